"""Where a tile's time goes inside the XCD-team chain kernel (LDC_CHAIN_STAMPS=1): per chain and conv, the mean s_memtime cycles a tile
spends pulling its ticket, reading its descriptor and waiting for its producers, in the prologue (first copies + tables), in the K loop,
in the epilogue (GroupNorm exchange, stores, flag), and how busy the team's workgroups were.  Workload: one UNet pass at the bench grid.
    LDC_CHAIN_STAMPS=1 python tools/chain_stamps.py [n_steps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LDC_CHAIN_STAMPS", "1")
from ladiffcodec_amd import lib as L, synth  # noqa: E402
from ladiffcodec_amd.model import Engine  # noqa: E402
from ladiffcodec_amd.spec import CodecConfig, UnetConfig  # noqa: E402


def main():
    B, T = int(os.environ.get("LDC_B", "32")), 38400
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
    u = UnetConfig(dim=256, upsampling_ratios=(5, 2), unet_scale_cond=True)
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
    sd = synth.ladiff_state_dict(mc, u, seed=1)
    e = Engine(mc, u, cc, dtype="bf16", device=0, noise_seed=4321)
    e.load_state_dict(L.MODEL_MAIN, {k: v for k, v in sd.items() if not k.startswith("diffusion.model.")})
    e.load_state_dict(L.MODEL_COND, synth.codec_state_dict(cc, seed=0))
    e.finalize(strict=True)
    wav = torch.from_numpy(synth.synthetic_wav(B, T, seed=1234)).cuda()
    out = e.decode(wav, N, noise=None, per_item=True)
    torch.cuda.synchronize()
    w = out["wav"] if isinstance(out, dict) else out
    assert torch.isfinite(w).all()
    lib = L.load()
    n_ch = C.c_int(0)
    L.check(lib.ldc_chain_stamps(e._ctx, -1, None, 0, None, C.byref(n_ch), None, 0))
    print(f"# chain stamps, B = {B}, last of {N} UNet passes; times are s_memtime ticks (shader cycles)" )
    for ch in range(n_ch.value):
        meta = (C.c_int * (2 + 8 * 17))()
        info = C.create_string_buffer(4096)
        stride = 4096
        buf = np.zeros(8 * stride * 12, np.uint64)
        L.check(lib.ldc_chain_stamps(e._ctx, ch, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size, meta, C.byref(n_ch), info, 4096))
        stride, nconv = meta[0], meta[1]
        first = np.array(meta[2:2 + 8 * 17]).reshape(8, 17)
        st = buf.reshape(8, stride, 12).astype(np.int64)
        names = info.value.decode().split(": ", 1)[1].split(" ") if ": " in info.value.decode() else []
        print(f"\n## chain {ch}: {nconv} convs")
        print("| conv | tiles/team | ticket | desc+dep wait | prologue | K loop | epilogue+flag | tile total | team span | slots busy |")
        print("|---|---|---|---|---|---|---|---|---|---|")
        t_lo, t_hi = [], []
        for c in range(nconv):
            rows = []
            spans, busy = [], []
            for x in range(8):
                a, b = first[x, c], first[x, c + 1]
                if b <= a:
                    continue
                s = st[x, a:b]
                s = s[s[:, 10] > 0]
                if not len(s):
                    continue
                rows.append(s)
                span = s[:, 10].max() - s[:, 8].min()
                spans.append(span)
                busy.append((s[:, 10] - s[:, 8]).sum() / max(1, span))
                t_lo.append(s[:, 8].min()); t_hi.append(s[:, 10].max())
            if not rows:
                continue
            s = np.concatenate(rows)
            full = s[s[:, 3] > 0]       # (split-K slices that left early have no end stamp)
            tk = (s[:, 9] - s[:, 8]).mean()
            dep = (s[:, 7] - s[:, 9]).mean()
            pro = (s[:, 1] - s[:, 7]).mean()
            loop = (s[:, 2] - s[:, 1]).mean()
            epi = (full[:, 10] - full[:, 2]).mean() if len(full) else 0
            tot = (s[:, 10] - s[:, 8]).mean()
            nm = names[c] if c < len(names) else str(c)
            print(f"| `{nm}` | {len(s) // 8} | {tk:.0f} | {dep:.0f} | {pro:.0f} | {loop:.0f} | {epi:.0f} | {tot:.0f} | {np.mean(spans):.0f} | {np.mean(busy):.1f} |")
        if t_lo:
            print(f"\nchain span (first ticket pull to last tile end, mean over teams): {(np.array(t_hi).max() - np.array(t_lo).min()):.0f} ticks")


if __name__ == "__main__":
    main()
