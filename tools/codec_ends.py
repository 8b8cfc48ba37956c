"""The codec front end (cond encoder -> RVQ) and back end (main decoder) of one batch part, for rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/codec_ends.py [B] [c1]
    (c1: the cond codec's own round trip, encode -> RVQ -> its decoder, BASELINE configs[0])
    python tools/codec_ends.py --summarise OUT      (dispatch order, grid, duration of the last repetition)"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarise(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last repetition: from the last conv_cin1 launch on
    last = max(i for i, r in enumerate(rows) if "conv_cin1" in r["Kernel_Name"])
    tot = 0.0
    print("| # | kernel | grid | us |\n|---|---|---|---|")
    for i, r in enumerate(rows[last:]):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        tot += us
        name = r["Kernel_Name"].replace("ldc::", "").split("(")[0][:70]
        print(f"| {i} | `{name}` | {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} | {us:.1f} |")
    print(f"\nsum of kernel time: {tot:.0f} us")


def main():
    import torch
    from ladiffcodec_amd import lib as L, synth
    from ladiffcodec_amd.model import Engine
    from ladiffcodec_amd.spec import CodecConfig, UnetConfig
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
    u = UnetConfig(dim=32, upsampling_ratios=(5, 2), unet_scale_cond=True)
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
    sd = synth.ladiff_state_dict(mc, u, seed=1)
    e = Engine(mc, u, cc, dtype="f32", device=0)
    e.load_state_dict(L.MODEL_MAIN, {k: v for k, v in sd.items() if not k.startswith("diffusion.model.")})
    e.load_state_dict(L.MODEL_COND, synth.codec_state_dict(cc, seed=0))
    e.finalize(strict=True)
    wav = torch.from_numpy(synth.synthetic_wav(B, 38400, seed=3)).cuda()
    z = torch.randn(B, 128, 1200, device="cuda") * 0.1
    c1 = len(sys.argv) > 2 and sys.argv[2] == "c1"
    for _ in range(3):
        cond = e.get_cond(wav)
        if c1:
            e.decode_latents(L.MODEL_COND, cond)
        else:
            e.decode_latents(L.MODEL_MAIN, z)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        main()
