"""One-shot GPU diagnosis: runs every stage against the golden vectors / oracle and prints a table
without stopping at the first mismatch.  Usage on the GPU box:  python tools/gpu_diag.py [f32|bf16]"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import CASES, COND_CFG, T, load_golden, main_sd_np, cond_sd_np  # noqa: E402
from gpu_common import engine, rel  # noqa: E402
from ladiffcodec_amd import lib as L, synth  # noqa: E402
from oracle import ldc_oracle as O  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
rows = []


def report(name, err, tol):
    ok = err < tol
    rows.append((name, err, tol, ok))
    print(f"{'OK  ' if ok else 'FAIL'} {name:40s} err={err:.3e} tol={tol:.1e}", flush=True)


def guard(fn):
    try:
        fn()
    except Exception as e:  # noqa: BLE001
        traceback.print_exc()
        rows.append((fn.__name__, float('nan'), 0, False))
        print("FAIL", fn.__name__, repr(e), flush=True)


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def prim():
    g = load_golden("primitives")
    e = engine("r84", "f32")
    for n in sorted({k.split(".")[0] for k in g if k.startswith("c_")}):
        k, s, d, causal = (int(v) for v in g[n + ".cfg"])
        if g[n + ".x"].shape[-1] <= (k - 1) * d:
            continue
        w = O.fold_weight_norm(T(g[n + ".g"]), T(g[n + ".v"])).numpy()
        y = e.sconv1d(cu(g[n + ".x"]), w, g[n + ".b"], stride=s, dilation=d, causal=bool(causal))
        report("sconv1d." + n, rel(y.cpu().numpy(), g[n + ".y"]), 1e-5)
    for n in sorted({k.split(".")[0] for k in g if k.startswith("t_")}):
        k, s, d, causal = (int(v) for v in g[n + ".cfg"])
        w = O.fold_weight_norm(T(g[n + ".g"]), T(g[n + ".v"])).numpy() if (n + ".g") in g else g[n + ".w"]
        y = e.sconvtr1d(cu(g[n + ".x"]), w, g[n + ".b"], s, bool(causal))
        report("sconvtr1d." + n, rel(y.cpu().numpy(), g[n + ".y"]), 1e-5)
    ws = []
    for layer in range(2):
        for nm in ("weight_ih_l", "weight_hh_l", "bias_ih_l", "bias_hh_l"):
            ws.append(g[f"lstm.sd.lstm.{nm}{layer}"])
    y = e.slstm(cu(g["lstm.x"]), ws, 2)
    report("slstm.h16", rel(y.cpu().numpy(), g["lstm.y"]), 1e-5)


def codec():
    g = load_golden("codec_c1")
    e = engine("r84", "f32")
    wav = cu(g["wav"])
    z = e.encode(L.MODEL_COND, wav)
    report("cond.encoder z", rel(z.cpu().numpy(), g["z"]), 1e-4)
    q, codes = e.rvq(cu(g["z"]), 6)
    sd = synth.to_torch(cond_sd_np())
    _, _, margins = O.rvq_forward(sd, T(g["z"]), 6)
    safe = margins.numpy() > 1e-3
    mism = int((codes.cpu().numpy()[safe] != g["codes"][safe]).sum())
    report("rvq codes mismatches (safe margin)", float(mism), 0.5)
    report("rvq quantized", rel(q.cpu().numpy(), g["quantized"]), 1e-5)
    cond, codes2 = e.get_cond(wav, return_codes=True)
    report("get_cond codes mismatches", float((codes2.cpu().numpy()[safe] != g["codes"][safe]).sum()), 0.5)
    report("get_cond quantized", rel(cond.cpu().numpy(), g["quantized"]), 1e-4)
    dec = e.decode_latents(L.MODEL_COND, cu(g["quantized"]))
    report("cond.decoder", rel(dec.cpu().numpy(), g["decoded"]), 1e-4)
    report("rvq_decode", rel(e.rvq_decode(cu(g["codes"])).cpu().numpy(), g["quantized"]), 1e-6)


def unet(tag):
    g = load_golden("ladiff_" + tag)
    mc, u, _ = CASES[tag]
    e = engine(tag, dtype)
    tol = 2e-4 if dtype == "f32" else 6e-2
    cond, x = cu(g["cond"]), cu(g["x"])
    sd = synth.to_torch(main_sd_np(tag))
    taps = {}
    O.unet_forward(sd, u, T(g["x"]), torch.full((2,), 37, dtype=torch.long), T(g["cond"]), taps=taps)
    eps = e.unet_forward(x, 37, cond)
    for name in ["cond_proc", "init"] + [f"down{i}" for i in range(5)] + ["mid"] + [f"up{i}" for i in range(5)]:
        ref = taps[name].numpy()
        got = e.debug_tap(name, ref.shape).cpu().numpy()
        report(f"{tag}.tap.{name}", rel(got, ref), tol)
    report(f"{tag}.eps_t37", rel(eps.cpu().numpy(), g["eps_t37"]), tol)
    report(f"{tag}.eps_t0", rel(e.unet_forward(x, 0, cond).cpu().numpy(), g["eps_t0"]), tol)
    report(f"{tag}.img_up", rel(e.cond_upsample(cond, 0).cpu().numpy(), g["img_up"]), 1e-5)
    report(f"{tag}.img0", rel(e.cond_upsample(cond, 1).cpu().numpy(), g["img0"]), 1e-5)
    n = int(g["meta"][2])
    noises = cu(g["noises"])
    report(f"{tag}.p_sample_t5", rel(e.p_sample(x, 5, cond, noises[n - 1]).cpu().numpy(), g["p_sample_t5"]), tol)
    lat = e.denoise(cu(g["img0"]), cond, n, noises)
    report(f"{tag}.latents(chain {n})", rel(lat.cpu().numpy(), g["latents"]), tol * 3)
    lat5 = e.denoise(cu(g["img0"]), cond, n, noises)
    report(f"{tag}.latents replay == first", rel(lat5.cpu().numpy(), lat.cpu().numpy()), 1e-6 if dtype == "f32" else 1e-2)
    wav = e.decode_latents(L.MODEL_MAIN, cu(g["latents"]))
    report(f"{tag}.decoder", rel(wav.cpu().numpy(), g["wav_raw"]), 1e-4)
    report(f"{tag}.out_norm", rel(e.output_normalise(cu(g["wav_raw"])).cpu().numpy(), g["wav_out"]), 1e-5)
    out = e.decode(cu(g["wav"]), n, noises, per_item=False, want_stages=True)
    report(f"{tag}.e2e.cond", rel(out["cond"].cpu().numpy(), g["cond"]), 1e-4)
    report(f"{tag}.e2e.latents", rel(out["latents"].cpu().numpy(), g["latents"]), tol * 3)
    report(f"{tag}.e2e.wav", rel(out["wav"].cpu().numpy(), g["wav_out"]), 5e-3 if dtype == "f32" else 0.3)


print("device:", torch.cuda.get_device_name(0), "| lib:", L.load().ldc_version().decode(), "| dtype:", dtype, flush=True)
guard(prim)
guard(codec)
guard(lambda: unet("r84"))
guard(lambda: unet("r8"))
bad = [r for r in rows if not r[3]]
print(f"\n{len(rows) - len(bad)}/{len(rows)} checks passed")
sys.exit(1 if bad else 0)
