"""Tile-shape sweep of the pipelined conv-GEMM on the UNet's layer shapes (random operands), plus the fast-vs-generic
self-check of every shape.   python tools/tile_sweep.py [B ...]      (GPU box)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ladiffcodec_amd import lib as L  # noqa: E402

lib = L.load()
SHAPES = [  # (name, L, cin1, cin2, cout, k, stride, ups, count per UNet step)
    ("k3 256->256 @1200", 1200, 256, 0, 256, 3, 1, 0, 10),
    ("k3 512->256 @1200 (cat)", 1200, 256, 256, 256, 3, 1, 0, 3),
    ("k3 256->256 @600", 600, 256, 0, 256, 3, 1, 0, 4),
    ("k3 512->512 @600", 600, 512, 0, 512, 3, 1, 0, 2),
    ("k3 768->512 @600 (cat)", 600, 512, 256, 512, 3, 1, 0, 2),
    ("k3 512->512 @300", 300, 512, 0, 512, 3, 1, 0, 6),
    ("k3 1024->512 @300 (cat)", 300, 512, 512, 512, 3, 1, 0, 2),
    ("k3 512->512 @150", 150, 512, 0, 512, 3, 1, 0, 4),
    ("k3 1024->1024 @150", 150, 1024, 0, 1024, 3, 1, 0, 2),
    ("k3 1536->1024 @150 (cat)", 150, 1024, 512, 1024, 3, 1, 0, 2),
    ("k3 1024->1024 @75", 75, 1024, 0, 1024, 3, 1, 0, 11),
    ("k3 2048->1024 @75 (cat)", 75, 1024, 1024, 1024, 3, 1, 0, 2),
    ("k1 256->384 @1200 (qkv)", 1200, 256, 0, 384, 1, 1, 0, 2),
    ("k1 512->256 @1200 (res)", 1200, 256, 256, 256, 1, 1, 0, 3),
    ("k1 768->512 @600 (res)", 600, 512, 256, 512, 1, 1, 0, 2),
    ("k1 1024->512 @300 (res)", 300, 512, 512, 512, 1, 1, 0, 2),
    ("k1 1536->1024 @150 (res)", 150, 1024, 512, 1024, 1, 1, 0, 2),
    ("k1 2048->1024 @75 (res)", 75, 1024, 1024, 1024, 1, 1, 0, 2),
    ("up k3 512->256 @600->1200", 600, 512, 0, 256, 3, 1, 1, 1),
    ("up k3 1024->1024 @75->150", 75, 1024, 0, 1024, 3, 1, 1, 1),
]
CFGS = [-1, 0, 1, 2]


def ctx_for(cfg, tall=1):
    if cfg >= 0:
        os.environ["LDC_TILE_CFG"] = str(cfg)
    else:
        os.environ.pop("LDC_TILE_CFG", None)
    c = L.LdcConfig()
    c.compute_dtype = L.LDC_BF16
    c.rep_dims, c.n_filters, c.n_residual_layers, c.lstm = 128, 32, 1, 2
    c.n_enc_ratios = 2
    c.enc_ratios[0], c.enc_ratios[1] = 8, 4
    c.diff_dims = 256
    c.n_upsampling_ratios = 2
    c.upsampling_ratios[0], c.upsampling_ratios[1] = 5, 2
    ctx = C.c_void_p()
    L.check(lib.ldc_create(C.byref(c), 0, C.byref(ctx)))
    return ctx


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [16, 32]
    ctxs = {cfg: ctx_for(cfg) for cfg in CFGS}
    cols = CFGS
    print("self-check (fast kernel with the forced tile vs generic kernel; bf16 outputs, relative to max |y|):")
    worst = 0.0
    for name, Lx, c1, c2, co, k, st, ups, cnt in SHAPES[::3] + [SHAPES[12], SHAPES[-1]]:
        for cfg in CFGS:
            d, m, r = C.c_double(), C.c_double(), C.c_double()
            L.check(lib.ldc_conv_compare(ctxs[-1], L.LDC_BF16, 3, Lx, c1, c2, co, k, st, ups, cfg, 1 if co % 8 == 0 and k == 3 else 0,
                                         1 if k == 1 else 0, 1 if cfg == 2 else 0, C.byref(d), C.byref(m), C.byref(r)))
            rel = d.value / (m.value + 1e-30)
            worst = max(worst, rel, r.value)
            flag = "" if rel < 1.2e-2 and r.value < 1e-3 else "   <-- MISMATCH"
            print(f"  {name:30s} cfg {cfg:2d}: out {rel:.2e} stats {r.value:.2e}{flag}")
    print(f"worst {worst:.2e}")
    for B in Bs:
        print(f"\nB = {B}: us per launch (TFLOP/s); columns: -1 = launcher's choice, then forced tiles 0 = 64x64, 1 = 128x64, 2 = 128x128")
        tot = {c: 0.0 for c in cols}
        best_tot = 0.0
        fl_tot = 0.0
        for name, Lx, c1, c2, co, k, st, ups, cnt in SHAPES:
            Lo = 2 * Lx if ups else (Lx // 2 if st == 2 else Lx)
            fl = 2.0 * B * Lo * co * (c1 + c2) * k
            row = []
            for cfg in cols:
                ms = C.c_double()
                L.check(lib.ldc_conv_microbench(ctxs[cfg], L.LDC_BF16, B, Lx, c1, c2, co, k, st, ups, 20, C.byref(ms)))
                row.append(ms.value * 1e3)
                tot[cfg] += ms.value * cnt
            best = min(row[1:])
            best_tot += best * 1e-3 * cnt
            fl_tot += fl * cnt
            print(f"  {name:30s} " + " ".join(f"{u:6.1f}" for u in row) + f"   best cfg {CFGS[row[1:].index(best) + 1]:2d} {fl / best / 1e6:6.0f} TF  (x{cnt})")
        print("  weighted ms/step: " + " ".join(f"{c}:{tot[c]:.3f}" for c in cols) + f"  best-of: {best_tot:.3f} ms = {fl_tot / best_tot / 1e9:.0f} TFLOP/s")
    for c in ctxs.values():
        lib.ldc_destroy(c)


if __name__ == "__main__":
    main()
