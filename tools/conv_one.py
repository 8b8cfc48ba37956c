"""Launch one conv-GEMM shape repeatedly (for rocprofv3 --pmc).  Usage: python tools/conv_one.py L cin1 cin2 cout k stride ups [iters] [dtype]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ladiffcodec_amd import lib as L  # noqa: E402

Lx, c1, c2, co, k, st, ups = (int(v) for v in sys.argv[1:8])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
dtype = sys.argv[9] if len(sys.argv) > 9 else "bf16"
Bn = int(os.environ.get("LDC_B", "32"))
lib = L.load()
cfg = L.LdcConfig()
cfg.compute_dtype = L.LDC_BF16
cfg.rep_dims, cfg.n_filters, cfg.n_residual_layers, cfg.lstm = 128, 32, 1, 2
cfg.n_enc_ratios = 1
cfg.enc_ratios[0] = 8
cfg.diff_dims = 256
ctx = C.c_void_p()
L.check(lib.ldc_create(C.byref(cfg), 0, C.byref(ctx)))
ms = C.c_double()
L.check(lib.ldc_conv_microbench(ctx, L.LDC_BF16 if dtype == "bf16" else L.LDC_F32, Bn, Lx, c1, c2, co, k, st, ups, iters, C.byref(ms)))
fl = 2.0 * Bn * (2 * Lx if ups else (Lx // 2 if st == 2 else Lx)) * co * (c1 + c2) * k
print(f"{ms.value * 1e3:.1f} us  {fl / ms.value / 1e9:.1f} TFLOP/s")
lib.ldc_destroy(ctx)
