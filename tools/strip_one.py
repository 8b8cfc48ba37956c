"""One strip-kernel shape, for profiling: python tools/strip_one.py L cin1 cin2 cout with_res [B] [iters]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ladiffcodec_amd import lib as L
lib = L.load()
cfg = L.LdcConfig()
cfg.compute_dtype = L.LDC_BF16
cfg.rep_dims, cfg.n_filters, cfg.n_residual_layers, cfg.lstm = 128, 32, 1, 2
cfg.n_enc_ratios = 2; cfg.enc_ratios[0], cfg.enc_ratios[1] = 8, 4
cfg.diff_dims = 256; cfg.n_upsampling_ratios = 2; cfg.upsampling_ratios[0], cfg.upsampling_ratios[1] = 5, 2
ctx = C.c_void_p()
L.check(lib.ldc_create(C.byref(cfg), 0, C.byref(ctx)))
Lx, c1, c2, co, res = (int(v) for v in sys.argv[1:6])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 32
it = int(sys.argv[7]) if len(sys.argv) > 7 else 10
ms = C.c_double()
L.check(lib.ldc_strip_microbench(ctx, L.LDC_BF16, B, Lx, c1, c2, co, res, it, C.byref(ms)))
print(f"L={Lx} {c1}+{c2}->{co} res={res} B={B}: {ms.value*1e3:.1f} us")
lib.ldc_destroy(ctx)
