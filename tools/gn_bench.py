"""Times gn_apply on the UNet level shapes (half batch).  Usage: python tools/gn_bench.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ladiffcodec_amd import lib as L
lib = L.load()
cfg = L.LdcConfig(); cfg.compute_dtype = L.LDC_BF16
cfg.rep_dims, cfg.n_filters, cfg.n_residual_layers, cfg.lstm = 128, 32, 1, 2
cfg.n_enc_ratios = 1; cfg.enc_ratios[0] = 8; cfg.diff_dims = 256
ctx = C.c_void_p(); L.check(lib.ldc_create(C.byref(cfg), 0, C.byref(ctx)))
for B in (16, 32):
    for Lx, Cc in ((1200, 256), (600, 256), (300, 512), (150, 512), (75, 1024)):
        for res in (0, 1):
            ms = C.c_double()
            L.check(lib.ldc_gn_microbench(ctx, L.LDC_BF16, B, Lx, Cc, res, 50, C.byref(ms)))
            gb = B * Lx * Cc * 2 * (2 + res) / 1e9
            print(f"B={B} L={Lx} C={Cc} res={res}: {ms.value*1e3:7.1f} us  {gb/ms.value*1e3:8.1f} GB/s")
lib.ldc_destroy(ctx)
