"""Which dies do the workgroups of a CU-masked stream land on?  (hipExtStreamCreateWithCUMask; MI355X: 8 XCCs x 32 CUs)
python tools/xcd_mask_probe.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ladiffcodec_amd import lib as L  # noqa: E402

lib = L.load()
cfg = L.LdcConfig()
cfg.compute_dtype = L.LDC_BF16
cfg.rep_dims, cfg.n_filters, cfg.n_residual_layers, cfg.lstm = 128, 32, 1, 2
cfg.n_enc_ratios = 1
cfg.enc_ratios[0] = 8
cfg.diff_dims = 256
ctx = C.c_void_p()
L.check(lib.ldc_create(C.byref(cfg), 0, C.byref(ctx)))


def census(bits, wgs=2048):
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    hist = (C.c_int * 16)()
    L.check(lib.ldc_xcc_census(ctx, arr if bits is not None else None, 8 if bits is not None else 0, wgs, hist))
    return list(hist)[:8]


hist = (C.c_int * 16)()
L.check(lib.ldc_xcc_census(ctx, None, 0, 2048, hist))
print("no mask          :", list(hist)[:8])
print("bits 0..31       :", census(range(0, 32)))
print("bits 32..63      :", census(range(32, 64)))
print("bits 0..7        :", census(range(0, 8)))
print("bits k*8 (k<32)  :", census(range(0, 256, 8)))
print("bits k*8+1       :", census(range(1, 256, 8)))
print("bits k*8+3       :", census(range(3, 256, 8)))
print("bits 0,1 only    :", census([0, 1]))
print("bits 0..127      :", census(range(0, 128)))
lib.ldc_destroy(ctx)
