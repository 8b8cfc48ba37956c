"""Times the conv-GEMM kernel on the UNet's layer shapes (C2: B=32, L=1200, dim=256).
Usage on the GPU box: [LDC_CONV_STAGES=n] python tools/conv_bench.py [bf16|f32]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ladiffcodec_amd import lib as L  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
lib = L.load()
cfg = L.LdcConfig()
cfg.compute_dtype = L.LDC_BF16
cfg.rep_dims, cfg.n_filters, cfg.n_residual_layers, cfg.lstm = 128, 32, 1, 2
cfg.n_enc_ratios = 2
cfg.enc_ratios[0], cfg.enc_ratios[1] = 8, 4
cfg.diff_dims = 256
cfg.n_upsampling_ratios = 2
cfg.upsampling_ratios[0], cfg.upsampling_ratios[1] = 5, 2
ctx = C.c_void_p()
L.check(lib.ldc_create(C.byref(cfg), 0, C.byref(ctx)))
B = 32
# (name, L, cin1, cin2, cout, k, stride, ups, count per UNet step)
shapes = [
    ("init k7 256->256 @1200", 1200, 128, 128, 256, 7, 1, 0, 1),
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
    ("k1 128->256 @1200 (out)", 1200, 128, 0, 256, 1, 1, 0, 2),
    ("k1 2048->1024 @75 (res)", 75, 1024, 1024, 1024, 1, 1, 0, 2),
    ("k4s2 256->256 @1200", 1200, 256, 0, 256, 4, 2, 0, 1),
    ("up k3 512->256 @600->1200", 600, 512, 0, 256, 3, 1, 1, 1),
]
dt = L.LDC_BF16 if dtype == "bf16" else L.LDC_F32
tot_ms = tot_fl = 0.0
print(f"dtype={dtype} stages={os.environ.get('LDC_CONV_STAGES', 'default')} v1={'LDC_CONV_V1' in os.environ}")
for name, Lx, c1, c2, co, k, st, ups, cnt in shapes:
    ms = C.c_double()
    L.check(lib.ldc_conv_microbench(ctx, dt, B, Lx, c1, c2, co, k, st, ups, 20, C.byref(ms)))
    Lo = 2 * Lx if ups else (Lx // 2 if st == 2 else Lx)
    fl = 2.0 * B * Lo * co * (c1 + c2) * k
    print(f"{name:34s} {ms.value * 1e3:8.1f} us  {fl / ms.value / 1e9:8.1f} TFLOP/s   x{cnt}")
    tot_ms += ms.value * cnt
    tot_fl += fl * cnt
print(f"weighted: {tot_ms:.3f} ms per (partial) step, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
lib.ldc_destroy(ctx)
