"""Times the strip-form ResnetBlock kernels (conv_strip.inc) on the UNet's block shapes (C2: B=32, dim=256).
Usage on the GPU box: python tools/strip_bench.py [bf16|f32] [B]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ladiffcodec_amd import lib as L  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
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
# (name, L, cin1, cin2, cout, with_res, count per UNet step)
shapes = [
    ("conv1 256->256 @1200", 1200, 256, 0, 256, 0, 10),
    ("conv1 512->256 @1200 (cat)", 1200, 256, 256, 256, 0, 3),
    ("conv2 256->256 +res 512 @1200", 1200, 256, 256, 256, 1, 3),
    ("conv1 256->256 @600", 600, 256, 0, 256, 0, 4),
    ("conv1 768->512 @600 (cat)", 600, 512, 256, 512, 0, 2),
    ("conv 512->512 @600", 600, 512, 0, 512, 0, 2),
    ("conv 512->512 @300", 300, 512, 0, 512, 0, 6),
    ("conv1 1024->512 @300 (cat)", 300, 512, 512, 512, 0, 2),
    ("conv 512->512 @150", 150, 512, 0, 512, 0, 4),
    ("conv 1024->1024 @150", 150, 1024, 0, 1024, 0, 2),
    ("conv1 1536->1024 @150 (cat)", 150, 1024, 512, 1024, 0, 2),
    ("conv 1024->1024 @75", 75, 1024, 0, 1024, 0, 10),
    ("conv1 2048->1024 @75 (cat)", 75, 1024, 1024, 1024, 0, 2),
    ("conv2 1024 +res 2048 @75", 75, 1024, 1024, 1024, 1, 2),
]
dt = L.LDC_BF16 if dtype == "bf16" else L.LDC_F32
tot_ms = tot_fl = 0.0
for name, Lx, c1, c2, co, res, cnt in shapes:
    ms = C.c_double()
    cin = c1 + c2
    if res:   # conv2: k3 over cout channels, res 1x1 over (c1 + c2)
        L.check(lib.ldc_strip_microbench(ctx, dt, B, Lx, co, 0, co, 0, 10, C.byref(ms)))   # plain part for reference
        base = ms.value
        # the res variant convolves (c1|c2) with k3 AND 1x1: a bound on the fused cost
        L.check(lib.ldc_strip_microbench(ctx, dt, B, Lx, c1, c2, co, 1, 10, C.byref(ms)))
        fl = 2.0 * B * Lx * co * cin * 4
        print(f"{name:34s} {ms.value * 1e3:8.1f} us  {fl / ms.value / 1e9:8.1f} TFLOP/s   (k3 {co}->{co} alone {base * 1e3:.1f} us)  x{cnt}")
    else:
        L.check(lib.ldc_strip_microbench(ctx, dt, B, Lx, c1, c2, co, 0, 10, C.byref(ms)))
        fl = 2.0 * B * Lx * co * cin * 3
        print(f"{name:34s} {ms.value * 1e3:8.1f} us  {fl / ms.value / 1e9:8.1f} TFLOP/s   x{cnt}")
    tot_ms += ms.value * cnt
    tot_fl += fl * cnt
print(f"weighted: {tot_ms:.3f} ms per step for these, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
lib.ldc_destroy(ctx)
