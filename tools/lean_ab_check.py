"""Round 6 diagnostic: ldc_conv_compare in its kernel-A/B modes (tile_cfg + 100: conv_fast vs conv_lean; + 200: conv_fast twice; + 300: conv_lean twice)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ladiffcodec_amd import lib as L  # noqa: E402
from gpu_common import engine  # noqa: E402

shapes = [(1200, 256, 0, 256, 3, 1, 0), (600, 512, 256, 512, 3, 1, 0), (600, 768, 0, 512, 3, 1, 0), (600, 512, 0, 512, 3, 1, 0), (150, 1024, 512, 1024, 3, 1, 0),
          (300, 512, 512, 512, 1, 1, 0), (1200, 256, 0, 384, 1, 1, 0), (75, 1024, 0, 1024, 3, 1, 0)]
for dtype in ("f32", "bf16"):
    e = engine("r84", dtype)
    dt = L.LDC_F32 if dtype == "f32" else L.LDC_BF16
    for sh in shapes:
        for B in (3, 16):
            row = []
            for base in (199, 299, 399):       # (tile_cfg % 100 == 99: the launcher's own tile choice)
                for cfg in (base, base - 99, base - 98):
                    d, m, r = C.c_double(), C.c_double(), C.c_double()
                    L.check(e.lib.ldc_conv_compare(e._ctx, dt, B, *sh, cfg, 0, 1 if sh[4] == 1 else 0, 0, C.byref(d), C.byref(m), C.byref(r)))
                    row.append("%.2e" % d.value)
            print(dtype, sh, "B", B, "| fast-vs-lean (auto, 64x64, 128x64):", *row[0:3], "| fast twice:", *row[3:6], "| lean twice:", *row[6:9])
