"""Accuracy / time probe of the training GEMMs (csrc/train_mm3.hip split-bf16 path, or convmm_kernel under LDC_TRAIN_FP32_MFMA=1)
against float64 conv1d on the CPU.  GPU box only:  python tools/mm3_probe.py"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ladiffcodec_amd import train as TR  # noqa: E402
from gpu_common import engine  # noqa: E402


def rel(a, b):
    return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))


def timed(fn, n=8):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


SHAPES = ((2, 256, 128, 1, 1, 0, 1200), (4, 512, 1024, 3, 1, 1, 75), (2, 256, 256, 4, 2, 1, 1200), (3, 130, 70, 7, 1, 3, 333),
          (32, 256, 256, 3, 1, 1, 1200), (32, 512, 256, 3, 1, 1, 1200), (32, 512, 256, 1, 1, 0, 1200), (32, 768, 512, 3, 1, 1, 600),
          (32, 512, 512, 3, 1, 1, 300), (32, 1536, 1024, 3, 1, 1, 150), (32, 1024, 1024, 3, 1, 1, 75), (32, 2048, 1024, 3, 1, 1, 75),
          (32, 512, 1024, 4, 2, 1, 150), (32, 256, 256, 7, 1, 3, 1200))


def main():
    e = engine("r84", "f32")
    gen = torch.Generator().manual_seed(3)
    print("path:", "fp32 MFMA" if os.environ.get("LDC_TRAIN_FP32_MFMA") else "split-bf16 MFMA")
    tot = [0.0, 0.0, 0.0]
    for B, cin, cout, k, st, pd, L in SHAPES:
        x = torch.randn(B, cin, L, generator=gen)
        w = torch.randn(cout, cin, k, generator=gen) * 0.05
        b = torch.randn(cout, generator=gen) * 0.1
        Lo = (L + 2 * pd - k) // st + 1
        dy = torch.randn(B, cout, Lo, generator=gen)
        check = B * cin * L * cout * k < 2e9
        cv = TR.Conv1d(e, w, b, st, pd)
        xg, dyg = x.cuda(), dy.cuda()
        got = cv.forward(xg)
        gr = cv.backward(dyg)
        t_f = timed(lambda: cv.forward(xg))
        t_b = timed(lambda: cv.backward(dyg))
        t_w = timed(lambda: cv.backward(dyg, want_dx=False))
        fl = 2.0 * B * cout * cin * k * Lo / 1e9          # GFLOP of one GEMM shape
        line = (f"B{B} {cin}->{cout} k{k} s{st} L{L}: fwd {t_f * 1e3:.0f} us {fl / t_f:.0f} TF | dx {(t_b - t_w) * 1e3:.0f} us {fl / max(t_b - t_w, 1e-6):.0f} TF"
                f" | dw+db {t_w * 1e3:.0f} us {fl / t_w:.0f} TF")
        if B == 32:
            tot[0] += t_f; tot[1] += t_b - t_w; tot[2] += t_w
        if check:
            xd, wd, bd = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
            y = F.conv1d(xd, wd, bd, stride=st, padding=pd)
            y.backward(dy.double())
            line += f"  err y {rel(got.cpu(), y.detach()):.1e} dx {rel(gr['dx'].cpu(), xd.grad):.1e} dw {rel(gr['dw'].cpu(), wd.grad):.1e} db {rel(gr['db'].cpu(), bd.grad):.1e}"
        print(line, flush=True)
    print(f"B = 32 shapes: fwd {tot[0]:.2f} ms, dx {tot[1]:.2f} ms, dw+db {tot[2]:.2f} ms")


if __name__ == "__main__":
    main()
