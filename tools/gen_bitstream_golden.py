"""tests/golden/bitstream.npz: outputs of the REFERENCE's bit packer and arithmetic coder (srcs/encodec/binary.py,
srcs/quantization/ac.py == srcs/encodec/quantization/ac.py), driven exactly as their own seeded self-tests drive them
(binary.py:125-149, ac.py:263-288) plus a codes-shaped frame in compress.py's push order.  Run in the build container only.
The fixture holds inputs and expected outputs; the big pdf tensors of ac.py's self-test are regenerated from its seeds
at test time (same torch build in this image), only their quantised-cdf checksums are stored."""
import importlib.util
import io
import os
import random
import sys
import types
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/srcs/encodec"


def load_reference():
    pkg = types.ModuleType("refenc"); pkg.__path__ = [REF]; sys.modules["refenc"] = pkg
    q = types.ModuleType("refenc.quantization"); q.__path__ = [REF + "/quantization"]; sys.modules["refenc.quantization"] = q
    out = {}
    for name, path in (("refenc.binary", REF + "/binary.py"), ("refenc.quantization.ac", REF + "/quantization/ac.py")):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        out[name.split(".")[-1]] = m
    return out["binary"], out["ac"]


def ac_selftest_streams(ac):
    """ac.py:263-288, verbatim driving (seeds 1234), collecting what the test produces."""
    torch.manual_seed(1234)
    random.seed(1234)
    cases = []
    for _ in range(4):
        cardinality = random.randrange(4000)
        steps = random.randrange(100, 500)
        fo = io.BytesIO()
        enc = ac.ArithmeticCoder(fo)
        symbols, crc = [], 0
        for _step in range(steps):
            pdf = torch.softmax(torch.randn(cardinality), dim=0)
            q_cdf = ac.build_stable_quantized_cdf(pdf, enc.total_range_bits)
            crc = zlib.crc32(q_cdf.numpy().astype(np.int64).tobytes(), crc)
            symbol = torch.multinomial(pdf, 1).item()
            symbols.append(symbol)
            enc.push(symbol, q_cdf)
        enc.flush()
        cases.append((cardinality, steps, np.array(symbols, np.int64), np.frombuffer(fo.getvalue(), np.uint8), crc))
    return cases


def main():
    binary, ac = load_reference()
    binary.test(); ac.test()                      # the reference's own self-tests pass in this container
    out = {}
    # --- BitPacker: the self-test's cases (binary.py:125-132) + a [K=6, T=120] code frame at 10 bits
    torch.manual_seed(1234)
    for rep in range(4):
        length = torch.randint(10, 2_000, (1,)).item()
        bits = torch.randint(1, 16, (1,)).item()
        tokens = torch.randint(2 ** bits, (length,)).tolist()
        fo = io.BytesIO()
        p = binary.BitPacker(bits, fo)
        for t in tokens:
            p.push(t)
        p.flush()
        out[f"pack{rep}.tokens"] = np.array(tokens, np.int64)
        out[f"pack{rep}.bits"] = np.array([bits], np.int64)
        out[f"pack{rep}.bytes"] = np.frombuffer(fo.getvalue(), np.uint8)
    g = torch.Generator().manual_seed(7)
    codes = torch.randint(0, 1024, (3, 6, 120), generator=g)            # [B, K, T]
    out["frame.codes"] = codes.numpy().astype(np.int64)
    for b in range(3):
        fo = io.BytesIO()
        p = binary.BitPacker(10, fo)
        for t in range(120):                                               # compress.py:74-84
            for value in codes[b, :, t].tolist():
                p.push(value)
        p.flush()
        out[f"frame.bytes{b}"] = np.frombuffer(fo.getvalue(), np.uint8)
    fo = io.BytesIO()
    binary.write_ecdc_header(fo, {"m": "ladiffcodec_16khz", "al": 38400, "nc": 6, "lm": False})
    out["header.bytes"] = np.frombuffer(fo.getvalue(), np.uint8)
    # --- arithmetic coder: the self-test's four streams (pdfs regenerated from the seeds at test time)
    for i, (card, steps, symbols, data, crc) in enumerate(ac_selftest_streams(ac)):
        out[f"ac{i}.meta"] = np.array([card, steps, crc], np.int64)
        out[f"ac{i}.symbols"] = symbols
        out[f"ac{i}.bytes"] = data
    # --- a small fully stored case: 3 streams x 200 steps over 64 symbols, per-step pdfs
    g = torch.Generator().manual_seed(99)
    pdf = torch.softmax(torch.randn(3, 200, 64, generator=g) * 2.0, dim=-1)
    cdf = torch.stack([torch.stack([ac.build_stable_quantized_cdf(pdf[b, s], 24) for s in range(200)]) for b in range(3)])
    sym = torch.multinomial(pdf.reshape(-1, 64), 1, generator=g).reshape(3, 200)
    out["small.pdf"] = pdf.numpy().astype(np.float32)
    out["small.cdf"] = cdf.numpy().astype(np.int64)
    out["small.symbols"] = sym.numpy().astype(np.int64)
    for b in range(3):
        fo = io.BytesIO()
        enc = ac.ArithmeticCoder(fo)
        for s in range(200):
            enc.push(int(sym[b, s]), cdf[b, s])
        enc.flush()
        out[f"small.bytes{b}"] = np.frombuffer(fo.getvalue(), np.uint8)
        dec = ac.ArithmeticDecoder(io.BytesIO(fo.getvalue()))
        assert [dec.pull(cdf[b, s]) for s in range(200)] == sym[b].tolist()
    # --- static per-codebook tables (zero-order model of RVQ codes): 6 codebooks x 1024, codes [K, T] in push order
    g = torch.Generator().manual_seed(5)
    pdf6 = torch.softmax(torch.randn(6, 1024, generator=g) * 1.5, dim=-1)
    cdf6 = torch.stack([ac.build_stable_quantized_cdf(pdf6[k], 24) for k in range(6)])
    codes6 = torch.stack([torch.multinomial(pdf6[k], 120, replacement=True, generator=g) for k in range(6)])   # [K, T]
    fo = io.BytesIO()
    enc = ac.ArithmeticCoder(fo)
    for t in range(120):
        for k in range(6):
            enc.push(int(codes6[k, t]), cdf6[k])
    enc.flush()
    out["static.pdf"] = pdf6.numpy().astype(np.float32)
    out["static.cdf"] = cdf6.numpy().astype(np.int64)
    out["static.codes"] = codes6.numpy().astype(np.int64)
    out["static.bytes"] = np.frombuffer(fo.getvalue(), np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bitstream.npz"), **out)
    print("bitstream.npz:", {k: v.shape for k, v in out.items() if k.endswith("bytes") or k.endswith("bytes0")})


if __name__ == "__main__":
    main()
