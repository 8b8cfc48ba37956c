"""Pins oracle/bitstream_oracle.py to the reference's bit packer / arithmetic coder outputs (tests/golden/bitstream.npz,
made by tools/gen_bitstream_golden.py from srcs/encodec/binary.py and srcs/quantization/ac.py).  Byte-for-byte."""
import random
import zlib

import numpy as np
import torch

from helpers import load_golden
from oracle import bitstream_oracle as BO


def test_bitpacker_matches_reference_selftest_vectors():
    g = load_golden("bitstream")
    for rep in range(4):
        bits = int(g[f"pack{rep}.bits"][0])
        tokens = g[f"pack{rep}.tokens"]
        data = BO.pack_bits(tokens.tolist(), bits)
        assert data == g[f"pack{rep}.bytes"].tobytes(), rep
        back = BO.unpack_bits(data, bits)
        assert len(tokens) <= len(back) <= len(tokens) + 8 // bits            # ghost values of the flush (binary.py:146)
        assert back[:len(tokens)] == tokens.tolist()
    assert BO.pack_bits([], 10) == b""


def test_code_frame_push_order_and_header():
    g = load_golden("bitstream")
    codes = g["frame.codes"]                                                   # [B, K, T]
    for b in range(codes.shape[0]):
        data = BO.pack_bits(BO.frame_code_order(codes[b]).tolist(), 10)
        assert data == g[f"frame.bytes{b}"].tobytes()
        assert len(data) == (6 * 120 * 10 + 7) // 8
        back = np.array(BO.unpack_bits(data, 10, 6 * 120)).reshape(120, 6).T
        assert np.array_equal(back, codes[b])
    meta = {"m": "ladiffcodec_16khz", "al": 38400, "nc": 6, "lm": False}
    hdr = BO.write_ecdc_header(meta)
    assert hdr == g["header.bytes"].tobytes()
    got, off = BO.read_ecdc_header(hdr + b"xyz")
    assert got == meta and off == len(hdr)


def test_quantized_cdf_and_arithmetic_coder_small_case():
    g = load_golden("bitstream")
    cdf = BO.build_stable_quantized_cdf(g["small.pdf"], 24)
    assert np.array_equal(cdf, g["small.cdf"])
    for b in range(3):
        data = BO.ac_encode(g["small.symbols"][b].tolist(), g["small.cdf"][b])
        assert data == g[f"small.bytes{b}"].tobytes(), b
        assert BO.ac_decode(data, 200, g["small.cdf"][b]) == g["small.symbols"][b].tolist()
    # static per-codebook tables, codes in compress.py's push order (t outer, k inner)
    cdf6 = BO.build_stable_quantized_cdf(g["static.pdf"], 24)
    assert np.array_equal(cdf6, g["static.cdf"])
    seq = BO.frame_code_order(g["static.codes"])
    rows = np.tile(np.arange(6), 120)
    data = BO.ac_encode(seq.tolist(), cdf6, rows)
    assert data == g["static.bytes"].tobytes()
    assert BO.ac_decode(data, len(seq), cdf6, rows) == seq.tolist()


def test_arithmetic_coder_reference_selftest_streams():
    """ac.py:263-288: four streams of 100-500 symbols over up to 4000 symbols, pdfs regenerated from the test's seeds."""
    g = load_golden("bitstream")
    torch.manual_seed(1234)
    random.seed(1234)
    for i in range(4):
        cardinality = random.randrange(4000)
        steps = random.randrange(100, 500)
        card, nsteps, crc_want = (int(v) for v in g[f"ac{i}.meta"])
        assert (cardinality, steps) == (card, nsteps)
        enc = BO.ArithmeticCoder()
        cdfs, crc = [], 0
        for step in range(steps):
            pdf = torch.softmax(torch.randn(cardinality), dim=0)
            cdf = BO.build_stable_quantized_cdf(pdf.numpy(), 24)
            crc = zlib.crc32(cdf.astype(np.int64).tobytes(), crc)
            symbol = torch.multinomial(pdf, 1).item()
            assert symbol == int(g[f"ac{i}.symbols"][step])
            enc.push(symbol, cdf)
            cdfs.append(cdf)
        assert crc == crc_want, "quantised cdfs differ from the reference's"
        data = enc.flush()
        assert data == g[f"ac{i}.bytes"].tobytes(), i
        dec = BO.ArithmeticDecoder(data)
        assert [dec.pull(c) for c in cdfs] == g[f"ac{i}.symbols"].tolist()
        assert dec.pull(np.zeros(1, np.int64)) is None
