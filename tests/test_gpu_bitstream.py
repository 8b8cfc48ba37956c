"""GPU bit-stream layer through the C ABI against the reference's outputs (tests/golden/bitstream.npz) -- byte for byte
-- and, at full batch sizes, through round trips and the oracle on fresh seeded inputs."""
import io
import random
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ladiffcodec_amd import bitstream as BS  # noqa: E402
from oracle import bitstream_oracle as BO  # noqa: E402
from helpers import load_golden  # noqa: E402
from gpu_common import engine  # noqa: E402


def bs():
    return BS.Bitstream(engine("r84", "f32"))


def test_pack_unpack_reference_vectors():
    g = load_golden("bitstream")
    b = bs()
    codes = torch.from_numpy(g["frame.codes"]).permute(1, 0, 2).contiguous()          # [K, B, T]
    packed = b.pack_codes(codes.cuda(), 10).cpu().numpy()
    for i in range(3):
        assert packed[i].tobytes() == g[f"frame.bytes{i}"].tobytes()
    assert torch.equal(b.unpack_codes(torch.from_numpy(packed).cuda(), 6, 120, 10).cpu(), codes)
    # the reference self-test's widths (1..15 bits, 10..2000 values): a [1, 1, n] "frame"
    for rep in range(4):
        bits = int(g[f"pack{rep}.bits"][0])
        tok = torch.from_numpy(g[f"pack{rep}.tokens"]).reshape(1, 1, -1)
        got = b.pack_codes(tok.cuda(), bits).cpu().numpy()[0].tobytes()
        assert got == g[f"pack{rep}.bytes"].tobytes(), (rep, bits)
        assert torch.equal(b.unpack_codes(torch.frombuffer(bytearray(got), dtype=torch.uint8).reshape(1, -1).cuda(), 1, tok.shape[2], bits).cpu(), tok)
    with pytest.raises(EOFError):                                                        # truncated stream
        b.unpack_codes(torch.from_numpy(packed[:, :100].copy()).cuda(), 6, 120, 10)


def test_quantized_cdf_and_range_coder_reference_vectors():
    g = load_golden("bitstream")
    b = bs()
    cdf = b.build_cdf(torch.from_numpy(g["small.pdf"]).cuda())
    assert np.array_equal(cdf.cpu().numpy().astype(np.int64), g["small.cdf"])
    streams = b.ac_encode(torch.from_numpy(g["small.symbols"]).cuda(), cdf)
    for i in range(3):
        assert streams[i] == g[f"small.bytes{i}"].tobytes(), i
    back = b.ac_decode(streams, 200, cdf)
    assert np.array_equal(back.cpu().numpy(), g["small.symbols"])
    with pytest.raises(EOFError):
        b.ac_decode([s[:20] for s in streams], 200, cdf)
    # static per-codebook tables in compress.py's push order
    cdf6 = b.build_cdf(torch.from_numpy(g["static.pdf"]).cuda())
    assert np.array_equal(cdf6.cpu().numpy().astype(np.int64), g["static.cdf"])
    seq = torch.from_numpy(BO.frame_code_order(g["static.codes"])).reshape(1, -1)
    stream = b.ac_encode(seq.cuda(), cdf6, static=True)
    assert stream[0] == g["static.bytes"].tobytes()
    assert np.array_equal(b.ac_decode(stream, seq.shape[1], cdf6, static=True).cpu().numpy(), seq.numpy())


def test_range_coder_reference_selftest_streams():
    """ac.py:263-288 on the GPU: four streams, up to 4000 symbols wide, pdfs regenerated from the test's seeds."""
    g = load_golden("bitstream")
    b = bs()
    torch.manual_seed(1234)
    random.seed(1234)
    for i in range(4):
        cardinality = random.randrange(4000)
        steps = random.randrange(100, 500)
        pdfs = []
        for step in range(steps):
            pdf = torch.softmax(torch.randn(cardinality), dim=0)
            torch.multinomial(pdf, 1)                       # the self-test's symbol draw advances the generator
            pdfs.append(pdf)
        pdfs = torch.stack(pdfs)[None]                      # [1, S, card]
        cdf = b.build_cdf(pdfs.cuda())
        crc = 0
        for row in cdf.cpu().numpy()[0].astype(np.int64):
            crc = zlib.crc32(row.tobytes(), crc)
        assert crc == int(g[f"ac{i}.meta"][2]), "quantised cdfs differ from the reference's"
        sym = torch.from_numpy(g[f"ac{i}.symbols"]).reshape(1, -1)
        stream = b.ac_encode(sym.cuda(), cdf)
        assert stream[0] == g[f"ac{i}.bytes"].tobytes(), i
        assert np.array_equal(b.ac_decode(stream, steps, cdf).cpu().numpy(), sym.numpy())


def test_bench_size_codes_round_trip_and_container():
    """B = 32 utterances x 6 codebooks x 120 frames (BASELINE configs[1]): packing equals the oracle on every stream, the
    ECDC container round-trips, the static-model coder round-trips and is smaller than plain packing on skewed codes."""
    b = bs()
    g = torch.Generator().manual_seed(3)
    codes = torch.randint(0, 1024, (6, 32, 120), generator=g)
    packed = b.pack_codes(codes.cuda(), 10).cpu().numpy()
    assert packed.shape == (32, 900)                       # 3 kbps: 900 bytes per 2.4 s
    for i in (0, 7, 31):
        assert packed[i].tobytes() == BO.pack_bits(BO.frame_code_order(codes[:, i].numpy()).tolist(), 10)
    blobs = b.compress_codes(codes.cuda(), audio_length=38400)
    meta, off = BO.read_ecdc_header(blobs[5])
    assert meta == {"m": "ladiffcodec_16khz", "al": 38400, "nc": 6, "lm": False, "hop": 320} and blobs[5][off:] == packed[5].tobytes()
    back, metas = b.decompress_codes(blobs, 120)
    back2, _ = b.decompress_codes(blobs)                    # F from the header (al / hop)
    assert torch.equal(back2.cpu(), codes)
    with pytest.raises(ValueError):
        b.decompress_codes(blobs, 119)                      # caller's F disagrees with the header
    assert torch.equal(back.cpu(), codes) and metas[0]["al"] == 38400
    pdf = torch.softmax(torch.randn(6, 1024, generator=g) * 2.0, dim=-1)
    skew = torch.stack([torch.multinomial(pdf[k], 32 * 120, replacement=True, generator=g).reshape(32, 120) for k in range(6)])
    cdf = b.build_cdf(pdf.cuda())
    blobs = b.compress_codes(skew.cuda(), 38400, static_cdf=cdf)
    back, _ = b.decompress_codes(blobs, 120, static_cdf=cdf)
    assert torch.equal(back.cpu(), skew)
    # a static-table stream says so in its own field and keeps the reference's `lm` false (compress.py:47-71: lm = payload coded
    # with the model's LM pdfs); decoding it as plain packing, or a plain stream with a table, is refused instead of mis-decoded
    meta_s, _ = BO.read_ecdc_header(blobs[0])
    assert meta_s["lm"] is False and meta_s["ac"] == "static"
    with pytest.raises(ValueError):
        b.decompress_codes(blobs, 120)
    with pytest.raises(ValueError):
        b.decompress_codes(b.compress_codes(codes.cuda(), 38400), 120, static_cdf=cdf)
    with pytest.raises(ValueError):
        b.compress_codes(skew.cuda(), 38400, static_cdf=cdf[:5])   # one table per codebook
    hdr = len(blobs[0]) - len(blobs[0][BO.read_ecdc_header(blobs[0])[1]:])
    assert max(len(x) for x in blobs) - hdr < 900
    want = BO.ac_encode(BO.frame_code_order(skew[:, 9].numpy()).tolist(), cdf.cpu().numpy().astype(np.int64), np.tile(np.arange(6), 120))
    assert blobs[9][BO.read_ecdc_header(blobs[9])[1]:] == want
    # an empty frame packs to nothing
    assert b.pack_codes(torch.zeros(6, 2, 0, dtype=torch.int64).cuda(), 10).shape == (2, 0)
