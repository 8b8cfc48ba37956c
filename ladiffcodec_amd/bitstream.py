"""On-wire format of the RVQ codes: host-side mirror of the reference's
`srcs/encodec/binary.py` (ECDC header, BitPacker / BitUnpacker) and `srcs/quantization/ac.py` (arithmetic coder) on
top of the C ABI.  The byte work runs on the GPU (csrc/bitstream.hip), one independent stream per utterance; the
container (header + per-utterance payload) is assembled on the host exactly as `srcs/encodec/compress.py:28-84` lays it out.
"""
from __future__ import annotations

import ctypes as C
import io
import json
import struct
import typing as tp

from . import lib as L

_encodec_header_struct = struct.Struct("!4sBI")      # binary.py:19
_ENCODEC_MAGIC = b"ECDC"


def write_ecdc_header(fo: tp.IO[bytes], metadata: tp.Any):
    """binary.py:22-28."""
    meta_dumped = json.dumps(metadata).encode("utf-8")
    fo.write(_encodec_header_struct.pack(_ENCODEC_MAGIC, 0, len(meta_dumped)))
    fo.write(meta_dumped)
    fo.flush()


def _read_exactly(fo: tp.IO[bytes], size: int) -> bytes:
    buf = b""
    while len(buf) < size:
        new_buf = fo.read(size)
        if not new_buf:
            raise EOFError("Impossible to read enough data from the stream, " f"{size} bytes remaining.")
        buf += new_buf
        size -= len(new_buf)
    return buf


def read_ecdc_header(fo: tp.IO[bytes]):
    """binary.py:43-52."""
    magic, version, meta_size = _encodec_header_struct.unpack(_read_exactly(fo, _encodec_header_struct.size))
    if magic != _ENCODEC_MAGIC:
        raise ValueError("File is not in ECDC format.")
    if version != 0:
        raise ValueError("Version not supported.")
    return json.loads(_read_exactly(fo, meta_size).decode("utf-8"))


class Bitstream:
    """GPU bit packer / range coder bound to an Engine's context and stream."""

    def __init__(self, eng):
        self.eng, self.lib, self.torch = eng, eng.lib, eng.torch

    # ---- BitPacker / BitUnpacker over code frames ---------------------------------------------------------------
    def pack_codes(self, codes, bits: int = 10):
        """codes [n_q, B, F] int64 -> uint8 [B, ceil(n_q*F*bits/8)], every row = BitPacker over `for t: for k: codes[k, b, t]`
        (compress.py:74-84) + flush."""
        t = self.torch
        codes = codes.to(self.eng.device, t.int64).contiguous()
        n_q, B, F = codes.shape
        nb = int(self.lib.ldc_packed_bytes(n_q, F, bits))
        out = t.empty(B, max(nb, 1), dtype=t.uint8, device=self.eng.device)
        if nb == 0:                       # an empty frame packs to nothing (BitPacker.flush without a push writes no byte)
            return out[:, :0]
        s = self.eng._enter()
        L.check(self.lib.ldc_pack_codes(self.eng._ctx, codes.data_ptr(), n_q, B, F, bits, out.data_ptr(), out.stride(0), s))
        self.eng._exit()
        return out[:, :nb]

    def unpack_codes(self, data, n_q: int, F: int, bits: int = 10):
        t = self.torch
        data = data.to(self.eng.device, t.uint8).contiguous()
        B = data.shape[0]
        if data.shape[1] < int(self.lib.ldc_packed_bytes(n_q, F, bits)):
            raise EOFError("The stream ended sooner than expected.")       # compress.py:148-149
        codes = t.empty(n_q, B, F, dtype=t.int64, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_unpack_codes(self.eng._ctx, data.data_ptr(), data.stride(0), n_q, B, F, bits, codes.data_ptr(), s))
        self.eng._exit()
        return codes

    # ---- arithmetic coder -----------------------------------------------------------------------------------------
    def build_cdf(self, pdf, total_range_bits: int = 24, roundoff: float = 1e-8, min_range: int = 2):
        """build_stable_quantized_cdf (ac.py:18-53) of every row of pdf [..., card] -> int32 cdf of the same shape."""
        t = self.torch
        pdf = pdf.to(self.eng.device, t.float32).contiguous()
        card = pdf.shape[-1]
        rows = pdf.numel() // card
        cdf = t.empty(pdf.shape, dtype=t.int32, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_ac_build_cdf(self.eng._ctx, pdf.data_ptr(), rows, card, total_range_bits, float(roundoff), min_range,
                                          cdf.data_ptr(), s))
        self.eng._exit()
        return cdf

    def ac_encode(self, symbols, cdf, static: bool = False, total_range_bits: int = 24, capacity: tp.Optional[int] = None):
        """symbols [B, S]; cdf [B, S, card] (static=False) or [n, card] with symbol s using table s % n (static=True).
        -> list of B `bytes` (ArithmeticCoder pushes + flush of every stream)."""
        t = self.torch
        symbols = symbols.to(self.eng.device, t.int32).contiguous()
        cdf = cdf.to(self.eng.device, t.int32).contiguous()
        B, S = symbols.shape
        card = cdf.shape[-1]
        cap = int(capacity) if capacity else S * 4 + 64           # 30-bit range: a symbol costs < 32 bits
        out = t.empty(B, cap, dtype=t.uint8, device=self.eng.device)
        nbytes = t.empty(B, dtype=t.int64, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_ac_encode(self.eng._ctx, symbols.data_ptr(), cdf.data_ptr(), B, S, card, cdf.shape[0] if static else 0,
                                       total_range_bits, out.data_ptr(), out.stride(0), nbytes.data_ptr(), s))
        self.eng._exit()
        n = nbytes.cpu().tolist()
        if min(n) < 0:
            raise ValueError("arithmetic coder: a symbol lies outside its table or the output capacity is too small")
        host = out.cpu().numpy()
        return [host[b, :n[b]].tobytes() for b in range(B)]

    def ac_decode(self, streams: tp.Sequence[bytes], S: int, cdf, static: bool = False, total_range_bits: int = 24):
        """-> symbols [B, S] int32; raises EOFError where ArithmeticDecoder.pull would return None (compress.py:148-149)."""
        import numpy as np
        t = self.torch
        B = len(streams)
        width = max(1, max(len(x) for x in streams))
        host = np.zeros((B, width), np.uint8)
        for b, x in enumerate(streams):
            host[b, :len(x)] = np.frombuffer(x, np.uint8)
        data = t.from_numpy(host).to(self.eng.device)
        nbytes = t.tensor([len(x) for x in streams], dtype=t.int64, device=self.eng.device)
        cdf = cdf.to(self.eng.device, t.int32).contiguous()
        symbols = t.zeros(B, S, dtype=t.int32, device=self.eng.device)
        status = t.zeros(B, dtype=t.int32, device=self.eng.device)
        s = self.eng._enter()
        L.check(self.lib.ldc_ac_decode(self.eng._ctx, data.data_ptr(), data.stride(0), nbytes.data_ptr(), cdf.data_ptr(), B, S,
                                       cdf.shape[-1], cdf.shape[0] if static else 0, total_range_bits, symbols.data_ptr(),
                                       status.data_ptr(), s))
        self.eng._exit()
        st = status.cpu().tolist()
        if any(v == 1 for v in st):
            raise EOFError("The stream ended sooner than expected.")
        if any(v == 2 for v in st):
            raise RuntimeError("Binary search failed")               # ac.py:241
        return symbols

    # ---- container: compress.py:28-84 / 87-156 with use_lm = False (plain packing) or a static per-codebook model ----------
    def compress_codes(self, codes, audio_length: int, model_name: str = "ladiffcodec_16khz", bits: int = 10,
                       static_cdf=None, hop_length: int = 320) -> tp.List[bytes]:
        """codes [n_q, B, F] -> one ECDC byte string per utterance (header + payload).

        Header fields as compress.py:47-71 writes them; `lm` keeps the reference's meaning (payload arithmetic-coded with the
        model's LM pdfs, compress.py:118-141), which this library never produces, so it is always false.  A payload coded with
        a caller-supplied static per-codebook table says so in its own field `ac: "static"`; a reference decoder that meets
        it reads `lm: false`, plain-unpacks and finds the stream too short instead of silently mis-decoding."""
        n_q, B, F = codes.shape
        if static_cdf is None:
            payloads = [bytes(r) for r in self.pack_codes(codes, bits).cpu().numpy()]
        else:
            if static_cdf.shape[0] != n_q:
                raise ValueError(f"static_cdf has {static_cdf.shape[0]} tables for {n_q} codebooks (symbol s uses table s % n_q)")
            sym = codes.permute(1, 2, 0).reshape(B, F * n_q)          # push order: t outer, k inner
            payloads = self.ac_encode(sym, static_cdf, static=True)
        if -(-int(audio_length) // int(hop_length)) != F:
            raise ValueError(f"audio_length {audio_length} does not give {F} frames at hop {hop_length}")
        out = []
        for b in range(B):
            fo = io.BytesIO()
            meta = {"m": model_name, "al": int(audio_length), "nc": int(n_q), "lm": False, "hop": int(hop_length)}
            if static_cdf is not None:
                meta["ac"] = "static"
            write_ecdc_header(fo, meta)
            fo.write(payloads[b])
            out.append(fo.getvalue())
        return out

    def decompress_codes(self, blobs: tp.Sequence[bytes], F: tp.Optional[int] = None, bits: int = 10, static_cdf=None):
        """-> (codes [n_q, B, F] int64, list of metadata).  F is derived from the header (`al`, `hop`) and, when given, must agree."""
        import numpy as np
        metas, payloads = [], []
        for blob in blobs:
            fo = io.BytesIO(blob)
            metas.append(read_ecdc_header(fo))
            payloads.append(fo.read())
        n_q = metas[0]["nc"]
        if any(m["nc"] != n_q for m in metas):
            raise ValueError("streams of one batch must share the number of codebooks")
        for m in metas:
            if m.get("lm"):
                raise ValueError("stream is coded with a language model (lm: true): not produced nor decodable here")
            mode = m.get("ac", "none")
            if mode not in ("none", "static"):
                raise ValueError(f"unknown entropy-coding mode {mode!r}")
            if (mode == "static") != (static_cdf is not None):
                raise ValueError("stream is static-table arithmetic-coded: pass the table it was coded with" if mode == "static"
                                 else "stream is plainly packed: static_cdf must not be given")
            frames = -(-int(m["al"]) // int(m.get("hop", 320)))
            if F is None:
                F = frames
            if frames != F:
                raise ValueError(f"header says {frames} frames (al {m['al']}), caller expects {F}")
        if static_cdf is None:
            width = max(len(p) for p in payloads)
            host = np.zeros((len(payloads), max(width, 1)), np.uint8)
            for b, p in enumerate(payloads):
                host[b, :len(p)] = np.frombuffer(p, np.uint8)
            if min(len(p) for p in payloads) < int(self.lib.ldc_packed_bytes(n_q, F, bits)):
                raise EOFError("The stream ended sooner than expected.")
            return self.unpack_codes(self.torch.from_numpy(host), n_q, F, bits), metas
        if static_cdf.shape[0] != n_q:
            raise ValueError(f"static_cdf has {static_cdf.shape[0]} tables for {n_q} codebooks")
        sym = self.ac_decode(payloads, F * n_q, static_cdf, static=True)
        return sym.reshape(len(payloads), F, n_q).permute(2, 0, 1).contiguous().to(self.torch.int64), metas
