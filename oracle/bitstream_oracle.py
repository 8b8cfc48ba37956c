"""CPU restatement of the reference's bit-stream layer -- TEST INFRASTRUCTURE ONLY
(used by tests/ and tools/, never by the product path).

Restated from (read-only) /root/reference:
  BitPacker / BitUnpacker            srcs/encodec/binary.py:55-118   (LSB-first, non-byte-aligned widths)
  ECDC header                        srcs/encodec/binary.py:14-52    ('!4sBI' magic/version/size + JSON metadata)
  build_stable_quantized_cdf         srcs/quantization/ac.py:18-53   (identical file: srcs/encodec/quantization/ac.py)
  ArithmeticCoder / ArithmeticDecoder srcs/quantization/ac.py:56-260
  code order of a frame              srcs/encodec/compress.py:68-84  (for t: for k: push(frame[0, k, t]))
Pinned by tests/golden/bitstream.npz, which tools/gen_bitstream_golden.py produced by running the reference's own
classes, including its two seeded self-tests (binary.py:125-149, ac.py:263-288).
"""
import json
import math
import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

ECDC_STRUCT = struct.Struct("!4sBI")
ECDC_MAGIC = b"ECDC"


def write_ecdc_header(metadata) -> bytes:
    """binary.py:22-28."""
    meta = json.dumps(metadata).encode("utf-8")
    return ECDC_STRUCT.pack(ECDC_MAGIC, 0, len(meta)) + meta


def read_ecdc_header(buf: bytes) -> Tuple[dict, int]:
    """binary.py:43-52 -> (metadata, offset of the payload)."""
    magic, version, size = ECDC_STRUCT.unpack(buf[:ECDC_STRUCT.size])
    if magic != ECDC_MAGIC:
        raise ValueError("File is not in ECDC format.")
    if version != 0:
        raise ValueError("Version not supported.")
    end = ECDC_STRUCT.size + size
    if len(buf) < end:
        raise EOFError("Impossible to read enough data from the stream")
    return json.loads(buf[ECDC_STRUCT.size:end].decode("utf-8")), end


def pack_bits(values: Sequence[int], bits: int) -> bytes:
    """BitPacker.push for every value, then flush (binary.py:69-87)."""
    cur, nb, out = 0, 0, bytearray()
    for v in values:
        cur += int(v) << nb
        nb += bits
        while nb >= 8:
            out.append(cur & 0xFF)
            cur >>= 8
            nb -= 8
    if nb:
        out.append(cur)
    return bytes(out)


def unpack_bits(data: bytes, bits: int, count: Optional[int] = None) -> List[int]:
    """BitUnpacker.pull until the stream ends (binary.py:103-118); `count` stops early (the flush can leave ghost values)."""
    cur, nb, pos, out = 0, 0, 0, []
    mask = (1 << bits) - 1
    while count is None or len(out) < count:
        while nb < bits:
            if pos >= len(data):
                return out
            cur += data[pos] << nb
            pos += 1
            nb += 8
        out.append(cur & mask)
        cur >>= bits
        nb -= bits
    return out


def frame_code_order(codes: np.ndarray) -> np.ndarray:
    """compress.py:68-84: codes [K, T] -> the sequence pushed: t outer, k inner."""
    return np.ascontiguousarray(codes.T).reshape(-1)


def build_stable_quantized_cdf(pdf: np.ndarray, total_range_bits: int, roundoff: float = 1e-8, min_range: int = 2) -> np.ndarray:
    """ac.py:18-53 in float32, as torch computes it on the CPU: tensor / python-float and tensor * python-float are
    float32 operations with the scalar rounded to float32."""
    p = np.asarray(pdf, np.float32)
    if roundoff:
        r = np.float32(roundoff)
        p = np.floor(p / r) * r
    total_range = 2 ** total_range_bits
    card = p.shape[-1]
    alpha = min_range * card / total_range
    assert alpha <= 1, "you must reduce min_range"
    ranges = np.floor(np.float32((1 - alpha) * total_range) * p).astype(np.int64) + min_range
    return np.cumsum(ranges, axis=-1)


class ArithmeticCoder:
    """ac.py:56-174 (bits are packed LSB-first, one at a time)."""

    def __init__(self, total_range_bits: int = 24):
        assert total_range_bits <= 30
        self.total_range_bits = total_range_bits
        self.low = 0
        self.high = 0
        self.max_bit = -1
        self.bits: List[int] = []

    def _flush_common_prefix(self):
        while self.max_bit >= 0:
            b1 = self.low >> self.max_bit
            b2 = self.high >> self.max_bit
            if b1 != b2:
                break
            self.low -= b1 << self.max_bit
            self.high -= b1 << self.max_bit
            self.max_bit -= 1
            self.bits.append(b1)

    def push(self, symbol: int, cdf: np.ndarray):
        while self.high - self.low + 1 < 2 ** self.total_range_bits:
            self.low *= 2
            self.high = self.high * 2 + 1
            self.max_bit += 1
        delta = self.high - self.low + 1
        range_low = 0 if symbol == 0 else int(cdf[symbol - 1])
        range_high = int(cdf[symbol]) - 1
        eff_low = int(math.ceil(range_low * (delta / (2 ** self.total_range_bits))))
        eff_high = int(math.floor(range_high * (delta / (2 ** self.total_range_bits))))
        self.high = self.low + eff_high
        self.low = self.low + eff_low
        assert self.low <= self.high
        self._flush_common_prefix()

    def flush(self) -> bytes:
        while self.max_bit >= 0:
            self.bits.append((self.low >> self.max_bit) & 1)
            self.max_bit -= 1
        return pack_bits(self.bits, 1)


class ArithmeticDecoder:
    """ac.py:177-260."""

    def __init__(self, data: bytes, total_range_bits: int = 24):
        self.total_range_bits = total_range_bits
        self.low = self.high = self.current = 0
        self.max_bit = -1
        self.data = data
        self.bitpos = 0

    def _pull_bit(self) -> Optional[int]:
        if self.bitpos >= 8 * len(self.data):
            return None
        b = (self.data[self.bitpos >> 3] >> (self.bitpos & 7)) & 1
        self.bitpos += 1
        return b

    def pull(self, cdf: np.ndarray) -> Optional[int]:
        while self.high - self.low + 1 < 2 ** self.total_range_bits:
            bit = self._pull_bit()
            if bit is None:
                return None
            self.low *= 2
            self.high = self.high * 2 + 1
            self.current = self.current * 2 + bit
            self.max_bit += 1
        delta = self.high - self.low + 1
        lo_i, hi_i = 0, len(cdf) - 1
        while True:
            if hi_i < lo_i:
                raise RuntimeError("Binary search failed")
            mid = (lo_i + hi_i) // 2
            range_low = int(cdf[mid - 1]) if mid > 0 else 0
            range_high = int(cdf[mid]) - 1
            low = int(math.ceil(range_low * (delta / (2 ** self.total_range_bits)))) + self.low
            high = int(math.floor(range_high * (delta / (2 ** self.total_range_bits)))) + self.low
            if self.current >= low:
                if self.current <= high:
                    break
                lo_i = mid + 1
            else:
                hi_i = mid - 1
        self.low, self.high = low, high
        while self.max_bit >= 0:
            b1 = self.low >> self.max_bit
            b2 = self.high >> self.max_bit
            if b1 != b2:
                break
            self.low -= b1 << self.max_bit
            self.high -= b1 << self.max_bit
            self.current -= b1 << self.max_bit
            self.max_bit -= 1
        return mid


def ac_encode(symbols: Sequence[int], cdfs: np.ndarray, rows: Optional[Sequence[int]] = None, total_range_bits: int = 24) -> bytes:
    """Encode symbols[s] with cdfs[rows[s]] (rows = None: row s)."""
    enc = ArithmeticCoder(total_range_bits)
    for s, sym in enumerate(symbols):
        enc.push(int(sym), cdfs[s if rows is None else rows[s]])
    return enc.flush()


def ac_decode(data: bytes, n: int, cdfs: np.ndarray, rows: Optional[Sequence[int]] = None, total_range_bits: int = 24) -> List[int]:
    dec = ArithmeticDecoder(data, total_range_bits)
    out = []
    for s in range(n):
        v = dec.pull(cdfs[s if rows is None else rows[s]])
        if v is None:
            raise EOFError("The stream ended sooner than expected.")
        out.append(v)
    return out
