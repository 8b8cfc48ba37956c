"""tests/golden/codec_c1_full.npz: BASELINE configs[0] at its full size -- EnCodec-style round trip of ONE 2.4 s clip through the
cond codec (enc_ratios 8 5 4 2, bandwidth 3): SEANetEncoder -> RVQ -> SEANetDecoder, computed by the REFERENCE
(/root/reference: srcs/model.py:223-231 get_cond, srcs/encodec/modules/seanet.py:66-248) on the seeded synthetic checkpoint the tests
regenerate (codec seed 11).  Run in the build container only:   python tools/gen_golden_c1.py
The 0.4 s fixture (codec_c1.npz) never ran the decoder's LSTM (H = 512) over F = 120 frames or the k16 s8 transposed conv at
that length."""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
warnings.filterwarnings("ignore")

from gen_golden import OUT, build_cond_model, np32  # noqa: E402
from ref_import import import_reference  # noqa: E402
from ladiffcodec_amd import synth  # noqa: E402
from ladiffcodec_amd.spec import CodecConfig  # noqa: E402


def main():
    torch.set_num_threads(8)
    ref = import_reference()
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
    m = build_cond_model(ref, cc, seed=11)
    T = 38400                                                     # 2.4 s at 16 kHz
    wav = torch.from_numpy(synth.synthetic_wav(1, T, seed=4242)) * 0.5
    with torch.no_grad():
        z = m.encoder(wav)
        q = m.quantizer(z, sample_rate=m.frame_rate, bandwidth=m.bandwidth)
        cond = m.get_cond(wav)
        dec = m.decoder(q.quantized)
    assert torch.equal(cond, q.quantized) and dec.shape[-1] == T
    # top-2 margin of every nearest-neighbour decision (where it is tiny a different summation order may flip the index)
    np.savez_compressed(os.path.join(OUT, "codec_c1_full.npz"), z=np32(z), quantized=np32(q.quantized), codes=q.codes.numpy().astype(np.int64),
                        decoded=np32(dec), meta=np.array([11, T, 4242], np.int64))
    print("codec_c1_full: z", tuple(z.shape), "codes", tuple(q.codes.shape), "decoded", tuple(dec.shape))


if __name__ == "__main__":
    main()
