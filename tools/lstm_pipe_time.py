"""Wall time of the main decoder (ldc_seanet_decode) of one batch part with the two-layer LSTM pipeline on / off."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from ladiffcodec_amd import lib as L, synth
    from ladiffcodec_amd.model import Engine
    from ladiffcodec_amd.spec import CodecConfig, UnetConfig
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
    u = UnetConfig(dim=32, upsampling_ratios=(5, 2), unet_scale_cond=True)
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
    sd = synth.ladiff_state_dict(mc, u, seed=1)
    e = Engine(mc, u, cc, dtype="f32", device=0)
    e.load_state_dict(L.MODEL_MAIN, {k: v for k, v in sd.items() if not k.startswith("diffusion.model.")})
    e.load_state_dict(L.MODEL_COND, synth.codec_state_dict(cc, seed=0))
    e.finalize(strict=True)
    z = torch.randn(B, 128, 1200, device="cuda") * 0.1
    for opt in (1, 0, 1, 0):
        e.set_option("lstm_pipe", opt)
        for _ in range(3):
            e.decode_latents(L.MODEL_MAIN, z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            e.decode_latents(L.MODEL_MAIN, z)
        torch.cuda.synchronize()
        print(f"lstm_pipe {opt}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per decoder pass of {B} items")


if __name__ == "__main__":
    main()
