"""Pin the CPU oracle (oracle/ldc_oracle.py) to outputs of the reference itself (tests/golden/,
made by tools/gen_golden.py).  CPU-only; this is what lets the GPU parity tests trust the oracle."""
import numpy as np
import pytest
import torch

from ladiffcodec_amd import synth
from oracle import ldc_oracle as O
from helpers import CASES, COND_CFG, T, cond_sd_np, driver_noises, load_golden, main_sd_np, rel_err

TOL = 2e-5   # fp32 CPU vs fp32 CPU, different op order only


def test_schedule_tables_match_reference():
    gold = load_golden("schedule")
    mine = synth.cosine_schedule_buffers(1000)
    assert set(mine) == set(gold)
    for k in mine:
        np.testing.assert_allclose(mine[k], gold[k], rtol=2e-6, atol=1e-12, err_msg=k)


def test_sconv1d_padding_rules():
    g = load_golden("primitives")
    names = sorted({k.split(".")[0] for k in g if k.startswith("c_")})
    assert len(names) == 7
    for n in names:
        k, s, d, causal = (int(v) for v in g[n + ".cfg"])
        w = O.fold_weight_norm(T(g[n + ".g"]), T(g[n + ".v"]))
        y = O.sconv1d(T(g[n + ".x"]), w, T(g[n + ".b"]), stride=s, dilation=d, causal=bool(causal))
        assert y.shape == g[n + ".y"].shape, n
        assert rel_err(y.numpy(), g[n + ".y"]) < TOL, n


def test_sconvtranspose1d_trim_rules():
    g = load_golden("primitives")
    names = sorted({k.split(".")[0] for k in g if k.startswith("t_")})
    assert len(names) == 5
    for n in names:
        k, s, d, causal = (int(v) for v in g[n + ".cfg"])
        w = O.fold_weight_norm(T(g[n + ".g"]), T(g[n + ".v"])) if (n + ".g") in g else T(g[n + ".w"])
        y = O.sconvtr1d(T(g[n + ".x"]), w, T(g[n + ".b"]), s, causal=bool(causal))
        assert y.shape == g[n + ".y"].shape, n
        assert rel_err(y.numpy(), g[n + ".y"]) < TOL, n


def test_lstm_skip():
    g = load_golden("primitives")
    sd = {"p." + k[len("lstm.sd."):]: T(v) for k, v in g.items() if k.startswith("lstm.sd.")}
    y = O.lstm_skip(T(g["lstm.x"]), sd, "p", 2)
    assert rel_err(y.numpy(), g["lstm.y"]) < TOL


def test_codec_encode_rvq_decode():
    g = load_golden("codec_c1")
    sd = synth.to_torch(cond_sd_np())
    wav = T(g["wav"])
    q, codes, margins, z = O.get_cond(sd, COND_CFG, wav)
    assert rel_err(z.numpy(), g["z"]) < TOL
    assert codes.shape == g["codes"].shape and codes.dtype == torch.int64
    safe = margins.numpy() > 1e-3
    assert safe.mean() > 0.95
    assert np.array_equal(codes.numpy()[safe], g["codes"][safe])
    if np.array_equal(codes.numpy(), g["codes"]):
        assert rel_err(q.numpy(), g["quantized"]) < TOL
    q15, codes15, _, _ = O.get_cond(sd, COND_CFG, wav, bandwidth=1.5)
    assert codes15.shape[0] == 3 and np.array_equal(codes15.numpy(), g["codes_1p5"])
    dec = O.seanet_decode(sd, COND_CFG, T(g["quantized"]))
    assert rel_err(dec.numpy(), g["decoded"]) < TOL


def test_codec_round_trip_full_size_c1():
    """BASELINE configs[0] at its full size (one 2.4 s clip, enc_ratios 8 5 4 2, bandwidth 3): the oracle against the REFERENCE's
    encoder -> RVQ -> decoder (tests/golden/codec_c1_full.npz from tools/gen_golden_c1.py; model.py:223-231, seanet.py:66-248)."""
    g = load_golden("codec_c1_full")
    sd = synth.to_torch(cond_sd_np())
    seed_w, Tn, seed_in = (int(v) for v in g["meta"])
    wav = torch.from_numpy(synth.synthetic_wav(1, Tn, seed=seed_in)) * 0.5
    q, codes, margins, z = O.get_cond(sd, COND_CFG, wav)
    assert rel_err(z.numpy(), g["z"]) < TOL
    safe = margins.numpy() > 1e-3
    assert safe.mean() > 0.95 and np.array_equal(codes.numpy()[safe], g["codes"][safe])
    dec = O.seanet_decode(sd, COND_CFG, T(g["quantized"]))
    assert dec.shape[-1] == Tn and rel_err(dec.numpy(), g["decoded"]) < TOL


@pytest.mark.parametrize("tag", ["r84", "r8"])
def test_unet_step_and_chain(tag):
    g = load_golden("ladiff_" + tag)
    mc, u, _ = CASES[tag]
    sd = synth.to_torch(main_sd_np(tag))
    cond, x = T(g["cond"]), T(g["x"])
    assert rel_err(O.process_cond(sd, u, cond).numpy(), g["cond_proc"]) < TOL
    for t in (0, 37):
        eps = O.unet_forward(sd, u, x, torch.full((2,), t, dtype=torch.long), cond)
        assert rel_err(eps.numpy(), g[f"eps_t{t}"]) < 5e-5, t
    assert rel_err(O.cond_upsample(sd, u, cond).numpy(), g["img_up"]) < TOL
    img0 = O.start_image(sd, u, cond)
    assert rel_err(img0.numpy(), g["img0"]) < TOL
    n = int(g["meta"][2])
    noises = T(g["noises"])
    one = O.p_sample(sd, u, x, 5, cond, noises[n - 1])
    assert rel_err(one.numpy(), g["p_sample_t5"]) < 5e-5
    lat = O.halfway_sampling(sd, u, img0, cond, n, noises)
    assert rel_err(lat.numpy(), g["latents"]) < 2e-4
    wav_raw = O.seanet_decode(sd, mc, T(g["latents"]))
    assert rel_err(wav_raw.numpy(), g["wav_raw"]) < TOL
    assert rel_err(O.output_normalise(T(g["wav_raw"])).numpy(), g["wav_out"]) < TOL


def test_end_to_end_stage_tensors():
    g = load_golden("ladiff_r84")
    mc, u, _ = CASES["r84"]
    out = O.decode_utterances(synth.to_torch(cond_sd_np()), COND_CFG, synth.to_torch(main_sd_np("r84")), mc, u,
                              T(g["wav"]), int(g["meta"][2]), T(g["noises"]))
    assert rel_err(out["cond"].numpy(), g["cond"]) < TOL
    assert rel_err(out["img0"].numpy(), g["img0"]) < TOL
    assert rel_err(out["latents"].numpy(), g["latents"]) < 2e-4
    assert rel_err(out["wav"].numpy(), g["wav_out"]) < 1e-3


def test_alternative_samplers_match_reference():
    """p_sample_loop (all 1000 steps from a normal start) and infilling, against the
    reference's own runs with the same start images and noise tapes."""
    g = load_golden("drivers_r84")
    mc, u, _ = CASES["r84"]
    sd = synth.to_torch(main_sd_np("r84"))
    loop_noise, fill_noise, midway_t = driver_noises(g)
    cond = T(g["cond"])
    out = O.p_sample_loop(sd, u, T(g["loop_img0"]), cond, loop_noise)
    assert rel_err(out.numpy(), g["loop_out"]) < 2e-4      # 1000 chained steps
    img, _ = O.infilling(sd, u, T(g["fill_img0"]), T(g["fill_infill0"]), cond, midway_t, fill_noise, lam=0.8)
    assert rel_err(img.numpy(), g["fill_out"]) < TOL


def test_flag_variants_match_reference():
    """--unet_scale_x (unet.py:432-433), upsampling_ratios=None (unet.py:411) and --final_activation (seanet.py:144-149)
    against the reference's own outputs (tests/golden/variants.npz)."""
    from ladiffcodec_amd.spec import CodecConfig, UnetConfig
    g = load_golden("variants")
    mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
    t37 = torch.full((2,), 37, dtype=torch.long)
    u = UnetConfig(dim=32, upsampling_ratios=(5, 2), unet_scale_cond=True, unet_scale_x=True)
    sd = synth.to_torch(synth.ladiff_state_dict(mc, u, int(g["meta"][0])))
    assert rel_err(O.unet_forward(sd, u, T(g["sx.x"]), t37, T(g["sx.cond"])).numpy(), g["sx.eps_t37"]) < 5e-5
    u = UnetConfig(dim=32, upsampling_ratios=None, unet_scale_cond=True)
    sd = synth.to_torch(synth.ladiff_state_dict(mc, u, int(g["meta"][1])))
    assert not any("upsampling_layers" in k for k in sd)
    assert rel_err(O.unet_forward(sd, u, T(g["nu.x"]), t37, T(g["nu.cond"])).numpy(), g["nu.eps_t37"]) < 5e-5
    assert rel_err(O.p_sample(sd, u, T(g["nu.x"]), 0, T(g["nu.cond"]), None).numpy(), g["nu.p_sample_t0"]) < 5e-5
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0, final_activation="Tanh")
    sdc = synth.to_torch(synth.codec_state_dict(cc, int(g["meta"][2])))
    q, codes, margins, z = O.get_cond(sdc, cc, T(g["fa.wav"]))
    assert rel_err(z.numpy(), g["fa.z"]) < TOL and float(np.abs(g["fa.z"]).max()) <= 1.0
    safe = margins.numpy() > 1e-3
    assert np.array_equal(codes.numpy()[safe], g["fa.codes"][safe])
