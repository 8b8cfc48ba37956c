"""Audio front end on the GPU: ldc_resample against the oracle's restatement of
torchaudio.functional.resample (srcs/sample.py:84), and the CLI on a non-16 kHz file."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ladiffcodec_amd import synth  # noqa: E402
from oracle import ldc_oracle as O, resample_oracle as RO  # noqa: E402
from helpers import (CASES, COND_CFG, T, cli50_c5_audio, cli50_c5_plan, cli50_c5_tape, cli50_default_audio, cli50_default_tape, cond_sd_np,  # noqa: E402
                     load_golden, main_sd_np)
from gpu_common import engine, rel  # noqa: E402


@pytest.mark.parametrize("sr", [8000, 22050, 44100, 48000, 16000])
def test_resample_matches_oracle(sr):
    e = engine("r84", "f32")
    x = np.random.default_rng(sr).standard_normal((2, 20000)).astype(np.float32) * 0.3
    got = e.resample(torch.from_numpy(x), sr, 16000).cpu().numpy()
    want = RO.resample(x, sr, 16000)
    assert got.shape == want.shape
    assert rel(got, want) < 1e-5            # same fp32 filter bank; only the summation order differs


def test_cli_resamples_non_16k_input(tmp_path):
    from scipy.io import wavfile
    from ladiffcodec_amd import sample as cli
    mc, u, _ = CASES["r84"]
    synth.save_amlt(main_sd_np("r84"), str(tmp_path / "ladiff.amlt"))
    synth.save_amlt(cond_sd_np(), str(tmp_path / "codec.amlt"))
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    sr = 22050
    n = 7500          # -> 5443 samples at 16 kHz -> trimmed to 5120 = a latent length (160) that survives the four halvings
    x = (0.3 * np.sin(2 * np.pi * 220.0 * np.arange(n) / sr) + 0.05 * np.random.default_rng(1).standard_normal(n)).astype(np.float32)
    wavfile.write(str(ind / "a.wav"), sr, x)
    args = cli.build_parser().parse_args([
        "--model_for_cond", str(tmp_path / "codec.amlt"), "--model_path", str(tmp_path / "ladiff.amlt"), "--run_diff", "--scaling_global",
        "--cond_bandwidth", "3", "--unet_scale_cond", "--enc_ratios", "8", "4", "--upsampling_ratios", "5", "2", "--diff_dims", "32",
        "--input_dir", str(ind) + "/", "--output_dir", str(outd) + "/", "--midway_t", "1", "--dtype", "f32"])
    written = cli.synthesis(args)
    assert len(written) == 1
    sr_out, y = wavfile.read(str(outd / "a.wav"))
    x16 = RO.resample(x[None], sr, 16000)[0]
    m = len(x16) // 640 * 640
    assert sr_out == 16000 and y.shape == (m,)
    ref = O.decode_utterances(synth.to_torch(cond_sd_np()), COND_CFG, synth.to_torch(main_sd_np("r84")), mc, u,
                              T(x16[:m]).reshape(1, 1, m), 1, None)
    assert rel(y, ref["wav"].numpy().reshape(-1)) < 5e-3


def test_cli_long_form_chunks_against_oracle(tmp_path):
    """--chunk_sec (BASELINE config 5): a recording of 2 chunks + a shorter tail through the CLI; every chunk's raw decoder
    output must equal the oracle's for that chunk decoded alone, and the normalisation must run over the joined recording."""
    from scipy.io import wavfile
    from ladiffcodec_amd import sample as cli
    mc, u, _ = CASES["r84"]
    synth.save_amlt(main_sd_np("r84"), str(tmp_path / "ladiff.amlt"))
    synth.save_amlt(cond_sd_np(), str(tmp_path / "codec.amlt"))
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    chunk, tail = 5120, 2560
    n = 2 * chunk + tail + 300
    t = np.arange(n) / 16000.0
    x = (0.3 * np.sin(2 * np.pi * 180.0 * t) + 0.1 * np.sin(2 * np.pi * 1230.0 * t)).astype(np.float32)
    wavfile.write(str(ind / "long.wav"), 16000, x)
    args = cli.build_parser().parse_args([
        "--model_for_cond", str(tmp_path / "codec.amlt"), "--model_path", str(tmp_path / "ladiff.amlt"), "--run_diff", "--scaling_global",
        "--cond_bandwidth", "3", "--unet_scale_cond", "--enc_ratios", "8", "4", "--upsampling_ratios", "5", "2", "--diff_dims", "32",
        "--input_dir", str(ind) + "/", "--output_dir", str(outd) + "/", "--midway_t", "1", "--dtype", "f32",
        "--chunk_sec", str(chunk / 16000.0)])
    written = cli.synthesis(args)
    assert len(written) == 1
    sr_out, y = wavfile.read(str(outd / "long.wav"))
    assert sr_out == 16000 and y.shape == (2 * chunk + tail,)
    sdc, sdm = synth.to_torch(cond_sd_np()), synth.to_torch(main_sd_np("r84"))
    raws = []
    for st, ln in ((0, chunk), (chunk, chunk), (2 * chunk, tail)):
        ref = O.decode_utterances(sdc, COND_CFG, sdm, mc, u, T(x[st:st + ln]).reshape(1, 1, ln), 1, None, per_item=True)
        raws.append(ref["wav_raw"])                                    # un-normalised decoder output of the chunk
    whole = O.output_normalise(torch.cat(raws, dim=-1))
    assert rel(y, whole.numpy().reshape(-1)) < 5e-3


def test_cli_two_batches_in_flight_against_oracle(tmp_path):
    """--in_flight 2 (the CLI's default when a run has several batches): three files at batch_size 1 go through two engines on
    two streams; every output must equal the oracle's for that file decoded alone."""
    from scipy.io import wavfile
    from ladiffcodec_amd import sample as cli
    mc, u, _ = CASES["r84"]
    synth.save_amlt(main_sd_np("r84"), str(tmp_path / "ladiff.amlt"))
    synth.save_amlt(cond_sd_np(), str(tmp_path / "codec.amlt"))
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    n = 5120
    xs = {}
    for k, f0 in enumerate((150.0, 310.0, 95.0)):
        t = np.arange(n) / 16000.0
        xs[f"u{k}.wav"] = (0.25 * np.sin(2 * np.pi * f0 * t) + 0.05 * np.sin(2 * np.pi * 7.3 * f0 * t)).astype(np.float32)
        wavfile.write(str(ind / f"u{k}.wav"), 16000, xs[f"u{k}.wav"])
    args = cli.build_parser().parse_args([
        "--model_for_cond", str(tmp_path / "codec.amlt"), "--model_path", str(tmp_path / "ladiff.amlt"), "--run_diff", "--scaling_global",
        "--cond_bandwidth", "3", "--unet_scale_cond", "--enc_ratios", "8", "4", "--upsampling_ratios", "5", "2", "--diff_dims", "32",
        "--input_dir", str(ind) + "/", "--output_dir", str(outd) + "/", "--midway_t", "1", "--dtype", "f32", "--batch_size", "1",
        "--in_flight", "2"])
    written = cli.synthesis(args)
    assert len(written) == 3
    sdc, sdm = synth.to_torch(cond_sd_np()), synth.to_torch(main_sd_np("r84"))
    for name, x in xs.items():
        y = wavfile.read(str(outd / name))[1]
        ref = O.decode_utterances(sdc, COND_CFG, sdm, mc, u, T(x).reshape(1, 1, n), 1, None)
        assert rel(y, ref["wav"].numpy().reshape(-1)) < 5e-3, name


def test_cli_default_mode_bf16_50_steps_in_flight(tmp_path):
    """The mode a user gets without flags of ours: --dtype bf16 (default), 50 reverse steps, graph replay, several batches
    and --in_flight 2 (default) -- seven files at batch_size 2 = four batches dealt to two engines on two streams, each batch
    one chain.  The noise of every file comes from a seeded tape (synthesis' test seam `noise_provider`; the device stream
    cannot be reproduced on the CPU), the oracle decodes every file alone with the same tape.  bf16 tolerance: 2x the drift
    measured on MI355X (tests/drift_tolerances.py, key wav_cli50)."""
    from scipy.io import wavfile
    from ladiffcodec_amd import sample as cli
    from drift_tolerances import check
    mc, u, _ = CASES["r84"]
    synth.save_amlt(main_sd_np("r84"), str(tmp_path / "ladiff.amlt"))
    synth.save_amlt(cond_sd_np(), str(tmp_path / "codec.amlt"))
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    n, steps = 5120, 50
    names = [f"u{k}.wav" for k in range(7)]                       # sorted glob order = this order
    for k, name in enumerate(names):
        wavfile.write(str(ind / name), 16000, cli50_default_audio(k, n))
    args = cli.build_parser().parse_args([
        "--model_for_cond", str(tmp_path / "codec.amlt"), "--model_path", str(tmp_path / "ladiff.amlt"), "--run_diff", "--scaling_global",
        "--cond_bandwidth", "3", "--unet_scale_cond", "--enc_ratios", "8", "4", "--upsampling_ratios", "5", "2", "--diff_dims", "32",
        "--input_dir", str(ind) + "/", "--output_dir", str(outd) + "/", "--midway_t", str(steps), "--batch_size", "2"])
    assert args.dtype == "bf16" and args.in_flight == 2
    args.noise_provider = lambda idxs, n_steps, L: torch.cat([cli50_default_tape(i, L, n_steps) for i in idxs], dim=1)
    written = cli.synthesis(args)
    assert len(written) == 7
    g = load_golden("cli50")                 # the oracle's decode of every file alone with the same tape (tools/gen_golden_cli.py)
    for i, name in enumerate(names):
        y = wavfile.read(str(outd / name))[1]
        check("bf16", "wav_cli50", rel(y, g[f"default.{i}"]), name)


def test_cli_config5_long_form_fp8_50_steps(tmp_path):
    """BASELINE configs[4] through synthesis(): ONE 30 s recording = 13 chunks of 2.4 s (12 full + a 1.2 s tail) decoded as batch
    items, --dtype fp8 (e4m3 UNet weights; fp8 x fp8 MFMA where a tensor's only consumer is a conv), 50 reverse steps, graph
    replay -- against the oracle run chunk by chunk on the SAME quantised weights and activations with the same noise tape, the
    chunks' raw decoder outputs joined and normalised over the whole recording (sample.py:133-134)."""
    from scipy.io import wavfile
    from ladiffcodec_amd import sample as cli
    from drift_tolerances import check
    mc, u, _ = CASES["r84"]
    synth.save_amlt(main_sd_np("r84"), str(tmp_path / "ladiff.amlt"))
    synth.save_amlt(cond_sd_np(), str(tmp_path / "codec.amlt"))
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    steps = 50
    x = cli50_c5_audio()
    plan, chunk = cli50_c5_plan(x.size)
    tail = plan[12][1]
    wavfile.write(str(ind / "long.wav"), 16000, x)
    args = cli.build_parser().parse_args([
        "--model_for_cond", str(tmp_path / "codec.amlt"), "--model_path", str(tmp_path / "ladiff.amlt"), "--run_diff", "--scaling_global",
        "--cond_bandwidth", "3", "--unet_scale_cond", "--enc_ratios", "8", "4", "--upsampling_ratios", "5", "2", "--diff_dims", "32",
        "--input_dir", str(ind) + "/", "--output_dir", str(outd) + "/", "--midway_t", str(steps), "--dtype", "fp8", "--chunk_sec", "2.4"])
    args.noise_provider = lambda keys, n_steps, Lz: torch.cat([cli50_c5_tape(k, Lz, n_steps) for _, k in keys], dim=1)
    written = cli.synthesis(args)
    assert len(written) == 1
    sr_out, y = wavfile.read(str(outd / "long.wav"))
    assert sr_out == 16000 and y.shape == (12 * chunk + tail,)
    # the oracle's result (chunk by chunk on the fake-quantised weights / activations, joined, normalised over the recording) is a
    # fixture since round 3: tools/gen_golden_cli.py, tests/golden/cli50.npz
    check("fp8", "wav_c5", rel(y, load_golden("cli50")["c5.whole"]))
