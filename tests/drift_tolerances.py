"""Tolerances of the bench-shape parity tests = 2x the drift MEASURED on MI355X (relative to max |reference|).

Regenerate the measurements with
    LDC_RECORD_DRIFT=gpurun_out/drift.json python -m pytest tests/test_gpu_bench_shape.py -m gpu -q
(the tests then record instead of asserting) and copy 2x the worst value of every key here; DESIGN.md section 2
quotes the same numbers.  f32 = exact-fp32 MFMA engine, bf16 = bf16 storage + MFMA with fp32 accumulation/state.
"""
import json
import os

MEASURED = {   # worst value seen over items / timesteps / layouts on MI355X, round 2 (two recordings, before and after the fused
    # attention tail; gpurun_out/drift.json, rounded up)
    "f32": {"eps_bench": 2.3e-6, "tap_bench": 3.2e-6, "lat_50": 1.6e-6, "wav_50": 1.0e-6, "chain_250": 1.8e-6, "lat_200": 2.8e-6,
            "wav_200": 1.4e-6, "eps_small": 2.2e-6, "chain_small": 9e-7, "wav_small": 1.2e-6, "repeat": 2e-7},
    "bf16": {"eps_bench": 1.13e-2, "tap_bench": 1.39e-2, "lat_50": 1.2e-3, "wav_50": 2.4e-4, "chain_250": 8.5e-3, "lat_200": 5.1e-3,
             "wav_200": 8.8e-4, "eps_small": 2.12e-2, "chain_small": 3.2e-4, "wav_small": 1.6e-4, "repeat": 1.8e-4,
             "wav_cli50": 3.2e-4},     # round 3: the CLI's default mode (50 steps, dim 32, two engines in flight)
    # fp8 (e4m3) UNet weights with per-channel scales against the UNQUANTISED fp32 oracle: the price of config 5's weights
    "fp8": {"eps_vs_unquantised": 0.125,     # measured: 0.125 on the synthetic (Gaussian) checkpoints at dim 256
            # round 3 (recorded on MI355X): the fp8 x fp8 path (act8: tensors whose only consumer is a conv are produced in fp8) against
            # the oracle on the same quantised weights + activations (a bf16-level difference upstream flips an activation to the
            # neighbouring e4m3 code, a 6-12 % step of that element: the kernel itself is pinned exactly by
            # test_conv_fp8_x_fp8_mfma_against_bf16_path_on_the_e4m3_grid), against the unquantised model, and both fp8 forms after
            # the timed 50-step decode against the reference's fp32 decode
            "eps_bench_act8": 5.7e-2, "eps_small_act8": 2.8e-2, "eps_vs_unquantised_act8": 0.124,
            "lat_50": 1.74e-2, "wav_50": 4.2e-3, "lat_50_act8": 1.76e-2, "wav_50_act8": 4.4e-3, "wav_c5": 3.5e-2},
}
# f32: 2x a 1e-6-class number would trip on a different reduction order; the floor keeps the f32 bar at 1e-5
_FLOOR = {"f32": 1e-5, "bf16": 0.0, "fp8": 0.0}
TOL = {dt: {k: max(2.0 * v, _FLOOR[dt]) for k, v in d.items()} for dt, d in MEASURED.items()}

_RECORD = os.environ.get("LDC_RECORD_DRIFT")


def check(dtype: str, key: str, value: float, what=""):
    """assert value < TOL[dtype][key]; with LDC_RECORD_DRIFT=<path> record the worst value per key instead."""
    if _RECORD:
        data = {}
        if os.path.exists(_RECORD):
            data = json.load(open(_RECORD))
        cur = data.setdefault(dtype, {})
        cur[key] = max(cur.get(key, 0.0), float(value))
        os.makedirs(os.path.dirname(_RECORD) or ".", exist_ok=True)
        json.dump(data, open(_RECORD, "w"), indent=1, sort_keys=True)
        return
    assert value < TOL[dtype][key], (dtype, key, value, TOL[dtype][key], what)
