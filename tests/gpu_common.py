"""Engine builders shared by the GPU parity tests and tools/gpu_diag.py."""
import numpy as np

from ladiffcodec_amd import lib as L
from ladiffcodec_amd.model import Engine
from helpers import CASES, COND_CFG, cond_sd_np, main_sd_np

_ENGINES = {}


def engine(tag: str, dtype: str) -> Engine:
    key = (tag, dtype)
    if key not in _ENGINES:
        mc, u, _ = CASES[tag]
        e = Engine(mc, u, COND_CFG, dtype=dtype)
        e.load_state_dict(L.MODEL_MAIN, main_sd_np(tag))
        e.load_state_dict(L.MODEL_COND, cond_sd_np())
        e.finalize(strict=True)
        _ENGINES[key] = e
    return _ENGINES[key]


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
