"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
exactly the entry points include/ladiffcodec.h declares; the host loader mirrors load_model."""
import ctypes
import os
import re
from collections import OrderedDict

import numpy as np
import pytest

from ladiffcodec_amd import checkpoint, lib as L, spec, synth
from helpers import CASES, COND_CFG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(L.LIB_PATH):
        L.build()
    return L.LIB_PATH


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ladiffcodec.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ldc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built):
    dll = ctypes.CDLL(built)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(dll, s), f"{s} declared in include/ladiffcodec.h but not exported"
    assert sorted(L.EXPORTS) == syms
    dll.ldc_version.restype = ctypes.c_char_p
    assert b"gfx950" in dll.ldc_version()      # no compute call: version string only


def test_no_gpu_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ladiffcodec_amd.model import Engine
    mc, u, _ = CASES["r84"]
    with pytest.raises(RuntimeError):
        Engine(mc, u, COND_CFG)


def test_strip_ddp_prefix_matches_load_model():
    sd = OrderedDict([("module.encoder.model.0.conv.conv.bias", 1), ("module.diffusion.model.init_conv.weight", 2)])
    out = checkpoint.strip_ddp_prefix(sd)
    assert list(out) == ["encoder.model.0.conv.conv.bias", "diffusion.model.init_conv.weight"]
    plain = OrderedDict([("encoder.model.0.conv.conv.bias", 1)])
    assert checkpoint.strip_ddp_prefix(plain) == plain


def test_amlt_round_trip(tmp_path):
    mc, u, seed = CASES["r84"]
    sd = synth.ladiff_state_dict(mc, u, seed)
    p = str(tmp_path / "model_best.amlt")
    synth.save_amlt(sd, p, ddp_prefix=True)
    back = checkpoint.read_amlt(p)
    assert list(back) == list(sd)
    for k in sd:
        assert np.array_equal(back[k], sd[k]), k


def test_key_counts_match_survey():
    # LaDiff checkpoint 745 keys (released layout: enc_ratios [8], upsampling
    # [5,4,2], dim 256; 340 of them the UNet), codec 148 keys
    mc = spec.CodecConfig(enc_ratios=(8,))
    u = spec.UnetConfig(dim=256, upsampling_ratios=(5, 4, 2))
    assert len(spec.ladiff_keys(mc, u)) == 745
    assert len(spec.unet_keys(u)) == 340
    assert len(spec.codec_keys(COND_CFG)) == 148


def test_fp8_weight_quantiser_is_ocp_e4m3_round_to_nearest_even(built):
    """The fp8 weight packer's rounding (csrc/conv_gemm.hip host_f32_to_e4m3) against torch.float8_e4m3fn on 300 000 values
    incl. subnormals, ties and the saturation edge."""
    import ctypes as C
    import torch
    lib = L.load()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * 100,
                        np.array([0, 1e-4, 2 ** -10, 2 ** -9, 1.5 * 2 ** -9, 2 ** -6, 447, 448, -448, 0.0009765625 * 1.0001], np.float32),
                        np.linspace(-448, 448, 100001).astype(np.float32)])
    codes, vals = np.empty(x.size, np.uint8), np.empty(x.size, np.float32)
    lib.ldc_quantize_e4m3(x.ctypes.data_as(C.c_void_p), x.size, codes.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p))
    ref = torch.from_numpy(np.clip(x, -448, 448)).to(torch.float8_e4m3fn)
    assert np.array_equal(vals, ref.float().numpy())
    nz = vals != 0
    assert np.array_equal(codes[nz], ref.view(torch.uint8).numpy()[nz])
    big = np.array([1e9, -1e9], np.float32)
    lib.ldc_quantize_e4m3(big.ctypes.data_as(C.c_void_p), 2, None, vals.ctypes.data_as(C.c_void_p))
    assert vals[0] == 448.0 and vals[1] == -448.0            # saturating, as the packer scales every channel to amax = 448
