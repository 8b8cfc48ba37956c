"""Import the read-only reference (/root/reference) in THIS container only.

Generator-side helper for tools/gen_golden.py.  The reference cannot be imported as
shipped (missing in-repo modules + absent third-party deps), so the
absent modules are registered as inert placeholders before `srcs.model` is imported.
Nothing here is shipped to, or usable on, the GPU box.
"""
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Inert(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


class _ZeroSDR(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, est, tgt):
        return torch.zeros(est.shape[0])


def import_reference():
    """Returns the reference's `srcs.model` module (DiffAudioRep lives there)."""
    if "srcs.model" in sys.modules:
        return sys.modules["srcs.model"]
    ta = _placeholder("torchaudio")
    ta.transforms = _placeholder("torchaudio.transforms", MelSpectrogram=_Inert, Spectrogram=_Inert)
    ta.functional = _placeholder("torchaudio.functional")
    _placeholder("asteroid")
    _placeholder("asteroid.losses")
    _placeholder("asteroid.losses.sdr", MultiSrcNegSDR=_ZeroSDR)
    _placeholder("labml_helpers")
    _placeholder("labml_helpers.module", Module=nn.Module)
    _placeholder("labml_nn")
    _placeholder("labml_nn.diffusion")
    _placeholder("labml_nn.diffusion.ddpm")
    _placeholder("labml_nn.diffusion.ddpm.utils",
                 gather=lambda c, t: c.gather(-1, t).reshape(-1, 1, 1, 1))
    _placeholder("pesq", pesq=None)
    _placeholder("librosa")
    _placeholder("srcs.modules.transformer_discrete", Transformer=object)
    _placeholder("srcs.losses.discrete_diff", AbsorbingDiffusion=object)
    # this repo ships its own `srcs` package (the `python -m srcs.sample` drop-in shim): make sure the name resolves
    # to the reference here
    for name in [n for n in sys.modules if n == "srcs" or (n.startswith("srcs.") and not hasattr(sys.modules[n], "Transformer")
                                                        and not hasattr(sys.modules[n], "AbsorbingDiffusion"))]:
        del sys.modules[name]
    # (the reference's `srcs` has no __init__.py, i.e. it is a namespace package, and a regular package of the same
    # name anywhere on sys.path wins over it: hide those entries while importing)
    import os
    hidden = [p for p in sys.path if os.path.exists(os.path.join(p or ".", "srcs", "__init__.py"))]
    saved_path = list(sys.path)
    sys.path[:] = [REFERENCE_ROOT] + [p for p in sys.path if p not in hidden]
    try:
        import srcs.model as ref_model  # noqa: E402
    finally:
        sys.path[:] = [REFERENCE_ROOT] + saved_path
    assert ref_model.__file__.startswith(REFERENCE_ROOT), ref_model.__file__
    return ref_model
