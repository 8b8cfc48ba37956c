"""`.amlt` checkpoint handling: the counterpart of reference srcs/utils.py:85-108 (`save_checkpoints`, `load_model`)."""
from __future__ import annotations

import re
from collections import OrderedDict
from typing import Dict

import numpy as np


def strip_ddp_prefix(state_dict: Dict[str, object]) -> "OrderedDict[str, object]":
    """utils.py:101-107: every occurrence of the pattern `module.` is removed from keys that contain
    'module' (DDP-wrapped checkpoints); other checkpoints pass through untouched."""
    out = OrderedDict()
    pattern = re.compile("module.")
    any_module = any(re.search("module", k) for k in state_dict)
    if not any_module:
        return OrderedDict(state_dict)
    for k, v in state_dict.items():
        out[re.sub(pattern, "", k) if re.search("module", k) else k] = v
    return out


def read_amlt(path: str) -> "OrderedDict[str, np.ndarray]":
    """torch.load(path) -> {key: float32 ndarray}.  The reference calls torch.load without
    map_location (utils.py:100); tensors are brought to the host here because the library folds and
    packs weights from host memory."""
    import torch
    sd = torch.load(path, map_location="cpu")
    sd = strip_ddp_prefix(sd)
    return OrderedDict((k, v.detach().to(torch.float32).contiguous().numpy()) for k, v in sd.items())


def merged_state_dict(base_sd: Dict[str, object], trained: Dict[str, object]) -> "OrderedDict[str, object]":
    """The main model's full state dict with the trained diffusion UNet written back: `trained` is keyed like
    DiffusionTrainer.state_dict() (the keys of model.diff_model); every `diff_model.<k>` AND its alias `diffusion.model.<k>`
    (GaussianDiffusion1D holds the same module: both prefixes are in the reference's model.state_dict()) get the new tensor,
    everything else (frozen codec, schedule buffers) is carried over.  Unknown keys in `trained` are an error."""
    import torch
    out = OrderedDict()
    left = dict(trained)
    for k, v in base_sd.items():
        for prefix in ("diff_model.", "diffusion.model."):
            if k.startswith(prefix) and k[len(prefix):] in trained:
                v = trained[k[len(prefix):]]
                left.pop(k[len(prefix):], None)
                break
        out[k] = v.detach().cpu().clone() if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))
    if left:
        raise KeyError(f"trained keys without a counterpart in the base state dict: {sorted(left)[:5]}")
    return out


def save_checkpoints(state_dict: Dict[str, object], output_dir: str, exp_name: str, note: str = "") -> str:
    """utils.py:85-95 for the model file: `torch.save(model.state_dict(), f'{output_dir}/{exp_name}/model_{note}.amlt')`, the
    directory created on demand (the reference's ema_ / disc_ files belong to components that are out of scope here).  `note` is
    'best' when the monitored neg_loss improves and str(step) every 100 outer steps (train.py:410-414).  Returns the path."""
    import os
    import torch
    directory = f"{output_dir}/{exp_name}"
    if not os.path.exists(directory):
        os.makedirs(directory)
    path = f"{output_dir}/{exp_name}/model_{note}.amlt"
    torch.save(OrderedDict((k, v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in state_dict.items()), path)
    return path
