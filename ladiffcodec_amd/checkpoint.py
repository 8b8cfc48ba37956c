"""`.amlt` checkpoint handling: the counterpart of reference srcs/utils.py:98-108 (`load_model`)."""
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
