import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import *
from ladiffcodec_amd import lib as L, synth
from ladiffcodec_amd.model import Engine
from oracle import ldc_oracle as O
def mk(tag, dtype, strip):
    mc, u, _ = CASES[tag]
    os.environ["LDC_STRIP"] = str(strip)
    e = Engine(mc, u, COND_CFG, dtype=dtype)
    e.load_state_dict(L.MODEL_MAIN, main_sd_np(tag)); e.load_state_dict(L.MODEL_COND, cond_sd_np()); e.finalize(strict=True)
    return e
for tag in ("r84", "r8"):
    g = load_golden("ladiff_" + tag); mc, u, _ = CASES[tag]
    taps = {}
    O.unet_forward(synth.to_torch(main_sd_np(tag)), u, T(g["x"]), torch.full((2,), 37, dtype=torch.long), T(g["cond"]), taps=taps)
    for dtype in ("f32", "bf16"):
        for strip in (0, 2):
            e = mk(tag, dtype, strip)
            eps = e.unet_forward(torch.from_numpy(g["x"]).cuda(), 37, torch.from_numpy(g["cond"]).cuda())
            row = [f"{tag} {dtype} strip={strip} eps {rel_err(eps.cpu().numpy(), g['eps_t37']):.2e}"]
            for n in ["init", "down0", "down1", "down2", "down3", "down4", "mid", "up0", "up1", "up2", "up3", "up4"]:
                row.append(f"{n} {rel_err(e.debug_tap(n, taps[n].shape).cpu().numpy(), taps[n].numpy()):.1e}")
            print(" | ".join(row)); e.close()
