"""Generate tests/golden/dataset_libri.npz: what the REFERENCE's Dataset_Libri (srcs/dataset_libri.py, imported read-only) returns on
the synthetic LibriSpeech-shaped tree tests/helpers.py builds (libri_tree), under fixed torch seeds.  The tree holds int16 wavs of
different lengths, one silent file, one file shorter than the crop, one with a -32768 sample (int16 abs wraps) and a silent stretch.
Run in the build container only:   python tools/gen_golden_dataset.py"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ref_import import import_reference  # noqa: E402
from helpers import libri_tree  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "dataset_libri.npz")


def main():
    import_reference()
    from srcs.dataset_libri import Dataset_Libri
    out = {}
    with tempfile.TemporaryDirectory() as d:
        libri_tree(d)
        for task in ("train", "valid", "eval"):
            ds = Dataset_Libri(task=task, seq_len_p_sec=0.5, data_folder_path=d)
            ds.files = sorted(ds.files)               # glob order is directory order: pin it for the comparison
            out[f"{task}.n"] = np.array(len(ds))
            torch.manual_seed(1234)
            for i in range(len(ds)):
                out[f"{task}.{i}"] = np.asarray(ds[i], np.float64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) / 1e3, "kB", {k: v.shape for k, v in out.items() if k.endswith(".0")})


if __name__ == "__main__":
    main()
