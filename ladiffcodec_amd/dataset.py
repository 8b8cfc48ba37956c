"""LibriSpeech-shaped dataset walker (second half): the reference's `Dataset_Libri`
(srcs/dataset_libri.py:13-94) and the batch hand-off of its training loop (srcs/train.py:113-115, 335-336), for the training row's
`DiffusionTrainer.step_from_wav(wav, next_wav=...)`.

`DatasetLibri` follows the reference item by item, including what is easy to get wrong:
  * files = glob(<root>/train-clean-100/*/*/*.wav)[:10000] ('train') or <root>/dev-clean/... ('valid' / 'eval'), in glob order;
  * every file is peak-normalised over the WHOLE file before cropping: x / max(|x| + 1e-20) on the int16 samples scipy returns (so
    |-32768| wraps to -32768 exactly as numpy's int16 abs does in the reference);
  * 'eval': the first seq_len seconds (shorter files come back shorter, unpadded);
  * otherwise: files shorter than the crop or silent (std ~ 0) are replaced by the next index; the crop start is
    torch.randint(len - seq_len, (1,)) -- one draw from the GLOBAL torch generator per attempt -- and silent crops are redrawn.
`BatchWalker` is the DataLoader(batch_size, pin_memory=True) + `batch.unsqueeze(1).to(torch.float).to(device)` of the reference as one
object: float32 [B, 1, T] batches staged through pinned host memory and uploaded on a side stream one batch ahead, so that the
trainer's prefetch of the frozen encoders has its audio before the current step's launches are queued.  No GPU code of its own: the
waveform is 0.6 MB per batch; the work worth a kernel (the encoders) is the trainer's."""
import glob
from typing import Iterator, List, Optional

import numpy as np


def read_wav_int(path: str):
    """scipy.io.wavfile.read semantics for the files LibriSpeech-as-wav holds: (sample_rate, int16 / float array)."""
    import scipy.io.wavfile as wavfile
    return wavfile.read(path)


class DatasetLibri:
    def __init__(self, task: str = "train", seq_len_p_sec: float = 5, data_folder_path: str = "/data/hy17/librispeech/librispeech",
                 max_files: int = 10000):
        self.task = task
        self.seq_len_p_sec = seq_len_p_sec
        if task == "train":
            path = data_folder_path + "/train-clean-100/*/*/*.wav"
        elif task in ("valid", "eval"):
            path = data_folder_path + "/dev-clean/*/*/*.wav"
        else:
            raise ValueError(f"task must be 'train', 'valid' or 'eval', got {task!r}")   # (the reference leaves `path` unbound here)
        self.files: List[str] = glob.glob(path)[:max_files]

    def __len__(self) -> int:
        return len(self.files)

    @staticmethod
    def normalize_data(x: np.ndarray) -> np.ndarray:
        return x / (np.abs(x) + 1e-20).max()        # dataset_libri.py:47-51 (python max over the array = numpy max)

    def _load(self, idx: int) -> np.ndarray:
        _, x = read_wav_int(self.files[idx])
        return self.normalize_data(x)

    def __getitem__(self, idx: int) -> np.ndarray:
        import torch
        in_data = self._load(idx)
        seq_length = int(self.seq_len_p_sec * 16000)
        if self.task == "eval":
            return in_data[0:seq_length]
        while len(in_data) < seq_length or np.isclose(np.std(in_data), 0):
            idx = (idx + 1) % len(self)
            in_data = self._load(idx)
        while True:
            loc = 0 if len(in_data) == seq_length else int(torch.randint(len(in_data) - seq_length, (1,)))
            seg = in_data[loc: loc + seq_length]
            if not np.isclose(np.std(seg), 0):      # exclude empty samples
                return seg


class BatchWalker:
    """for wav in BatchWalker(ds, batch_size, device): ...   ->  float32 [B, 1, T] on `device`, sequential order (the reference's
    DataLoader without a sampler), last partial batch kept.  `peek()` returns the batch after the current one (already uploading)."""

    def __init__(self, dataset: DatasetLibri, batch_size: int, device=None, indices: Optional[List[int]] = None):
        import torch
        self.ds, self.bs, self.torch = dataset, int(batch_size), torch
        self.device = device
        self.indices = list(range(len(dataset))) if indices is None else list(indices)
        self._cuda = device is not None and torch.device(device).type == "cuda"
        self._stream = torch.cuda.Stream(device=device) if self._cuda else None
        self._next = None
        self._pos = 0

    def __len__(self) -> int:
        return (len(self.indices) + self.bs - 1) // self.bs

    def _stage(self):
        t = self.torch
        if self._pos >= len(self.indices):
            return None
        idx = self.indices[self._pos:self._pos + self.bs]
        self._pos += len(idx)
        segs = [np.asarray(self.ds[i]) for i in idx]
        host = t.from_numpy(np.stack(segs)).unsqueeze(1).to(t.float)          # train.py:115
        if not self._cuda:
            return host if self.device is None else host.to(self.device)
        host = host.pin_memory()
        with t.cuda.stream(self._stream):
            dev = host.to(self.device, non_blocking=True)
            ev = t.cuda.Event()
            ev.record(self._stream)
        return (dev, ev, host)

    def _ready(self, staged):
        if staged is None or not self._cuda:
            return staged
        dev, ev, _ = staged
        self.torch.cuda.current_stream(self.device).wait_event(ev)
        dev.record_stream(self.torch.cuda.current_stream(self.device))
        return dev

    def __iter__(self) -> Iterator:
        self._pos = 0
        self._next = self._stage()
        while self._next is not None:
            cur, self._next = self._next, self._stage()
            yield self._ready(cur)

    def peek(self):
        """the batch the next iteration will yield (None at the end); safe to hand to DiffusionTrainer.step_from_wav(next_wav=...)"""
        return self._ready(self._next)
