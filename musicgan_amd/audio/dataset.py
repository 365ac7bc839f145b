"""Datasets over what create_dataset writes.

`AudioDataset(dataset_path)` is the reference's interface (/root/reference/music_gan/audio/dataset.py:14-44): items are the float64
(2, 512, 512) tensors of `magn_phase_{idx}.pt`, in file-name order, each read with `th.load` (4 MiB of pickle per sample -- about
300 samples/s per DataLoader worker, i.e. under 2 000 samples/s with the reference's 6 workers against the 3 900 images/s one
MI355X consumes at level 5).

`PackedAudioDataset` / `PackedLoader` read the optional side-car `create_dataset` writes next to those files: ONE memory-mapped
float32 array `magn_phase_f32.bin` of shape (S, 2, 512, 512) (the stored float64 values are float32 numbers widened -- the codec
computes in float32 -- so nothing is lost) plus `magn_phase_f32.json` (count, shape, the .pt file of every row).  A batch is then
one gather from the page cache into a pinned buffer and one asynchronous upload, double-buffered on a background thread: no
pickle, no worker processes, no float64 on the wire.
"""
from __future__ import annotations

import fnmatch
import json
import os
import queue
import threading
from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset

_PATTERN = "magn_phase_*.pt"
PACKED_BIN, PACKED_META = "magn_phase_f32.bin", "magn_phase_f32.json"
_SAMPLE_SHAPE = (2, 512, 512)


def _sample_files(folder: str):
    with os.scandir(folder) as it:
        names = [e.name for e in it if e.is_file() and fnmatch.fnmatchcase(e.name, _PATTERN)
                 and e.name[len("magn_phase_"):-len(".pt")].isdigit()]
    names.sort()  # plain string order, as the reference sorts them
    return tuple(names)


class AudioDataset(Dataset):
    def __init__(self, dataset_path: str) -> None:
        super().__init__()
        assert os.path.isdir(dataset_path)
        self._root = dataset_path
        self._names = _sample_files(dataset_path)

    def __len__(self) -> int:
        return len(self._names)

    def __getitem__(self, index: int) -> torch.Tensor:
        return torch.load(os.path.join(self._root, self._names[index]))


# ---------------------------------------------------------------------------------------------------------------- packed side-car
def write_packed(dataset_path: str) -> int:
    """(Re)build the side-car from the `magn_phase_{idx}.pt` files of a dataset directory, rows in AudioDataset order (so index i
    of either dataset is the same sample).  Returns the number of samples."""
    names = _sample_files(dataset_path)
    if not names:
        return 0
    path = os.path.join(dataset_path, PACKED_BIN)
    mm = np.memmap(path + ".tmp", dtype=np.float32, mode="w+", shape=(len(names),) + _SAMPLE_SHAPE)
    for i, n in enumerate(names):
        x = torch.load(os.path.join(dataset_path, n))
        assert tuple(x.shape) == _SAMPLE_SHAPE, f"{n}: shape {tuple(x.shape)}"
        f = x.to(torch.float32)
        assert torch.equal(f.to(x.dtype), x), f"{n}: values are not float32-representable"
        mm[i] = f.numpy()
    mm.flush()
    del mm
    os.replace(path + ".tmp", path)
    with open(os.path.join(dataset_path, PACKED_META), "w") as fh:
        json.dump({"count": len(names), "shape": list(_SAMPLE_SHAPE), "dtype": "float32", "files": list(names)}, fh)
    return len(names)


def has_packed(dataset_path: str) -> bool:
    """True when the side-car exists AND still matches the directory's .pt files (same names, same order)."""
    meta = os.path.join(dataset_path, PACKED_META)
    if not (os.path.exists(meta) and os.path.exists(os.path.join(dataset_path, PACKED_BIN))):
        return False
    with open(meta) as fh:
        m = json.load(fh)
    return tuple(m.get("files", ())) == _sample_files(dataset_path)


class PackedAudioDataset(Dataset):
    """Same items, same order as AudioDataset, served from the memory-mapped float32 side-car (items are float32)."""

    def __init__(self, dataset_path: str) -> None:
        super().__init__()
        assert has_packed(dataset_path), f"no valid {PACKED_BIN} in \"{dataset_path}\" (musicgan_amd.audio.dataset.write_packed)"
        with open(os.path.join(dataset_path, PACKED_META)) as fh:
            meta = json.load(fh)
        self._count = int(meta["count"])
        self._mm = np.memmap(os.path.join(dataset_path, PACKED_BIN), dtype=np.float32, mode="r",
                             shape=(self._count,) + _SAMPLE_SHAPE)

    def __len__(self) -> int:
        return self._count

    def __getitem__(self, index: int) -> torch.Tensor:
        return torch.from_numpy(np.array(self._mm[index]))

    def gather(self, indices: Sequence[int], out: torch.Tensor) -> torch.Tensor:
        """out[k] = sample indices[k]; `out` is a (len(indices), 2, 512, 512) float32 host tensor (pinned for async upload)."""
        dst = out.numpy()
        for k, i in enumerate(indices):
            dst[k] = self._mm[i]
        return out


class PackedLoader:
    """Batches of a PackedAudioDataset already on the device: a background thread gathers batch b+1 into one of `depth` pinned
    buffers and queues its upload on a side stream while the caller trains on batch b.  Iterating yields float32 device tensors
    (B, 2, 512, 512) in sampler order with the tail dropped (the reference's drop_last=True)."""

    def __init__(self, dataset: PackedAudioDataset, batch_size: int, sampler, device, depth: int = 3):
        self.ds, self.bs, self.sampler, self.device, self.depth = dataset, batch_size, sampler, torch.device(device), depth
        self._pinned = [torch.empty((batch_size,) + _SAMPLE_SHAPE, dtype=torch.float32).pin_memory() for _ in range(depth)]
        self._dev = [torch.empty((batch_size,) + _SAMPLE_SHAPE, dtype=torch.float32, device=self.device) for _ in range(depth)]
        self._stream = torch.cuda.Stream(device=self.device)

    def __len__(self) -> int:
        return len(self.sampler) // self.bs

    def __iter__(self) -> Iterator[torch.Tensor]:
        idx: List[int] = list(iter(self.sampler))
        nb = len(idx) // self.bs
        ready: "queue.Queue" = queue.Queue(maxsize=self.depth - 1)
        free: "queue.Queue" = queue.Queue()
        for s in range(self.depth):
            free.put(s)
        stop = threading.Event()

        def producer():
            torch.cuda.set_device(self.device)
            for b in range(nb):
                slot = free.get()
                if stop.is_set():
                    return
                self.ds.gather(idx[b * self.bs:(b + 1) * self.bs], self._pinned[slot])
                with torch.cuda.stream(self._stream):
                    self._dev[slot].copy_(self._pinned[slot], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                ready.put((slot, ev))
            ready.put(None)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        prev: Optional[int] = None
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                slot, ev = item
                torch.cuda.current_stream(self.device).wait_event(ev)
                if prev is not None:  # the consumer moved on: its buffer may be refilled once the work queued on it has run
                    done = torch.cuda.Event()
                    done.record(torch.cuda.current_stream(self.device))
                    self._stream.wait_event(done)
                    free.put(prev)
                prev = slot
                yield self._dev[slot]
        finally:
            stop.set()
            free.put(-1)
            th.join(timeout=5)
