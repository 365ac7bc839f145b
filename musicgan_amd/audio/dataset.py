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
# The array may be split over `shards` files (meta "shards" / "block_rows"; 1 = the single file PACKED_BIN): row r lives in shard
# (r // B) % K at local row (r // (B * K)) * B + r % B.  create_dataset's writer threads fill it with concurrent pwrites, and writes
# to ONE file serialise on its inode lock (measured: 0.25 ms per 2 MiB row, i.e. 20 files/s whatever the thread count).
PACKED_SHARDS, PACKED_BLOCK_ROWS = 16, 8


def shard_name(k: int, shards: int) -> str:
    return PACKED_BIN if shards == 1 else f"{PACKED_BIN}.{k}"


def shard_of_row(r: int, shards: int, block_rows: int):
    """(shard, local row) of global row r."""
    b = r // block_rows
    return b % shards, (b // shards) * block_rows + r % block_rows


def shard_rows(count: int, shards: int, block_rows: int):
    """Rows each shard holds for `count` global rows."""
    out = [0] * shards
    full, rest = divmod(count, block_rows)
    for k in range(shards):
        out[k] = (full // shards + (1 if k < full % shards else 0)) * block_rows
    if rest:
        out[full % shards] += rest
    return out


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
def file_probe(path: str) -> int:
    """CRC-32 of 4 KiB from the middle of a sample file (tensor payload).  Every valid sample has the same size, so sizes alone
    cannot tell a regenerated dataset from the one the side-car was built from; this does, at one small read per file."""
    import zlib
    size = os.path.getsize(path)
    with open(path, "rb") as fh:
        fh.seek(max(0, size // 2 - 2048))
        return zlib.crc32(fh.read(4096))


def write_packed(dataset_path: str) -> int:
    """(Re)build the side-car from the `magn_phase_{idx}.pt` files of a dataset directory, rows in AudioDataset order (so index i
    of either dataset is the same sample).  Returns the number of samples."""
    names = _sample_files(dataset_path)
    if not names:
        return 0
    path = os.path.join(dataset_path, PACKED_BIN)
    try:
        os.remove(os.path.join(dataset_path, PACKED_META))  # never leave a meta file describing a half-written array
    except FileNotFoundError:
        pass
    try:
        mm = np.memmap(path + ".tmp", dtype=np.float32, mode="w+", shape=(len(names),) + _SAMPLE_SHAPE)
        for i, n in enumerate(names):
            x = torch.load(os.path.join(dataset_path, n))
            assert tuple(x.shape) == _SAMPLE_SHAPE, f"{n}: shape {tuple(x.shape)}"
            f = x.to(torch.float32)
            assert torch.equal(f.to(x.dtype), x), f"{n}: values are not float32-representable"
            mm[i] = f.numpy()
        mm.flush()
        del mm
    except BaseException:
        try:
            os.remove(path + ".tmp")
        except FileNotFoundError:
            pass
        raise
    os.replace(path + ".tmp", path)
    with open(os.path.join(dataset_path, PACKED_META), "w") as fh:
        json.dump({"count": len(names), "shape": list(_SAMPLE_SHAPE), "dtype": "float32", "files": list(names),
                   "rows": list(range(len(names))), "sizes": [os.path.getsize(os.path.join(dataset_path, n)) for n in names],
                   "probes": [file_probe(os.path.join(dataset_path, n)) for n in names]}, fh)
    return len(names)


def _read_meta(dataset_path: str):
    meta = os.path.join(dataset_path, PACKED_META)
    if not os.path.exists(meta):
        return None
    with open(meta) as fh:
        m = json.load(fh)
    k = int(m.get("shards", 1))
    if not all(os.path.exists(os.path.join(dataset_path, shard_name(i, k))) for i in range(k)):
        return None
    return m


def has_packed(dataset_path: str) -> bool:
    """True when the side-car exists AND still matches the directory's .pt files: same names in the same order, the same file
    sizes, the same CRC of a 4 KiB probe from the middle of every file, and an array file of the recorded length.  (create_dataset
    removes the side-car before it rewrites any .pt file, so a side-car never outlives the files it was built from; every valid
    sample has the same size, so it is the probe that catches files replaced behind its back by another tool.)"""
    m = _read_meta(dataset_path)
    if m is None:
        return False
    names = _sample_files(dataset_path)
    if tuple(m.get("files", ())) != names:
        return False
    if "sizes" in m and list(m["sizes"]) != [os.path.getsize(os.path.join(dataset_path, n)) for n in names]:
        return False
    if "probes" in m and list(m["probes"]) != [file_probe(os.path.join(dataset_path, n)) for n in names]:
        return False
    k, b = int(m.get("shards", 1)), int(m.get("block_rows", PACKED_BLOCK_ROWS))
    row_bytes = int(np.prod(_SAMPLE_SHAPE)) * 4
    sizes_ok = all(os.path.getsize(os.path.join(dataset_path, shard_name(i, k))) == n * row_bytes
                   for i, n in enumerate(shard_rows(int(m["count"]), k, b)))
    return sizes_ok and sorted(m.get("rows", range(len(names)))) == list(range(len(names)))


class PackedAudioDataset(Dataset):
    """Same items, same order as AudioDataset, served from the memory-mapped float32 side-car (items are float32).  Item i (file
    names in the reference's plain string order) lives in row `rows[i]` of the array: create_dataset streams the rows out in the
    order it produces them (idx 0, 1, 2, ...), which is not the string order of their names."""

    def __init__(self, dataset_path: str) -> None:
        super().__init__()
        assert has_packed(dataset_path), f"no valid {PACKED_BIN} in \"{dataset_path}\" (musicgan_amd.audio.dataset.write_packed)"
        meta = _read_meta(dataset_path)
        self._count = int(meta["count"])
        self._rows = np.asarray(meta.get("rows", range(self._count)), dtype=np.int64)
        self._k, self._b = int(meta.get("shards", 1)), int(meta.get("block_rows", PACKED_BLOCK_ROWS))
        self._mm = [np.memmap(os.path.join(dataset_path, shard_name(i, self._k)), dtype=np.float32, mode="r", shape=(n,) + _SAMPLE_SHAPE)
                    if n else None for i, n in enumerate(shard_rows(self._count, self._k, self._b))]

    def __len__(self) -> int:
        return self._count

    def _row(self, index: int):
        k, local = shard_of_row(int(self._rows[index]), self._k, self._b)
        return self._mm[k][local]

    def __getitem__(self, index: int) -> torch.Tensor:
        return torch.from_numpy(np.array(self._row(index)))

    def gather(self, indices: Sequence[int], out: torch.Tensor) -> torch.Tensor:
        """out[k] = sample indices[k]; `out` is a (len(indices), 2, 512, 512) float32 host tensor (pinned for async upload)."""
        dst = out.numpy()
        for k, i in enumerate(indices):
            dst[k] = self._row(i)
        return out


class PackedLoader:
    """Batches of a PackedAudioDataset already on the device: a background thread gathers batch b+1 into one of `depth` pinned
    buffers and queues its upload on a side stream while the caller trains on batch b.  Iterating yields float32 device tensors
    (B, 2, 512, 512) in sampler order with the tail dropped (the reference's drop_last=True).

    Buffer life cycle (slot s = pinned[s] + dev[s]):  the consumer frees a slot with STREAM-level ordering only (its kernels on
    dev[s] are queued, not finished), so the upload into dev[s] waits on that event on the loader stream -- and therefore may sit
    in the queue long after the producer got the slot back.  The producer must not gather into pinned[s] while an EARLIER upload
    out of pinned[s] is still pending: it waits on the HOST for that upload's event first (a training loop under graph replay
    has no host synchronisation of its own and runs many steps ahead of the GPU)."""

    def __init__(self, dataset: PackedAudioDataset, batch_size: int, sampler, device, depth: int = 3):
        self.ds, self.bs, self.sampler, self.device, self.depth = dataset, batch_size, sampler, torch.device(device), depth
        self._pinned = [torch.empty((batch_size,) + _SAMPLE_SHAPE, dtype=torch.float32).pin_memory() for _ in range(depth)]
        self._dev = [torch.empty((batch_size,) + _SAMPLE_SHAPE, dtype=torch.float32, device=self.device) for _ in range(depth)]
        self._uploaded: List[Optional[torch.cuda.Event]] = [None] * depth  # last upload out of pinned[s] (kept across epochs)
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

        def post(item) -> bool:  # ready.put that gives up when the consumer has left
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def producer():
            try:
                torch.cuda.set_device(self.device)
                for b in range(nb):
                    slot = free.get()
                    if stop.is_set() or slot < 0:
                        return
                    if self._uploaded[slot] is not None:
                        self._uploaded[slot].synchronize()  # HOST wait: the previous upload has read pinned[slot]
                    self.ds.gather(idx[b * self.bs:(b + 1) * self.bs], self._pinned[slot])
                    with torch.cuda.stream(self._stream):
                        self._dev[slot].copy_(self._pinned[slot], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self._stream)
                    self._uploaded[slot] = ev
                    if not post((slot, ev)):
                        return
                post(None)
            except BaseException as e:  # noqa: BLE001  (I/O error on the memmap, HIP error: re-raised by the consumer)
                post(e)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        prev: Optional[int] = None
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
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
            while True:  # a producer blocked in ready.put returns as soon as there is room or it sees `stop`
                try:
                    ready.get_nowait()
                except queue.Empty:
                    break
            th.join(timeout=5)
