"""`create_dataset(audio_path_glob, dataset_output_dir)` (/root/reference/music_gan/create_dataset.py:13-64): wav files ->
STFT -> (magnitude, phase-delta) images -> `magn_phase_{idx}.pt` (float64, shape (2,512,512)), STFT and codec on the GPU.
Under torchrun the files are dealt round-robin to the ranks (independent units, no collective); the global sample numbering
stays the reference's (files in glob order) because each file's sample count follows from its length alone.

The GPU side of a 10-minute file is ~0.5 ms; what the loop spends its time on is the 843 MB of float64 `.pt` files it has to
leave behind (201 samples x 4 MiB, create_dataset.py:52-62).  So the loop is a pipeline: the codec writes the stacked
(S, 2, 512, 512) float32 tensor in one pass (`mg_codec_fwd_strided`), it comes back in chunks through a ring of pinned buffers
(float32 on the wire: half the bytes of the reference's `.to(th.float64).cpu()`), a pool of writer threads widens each sample to
float64 and `th.save`s it while the GPU and the copy engine work on the next chunk / file, and the float32 side-car of the fast
loader (audio/dataset.py) is streamed out of the same pinned chunks instead of re-reading every `.pt` afterwards.
`stats` (optional dict) receives where the time went: bench.py's `create_dataset_e2e` record.
"""
import glob
import json
import os
import queue
import threading
import time
from os.path import exists, isdir, join

import torch as th

from . import audio
from .audio import dataset as _ds

CHUNK_SAMPLES = 32  # samples per pinned chunk (64 MiB of float32)


def _nb_samples(nb_frames_wav: int, nb_vec: int) -> int:
    t = 1 + nb_frames_wav // audio.STFT_STRIDE
    return 0 if t < nb_vec else (t - 1) // nb_vec


def _remove_sidecar(folder: str) -> None:
    for name in (_ds.PACKED_META, _ds.PACKED_BIN, _ds.PACKED_BIN + ".tmp"):  # meta first: a reader never sees meta without data
        try:
            os.remove(join(folder, name))
        except FileNotFoundError:
            pass


class _Writers:
    """`n` threads that widen a float32 sample to float64 and th.save it (create_dataset.py:52-62); the first failure is kept and
    re-raised by `close()` / the next `submit()`."""

    def __init__(self, n: int):
        self.q: "queue.Queue" = queue.Queue(maxsize=4 * n)
        self.err = None
        self.busy_s = 0.0
        self._lock = threading.Lock()
        self.threads = [threading.Thread(target=self._run, daemon=True) for _ in range(n)]
        for t in self.threads:
            t.start()

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                return
            chunk, row, path = job
            try:
                if self.err is None:
                    chunk.event.synchronize()  # the chunk's device-to-host copy has landed
                    t0 = time.perf_counter()
                    th.save(chunk.host[row].to(th.float64), path)
                    with self._lock:
                        self.busy_s += time.perf_counter() - t0
            except BaseException as e:  # noqa: BLE001  (kept for the submitting thread)
                self.err = self.err or e
            finally:
                chunk.release()

    def submit(self, chunk, row, path):
        if self.err is not None:
            raise self.err
        chunk.acquire()
        self.q.put((chunk, row, path))

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.err is not None:
            raise self.err


class _Chunk:
    """One pinned (CHUNK_SAMPLES, 2, 512, nb_vec) float32 buffer of the ring; reusable once every consumer released it."""

    def __init__(self, nb_vec: int, ring: "queue.Queue"):
        self.host = th.empty((CHUNK_SAMPLES, 2, audio.N_FFT // 2, nb_vec), dtype=th.float32).pin_memory()
        self.event = th.cuda.Event()
        self._ring, self._refs, self._lock = ring, 0, threading.Lock()

    def acquire(self):
        with self._lock:
            self._refs += 1

    def release(self):
        with self._lock:
            self._refs -= 1
            done = self._refs == 0
        if done:
            self._ring.put(self)


def create_dataset(audio_path: str, dataset_output_dir: str, *, packed: bool = True, writer_threads: int = 0,
                   stats: dict = None) -> None:
    """`packed` (extension, single-process runs): also write the float32 memory-mapped side-car the fast loader reads
    (audio/dataset.py); the reference-format `magn_phase_{idx}.pt` files are written either way.  `writer_threads`: 0 = one per
    available CPU (at most 16)."""
    w_p = glob.glob(audio_path)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if not exists(dataset_output_dir):
        os.makedirs(dataset_output_dir, exist_ok=True)
    elif not isdir(dataset_output_dir):
        raise NotADirectoryError(f"\"{dataset_output_dir}\" is not a directory")
    # whatever side-car an earlier run left describes OTHER .pt files than the ones about to be written (same names!)
    _remove_sidecar(dataset_output_dir)
    nb_vec = audio.N_VEC
    if world > 1:
        th.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        from scipy.io import wavfile
        counts = []
        for p in w_p:
            _, data = wavfile.read(p, mmap=True)
            counts.append(_nb_samples(data.shape[0], nb_vec))
    t_setup = time.perf_counter()
    n_thr = writer_threads or max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4))
    ring: "queue.Queue" = queue.Queue()
    for _ in range(4):
        ring.put(_Chunk(nb_vec, ring))
    writers = _Writers(n_thr)
    side = None
    names = []
    if packed and world == 1:
        side = open(join(dataset_output_dir, _ds.PACKED_BIN + ".tmp"), "wb")
    t_start = time.perf_counter()
    t_setup = t_start - t_setup  # pinned ring (4 x 64 MiB page-locked) + writer threads: paid once per call
    t_gpu = t_load = t_wait = t_drain = 0.0
    idx = n_files = 0
    ok = False
    try:
        for f_i, wav_p in enumerate(w_p):
            if world > 1 and f_i % world != rank:
                idx += counts[f_i]
                continue
            t0 = time.perf_counter()
            complex_values = audio.wav_to_stft(wav_p, nperseg=audio.N_FFT, stride=audio.STFT_STRIDE)
            t1 = time.perf_counter()
            t_load += t1 - t0
            # create_dataset.py:41-42 skips files of fewer than nb_vec frames; a file of EXACTLY nb_vec frames has nb_vec - 1 phase
            # differences, i.e. no complete image either (the reference would write one degenerate (2, 512, 0) tensor for it)
            if complex_values.size()[1] - 1 < nb_vec:
                continue
            both = audio.stft_to_stacked_phase_magn(complex_values, nb_vec=nb_vec)  # (S, 2, 512, nb_vec) float32, on the device
            n_files += 1
            for c0 in range(0, both.size()[0], CHUNK_SAMPLES):
                t2 = time.perf_counter()
                chunk = ring.get()
                t_wait += time.perf_counter() - t2
                n = min(CHUNK_SAMPLES, both.size()[0] - c0)
                chunk.acquire()  # held by this loop until everything that reads the chunk has been queued
                chunk.host[:n].copy_(both[c0:c0 + n], non_blocking=True)
                chunk.event.record()
                for r in range(n):
                    name = f"magn_phase_{idx}.pt"
                    writers.submit(chunk, r, join(dataset_output_dir, name))
                    names.append(name)
                    idx += 1
                if side is not None:  # rows in idx order == AudioDataset order only after the sort below; see _finish_sidecar
                    chunk.event.synchronize()
                    side.write(memoryview(chunk.host[:n].numpy()).cast("B"))
                chunk.release()
            t_gpu += time.perf_counter() - t1
        t_drain = time.perf_counter()
        writers.close()  # the writers finish what is queued
        t_drain = time.perf_counter() - t_drain
        ok = True
    finally:
        if not ok:
            try:
                writers.close()
            except BaseException:  # noqa: BLE001  (the original error is the one to report)
                pass
        if side is not None:
            side.close()
            if not ok:
                _remove_sidecar(dataset_output_dir)
    if side is not None:
        _finish_sidecar(dataset_output_dir, names)
    if stats is not None:
        wall = time.perf_counter() - t_start
        stats.update({"files": n_files, "samples": len(names), "wall_s": wall, "setup_s": t_setup, "drain_s": t_drain,
                      "load_stft_s": t_load,
                      "codec_copy_submit_s": t_gpu, "ring_wait_s": t_wait, "writer_threads": n_thr,
                      "writer_busy_s": writers.busy_s, "pt_bytes": len(names) * 2 * (audio.N_FFT // 2) * nb_vec * 8})


def _finish_sidecar(folder: str, names_in_write_order) -> None:
    """The rows were streamed in write order (idx 0, 1, 2, ...); AudioDataset / the loader index samples in file-NAME order
    (plain string sort, as the reference does: magn_phase_10.pt < magn_phase_2.pt).  The meta file carries the row of every
    sorted name, so the stream never has to be permuted on disk."""
    tmp = join(folder, _ds.PACKED_BIN + ".tmp")
    if not names_in_write_order:
        os.remove(tmp)
        return
    order = sorted(range(len(names_in_write_order)), key=lambda i: names_in_write_order[i])
    files = [names_in_write_order[i] for i in order]
    if tuple(files) != _ds._sample_files(folder):
        # the directory also holds magn_phase_*.pt files of an earlier, longer run: AudioDataset would serve them too (as the
        # reference's does), so no side-car describes this directory -- the loader then takes the reference path
        os.remove(tmp)
        return
    os.replace(tmp, join(folder, _ds.PACKED_BIN))
    with open(join(folder, _ds.PACKED_META), "w") as fh:
        json.dump({"count": len(files), "shape": list(_ds._SAMPLE_SHAPE), "dtype": "float32", "files": files, "rows": order,
                   "sizes": [os.path.getsize(join(folder, f)) for f in files],
                   "probes": [_ds.file_probe(join(folder, f)) for f in files]}, fh)
