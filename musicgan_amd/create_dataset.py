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
RING_CHUNKS = 6     # pinned chunks in flight between the copy streams and the writer threads


def host_cpus() -> int:
    """CPUs this process may really use: min(affinity mask, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _nb_samples(nb_frames_wav: int, nb_vec: int) -> int:
    t = 1 + nb_frames_wav // audio.STFT_STRIDE
    return 0 if t < nb_vec else (t - 1) // nb_vec


def _remove_sidecar(folder: str) -> None:
    shards = [os.path.basename(p) for p in glob.glob(join(glob.escape(folder), _ds.PACKED_BIN + ".*"))]
    for name in [_ds.PACKED_META, _ds.PACKED_BIN] + shards:  # meta first: a reader never sees meta without data
        try:
            os.remove(join(folder, name))
        except FileNotFoundError:
            pass


class _Writers:
    """`n` threads that widen float32 samples to float64 and write them as `.pt` files (create_dataset.py:52-62); the first failure
    is kept and re-raised by `close()` / the next `submit()`.  A job is a run of consecutive rows of one pinned chunk.  With a
    template (fast_pt.PtTemplate) the per-sample work -- side-car row, widening, container prefix | payload | suffix -- is ONE native
    call per job (`mg_pt_write_samples`, made without the interpreter lock); the Python left per sample is the 600-byte suffix."""

    JOB_ROWS = 8

    def __init__(self, n: int, template=None):
        self.q: "queue.Queue" = queue.Queue(maxsize=4 * n)
        self.template = template  # fast_pt.PtTemplate (byte-identical th.save output from a template + the payload's CRC) or None
        self.err = None
        self.busy_s = 0.0
        self._lock = threading.Lock()
        self.threads = [threading.Thread(target=self._run, daemon=True) for _ in range(n)]
        for t in self.threads:
            t.start()

    def _run(self):
        import ctypes
        from . import _lib
        while True:
            job = self.q.get()
            if job is None:
                return
            chunk, row, paths, side_fd, side_off, check_first = job
            try:
                if self.err is None:
                    chunk.event.synchronize()  # the chunk's device-to-host copy has landed
                    t0 = time.perf_counter()
                    n = len(paths)
                    if self.template is not None:
                        tp = self.template
                        crcs = [int(v) & 0xFFFFFFFF for v in chunk.crc[row:row + n].tolist()]  # from the GPU, came with the samples
                        if check_first:  # spot check (first sample of a file): the GPU's CRC against zlib's on the same bytes
                            import zlib
                            wide = chunk.host[row].numpy().astype("float64")
                            assert zlib.crc32(wide) & 0xFFFFFFFF == crcs[0], f"device CRC-32 of {paths[0]} disagrees with zlib"
                        suffixes = b"".join(tp.suffix(c) for c in crcs)
                        rows = chunk.host[row:row + n]
                        _lib.check(_lib.load().mg_pt_write_samples(
                            ctypes.c_void_p(rows.data_ptr()), n, rows[0].numel(), b"\0".join(os.fsencode(p) for p in paths) + b"\0",
                            tp.prefix, len(tp.prefix), suffixes, len(suffixes) // n, -1 if side_fd is None else side_fd, side_off),
                            "mg_pt_write_samples")
                    else:
                        for i, path in enumerate(paths):
                            arr = chunk.host[row + i].numpy()
                            if side_fd is not None:  # the float32 side-car row of this sample, at its place in the array file
                                view, done = memoryview(arr).cast("B"), 0
                                while done < len(view):
                                    done += os.pwrite(side_fd, view[done:], side_off + i * arr.nbytes + done)
                            # (numpy, not torch: its intra-op pool under 16 writer threads made `.to(float64)` cost 20-50 ms a sample)
                            th.save(th.from_numpy(arr.astype("float64")), path)
                    with self._lock:
                        self.busy_s += time.perf_counter() - t0
            except BaseException as e:  # noqa: BLE001  (kept for the submitting thread)
                self.err = self.err or e
            finally:
                chunk.release()

    def submit(self, chunk, row, paths, side_fd=None, side_off=0, check_first=False):
        """`paths`: the files of rows row .. row + len(paths) - 1 of `chunk` (side-car rows consecutive from byte `side_off`)."""
        if self.err is not None:
            raise self.err
        chunk.acquire()
        self.q.put((chunk, row, list(paths), side_fd, side_off, check_first))

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.err is not None:
            raise self.err


class _Loader:
    """Reads, page-locks and uploads the wav files ahead of the loop that transforms them (create_dataset.py:34-38 does
    `wav_to_stft(path)` in line): WORKERS threads, worker i taking files i, i + WORKERS, ...; each maps its file, copies the PCM
    frames AS STORED (int16 stays int16: half of float32's bytes; de-interleaving, scaling and the mono mean happen inside the STFT
    kernel) through two pinned staging buffers of its own and queues the host-to-device copies on its own stream, while the main
    thread is busy with the device-to-host copies and writers of earlier files.  (One thread moved a 106 MB float32 file in ~27 ms
    -- page cache -> pinned memory is a single-core copy -- and was the loop's slowest stage once the writers were fixed, r05.)
    Items come out in file order: (path, device PCM tensor (frames, channels) or None, sample rate, ready event)."""

    STAGE_BYTES = 32 << 20  # two pinned staging buffers of this size per worker, page-locked by the worker while the ring is set up
    WORKERS = 3

    def __init__(self, paths, device, workers: int = 0):
        paths = list(paths)
        self.device = device
        self.busy_s = 0.0  # summed over the workers
        self._stop = False
        self._lock = threading.Lock()
        n = max(1, min(workers or self.WORKERS, len(paths)))
        self.workers = n
        self._n_files = len(paths)
        self._qs = [queue.Queue(maxsize=1) for _ in range(n)]
        self._threads = [threading.Thread(target=self._run, args=(paths[i::n], self._qs[i]), daemon=True) for i in range(n)]
        for t in self._threads:
            t.start()

    def _run(self, paths, out_q):
        import numpy as np
        try:
            th.cuda.set_device(self.device)
            pins = [th.empty(self.STAGE_BYTES, dtype=th.uint8).pin_memory() for _ in range(2)]
            stream = th.cuda.Stream(device=self.device)
            events = [None, None]
            tdt = {np.dtype(np.int16): th.int16, np.dtype(np.int32): th.int32, np.dtype(np.uint8): th.uint8,
                   np.dtype(np.float32): th.float32}
            slot = 0
            for path in paths:
                if self._stop:
                    return
                t0 = time.perf_counter()
                pcm, sr = audio.wavio.load_pcm(path)  # a memory map where the format allows: header parsed, no sample read yet
                nbytes = pcm.size * pcm.dtype.itemsize
                with th.cuda.stream(stream):
                    dev = th.empty(nbytes, dtype=th.uint8, device=self.device)
                offset = getattr(pcm, "offset", None)
                fh = open(path, "rb") if offset is not None else None
                flat = None if fh is not None else np.ascontiguousarray(pcm).reshape(-1).view(np.uint8)
                try:
                    if fh is not None:
                        fh.seek(offset)
                    for o in range(0, nbytes, self.STAGE_BYTES):
                        n = min(self.STAGE_BYTES, nbytes - o)
                        if events[slot] is not None:
                            events[slot].synchronize()  # the upload that last read this staging buffer has finished (HOST wait)
                        host = pins[slot][:n]
                        if fh is not None:  # file (page cache) -> pinned memory in one pass, no intermediate array
                            got = fh.readinto(memoryview(host.numpy()))
                            assert got == n, f"short read from {path}"
                        else:
                            host.numpy()[...] = flat[o:o + n]
                        with th.cuda.stream(stream):
                            dev[o:o + n].copy_(host, non_blocking=True)
                            ev = th.cuda.Event()
                            ev.record(stream)
                        events[slot] = ev
                        slot ^= 1
                finally:
                    if fh is not None:
                        fh.close()
                with th.cuda.stream(stream):
                    ready = th.cuda.Event()
                    ready.record(stream)
                shape, dt = (pcm.shape[0], pcm.shape[1]), tdt[pcm.dtype]
                del pcm
                with self._lock:
                    self.busy_s += time.perf_counter() - t0
                out_q.put((path, dev.view(dt).view(*shape), sr, ready))
        except BaseException as e:  # noqa: BLE001  (re-raised by the consumer)
            out_q.put(e)

    def __iter__(self):
        for k in range(self._n_files):  # file k comes from worker k % workers
            item = self._qs[k % self.workers].get()
            if isinstance(item, BaseException):
                raise item
            yield item

    def close(self):
        self._stop = True
        for t, q in zip(self._threads, self._qs):
            while t.is_alive():  # a producer blocked on a full queue needs room to see the flag
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.05)


class _Chunk:
    """One pinned (CHUNK_SAMPLES, 2, 512, nb_vec) float32 buffer of the ring; reusable once every consumer released it."""

    def __init__(self, nb_vec: int, ring: "queue.Queue"):
        self.host = th.empty((CHUNK_SAMPLES, 2, audio.N_FFT // 2, nb_vec), dtype=th.float32).pin_memory()
        self.crc = th.empty((CHUNK_SAMPLES,), dtype=th.int64).pin_memory()  # CRC-32 of each sample's float64 bytes (mg_crc32_f64)
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
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC, before the first HIP call: train.py)
    if not exists(dataset_output_dir):
        os.makedirs(dataset_output_dir, exist_ok=True)
    elif not isdir(dataset_output_dir):
        raise NotADirectoryError(f"\"{dataset_output_dir}\" is not a directory")
    # whatever side-car an earlier run left describes OTHER .pt files than the ones about to be written (same names!)
    _remove_sidecar(dataset_output_dir)
    nb_vec = audio.N_VEC
    if world > 1:
        th.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        counts = []
        for p in w_p:
            data, _ = audio.wavio.load_pcm(p)
            counts.append(_nb_samples(data.shape[0], nb_vec))
    t_setup = time.perf_counter()
    mine = [f_i for f_i in range(len(w_p)) if world == 1 or f_i % world == rank]
    first_idx = {}
    if world > 1:  # global sample numbering: files of other ranks in front of each of mine
        run = 0
        for f_i in range(len(w_p)):
            first_idx[f_i] = run
            run += counts[f_i]
    dev = th.device("cuda", th.cuda.current_device())
    # started first: it reads and uploads the first files while the chunk ring below is being page-locked
    loader = _Loader([w_p[f_i] for f_i in mine], dev, workers=int(os.environ.get("MG_LOADER_THREADS", "0")) or max(1, min(
        _Loader.WORKERS, host_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) // 4)))
    # writer threads: one per CPU this process may use (at most 16) -- divided by the ranks sharing the host: 8 ranks x 16 threads
    # on the cores of one node only take turns (the widen + write path is memory-bandwidth-bound)
    cpus = host_cpus()
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    n_thr = writer_threads or max(1, min(16, cpus // local_world))
    ring: "queue.Queue" = queue.Queue()
    for _ in range(RING_CHUNKS):
        ring.put(_Chunk(nb_vec, ring))
    # device-to-host copies go out on two streams of their own, alternating by chunk: the loop's stream never waits for a copy
    # (it used to carry them, and the per-file `.cpu()` of the CRCs then waited for 421 MB of copies each time)
    copy_streams = [th.cuda.Stream(device=dev) for _ in range(2)]
    n_chunks = 0
    # th.save's output for a (2, 512, nb_vec) float64 tensor as a template (checked against th.save at construction; None: keep
    # calling th.save): the container's CRC-32 then comes from the GPU (mg_crc32_f64) instead of one host core per sample
    template = None
    if os.environ.get("MG_FAST_PT", "1") != "0":
        from .fast_pt import PtTemplate
        template = PtTemplate((2, audio.N_FFT // 2, nb_vec))
        if not template.ok:
            template = None
    writers = _Writers(n_thr, template)
    side = None
    names = []
    if packed and world == 1:
        # written row by row by the writer threads (os.pwrite at row idx): one thread streaming 421 MB per file was the
        # slowest stage of the loop
        # into PACKED_SHARDS files, blocks of PACKED_BLOCK_ROWS rows dealt round-robin: concurrent writes to ONE file take turns on
        # its inode lock (0.25 ms per 2 MiB row = 20 files/s however many threads write)
        side = [os.open(join(dataset_output_dir, _ds.shard_name(k, _ds.PACKED_SHARDS) + ".tmp"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
                for k in range(_ds.PACKED_SHARDS)]
    t_start = time.perf_counter()
    t_setup = t_start - t_setup  # pinned ring (RING_CHUNKS x 64 MiB page-locked) + writer threads: paid once per call
    t_gpu = t_load = t_wait = t_drain = t_loader = 0.0
    row_bytes = 2 * (audio.N_FFT // 2) * nb_vec * 4
    idx = n_files = 0
    ok = False
    try:
        for f_i, (wav_p, pcm, sr, ready) in zip(mine, loader):
            if world > 1:
                idx = first_idx[f_i]
            t0 = time.perf_counter()
            assert sr == audio.SAMPLE_RATE, \
                f"Audio sample rate must be {audio.SAMPLE_RATE}Hz, " \
                f"file \"{wav_p}\" is {sr}Hz"
            th.cuda.current_stream().wait_event(ready)  # the upload (loader's stream) is ordered in front of the STFT
            pcm.record_stream(th.cuda.current_stream())
            complex_values = audio.functions.stft_from_pcm(pcm, nperseg=audio.N_FFT, stride=audio.STFT_STRIDE)
            del pcm
            t1 = time.perf_counter()
            t_load += t1 - t0
            # create_dataset.py:41-42 skips files of fewer than nb_vec frames; a file of EXACTLY nb_vec frames has nb_vec - 1 phase
            # differences, i.e. no complete image either (the reference would write one degenerate (2, 512, 0) tensor for it)
            if complex_values.size()[1] - 1 < nb_vec:
                continue
            both = audio.stft_to_stacked_phase_magn(complex_values, nb_vec=nb_vec)  # (S, 2, 512, nb_vec) float32, on the device
            n_files += 1
            crcs = None
            if template is not None:
                from . import ops
                crcs = ops.crc32_of_float64(both)  # int64 [S] on the device; it travels with the samples, no host sync here
            produced = th.cuda.Event()
            produced.record()  # codec (+ CRC) of this file queued: the copy streams start behind it
            for c0 in range(0, both.size()[0], CHUNK_SAMPLES):
                t2 = time.perf_counter()
                chunk = ring.get()
                t_wait += time.perf_counter() - t2
                n = min(CHUNK_SAMPLES, both.size()[0] - c0)
                chunk.acquire()  # held by this loop until everything that reads the chunk has been queued
                cs = copy_streams[n_chunks % 2]
                n_chunks += 1
                with th.cuda.stream(cs):
                    cs.wait_event(produced)
                    chunk.host[:n].copy_(both[c0:c0 + n], non_blocking=True)
                    if crcs is not None:
                        chunk.crc[:n].copy_(crcs[c0:c0 + n], non_blocking=True)
                        crcs.record_stream(cs)
                    both.record_stream(cs)
                    chunk.event.record(cs)
                r0 = 0
                while r0 < n:
                    # a job = consecutive rows of this chunk inside ONE block of side-car rows (global row = len(names); rows in idx
                    # order == AudioDataset order only after the sort in _finish_sidecar)
                    g_row = len(names)
                    m = min(n - r0, _ds.PACKED_BLOCK_ROWS - g_row % _ds.PACKED_BLOCK_ROWS)
                    batch = [f"magn_phase_{idx + k}.pt" for k in range(m)]
                    fd, off = None, 0
                    if side is not None:
                        k_sh, local = _ds.shard_of_row(g_row, _ds.PACKED_SHARDS, _ds.PACKED_BLOCK_ROWS)
                        fd, off = side[k_sh], local * row_bytes
                    writers.submit(chunk, r0, [join(dataset_output_dir, b) for b in batch], fd, off,
                                   check_first=crcs is not None and c0 + r0 == 0)
                    names.extend(batch)
                    idx += m
                    r0 += m
                chunk.release()
            t_gpu += time.perf_counter() - t1
        t_loader = loader.busy_s
        t_drain = time.perf_counter()
        writers.close()  # the writers finish what is queued
        t_drain = time.perf_counter() - t_drain
        ok = True
    finally:
        loader.close()
        if not ok:
            try:
                writers.close()
            except BaseException:  # noqa: BLE001  (the original error is the one to report)
                pass
        if side is not None:
            for fd in side:
                os.close(fd)
            if not ok:
                _remove_sidecar(dataset_output_dir)
    if side is not None:
        _finish_sidecar(dataset_output_dir, names)
    if stats is not None:
        wall = time.perf_counter() - t_start
        stats.update({"files": n_files, "samples": len(names), "wall_s": wall, "setup_s": t_setup, "drain_s": t_drain,
                      "load_stft_s": t_load, "loader_thread_busy_s": t_loader, "loader_threads": loader.workers,
                      "codec_copy_submit_s": t_gpu, "ring_wait_s": t_wait, "writer_threads": n_thr,
                      "writer_busy_s": writers.busy_s, "pt_bytes": len(names) * 2 * (audio.N_FFT // 2) * nb_vec * 8})


def _finish_sidecar(folder: str, names_in_write_order) -> None:
    """The rows were streamed in write order (idx 0, 1, 2, ...); AudioDataset / the loader index samples in file-NAME order
    (plain string sort, as the reference does: magn_phase_10.pt < magn_phase_2.pt).  The meta file carries the row of every
    sorted name, so the stream never has to be permuted on disk."""
    k = _ds.PACKED_SHARDS
    tmps = [join(folder, _ds.shard_name(i, k) + ".tmp") for i in range(k)]

    def drop():
        for t in tmps:
            try:
                os.remove(t)
            except FileNotFoundError:
                pass
    if not names_in_write_order:
        drop()
        return
    order = sorted(range(len(names_in_write_order)), key=lambda i: names_in_write_order[i])
    files = [names_in_write_order[i] for i in order]
    if tuple(files) != _ds._sample_files(folder):
        # the directory also holds magn_phase_*.pt files of an earlier, longer run: AudioDataset would serve them too (as the
        # reference's does), so no side-car describes this directory -- the loader then takes the reference path
        drop()
        return
    for i, t in enumerate(tmps):
        os.replace(t, join(folder, _ds.shard_name(i, k)))
    with open(join(folder, _ds.PACKED_META), "w") as fh:
        json.dump({"count": len(files), "shape": list(_ds._SAMPLE_SHAPE), "dtype": "float32", "files": files, "rows": order,
                   "shards": k, "block_rows": _ds.PACKED_BLOCK_ROWS,
                   "sizes": [os.path.getsize(join(folder, f)) for f in files],
                   "probes": [_ds.file_probe(join(folder, f)) for f in files]}, fh)
