"""create_dataset on eight synthetic 10-minute wav files, repeated with 1..4 loader threads (MG_LOADER_THREADS): files/s after set-up
and the loop's split.   python tools/bench_create_dataset.py [nfiles] [reps]"""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import create_dataset
from musicgan_amd.audio import wavio

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tmp = tempfile.mkdtemp(prefix="mg_cd_")
try:
    wav = os.path.join(tmp, "wav")
    os.mkdir(wav)
    g = torch.Generator().manual_seed(7)
    for i in range(nfiles):
        wavio.save(os.path.join(wav, f"track_{i}.wav"), torch.rand(1, 44100 * 600, generator=g) - 0.5, 44100)
    print("loader_threads rep files/s_after_setup wall setup loader_busy load_wait codec_d2h ring_wait drain writer_busy", flush=True)
    for thr in (3, 1, 3, 1, 2, 1, 3):
        os.environ["MG_LOADER_THREADS"] = str(thr)
        for r in range(reps):
            out = os.path.join(tmp, "data")
            st = {}
            t0 = time.perf_counter()
            create_dataset(os.path.join(wav, "track_*.wav"), out, stats=st)
            wall = time.perf_counter() - t0
            print(f"{thr} {r} {nfiles / (wall - st['setup_s']):7.1f} {wall:.3f} {st['setup_s']:.3f} {st['loader_thread_busy_s']:.3f} "
                  f"{st['load_stft_s']:.3f} {st['codec_copy_submit_s']:.3f} {st['ring_wait_s']:.3f} {st['drain_s']:.3f} {st['writer_busy_s']:.3f}",
                  flush=True)
            shutil.rmtree(out)
            os.sync()  # the next run starts without this one's dirty pages
            time.sleep(1.0)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
