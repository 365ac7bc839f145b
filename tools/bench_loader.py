"""Loader throughput: samples/s delivered to the GPU by (a) the reference's path -- AudioDataset + DataLoader
workers, one th.load of a 4 MiB float64 .pt per sample -- and (b) the float32 memory-mapped side-car + PackedLoader, against the
images/s one MI355X consumes.   python tools/bench_loader.py [samples] [batch] [workers]"""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

from musicgan_amd import audio  # noqa: E402
from musicgan_amd.train import ShardedShuffle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
root = tempfile.mkdtemp(prefix="mg_loader_")
try:
    g = torch.Generator().manual_seed(0)
    for i in range(n):
        torch.save((torch.rand(2, 512, 512, generator=g) * 2 - 1).double(), os.path.join(root, f"magn_phase_{i}.pt"))
    t0 = time.perf_counter()
    audio.write_packed(root)
    t_pack = time.perf_counter() - t0
    out = {"samples": n, "batch": batch, "pack_seconds": t_pack}

    def run(loader, epochs):
        cnt = 0
        t0 = time.perf_counter()
        for e in range(epochs):
            if hasattr(loader, "sampler") and hasattr(loader.sampler, "set_epoch"):
                loader.sampler.set_epoch(e)
            for x in loader:
                x = x.to(dev, non_blocking=True)
                cnt += x.shape[0]
        torch.cuda.synchronize()
        return cnt / (time.perf_counter() - t0)

    ref = DataLoader(audio.AudioDataset(root), batch_size=batch, sampler=ShardedShuffle(n, 0), num_workers=workers,
                     drop_last=True, pin_memory=True, persistent_workers=workers > 0)
    run(ref, 1)
    out["reference_loader_samples_per_s"] = run(ref, 3)
    out["reference_loader"] = f"AudioDataset + DataLoader(num_workers={workers}, pin_memory): th.load of float64 .pt per sample"
    ds = audio.PackedAudioDataset(root)
    pl = audio.PackedLoader(ds, batch, ShardedShuffle(n, 0), dev)
    run(pl, 1)
    out["packed_loader_samples_per_s"] = run(pl, 6)
    out["packed_loader"] = "float32 memmap side-car, background gather into pinned buffers + async upload"
    print(json.dumps(out))
finally:
    shutil.rmtree(root, ignore_errors=True)
