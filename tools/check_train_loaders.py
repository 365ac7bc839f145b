import sys, os, tempfile, time, shutil
sys.path.insert(0, '/root/repo')
import torch
from musicgan_amd.train import train
root = tempfile.mkdtemp(prefix="mg_tw_")
try:
    data = os.path.join(root, "data"); os.mkdir(data)
    g = torch.Generator().manual_seed(0)
    for i in range(24):
        torch.save((torch.rand(2, 512, 512, generator=g) * 2 - 1).double(), os.path.join(data, f"magn_phase_{i}.pt"))
    t0 = time.time()
    train("w", data, os.path.join(root, "out"), nb_epoch=50, batch_size=6, num_workers=2, max_iters=40, save_every=1000,
          use_packed_loader=False, fadein_lengths=[1, 30, 30, 30, 30, 30, 30, 30], train_lengths=[40, 60, 1000, 1000, 1000, 1000, 1000])
    print("reference loader path with 2 workers + graphs: ok, %.1f s" % (time.time() - t0))
    from musicgan_amd import audio
    audio.write_packed(data)
    t0 = time.time()
    train("p", data, os.path.join(root, "out2"), nb_epoch=50, batch_size=6, num_workers=0, max_iters=40, save_every=1000,
          fadein_lengths=[1, 30, 30, 30, 30, 30, 30, 30], train_lengths=[40, 60, 1000, 1000, 1000, 1000, 1000])
    print("packed loader + graphs: ok, %.1f s" % (time.time() - t0))
finally:
    shutil.rmtree(root, ignore_errors=True)
