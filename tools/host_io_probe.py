"""The host-side bound of create_dataset on this box (bench.host_io_probe): pinned D2H rate and the writer path's rate, as JSON.
python tools/host_io_probe.py [threads] > gpurun_out/host_io_probe.json   (copy into profiles/ to keep it)"""
import json, os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from musicgan_amd.create_dataset import host_cpus
thr = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, min(16, host_cpus()))
tmp = tempfile.mkdtemp(prefix="mg_probe_")
try:
    print(json.dumps(bench.host_io_probe(torch.device("cuda", 0), tmp, thr), indent=1))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
