"""STFT throughput as a function of time under load: batches of 50 launches after 2 s of idle (the clock ramp behind bench.py's 250-launch warm-up of the STFT record).  python tools/bench_stft_ramp.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgan_amd import ops
dev = torch.device("cuda", 0)
L = 44100 * 600
wav = torch.rand(L, device=dev) - 0.5
T = 1 + L // 256
ops.stft_1024(wav); torch.cuda.synchronize()
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 2.0)  # idle, as after host-side work
res = []
for b in range(int(sys.argv[2]) if len(sys.argv) > 2 else 14):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.stft_1024(wav)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 50
    res.append(f"{5120*T/(ms*1e-3)/8e12:.3f}")
print("frac per batch of 50 launches after 2 s idle:", " ".join(res))
