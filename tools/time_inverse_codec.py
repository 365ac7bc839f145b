import sys, os, torch
sys.path.insert(0, '/root/repo')
from musicgan_amd import audio
dev = torch.device("cuda", 0)
for n in (1, 10, 40):
    mp = (torch.rand(n, 2, 512, 512, device=dev) * 2 - 1)
    for _ in range(2): w = audio.magn_phase_to_waveform(mp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): w = audio.magn_phase_to_waveform(mp)
    e1.record(); e1.synchronize()
    print(n, "items ->", w.numel(), "samples:", e0.elapsed_time(e1) / 5, "ms", flush=True)
