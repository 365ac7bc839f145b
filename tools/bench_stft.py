"""STFT / codec throughput (BASELINE.json configs[4] shape: 10 min of 44.1 kHz audio per file, n_fft 1024, hop 256).
Prints frames/s and the fraction of the HBM roofline at 5 120 algorithmic bytes per frame."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import audio, ops

dev = torch.device("cuda", 0)
L = 44100 * 600
g = torch.Generator(device=dev).manual_seed(7)
wav = torch.rand(L, device=dev, generator=g) - 0.5
T = 1 + L // 256

def timeit(fn, iters):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters

ms_stft = timeit(lambda: ops.stft_1024(wav), 20)
c = ops.stft_1024(wav)
ms_codec = timeit(lambda: audio.stft_to_phase_magn(c), 3)
fps = T / (ms_stft * 1e-3)
out = {"frames": T, "stft_ms": ms_stft, "stft_frames_per_s": fps, "stft_alg_GBps": 5120 * fps / 1e9,
       "stft_frac_of_8TBps": 5120 * fps / 8e12, "codec_ms": ms_codec,
       "stft_plus_codec_samples_per_s": (T - 1) // 512 / ((ms_stft + ms_codec) * 1e-3)}
# CPU baseline: torch.stft on the host cores (the reference's own path goes through torchaudio -> torch.stft)
import bench
torch.set_num_threads(bench.host_cpu_share())
w = wav.cpu()
win = torch.hann_window(1024)
t0 = time.perf_counter()
torch.stft(w, 1024, 256, 1024, win, center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
out["cpu_torch_stft_frames_per_s"] = T / (time.perf_counter() - t0)
out["cpu_threads"] = torch.get_num_threads()
print(json.dumps(out))
