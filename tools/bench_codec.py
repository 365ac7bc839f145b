"""Time the forward codec (stft_to_phase_magn) on one 10-minute track's STFT: `python tools/bench_codec.py [iters]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from musicgan_amd import audio, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
wav = torch.rand(44100 * 600, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) - 0.5
c = ops.stft_1024(wav)
bark = audio.functions._bark_vector(512, dev)
for stacked in (False, True):
    for _ in range(20):
        ops.codec_fwd(c, bark, 512, stacked=stacked)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops.codec_fwd(c, bark, 512, stacked=stacked)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bins = 512 * c.shape[1]
    print(f"codec_fwd stacked={stacked}: {ms:.4f} ms per 10-minute track; {16 * bins / ms / 1e9:.3f} TB/s algorithmic "
          f"(16 B/bin) = {16 * bins / ms / 1e9 / 8.0:.3f} of 8 TB/s")
for _ in range(20):
    ops.stft_1024(wav)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    cc = ops.stft_1024(wav)
    ops.codec_fwd(cc, bark, 512, stacked=True)
e1.record()
torch.cuda.synchronize()
print(f"stft + codec: {e0.elapsed_time(e1) / iters:.4f} ms per file")
