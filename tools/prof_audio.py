"""The audio path a few times for rocprofv3 --kernel-trace --stats: STFT of one 10-minute file, the codec, the inverse codec.
    rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/prof_audio.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from musicgan_amd import audio, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
wav = torch.rand(44100 * 600, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) - 0.5
for _ in range(iters):
    c = ops.stft_1024(wav)
    mp = audio.stft_to_stacked_phase_magn(c)[:8].contiguous()
    audio.functions.magn_phase_to_waveform(mp)
torch.cuda.synchronize()
