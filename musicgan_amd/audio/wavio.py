"""Minimal WAV file IO (the reference uses torchaudio.load/save, functions.py:43,139; torchaudio is not a dependency here).
load() mirrors torchaudio.load(normalize=True): float32 tensor (channels, samples) in [-1, 1] and the sample rate."""
from __future__ import annotations

import numpy as np
import torch
from scipy.io import wavfile


def load(path: str):
    sr, data = wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.T)), int(sr)


def load_pcm(path: str, mmap: bool = True):
    """The file's frames as stored: array (frames, channels) of int16 / int32 / uint8 / float32 (a read-only memory map where the
    format allows) and the sample rate -- what `load` normalises and transposes; the device path does both inside the STFT kernel
    (ops.stft_1024_pcm), so a file's bytes travel to the GPU as they are (int16: half of float32's)."""
    try:
        sr, data = wavfile.read(path, mmap=mmap)
    except ValueError:  # formats scipy cannot map (e.g. 24-bit)
        sr, data = wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype not in (np.int16, np.int32, np.uint8, np.float32):
        data = np.asarray(data, dtype=np.float32)  # (64-bit float files)
    return data, int(sr)


def save(path: str, wav: torch.Tensor, sample_rate: int) -> None:
    """(channels, samples) float tensor -> 32-bit float WAV (what torchaudio.save writes for float32 input)."""
    x = wav.detach().to("cpu", torch.float32).numpy()
    wavfile.write(path, int(sample_rate), np.ascontiguousarray(x.T))
