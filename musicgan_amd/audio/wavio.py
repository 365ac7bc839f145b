"""Minimal audio file IO (the reference uses torchaudio.load/save, functions.py:43,139; torchaudio is not a dependency here).
load() mirrors torchaudio.load(normalize=True): float32 tensor (channels, samples) in [-1, 1] and the sample rate.
Containers: RIFF WAV (scipy), AIFF / AIFF-C and Sun AU with linear PCM (Python's standard library) -- the uncompressed formats
torchaudio's backends read without a codec; compressed formats (flac, mp3, ogg) need decoders this image does not have and raise."""
from __future__ import annotations

import os

import numpy as np
import torch
from scipy.io import wavfile


def _linear_pcm(raw: bytes, width: int, channels: int, what: str) -> np.ndarray:
    """Big-endian signed linear PCM (AIFF, AU) -> (frames, channels) int16 / int32 with the value torchaudio normalises:
    8-bit v -> v * 2^8 (v / 2^7 == that / 2^15), 24-bit v -> v * 2^8 (v / 2^23 == that / 2^31)."""
    if width == 1:
        x = np.frombuffer(raw, dtype=np.int8).astype(np.int16) * 256
    elif width == 2:
        x = np.frombuffer(raw, dtype=">i2").astype(np.int16)
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        x = ((b[:, 0] << 24) | (b[:, 1] << 16) | (b[:, 2] << 8)).astype(np.int32)  # sign bit lands in bit 31
    elif width == 4:
        x = np.frombuffer(raw, dtype=">i4").astype(np.int32)
    else:
        raise ValueError(f"{what}: {8 * width}-bit samples are not supported")
    return np.ascontiguousarray(x.reshape(-1, channels))


def _read_frames(path: str, mmap: bool = False):
    """(frames, channels) array as stored (int16 / int32 / uint8 / float32 / float64) and the sample rate."""
    ext = os.path.splitext(path)[1].lower()
    if ext in (".aif", ".aiff", ".aifc"):
        import aifc
        with aifc.open(path, "rb") as f:
            if f.getcomptype() not in (b"NONE", b"sowt"):
                raise ValueError(f"{path}: compressed AIFF-C ({f.getcomptype().decode()}) is not supported")
            raw, width, ch, sr = f.readframes(f.getnframes()), f.getsampwidth(), f.getnchannels(), f.getframerate()
            if f.getcomptype() == b"sowt":  # little-endian 16-bit variant
                return np.ascontiguousarray(np.frombuffer(raw, dtype="<i2").astype(np.int16).reshape(-1, ch)), int(sr)
        return _linear_pcm(raw, width, ch, path), int(sr)
    if ext in (".au", ".snd"):
        import sunau
        with sunau.open(path, "rb") as f:
            if f.getcomptype() != "NONE":
                raise ValueError(f"{path}: {f.getcompname()} AU files are not supported (linear PCM only)")
            raw, width, ch, sr = f.readframes(f.getnframes()), f.getsampwidth(), f.getnchannels(), f.getframerate()
        return _linear_pcm(raw, width, ch, path), int(sr)
    if ext not in (".wav", ".wave", ""):
        raise ValueError(f"{path}: only WAV, AIFF and AU (linear PCM) files can be read here; torchaudio's codec-backed formats "
                         f"(flac, mp3, ogg, ...) need a decoder this build does not ship")
    try:
        sr, data = wavfile.read(path, mmap=mmap)
    except ValueError:  # formats scipy cannot map (e.g. 24-bit)
        sr, data = wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    return data, int(sr)


def load(path: str):
    data, sr = _read_frames(path)
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
    data, sr = _read_frames(path, mmap=mmap)
    if data.dtype not in (np.int16, np.int32, np.uint8, np.float32):
        data = np.asarray(data, dtype=np.float32)  # (64-bit float files)
    return data, int(sr)


def save(path: str, wav: torch.Tensor, sample_rate: int) -> None:
    """(channels, samples) float tensor -> 32-bit float WAV (what torchaudio.save writes for float32 input)."""
    x = wav.detach().to("cpu", torch.float32).numpy()
    wavfile.write(path, int(sample_rate), np.ascontiguousarray(x.T))
