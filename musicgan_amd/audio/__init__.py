"""Public surface of the reference's `audio` package (/root/reference/music_gan/audio/__init__.py) plus the two
waveform-level helpers the drivers use (`stft_from_waveform`, `magn_phase_to_waveform`)."""
from . import constant as _constant
from . import functions as _functions
from .constant import N_FFT, N_VEC, SAMPLE_RATE, STFT_STRIDE
from .dataset import AudioDataset, PackedAudioDataset, PackedLoader, has_packed, write_packed
from .transforms import ChangeRange, ChannelMinMaxNorm

for _name in ("wav_to_stft", "stft_to_phase_magn", "magn_phase_to_wav", "bark_magn_scale", "stft_from_waveform",
              "magn_phase_to_waveform", "stft_to_stacked_phase_magn"):
    globals()[_name] = getattr(_functions, _name)
del _name

__all__ = ["wav_to_stft", "stft_to_phase_magn", "magn_phase_to_wav", "bark_magn_scale", "stft_from_waveform",
           "magn_phase_to_waveform", "stft_to_stacked_phase_magn", "AudioDataset", "PackedAudioDataset", "PackedLoader", "has_packed", "write_packed", "ChannelMinMaxNorm", "ChangeRange", *_constant.__all__]
