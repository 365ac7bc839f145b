from .functions import (
    wav_to_stft,
    bark_magn_scale,
    stft_to_phase_magn,
    magn_phase_to_wav,
    stft_from_waveform,
    magn_phase_to_waveform,
)
from .dataset import AudioDataset
from .transforms import ChannelMinMaxNorm, ChangeRange
from .constant import *
