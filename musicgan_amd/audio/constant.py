"""Signal constants of the data format (/root/reference/music_gan/audio/constant.py:1-4): every stored sample is a
(2, N_FFT // 2, N_VEC) magnitude / phase image cut from an STFT with a window of N_FFT samples hopping by STFT_STRIDE."""
SAMPLE_RATE = 44_100                 # Hz; wav_to_stft refuses anything else
N_FFT = 1 << 10                      # window length = FFT size; 512 frequency rows are kept (Nyquist dropped)
STFT_STRIDE = N_FFT >> 2             # hop: 75 % overlap of the periodic Hann window
N_VEC = N_FFT >> 1                   # frames per stored sample: square 512 x 512 images

__all__ = ["SAMPLE_RATE", "N_FFT", "STFT_STRIDE", "N_VEC"]
