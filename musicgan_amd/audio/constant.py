# /root/reference/music_gan/audio/constant.py:1-4
N_FFT = 1024
N_VEC = 512
STFT_STRIDE = 256
SAMPLE_RATE = 44100
