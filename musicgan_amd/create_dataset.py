"""`create_dataset(audio_path_glob, dataset_output_dir)` (/root/reference/music_gan/create_dataset.py:13-64): wav files ->
STFT -> (magnitude, phase-delta) images -> `magn_phase_{idx}.pt` (float64, shape (2,512,512)), STFT and codec on the GPU.
Under torchrun the files are dealt round-robin to the ranks (independent units, no collective); the global sample numbering
stays the reference's (files in glob order) because each file's sample count follows from its length alone."""
import glob
import os
from os import mkdir
from os.path import exists, isdir, join

import torch as th

from . import audio
from .audio import wavio


def _nb_samples(nb_frames_wav: int, nb_vec: int) -> int:
    t = 1 + nb_frames_wav // audio.STFT_STRIDE
    return 0 if t < nb_vec else (t - 1) // nb_vec


def create_dataset(audio_path: str, dataset_output_dir: str, *, packed: bool = True) -> None:
    """`packed` (extension, single-process runs): also write the float32 memory-mapped side-car the fast loader reads
    (audio/dataset.py); the reference-format `magn_phase_{idx}.pt` files are written either way."""
    w_p = glob.glob(audio_path)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if not exists(dataset_output_dir):
        os.makedirs(dataset_output_dir, exist_ok=True)
    elif not isdir(dataset_output_dir):
        raise NotADirectoryError(f"\"{dataset_output_dir}\" is not a directory")
    nb_vec = audio.N_VEC
    if world > 1:
        th.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        from scipy.io import wavfile
        counts = []
        for p in w_p:
            _, data = wavfile.read(p, mmap=True)
            counts.append(_nb_samples(data.shape[0], nb_vec))
    idx = 0
    for f_i, wav_p in enumerate(w_p):
        if world > 1 and f_i % world != rank:
            idx += counts[f_i]
            continue
        complex_values = audio.wav_to_stft(wav_p, nperseg=audio.N_FFT, stride=audio.STFT_STRIDE)
        if complex_values.size()[1] < nb_vec:
            continue
        magn, phase = audio.stft_to_phase_magn(complex_values, nb_vec=nb_vec)
        both = th.stack([magn, phase], dim=1).to(th.float64).cpu()  # (S, 2, 512, nb_vec)
        for s_idx in range(both.size()[0]):
            th.save(both[s_idx].clone(), join(dataset_output_dir, f"magn_phase_{idx}.pt"))
            idx += 1
    if packed and world == 1:
        audio.write_packed(dataset_output_dir)
