"""`generate(output_dir, rand_channels, gen_dict_state, nb_vec, nb_music)` (/root/reference/music_gan/generate.py:12-65):
load a level-7 generator checkpoint, run it on a wide latent and write wav files -- on the GPU end to end."""
from os import mkdir
from os.path import exists, isdir, join

import torch as th

from . import audio
from .networks import Generator


def generate(output_dir: str, rand_channels: int, gen_dict_state: str, nb_vec: int, nb_music: int) -> None:
    if not exists(output_dir):
        mkdir(output_dir)
    elif not isdir(output_dir):
        raise NotADirectoryError(f"\"{output_dir}\" is not a directory")
    print("Load model...")
    device = th.device("cuda", th.cuda.current_device())
    gen = Generator(rand_channels, end_layer=7)
    gen.load_state_dict(th.load(gen_dict_state, map_location="cpu"))
    gen.to(device).eval()
    height, width = 2, 2
    with th.no_grad():
        print("Pass rand data to generator...")
        z = th.randn(nb_music, rand_channels, height, width * nb_vec, device=device)
        print("Saving sound...")
        for i in range(nb_music):  # one item at a time: the level-7 activations of a 512 x 5120 image are ~1 GB each
            gen_sound = gen(z[i:i + 1].contiguous(), 1.0)
            audio.magn_phase_to_wav(gen_sound.detach(), join(output_dir, f"sound_{i}.wav"), audio.SAMPLE_RATE)
