"""Sampling driver with the reference's entry point `generate(output_dir, rand_channels, gen_dict_state, nb_vec, nb_music)`
(/root/reference/music_gan/generate.py:12-65): a fully grown generator turns wide latents into (2, 512, 512 * nb_vec)
magnitude / phase images, which the inverse codec + inverse STFT turn into `sound_{i}.wav` -- all of it on the GPU."""
import os

import torch

from . import audio
from .networks import Generator

_LATENT_H, _LATENT_W = 2, 2   # latent tile that grows to one 512 x 512 image; nb_vec tiles side by side along time
_FINAL_LEVEL = 7


def _load_generator(rand_channels: int, checkpoint: str, device: torch.device) -> Generator:
    net = Generator(rand_channels, end_layer=_FINAL_LEVEL)
    net.load_state_dict(torch.load(checkpoint, map_location="cpu"))
    return net.to(device).eval()


def generate(output_dir: str, rand_channels: int, gen_dict_state: str, nb_vec: int, nb_music: int) -> None:
    if os.path.exists(output_dir) and not os.path.isdir(output_dir):
        raise NotADirectoryError(f"\"{output_dir}\" is not a directory")
    os.makedirs(output_dir, exist_ok=True)

    device = torch.device("cuda", torch.cuda.current_device())
    print("Load model...")
    gen = _load_generator(rand_channels, gen_dict_state, device)

    print("Pass rand data to generator...")
    latents = torch.randn(nb_music, rand_channels, _LATENT_H, _LATENT_W * nb_vec, device=device)
    print("Saving sound...")
    with torch.no_grad():
        # one item at a time: the level-7 activations of a 512 x (512 * nb_vec) image are ~1 GB each
        for idx, z in enumerate(latents.split(1, dim=0)):
            image = gen(z.contiguous(), 1.0)
            audio.magn_phase_to_wav(image, os.path.join(output_dir, f"sound_{idx}.wav"), audio.SAMPLE_RATE)
