"""`view_audio(audio_path, image_idx)` (/root/reference/music_gan/view_audio.py:6-26): matplotlib preview of one sample."""
from . import audio


def view_audio(audio_path: str, image_idx: int) -> None:
    import matplotlib.pyplot as plt
    magn, phase = audio.stft_to_phase_magn(audio.wav_to_stft(audio_path))
    fig, (a0, a1) = plt.subplots(1, 2)
    a0.matshow(magn[image_idx].cpu().numpy(), cmap="plasma")
    a0.set_title("magnitude")
    a1.matshow(phase[image_idx].cpu().numpy(), cmap="plasma")
    a1.set_title("phase delta")
    plt.show()
