"""Container-only helper: import the reference's sub-packages from /root/reference for golden-vector generation.

The reference never ships; nothing under tests/, bench.py or the product imports this at run time on the GPU box.
`music_gan/__init__.py` pulls in mlflow/torchvision/torchaudio (absent here), so the torch-only
`music_gan.networks` sub-package is loaded under an empty synthetic parent package (SURVEY 8(c) recipe).
For `music_gan.audio` the absent, unpinned `torchaudio` is replaced by a stand-in that restates the two
functional wrappers the reference calls on `torch.stft` / `torch.istft`; that boundary is
therefore pinned on torch.stft, not on torchaudio itself (recorded in DESIGN.md).
"""
import importlib.util
import os
import sys
import types

REF_ROOT = "/root/reference/music_gan"


def _synthetic_parent():
    sys.dont_write_bytecode = True
    if "music_gan" not in sys.modules or not hasattr(sys.modules["music_gan"], "__graft_synthetic__"):
        pkg = types.ModuleType("music_gan")
        pkg.__path__ = [REF_ROOT]
        pkg.__graft_synthetic__ = True
        sys.modules["music_gan"] = pkg


def _load(sub):
    _synthetic_parent()
    name = f"music_gan.{sub}"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(REF_ROOT, sub, "__init__.py"),
        submodule_search_locations=[os.path.join(REF_ROOT, sub)])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_networks():
    return _load("networks")


_WAV_STORE = {}


def _install_torchaudio_standin():
    import torch

    if "torchaudio" in sys.modules:
        return
    ta = types.ModuleType("torchaudio")
    taf = types.ModuleType("torchaudio.functional")

    def spectrogram(waveform, pad, window, n_fft, hop_length, win_length, power, normalized,
                    center=True, pad_mode="reflect", onesided=True, return_complex=None):
        if pad > 0:
            waveform = torch.nn.functional.pad(waveform, (pad, pad), "constant")
        shape = waveform.size()
        waveform = waveform.reshape(-1, shape[-1])
        spec = torch.stft(waveform, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window,
                          center=center, pad_mode=pad_mode, normalized=False, onesided=onesided,
                          return_complex=True)
        spec = spec.reshape(shape[:-1] + spec.shape[-2:])
        if normalized:
            spec = spec / window.pow(2.0).sum().sqrt()
        if power is not None:
            return spec.abs() if power == 1.0 else spec.abs().pow(power)
        return spec

    def inverse_spectrogram(spectrogram, length, pad, window, n_fft, hop_length, win_length, normalized,
                            center=True, pad_mode="reflect", onesided=True):
        if normalized:
            spectrogram = spectrogram * window.pow(2.0).sum().sqrt()
        shape = spectrogram.size()
        spectrogram = spectrogram.reshape(-1, shape[-2], shape[-1])
        wav = torch.istft(spectrogram, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window,
                          center=center, normalized=False, onesided=onesided,
                          length=length + 2 * pad if length is not None else None, return_complex=False)
        if length is not None and pad > 0:
            wav = wav[:, pad:-pad]
        return wav.reshape(shape[:-2] + wav.shape[-1:])

    def load(path):
        return _WAV_STORE[path]

    def save(path, tensor, sr):
        _WAV_STORE[path] = (tensor.clone(), sr)

    taf.spectrogram = spectrogram
    taf.inverse_spectrogram = inverse_spectrogram
    ta.functional = taf
    ta.load = load
    ta.save = save
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.functional"] = taf


def load_audio():
    _install_torchaudio_standin()
    return _load("audio")


def wav_store():
    return _WAV_STORE


def _install_torchvision_standin():
    """`music_gan/utils.py:7` needs torchvision.transforms.{Compose, Resize} (absent, unpinned).  Compose is restated
    verbatim (call the transforms in order); Resize(int) on a square (N,C,H,W) tensor is restated on aten's bilinear
    interpolation with anti-aliasing, torchvision's tensor default -- so anything that goes through Resize is pinned on torch,
    not on torchvision (recorded in DESIGN.md).  The schedule arithmetic of `Grower` (grow / alpha) never touches it."""
    import torch

    if "torchvision" in sys.modules:
        return
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size):
            self.size = size

        def __call__(self, x):
            if tuple(x.shape[-2:]) == (self.size, self.size):
                return x
            return torch.nn.functional.interpolate(x, size=(self.size, self.size), mode="bilinear", antialias=True,
                                                   align_corners=False)

    tvt.Compose, tvt.Resize = Compose, Resize
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt


def load_utils():
    """The reference's `utils.py` (Grower, Saver) as module `music_gan.utils`."""
    _install_torchaudio_standin()
    _install_torchvision_standin()
    import matplotlib
    matplotlib.use("Agg")
    _load("networks")
    _load("audio")
    name = "music_gan.utils"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, "utils.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m
