"""musicgan_amd -- the Ipsedo/MusicGAN training hot path on AMD MI355X (gfx950): ProGAN WGAN-GP generator/discriminator step
and the STFT / magnitude-phase codec, hand-written HIP behind a C ABI (include/musicgan_hip.h), with the reference's Python
surface (`networks`, `audio`, `train`, `generate`, `create_dataset`, `view_audio`) on top."""


def __getattr__(name):  # lazy: importing the package must not drag in the data/IO stack
    if name in ("train", "generate", "create_dataset", "view_audio"):
        import importlib
        fn = getattr(importlib.import_module(f".{name}", __name__), name)
        globals()[name] = fn  # the import system bound the sub-module under this name; re-bind the function like the reference
        return fn
    if name in ("networks", "audio", "ops", "optim", "dist", "utils", "train_step"):
        import importlib
        return importlib.import_module(f".{name}", __name__)
    raise AttributeError(name)
