"""`python -m musicgan_amd <mode> ...`: the reference's four sub-commands with its positional names and flags
(/root/reference/music_gan/__main__.py:11-124), declared as data and dispatched lazily (importing `train` pulls in the GPU
library, `view_audio` pulls in matplotlib)."""
import argparse
import importlib

# mode -> (module, function, [(flags, kwargs)], lambda args: call arguments)
_MODES = {
    "create_dataset": ("create_dataset", "create_dataset", [
        (("audio_path",), dict(type=str, help="can be /path/to/*.wav")),
        (("-o", "--output-dir"), dict(type=str, required=True, help="The folder where the tensor files will be saved")),
    ], lambda a: (a.audio_path, a.output_dir)),
    "train": ("train", "train", [
        (("run",), dict(type=str, metavar="RUN_NAME")),
        (("-o", "--out-path"), dict(dest="out_path", type=str, required=True)),
        (("-i", "--input-dataset"), dict(dest="input_dataset", type=str, required=True)),
    ], lambda a: (a.run, a.input_dataset, a.out_path)),
    "generate": ("generate", "generate", [
        (("gen_dict_state",), dict(type=str)),
        (("rand_channels",), dict(type=int)),
        (("-n", "--nb-vec"), dict(type=int, default=10)),
        (("-m", "--nb-music"), dict(type=int, default=5)),
        (("-o", "--output-dir"), dict(type=str, required=True)),
    ], lambda a: (a.output_dir, a.rand_channels, a.gen_dict_state, a.nb_vec, a.nb_music)),
    "view_audio": ("view_audio", "view_audio", [
        (("--input-audio",), dict(type=str, required=True)),
        (("--image-idx",), dict(type=int, required=True)),
    ], lambda a: (a.input_audio, a.image_idx)),
}


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser("MusicGAN")
    modes = parser.add_subparsers(dest="mode")
    modes.required = True
    for mode, (_, _, arguments, _) in _MODES.items():
        sub = modes.add_parser(mode)
        for flags, kwargs in arguments:
            sub.add_argument(*flags, **kwargs)
    return parser


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    module, function, _, call_args = _MODES[args.mode]
    getattr(importlib.import_module(f".{module}", __package__), function)(*call_args(args))


if __name__ == "__main__":
    main()
