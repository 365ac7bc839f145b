"""CLI with the reference's sub-commands and flags (/root/reference/music_gan/__main__.py:11-124)."""
import argparse


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser("MusicGAN")
    sub = parser.add_subparsers()
    sub.required = True
    sub.dest = "mode"

    p = sub.add_parser("create_dataset")
    p.add_argument("audio_path", type=str, help="can be /path/to/*.wav")
    p.add_argument("-o", "--output-dir", type=str, required=True,
                   help="The folder where the tensor files will be saved")

    p = sub.add_parser("train")
    p.add_argument("run", type=str, metavar="RUN_NAME")
    p.add_argument("-o", "--out-path", dest="out_path", type=str, required=True)
    p.add_argument("-i", "--input-dataset", dest="input_dataset", required=True, type=str)

    p = sub.add_parser("generate")
    p.add_argument("gen_dict_state", type=str)
    p.add_argument("rand_channels", type=int)
    p.add_argument("-n", "--nb-vec", type=int, default=10)
    p.add_argument("-m", "--nb-music", type=int, default=5)
    p.add_argument("-o", "--output-dir", type=str, required=True)

    p = sub.add_parser("view_audio")
    p.add_argument("--input-audio", type=str, required=True)
    p.add_argument("--image-idx", type=int, required=True)
    return parser


def main() -> None:
    args = build_parser().parse_args()
    if args.mode == "create_dataset":
        from .create_dataset import create_dataset
        create_dataset(args.audio_path, args.output_dir)
    elif args.mode == "train":
        from .train import train
        train(args.run, args.input_dataset, args.out_path)
    elif args.mode == "generate":
        from .generate import generate
        generate(args.output_dir, args.rand_channels, args.gen_dict_state, args.nb_vec, args.nb_music)
    elif args.mode == "view_audio":
        from .view_audio import view_audio
        view_audio(args.input_audio, args.image_idx)


if __name__ == '__main__':
    main()
