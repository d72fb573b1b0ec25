"""
`python -m open_universe_amd.bin.enhance <input> <output> [--model ...] [enhance arguments]`

The reference's inference script (open_universe/bin/enhance.py:83-192) on the MI355X-native model: enhance one audio
file or every .wav/.mp3/.flac below a folder (folder structure retained), channels of a file = batch of the
`enhance` call, resample to the model rate and back, ONE torch.Generator seeded with --seed shared by all files in
processing order (so a run is reproducible file by file, exactly like the reference), model-specific arguments
generated from the signature of `model.enhance`.

Extension (not in the reference, whose loop is serial on one device): launched under `torch.distributed.run` with N
processes, the files are sharded over the ranks (one GPU each, LPT by duration so the ranks finish together); the
packed weights are loaded per rank.  Because the reference draws the noise of file k from the generator state
left by files 0..k-1, a sharded run cannot reproduce the serial noise; with `--per-file-seed` (forced when N > 1)
file k uses its own generator seeded with `seed + k` (k = index in the sorted file list), which makes the result of
a file independent of how the list was sharded.
"""
import argparse
import os
import sys
from pathlib import Path

import torch

from .. import inference_utils
from ..audio import AUDIO_SUFFIXES, can_decode, load, resample, save


def handle_help(argv):
    """bin/enhance.py:36-57: defer --help until the model's own arguments are known."""
    if "--model" not in argv:
        return False
    for flag in ("--help", "-h"):
        if flag in argv:
            argv.remove(flag)
            return True
    return False


def find_files(path):
    """bin/enhance.py:60-74 (sorted, so that the processing order -- and with it the noise -- is deterministic)."""
    path = Path(path)
    if not path.is_dir():
        return [path], path.parent, False
    files = sorted(p for p in path.rglob("*") if p.suffix in AUDIO_SUFFIXES)
    return files, path, True


def plan_files(files, world, rank):
    """(index, path) pairs this rank processes: all of them in order for one process, else an LPT shard by file size
    (a proxy for the duration) -- the index k is what --per-file-seed adds to the seed."""
    todo = list(enumerate(files))
    if world <= 1:
        return todo
    from ..distributed import shard_utterances

    sizes = [os.path.getsize(p) for p in files]
    return [todo[i] for i in sorted(shard_utterances(sizes, world)[rank])]


def build_parser():
    parser = argparse.ArgumentParser(description="Enhance a file or a directory of audio files")
    parser.add_argument("input", type=Path, help="Path to an audio file or a folder of audio files")
    parser.add_argument("output", type=Path,
                        help="Output path for the enhanced files. In the case of a folder, the orignal structure is retained.")
    parser.add_argument("--model", type=str, default="line-corporation/open-universe:plusplus",
                        help="A checkpoint file (config.yaml beside it) or a Huggingface model id repo[:revision]")
    parser.add_argument("--hf-token", type=str, help="Huggingface access token")
    parser.add_argument("--model-strict", action="store_true",
                        help="Use strict policy to load the model. Can help uncover problems.")
    parser.add_argument("--seed", type=int, default=1028282, help="Set a deterministic seed to get reproducible results")
    parser.add_argument("--device", type=str, default="cuda:0", help="The device to use, e.g. cuda:0. Default: cuda:0.")
    parser.add_argument("--per-file-seed", action="store_true",
                        help="Seed the generator of file k with seed + k instead of sharing one generator across files "
                             "(always on when the files are sharded over several processes)")
    return parser


def main(argv=None, model=None):
    """`model`: an already-loaded model (tests); otherwise --model is loaded like the reference does."""
    argv = list(sys.argv[1:] if argv is None else argv)
    parser = build_parser()
    requires_help = handle_help(argv)
    args, _ = parser.parse_known_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if model is None:
        device = args.device
        if world > 1:
            # one GPU per rank; ranks wrap around the visible devices when there are fewer GPUs than ranks
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', rank)) % max(1, torch.cuda.device_count())}"
        if not device.startswith("cuda"):
            raise ValueError("Device name should be 'cuda:X' where X is an integer (this build has no CPU path). "
                             f"Provided {device}")
        model = inference_utils.load_model(args.model, device=device, strict=args.model_strict, hf_token=args.hf_token)
    device = str(getattr(model, "device", args.device))

    inference_utils.add_enhance_arguments(model, parser)
    if requires_help:
        argv.append("--help")
    args = parser.parse_args(argv)
    enhance_kwargs = {}
    for group in parser._action_groups:
        if group.title == "enhance":
            enhance_kwargs = {a.dest: getattr(args, a.dest, None) for a in group._group_actions}

    files, rel_path, dir_proc = find_files(args.input)
    per_file_seed = args.per_file_seed or world > 1
    rng = torch.Generator(device=device)
    rng.manual_seed(args.seed)

    undecodable = [str(p) for p in files if not can_decode(p)]
    if undecodable:
        # fail before the first file is enhanced, not in the middle of a directory (torchaudio is what the reference
        # decodes .mp3 / .flac with, bin/enhance.py:183)
        raise RuntimeError(f"{len(undecodable)} input file(s) need torchaudio to be decoded (only .wav is read natively): "
                           + ", ".join(undecodable[:5]))
    todo = plan_files(files, world, rank)

    done = []
    for k, path in todo:
        if dir_proc:
            output_path = args.output / path.relative_to(rel_path)
            output_path.parent.mkdir(exist_ok=True, parents=True)
        elif args.output.is_dir():
            output_path = args.output / path.name
        else:
            output_path = args.output
        audio, fs = load(path)
        audio = audio.to(device)
        if per_file_seed:
            rng.manual_seed(args.seed + k)
        with torch.no_grad():
            audio = resample(audio, fs, model.fs)
            enh = model.enhance(audio, **dict(enhance_kwargs, rng=rng))
            enh = resample(enh, model.fs, fs)
        save(output_path, enh.cpu(), fs)
        done.append(output_path)
    return done


if __name__ == "__main__":
    main()
