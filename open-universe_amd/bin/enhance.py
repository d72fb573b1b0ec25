"""
`python -m open_universe_amd.bin.enhance <input> <output> [--model ...] [enhance arguments]`

The reference's inference script (open_universe/bin/enhance.py:83-192) on the MI355X-native model: enhance one audio
file or every .wav/.mp3/.flac below a folder (folder structure retained), channels of a file = batch of the
`enhance` call, resample to the model rate and back, ONE torch.Generator seeded with --seed shared by all files in
processing order (so a run is reproducible file by file, exactly like the reference), model-specific arguments
generated from the signature of `model.enhance`.

Extensions (not in the reference, whose loop is serial on one device, one file per `enhance` call):
* launched under `torch.distributed.run` with N processes, the files are sharded over the ranks (one GPU each, LPT by
  file size so the ranks finish together); rank 0 reads the checkpoint and the packed weights reach the other ranks
  with one broadcast (`inference_utils.load_model_sharded`).  Because the reference draws the noise of file k from the
  generator state left by files 0..k-1, a sharded run cannot reproduce the serial noise; with `--per-file-seed` (forced
  when N > 1) file k uses its own generator seeded with `seed + k` (k = index in the sorted file list), which makes the
  result of a file independent of how the list was sharded;
* `--batch-size K`: up to K files of the same sample rate -- of ANY lengths -- share
  one `enhance` call (`Universe.enhance_many`; the channels of a file stay rows of the batch).  Every row keeps the
  geometry of the call on that file alone (own padding, normalisation, conv zero padding, GRU length: exact batching,
  ou_enhance_var), and the noise is drawn file by file in processing order with the shapes of the serial loop, so the
  shared generator advances exactly as in the reference and every file gets the noise -- and, to fp32 round-off, the
  result -- it would get alone.  Files are read `--batch-window` ahead (default 4 K) and grouped by length inside the window
  (a call costs what its longest row costs); every file's generator starts from the state the shared generator has in front
  of that file in processing order, so the grouping does not change anybody's noise.  `--pad-batch` selects the reference's own batch semantics instead: right-zero-padded to
  the longest like `max_collator` (datasets/datamodule.py:24-42; no mask: the padding takes part in the normalisation,
  as in a reference batch).
"""
import argparse
import os
import sys
from pathlib import Path

import torch

from .. import inference_utils
from ..audio import AUDIO_SUFFIXES, can_decode, channels, load, resample, save


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
    parser.add_argument("--batch-size", type=int, default=1,
                        help="Enhance up to this many consecutive files of equal sample rate (any lengths) in one call; "
                             "every file gets the result it would get alone")
    parser.add_argument("--batch-window", type=int, default=0,
                        help="With --batch-size: read this many files ahead and group them by length (longest first) -- a call "
                             "costs what its longest file costs.  Every file still gets the noise of the file-by-file loop "
                             "(its generator state is taken in processing order).  Default: 4 x batch-size; 1: files as they come")
    parser.add_argument("--in-flight", type=int, default=1,
                        help="Keep this many enhance calls in flight side by side on the device (one stream and workspace "
                             "each, 1..8): same result as the file-by-file loop, bit for bit, at a multiple of its "
                             "throughput (--batch-size reaches a higher rate, equal to fp32 round-off)")
    parser.add_argument("--pad-batch", action="store_true",
                        help="With --batch-size: files of different lengths are zero-padded to the longest WITHOUT a mask "
                             "(the reference's batch semantics: the padding changes every result)")
    return parser


def group_files(todo, infos, batch_size, pad_batch=False):
    """Consecutive runs of `todo` (in processing order) that may share one enhance call: same sample rate (any number of
    samples: rows keep their own geometry, or are zero-padded with pad_batch), at most batch_size files.
    infos[k] = (fs, n_samples)."""
    groups, cur = [], []
    for item in todo:
        k = item[0]
        if cur:
            k0 = cur[0][0]
            same = infos[k][0] == infos[k0][0]
            if not same or len(cur) >= batch_size:
                groups.append(cur)
                cur = []
        cur.append(item)
    if cur:
        groups.append(cur)
    return groups


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
        if world > 1:
            from .. import distributed

            distributed.init()  # nccl (= RCCL) with one GPU per rank, gloo when ranks share a device
            model = inference_utils.load_model_sharded(args.model, device=device, strict=args.model_strict,
                                                       hf_token=args.hf_token)
        else:
            model = inference_utils.load_model(args.model, device=device, strict=args.model_strict,
                                               hf_token=args.hf_token)
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
        raise RuntimeError(f"{len(undecodable)} input file(s) need torchaudio to be decoded (.wav and .flac are read natively): "
                           + ", ".join(undecodable[:5]))
    todo = plan_files(files, world, rank)

    def out_path(path):
        if dir_proc:
            output_path = args.output / path.relative_to(rel_path)
            output_path.parent.mkdir(exist_ok=True, parents=True)
            return output_path
        if args.output.is_dir():
            return args.output / path.name
        return args.output

    done = []
    if args.batch_size <= 1 and args.in_flight > 1 and enhance_kwargs.get("target") is None \
            and getattr(model, "fork", None) is not None:
        # The serial loop below with K calls in flight: file k + K is read, resampled and enqueued while files k .. k + K - 1
        # are still on the device; a file is written when its lane comes round again.  The noise is drawn at enqueue time
        # in processing order, so the shared generator advances exactly as in the serial loop.
        from ..lanes import LanePool

        def finish(item):
            lane, output_path, enh, fs = item
            pool.wait(lane)
            save(output_path, enh.cpu(), fs)
            done.append(output_path)

        # one call per file, batch size = its channel count: files that differ there put calls of different sizes in flight,
        # and the lanes then have to agree on how their GRU clusters share the device (headers only are read here)
        chans = {channels(path) for _, path in todo}
        mb = max(chans, default=1)
        with LanePool(model, min(int(args.in_flight), LanePool.MAX_LANES),
                      max_batch=mb if (len(chans) > 1 or mb > 1) else 0) as pool:
            pending = []
            for k, path in todo:
                output_path = out_path(path)
                audio, fs = load(path)
                audio = audio.to(device)
                if per_file_seed:
                    # a generator of its own per file: the draws of a call in flight must not see the next file's re-seed
                    file_rng = torch.Generator(device=device)
                    file_rng.manual_seed(args.seed + k)
                else:
                    file_rng = rng
                if len(pending) >= pool.lanes:
                    finish(pending.pop(0))

                def work(m, audio=audio, fs=fs, file_rng=file_rng):
                    with torch.no_grad():
                        x = resample(audio, fs, m.fs)
                        return resample(m.enhance(x, **dict(enhance_kwargs, rng=file_rng)), m.fs, fs)

                lane, enh = pool.submit(work, audio)
                pending.append((lane, output_path, enh, fs))
            for item in pending:
                finish(item)
        return done
    if args.batch_size <= 1:
        for k, path in todo:
            output_path = out_path(path)
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

    # --batch-size: files are read in processing order into a window (one sample rate, --batch-window files), sorted by length
    # inside it and enhanced batch_size at a time
    if any(enhance_kwargs.get(key) is not None for key in ("ensemble", "target")):
        raise ValueError("--batch-size cannot be combined with --ensemble (one call per file needed)")
    kw = {key: v for key, v in enhance_kwargs.items() if key not in ("rng", "ensemble", "ensemble_stat", "target",
                                                                    "fake_score_snr")}

    window_size = max(1, args.batch_window if args.batch_window > 0 else 4 * args.batch_size)
    # Sorting a window by length needs a generator per file that starts where the serial loop's shared generator would stand
    # in front of that file: with per-file seeds by construction, with the shared generator by taking its state file by file
    # in processing order and advancing it by the file's draws (Universe.advance_generator_like_enhance).
    can_sort = per_file_seed or hasattr(model, "advance_generator_like_enhance")
    if not can_sort:
        window_size = min(window_size, args.batch_size)

    def flush(window):
        """window: list of (k, path, audio, fs) with one fs, in processing order"""
        if not window:
            return
        fs = window[0][3]
        with torch.no_grad():
            items = []
            for k, path, a, _ in window:
                sig = resample(a.to(device), fs, model.fs)
                if per_file_seed:
                    g = torch.Generator(device=device)
                    g.manual_seed(args.seed + k)
                elif can_sort:
                    g = torch.Generator(device=device)
                    g.set_state(rng.get_state())
                    model.advance_generator_like_enhance(rng, sig.shape[0] if sig.ndim == 2 else 1, sig.shape[-1],
                                                         n_steps=kw.get("n_steps"), warm_start=kw.get("warm_start"),
                                                         use_aux_signal=bool(kw.get("use_aux_signal")))
                else:
                    g = None  # (a model without the hook: consecutive files, the shared generator drawn from in order)
                items.append((k, path, sig, g))
            order = sorted(range(len(items)), key=lambda i: (-items[i][2].shape[-1], i)) if can_sort else list(range(len(items)))
            results = {}
            for b0 in range(0, len(order), args.batch_size):
                grp = [items[i] for i in order[b0:b0 + args.batch_size]]
                rngs = [g for _, _, _, g in grp] if can_sort else rng
                enhs = model.enhance_many([sig for _, _, sig, _ in grp], rngs, pad_batch=args.pad_batch, **kw)
                for (k, _, _, _), e in zip(grp, enhs):
                    results[k] = resample(e, model.fs, fs)
        for k, path, _, _ in window:  # written in processing order
            output_path = out_path(path)
            save(output_path, results[k].cpu(), fs)
            done.append(output_path)

    window = []
    for k, path in todo:
        audio, fs = load(path)
        if window and (fs != window[0][3] or len(window) >= window_size):
            flush(window)
            window = []
        window.append((k, path, audio, fs))
    flush(window)
    return done


if __name__ == "__main__":
    main()
