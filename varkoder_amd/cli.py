"""`python -m varkoder_amd image ...`: steps D+E of `varKoder image` on the GPU
(`python -m varkoder_amd convert ...`: `varKoder convert`, see convert.convert_folder).

Same flags as the reference's `image` sub-command (varKoder/cli.py:69-166).  Steps B/C of
the reference (fastp cleaning, reformat.sh subsampling) are external CPU tools outside this
path: run the reference once with `-X/--no-image -i INT` (clean + split only, cli.py:155-160)
and point this command at INT (or at its `split_fastqs/` folder).  Every file
`<sample>@<bp>K.fq[.gz]` becomes `<outdir>/<sample>@<bp>K+<mapping>+k<k>.png` with the
reference's metadata; `stats.csv` (and `labels.csv` with -t) are written like
process_stats does (commands/image.py:1144-1185).  With torchrun, ranks shard the files.
"""
import argparse
import math
import os
import shutil
import sys
from collections import OrderedDict, defaultdict
from pathlib import Path

from . import __version__
from .config import (DEFAULT_KMER_MAPPING, DEFAULT_KMER_SIZE, KMER_MAX, KMER_MIN, LABELS_SEP, MAPPING_CHOICES,
                     QUAL_THRESH, SAMPLE_BP_SEP)
from .image import eprint


def setup_parser():
    main = argparse.ArgumentParser(prog="varkoder_amd", formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                   description="MI355X-native k-mer counting and varKode/rfCGR imaging")
    sub = main.add_subparsers(required=True, dest="command")
    p = sub.add_parser("image", formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                       help="count k-mers and write images for already cleaned + split reads")
    p.add_argument("input", help="intermediate folder of a `varKoder image -X -i` run (or its split_fastqs/)")
    p.add_argument("-R", "--seed", type=int, help="random seed (accepted for parity; nothing here is random)")
    p.add_argument("-x", "--overwrite", action="store_true", help="overwrite existing results.")
    p.add_argument("-v", "--verbose", action="store_true", default=False)
    p.add_argument("-vv", "--version", action="version", version=f"%(prog)s {__version__}")
    p.add_argument("-k", "--kmer-size", type=int, default=DEFAULT_KMER_SIZE, help="size of kmers to count (5-9)")
    p.add_argument("-p", "--kmer-mapping", type=str, default=DEFAULT_KMER_MAPPING, choices=MAPPING_CHOICES)
    p.add_argument("-n", "--n-threads", type=int, default=1, help="host threads for file reading / PNG writing")
    p.add_argument("-c", "--cpus-per-thread", type=int, default=1, help="accepted for parity, unused")
    p.add_argument("-o", "--outdir", default="images", help="path to folder where to write final images.")
    p.add_argument("-f", "--stats-file", default="stats.csv", help="path to file where sample statistics will be saved.")
    p.add_argument("-i", "--int-folder", help="accepted for parity (the input IS the intermediate folder)")
    p.add_argument("-m", "--min-bp", type=str, default="500K", help="applied upstream by the split step; accepted for parity")
    p.add_argument("-M", "--max-bp", default="200M", help="applied upstream by the split step; accepted for parity")
    p.add_argument("-t", "--label-table", action="store_true", help="also write labels.csv")
    p.add_argument("-a", "--no-adapter", action="store_true", help="upstream (fastp) option; accepted for parity")
    p.add_argument("-D", "--no-deduplicate", action="store_true", help="upstream (fastp) option; accepted for parity")
    p.add_argument("-r", "--no-merge", action="store_true", help="upstream (fastp) option; accepted for parity")
    p.add_argument("-X", "--no-image", action="store_true", help="nothing to do here without images")
    p.add_argument("-T", "--trim-bp", default="10,10", help="upstream (fastp) option; accepted for parity")
    p.add_argument("--labels-csv", help="optional CSV `sample,labels` (labels separated by ';')")
    p.add_argument("--from-clean", action="store_true",
                   help="input holds cleaned but UNSPLIT reads (`<int>/clean_reads/<sample>.fq.gz`): draw the "
                        "1-2-5 ladder of subsamples between --min-bp and --max-bp on the GPU (stands in for "
                        "reformat.sh; statistically equivalent, not the same random reads)")
    q = sub.add_parser("query", formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                       help="Query cleaned reads or images against a trained network (cli.py:327-446).")
    q.add_argument("input", help="folder with cleaned reads (`<sample>.fq[.gz]`, e.g. <int>/clean_reads) or, with "
                                 "--images, with varKode / rfCGR png files")
    q.add_argument("outdir", help="path to the folder where results will be saved.")
    q.add_argument("-R", "--seed", type=int, help="random seed.")
    q.add_argument("-x", "--overwrite", action="store_true", help="overwrite existing results.")
    q.add_argument("-v", "--verbose", action="store_true", default=False)
    q.add_argument("-l", "--model", required=True,
                   help="local TorchScript archive or pickled torch module (fastai / hub models are not loadable here)")
    q.add_argument("--vocab", required=True, help="text file, one label per output unit of the model")
    q.add_argument("--single-label", action="store_true", help="softmax + best label instead of sigmoid >= threshold")
    q.add_argument("--input-size", type=int, default=224, help="side of the model's input (squish, BOX filter)")
    q.add_argument("--half", action="store_true", help="run the model under fp16 autocast")
    q.add_argument("-1", "--no-pairs", action="store_true", help="accepted for parity")
    q.add_argument("-I", "--images", action="store_true", help="input folder contains processed images instead of reads.")
    q.add_argument("-k", "--kmer-size", type=int, default=DEFAULT_KMER_SIZE, help="size of kmers to count (5-9)")
    q.add_argument("-p", "--kmer-mapping", type=str, default=DEFAULT_KMER_MAPPING, choices=MAPPING_CHOICES)
    q.add_argument("-n", "--n-threads", type=int, default=1, help="host threads for file reading / PNG writing")
    q.add_argument("-c", "--cpus-per-thread", type=int, default=1, help="accepted for parity, unused")
    q.add_argument("-f", "--stats-file", default="stats.csv", help="path to file where sample statistics will be saved.")
    q.add_argument("-d", "--threshold", type=float, default=0.7, help="confidence threshold to make a prediction.")
    q.add_argument("-i", "--int-folder", help="accepted for parity")
    q.add_argument("-m", "--keep-images", action="store_true",
                   help="whether barcode images should be saved to a directory named 'query_images'.")
    q.add_argument("-P", "--include-probs", action="store_true",
                   help="whether probabilities for each label should be included in the output.")
    q.add_argument("-a", "--no-adapter", action="store_true", help="upstream (fastp) option; accepted for parity")
    q.add_argument("-r", "--no-merge", action="store_true", help="upstream (fastp) option; accepted for parity")
    q.add_argument("-D", "--no-deduplicate", action="store_true", help="upstream (fastp) option; accepted for parity")
    q.add_argument("-T", "--trim-bp", default="10,10", help="upstream (fastp) option; accepted for parity")
    q.add_argument("-M", "--max-bp", default="200M",
                   help="number of post-cleaning basepairs to use for making image. Use '0' to use all of the available data.")
    q.add_argument("-b", "--max-batch-size", type=int, default=64, help="maximum batch size for predictions.")
    c = sub.add_parser("convert", formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                       help="Convert images between different kmer mappings.")          # cli.py:447-482
    c.add_argument("-R", "--seed", type=int, help="accepted for parity")
    c.add_argument("-x", "--overwrite", action="store_true", help="overwrite existing results.")
    c.add_argument("-v", "--verbose", action="store_true", default=False)
    c.add_argument("-n", "--n-threads", type=int, default=1, help="host threads for reading / writing PNG files")
    c.add_argument("-k", "--kmer-size", type=int, default=DEFAULT_KMER_SIZE, choices=[5, 6, 7, 8, 9],
                   help="size of kmers used to produce original images (as in the reference, this value "
                        "takes priority over the file names)")
    c.add_argument("-p", "--input-mapping", choices=MAPPING_CHOICES,
                   help="kmer mapping of input images. Will be inferred from file names if omitted.")
    c.add_argument("-r", "--sum-reverse-complements", action="store_true",
                   help="When converting from CGR to varKode, add together counts from canonical kmers and "
                        "their reverse complements.")
    c.add_argument("output_mapping", choices=MAPPING_CHOICES, help="kmer mapping of output images.")
    c.add_argument("input", help="path to folder with png images files to be converted.")
    c.add_argument("outdir", help="path to the folder where results will be saved.")
    return main


def run_convert(args):
    from .convert import convert_folder
    n = convert_folder(args.input, args.outdir, args.output_mapping, args.input_mapping, args.kmer_size,
                       args.sum_reverse_complements, args.overwrite, max(1, args.n_threads))
    eprint(f"Converted {n} images; written to {args.outdir}")


def run_query(args):
    """`varKoder query` from step C on (commands/query.py:188-324): images are made on the GPU and
    stay there for the model's input transform; predictions.csv has the reference's columns.
    Under a launcher (one process per GPU) the inputs are sharded by size over the ranks (shard.shard_by_size) -- every rank
    makes its images and runs the model on its GPU -- and rank 0 writes predictions.csv in input order
    (BASELINE config 5: images + batched inference on 8 GPUs; no data-path collective, one object gather)."""
    import numpy as np
    import torch
    from PIL import Image
    from . import query as Q
    from .convert import get_metadata_from_img_filename
    from .engine import ImageEngine
    from .image import write_png
    from .shard import agreed_weights, shard_by_size, world_info
    from .subsample import ladder_counts, split_name
    rank, world, device = world_info()
    outdir = Path(args.outdir)
    if not args.overwrite and (outdir / "predictions.csv").exists():
        raise Exception("Output directory exists, use --overwrite if you want to overwrite it.")
    if args.images:
        inputs = sorted(Path(args.input).rglob("*.png"))
    else:
        src = Path(args.input)
        if (src / "clean_reads").is_dir():
            src = src / "clean_reads"
        inputs = sorted(f for f in src.iterdir() if f.is_file() and f.name.endswith((".fq", ".fq.gz", ".fastq", ".fastq.gz")))
    if not inputs:
        raise Exception("No images found to query. Please check your input.")
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)        # control plane only
    # A rank that fails must not leave the others waiting in the gather: its work runs inside one try block, the
    # error travels WITH the gathered results, and every rank -- failed or not -- takes part in the gather and the
    # closing barrier; the exception is raised after the process group is gone.
    state = {"eng": None}
    model = vocab = None
    weights = agreed_weights(inputs)   # (a collective: outside the try block, every rank is still here)

    def rank_work():
        nonlocal model, vocab
        mine = shard_by_size(weights, rank, world)
        model, vocab = Q.load_model(args.model), Q.read_vocab(args.vocab)
        records, images, order = [], None, []   # order: index of each record's input in `inputs`
        eng = None
        if args.images:
            arrays = []
            for i in mine:
                p = inputs[i]
                im = Image.open(p)
                md = get_metadata_from_img_filename(p)
                labels, qual, sd = Q.image_metadata(im.info)
                records.append(dict(path=str(p), sample=md["sample"], bp=md["bp"], k=md["img_kmer_size"],
                                    mapping=md["img_kmer_mapping"], labels=labels, qual=qual, freq_sd=sd))
                arrays.append(np.array(im))
                order.append(i)
            shapes = {a.shape for a in arrays}
            if len(shapes) > 1:
                raise Exception("Images of different sizes in one query are not supported.")
            if arrays:
                eng = state["eng"] = ImageEngine(k=records[0]["k"], mapping="cgr", device=device)
                images = torch.from_numpy(np.stack(arrays)).to(eng.device)
        else:
            max_bp = None if str(args.max_bp) == "0" else parse_size(args.max_bp)
            rng = np.random.default_rng(args.seed)
            # one seed per sample in input order, as image.py:1017 -- drawn for every input on every rank
            seeds = [int(str(i) + str(rng.integers(low=0, high=2 ** 32))) % (1 << 63) for i in range(len(inputs))]
            hists, keep = [], []
            if mine:
                eng = state["eng"] = ImageEngine(k=args.kmer_size, mapping=args.kmer_mapping, device=device)
                files = [inputs[i] for i in mine]
                dev, offs, lens = eng.upload_files(files)
                for j, (i, f) in enumerate(zip(mine, files)):
                    sample = str(f.name.removesuffix("".join(f.suffixes)))
                    rec = ladder_counts(eng, dev, offs[j:j + 1], lens[j:j + 1], seed=seeds[i], max_bp=max_bp, is_query=True)[0]
                    if rec["error"] or not rec["steps"]:
                        eprint("SPLIT FAIL:", f, "-", rec["error"])
                        continue
                    bp, hist, _ = rec["steps"][0]
                    name = split_name(sample, bp) + f"+{args.kmer_mapping}+k{args.kmer_size}.png"
                    path = str(outdir / "query_images" / name) if args.keep_images else name
                    records.append(dict(path=path, sample=sample, bp=int(bp / 1000) * 1000, k=args.kmer_size,
                                        mapping=args.kmer_mapping, labels="", qual=bool("False"), freq_sd=0.0))
                    hists.append(hist)
                    keep.append(name)
                    order.append(i)
                if hists:
                    images = eng.images(torch.stack(hists))
                    if args.keep_images:
                        (outdir / "query_images").mkdir(parents=True, exist_ok=True)
                        host = images.cpu().numpy()
                        for j, name in enumerate(keep):
                            write_png(host[j], outdir / "query_images" / name, [], 0, QUAL_THRESH, args.kmer_mapping)
        if rank == 0:
            if args.single_label:
                eprint("This is a single label classification model, each input may will have only one prediction.")
            else:
                eprint("This is a multilabel classification model, each input may have 0 or more predictions.")
        probs = None
        if images is not None:
            probs = Q.probabilities(eng, images, model, batch_size=args.max_batch_size, multilabel=not args.single_label,
                                    input_size=args.input_size, half=args.half)
        return order, records, None if probs is None else np.asarray(probs)

    failure = None
    try:
        part = rank_work() + (None,)
    except Exception as e:   # noqa: BLE001 -- reported below, once every rank is past its collectives
        failure = e
        part = ([], [], None, "rank %d: %r" % (rank, e))
    parts = [part]
    if world > 1:
        import torch.distributed as dist
        bucket = [None] * world if rank == 0 else None
        dist.gather_object(part, bucket, dst=0)
        parts = bucket if rank == 0 else []
    if rank == 0:
        errors = [e for _, _, _, e in parts if e]
        try:
            if errors:
                raise failure if failure is not None else Exception("query failed on " + "; ".join(errors))
            rows = sorted(((i, r, p) for o, rs, ps, _ in parts if ps is not None for i, r, p in zip(o, rs, ps)), key=lambda t: t[0])
            if not rows:
                raise Exception("No images found to query. Please check your input.")
            df = Q.predictions_frame([r for _, r, _ in rows], np.stack([p for _, _, p in rows]), vocab, args.model,
                                     args.threshold, not args.single_label, args.include_probs)
            outdir.mkdir(parents=True, exist_ok=True)
            df.to_csv(outdir / "predictions.csv", index=False)
            eprint("Predictions saved to", str(outdir / "predictions.csv"))
        except Exception as e:   # noqa: BLE001
            failure = e
    if state["eng"] is not None:
        state["eng"].close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if failure is not None:
        raise failure


def read_labels(path):
    import pandas as pd
    if not path:
        return {}
    df = pd.read_csv(path, dtype=str).fillna("")
    return {r["sample"]: [x for x in r["labels"].split(LABELS_SEP) if x] for _, r in df.iterrows()}


def parse_size(text):
    """humanfriendly.parse_size for the forms the reference's flags take ("500K", "200M", "0"):
    decimal multiples, optional trailing B (commands/image.py:1013)."""
    t = str(text).strip().upper().removesuffix("B")
    mult = {"K": 10 ** 3, "M": 10 ** 6, "G": 10 ** 9, "T": 10 ** 12}
    if t and t[-1] in mult:
        return int(float(t[:-1]) * mult[t[-1]])
    return int(float(t))


def run_image_from_clean(args, outdir, rank, world, local_rank):
    import numpy as np
    from .pipeline import clean_to_images
    from .shard import io_threads_per_rank
    src = Path(args.input)
    if (src / "clean_reads").is_dir():
        src = src / "clean_reads"
    files = sorted(f for f in src.iterdir() if f.is_file() and f.name.endswith((".fq", ".fq.gz", ".fastq", ".fastq.gz")))
    if not files:
        raise Exception("No files found in input. Please check.")
    samples = [str(f.name.removesuffix("".join(f.suffixes))) for f in files]
    max_bp = None if str(args.max_bp) == "0" else parse_size(args.max_bp)     # cli.py:496-501
    rng = np.random.default_rng(args.seed)
    # image.py:1017: str(row index) + str(random integer), one draw per sample in row order
    seeds = {s: int(str(i) + str(rng.integers(low=0, high=2 ** 32))) % (1 << 63) for i, s in enumerate(samples)}
    labels = read_labels(args.labels_csv)
    from .image import base_sd_table
    base_sd = base_sd_table(src, samples)                                     # image.py:1094-1097
    eprint("Subsampling, counting kmers and creating images for", len(files), "samples")
    per_sample, error = OrderedDict(), None
    from .shard import agreed_weights
    weights = agreed_weights(files)   # (a collective: before the try block, while every rank is still here)
    try:   # (a rank whose share fails still reaches the gather below: see finish_image_job)
        failpoint(rank)
        per_sample = clean_to_images(files, outdir, weights=weights, k=args.kmer_size, mapping_code=args.kmer_mapping,
                                     min_bp=parse_size(args.min_bp), max_bp=max_bp, seeds=seeds, labels=labels,
                                     base_sd=base_sd,
                                     device=local_rank, rank=rank, world=world, io_threads=io_threads_per_rank(args.n_threads),
                                     verbose=args.verbose)
        for s, v in per_sample.items():
            v["base_frequencies_sd"] = base_sd.get(s, 0)
    except Exception as e:   # noqa: BLE001 -- reported by finish_image_job, once every rank is past its collectives
        error = e
    finish_image_job(args, outdir, rank, world, per_sample, error, samples, labels, base_sd)


def failpoint(rank):
    """Test hook: VARKODER_AMD_FAULT=rank<r> makes rank r's share of an `image` job raise before it starts (the tests of
    the job's failure isolation need a rank that fails for a reason no input file carries: a device out of memory, a
    full disk)."""
    if os.environ.get("VARKODER_AMD_FAULT") == "rank%d" % rank:
        raise RuntimeError("injected fault on rank %d (VARKODER_AMD_FAULT)" % rank)


def finish_image_job(args, outdir, rank, world, per_sample, error, samples, labels, base_sd):
    """The closing half of `image` on every rank: gather the ranks' per-sample stats (and errors) on rank 0, write
    stats.csv / labels.csv there, meet at the barrier, leave the group -- and only then raise what went wrong.

    The reference's pool loses ONE sample when a worker dies on it (commands/image.py:1070-1075, :1281-1284); a job of
    ranks that share collectives loses every rank unless the failing one still shows up for them: an exception on one
    rank (the device out of memory in a copy, a PNG that cannot be written) used to leave the others inside
    gather_object until the gloo timeout.  Now the error travels with the gathered object, every rank reaches the
    gather and the barrier, rank 0 still writes the stats of what did finish, and every rank exits non-zero."""
    import pandas as pd
    from .shard import gather_stats
    text = None if error is None else "rank %d: %r" % (rank, error)
    merged, errors = gather_stats(per_sample, error=text, with_errors=True)
    failure = error
    if rank == 0:
        try:
            rows = [OrderedDict([("sample", s)] + list(v.items())) for s, v in merged.items()]
            pd.DataFrame(rows).to_csv(args.stats_file, index=False)
            if args.label_table and not errors:                               # image.py:1172-1185
                lt = pd.DataFrame({"sample": samples,
                                   "labels": [LABELS_SEP.join(labels.get(s, [])) for s in samples],
                                   "possible_low_quality": [base_sd.get(s, 0) > QUAL_THRESH for s in samples]})
                lt.to_csv(outdir / "labels.csv", index=False)
            if errors:
                eprint("image failed on", "; ".join(errors))
                if failure is None:
                    failure = Exception("image failed on " + "; ".join(errors))
            else:
                eprint("All images done, saved in", str(outdir))
        except Exception as e:   # noqa: BLE001
            failure = failure or e
    if world > 1:
        import torch.distributed as dist
        # every rank learns whether the job failed (a rank whose own share was fine must not report success)
        flag = [failure is not None]
        verdict = [None] * world
        dist.all_gather_object(verdict, flag[0])
        dist.barrier()
        dist.destroy_process_group()
        if failure is None and any(verdict):
            failure = Exception("image failed on another rank")
    if failure is not None:
        raise failure


def run_image(args):
    from .pipeline import fastqs_to_images
    from .shard import io_threads_per_rank, world_info
    if args.kmer_size not in range(KMER_MIN, KMER_MAX + 1):
        raise ValueError("kmer size must be between 5 and 9")               # image.py:1209-1210
    rank, world, local_rank = world_info()
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)        # control plane only
    outdir = Path(args.outdir)
    refuse = [False]
    if rank == 0:
        refuse[0] = (not args.overwrite) and outdir.exists()                # image.py:1219-1223
        if not refuse[0]:
            if outdir.is_dir():
                shutil.rmtree(outdir)
            outdir.mkdir(parents=True)
    if world > 1:
        dist.broadcast_object_list(refuse, src=0)   # every rank learns the verdict: nobody is left in a barrier
    if refuse[0]:
        if world > 1:
            dist.destroy_process_group()
        raise Exception("Output directory exists, use --overwrite if you want to overwrite it.")
    if args.from_clean:
        return run_image_from_clean(args, outdir, rank, world, local_rank)
    src = Path(args.input)
    if (src / "split_fastqs").is_dir():
        src = src / "split_fastqs"
    files = sorted(f for f in src.iterdir() if SAMPLE_BP_SEP in f.name and f.is_file())
    if not files:
        raise Exception("No files found in input. Please check.")            # image.py:1309-1310
    samples = sorted({f.name.split(SAMPLE_BP_SEP)[0] for f in files})
    levels = math.floor(math.log(len(samples) / 1000, 16)) if samples else 0  # image.py:1246
    levels = max(levels, 0)
    labels = read_labels(args.labels_csv)
    from .image import base_sd_table
    # the fastp reports sit next to the split files in the intermediate folder (image.py:1094-1097)
    base_sd = base_sd_table(src.parent / "clean_reads", samples)
    eprint("varkoder_amd")
    eprint("Kmer size:", str(args.kmer_size))
    eprint("Counting kmers and creating images for", len(files), "files of", len(samples), "samples")
    if args.no_image:
        return
    mine, error = defaultdict(OrderedDict), None
    from .shard import agreed_weights
    weights = agreed_weights(files)   # (a collective: before the try block, while every rank is still here)
    try:   # (a rank whose share fails still reaches the gather: see finish_image_job)
        failpoint(rank)
        per_file = fastqs_to_images(files, outdir, weights=weights, k=args.kmer_size, mapping_code=args.kmer_mapping, labels=labels,
                                    base_sd=base_sd, overwrite=True, subfolder_levels=levels, device=local_rank, rank=rank,
                                    world=world, io_threads=io_threads_per_rank(args.n_threads), verbose=args.verbose)
        # fold the per-file stats into per-sample stats like run_clean2img does (image.py:1057-1125); files are dealt by
        # size, so other ranks may hold further files of the same sample: shard.merge_stats adds the ranks' shares up
        ck, ik = f"{args.kmer_size}mer_counting_time", f"k{args.kmer_size}_img_time"
        if os.environ.get("VARKODER_AMD_PER_FILE_STATS"):   # (tests: this rank's per-file rows, before they are folded)
            import json
            with open(os.environ["VARKODER_AMD_PER_FILE_STATS"] + f".rank{rank}.json", "w") as f:
                json.dump(per_file, f)
        for key, st in per_file.items():
            s = mine[key.split(SAMPLE_BP_SEP)[0]]
            for name in (ck, ik):
                if name in st:
                    s[name] = s.get(name, 0) + st[name]
            if "failed_step" in st:
                s["failed_step"] = st["failed_step"]
        for name, st in mine.items():
            st["base_frequencies_sd"] = base_sd.get(name, 0)
    except Exception as e:   # noqa: BLE001 -- reported by finish_image_job, once every rank is past its collectives
        error = e
    finish_image_job(args, outdir, rank, world, mine, error, samples, labels, base_sd)


def main(argv=None):
    args = setup_parser().parse_args(argv)
    if not Path(args.input).exists():                                       # cli.py:503-505
        raise Exception("Input path", args.input, "does not exist. Please check.")
    if args.command == "image":
        run_image(args)
    elif args.command == "convert":
        run_convert(args)
    elif args.command == "query":
        run_query(args)
    eprint("DONE")


if __name__ == "__main__":
    main(sys.argv[1:])
