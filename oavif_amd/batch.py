"""Batch driver: the MI355X-side counterpart of /root/reference/scripts/measure.py.

measure.py runs the `oavif` binary once per image, one after another (measure.py:151-158).
Here every rank of a `torch.distributed` job (one process per GPU) takes its share of the same
sorted list (measure.py:137-140) -- dealt largest file first to the least loaded rank, so the
ranks' byte loads are even ("sort/largest-first if sizes vary") -- runs the
target-quality search for each (CPU libavif/aom encode + dav1d decode through oavif_amd.avif_bridge, scorer on
this rank's GPU) on host cores near that GPU (the rank pins itself to its slice of the node's
cores before its first GPU call, oavif_amd.hostinfo), and the per-image result records are
gathered to every rank with ONE all_gather (RCCL when the backend is nccl).  There is no
data-path collective: images are independent.

Kept from the reference: image selection (.png/.jpg/.jpeg, sorted), CSV header, column order
and number formats (measure.py:178-206), the summary statistics (measure.py:209-269), the
"--tolerance" / "--keep" flags, and main.zig's rule that the last probe's bytes are reused
only when its quantizer is the chosen one (main.zig:109-113).

    python -m oavif_amd.batch IMAGES_DIR [OAVIF_PATH] OUTPUT_CSV [--tolerance T] [--keep]
    python -m oavif_amd.batch --gpus 8 IMAGES_DIR OUTPUT_CSV        (one command, as measure.py is: starts its own 8 ranks)
    python -m torch.distributed.run --nproc-per-node 8 -m oavif_amd.batch IMAGES_DIR OUTPUT_CSV     (the same ranks under a launcher)

The positional arguments are measure.py's (measure.py:111-123: images_dir oavif_path output_csv),
so an existing invocation keeps working; OAVIF_PATH is accepted and not used (the search runs in
this process instead of one `oavif` process per image), and the two-positional form omits it.
Every image goes through the same load / encode path as the CLI mirror (oavif_amd.cli): the
source's own channels (alpha included), ICC profile, 16-bit >> 8, and the reference's encoder
defaults (one encoder thread, parse_args.zig:51).
"""
from __future__ import annotations

import argparse
import csv
import os
import statistics
import sys
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, List, Optional, Sequence

import numpy as np

IMAGE_EXTENSIONS = {".png", ".jpg", ".jpeg"}  # measure.py:136
CSV_HEADER = ["Image", "Original Bytes", "Final Bytes", "Savings Bytes", "Savings %",
              "Encoding Time (ms)", "Passes", "Status", "Error"]  # measure.py:181-191
STATUS = ["ok", "no-output", "error"]
RECORD_FIELDS = 8  # idx, status, q, score, passes, orig_bytes, final_bytes, time_ms


@dataclass
class ImageResult:
    index: int
    image: str
    status: str = "ok"
    q: int = 0
    score: float = 0.0
    passes: Optional[int] = None
    orig_bytes: int = 0
    final_bytes: Optional[int] = None
    encoding_time_ms: Optional[float] = None
    error: str = ""

    @property
    def savings_bytes(self) -> Optional[int]:
        if self.final_bytes is None or self.orig_bytes == 0:
            return None
        return max(self.orig_bytes - self.final_bytes, 0)

    @property
    def savings_pct(self) -> Optional[float]:
        s = self.savings_bytes
        return None if s is None else s / self.orig_bytes * 100.0


def list_images(images_dir) -> List[Path]:
    d = Path(images_dir)
    return sorted(f for f in d.iterdir() if f.is_file() and f.suffix.lower() in IMAGE_EXTENSIONS)


def shard(n_items: int, rank: int, world: int) -> List[int]:
    """Image i goes to rank i mod world (equal-sized inputs)."""
    return list(range(rank, n_items, world))


def deal_largest_first(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time dealing: images in order of decreasing size (ties: lower index
    first), each to the rank with the smallest byte load so far (ties: lower rank).  Returns the
    index list of every rank, each in dealing order (largest first, so a rank's worker threads
    start its long encodes first and the tail is short).  Deterministic: every rank computes the
    same table from the same directory listing.  With equal sizes this is i mod world."""
    world = max(1, int(world))
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(sizes)), key=lambda k: (-int(sizes[k]), k)):
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += int(sizes[i])
    return out


# ---- one image ---------------------------------------------------------------------------------

def encode_image(scorer, path: Path, out_path: Optional[Path], score_tgt: float = 80.0,
                 tolerance: float = 2.0, max_pass: int = 6, speed: int = 9, options=None):
    """main.zig:73-116 for one file with the search on the GPU scorer, through the CLI mirror's
    own load and encode functions (one path for `python -m oavif_amd.cli` and the batch).

    Returns (q, score, passes, final_bytes)."""
    from . import cli, tq
    o = options
    if o is None:
        o = cli.AvifEncOptions()   # the reference's defaults (parse_args.zig:48-63): 1 thread, auto tiling
        o.speed, o.score_tgt, o.tolerance, o.max_pass = speed, score_tgt, tolerance, max_pass
    src = cli.load_source(str(path))
    prepared = cli.encoder_input(src.pixels, o, src.icc) if cli._bridge_on() else None   # hoisted out of the pass loop
    cache = {}

    def codec(q: int):
        data = cli._encode(src.pixels, o, q, icc=src.icc, prepared=prepared)
        cache.clear()
        cache[q] = data                      # EncBuffer holds only the last probe (tq.zig:31-35)
        return cli._decode_rgb(data), len(data)

    try:
        if cli._bridge_on():
            from . import avif_bridge

            def codec_frame(q: int):             # the decoded frame stays in libavif's buffer
                data = cli._encode(src.pixels, o, q, icc=src.icc, prepared=prepared)
                cache.clear()
                cache[q] = data
                return avif_bridge.decode_common(data), len(data)
            r = tq.search_hip_frames(scorer, src.rgb, codec_frame, score_tgt=o.score_tgt, tolerance=o.tolerance,
                                     max_pass=o.max_pass)
        else:
            r = tq.search_hip(scorer, src.rgb, codec, score_tgt=o.score_tgt, tolerance=o.tolerance,
                              max_pass=o.max_pass)
        data = cache.get(r.q) if r.buf_q == r.q else None
        if data is None:                         # main.zig:109-113: re-encode at the chosen q
            data = cli._encode(src.pixels, o, r.q, icc=src.icc, prepared=prepared)
    finally:
        if prepared is not None:
            prepared.close()
    if out_path is not None:
        out_path.write_bytes(data)
    return r.q, r.score, r.num_pass, len(data)


def exec_image(oavif_path: str, path: Path, out_path: Path, tolerance: Optional[float] = None, env=None):
    """measure.py:41-107 for one file: run an `oavif` executable -- the reference's, or this repo's compiled
    host (oavif_amd/lib/oavif_host) -- as `oavif [--tolerance T] <in> <out.avif>` and read the pass count off
    its stderr with measure.py's own expression (measure.py:27).  q and score come from the same line
    (main.zig:106); they are not in measure.py's CSV.  Returns (q, score, passes, final_bytes)."""
    import re
    import subprocess
    cmd = [str(oavif_path)]
    if tolerance is not None:
        cmd += ["--tolerance", str(tolerance)]
    cmd += [str(path), str(out_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    if r.returncode != 0:   # measure.py:94: check=True -> CalledProcessError
        raise RuntimeError(f"Command {cmd!r} returned non-zero exit status {r.returncode}: {(r.stderr or '').strip()[-300:]}")
    m = re.search(r"(\d+)\s+passes?", r.stderr or "", re.IGNORECASE)
    f = re.search(r"Found q(\d+) \(score (-?\d+(?:\.\d+)?)", r.stderr or "")
    final_bytes = out_path.stat().st_size if out_path.exists() else None
    return (int(f.group(1)) if f else 0, float(f.group(2)) if f else 0.0, int(m.group(1)) if m else -1, final_bytes)


def output_names(image_files: Sequence[Path]) -> List[str]:
    """One .avif name per input: `<stem>.avif` as measure.py writes it (measure.py:49), or
    `<stem>_<ext>.avif` for inputs whose stem is shared (a.png + a.jpg); a name that is still
    taken (a.png + a.jpg + a_png.webp-style stems, a.JPG + a.jpg) gets the input's position in
    the list appended.  The names are unique among themselves, so concurrent workers / ranks
    never write or unlink the same file."""
    stems = [p.stem for p in image_files]
    names = [f"{p.stem}.avif" if stems.count(p.stem) == 1 else f"{p.stem}_{p.suffix.lstrip('.').lower()}.avif"
             for p in image_files]
    taken = set()
    for i, name in enumerate(names):
        if name in taken or names.count(name) > 1:
            name = f"{name[:-5]}_{i}.avif"
            while name in taken:
                name = f"{name[:-5]}_.avif"
            names[i] = name
        taken.add(name)
    return names


# ---- gather ----------------------------------------------------------------------------------------

def pack_records(results: Sequence[ImageResult]) -> np.ndarray:
    rec = np.zeros((len(results), RECORD_FIELDS), np.float64)
    for i, r in enumerate(results):
        rec[i] = [r.index, STATUS.index(r.status), r.q, r.score,
                  -1 if r.passes is None else r.passes, r.orig_bytes,
                  -1 if r.final_bytes is None else r.final_bytes,
                  -1.0 if r.encoding_time_ms is None else r.encoding_time_ms]
    return rec


def gather_records(local: np.ndarray, n_total: int, device=None, per_rank: Optional[int] = None,
                   always: bool = False) -> np.ndarray:
    """All-gather the (n_local, 8) float64 record arrays of all ranks -> (n_total, 8) sorted by
    image index.  One collective; works on gloo (CPU tensors) and nccl = RCCL (GPU tensors).
    `per_rank`: rows of the fixed-size buffer every rank contributes (the largest shard).
    `always`: run the collective even in a process group of one rank (a one-GPU box can then push
    the device-tensor all_gather through RCCL: tests/test_gpu_batch.py; OAVIF_GATHER_ALWAYS=1 for
    the batch driver) instead of returning the local records directly."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not always):
        return local[np.argsort(local[:, 0])] if len(local) else local
    world = dist.get_world_size()
    if per_rank is None:
        per_rank = (n_total + world - 1) // world
    buf = torch.full((per_rank, RECORD_FIELDS), -2.0, dtype=torch.float64)
    if len(local):
        buf[: len(local)] = torch.from_numpy(local)
    if device is not None:
        buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    allrec = torch.cat(out).cpu().numpy()
    allrec = allrec[allrec[:, 0] >= 0]
    return allrec[np.argsort(allrec[:, 0])]


# ---- CSV + summary (formats of measure.py:178-269) ---------------------------------------------------

def records_to_results(rec: np.ndarray, names: Sequence[str], errors=None) -> List[ImageResult]:
    out = []
    for row in rec:
        i = int(row[0])
        out.append(ImageResult(
            index=i, image=names[i], status=STATUS[int(row[1])], q=int(row[2]), score=float(row[3]),
            passes=None if row[4] < 0 else int(row[4]), orig_bytes=int(row[5]),
            final_bytes=None if row[6] < 0 else int(row[6]),
            encoding_time_ms=None if row[7] < 0 else float(row[7]),
            error=(errors or {}).get(i, "")))
    return out


def write_csv(path, results: Sequence[ImageResult]) -> None:
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(CSV_HEADER)
        for m in results:
            w.writerow([
                m.image, m.orig_bytes,
                m.final_bytes if m.final_bytes is not None else "",
                m.savings_bytes if m.savings_bytes is not None else "",
                f"{m.savings_pct:.2f}" if m.savings_pct is not None else "",
                f"{m.encoding_time_ms:.2f}" if m.encoding_time_ms is not None else "",
                m.passes if m.passes is not None else "",
                m.status, m.error or ""])


def human_bytes(n: float) -> str:
    """`1536 -> "1.50 KiB"`: binary units up to TiB, two decimals (the format of measure.py's
    summary lines, measure.py:31-38)."""
    units = ("B", "KiB", "MiB", "GiB", "TiB")
    k = 0
    v = float(n)
    while v >= 1024.0 and k < len(units) - 1:
        v /= 1024.0
        k += 1
    return f"{v:.2f} {units[k]}"


def summarize(results: Sequence[ImageResult], wall_s: float, world: int = 1) -> str:
    ok = [m for m in results if m.status == "ok"]
    errors = [m for m in results if m.status == "error"]
    no_out = [m for m in results if m.status == "no-output"]
    times = [m.encoding_time_ms for m in ok if m.encoding_time_ms is not None]
    passes = [m.passes for m in ok if m.passes is not None]
    orig_total = sum(m.orig_bytes for m in ok)
    final_total = sum(m.final_bytes for m in ok if m.final_bytes is not None)
    savings_total = max(orig_total - final_total, 0) if ok else 0
    ratios = [m.final_bytes / m.orig_bytes for m in ok if m.final_bytes is not None and m.orig_bytes > 0]
    lines = ["", "Run Summary",
             f"Images: {len(ok)} ok, {len(no_out)} no-output, {len(errors)} errors",
             f"Ranks (GPUs): {world}",
             f"Total wall time: {wall_s:.2f} s",
             f"Throughput: {(len(ok) / wall_s) if wall_s > 0 else 0.0:.2f} images/s",
             f"Input bytes throughput: {human_bytes(orig_total / wall_s if wall_s > 0 else 0)}/s",
             f"Output bytes throughput: {human_bytes(final_total / wall_s if wall_s > 0 else 0)}/s",
             "", "Compression Totals",
             f"Original total bytes: {orig_total} ({human_bytes(orig_total)})",
             f"Final total bytes:    {final_total} ({human_bytes(final_total)})",
             f"Savings (bytes):      {savings_total} ({human_bytes(savings_total)})",
             f"% saved (overall):    {(savings_total / orig_total * 100.0) if orig_total else 0.0:.2f}%"]
    if ratios:
        try:
            lines.append(f"% saved (geometric mean across files): "
                         f"{(1.0 - statistics.geometric_mean(ratios)) * 100.0:.2f}%")
        except ValueError:
            pass
    if times:
        sd = statistics.stdev(times) if len(times) > 1 else 0.0
        psd = statistics.stdev(passes) if len(passes) > 1 else 0.0
        lines += ["", "Timing & Passes",
                  f"Average encoding time: {sum(times) / len(times):.2f} ms ± {sd:.2f}",
                  f"Median encoding time:  {statistics.median(times):.2f} ms",
                  f"Average passes:        {(sum(passes) / len(passes)) if passes else 0.0:.2f} ± {psd:.2f} "
                  f"(max: {max(passes) if passes else 0}, min: {min(passes) if passes else 0})"]
    return "\n".join(lines)


# ---- driver ----------------------------------------------------------------------------------------

def run_batch(image_files: Sequence[Path], encode_fn: Callable[[int, Path], tuple], rank: int = 0,
              world: int = 1, gather_device=None, log=None, workers: int = 1,
              deal: Optional[List[List[int]]] = None, gather_always: bool = False) -> List[ImageResult]:
    """Process this rank's shard with `encode_fn(index, path) -> (q, score, passes, final_bytes)`
    and return the gathered, index-sorted results of ALL ranks.  `deal`: the index list of every
    rank (default: deal_largest_first over the files' sizes).

    `workers` > 1 runs that many images of the shard concurrently in threads: the CPU codec
    (libavif/aom, called with the GIL released) is 3-4 orders of magnitude slower than the GPU
    score, so one image at a time leaves both the host cores and the GPU idle.  `encode_fn` must
    then be thread-safe (one scorer context per thread: contexts are not re-entrant)."""
    errors = {}

    def one(i: int) -> ImageResult:
        path = image_files[i]
        res = ImageResult(index=i, image=path.name, orig_bytes=path.stat().st_size)
        t0 = time.perf_counter()
        try:
            q, score, passes, final_bytes = encode_fn(i, path)
            res.q, res.score, res.passes, res.final_bytes = int(q), float(score), int(passes), final_bytes
            res.encoding_time_ms = (time.perf_counter() - t0) * 1000.0
            res.status = "ok" if final_bytes is not None else "no-output"
            if log:
                # the reference's stderr contract (main.zig:106; parsed by measure.py:27)
                log(f"[rank {rank}] {path.name}: Found q{res.q} (score {res.score:.2f}, {res.passes} passes)")
        except Exception as e:  # measure.py:94-107: record and continue
            res.status = "error"
            res.error = f"Error processing {path}: {e}"
            errors[i] = res.error
            if log:
                log(res.error)
        return res

    if deal is None:
        deal = deal_largest_first([_size_or_zero(p) for p in image_files], world)
    mine = deal[rank]
    if workers > 1 and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            local = list(ex.map(one, mine))
    else:
        local = [one(i) for i in mine]
    rec = gather_records(pack_records(local), len(image_files), gather_device,
                         per_rank=max(len(d) for d in deal), always=gather_always)
    names = [p.name for p in image_files]
    results = records_to_results(rec, names)
    for r in results:  # error strings stay on the rank that produced them; keep local ones
        if r.index in errors:
            r.error = errors[r.index]
        elif r.status == "error":
            r.error = "error on another rank (see its log)"
    return results


def _size_or_zero(p: Path) -> int:
    try:
        return p.stat().st_size
    except OSError:
        return 0


def default_workers() -> int:
    """Images encoded concurrently per rank: this rank's share of the host cores the job may use
    (cgroup quota / affinity mask, not os.cpu_count(): the GPU host has 256 threads, a job's
    cgroup 16 per GPU), one encoder thread per image as oavif defaults (parse_args.zig:51)."""
    from . import hostinfo
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    return max(1, min(16, hostinfo.usable_cores(cap=16 * local_world) // local_world))


def parse_cli(argv=None):
    """measure.py's command line (measure.py:111-123) plus the search options."""
    ap = argparse.ArgumentParser(description="Target-quality AVIF encoding of a directory of images, "
                                             "sharded over the GPUs of one node")
    ap.add_argument("paths", nargs="+", metavar="PATH",
                    help="IMAGES_DIR OUTPUT_CSV, or measure.py's IMAGES_DIR OAVIF_PATH OUTPUT_CSV "
                         "(OAVIF_PATH is accepted for compatibility and not used)")
    ap.add_argument("--tolerance", type=float, default=2.0)   # parse_args.zig:58
    ap.add_argument("--score-tgt", type=float, default=80.0)  # parse_args.zig:55
    ap.add_argument("--max-pass", type=int, default=6)        # parse_args.zig:59
    ap.add_argument("--speed", type=int, default=9)           # parse_args.zig:50
    ap.add_argument("--keep", action="store_true", help="keep the generated .avif files")
    ap.add_argument("--gpus", type=int, default=None, metavar="N",
                    help="shard the images over N GPUs of this node, one rank per GPU (times --procs-per-gpu): the command "
                         "starts its own ranks (oavif_amd/launch.py) unless a launcher such as torch.distributed.run has "
                         "already announced a world.  Default: 1, or the launcher's WORLD_SIZE")
    ap.add_argument("--workers", type=int, default=default_workers(),
                    help="images encoded concurrently per rank (threads; one scorer context each); "
                         "default: the rank's share of the usable host cores")
    ap.add_argument("--no-pin", action="store_true",
                    help="do not pin the rank to its slice of the node's host cores")
    ap.add_argument("--pin", choices=("slice", "idle"), default=os.environ.get("OAVIF_PIN", "slice") or "slice",
                    help="slice (default): the rank's fixed contiguous slice of the cores near its GPU; idle: a "
                         "single rank on a host shared with other tenants may take the idlest cores near its GPU "
                         "(sampled for 1 s) instead")
    ap.add_argument("--procs-per-gpu", type=int, default=int(os.environ.get("OAVIF_PROCS_PER_GPU", "1") or 1),
                    help="ranks that share one GPU (launch nproc-per-node = GPUs x this): the CPU codec is the "
                         "cost of a pass and several processes per GPU use the host's cores better than one "
                         "process with as many threads; the gather of such a job runs over gloo")
    ap.add_argument("--exec", dest="exec_oavif", action="store_true",
                    help="run OAVIF_PATH once per image exactly as measure.py does (measure.py:41-107), in this rank's "
                         "shard and on this rank's GPU (LOCAL_RANK is handed to the child), instead of searching in "
                         "this process: e.g. oavif_amd/lib/oavif_host, the compiled C host of this repository")
    ap.add_argument("--out-dir", default="temp_avif_output")
    ap.add_argument("--collective-json", default=None, metavar="PATH",
                    help="rank 0 writes the job's `collective` record (oavif_amd/collective.py: backend, world size, every "
                         "rank's device / PCI bus id / NUMA node / pinned cores, library versions) and the run's totals "
                         "(images, wall seconds, images per second) to PATH")
    args = ap.parse_args(argv)
    args.oavif_path = None
    if len(args.paths) == 2:
        args.images_dir, args.output_csv = args.paths
        if args.exec_oavif:
            ap.error("--exec needs OAVIF_PATH: IMAGES_DIR OAVIF_PATH OUTPUT_CSV")
    elif len(args.paths) == 3:   # measure.py:111-123
        args.images_dir, oavif_path, args.output_csv = args.paths
        args.oavif_path = oavif_path
        if int(os.environ.get("RANK", "0")) == 0 and not args.exec_oavif:
            print(f"note: {oavif_path} is not run; the search runs in this process on the GPU scorer",
                  file=sys.stderr)
    else:
        ap.error("expected IMAGES_DIR [OAVIF_PATH] OUTPUT_CSV")
    return args


def main(argv=None) -> int:
    args = parse_cli(argv)
    # One command, as measure.py is one command (measure.py:110-158): `--gpus N` without a launcher's world in the
    # environment starts N x --procs-per-gpu fresh ranks of this very command before torch is imported or a GPU touched,
    # relays rank 0's summary and leaves with the first non-zero rank code (oavif_amd/launch.py).
    from . import launch
    ppg_cli = max(1, int(args.procs_per_gpu))
    if args.gpus is not None and launch.needs_self_launch(args.gpus * ppg_cli):
        return launch.spawn_ranks([sys.executable, "-m", "oavif_amd.batch"] + list(sys.argv[1:] if argv is None else argv),
                                  args.gpus * ppg_cli, label="oavif_amd.batch")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if args.gpus is not None and args.gpus * ppg_cli != world:
        print(f"oavif_amd.batch: --gpus {args.gpus} x --procs-per-gpu {ppg_cli} but the launcher announced WORLD_SIZE={world}",
              file=sys.stderr)
        return 2
    # Host placement first -- before torch is imported, before any GPU call, before a thread
    # exists: every thread started later (encoder workers, the HIP runtime's) inherits the mask.
    from . import hostinfo
    ppg = max(1, int(args.procs_per_gpu))
    core_sets = hostinfo.node_core_sets(local_world, procs_per_gpu=ppg)
    # one rank alone is pinned too when the cgroup grants fewer CPUs than the affinity mask holds:
    # a quota is enforced by throttling, and threads that float over the whole host hit it
    quota = hostinfo.cgroup_cpu_quota()
    want_pin = not args.no_pin and (world > 1 or (quota is not None and quota < len(hostinfo.allowed_cpus())))
    pinned, pin_note = False, "not pinned"
    if want_pin:
        mine = hostinfo.pin_rank(local_rank, local_world, procs_per_gpu=ppg, idle=(args.pin == "idle"))
        pinned = mine.pinned
        if not mine.pinned:   # said, not swallowed: the summary then reads "not pinned"
            pin_note = f"not pinned: sched_setaffinity failed on rank {rank} ({mine.error})"
            print(f"oavif_amd.batch: rank {rank}: {pin_note}", file=sys.stderr)
        if world == 1:
            core_sets = [list(mine)]

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        print("oavif_amd.batch: no GPU visible; the scorer has no CPU fallback", file=sys.stderr)
        return 3
    # One process per GPU over RCCL (backend "nccl").  With --procs-per-gpu K > 1, K consecutive
    # local ranks share a GPU (rank r -> GPU r // K) and the one gather of 64-byte records runs
    # over gloo on CPU tensors: RCCL does not place two ranks on one device.  OAVIF_BENCH_BACKEND=gloo
    # is the rehearsal mode bench.py also has, for boxes with fewer GPUs than ranks (local_rank
    # modulo the device count).
    backend = os.environ.get("OAVIF_BENCH_BACKEND", "gloo" if ppg > 1 else "nccl")
    from . import collective
    if world > 1:
        why = collective.preflight(backend, local_world)
        if why:   # every rank of the host sees the same count and leaves with the same code
            print(f"oavif_amd.batch: rank {rank}: refusing to run: {why}", file=sys.stderr)
            return 4
    launcher_local_rank = local_rank
    local_rank = local_rank // ppg
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # OAVIF_GATHER_ALWAYS=1: a single rank still opens its process group and sends its records through
    # the collective (what a one-GPU box can exercise of the RCCL path: the same call on device tensors)
    gather_always = os.environ.get("OAVIF_GATHER_ALWAYS", "") == "1"
    if world == 1 and gather_always:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    # Before any communicator exists: every rank's description of itself through the rendezvous store, judged on every
    # rank (collective.open_group).  Two RCCL ranks on one GPU -- the case RCCL answers with a hang -- or a rank set that
    # is not what the launcher announced end the run here on every rank (rc 4), no communicator ever created; then the
    # process group is opened on that store and one all_gather over it confirms the records.  An RCCL failure at that
    # point ends every rank with rc 5 and RCCL's own message.
    coll_group = world > 1 or gather_always
    if coll_group:
        coll, rc_ = collective.open_group(rank, launcher_local_rank, local_rank, backend, world, local_world,
                                          pinned=pinned if want_pin else None, label="oavif_amd.batch")
    else:
        coll, bad_ = collective.check_in(rank, launcher_local_rank, local_rank, backend, world, local_world,
                                         pinned=pinned if want_pin else None, grouped=False)
        rc_ = collective.RC_REFUSED if bad_ else 0
        if bad_:
            print("oavif_amd.batch: refusing to run:\n  " + "\n  ".join(bad_), file=sys.stderr)
    if ppg > 1:
        coll["ranks_per_gpu"] = ppg
    if rc_:
        return rc_

    files = list_images(args.images_dir)
    if not files:
        print(f"No images found in {args.images_dir}", file=sys.stderr)
        return 1
    out_dir = Path(args.out_dir)
    out_dir.mkdir(exist_ok=True)

    import threading

    from . import Ssimu2
    tls = threading.local()
    all_scorers = []

    names = output_names(files)

    child_env = None
    if args.exec_oavif:
        from . import avif_bridge
        child_env = dict(os.environ, LOCAL_RANK=str(local_rank))        # the child scores on this rank's GPU
        if "OAVIF_LIBAVIF" not in child_env and avif_bridge._find_library():
            child_env["OAVIF_LIBAVIF"] = avif_bridge._find_library()    # the C host opens libavif by this name

    def encode_fn(i, path):
        if args.exec_oavif:
            q, score, passes, nbytes = exec_image(args.oavif_path, path, out_dir / names[i],
                                                  None if args.tolerance == 2.0 else args.tolerance, child_env)
            return q, score, (passes if passes >= 0 else 0), nbytes
        if not hasattr(tls, "scorer"):  # one context (HIP stream + scratch) per worker thread
            from . import cli
            tls.scorer = Ssimu2(local_rank, blur=cli.blur_from_env())
            all_scorers.append(tls.scorer)
        return encode_image(tls.scorer, path, out_dir / names[i], args.score_tgt,
                            args.tolerance, args.max_pass, args.speed)

    if rank == 0:
        print(f"Found {len(files)} images. Starting encoding on {world} GPU(s)...", file=sys.stderr)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    results = run_batch(files, encode_fn, rank, world,
                        gather_device=torch.device("cuda", local_rank) if (world > 1 or gather_always) and backend == "nccl" else None,
                        log=lambda s: print(s, file=sys.stderr), workers=args.workers, gather_always=gather_always)
    wall = time.perf_counter() - t0
    if not args.keep:
        for i in deal_largest_first([_size_or_zero(p) for p in files], world)[rank]:
            try:
                (out_dir / names[i]).unlink()
            except OSError:
                pass
    if rank == 0:
        write_csv(args.output_csv, results)
        print(summarize(results, wall, world))
        print(f"Host cores per rank{'' if pinned else ' (' + pin_note + ')'}: " + "; ".join(
            f"rank {r}: {len(cs)} ({hostinfo.format_cpus(cs)})" for r, cs in enumerate(core_sets)))
        print(f"Worker threads per rank: {args.workers}; ranks per GPU: {ppg}; dealing: largest file first")
        print(f"Collective: backend {coll['backend']}, world {coll['world_size']}, {coll['distinct_devices']} distinct device(s): "
              + "; ".join(f"rank {r_.get('rank')} -> GPU {r_.get('device')} at {r_.get('pci_bus_id')} (NUMA {r_.get('numa_node')})"
                          for r_ in coll["ranks"]))
        print(f"Collective record exchanged through: {coll['gathered_through']}")
        if args.collective_json:
            import json
            ok_n = sum(1 for r_ in results if r_.status == "ok")
            with open(args.collective_json, "w") as f:
                json.dump({"collective": coll, "images": len(results), "images_ok": ok_n, "wall_s": round(wall, 3),
                           "images_per_s": round(ok_n / wall, 3) if wall > 0 else None,
                           "workers_per_rank": args.workers, "ranks_per_gpu": ppg}, f, indent=1)
        from . import cli
        depth, note = cli.codec_depth(cli.AvifEncOptions().tenbit, False)   # the batch runs the reference's defaults
        if note:
            print("Note: oavif defaults to 10-bit AVIF (parse_args.zig:56) and this host's libaom writes 8-bit only: "
                  "byte sizes and chosen quantizers are not those of the reference's measure.py run")
        print(f"\nResults written to {args.output_csv}")
    for sc in all_scorers:
        sc.close()
    if world > 1 or gather_always:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
