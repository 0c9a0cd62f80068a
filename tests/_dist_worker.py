"""Worker for tests/test_distributed_cpu.py: one rank of a gloo job (CPU only).

Runs oavif_amd.batch.run_batch over a directory of PNGs with a scripted, GPU-free encode
function (the C++ search through the C ABI driven by a score table derived from the file),
so the sharding, the record packing and the single all_gather are exercised exactly as on
the GPU node, with gloo in place of RCCL."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def scripted_encode(i, path):
    from oavif_amd import tq
    seed = int(hashlib.sha256(path.name.encode()).hexdigest()[:8], 16)
    base, slope = 20.0 + seed % 30, 0.55 + (seed % 7) * 0.05
    if path.name.startswith("bad"):
        raise RuntimeError("scripted decode failure")
    r = tq.find_target_quality(lambda q: base + slope * q, score_tgt=80.0, tolerance=2.0, max_pass=6)
    return r.q, r.score, r.num_pass, 1000 + 10 * r.q + i


def main():
    images_dir, out_json = sys.argv[1], sys.argv[2]
    import torch.distributed as dist
    from oavif_amd import batch
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    from oavif_amd import hostinfo
    if world > 1:   # what batch.main does before its first GPU call
        hostinfo.pin_rank(rank, world)
        dist.init_process_group(backend="gloo")
    files = batch.list_images(images_dir)
    results = batch.run_batch(files, scripted_encode, rank, world, workers=int(os.environ.get('BATCH_WORKERS', '1')))
    deal = batch.deal_largest_first([p.stat().st_size for p in files], world)
    json.dump({"rank": rank, "affinity": sorted(os.sched_getaffinity(0)), "indices": deal[rank],
               "gather_rows_per_rank": max(len(d) for d in deal),
               "bytes": sum(files[i].stat().st_size for i in deal[rank])},
              open(f"{out_json}.rank{rank}", "w"))
    if rank == 0:
        batch.write_csv(out_json + ".csv", results)
        json.dump([[r.index, r.image, r.status, r.q, r.score, r.passes, r.orig_bytes, r.final_bytes]
                   for r in results], open(out_json, "w"))
        print(batch.summarize(results, 1.0, world))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
