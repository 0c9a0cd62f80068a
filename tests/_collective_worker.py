"""Worker for tests/test_collective.py: one rank of a gloo job (CPU only) doing what bench.py and the batch driver
do first with their process group -- oavif_amd.collective.check_in -- with the device description injected (there
is no GPU here): FAKE_BUS holds one PCI bus id per rank, FAKE_DEVCOUNT the visible device count, CLAIM_BACKEND the
backend the job claims ("nccl" = the rules of a real multi-GPU run; the bytes still travel over gloo).
Exit code 4 = the placement was refused, as bench.py / batch.py leave."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    import torch.distributed as dist
    from oavif_amd import collective, hostinfo
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    pinned = bool(hostinfo.pin_rank(rank, world).pinned)
    claim = os.environ.get("CLAIM_BACKEND", "nccl")
    why = None
    if os.environ.get("PREFLIGHT_DEVCOUNT"):   # collective.preflight with torch.cuda.device_count() replaced
        import torch
        torch.cuda.device_count = lambda: int(os.environ["PREFLIGHT_DEVCOUNT"])
        why = collective.preflight(claim, world)
        if why:
            json.dump({"preflight": why}, open(f"{out}.rank{rank}", "w"))
            return 4
    dist.init_process_group(backend="gloo")
    bus = os.environ["FAKE_BUS"].split(",")
    info = {"pci_bus_id": bus[rank] or None, "numa_node": rank % 2, "arch": "gfx950:sramecc+:xnack-"}
    coll, bad = collective.check_in(rank, rank, rank, claim, world, world, tensor_device=None, pinned=pinned,
                                    device_info=info, device_count=int(os.environ.get("FAKE_DEVCOUNT", str(world))))
    json.dump({"collective": coll, "problems": bad}, open(f"{out}.rank{rank}", "w"))
    dist.barrier()
    dist.destroy_process_group()
    return 4 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
