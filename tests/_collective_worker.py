"""Worker for tests/test_collective.py: one rank of a CPU-only job doing what bench.py and the batch driver do before
their first collective -- oavif_amd.collective.open_group -- with the device description injected (there is no GPU
here): FAKE_BUS holds one PCI bus id per rank, FAKE_DEVCOUNT the visible device count, CLAIM_BACKEND the backend the
job claims ("nccl" = the rules of a real multi-GPU run).  The process group itself is a gloo one, opened by a stand-in
for RCCL's communicator creation that FAILS when two ranks claim one bus id (RCCL answers that placement with a hang or
an obscure error): it leaves a marker file when it is called, so the test can see that a refused placement never
reached it.  FAIL_INIT=1 makes the stand-in fail on a legal placement (an RCCL init failure: rc 5 on every rank).
Exit code 4 = the placement was refused, 5 = the process group failed, as bench.py / batch.py leave."""
import datetime
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
    bus = os.environ["FAKE_BUS"].split(",")
    info = {"pci_bus_id": bus[rank] or None, "numa_node": rank % 2, "arch": "gfx950:sramecc+:xnack-"}

    def stand_in_for_rccl_init(store):
        open(f"{out}.init_called.rank{rank}", "w").write("1")
        if len(set(bus[:world])) < world and claim == "nccl":
            raise RuntimeError("ncclInvalidUsage: Duplicate GPU detected (stand-in for what RCCL does with two ranks on one device)")
        if os.environ.get("FAIL_INIT") == "1":
            raise RuntimeError("ncclSystemError: stand-in for an RCCL initialisation failure")
        dist.init_process_group(backend="gloo", store=store, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=120))

    if os.environ.get("LEGACY_GROUP_CHECK") == "1":   # the pre-round-6 order: communicator first, then the check over it
        try:
            stand_in_for_rccl_init(collective.rendezvous_store(rank, world, 120))
        except RuntimeError as e:
            json.dump({"legacy_error": str(e)}, open(f"{out}.rank{rank}", "w"))
            return 1
    coll, rc = collective.open_group(rank, rank, rank, claim, world, world, pinned=pinned, label="worker", timeout_s=120,
                                     device_info=info, device_count=int(os.environ.get("FAKE_DEVCOUNT", str(world))),
                                     init_fn=stand_in_for_rccl_init, cpu_tensors=True)
    json.dump({"collective": coll, "problems": coll["problems"], "rc": rc}, open(f"{out}.rank{rank}", "w"))
    if rc == 0:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
