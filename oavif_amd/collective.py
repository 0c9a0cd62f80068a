"""The `collective` record of a multi-rank run: what the process group really was.

A batch of independent images needs no data-path collective, so a rank that landed on the wrong
GPU -- two ranks on one device, a rank on a device of another NUMA node than its pinned cores -- still produces a
plausible throughput line.  This module makes the line self-describing and refuses the run when the placement is
wrong: every rank describes itself (rank, local rank, host, HIP device index, PCI bus id, NUMA node of the device,
the host cores it is pinned to, the HSA_* / HIP_* / ROCR_* / NCCL_* / RCCL_* variables it ran under), the descriptions
are exchanged through the rendezvous store (TCP) BEFORE any communicator exists -- the one placement the rules are
there to catch, two ranks on one GPU, is exactly the one RCCL answers with a hang or an obscure error at communicator
creation -- judged identically on every rank, and only then is the process group opened; one all_gather
over that group and backend (device tensors through RCCL when the backend is "nccl") confirms that the group carries
what the store carried.  `problems()` lists what is wrong:

  * backend "nccl" and two ranks of one host report the same PCI bus id (one GPU serving two ranks: RCCL itself
    refuses that at its first collective on most builds, but only as a hang or an obscure error);
  * fewer visible devices than ranks on the host (`torch.cuda.device_count() < local world`);
  * gathered world size / rank set differs from what the launcher announced.

Used by bench.py (`"collective"` on the JSON line) and by the batch driver (summary + `--collective-json`).
The counterpart in the reference is the sequential loop of scripts/measure.py:137-158, which has no ranks at all.
"""
from __future__ import annotations

import json
import os
import socket
from typing import Dict, List, Optional, Sequence

RECORD_BYTES = 4096   # fixed-size slot of one rank's JSON description in the gathered buffer
ENV_PREFIXES = ("HSA_", "HIP_", "ROCR_", "NCCL_", "RCCL_", "TORCH_NCCL_", "CUDA_VISIBLE", "GPU_DEVICE")
RC_REFUSED = 4        # the placement is wrong: every rank leaves with this code
RC_COLLECTIVE = 5     # the process group could not be opened / its first collective failed (RCCL's message on stderr)


def runtime_env() -> Dict[str, str]:
    """The runtime-steering variables this rank runs under (values cut to 96 characters): what the `collective`
    record shows of the environment, so that an override nobody remembers setting is visible on the line."""
    return {k: v[:96] for k, v in sorted(os.environ.items()) if k.startswith(ENV_PREFIXES)}


def library_versions() -> Dict[str, object]:
    """Versions of what carries the collective: torch, HIP, RCCL (torch.cuda.nccl.version() IS RCCL's on ROCm)."""
    out: Dict[str, object] = {}
    try:
        import torch
        out["torch"] = torch.__version__
        out["hip"] = getattr(torch.version, "hip", None)
        try:
            v = torch.cuda.nccl.version()
            out["rccl"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception as e:   # a CPU-only build has no nccl module
            out["rccl"] = f"unavailable ({type(e).__name__})"
    except Exception as e:
        out["torch"] = f"unavailable ({type(e).__name__})"
    return out


def rank_record(rank: int, local_rank: int, device_index: Optional[int], device_info: Optional[dict] = None,
                pinned: Optional[bool] = None) -> dict:
    """This rank as the record sees it.  `device_info`: ssimu2_query_device's dict for `device_index` (arch, PCI bus
    id, NUMA node); None = ask the library (needs a GPU).  `pinned`: whether the rank restricted itself to a core
    set (the cpus listed are the affinity mask either way)."""
    if device_info is None and device_index is not None:
        from . import scorer
        device_info = scorer.query_device(device_index)
    info = device_info or {}
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except Exception:
        cpus = []
    from . import hostinfo
    # what the placement code ASSUMED this device to be before HIP was up (hostinfo.visible_gpus: KFD topology order):
    # compared with what the runtime says in warnings()
    assumed = None
    try:
        vg = hostinfo.visible_gpus()
        if vg and device_index is not None and 0 <= device_index < len(vg):
            assumed = vg[device_index]
    except Exception:
        assumed = None
    return {"rank": int(rank), "local_rank": int(local_rank), "host": socket.gethostname(), "pid": os.getpid(),
            "assumed_pci_bus_id": assumed,
            "device": device_index, "pci_bus_id": info.get("pci_bus_id"), "numa_node": info.get("numa_node"),
            "arch": info.get("arch"), "cpus": hostinfo.format_cpus(cpus), "n_cpus": len(cpus),
            "cpu_numa_nodes": hostinfo.cpu_numa_nodes(cpus), "pinned": pinned, "env": runtime_env()}


def _encode(rec: dict) -> bytes:
    raw = json.dumps(rec, separators=(",", ":")).encode()
    if len(raw) > RECORD_BYTES - 1:   # a very fragmented cpu list: keep the record valid JSON
        rec = dict(rec, cpus=rec.get("cpus", "")[:200] + "...")
        raw = json.dumps(rec, separators=(",", ":")).encode()
    if len(raw) > RECORD_BYTES - 1 and rec.get("env"):   # a very long environment: names only
        rec = dict(rec, env={k: "..." for k in rec["env"]})
        raw = json.dumps(rec, separators=(",", ":")).encode()
    if len(raw) > RECORD_BYTES - 1:
        rec = dict(rec, env={"truncated": str(len(rec.get("env") or {}))})
        raw = json.dumps(rec, separators=(",", ":")).encode()[: RECORD_BYTES - 1]
    return raw + b"\0" * (RECORD_BYTES - len(raw))


def _decode(buf: bytes) -> dict:
    raw = bytes(buf).split(b"\0", 1)[0]
    try:
        return json.loads(raw.decode())
    except Exception:
        return {"undecodable": raw[:80].decode(errors="replace")}


def gather(rec: dict, device=None, group=None) -> List[dict]:
    """All ranks' records, in rank order, gathered with ONE all_gather of fixed-size byte tensors over the current
    process group (`device`: where the tensors live -- torch.device("cuda", i) sends them through RCCL, None / cpu
    through gloo).  Without an initialised process group: the caller's record alone."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [rec]
    world = dist.get_world_size(group)
    mine = torch.frombuffer(bytearray(_encode(rec)), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [_decode(t.cpu().numpy().tobytes()) for t in out]


def problems(records: Sequence[dict], backend: str, world: int, local_world: Optional[int] = None,
             device_count: Optional[int] = None) -> List[str]:
    """What is wrong with the placement the records describe ([] = nothing)."""
    bad: List[str] = []
    if len(records) != world:
        bad.append(f"gathered {len(records)} rank records, the launcher announced a world of {world}")
    ranks = sorted(r.get("rank", -1) for r in records)
    if ranks != list(range(len(records))):
        bad.append(f"rank set {ranks} is not 0..{len(records) - 1}")
    if any("undecodable" in r for r in records):
        bad.append("a rank's record did not survive the gather")
    if backend == "nccl":
        seen: Dict[tuple, int] = {}
        for r in records:
            if r.get("pci_bus_id") is None:
                bad.append(f"rank {r.get('rank')} reports no PCI bus id (no device?)")
                continue
            key = (r.get("host"), r.get("pci_bus_id"))
            if key in seen:
                bad.append(f"ranks {seen[key]} and {r.get('rank')} both sit on the GPU at {key[1]} of host {key[0]}: "
                           f"one process per GPU means one GPU per process")
            else:
                seen[key] = r.get("rank")
        if device_count is not None and local_world is not None and device_count < local_world:
            bad.append(f"{device_count} visible device(s) for {local_world} ranks on this host "
                       f"(torch.cuda.device_count() < local world)")
    return bad


def warnings(records: Sequence[dict]) -> List[str]:
    """Placements that are legal but not what the design intends (reported on the line, never a refusal): a rank that
    pinned itself to cores of another NUMA node than the one its GPU hangs off ("host cores near that GPU")."""
    out: List[str] = []
    for r in records:
        nodes, gpu = r.get("cpu_numa_nodes") or [], r.get("numa_node")
        if r.get("pinned") and gpu is not None and gpu >= 0 and nodes and gpu not in nodes:
            out.append(f"rank {r.get('rank')} is pinned to cpus {r.get('cpus')} of NUMA node(s) {nodes}, its GPU at "
                       f"{r.get('pci_bus_id')} hangs off node {gpu}")
        if r.get("assumed_pci_bus_id") and r.get("pci_bus_id") and r["assumed_pci_bus_id"] != r["pci_bus_id"]:
            out.append(f"rank {r.get('rank')}: the placement code took HIP device {r.get('device')} for the GPU at "
                       f"{r['assumed_pci_bus_id']} (KFD topology order), the runtime says it is {r['pci_bus_id']}")
    return out


def describe(backend: str, world: int, records: Sequence[dict], through: str, bad: Sequence[str] = ()) -> dict:
    """The `collective` object of the JSON line."""
    distinct = len({(r.get("host"), r.get("pci_bus_id")) for r in records if r.get("pci_bus_id")})
    return {"backend": backend, "world_size": int(world), "gathered_through": through,
            "distinct_devices": distinct, "ranks": list(records), "versions": library_versions(),
            "problems": list(bad), "warnings": warnings(records)}


def rendezvous_store(rank: int, world: int, timeout_s: float = 300.0):
    """The job's rendezvous store, opened the way `init_process_group("env://")` would (MASTER_ADDR / MASTER_PORT; under
    torch.distributed.run the agent's store, otherwise rank 0 hosts it) -- but WITHOUT creating a process group.  The
    same store is later handed to init_process_group(store=...)."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    store, _r, _w = next(iter(dist.rendezvous("env://", int(rank), int(world),
                                              timeout=datetime.timedelta(seconds=timeout_s))))
    store.set_timeout(datetime.timedelta(seconds=timeout_s))
    return store


def exchange(store, rec: dict, rank: int, world: int, tag: str = "checkin") -> List[dict]:
    """Every rank's record through the key-value store: no communicator, no device memory, nothing RCCL.  Rank 0 (which
    may host the store) does not return before every rank has read every record."""
    store.set(f"oavif/{tag}/{rank}", json.dumps(rec, separators=(",", ":")))
    recs = []
    for r in range(world):
        try:
            recs.append(json.loads(bytes(store.get(f"oavif/{tag}/{r}")).decode()))
        except Exception as e:
            recs.append({"undecodable": f"rank {r}: {type(e).__name__}: {str(e)[:80]}"})
    store.set(f"oavif/{tag}/read/{rank}", "1")
    if rank == 0:
        store.wait([f"oavif/{tag}/read/{r}" for r in range(world)])
    return recs


def check_in(rank: int, launcher_local_rank: int, device_index: Optional[int], backend: str, world: int,
             local_world: int, tensor_device=None, pinned: Optional[bool] = None, device_info: Optional[dict] = None,
             device_count: Optional[int] = None, grouped: bool = True, store=None):
    """Describe this rank, exchange every rank's description, judge the placement.  With `store` (what bench.py and
    the batch driver do): through the rendezvous store, before any communicator exists.  Without it: over the current
    process group (`tensor_device`: where the gathered tensors live; `grouped` False = no process group, one rank).
    Returns (collective record, problems): a non-empty problem list means every rank must leave with a non-zero
    code -- all ranks judge the same records, so all of them do.
    `device_info` / `device_count` default to what the library and torch report (tests inject them)."""
    me = rank_record(rank, launcher_local_rank, device_index, device_info=device_info, pinned=pinned)
    if store is not None:
        recs = exchange(store, me, rank, world)
    else:
        recs = gather(me, tensor_device) if grouped else [me]
    if device_count is None:
        import torch
        device_count = torch.cuda.device_count()
    eff_backend = backend if (grouped or store is not None) else "none"
    bad = problems(recs, eff_backend, world if (grouped or store is not None) else 1, local_world, device_count)
    if store is not None:
        through = "the rendezvous store (TCP key-value exchange), before any communicator existed"
    elif not grouped:
        through = "no process group (one rank)"
    elif tensor_device is not None and str(tensor_device).startswith("cuda"):
        through = f"the job's process group ({'RCCL' if backend == 'nccl' else backend}, device tensors)"
    else:
        through = f"the job's process group ({backend}, CPU tensors)"
    return describe(eff_backend, world, recs, through, bad), bad


_IDENTITY = ("rank", "local_rank", "host", "pid", "device", "pci_bus_id", "numa_node")


def confirm(coll: dict, rank: int, tensor_device, backend: str) -> List[str]:
    """After the process group is open: the same records once more, through ONE all_gather over that group and backend
    (device tensors through RCCL for "nccl").  They must describe the same processes on the same devices as the
    records the store carried; the `collective` object says that both paths were taken."""
    mine = next((r for r in coll["ranks"] if r.get("rank") == rank), None) or {}
    again = gather(mine, tensor_device)
    bad = []
    if len(again) != len(coll["ranks"]):
        bad.append(f"the process group gathered {len(again)} records, the store carried {len(coll['ranks'])}")
    for a, b in zip(again, coll["ranks"]):
        if any(a.get(k) != b.get(k) for k in _IDENTITY):
            bad.append(f"rank {b.get('rank')}: the record gathered over the process group differs from the one the store "
                       f"carried ({ {k: a.get(k) for k in _IDENTITY} } vs { {k: b.get(k) for k in _IDENTITY} })")
    dev = tensor_device is not None and str(tensor_device).startswith("cuda")
    coll["confirmed_through"] = (f"one all_gather over the job's process group ({'RCCL' if backend == 'nccl' else backend}, "
                                 f"{'device' if dev else 'CPU'} tensors)")
    coll["gathered_through"] += "; confirmed by " + coll["confirmed_through"]
    coll["problems"] = list(coll["problems"]) + bad
    return bad


def open_group(rank: int, launcher_local_rank: int, device_index: Optional[int], backend: str, world: int, local_world: int,
               pinned: Optional[bool] = None, label: str = "oavif_amd", timeout_s: float = 300.0,
               device_info: Optional[dict] = None, device_count: Optional[int] = None, init_fn=None, cpu_tensors: bool = False):
    """What bench.py and the batch driver do before their first collective.  (1) The rendezvous store; (2) check_in
    through it; a wrong placement returns (record, RC_REFUSED) on EVERY rank and no process group, no RCCL
    communicator, was ever created.  (3) init_process_group on that store (`init_fn(store)` replaces it in tests;
    backend "nccl": device_id = this rank's GPU, the communicator is created eagerly) and (4) `confirm`.  A failure of
    (3) or (4) is printed with the backend's own message on every rank and returns (record, RC_COLLECTIVE): nothing is
    retried in a process whose GPU is initialised, nothing is re-executed.  (record, 0) = the group is open.
    `cpu_tensors`: the confirming all_gather uses CPU tensors whatever the backend claims (CPU tests)."""
    import sys
    store = rendezvous_store(rank, world, timeout_s)
    coll, bad = check_in(rank, launcher_local_rank, device_index, backend, world, local_world, pinned=pinned,
                         device_info=device_info, device_count=device_count, store=store)
    if bad:
        if rank == 0:
            print(f"{label}: refusing to run:\n  " + "\n  ".join(bad), file=sys.stderr, flush=True)
        return coll, RC_REFUSED
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", device_index) if backend == "nccl" and device_index is not None and not cpu_tensors else None
    try:
        if init_fn is not None:
            init_fn(store)
        else:
            import datetime
            kw = {"device_id": dev} if dev is not None else {}
            dist.init_process_group(backend=backend, store=store, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=timeout_s), **kw)
        bad = confirm(coll, rank, dev, backend)
    except Exception as e:
        what = "RCCL" if backend == "nccl" else backend
        print(f"{label}: rank {rank}: the {what} process group could not be opened or its first all_gather failed: "
              f"{type(e).__name__}: {e}", file=sys.stderr, flush=True)
        coll["error"] = f"{type(e).__name__}: {str(e)[:400]}"
        return coll, RC_COLLECTIVE
    if bad:
        if rank == 0:
            print(f"{label}: refusing to run:\n  " + "\n  ".join(bad), file=sys.stderr, flush=True)
        try:
            dist.destroy_process_group()
        except Exception:
            pass
        return coll, RC_REFUSED
    return coll, 0


def preflight(backend: str, local_world: int) -> Optional[str]:
    """Before the process group exists and before set_device: are there enough devices for the ranks of this host?"""
    if backend != "nccl":
        return None
    import torch
    n = torch.cuda.device_count()
    if n < local_world:
        return (f"{n} visible device(s) for {local_world} ranks on this host over RCCL "
                f"(torch.cuda.device_count() < local world): one process per GPU needs a GPU per process")
    return None
