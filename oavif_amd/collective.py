"""The `collective` record of a multi-rank run: what the process group really was.

A batch of independent images needs no data-path collective, so a rank that landed on the wrong
GPU -- two ranks on one device, a rank on a device of another NUMA node than its pinned cores -- still produces a
plausible throughput line.  This module makes the line self-describing and refuses the run when the placement is
wrong: every rank describes itself (rank, local rank, host, HIP device index, PCI bus id, NUMA node of the device,
the host cores it is pinned to), the descriptions are gathered over the SAME process group and backend the
measured job uses (device tensors through RCCL when the backend is "nccl"), and `problems()` lists what is wrong:

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

RECORD_BYTES = 1024   # fixed-size slot of one rank's JSON description in the gathered buffer


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
            "cpu_numa_nodes": hostinfo.cpu_numa_nodes(cpus), "pinned": pinned}


def _encode(rec: dict) -> bytes:
    raw = json.dumps(rec, separators=(",", ":")).encode()
    if len(raw) > RECORD_BYTES - 1:   # a very fragmented cpu list: keep the record valid JSON
        rec = dict(rec, cpus=rec.get("cpus", "")[:200] + "...")
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


def check_in(rank: int, launcher_local_rank: int, device_index: Optional[int], backend: str, world: int,
             local_world: int, tensor_device=None, pinned: Optional[bool] = None, device_info: Optional[dict] = None,
             device_count: Optional[int] = None, grouped: bool = True):
    """What bench.py and the batch driver do first with their process group: describe this rank, gather every rank's
    description over the group (`tensor_device`: where the gathered tensors live; `grouped` False = no process group,
    one rank), judge the placement.  Returns (collective record, problems): a non-empty problem list means every rank
    must leave with a non-zero code -- all ranks judge the same gathered records, so all of them do.
    `device_info` / `device_count` default to what the library and torch report (tests inject them)."""
    me = rank_record(rank, launcher_local_rank, device_index, device_info=device_info, pinned=pinned)
    recs = gather(me, tensor_device) if grouped else [me]
    if device_count is None:
        import torch
        device_count = torch.cuda.device_count()
    eff_backend = backend if grouped else "none"
    bad = problems(recs, eff_backend, world if grouped else 1, local_world, device_count)
    if not grouped:
        through = "no process group (one rank)"
    elif tensor_device is not None and str(tensor_device).startswith("cuda"):
        through = f"the job's process group ({'RCCL' if backend == 'nccl' else backend}, device tensors)"
    else:
        through = f"the job's process group ({backend}, CPU tensors)"
    return describe(eff_backend, world, recs, through, bad), bad


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
