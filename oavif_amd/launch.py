"""One command, N ranks: the self-launch of `bench.py --gpus N` and `python -m oavif_amd.batch --gpus N ...`.

The reference's batch entry is ONE command (scripts/measure.py:110-158: a sequential loop, no launcher), and the
command shape the driver uses for one GPU (`python3 bench.py --gpus 1 ...`) must also work for N > 1.  When an entry
point is asked for N > 1 ranks and no launcher has announced a world (no WORLD_SIZE / RANK in the environment), it
calls `spawn_ranks` FIRST -- before torch is imported, before any HIP call: the parent never touches a GPU -- and
becomes a plain supervisor of N fresh children, one per GPU:

  * each child is the same command with RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT (a free port) in its environment (the rest of the parent's environment is handed on unchanged), which
    is exactly what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` sets:
    the entry points run the same code under either launcher;
  * rank 0's stdout is relayed line by line as the parent's own stdout (its JSON line is the parent's last stdout
    line); the other ranks' stdout goes to the parent's stderr with a `[rank k]` prefix; every child's stderr is the
    parent's stderr;
  * the parent exits with the FIRST non-zero child code (a placement refusal, rc 4, stays 4; a child killed by
    signal S counts as 128 + S), 0 when every rank returned 0.  A failed child is reported, never retried; after the
    first failure the remaining ranks get `grace` seconds to leave by themselves (a refusal ends every rank with the
    same code) and are then terminated by PID -- a rank that lost its peer would otherwise sit in a rendezvous;
  * nothing is ever exec'ed over the parent (no `os.exec*`): the children are ordinary child processes.

Pure standard library: importing this module must not import torch (tests/test_launch.py checks it).
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence

LAUNCHED_ENV = "OAVIF_LAUNCHED_BY"   # pid of the supervising parent, in every child's environment


def launcher_announced() -> bool:
    """Has a launcher (torch.distributed.run, or this module) already described a world to this process?"""
    return "WORLD_SIZE" in os.environ or "RANK" in os.environ


def needs_self_launch(gpus: int) -> bool:
    return int(gpus) > 1 and not launcher_announced()


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environment(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """What one child of a single-node job of `world` ranks finds in its environment."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "GROUP_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), LAUNCHED_ENV: str(os.getpid())})
    return env


def exit_code_of(returncode: int) -> int:
    return 128 - returncode if returncode < 0 else returncode   # killed by signal S: returncode = -S


def _relay(src, dst, prefix: str = "") -> None:
    for line in iter(src.readline, ""):
        try:
            dst.write(prefix + line if prefix else line)
            dst.flush()
        except Exception:
            pass
    src.close()


def spawn_ranks(cmd: Sequence[str], world: int, grace: float = 30.0, timeout: Optional[float] = None,
                label: str = "oavif_amd.launch", env: Optional[Dict[str, str]] = None, cwd: Optional[str] = None) -> int:
    """Run `cmd` as `world` ranks of one node and supervise them (module docstring).  Returns the exit code the
    caller should leave with."""
    world = int(world)
    if world < 1:
        raise ValueError("world must be >= 1")
    if "torch" in sys.modules:   # not a correctness problem for the children; a rule of this repository's entry points
        print(f"{label}: note: torch was imported before the ranks were launched (the supervisor should hold no GPU state)",
              file=sys.stderr)
    port = free_port()
    print(f"{label}: launching {world} ranks (one per GPU), rendezvous 127.0.0.1:{port}: {' '.join(cmd)}", file=sys.stderr, flush=True)
    procs: List[subprocess.Popen] = []
    relays: List[threading.Thread] = []
    try:
        for rank in range(world):
            p = subprocess.Popen(list(cmd), env=rank_environment(rank, world, port, env), cwd=cwd, stdout=subprocess.PIPE,
                                 stderr=None, text=True, bufsize=1)
            procs.append(p)
            t = threading.Thread(target=_relay, args=(p.stdout, sys.stdout if rank == 0 else sys.stderr,
                                                      "" if rank == 0 else f"[rank {rank}] "), daemon=True)
            t.start()
            relays.append(t)
    except Exception as e:   # a rank that could not be started: the ones already running have no peer
        print(f"{label}: could not start rank {len(procs)}: {type(e).__name__}: {e}", file=sys.stderr)
        _stop(procs, label)
        return 1

    def on_signal(signum, _frame):
        print(f"{label}: signal {signum}: stopping the ranks", file=sys.stderr)
        _stop(procs, label)
        sys.exit(128 + signum)

    old = {}
    for s in (signal.SIGTERM, signal.SIGINT):
        try:
            old[s] = signal.signal(s, on_signal)
        except Exception:   # not the main thread
            pass
    first_bad, t_bad, t0 = None, None, time.monotonic()
    reported = set()
    try:
        while True:
            running = 0
            for rank, p in enumerate(procs):
                rc = p.poll()
                if rc is None:
                    running += 1
                elif rank not in reported:
                    reported.add(rank)
                    if rc != 0:
                        print(f"{label}: rank {rank} (pid {p.pid}) left with code {exit_code_of(rc)}", file=sys.stderr, flush=True)
                        if first_bad is None:
                            first_bad, t_bad = exit_code_of(rc), time.monotonic()
            if running == 0:
                break
            now = time.monotonic()
            if first_bad is not None and now - t_bad > grace:
                print(f"{label}: {running} rank(s) still running {grace:.0f} s after the first failure: terminating them",
                      file=sys.stderr, flush=True)
                _stop(procs, label)
                break
            if timeout is not None and now - t0 > timeout:
                print(f"{label}: the job ran longer than {timeout:.0f} s: terminating it", file=sys.stderr, flush=True)
                _stop(procs, label)
                first_bad = first_bad if first_bad is not None else 124
                break
            time.sleep(0.05)
    finally:
        for s, h in old.items():
            try:
                signal.signal(s, h)
            except Exception:
                pass
    for t in relays:
        t.join(timeout=10)
    try:
        sys.stdout.flush()
    except Exception:
        pass
    return 0 if first_bad is None else int(first_bad)


def _stop(procs: Sequence[subprocess.Popen], label: str) -> None:
    """Terminate exactly the children this supervisor started (by PID), then kill what ignores it."""
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        try:
            p.terminate()
        except Exception:
            pass
    t_end = time.monotonic() + 10.0
    for p in alive:
        try:
            p.wait(timeout=max(0.1, t_end - time.monotonic()))
        except Exception:
            try:
                p.kill()
                p.wait(timeout=5)
            except Exception:
                print(f"{label}: pid {p.pid} could not be stopped", file=sys.stderr)
