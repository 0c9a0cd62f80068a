"""The self-launch of `bench.py --gpus N` and `python -m oavif_amd.batch --gpus N` (oavif_amd/launch.py; VERDICT r05
item 1): the bare command -- the shape the driver uses for one GPU, and the shape of the reference's one-command batch
entry, scripts/measure.py:110-158 -- starts its own ranks when no launcher announced a world.  CPU only: the ranks are
a stub script, or the real entry points, which find no GPU here and leave with rc 3 on every rank."""
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent('''
    import json, os, sys, time
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                          "OAVIF_LAUNCHED_BY", "HSA_ENABLE_IPC_MODE_LEGACY", "KEPT_FROM_PARENT")}
    rec["argv"] = sys.argv[1:]
    open(os.path.join(sys.argv[1], f"rank{rank}.json"), "w").write(json.dumps(rec))
    print(f"stderr of rank {rank}", file=sys.stderr, flush=True)
    print(f"chatter of rank {rank}", flush=True)
    codes = [int(c) for c in os.environ.get("STUB_CODES", "").split(",") if c]
    code = codes[rank] if rank < len(codes) else 0
    if os.environ.get("STUB_HANG_RANK") == str(rank):
        time.sleep(600)
    if code < 0:
        os.kill(os.getpid(), -code)
    time.sleep(float(os.environ.get("STUB_SLEEP", "0.2")) * (world - rank))    # rank 0 leaves LAST: its line is still the last stdout line
    if rank == 0 and code == 0:
        print(json.dumps({"metric": "stub", "n_gpus": world}), flush=True)
    sys.exit(code)
''')

DRIVER = textwrap.dedent('''
    import sys
    sys.path.insert(0, {root!r})
    from oavif_amd import launch
    assert "torch" not in sys.modules                      # the supervisor holds no torch / GPU state
    rc = launch.spawn_ranks([sys.executable, {stub!r}, {out!r}, "--flag", "7"], {world}, grace={grace}, label="test")
    assert "torch" not in sys.modules
    sys.exit(rc)
''')


def _drive(tmp_path, world, grace=30.0, **env):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update({k: str(v) for k, v in env.items()})
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", DRIVER.format(root=ROOT, stub=str(stub), out=str(tmp_path), world=world, grace=grace)],
                       capture_output=True, text=True, timeout=120, env=e)
    return p, time.time() - t0


def test_ranks_get_the_launchers_environment_and_rank_0s_line_is_the_parents_last_line(tmp_path):
    p, _ = _drive(tmp_path, 3, KEPT_FROM_PARENT="yes", HSA_ENABLE_IPC_MODE_LEGACY="0")
    assert p.returncode == 0, p.stderr
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    for r, rec in enumerate(recs):
        assert (rec["RANK"], rec["LOCAL_RANK"], rec["WORLD_SIZE"], rec["LOCAL_WORLD_SIZE"]) == (str(r), str(r), "3", "3")
        assert rec["MASTER_ADDR"] == "127.0.0.1" and rec["MASTER_PORT"] == recs[0]["MASTER_PORT"] and int(rec["MASTER_PORT"]) > 1024
        assert rec["argv"][1:] == ["--flag", "7"] and rec["OAVIF_LAUNCHED_BY"].isdigit()
        # the parent's environment is handed on unchanged: what the pool exports (HSA_ENABLE_IPC_MODE_LEGACY) reaches every rank
        assert rec["KEPT_FROM_PARENT"] == "yes" and rec["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = p.stdout.splitlines()
    assert json.loads(out[-1]) == {"metric": "stub", "n_gpus": 3}           # rank 0's JSON line is the parent's last stdout line
    assert out[:-1] == ["chatter of rank 0"]                                 # other ranks' stdout does not reach the parent's stdout
    assert "[rank 1] chatter of rank 1" in p.stderr and "[rank 2] chatter of rank 2" in p.stderr
    assert all(f"stderr of rank {r}" in p.stderr for r in range(3))          # every rank's stderr is forwarded
    assert "launching 3 ranks" in p.stderr


def test_the_first_non_zero_code_is_the_parents_and_a_refusal_stays_4(tmp_path):
    p, _ = _drive(tmp_path, 2, STUB_CODES="4,4")
    assert p.returncode == 4 and "left with code 4" in p.stderr and '"metric"' not in p.stdout
    p, _ = _drive(tmp_path, 3, STUB_CODES="0,0,7")       # the rank that fails first decides (rank 2 leaves first in the stub)
    assert p.returncode == 7
    p, _ = _drive(tmp_path, 2, STUB_CODES="0,-9")        # a rank killed by SIGKILL counts as 128 + 9
    assert p.returncode == 137 and "left with code 137" in p.stderr


def test_ranks_that_outlive_a_failed_peer_are_terminated_not_waited_for_and_nothing_is_retried(tmp_path):
    p, wall = _drive(tmp_path, 2, grace=1.0, STUB_CODES="0,5", STUB_HANG_RANK="0")    # rank 0 would sit in its rendezvous for ever
    assert p.returncode == 5 and wall < 60
    assert "still running 1 s after the first failure: terminating them" in p.stderr
    assert p.stderr.count("launching 2 ranks") == 1                                    # reported, never retried


def test_bare_bench_and_batch_commands_launch_their_own_ranks(tmp_path):
    """The real entry points, bare (`python3 bench.py --gpus 2 ...`): the parent starts two ranks of itself before torch is
    imported; here they find no GPU and every rank leaves with rc 3 (the scorer has no CPU fallback), which becomes the
    parent's code.  Until round 5 this command printed "needs torch.distributed.run" and returned 2."""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    (tmp_path / "imgs").mkdir()
    for cmd, label in (([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], "bench.py"),
                       (["-m", "oavif_amd.batch", "--gpus", "2", str(tmp_path / "imgs"), str(tmp_path / "o.csv")], "oavif_amd.batch")):
        p = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=300, env=e, cwd=str(tmp_path))
        assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
        assert f"{label}: launching 2 ranks" in p.stderr and p.stderr.count("no GPU visible; the scorer has no CPU fallback") == 2
        assert "needs torch.distributed.run" not in p.stderr and '"value"' not in p.stdout
    # a launcher's world that disagrees with --gpus is said, not guessed around
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                       env=dict(e, RANK="0", WORLD_SIZE="4", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"))
    assert p.returncode == 2 and "the launcher announced WORLD_SIZE=4" in p.stderr


def test_the_launcher_module_needs_nothing_but_the_standard_library():
    p = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import oavif_amd.launch as l; "
                        "print('torch' in sys.modules, l.needs_self_launch(1), l.exit_code_of(-15))" % ROOT],
                       capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")})
    assert p.stdout.split() == ["False", "False", "143"], p.stderr
