"""A plain C program against the C ABI: compiles and links on CPU (no GPU needed), runs on the
MI355X and must report the score Python gets through ctypes for the same frames."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")


def _build(tmp_path, hip_lib):
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "oavif_amd", "lib")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), SRC,
           "-o", exe, "-L", libdir, "-loavif_hip", f"-Wl,-rpath,{libdir}",
           "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True)
    return exe


def test_c_program_compiles_and_links(tmp_path, hip_lib):
    exe = _build(tmp_path, hip_lib)
    assert os.path.exists(exe)
    import torch
    if not torch.cuda.is_available():
        # without a GPU the program must fail loudly with NO_DEVICE (exit 77), not compute
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 77 and "no usable HIP device" in r.stderr


def _xorshift_frames(w, h):
    xs = 2463534242

    def rnd():
        nonlocal xs
        xs ^= (xs << 13) & 0xFFFFFFFF
        xs ^= xs >> 17
        xs ^= (xs << 5) & 0xFFFFFFFF
        return xs
    n = w * h * 3
    ref = np.empty(n, np.uint8)
    for i in range(n):
        px = i // 3
        x, y = px % w, px // w
        ref[i] = (x * 3 + y * 2 + (i % 3) * 40 + (rnd() & 15)) & 255
    dist = np.empty(n, np.uint8)
    for i in range(n):
        v = int(ref[i]) + rnd() % 9 - 4
        dist[i] = min(max(v, 0), 255)
    return ref.reshape(h, w, 3), dist.reshape(h, w, 3)


@pytest.mark.gpu
def test_c_program_matches_python_binding(tmp_path, hip_lib, scorer):
    exe = _build(tmp_path, hip_lib)
    w, h = 96, 64
    r = subprocess.run([exe, str(w), str(h)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    m = re.match(r"score (\S+) (\S+) (\d) passes (\d+) q (\d+)", r.stdout)
    assert m, r.stdout
    ref, dist = _xorshift_frames(w, h)
    assert float(m.group(1)) == pytest.approx(scorer.compute_ssimu2(ref, dist), abs=1e-11)
    assert m.group(3) == "1"                      # set_reference path identical to the pair score
    assert 1 <= int(m.group(4)) <= 6 and 0 <= int(m.group(5)) <= 100
