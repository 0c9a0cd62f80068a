"""A/B of library builds on the GPU box: same-box kernel timings plus the exact bits of a set
of scores, so that a faster build can be shown to return the same doubles.

    python scripts/gpu_ab.py libA.so libB.so ...        (driver: one child process per library)
    OAVIF_AMD_INSTR_LIB=lib.so python scripts/gpu_ab.py --one (child)

Every library must carry the timing hooks (include/ssimu2_hip_internal.h): an instrumented
build (oavif_amd/lib/liboavif_hip_instr.so, or `hipcc ... ssimu2_instrument.hip tq.cpp`), or a
round-1 build, whose product library still had them.

Each child prints `bits <case> <hex of score> <sha of the 108 averages>` lines and timing lines;
the driver diffs the bits lines against the first library's.
"""
import hashlib
import os
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [(64, 40, "noise", 2), (127, 129, "blockq", 3), (513, 259, "blur", 2), (640, 360, "band", 4),
         (1920, 1080, "blockq", 1), (3840, 2160, "blockq", 2), (1000, 2000, "noise", 1), (120, 300, "noise", 4),
         (121, 9, "blockq", 2), (8, 8, "noise", 3)]


def child():
    import torch
    import oavif_amd
    from oavif_amd import synth
    s = oavif_amd.Ssimu2(0, instrumented=True)
    print("version", s._L.ssimu2_version().decode(), flush=True)
    for (w, h, kind, strength) in CASES:
        ref = synth.make_ref(w, h, w * 7 + h)
        dst = synth.distort(ref, kind, strength, seed=3)
        sc = s.compute_ssimu2(ref, dst)
        avg, ns = s.last_averages()
        s.set_reference(ref)
        sc2 = s.score_against_reference(dst)
        print("bits", f"{w}x{h}-{kind}{strength}", struct.pack("<d", sc).hex(),
              hashlib.sha1(avg.tobytes()).hexdigest()[:16], struct.pack("<d", sc2).hex(), f"{sc:.6f}", flush=True)
    w, h = 3840, 2160
    ref = synth.make_ref(w, h, 0)
    dst = synth.distort(ref, "blockq", 2)
    tr = torch.from_numpy(ref).cuda().contiguous()
    td = torch.from_numpy(dst).cuda().contiguous()
    torch.cuda.synchronize()
    for _ in range(300):   # clocks
        s.enqueue_device(tr.data_ptr(), td.data_ptr(), w, h)
    s.wait()
    for rep in range(3):
        ks = [s.time_stage(tr.data_ptr(), td.data_ptr(), w, h, st, 50) * 1e3 for st in range(3)]
        ms, _ = s.time_device(tr.data_ptr(), td.data_ptr(), w, h, 50)
        s.set_reference_device(tr.data_ptr(), w, h)
        for _ in range(5):
            s.enqueue_against_reference_device(td.data_ptr())
        s.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            s.enqueue_against_reference_device(td.data_ptr())
        s.wait()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        # one stream, whole pair scores back to back, wall clock (what bench.py's `ms_per_score_one_stream` is)
        for _ in range(20):
            s.enqueue_device(tr.data_ptr(), td.data_ptr(), w, h)
        s.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            s.enqueue_device(tr.data_ptr(), td.data_ptr(), w, h)
        s.wait()
        torch.cuda.synchronize()
        dp = (time.perf_counter() - t0) / 200
        # a blocking score (enqueue + wait each time): the latency one probe of a search sees
        s.set_reference_device(tr.data_ptr(), w, h)   # the pair scores above dropped the cached reference
        s.enqueue_against_reference_device(td.data_ptr())
        s.wait()
        t0 = time.perf_counter()
        for _ in range(100):
            s.enqueue_against_reference_device(td.data_ptr())
            s.wait()
        db = (time.perf_counter() - t0) / 100
        print(f"time rep{rep} stage_us[pyramid,march,finalize]={[round(k, 1) for k in ks]} "
              f"whole_score_us={ms / 50 * 1e3:.1f} cached_pass_us={dt * 1e6:.1f} pair_score_one_stream_us={dp * 1e6:.1f} "
              f"blocking_cached_pass_us={db * 1e6:.1f}", flush=True)
    # the search path's default mode: the published recursion, cached reference (same 4K pair)
    from oavif_amd import _lib
    if hasattr(s._L, "ssimu2_ctx_set_blur"):
        s.set_blur(_lib.BLUR_RECURSIVE)
        rs = s.score_device(tr.data_ptr(), td.data_ptr(), w, h)
        avg, _ns = s.last_averages()
        s.set_reference_device(tr.data_ptr(), w, h)
        s.enqueue_against_reference_device(td.data_ptr())
        rs2 = s.wait()
        print("bits", "4k-recursive", struct.pack("<d", rs).hex(), hashlib.sha1(avg.tobytes()).hexdigest()[:16],
              struct.pack("<d", rs2).hex(), f"{rs:.6f}", flush=True)
        for rep in range(3):
            for _ in range(10):
                s.enqueue_against_reference_device(td.data_ptr())
            s.wait()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                s.enqueue_against_reference_device(td.data_ptr())
            s.wait()
            torch.cuda.synchronize()
            dr = (time.perf_counter() - t0) / 100
            t0 = time.perf_counter()
            for _ in range(50):
                s.enqueue_against_reference_device(td.data_ptr())
                s.wait()
            drb = (time.perf_counter() - t0) / 50
            print(f"time rep{rep} recursive cached_pass_us={dr * 1e6:.1f} blocking_cached_pass_us={drb * 1e6:.1f}", flush=True)
        s.set_blur(_lib.BLUR_FIR)
    # 1080p too (config[3]'s frame size)
    w, h = 1920, 1080
    ref = synth.make_ref(w, h, 1)
    dst = synth.distort(ref, "blockq", 2)
    tr = torch.from_numpy(ref).cuda().contiguous()
    td = torch.from_numpy(dst).cuda().contiguous()
    torch.cuda.synchronize()
    ks = [s.time_stage(tr.data_ptr(), td.data_ptr(), w, h, st, 50) * 1e3 for st in range(3)]
    print(f"time 1080p stage_us={[round(k, 1) for k in ks]}", flush=True)
    s.close()


def main():
    if "--one" in sys.argv:
        child()
        return 0
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    base = None
    rc = 0
    for lib in libs:
        env = dict(os.environ, OAVIF_AMD_INSTR_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env,
                           capture_output=True, text=True, timeout=600)
        print(f"==== {lib} rc={p.returncode}")
        out = p.stdout
        print(out)
        if p.returncode != 0:
            print(p.stderr[-2000:])
            rc = 1
            continue
        bits = [ln.split()[1:5] for ln in out.splitlines() if ln.startswith("bits")]
        if base is None:
            base = bits
        else:
            diff = [(a, b) for a, b in zip(base, bits) if a != b]
            print(f"bit-identical to {libs[0]}: {not diff}")
            for a, b in diff:
                print("   DIFF", a, b)
            if diff:
                rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
