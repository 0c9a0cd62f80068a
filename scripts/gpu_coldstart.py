"""Cold-start cost of the scorer in a fresh process (what a one-image CLI run pays once):
library load + HIP init, context creation, first reference upload, first and second score."""
import os, sys, time
t0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("OAVIF_AMD_NO_TORCH", "1")
import ctypes
lib_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oavif_amd", "lib", "liboavif_hip.so")
L = ctypes.CDLL(lib_path)
PREFETCH = "--prefetch" in sys.argv
if PREFETCH:
    L.ssimu2_prefetch(0)   # returns at once; the frame synthesis below stands for load + first encode
t1 = time.perf_counter()
W, H = 3840, 2160
rng = np.random.default_rng(0)
ref = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
dst = np.clip(ref.astype(np.int16) + rng.integers(-6, 7, ref.shape), 0, 255).astype(np.uint8)
t2 = time.perf_counter()
u8p = ctypes.POINTER(ctypes.c_uint8)
ctx = ctypes.c_void_p()
rc = L.ssimu2_ctx_create(0, None, ctypes.byref(ctx)); assert rc == 0, rc
t3 = time.perf_counter()
rc = L.ssimu2_set_reference(ctx, ref.ctypes.data_as(u8p), W, H); assert rc == 0, rc
t4 = time.perf_counter()
out = ctypes.c_double()
rc = L.ssimu2_score_against_reference(ctx, dst.ctypes.data_as(u8p), ctypes.byref(out)); assert rc == 0, rc
t5 = time.perf_counter()
rc = L.ssimu2_score_against_reference(ctx, dst.ctypes.data_as(u8p), ctypes.byref(out)); assert rc == 0, rc
t6 = time.perf_counter()
ctx2 = ctypes.c_void_p()
t7 = time.perf_counter()
rc = L.ssimu2_ctx_create(0, None, ctypes.byref(ctx2)); assert rc == 0, rc
t8 = time.perf_counter()
print(f"[second ctx_create in the same process {1e3*(t8-t7):.1f} ms]", end=" ")
print("prefetch" if PREFETCH else "plain   ", end=" ")
print(f"frames ready after {1e3*(t2-t1):.0f} ms of CPU work |", end=" ")
print(f"dlopen {1e3*(t1-t0):.1f} ms | ctx_create (HIP init, stream, constants) {1e3*(t3-t2):.1f} ms | "
      f"set_reference 4K (alloc + upload + kernels load) {1e3*(t4-t3):.1f} ms | first score {1e3*(t5-t4):.2f} ms | "
      f"second score {1e3*(t6-t5):.2f} ms | score {out.value:.4f}")
