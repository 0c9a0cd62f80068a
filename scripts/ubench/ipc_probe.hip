// ipc_probe: does cross-process sharing of device memory work on this box under the HSA_ENABLE_IPC_MODE_LEGACY
// value this process was started with?  (bench.py / batch.py set that variable for multi-rank
// runs with no record of why.)  RCCL's intra-node transport between two ranks of one host opens the peer's buffers
// through exactly these two calls, so a failure here is the failure an N > 1 RCCL job would meet at its first
// collective -- and this probe needs ONE GPU, which is all this pool's boxes have.
//
//   exporter (parent): hipMalloc 1 MiB, fill with a pattern, hipIpcGetMemHandle, hand the 64-byte handle to the child
//   importer (child, forked BEFORE any HIP call in either process): hipIpcOpenMemHandle, read the pattern back
//
// Prints one line: the variable's value, each call's hipError_t name, and whether the pattern arrived.
// Exit code 0 whatever the outcome (it is a measurement); 2 = the probe itself could not run.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

__global__ void k_fill(unsigned* p, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0xA5000000u ^ i;
}

struct Reply {
    int open_err;
    int copy_err;
    int ok;
};

int main() {
    const char* mode = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
    int to_child[2], to_parent[2];
    if (pipe(to_child) || pipe(to_parent)) return 2;
    const pid_t pid = fork();  // before any HIP call: both processes initialise their own runtime
    if (pid < 0) return 2;
    const unsigned n = 1u << 18;
    if (pid == 0) {
        hipIpcMemHandle_t h;
        if (read(to_child[0], &h, sizeof h) != (ssize_t)sizeof h) _exit(2);
        Reply r{0, 0, 0};
        void* p = nullptr;
        r.open_err = (int)hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (r.open_err == 0) {
            unsigned* host = (unsigned*)malloc(n * sizeof(unsigned));
            r.copy_err = (int)hipMemcpy(host, p, n * sizeof(unsigned), hipMemcpyDeviceToHost);
            if (r.copy_err == 0) {
                r.ok = 1;
                for (unsigned i = 0; i < n; i += 4097) r.ok &= host[i] == (0xA5000000u ^ i);
            }
            (void)hipIpcCloseMemHandle(p);
            free(host);
        }
        if (write(to_parent[1], &r, sizeof r) != (ssize_t)sizeof r) _exit(2);
        _exit(0);
    }
    unsigned* d = nullptr;
    hipError_t e = hipMalloc(&d, n * sizeof(unsigned));
    if (e != hipSuccess) {
        printf("ipc_probe: hipMalloc failed: %s\n", hipGetErrorName(e));
        return 2;
    }
    hipLaunchKernelGGL(k_fill, dim3(n / 256), dim3(256), 0, 0, d, n);
    (void)hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof h);
    const hipError_t ge = hipIpcGetMemHandle(&h, d);
    Reply r{-1, -1, 0};
    if (ge == hipSuccess) {
        if (write(to_child[1], &h, sizeof h) != (ssize_t)sizeof h) return 2;
        if (read(to_parent[0], &r, sizeof r) != (ssize_t)sizeof r) r = Reply{-2, -2, 0};
    } else {
        close(to_child[1]);  // the child sees end-of-file and leaves
    }
    int st = 0;
    (void)waitpid(pid, &st, 0);
    printf("HSA_ENABLE_IPC_MODE_LEGACY=%s  hipIpcGetMemHandle=%s  hipIpcOpenMemHandle(other process)=%s  copy=%s  pattern_arrived=%s\n",
           mode ? mode : "(unset)", hipGetErrorName(ge), r.open_err >= 0 ? hipGetErrorName((hipError_t)r.open_err) : "not attempted",
           r.copy_err >= 0 ? hipGetErrorName((hipError_t)r.copy_err) : "not attempted", r.ok ? "yes" : "no");
    (void)hipFree(d);
    return 0;
}
