// What does ONE cross-workgroup hand-off cost on an MI355X?  (the tile-wavefront form of the
// recursive blur was killed on arithmetic -- its critical path has one such hand-off per tile; this is the measured
// term of that arithmetic.)
//
// A chain of G workgroups, one per CU (48 KB of LDS each keeps two from sharing a CU only loosely; G <= the CU count
// keeps all of them resident): workgroup i waits for the flag of workgroup i - 1, reads its payload (the recursion
// state a 64-column tile would hand on: P floats), writes its own payload, releases its own flag.  Nothing else runs,
// so (kernel time) / (G - 1) is the latency of one hop: release + visibility across CUs / XCDs + acquire + payload.
//   flag protocol: payload stores, __threadfence() (agent-scope release), workgroup barrier, lane 0 stores the flag
//   with an agent-scope atomic; the consumer's lane 0 polls it with agent-scope atomic loads (relaxed) and issues ONE
//   agent-scope acquire fence after the poll has matched, then a workgroup barrier, then everybody loads the payload.
//   Every wait is bounded (2^22 polls): a broken chain reports `failed`, it cannot hang the GPU.
// Hops between neighbouring workgroup indices alternate XCDs (workgroups are dealt round-robin over the 8 XCDs), so
// this is the cross-XCD figure a wavefront over the whole chip would pay.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k_chain(unsigned* flags, float* payload, int P, unsigned* failed, unsigned epoch) {
    __shared__ float s_pad[12 * 1024];  // 48 KB: at most three of these per CU; with G <= #CUs the dispatcher spreads them
    const int i = blockIdx.x, t = threadIdx.x;
    float acc = 0.f;
    if (i > 0) {
        if (t == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(&flags[(size_t)(i - 1) * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                if (++spins > (1u << 22)) {
                    atomicAdd(failed, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);  // agent scope by default in HIP device code
        }
        __syncthreads();
        for (int k = t; k < P; k += 256) acc += payload[(size_t)(i - 1) * P + k];
    }
    s_pad[t] = acc;
    for (int k = t; k < P; k += 256) payload[(size_t)i * P + k] = acc + (float)k;
    __threadfence();
    __syncthreads();
    if (t == 0) __hip_atomic_store(&flags[(size_t)i * 32], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (s_pad[t] == -1.f) payload[0] = 0.f;  // keeps the LDS allocation
}

int main(int argc, char** argv) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int G = argc > 1 ? atoi(argv[1]) : (cus > 0 ? cus : 256);
    unsigned *flags, *failed;
    float* payload;
    const int Pmax = 16384;
    hipMalloc(&flags, (size_t)G * 32 * sizeof(unsigned));
    hipMalloc(&failed, sizeof(unsigned));
    hipMalloc(&payload, (size_t)G * Pmax * sizeof(float));
    hipMemset(flags, 0, (size_t)G * 32 * sizeof(unsigned));
    hipMemset(failed, 0, sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned epoch = 0;
    printf("hand-off chain over %d workgroups (one per CU, %d CUs): us per hop, best of 5 launches\n", G, cus);
    const int Ps[] = {0, 64, 3072, 16384};  // payload floats: none, one cache line pair, the v-state of a 64-column tile x 3 planes, 64 KB
    for (int pi = 0; pi < 4; ++pi) {
        const int P = Ps[pi];
        double best = 1e30;
        for (int rep = 0; rep < 6; ++rep) {
            ++epoch;
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_chain, dim3(G), dim3(256), 0, 0, flags, payload, P, failed, epoch);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;  // the first launch loads the code object
        }
        unsigned f = 0;
        hipMemcpy(&f, failed, sizeof f, hipMemcpyDeviceToHost);
        printf("  payload %6d floats (%6.1f KB): kernel %8.1f us  ->  %.2f us per hop%s\n", P, P * 4 / 1024.0, best * 1e3,
               best * 1e3 / (G - 1), f ? "   (FAILED: a wait ran out)" : "");
    }
    return 0;
}
