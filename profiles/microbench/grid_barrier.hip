// Microbenchmark: what does one grid-wide barrier cost on MI355X next to one kernel boundary?
// (decides whether the round loop is worth a persistent kernel)
//   build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) { ok = false; break; }  // never hang the box
        }
    }
    __syncthreads();
    return ok;
}

// every iteration: write a value, barrier, read the neighbour block's value (must be this iteration's)
__global__ void k_persistent(int iters, unsigned* bar, int* data, int* errs) {
    const int nb = gridDim.x;
    int bad = 0;
    for (int it = 1; it <= iters; ++it) {
        if (threadIdx.x == 0) __hip_atomic_store(&data[blockIdx.x], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier(bar, (unsigned)nb * (2 * it - 1))) { bad = 1 << 20; break; }
        const int v = __hip_atomic_load(&data[(blockIdx.x + 1) % nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != it) ++bad;
        if (!grid_barrier(bar, (unsigned)nb * (2 * it))) { bad = 1 << 20; break; }
    }
    if (threadIdx.x == 0 && bad) atomicAdd(errs, bad);
}

// the same with plain (non-atomic) stores and loads of a 64 KB array per block pair + fences
__global__ void k_persistent_bulk(int iters, unsigned* bar, int* buf, int per, int* errs) {
    const int nb = gridDim.x;
    int bad = 0;
    for (int it = 1; it <= iters; ++it) {
        for (int i = threadIdx.x; i < per; i += blockDim.x) buf[(size_t)blockIdx.x * per + i] = it + i;
        __threadfence();
        if (!grid_barrier(bar, (unsigned)nb * (2 * it - 1))) { bad = 1 << 20; break; }
        const int nbk = (blockIdx.x + nb / 2 + 1) % nb;  // a block that most likely sits on another XCD
        for (int i = threadIdx.x; i < per; i += blockDim.x) {
            const int v = __builtin_nontemporal_load(&buf[(size_t)nbk * per + i]);
            if (v != it + i) ++bad;
        }
        if (!grid_barrier(bar, (unsigned)nb * (2 * it))) { bad = 1 << 20; break; }
    }
    if (bad) atomicAdd(errs, bad);
}

__global__ void k_step(int it, int* data, int* errs, int phase) {
    const int nb = gridDim.x;
    if (phase == 0) { if (threadIdx.x == 0) data[blockIdx.x] = it; }
    else { if (threadIdx.x == 0 && data[(blockIdx.x + 1) % nb] != it) atomicAdd(errs, 1); }
}

int main() {
    int* data; int* errs; unsigned* bar; int* buf;
    CK(hipMalloc(&data, 1 << 20)); CK(hipMalloc(&errs, 4)); CK(hipMalloc(&bar, 4));
    CK(hipMalloc(&buf, 64 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    struct Cfg { int nb, bt; } cfgs[] = {{256, 256}, {256, 1024}, {512, 256}, {512, 512}, {1024, 256}, {2048, 256}};
    for (auto cf : cfgs) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(bar, 0, 4, s)); CK(hipMemsetAsync(errs, 0, 4, s)); CK(hipMemsetAsync(data, 0, 1 << 20, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k_persistent, dim3(cf.nb), dim3(cf.bt), 0, s, iters, bar, data, errs);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); int h; CK(hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost));
            if (rep) printf("persistent  %4d blocks x %4d thr: %.2f us per barrier (2 per iteration), errs %d\n", cf.nb, cf.bt, ms * 1e3 / (2 * iters), h);
        }
        for (int per : {1024, 16384}) {
            if ((size_t)cf.nb * per * 4 > (64u << 20)) continue;
            CK(hipMemsetAsync(bar, 0, 4, s)); CK(hipMemsetAsync(errs, 0, 4, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k_persistent_bulk, dim3(cf.nb), dim3(cf.bt), 0, s, iters / 4, bar, buf, per, errs);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); int h; CK(hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost));
            printf("   bulk %6d ints/block + fence: %.2f us per iteration (2 barriers), errs %d\n", per, ms * 1e3 / (iters / 4), h);
        }
        // kernel boundaries: a graph of 2*200 tiny kernels, replayed
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipMemsetAsync(errs, 0, 4, s));
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int it = 1; it <= 200; ++it) {
            hipLaunchKernelGGL(k_step, dim3(cf.nb), dim3(cf.bt), 0, s, it, data, errs, 0);
            hipLaunchKernelGGL(k_step, dim3(cf.nb), dim3(cf.bt), 0, s, it, data, errs, 1);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); int h; CK(hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost));
        printf("graph       %4d blocks x %4d thr: %.2f us per kernel boundary, errs %d\n", cf.nb, cf.bt, ms * 1e3 / (5 * 400), h);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
