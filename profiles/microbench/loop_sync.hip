// Microbenchmark (round 4): is ONE persistent round-loop kernel cheaper than two kernel boundaries per
// iteration on MI355X?  A miniature of the round loop's communication pattern, run three ways:
//   (a) two kernels per iteration, replayed from a hipGraph                         [what the library does]
//   (b) one persistent kernel, two grid barriers per iteration, every cross-workgroup word exchanged with
//       write-through (sc1) stores and L1-bypassing (sc1) loads: no fences
//   (c) the barriers alone (no payload), several designs: flat counter, per-XCD counters -> master -> per-XCD
//       release flags (last arriver releases), the same keyed by the real XCC_ID
// Miniature of one iteration (256 members, K = 28):
//   band:   E events' rows (1 KB each, from a table much larger than L2) -> ballot vs thresholds -> 32-byte mask
//   tally:  NC candidates: one row each + 256 gathered masks (8 KB from the band's table) -> a verdict per member
//           (atomicMin); every workgroup then reads the 256 verdicts (the replicated resolve step)
// Every spin is bounded: a barrier that does not complete sets `abort` and every block leaves.
//   build: hipcc --offload-arch=gfx950 -O3 loop_sync.hip -o loop_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

struct Sync {
    unsigned* cnt;     // [8][32] per-XCD arrival counters (one 128-byte line each)
    unsigned* master;  // [32]
    unsigned* flag;    // [8][32] per-XCD release words
    unsigned* abort_;  // [1]
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

__device__ __forceinline__ bool spin_until(unsigned* p, unsigned target, unsigned* abort_) {
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(p, RLX, AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (((++spins) & 1023u) == 0) {
            if (__hip_atomic_load(abort_, RLX, AGENT)) return false;
            if (spins > (1u << 21)) { __hip_atomic_store(abort_, 1u, RLX, AGENT); return false; }
        }
    }
    return true;
}

// MODE 0: flat counter.  MODE 1: per-XCD counters keyed by blockIdx % 8.  MODE 2: keyed by XCC_ID (per-XCD
// populations counted in a prologue).  `epoch` counts barriers from 1.  Called by every thread.
template <int MODE>
__device__ __forceinline__ bool grid_barrier(const Sync& S, unsigned epoch, unsigned nblocks, unsigned xcd, unsigned xcd_pop) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through stores have left
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (MODE == 0) {
            __hip_atomic_fetch_add(S.master, 1u, RLX, AGENT);
            ok = spin_until(S.master, epoch * nblocks, S.abort_);
        } else {
            const unsigned old = __hip_atomic_fetch_add(S.cnt + xcd * 32, 1u, RLX, AGENT);
            if (old + 1 == epoch * xcd_pop) {
                const unsigned m = __hip_atomic_fetch_add(S.master, 1u, RLX, AGENT);
                if (m + 1 == epoch * 8u) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) __hip_atomic_store(S.flag + x * 32, epoch, RLX, AGENT);
                }
            }
            ok = spin_until(S.flag + xcd * 32, epoch, S.abort_);
        }
    }
    __syncthreads();
    return ok;
}

// ---- the miniature ---------------------------------------------------------------------------------
struct Mini {
    const int* rows;     // [NROWS][256] row table (>> L2)
    unsigned* thr;       // [256] thresholds of the iteration (written by block 0, read by all)
    unsigned* masks;     // [E][8] 32-bit mask words (written by the band, gathered by the tally)
    u64* found;          // [256] verdict per member (atomicMin)
    int nrows, E, NC;
    unsigned* errs;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, RLX, AGENT); }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, RLX, AGENT); }

// band share of this workgroup: events e = wave, wave + nwaves, ...  COH: cross-workgroup words by sc1
template <bool COH>
__device__ __forceinline__ void band_phase(const Mini& M, int it, int wave, int nwaves, int lane) {
    unsigned t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = COH ? ld_sc1(M.thr + j * 64 + lane) : M.thr[j * 64 + lane];
    const int rot = (it * 7919) % (M.nrows - M.E);
    for (int e = wave; e < M.E; e += nwaves) {
        const int* row = M.rows + (size_t)(rot + e) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned v = (unsigned)row[j * 64 + lane];
            const u64 b = __ballot(v >= t[j]);
            // the mask word also carries the iteration, so the tally can check freshness
            if (lane == 0) {
                const unsigned w0 = ((unsigned)b & 0xffff0000u) | (unsigned)(it & 0xffff), w1 = ((unsigned)(b >> 32) & 0xffff0000u) | (unsigned)(it & 0xffff);
                if (COH) { st_sc1(M.masks + (size_t)e * 8 + 2 * j, w0); st_sc1(M.masks + (size_t)e * 8 + 2 * j + 1, w1); }
                else { M.masks[(size_t)e * 8 + 2 * j] = w0; M.masks[(size_t)e * 8 + 2 * j + 1] = w1; }
            }
        }
    }
}

template <bool COH>
__device__ __forceinline__ void tally_phase(const Mini& M, int it, int wave, int nwaves, int lane) {
    for (int cnd = wave; cnd < M.NC; cnd += nwaves) {
        const int rot = (it * 104729 + cnd * 131) % (M.nrows - 1);
        const int* row = M.rows + (size_t)rot * 256;
        unsigned acc = 0, stale = 0;
        int hop[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) hop[j] = (unsigned)(row[j * 64 + lane] * 2654435761u + cnd) % (unsigned)M.E;
        // lane (g, w): word w of the masks of hops g, g + 8, ... (the library's bit-sliced gather pattern)
        const int w = lane & 7, g = lane >> 3;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int h = g + 8 * i;               // hop index 0..255
            const int k = __shfl(hop[h >> 6], h & 63);
            const unsigned x = COH ? ld_sc1(M.masks + (size_t)k * 8 + w) : M.masks[(size_t)k * 8 + w];
            acc += x >> 16;
            stale |= (x & 0xffffu) ^ (unsigned)(it & 0xffff);
        }
        for (int off = 32; off; off >>= 1) acc += (unsigned)__shfl_xor((int)acc, off);
        if (__any(stale != 0) && lane == 0) atomicAdd(M.errs, 1u);
        if (lane == 0) atomicMin(reinterpret_cast<unsigned long long*>(M.found + (size_t)(it & 1) * 256 + (cnd & 255)), ((u64)(unsigned)it << 32) | (acc & 0xffffffu));
    }
}

// replicated resolve: every workgroup reads the verdicts; block 0 publishes the next thresholds
template <bool COH>
__device__ __forceinline__ void resolve_phase(const Mini& M, int it, int lane_in_block) {
    __shared__ unsigned s_x;
    if (lane_in_block < 256) {
        const u64* fb = M.found + (size_t)((it + 1) & 1) * 256;   // written by the tally of iteration it - 1
        const u64 f = COH ? __hip_atomic_load(fb + lane_in_block, RLX, AGENT) : fb[lane_in_block];
        // verdicts of the previous tally carry iteration it - 1 (or ~0 before the first)
        if (it > 0 && (unsigned)(f >> 32) != (unsigned)(it - 1) && f != ~0ull) atomicAdd(M.errs, 1u << 12);
        if (lane_in_block == 0) s_x = (unsigned)f;
    }
    __syncthreads();
    if (blockIdx.x == 0 && lane_in_block < 256) {
        const unsigned v = 1000u + (unsigned)it * 3u + (s_x & 1u);
        if (COH) st_sc1(M.thr + lane_in_block, v); else M.thr[lane_in_block] = v;
    }
}

template <int MODE>
__global__ void __launch_bounds__(1024) k_persistent(Sync S, Mini M, int iters, int payload, unsigned* pop) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int wave = blockIdx.x * wpb + wib, nwaves = gridDim.x * wpb;
    unsigned xcd = blockIdx.x & 7, xcd_pop = (gridDim.x + 7 - xcd) / 8;
    unsigned epoch = 0;
    if (MODE == 2) {   // census: how many blocks sit on each XCD (a flat barrier closes it)
        xcd = xcc_id();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(pop + xcd, 1u, RLX, AGENT);
        Sync C = S; C.master = pop + 32;   // the census barrier has a counter of its own
        if (!grid_barrier<0>(C, 1, gridDim.x, 0, 0)) return;
        xcd_pop = __hip_atomic_load(pop + xcd, RLX, AGENT);
    }
    for (int it = 0; it < iters; ++it) {
        if (payload) {
            resolve_phase<true>(M, it, threadIdx.x);
            // the verdict buffer of this iteration's tally is re-armed by block 0 (its last readers passed two barriers ago)
            if (blockIdx.x == 0 && threadIdx.x < 256) __hip_atomic_store(M.found + (size_t)(it & 1) * 256 + threadIdx.x, ~0ull, RLX, AGENT);
            if (!grid_barrier<MODE>(S, ++epoch, gridDim.x, xcd, xcd_pop)) return;   // thresholds visible (a real loop folds this into the fan-in/broadcast)
            band_phase<true>(M, it, wave, nwaves, lane);
        }
        if (!grid_barrier<MODE>(S, ++epoch, gridDim.x, xcd, xcd_pop)) return;
        if (payload) tally_phase<true>(M, it, wave, nwaves, lane);
        if (!grid_barrier<MODE>(S, ++epoch, gridDim.x, xcd, xcd_pop)) return;
    }
}

__global__ void __launch_bounds__(1024) k_band(Mini M, int it) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    resolve_phase<false>(M, it, threadIdx.x);
    __syncthreads();
    // (the library recomputes the thresholds in every block; here block 0's store is read by the others only in
    // the next launch, so the miniature uses the thresholds of the previous iteration: same traffic)
    if (blockIdx.x == 0 && threadIdx.x < 256) M.found[(size_t)(it & 1) * 256 + threadIdx.x] = ~0ull;
    band_phase<false>(M, it, blockIdx.x * wpb + wib, gridDim.x * wpb, lane);
}
__global__ void __launch_bounds__(256) k_tally(Mini M, int it) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    tally_phase<false>(M, it, blockIdx.x * wpb + wib, gridDim.x * wpb, lane);
}

// a streaming kernel for the "loaded chip" runs (what the can_see sweep is to the loop): low-priority stream
__global__ void k_stream(const int4* src, int4* dst, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            int4 v = src[i]; v.x += r; dst[i] = v;
        }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    Sync S; Mini M;
    unsigned* sync_mem; CK(hipMalloc(&sync_mem, 4096 * 4));
    S.cnt = sync_mem; S.master = sync_mem + 8 * 32; S.flag = sync_mem + 9 * 32; S.abort_ = sync_mem + 17 * 32;
    unsigned* pop = sync_mem + 18 * 32;
    M.nrows = 1 << 20; M.E = 12288; M.NC = 7168;
    int* rows; CK(hipMalloc(&rows, (size_t)M.nrows * 256 * 4));
    { std::vector<int> h((size_t)M.nrows * 256); unsigned s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (int)(s >> 8) % 4000; } CK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    M.rows = rows;
    CK(hipMalloc(&M.thr, 1024)); CK(hipMalloc(&M.masks, (size_t)M.E * 32)); CK(hipMalloc(&M.found, 4096)); CK(hipMalloc(&M.errs, 4));
    int4 *sa, *sb; const size_t sn = (size_t)64 << 20; CK(hipMalloc(&sa, sn * 16)); CK(hipMalloc(&sb, sn * 16)); CK(hipMemset(sa, 1, sn * 16));
    int lo_p, hi_p; CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    hipStream_t s, sl; CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi_p)); CK(hipStreamCreateWithPriority(&sl, hipStreamNonBlocking, lo_p));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset = [&]() {
        CK(hipMemsetAsync(sync_mem, 0, 4096 * 4, s)); CK(hipMemsetAsync(M.errs, 0, 4, s)); CK(hipMemsetAsync(M.found, 0xff, 4096, s));
        CK(hipMemsetAsync(M.thr, 0, 1024, s)); CK(hipMemsetAsync(M.masks, 0, (size_t)M.E * 32, s));
    };
    auto report = [&](const char* what, int nb, int bt, int per_it_barriers, bool loaded) {
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[2]; CK(hipMemcpy(&h[0], M.errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h[1], S.abort_, 4, hipMemcpyDeviceToHost));
        printf("%-34s %4d blocks x %4d thr %s: %7.2f us per iteration", what, nb, bt, loaded ? "(loaded)" : "(idle)  ", ms * 1e3 / iters);
        if (per_it_barriers) printf("  (%.2f us per barrier)", ms * 1e3 / iters / per_it_barriers);
        printf("  errs %u%s\n", h[0], h[1] ? "  ** BARRIER TIMED OUT **" : "");
        fflush(stdout);
    };
    for (int loaded = 0; loaded < 2; ++loaded) {
        auto load_on = [&]() { if (loaded) hipLaunchKernelGGL(k_stream, dim3(256), dim3(320), 0, sl, (const int4*)sa, sb, sn, 40); };
        auto load_off = [&]() { if (loaded) CK(hipStreamSynchronize(sl)); };
        struct Cfg { int nb, bt; } cfgs[] = {{256, 256}, {256, 1024}, {512, 256}, {512, 512}, {512, 768}, {1024, 256}};
        // (c) barriers alone
        for (auto cf : cfgs) {
            for (int mode = 0; mode < 3; ++mode) {
                reset(); load_on();
                CK(hipEventRecord(e0, s));
                if (mode == 0) hipLaunchKernelGGL(k_persistent<0>, dim3(cf.nb), dim3(cf.bt), 0, s, S, M, iters, 0, pop);
                if (mode == 1) hipLaunchKernelGGL(k_persistent<1>, dim3(cf.nb), dim3(cf.bt), 0, s, S, M, iters, 0, pop);
                if (mode == 2) hipLaunchKernelGGL(k_persistent<2>, dim3(cf.nb), dim3(cf.bt), 0, s, S, M, iters, 0, pop);
                CK(hipEventRecord(e1, s));
                report(mode == 0 ? "barrier only, flat counter" : mode == 1 ? "barrier only, per-XCD (b % 8)" : "barrier only, per-XCD (XCC_ID)", cf.nb, cf.bt, 2, loaded);
                load_off();
            }
        }
        // (b) persistent miniature
        for (auto cf : cfgs) {
            if (cf.nb * cf.bt < 256 * 512) continue;
            for (int mode = 1; mode < 3; ++mode) {
                reset(); load_on();
                CK(hipEventRecord(e0, s));
                if (mode == 1) hipLaunchKernelGGL(k_persistent<1>, dim3(cf.nb), dim3(cf.bt), 0, s, S, M, iters, 1, pop);
                if (mode == 2) hipLaunchKernelGGL(k_persistent<2>, dim3(cf.nb), dim3(cf.bt), 0, s, S, M, iters, 1, pop);
                CK(hipEventRecord(e1, s));
                report(mode == 1 ? "persistent miniature (b % 8)" : "persistent miniature (XCC_ID)", cf.nb, cf.bt, 0, loaded);
                load_off();
            }
        }
        // (a) two kernels per iteration from a graph (512 x 256 band blocks, 1792 x 256 tally blocks: the library's shapes)
        {
            reset();
            hipGraph_t g; hipGraphExec_t ge;
            const int per = 40;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int it = 0; it < per; ++it) {
                hipLaunchKernelGGL(k_band, dim3(512), dim3(256), 0, s, M, it);
                hipLaunchKernelGGL(k_tally, dim3(1792), dim3(256), 0, s, M, it);
            }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            reset(); load_on();
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < iters / per; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            report("two kernels per iteration (graph)", 512, 256, 0, loaded);
            load_off();
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
