// Does a virtual address that was unmapped and mapped again to ANOTHER physical chunk translate to the
// new chunk, or can the old translation survive in the GPU's translation caches?  (DESIGN.md §10 N2:
// the windowed can_see table recycles physical chunks and, on a rewind, maps evicted addresses again.)
//
//   chunks A, B;  slot0 <- A;  kernel touches slot0 (translation cached);  unmap slot0;
//   slot1 <- A, slot0 <- B;  kernel writes pattern P to slot0, then slot1 is read back:
//   slot1 must still hold A's old content; if it holds P, the write to slot0 went through the
//   stale translation slot0 -> A.
//
// Modes: 0 = nothing between unmap and re-map, 1 = a 4 MB hipMalloc + hipFree in between (the
// driver's map/unmap calls for ordinary allocations flush the translation caches).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void fill(int* p, size_t n, int v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void count_eq(const int* p, size_t n, int v, unsigned long long* out) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n && p[i] == v) atomicAdd(out, 1ull); }

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int rounds = argc > 2 ? atoi(argv[2]) : 8;
    const size_t chunk_mb = argc > 3 ? atoi(argv[3]) : 2;
    CK(hipSetDevice(0));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t chunk = ((chunk_mb << 20) + gran - 1) / gran * gran;
    const size_t n = chunk / 4;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned long long* d_cnt;
    CK(hipMalloc(&d_cnt, 8));
    int stale_rounds = 0;
    for (int r = 0; r < rounds; ++r) {
        void* base = nullptr;
        CK(hipMemAddressReserve(&base, 64 * chunk, 0, nullptr, 0));
        char* s0 = (char*)base;
        char* s1 = (char*)base + 16 * chunk;
        hipMemGenericAllocationHandle_t A, B;
        CK(hipMemCreate(&A, chunk, &prop, 0));
        CK(hipMemCreate(&B, chunk, &prop, 0));
        CK(hipMemMap(s0, chunk, 0, A, 0));
        CK(hipMemSetAccess(s0, chunk, &acc, 1));
        fill<<<blocks, 256>>>((int*)s0, n, 111);          // A holds 111; slot0's translation is cached
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(s0, chunk));
        if (mode == 1) { void* t = nullptr; CK(hipMalloc(&t, 4 << 20)); CK(hipFree(t)); }
        CK(hipMemMap(s1, chunk, 0, A, 0));
        CK(hipMemSetAccess(s1, chunk, &acc, 1));
        CK(hipMemMap(s0, chunk, 0, B, 0));
        CK(hipMemSetAccess(s0, chunk, &acc, 1));
        fill<<<blocks, 256>>>((int*)s0, n, 222);          // must land in B
        CK(hipDeviceSynchronize());
        CK(hipMemset(d_cnt, 0, 8));
        count_eq<<<blocks, 256>>>((const int*)s1, n, 222, d_cnt);   // A through slot1: 222 here = stale translation
        unsigned long long bad = 0;
        CK(hipMemcpy(&bad, d_cnt, 8, hipMemcpyDeviceToHost));
        std::vector<int> host(n);
        CK(hipMemcpy(host.data(), s0, chunk, hipMemcpyDeviceToHost));  // the copy engine's view of slot0 (B)
        size_t b_ok = 0;
        for (size_t i = 0; i < n; ++i) b_ok += host[i] == 222;
        printf("round %d: base %p, words of A overwritten through the old slot0 translation: %llu of %zu; B holds the new pattern in %zu of %zu words (copy engine)\n",
               r, base, bad, n, b_ok, n);
        stale_rounds += bad != 0 || b_ok != n;
        CK(hipMemUnmap(s0, chunk));
        CK(hipMemUnmap(s1, chunk));
        CK(hipMemRelease(A));
        CK(hipMemRelease(B));
        CK(hipMemAddressFree(base, 64 * chunk));
    }
    printf("mode %d: %d of %d rounds saw a stale translation\n", mode, stale_rounds, rounds);
    return 0;
}
