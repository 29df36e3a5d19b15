// Microbenchmark / feasibility check for the windowed can_see table: one virtual
// address range reserved for the whole table, physical chunks mapped as events arrive and unmapped
// below the eviction horizon — the kernels keep one unchanged base pointer.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void fill(int* p, size_t n, int v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v + (int)(i & 1023); }
__global__ void sum(const int* p, size_t n, unsigned long long* out) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) atomicAdd(out, (unsigned long long)p[i]); }
int main() {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    size_t gran_rec = 0;
    CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity min %zu recommended %zu\n", gran, gran_rec);
    const size_t chunk = ((64ull << 20) + gran - 1) / gran * gran;
    const size_t va = 256ull << 30;  // 256 GB of address space
    void* base = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    CK(hipMemAddressReserve(&base, va, 0, nullptr, 0));
    auto t1 = std::chrono::steady_clock::now();
    printf("reserved %zu GB at %p in %.1f us\n", va >> 30, base, std::chrono::duration<double, std::micro>(t1 - t0).count());
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const int NCH = 6;
    std::vector<hipMemGenericAllocationHandle_t> h(NCH);
    double map_us = 0;
    for (int i = 0; i < NCH; ++i) {
        auto a = std::chrono::steady_clock::now();
        CK(hipMemCreate(&h[i], chunk, &prop, 0));
        CK(hipMemMap((char*)base + i * chunk, chunk, 0, h[i], 0));
        CK(hipMemSetAccess((char*)base + i * chunk, chunk, &acc, 1));
        map_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
    }
    printf("mapped %d chunks of %zu MB, %.1f us per chunk\n", NCH, chunk >> 20, map_us / NCH);
    const size_t n = NCH * chunk / 4;
    fill<<<(unsigned)((n + 255) / 256), 256>>>((int*)base, n, 7);
    CK(hipDeviceSynchronize());
    // evict the first two chunks, keep using the rest through the same base pointer
    auto a = std::chrono::steady_clock::now();
    for (int i = 0; i < 2; ++i) { CK(hipMemUnmap((char*)base + i * chunk, chunk)); CK(hipMemRelease(h[i])); }
    printf("unmapped 2 chunks in %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count());
    unsigned long long* d_out; CK(hipMalloc(&d_out, 8)); CK(hipMemset(d_out, 0, 8));
    const size_t off = 2 * chunk / 4, m = n - off;
    sum<<<(unsigned)((m + 255) / 256), 256>>>((const int*)base + off, m, d_out);
    unsigned long long s = 0; CK(hipMemcpy(&s, d_out, 8, hipMemcpyDeviceToHost));
    unsigned long long exp = 0; for (size_t i = off; i < n; ++i) exp += 7 + (i & 1023);
    printf("sum over the resident part %llu (expected %llu) %s\n", s, exp, s == exp ? "OK" : "MISMATCH");
    // map a chunk again at an evicted address (rewind) and free memory accounting
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    CK(hipMemCreate(&h[0], chunk, &prop, 0)); CK(hipMemMap(base, chunk, 0, h[0], 0)); CK(hipMemSetAccess(base, chunk, &acc, 1));
    fill<<<(unsigned)((chunk / 4 + 255) / 256), 256>>>((int*)base, chunk / 4, 1);
    CK(hipDeviceSynchronize());
    printf("re-mapped chunk 0: OK; free %zu MB of %zu MB\n", fr >> 20, tot >> 20);
    for (int i = 0; i < NCH; ++i) if (i != 1) { CK(hipMemUnmap((char*)base + i * chunk, chunk)); CK(hipMemRelease(h[i])); }
    CK(hipMemAddressFree(base, va));
    printf("done\n");
    return 0;
}
