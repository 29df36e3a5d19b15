// Microbenchmark: is straight-line code executed once per launch paying instruction-cache misses?
// A kernel runs the same 16 KB unrolled body three times and stamps each pass (100 MHz clock);
// launched back to back several times.  pass 0 >> pass 1,2 means cold instruction fetch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int UN>
__global__ void k_body(unsigned* out, unsigned long long* stamps, int launch, unsigned seed) {
    unsigned a = seed + threadIdx.x, b = a ^ 0x9e3779b9u, c = a + 77u, d = b + 13u;
    unsigned long long t[4];
    for (int pass = 0; pass < 3; ++pass) {
        t[pass] = wall_clock64();
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            a = a * 3u + (unsigned)i; b = b * 5u + a; c = c * 7u + 1u; d = d * 9u + c;
        }
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    t[3] = wall_clock64();
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
        const int k = (launch * 2 + (blockIdx.x ? 1 : 0)) * 4;
        for (int i = 0; i < 4; ++i) stamps[k + i] = t[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

template <int UN>
void run(const char* name, int nb, unsigned* out, unsigned long long* stamps, hipStream_t s) {
    const int L = 6;
    CK(hipMemsetAsync(stamps, 0, 4096, s));
    for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_body<UN>, dim3(nb), dim3(256), 0, s, out, stamps, l, 17u + l);
    CK(hipStreamSynchronize(s));
    unsigned long long h[64]; CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
    printf("%s, %d blocks x 256:\n", name, nb);
    for (int l = 0; l < L; ++l)
        for (int b = 0; b < 2; ++b) {
            unsigned long long* t = h + (l * 2 + b) * 4;
            printf("  launch %d %s block: pass0 %.2f us  pass1 %.2f us  pass2 %.2f us\n", l, b ? "last " : "first",
                   (t[1] - t[0]) / 100.0, (t[2] - t[1]) / 100.0, (t[3] - t[2]) / 100.0);
        }
}

int main() {
    unsigned* out; unsigned long long* stamps;
    CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc(&stamps, 4096));
    hipStream_t s; CK(hipStreamCreate(&s));
    run<128>("body 128 x 8 instr (~8 KB)", 512, out, stamps, s);
    run<512>("body 512 x 8 instr (~32 KB)", 512, out, stamps, s);
    run<512>("body 512 x 8 instr (~32 KB)", 8, out, stamps, s);
    return 0;
}
