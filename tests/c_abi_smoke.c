/* A plain-C user of include/swirld_hip.h (no Python, no torch): what a cgo/JNI/FFI binding of
 * any host language would do.  Built and run by tests/test_gpu_c_abi.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "swirld_hip.h"

#define CK(x) do { int rc_ = (x); if (rc_ != SW_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, sw_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 32;
    const long long N = argc > 2 ? atoll(argv[2]) : 20000;
    sw_ctx* ctx = NULL;
    uint64_t* stake = malloc(sizeof(uint64_t) * n);
    for (int i = 0; i < n; ++i) stake[i] = 1;
    int32_t *cr = malloc(4 * N), *sp = malloc(4 * N), *op = malloc(4 * N), *rnd = malloc(4 * N);
    double* t = malloc(8 * N);
    uint8_t* sig = malloc(64 * N);
    if (sw_synth_hashgraph(n, N, 11, 0, 0.0, 0.0, cr, sp, op, t, sig) != SW_OK) return 2;
    if (sw_create(n, stake, 6, 0, &ctx) != SW_OK) { fprintf(stderr, "sw_create: %s\n", sw_last_error(NULL)); return 3; }
    CK(sw_append_events(ctx, N, cr, sp, op, t, sig));
    CK(sw_divide_rounds(ctx, 0, N));
    int maxr = -1, n_new = 0;
    CK(sw_max_round(ctx, &maxr));
    int32_t* newc = malloc(4 * (maxr + 2));
    CK(sw_decide_fame(ctx, newc, maxr + 2, &n_new));
    int32_t* order = malloc(4 * N);
    int64_t n_ord = 0;
    CK(sw_find_order(ctx, newc, n_new, order, N, &n_ord));
    CK(sw_get_round(ctx, 0, N, rnd));
    unsigned long long h = 1469598103934665603ull; /* FNV-1a over rounds and the order */
    for (long long i = 0; i < N; ++i) { h ^= (unsigned)rnd[i]; h *= 1099511628211ull; }
    for (long long i = 0; i < n_ord; ++i) { h ^= (unsigned)order[i]; h *= 1099511628211ull; }
    printf("%d %d %lld %llu\n", maxr, n_new, (long long)n_ord, h);
    sw_destroy(ctx);
    return 0;
}
