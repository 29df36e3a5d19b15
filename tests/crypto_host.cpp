// Host build of crypto.hip.h (g++ -DSW_CRYPTO_HOST) — TEST INFRASTRUCTURE: lets the CPU suite run the
// very functions the batch kernels execute against libsodium (tests/test_crypto_host.py).  The
// product never loads this library; it launches the kernels of swirld_hip.hip.
#define SW_CRYPTO_HOST 1
#include "../py-swirld_amd/csrc/crypto.hip.h"

extern "C" {
int swc_host_verify(const uint8_t* sig, const uint8_t* m, uint64_t mlen, const uint8_t* pk) { return swc::ed25519_verify(sig, m, mlen, pk) ? 1 : 0; }
void swc_host_blake2b_256(const uint8_t* m, uint64_t n, uint8_t* out) { swc::blake2b_256(m, n, out); }
void swc_host_sha512(const uint8_t* m, uint64_t n, uint8_t* out) {
    swc::Sha512 s;
    swc::sha512_init(&s);
    swc::sha512_update(&s, m, n);
    swc::sha512_final(&s, out);
}
void swc_host_sc_reduce(const uint8_t* h64, uint8_t* out32) {
    uint64_t r[4];
    swc::sc_reduce512(h64, r);
    for (int i = 0; i < 4; ++i) swc::store64_le(out32 + 8 * i, r[i]);
}
void swc_host_scalarmult_base(const uint8_t* s32, uint8_t* out32) {  // [s]B, for cross-checks against libsodium
    swc::ge R = swc::ge_identity();
    const swc::ge B = swc::ge_base();
    for (int bit = 255; bit >= 0; --bit) {
        R = swc::ge_double(R);
        if ((s32[bit >> 3] >> (bit & 7)) & 1) R = swc::ge_add(R, B);
    }
    swc::ge_tobytes(out32, R);
}
}
