// Ingest-side crypto of the hot path's caller: what Node.is_valid_event spends its
// time in (swirld.py:97-103) — Ed25519 signature verification and the BLAKE2b-256 event id — as
// batch kernels, one thread per event.  Written against the published algorithms (RFC 8032 Ed25519,
// FIPS 180-4 SHA-512, RFC 7693 BLAKE2b) and against the ACCEPT / REJECT behaviour of libsodium
// 1.0.18's crypto_sign_verify_detached, which is what the reference reaches through pysodium
// (swirld.py:99-100): S must be canonical (< L), R and the public key must not be of small order,
// the public key must be canonically encoded and decompress, and the check is
// encode([S]B - [h]A) == R byte for byte with h = SHA-512(R || A || M) mod L.
//
// The same source compiles for the device (hipcc) and for the host (g++ -DSW_CRYPTO_HOST): the CPU
// tests run these very functions against libsodium (tests/test_crypto_host.py); the product only
// ever launches the kernels.  Field elements are 4 x 64-bit limbs (values < 2^256, folded with
// 2^256 = 38 mod p); nothing here needs to be constant-time (verification of public data).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) && !defined(SW_CRYPTO_HOST)
#define SW_HD __host__ __device__
#else
#define SW_HD
#endif

namespace swc {

typedef unsigned __int128 u128;
typedef uint64_t u64;

// ------------------------------------------------------------------ SHA-512 (FIPS 180-4)
SW_HD inline u64 sha512_k(int i) {
    const u64 K[80] = {
    0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull,
    0x3956c25bf348b538ull, 0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull,
    0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
    0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull,
    0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
    0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
    0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull,
    0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull, 0x06ca6351e003826full, 0x142929670a0e6e70ull,
    0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
    0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
    0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull,
    0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
    0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull,
    0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull,
    0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
    0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull,
    0xca273eceea26619cull, 0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull,
    0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
    0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull,
    0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
    return K[i];
}
SW_HD inline u64 rotr64(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
SW_HD inline u64 load64_be(const uint8_t* p) {
    u64 v = 0;
    for (int i = 0; i < 8; ++i) v = (v << 8) | p[i];
    return v;
}
SW_HD inline u64 load64_le(const uint8_t* p) {
    u64 v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}
SW_HD inline void store64_le(uint8_t* p, u64 v) { for (int i = 0; i < 8; ++i) { p[i] = (uint8_t)v; v >>= 8; } }
SW_HD inline void store64_be(uint8_t* p, u64 v) { for (int i = 7; i >= 0; --i) { p[i] = (uint8_t)v; v >>= 8; } }

struct Sha512 {
    u64 h[8];
    uint8_t buf[128];
    u64 len;   // bytes absorbed
};
SW_HD inline void sha512_block(u64* h, const uint8_t* blk) {
    u64 w[80];
    for (int i = 0; i < 16; ++i) w[i] = load64_be(blk + 8 * i);
    for (int i = 16; i < 80; ++i) {
        const u64 s0 = rotr64(w[i - 15], 1) ^ rotr64(w[i - 15], 8) ^ (w[i - 15] >> 7);
        const u64 s1 = rotr64(w[i - 2], 19) ^ rotr64(w[i - 2], 61) ^ (w[i - 2] >> 6);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    u64 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 80; ++i) {
        const u64 S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41);
        const u64 ch = (e & f) ^ (~e & g);
        const u64 t1 = hh + S1 + ch + sha512_k(i) + w[i];
        const u64 S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39);
        const u64 mj = (a & b) ^ (a & c) ^ (b & c);
        const u64 t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
SW_HD inline void sha512_init(Sha512* s) {
    const u64 iv[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                       0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    for (int i = 0; i < 8; ++i) s->h[i] = iv[i];
    s->len = 0;
}
SW_HD inline void sha512_update(Sha512* s, const uint8_t* m, u64 n) {
    for (u64 i = 0; i < n; ++i) {
        s->buf[s->len & 127] = m[i];
        s->len++;
        if ((s->len & 127) == 0) sha512_block(s->h, s->buf);
    }
}
SW_HD inline void sha512_final(Sha512* s, uint8_t out[64]) {
    const u64 bits = s->len * 8;
    u64 r = s->len & 127;
    s->buf[r++] = 0x80;
    if (r > 112) { while (r < 128) s->buf[r++] = 0; sha512_block(s->h, s->buf); r = 0; }
    while (r < 120) s->buf[r++] = 0;   // (the high 64 bits of the 128-bit length are zero)
    store64_be(s->buf + 120, bits);
    sha512_block(s->h, s->buf);
    for (int i = 0; i < 8; ++i) store64_be(out + 8 * i, s->h[i]);
}

// ------------------------------------------------------------------ BLAKE2b (RFC 7693), unkeyed
SW_HD inline uint8_t blake2b_sigma(int r, int i) {
    const uint8_t S[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    return S[r % 10][i];
}
SW_HD inline void blake2b_compress(u64* h, const uint8_t* blk, u64 t, bool last) {
    const u64 iv[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                       0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    u64 m[16], v[16];
    for (int i = 0; i < 16; ++i) m[i] = load64_le(blk + 8 * i);
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[8 + i] = iv[i]; }
    v[12] ^= t;   // (message length < 2^64)
    if (last) v[14] = ~v[14];
#define SW_B2G(a, b, c, d, x, y)                                     \
    do {                                                             \
        v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32);    \
        v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24);    \
        v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16);    \
        v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63);    \
    } while (0)
    for (int r = 0; r < 12; ++r) {
        SW_B2G(0, 4, 8, 12, m[blake2b_sigma(r, 0)], m[blake2b_sigma(r, 1)]);
        SW_B2G(1, 5, 9, 13, m[blake2b_sigma(r, 2)], m[blake2b_sigma(r, 3)]);
        SW_B2G(2, 6, 10, 14, m[blake2b_sigma(r, 4)], m[blake2b_sigma(r, 5)]);
        SW_B2G(3, 7, 11, 15, m[blake2b_sigma(r, 6)], m[blake2b_sigma(r, 7)]);
        SW_B2G(0, 5, 10, 15, m[blake2b_sigma(r, 8)], m[blake2b_sigma(r, 9)]);
        SW_B2G(1, 6, 11, 12, m[blake2b_sigma(r, 10)], m[blake2b_sigma(r, 11)]);
        SW_B2G(2, 7, 8, 13, m[blake2b_sigma(r, 12)], m[blake2b_sigma(r, 13)]);
        SW_B2G(3, 4, 9, 14, m[blake2b_sigma(r, 14)], m[blake2b_sigma(r, 15)]);
    }
#undef SW_B2G
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[8 + i];
}
// BLAKE2b with a 32-byte digest and no key == libsodium crypto_generichash(m) with default arguments
// (swirld.py:95, 103): the event id.
SW_HD inline void blake2b_256(const uint8_t* m, u64 n, uint8_t out[32]) {
    const u64 iv[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                       0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    u64 h[8];
    for (int i = 0; i < 8; ++i) h[i] = iv[i];
    h[0] ^= 0x01010000ull ^ 32ull;  // digest length 32, no key, fanout = depth = 1
    uint8_t blk[128];
    u64 off = 0;
    while (n - off > 128) {
        for (int i = 0; i < 128; ++i) blk[i] = m[off + i];
        off += 128;
        blake2b_compress(h, blk, off, false);
    }
    const u64 rem = n - off;
    for (u64 i = 0; i < 128; ++i) blk[i] = i < rem ? m[off + i] : 0;
    blake2b_compress(h, blk, n, true);
    for (int i = 0; i < 4; ++i) store64_le(out + 8 * i, h[i]);
}

// ------------------------------------------------------------------ GF(2^255 - 19), 4 x 64-bit limbs
struct fe { u64 v[4]; };
SW_HD inline fe fe_const(u64 a, u64 b, u64 c, u64 d) { fe r; r.v[0] = a; r.v[1] = b; r.v[2] = c; r.v[3] = d; return r; }
SW_HD inline fe fe_zero() { return fe_const(0, 0, 0, 0); }
SW_HD inline fe fe_one() { return fe_const(1, 0, 0, 0); }
// r = a + 38 * c (c small), value stays < 2^256 after at most two folds
SW_HD inline void fe_fold(fe& r, u64 c) {
    while (c) {
        u128 t = (u128)r.v[0] + (u128)c * 38u;
        r.v[0] = (u64)t;
        u64 carry = (u64)(t >> 64);
        for (int i = 1; i < 4; ++i) { t = (u128)r.v[i] + carry; r.v[i] = (u64)t; carry = (u64)(t >> 64); }
        c = carry;
    }
}
SW_HD inline fe fe_add(const fe& a, const fe& b) {
    fe r;
    u64 carry = 0;
    for (int i = 0; i < 4; ++i) { const u128 t = (u128)a.v[i] + b.v[i] + carry; r.v[i] = (u64)t; carry = (u64)(t >> 64); }
    fe_fold(r, carry);
    return r;
}
SW_HD inline fe fe_sub(const fe& a, const fe& b) {
    fe r;
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        const u128 t = (u128)a.v[i] - b.v[i] - borrow;
        r.v[i] = (u64)t;
        borrow = (u64)(t >> 64) & 1u;
    }
    while (borrow) {  // the wrap added 2^256 = 38 (mod p): take it out again
        u128 t = (u128)r.v[0] - 38u;
        r.v[0] = (u64)t;
        u64 bw = (u64)(t >> 64) & 1u;
        for (int i = 1; i < 4; ++i) { t = (u128)r.v[i] - bw; r.v[i] = (u64)t; bw = (u64)(t >> 64) & 1u; }
        borrow = bw;
    }
    return r;
}
SW_HD inline fe fe_mul(const fe& a, const fe& b) {
    u64 t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u64 carry = 0;
        for (int j = 0; j < 4; ++j) {
            const u128 p = (u128)a.v[i] * b.v[j] + t[i + j] + carry;
            t[i + j] = (u64)p;
            carry = (u64)(p >> 64);
        }
        t[i + 4] = carry;
    }
    fe r;
    u64 carry = 0;
    for (int i = 0; i < 4; ++i) {  // lo + 38 * hi
        const u128 p = (u128)t[i + 4] * 38u + t[i] + carry;
        r.v[i] = (u64)p;
        carry = (u64)(p >> 64);
    }
    fe_fold(r, carry);
    return r;
}
SW_HD inline fe fe_sq(const fe& a) { return fe_mul(a, a); }
SW_HD inline fe fe_neg(const fe& a) { return fe_sub(fe_zero(), a); }
// canonical 32-byte little-endian encoding (fully reduced)
SW_HD inline void fe_tobytes(uint8_t out[32], const fe& a) {
    fe r = a;
    for (int pass = 0; pass < 2; ++pass) {  // fold bit 255: 2^255 = 19 (mod p)
        const u64 top = r.v[3] >> 63;
        r.v[3] &= 0x7fffffffffffffffull;
        u128 t = (u128)r.v[0] + (u128)top * 19u;
        r.v[0] = (u64)t;
        u64 carry = (u64)(t >> 64);
        for (int i = 1; i < 4; ++i) { t = (u128)r.v[i] + carry; r.v[i] = (u64)t; carry = (u64)(t >> 64); }
    }
    // now r < 2^255: subtract p once if r >= p
    const u64 p0 = 0xffffffffffffffedull, p1 = 0xffffffffffffffffull, p3 = 0x7fffffffffffffffull;
    const bool ge = r.v[3] == p3 && r.v[2] == p1 && r.v[1] == p1 && r.v[0] >= p0;
    if (ge) { r.v[0] -= p0; r.v[1] = 0; r.v[2] = 0; r.v[3] = 0; }
    for (int i = 0; i < 4; ++i) store64_le(out + 8 * i, r.v[i]);
}
SW_HD inline fe fe_frombytes(const uint8_t in[32]) {  // ignores bit 255, accepts y >= p like ref10
    fe r;
    for (int i = 0; i < 4; ++i) r.v[i] = load64_le(in + 8 * i);
    r.v[3] &= 0x7fffffffffffffffull;
    return r;
}
SW_HD inline bool fe_iszero(const fe& a) {
    uint8_t b[32];
    fe_tobytes(b, a);
    uint8_t acc = 0;
    for (int i = 0; i < 32; ++i) acc |= b[i];
    return acc == 0;
}
SW_HD inline int fe_isnegative(const fe& a) {
    uint8_t b[32];
    fe_tobytes(b, a);
    return b[0] & 1;
}
SW_HD inline fe fe_pow2k(fe a, int k) { for (int i = 0; i < k; ++i) a = fe_sq(a); return a; }
// a^(2^252 - 3) = a^((p-5)/8) and a^(p-2) by the usual addition chain on 2^k - 1 exponents
SW_HD inline void fe_chain(const fe& z, fe* z11_out, fe* z2_250_0_out) {
    const fe z2 = fe_sq(z);
    const fe z9 = fe_mul(fe_pow2k(z2, 2), z);
    const fe z11 = fe_mul(z9, z2);
    const fe z2_5_0 = fe_mul(fe_sq(z11), z9);                    // 2^5 - 1
    const fe z2_10_0 = fe_mul(fe_pow2k(z2_5_0, 5), z2_5_0);      // 2^10 - 1
    const fe z2_20_0 = fe_mul(fe_pow2k(z2_10_0, 10), z2_10_0);
    const fe z2_40_0 = fe_mul(fe_pow2k(z2_20_0, 20), z2_20_0);
    const fe z2_50_0 = fe_mul(fe_pow2k(z2_40_0, 10), z2_10_0);
    const fe z2_100_0 = fe_mul(fe_pow2k(z2_50_0, 50), z2_50_0);
    const fe z2_200_0 = fe_mul(fe_pow2k(z2_100_0, 100), z2_100_0);
    *z2_250_0_out = fe_mul(fe_pow2k(z2_200_0, 50), z2_50_0);     // 2^250 - 1
    *z11_out = z11;
}
SW_HD inline fe fe_invert(const fe& z) {  // z^(p-2) = z^(2^255 - 21)
    fe z11, t;
    fe_chain(z, &z11, &t);
    return fe_mul(fe_pow2k(t, 5), z11);   // (2^250 - 1) * 32 + 11
}
SW_HD inline fe fe_pow22523(const fe& z) {  // z^(2^252 - 3)
    fe z11, t;
    fe_chain(z, &z11, &t);
    return fe_mul(fe_pow2k(t, 2), z);     // (2^250 - 1) * 4 + 1
}

// ------------------------------------------------------------------ edwards25519 points (extended coordinates, a = -1)
struct ge { fe X, Y, Z, T; };
SW_HD inline fe ge_d() { return fe_const(0x75eb4dca135978a3ull, 0x00700a4d4141d8abull, 0x8cc740797779e898ull, 0x52036cee2b6ffe73ull); }
SW_HD inline fe ge_2d() { return fe_const(0xebd69b9426b2f159ull, 0x00e0149a8283b156ull, 0x198e80f2eef3d130ull, 0x2406d9dc56dffce7ull); }
SW_HD inline fe ge_sqrtm1() { return fe_const(0xc4ee1b274a0ea0b0ull, 0x2f431806ad2fe478ull, 0x2b4d00993dfbd7a7ull, 0x2b8324804fc1df0bull); }
SW_HD inline ge ge_base() {
    ge b;
    b.X = fe_const(0xc9562d608f25d51aull, 0x692cc7609525a7b2ull, 0xc0a4e231fdd6dc5cull, 0x216936d3cd6e53feull);
    b.Y = fe_const(0x6666666666666658ull, 0x6666666666666666ull, 0x6666666666666666ull, 0x6666666666666666ull);
    b.Z = fe_one();
    b.T = fe_const(0x6dde8ab3a5b7dda3ull, 0x20f09f80775152f5ull, 0x66ea4e8e64abe37dull, 0x67875f0fd78b7665ull);
    return b;
}
SW_HD inline ge ge_identity() { ge r; r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.T = fe_zero(); return r; }
SW_HD inline ge ge_add(const ge& p, const ge& q) {  // add-2008-hwcd-3 (unified)
    const fe A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    const fe B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    const fe C = fe_mul(fe_mul(p.T, ge_2d()), q.T);
    const fe ZZ = fe_mul(p.Z, q.Z);
    const fe D = fe_add(ZZ, ZZ);
    const fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    ge r;
    r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(F, G);
    return r;
}
SW_HD inline ge ge_double(const ge& p) {  // dbl-2008-hwcd
    const fe A = fe_sq(p.X), B = fe_sq(p.Y);
    const fe ZZ = fe_sq(p.Z);
    const fe C = fe_add(ZZ, ZZ);
    const fe D = fe_neg(A);
    const fe xy = fe_add(p.X, p.Y);
    const fe E = fe_sub(fe_sub(fe_sq(xy), A), B);
    const fe G = fe_add(D, B), F = fe_sub(G, C), H = fe_sub(D, B);
    ge r;
    r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(F, G);
    return r;
}
SW_HD inline ge ge_neg(const ge& p) { ge r = p; r.X = fe_neg(p.X); r.T = fe_neg(p.T); return r; }
SW_HD inline void ge_tobytes(uint8_t out[32], const ge& p) {
    const fe zi = fe_invert(p.Z);
    const fe x = fe_mul(p.X, zi), y = fe_mul(p.Y, zi);
    fe_tobytes(out, y);
    out[31] ^= (uint8_t)(fe_isnegative(x) << 7);
}
// decompression as ref10's ge_frombytes (no negate): false when the encoding is not on the curve
SW_HD inline bool ge_frombytes(ge* h, const uint8_t s[32]) {
    h->Y = fe_frombytes(s);
    h->Z = fe_one();
    const fe y2 = fe_sq(h->Y);
    const fe u = fe_sub(y2, fe_one());              // y^2 - 1
    const fe v = fe_add(fe_mul(y2, ge_d()), fe_one());  // d y^2 + 1
    const fe v3 = fe_mul(fe_sq(v), v);
    const fe v7 = fe_mul(fe_sq(v3), v);
    fe x = fe_mul(fe_mul(fe_pow22523(fe_mul(u, v7)), v3), u);  // u v^3 (u v^7)^((p-5)/8)
    const fe vxx = fe_mul(fe_sq(x), v);
    if (!fe_iszero(fe_sub(vxx, u))) {
        if (!fe_iszero(fe_add(vxx, u))) return false;
        x = fe_mul(x, ge_sqrtm1());
    }
    if (fe_isnegative(x) != (s[31] >> 7)) x = fe_neg(x);
    h->X = x;
    h->T = fe_mul(h->X, h->Y);
    return true;
}

// ------------------------------------------------------------------ scalars mod L = 2^252 + 27742317777372353535851937790883648493
SW_HD inline u64 sc_L(int i) {
    const u64 L[4] = {0x5812631a5cf5d3edull, 0x14def9dea2f79cd6ull, 0x0000000000000000ull, 0x1000000000000000ull};
    return L[i];
}
SW_HD inline bool sc_is_canonical(const uint8_t s[32]) {  // s < L
    for (int i = 3; i >= 0; --i) {
        const u64 w = load64_le(s + 8 * i);
        if (w < sc_L(i)) return true;
        if (w > sc_L(i)) return false;
    }
    return false;
}
// 512-bit little-endian h -> h mod L (bitwise shift-and-subtract: 512 cheap steps, verification only)
SW_HD inline void sc_reduce512(const uint8_t h[64], u64 r[4]) {
    u64 acc[5] = {0, 0, 0, 0, 0};
    for (int bit = 511; bit >= 0; --bit) {
        // acc = 2 * acc + bit
        u64 carry = (h[bit >> 3] >> (bit & 7)) & 1u;
        for (int i = 0; i < 5; ++i) { const u64 nc = acc[i] >> 63; acc[i] = (acc[i] << 1) | carry; carry = nc; }
        // if acc >= L: acc -= L   (acc < 2L always)
        bool ge_ = acc[4] != 0;
        if (!ge_) {
            ge_ = true;
            for (int i = 3; i >= 0; --i) {
                if (acc[i] > sc_L(i)) break;
                if (acc[i] < sc_L(i)) { ge_ = false; break; }
            }
        }
        if (ge_) {
            u64 borrow = 0;
            for (int i = 0; i < 4; ++i) {
                const u128 t = (u128)acc[i] - sc_L(i) - borrow;
                acc[i] = (u64)t;
                borrow = (u64)(t >> 64) & 1u;
            }
            acc[4] -= borrow;
        }
    }
    for (int i = 0; i < 4; ++i) r[i] = acc[i];
}

// ------------------------------------------------------------------ Ed25519 verification (libsodium 1.0.18 behaviour)
SW_HD inline bool ge_has_small_order(const uint8_t s[32]) {
    const uint8_t bl[7][32] = {
        {0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00},
        {0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00},
        {0x26, 0xe8, 0x95, 0x8f, 0xc2, 0xb2, 0x27, 0xb0, 0x45, 0xc3, 0xf4, 0x89, 0xf2, 0xef, 0x98, 0xf0, 0xd5, 0xdf, 0xac, 0x05, 0xd3, 0xc6, 0x33, 0x39, 0xb1, 0x38, 0x02, 0x88, 0x6d, 0x53, 0xfc, 0x05},
        {0xc7, 0x17, 0x6a, 0x70, 0x3d, 0x4d, 0xd8, 0x4f, 0xba, 0x3c, 0x0b, 0x76, 0x0d, 0x10, 0x67, 0x0f, 0x2a, 0x20, 0x53, 0xfa, 0x2c, 0x39, 0xcc, 0xc6, 0x4e, 0xc7, 0xfd, 0x77, 0x92, 0xac, 0x03, 0x7a},
        {0xec, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0x7f},
        {0xed, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0x7f},
        {0xee, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0x7f}};
    for (int k = 0; k < 7; ++k) {
        uint8_t acc = 0;
        for (int j = 0; j < 31; ++j) acc |= s[j] ^ bl[k][j];
        acc |= (s[31] & 0x7f) ^ bl[k][31];
        if (acc == 0) return true;
    }
    return false;
}
SW_HD inline bool ge_is_canonical(const uint8_t s[32]) {  // y < p, sign bit ignored
    const u64 w3 = load64_le(s + 24) & 0x7fffffffffffffffull;
    if (w3 != 0x7fffffffffffffffull) return true;
    if (load64_le(s + 16) != ~0ull || load64_le(s + 8) != ~0ull) return true;
    return load64_le(s) < 0xffffffffffffffedull;
}
// true iff libsodium's crypto_sign_verify_detached(sig, m, mlen, pk) returns 0
SW_HD inline bool ed25519_verify(const uint8_t sig[64], const uint8_t* m, u64 mlen, const uint8_t pk[32]) {
    if (!sc_is_canonical(sig + 32)) return false;
    if (ge_has_small_order(sig)) return false;
    if (!ge_is_canonical(pk) || ge_has_small_order(pk)) return false;
    ge A;
    if (!ge_frombytes(&A, pk)) return false;
    uint8_t hbytes[64];
    {
        Sha512 sh;
        sha512_init(&sh);
        sha512_update(&sh, sig, 32);
        sha512_update(&sh, pk, 32);
        sha512_update(&sh, m, mlen);
        sha512_final(&sh, hbytes);
    }
    u64 h[4], S[4];
    sc_reduce512(hbytes, h);
    for (int i = 0; i < 4; ++i) S[i] = load64_le(sig + 32 + 8 * i);
    // R' = [S]B + [h](-A): joint double-and-add over the 253 bits of S and h
    const ge B = ge_base(), nA = ge_neg(A), BnA = ge_add(B, nA);
    ge R = ge_identity();
    for (int bit = 252; bit >= 0; --bit) {
        R = ge_double(R);
        const int sb = (int)((S[bit >> 6] >> (bit & 63)) & 1u), hb = (int)((h[bit >> 6] >> (bit & 63)) & 1u);
        if (sb && hb) R = ge_add(R, BnA);
        else if (sb) R = ge_add(R, B);
        else if (hb) R = ge_add(R, nA);
    }
    uint8_t rcheck[32];
    ge_tobytes(rcheck, R);
    uint8_t acc = 0;
    for (int i = 0; i < 32; ++i) acc |= rcheck[i] ^ sig[i];
    return acc == 0;
}

}  // namespace swc
