"""CPU: the functions the crypto batch kernels execute (py-swirld_amd/csrc/crypto.hip.h: SHA-512,
BLAKE2b-256, GF(2^255-19) / edwards25519 arithmetic, Ed25519 verification), compiled for the HOST
by g++ (tests/crypto_host.cpp) and run against hashlib and libsodium 1.0.18 — the library the
reference reaches through pysodium (swirld.py:10-12, 95-103): equal digests, equal scalars, equal
points, and the same accept / reject decision on valid, corrupted, non-canonical and small-order
inputs.  The GPU run of the same vectors is tests/test_gpu_crypto.py."""
import ctypes as C
import hashlib
import os
import random

import pytest

import hostlibs
from conftest import ROOT

HERE = os.path.dirname(os.path.abspath(__file__))
L_ORDER = 2 ** 252 + 27742317777372353535851937790883648493


def build_host_lib():
    return hostlibs.build_crypto_host()


def load_sodium():
    for cand in (os.environ.get("SWIRLD_LIBSODIUM"), "/opt/conda/lib/libsodium.so", "libsodium.so.23", "libsodium.so"):
        if not cand:
            continue
        try:
            s = C.CDLL(cand)
            if s.sodium_init() >= 0:
                return s
        except OSError:
            pass
    return None


def signed_cases(sod, rng, n_keys, with_adversarial=True):
    """(sig, msg, pk) triples: valid signatures, bit flips in R / S / message / key, S + L, and the
    small-order / non-canonical encodings libsodium refuses."""
    cases = []
    for _ in range(n_keys):
        seed = bytes(rng.getrandbits(8) for _ in range(32))
        pk, sk = C.create_string_buffer(32), C.create_string_buffer(64)
        sod.crypto_sign_seed_keypair(pk, sk, seed)
        m = bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 400)))
        sig = C.create_string_buffer(64)
        sod.crypto_sign_detached(sig, None, m, C.c_ulonglong(len(m)), sk)
        cases.append((sig.raw, m, pk.raw))
        for _ in range(4):
            s2 = bytearray(sig.raw)
            s2[rng.randrange(64)] ^= 1 << rng.randrange(8)
            cases.append((bytes(s2), m, pk.raw))
        if m:
            m2 = bytearray(m)
            m2[rng.randrange(len(m))] ^= 1
            cases.append((sig.raw, bytes(m2), pk.raw))
        p2 = bytearray(pk.raw)
        p2[rng.randrange(32)] ^= 1 << rng.randrange(8)
        cases.append((sig.raw, m, bytes(p2)))
        S = int.from_bytes(sig.raw[32:], "little") + L_ORDER
        if S < 2 ** 256:
            cases.append((sig.raw[:32] + S.to_bytes(32, "little"), m, pk.raw))
    if with_adversarial:
        p = 2 ** 255 - 19
        sig, m, pk = cases[0]
        bad = [0, 1, p - 1, p, p + 1, p + 2, 2 ** 255 - 1,
               2707385501144840649318225287225658788936804267575313519463743609750303402022,
               55188659117513257062467267217118295137698188065244968500265048394206261417927]
        for v in bad:
            for sign in (0, 1):
                enc = (v | (sign << 255)).to_bytes(32, "little")
                cases += [(enc + sig[32:], m, pk), (sig, m, enc), (enc + bytes(32), m, enc), (enc + (1).to_bytes(32, "little"), m, enc)]
        for _ in range(100):
            cases.append((bytes(rng.getrandbits(8) for _ in range(64)), m, bytes(rng.getrandbits(8) for _ in range(32))))
    return cases


def sodium_verify(sod, sig, m, pk):
    return sod.crypto_sign_verify_detached(sig, m, C.c_ulonglong(len(m)), pk) == 0


def test_hashes_scalars_points_and_verification_match_libsodium():
    L = C.CDLL(build_host_lib())
    rng = random.Random(1)
    out32, out64 = C.create_string_buffer(32), C.create_string_buffer(64)
    for n in list(range(0, 270)) + [511, 512, 513, 1000, 4096]:
        m = bytes(rng.getrandbits(8) for _ in range(n))
        L.swc_host_blake2b_256(m, C.c_uint64(n), out32)
        assert out32.raw == hashlib.blake2b(m, digest_size=32).digest()      # == crypto_generichash(m)
        L.swc_host_sha512(m, C.c_uint64(n), out64)
        assert out64.raw == hashlib.sha512(m).digest()
    for h in [bytes(rng.getrandbits(8) for _ in range(64)) for _ in range(200)] + [b"\xff" * 64, bytes(64), L_ORDER.to_bytes(64, "little"),
                                                                                 (L_ORDER - 1).to_bytes(64, "little"), (2 * L_ORDER).to_bytes(64, "little")]:
        L.swc_host_sc_reduce(h, out32)
        assert int.from_bytes(out32.raw, "little") == int.from_bytes(h, "little") % L_ORDER
    sod = load_sodium()
    if sod is None:
        pytest.skip("libsodium not found: digests and scalars checked, signatures need libsodium")
    o2 = C.create_string_buffer(32)
    for _ in range(30):
        s = (rng.getrandbits(255) % L_ORDER).to_bytes(32, "little")
        L.swc_host_scalarmult_base(s, out32)
        if sod.crypto_scalarmult_ed25519_base_noclamp(o2, s) == 0:
            assert out32.raw == o2.raw
    acc = rej = 0
    for sig, m, pk in signed_cases(sod, rng, 150):
        a = sodium_verify(sod, sig, m, pk)
        assert (L.swc_host_verify(sig, m, C.c_uint64(len(m)), pk) == 1) == a
        acc += a
        rej += not a
    assert acc >= 150 and rej > 1000
