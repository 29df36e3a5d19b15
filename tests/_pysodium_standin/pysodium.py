"""Test scaffolding ONLY: a minimal stand-in for the third-party `pysodium`
module that /root/reference/swirld.py imports (swirld.py:10-12, utils.py:5).

It exists so the *unmodified* reference can be imported in the authoring
container to generate golden vectors (tests/golden/make_golden.py) and to run
differential tests.  It is not product code and is never imported by the
package.  Backed by libsodium through ctypes when a libsodium shared object is
found; otherwise by hashlib.blake2b plus a deterministic keyed-hash "signature"
(the virtual-voting hot path never verifies signatures, it only consumes the
signature bytes as opaque data: swirld.py:272, 281).
"""
import ctypes
import ctypes.util
import hashlib
import os

_lib = None
for _cand in ("/opt/conda/lib/libsodium.so", ctypes.util.find_library("sodium")):
    if _cand:
        try:
            _lib = ctypes.CDLL(_cand)
            if _lib.sodium_init() < 0:
                _lib = None
            else:
                break
        except OSError:
            _lib = None

crypto_sign_PUBLICKEYBYTES = 32
crypto_sign_SECRETKEYBYTES = 64
crypto_sign_BYTES = 64
crypto_generichash_BYTES = 32

# optional deterministic RNG hook used by the golden generator
_rng_hook = None


def set_rng(fn):
    """fn(n) -> n bytes; None restores the system RNG."""
    global _rng_hook
    _rng_hook = fn


def randombytes(n):
    if _rng_hook is not None:
        return _rng_hook(n)
    return os.urandom(n)


def crypto_generichash(m, k=b"", outlen=crypto_generichash_BYTES):
    return hashlib.blake2b(m, digest_size=outlen, key=k).digest()


def crypto_sign_seed_keypair(seed):
    assert len(seed) == 32
    if _lib is not None:
        pk = ctypes.create_string_buffer(32)
        sk = ctypes.create_string_buffer(64)
        _lib.crypto_sign_seed_keypair(pk, sk, seed)
        return pk.raw, sk.raw
    pk = hashlib.blake2b(b"pk" + seed, digest_size=32).digest()
    return pk, seed + pk


def crypto_sign_keypair():
    return crypto_sign_seed_keypair(randombytes(32))


def crypto_sign_detached(m, sk):
    if _lib is not None:
        sig = ctypes.create_string_buffer(64)
        _lib.crypto_sign_detached(sig, None, m, ctypes.c_ulonglong(len(m)), sk)
        return sig.raw
    pk = sk[32:]
    return hashlib.blake2b(m, digest_size=64, key=pk).digest()


def crypto_sign_verify_detached(sig, m, pk):
    if _lib is not None:
        if _lib.crypto_sign_verify_detached(sig, m, ctypes.c_ulonglong(len(m)), pk) != 0:
            raise ValueError("signature verification failed")
        return
    if hashlib.blake2b(m, digest_size=64, key=pk).digest() != sig:
        raise ValueError("signature verification failed")


def crypto_sign(m, sk):
    return crypto_sign_detached(m, sk) + m


def crypto_sign_open(sm, pk):
    sig, m = sm[:64], sm[64:]
    crypto_sign_verify_detached(sig, m, pk)
    return m
