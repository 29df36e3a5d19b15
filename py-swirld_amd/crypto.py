"""Ed25519 / BLAKE2b / RNG used by the gossip side of Node (swirld.py:10-12, 91-103).

Out of the accelerated path: the voting kernels consume the
signature only as opaque bytes.  Backed by libsodium through ctypes (the reference uses
pysodium, a ctypes wrapper of the same library).  Like the reference, which hard-fails on a
missing pysodium (swirld.py:10), this module FAILS LOUDLY when libsodium cannot be loaded:
signing raises ImportError.  The only exception is an explicit opt-in for single-process
simulations, SWIRLD_INSECURE_CRYPTO=1, which installs a keyed-BLAKE2b stand-in that anyone
knowing a public key can forge — it warns at import and must never face a real network.
"""
import ctypes
import ctypes.util
import hashlib
import os
import warnings

_sodium = None
for _cand in (os.environ.get("SWIRLD_LIBSODIUM"), ctypes.util.find_library("sodium"),
              "/opt/conda/lib/libsodium.so"):
    if not _cand:
        continue
    try:
        _l = ctypes.CDLL(_cand)
        _l.sodium_init.restype = ctypes.c_int
        if _l.sodium_init() >= 0:
            _sodium = _l
            break
    except (OSError, AttributeError):
        pass

HAVE_SODIUM = _sodium is not None
INSECURE_STANDIN = (not HAVE_SODIUM) and os.environ.get("SWIRLD_INSECURE_CRYPTO") == "1"

if HAVE_SODIUM:
    _cp, _ull = ctypes.c_char_p, ctypes.c_ulonglong
    _sodium.crypto_sign_seed_keypair.argtypes = [_cp, _cp, _cp]
    _sodium.crypto_sign_seed_keypair.restype = ctypes.c_int
    _sodium.crypto_sign_detached.argtypes = [_cp, ctypes.c_void_p, _cp, _ull, _cp]
    _sodium.crypto_sign_detached.restype = ctypes.c_int
    _sodium.crypto_sign_verify_detached.argtypes = [_cp, _cp, _ull, _cp]
    _sodium.crypto_sign_verify_detached.restype = ctypes.c_int
elif INSECURE_STANDIN:
    warnings.warn("py-swirld_amd.crypto: libsodium not found and SWIRLD_INSECURE_CRYPTO=1 — using a keyed-BLAKE2b "
                  "stand-in for Ed25519 that is FORGEABLE by anyone who knows a public key; simulation only",
                  RuntimeWarning, stacklevel=2)


def _need_backend():
    if not (HAVE_SODIUM or INSECURE_STANDIN):
        raise ImportError("libsodium not found (set SWIRLD_LIBSODIUM=/path/to/libsodium.so); the reference needs "
                          "pysodium/libsodium as well (swirld.py:10).  For a single-process simulation without real "
                          "signatures set SWIRLD_INSECURE_CRYPTO=1 (forgeable stand-in).")


def _bytes(x, n, what):
    """pysodium-style argument check: exact type and length, ValueError otherwise (so that
    is_valid_event, which catches ValueError like swirld.py:100, rejects malformed input)."""
    if not isinstance(x, (bytes, bytearray)):
        raise ValueError("%s must be bytes" % what)
    if n is not None and len(x) != n:
        raise ValueError("%s must be %d bytes, got %d" % (what, n, len(x)))
    return bytes(x)


def randombytes(n):
    return os.urandom(n)


def generichash(m):
    """BLAKE2b-256 == libsodium crypto_generichash with default arguments."""
    return hashlib.blake2b(m, digest_size=32).digest()


def sign_seed_keypair(seed):
    _need_backend()
    seed = _bytes(seed, 32, "seed")
    if HAVE_SODIUM:
        pk, sk = ctypes.create_string_buffer(32), ctypes.create_string_buffer(64)
        _sodium.crypto_sign_seed_keypair(pk, sk, seed)
        return pk.raw, sk.raw
    pk = hashlib.blake2b(b"swirld-pk" + seed, digest_size=32).digest()
    return pk, seed + pk


def sign_keypair():
    return sign_seed_keypair(randombytes(32))


def sign_detached(m, sk):
    _need_backend()
    m, sk = _bytes(m, None, "message"), _bytes(sk, 64, "secret key")
    if HAVE_SODIUM:
        sig = ctypes.create_string_buffer(64)
        _sodium.crypto_sign_detached(sig, None, m, len(m), sk)
        return sig.raw
    return hashlib.blake2b(m, digest_size=64, key=sk[32:]).digest()


def verify_detached(sig, m, pk):
    """Raises ValueError on a bad signature (what swirld.py:100 expects)."""
    _need_backend()
    sig, m, pk = _bytes(sig, 64, "signature"), _bytes(m, None, "message"), _bytes(pk, 32, "public key")
    if HAVE_SODIUM:
        if _sodium.crypto_sign_verify_detached(sig, m, len(m), pk) != 0:
            raise ValueError("invalid signature")
    elif hashlib.blake2b(m, digest_size=64, key=pk).digest() != sig:
        raise ValueError("invalid signature")


def sign(m, sk):
    return sign_detached(m, sk) + m


def sign_open(sm, pk):
    sm = _bytes(sm, None, "signed message")
    if len(sm) < 64:
        raise ValueError("signed message shorter than a signature")
    verify_detached(sm[:64], sm[64:], pk)
    return sm[64:]
