"""Ed25519 / BLAKE2b / RNG used by the gossip side of Node (swirld.py:10-12, 91-103).

Out of the accelerated path: the voting kernels consume the
signature only as opaque bytes.  Backed by libsodium through ctypes when a shared object
can be found (the reference uses pysodium, a ctypes wrapper of the same library);
otherwise by hashlib with a keyed-BLAKE2b "signature" that is NOT secure and only keeps
the single-process simulation (`test()`) runnable.
"""
import ctypes
import ctypes.util
import hashlib
import os

_sodium = None
for _cand in (os.environ.get("SWIRLD_LIBSODIUM"), ctypes.util.find_library("sodium"),
              "/opt/conda/lib/libsodium.so"):
    if not _cand:
        continue
    try:
        _l = ctypes.CDLL(_cand)
        if _l.sodium_init() >= 0:
            _sodium = _l
            break
    except OSError:
        pass

HAVE_SODIUM = _sodium is not None


def randombytes(n):
    return os.urandom(n)


def generichash(m):
    """BLAKE2b-256 == libsodium crypto_generichash with default arguments."""
    return hashlib.blake2b(m, digest_size=32).digest()


def sign_seed_keypair(seed):
    if HAVE_SODIUM:
        pk, sk = ctypes.create_string_buffer(32), ctypes.create_string_buffer(64)
        _sodium.crypto_sign_seed_keypair(pk, sk, seed)
        return pk.raw, sk.raw
    pk = hashlib.blake2b(b"swirld-pk" + seed, digest_size=32).digest()
    return pk, seed + pk


def sign_keypair():
    return sign_seed_keypair(randombytes(32))


def sign_detached(m, sk):
    if HAVE_SODIUM:
        sig = ctypes.create_string_buffer(64)
        _sodium.crypto_sign_detached(sig, None, m, ctypes.c_ulonglong(len(m)), sk)
        return sig.raw
    return hashlib.blake2b(m, digest_size=64, key=sk[32:]).digest()


def verify_detached(sig, m, pk):
    """Raises ValueError on a bad signature (what swirld.py:100 expects)."""
    if HAVE_SODIUM:
        if _sodium.crypto_sign_verify_detached(sig, m, ctypes.c_ulonglong(len(m)), pk) != 0:
            raise ValueError("invalid signature")
    elif hashlib.blake2b(m, digest_size=64, key=pk).digest() != sig:
        raise ValueError("invalid signature")


def sign(m, sk):
    return sign_detached(m, sk) + m


def sign_open(sm, pk):
    verify_detached(sm[:64], sm[64:], pk)
    return sm[64:]
