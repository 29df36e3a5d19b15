#!/usr/bin/env python3
"""N3: throughput of the crypto batch kernels (one GPU thread per message) next to libsodium on one host core."""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
pkg = importlib.import_module("py-swirld_amd")
from test_crypto_host import load_sodium
sod = load_sodium()
import random
rng = random.Random(3)
keys = []
for _ in range(64):
    pk, sk = C.create_string_buffer(32), C.create_string_buffer(64)
    sod.crypto_sign_seed_keypair(pk, sk, bytes(rng.getrandbits(8) for _ in range(32)))
    keys.append((pk.raw, sk.raw))
for K in (1, 64, 4096, 65536, 262144):
    msgs, sigs, pks = [], [], []
    base = bytes(rng.getrandbits(8) for _ in range(180))      # a pickled event is ~180 bytes
    for i in range(K):
        pk, sk = keys[i % 64]
        m = base + i.to_bytes(8, "little")
        sig = C.create_string_buffer(64)
        sod.crypto_sign_detached(sig, None, m, C.c_ulonglong(len(m)), sk)
        msgs.append(m); sigs.append(sig.raw); pks.append(pk)
    pkg.verify_batch(msgs[:1], sigs[:1], pks[:1])
    t0 = time.perf_counter(); ok = pkg.verify_batch(msgs, sigs, pks); t1 = time.perf_counter()
    ids = pkg.hash_batch(msgs); t2 = time.perf_counter()
    assert ok.all()
    M = min(K, 2000)
    t3 = time.perf_counter()
    for i in range(M):
        assert sod.crypto_sign_verify_detached(sigs[i], msgs[i], C.c_ulonglong(len(msgs[i])), pks[i]) == 0
    t4 = time.perf_counter()
    print("K=%7d  GPU verify %8.2f ms (%9.0f /s)   GPU BLAKE2b %7.2f ms (%10.0f /s)   libsodium 1 core %9.0f verifies/s" % (
        K, (t1 - t0) * 1e3, K / (t1 - t0), (t2 - t1) * 1e3, K / (t2 - t1), M / (t4 - t3)), flush=True)
