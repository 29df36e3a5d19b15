"""GPU (-m gpu): the crypto batch kernels (sw_crypto_verify_batch, sw_crypto_hash_batch; SURVEY.md §8f
N3) against libsodium / hashlib on the vectors of tests/test_crypto_host.py — valid signatures,
corrupted ones, non-canonical and small-order encodings — and a whole gossip simulation whose
sync payloads are validated on the device."""
import contextlib
import hashlib
import io
import random

import numpy as np
import pytest

from test_crypto_host import load_sodium, signed_cases, sodium_verify

pytestmark = pytest.mark.gpu


def test_batches_match_libsodium(pkg):
    sod = load_sodium()
    assert sod is not None, "libsodium is part of the image"
    rng = random.Random(2)
    cases = signed_cases(sod, rng, 300)
    exp = np.array([sodium_verify(sod, s, m, p) for s, m, p in cases])
    got = pkg.verify_batch([m for _, m, _ in cases], [s for s, _, _ in cases], [p for _, _, p in cases])
    assert np.array_equal(got, exp)
    assert exp.sum() >= 300 and (~exp).sum() > 2000
    msgs = [bytes(rng.getrandbits(8) for _ in range(n)) for n in list(range(0, 300)) + [511, 512, 513, 4096]]
    assert pkg.hash_batch(msgs) == [hashlib.blake2b(m, digest_size=32).digest() for m in msgs]
    assert pkg.verify_batch([], [], []).shape == (0,) and pkg.hash_batch([]) == []


def test_node_sync_with_device_crypto(pkg):
    """Every sync payload validated by the device batches (threshold 1): the simulation must run as with
    libsodium on the host, and a tampered event must still be rejected."""
    node_mod = pkg.node
    rng = random.Random(20260923)
    orig_rb, orig_time, orig_thr = node_mod.crypto.randombytes, node_mod.time, node_mod.Node.device_crypto_threshold
    node_mod.crypto.randombytes = lambda k: bytes(rng.getrandbits(8) for _ in range(k))
    clock = iter(range(1, 1 << 30))
    node_mod.time = lambda: 1.0e9 + 0.001 * next(clock)
    results = {}
    try:
        for thr in (None, 1):
            rng.seed(20260923)
            clock = iter(range(1, 1 << 30))
            node_mod.Node.device_crypto_threshold = thr
            with contextlib.redirect_stdout(io.StringIO()):
                nodes = pkg.test(4, 120)
            results[thr] = [(len(nd._ids), sorted(nd.consensus), len(nd.transactions)) for nd in nodes]
            nd = nodes[0]
        assert results[None] == results[1], "device-validated gossip reproduces host-validated gossip"
        # a tampered payload: flip one bit of one signature
        other = nodes[1]
        h = other._ids[-1]
        ev = other.hg[h]
        bad = ev._replace(s=bytes([ev.s[0] ^ 1]) + ev.s[1:])
        pre = nd._batch_crypto([h], {h: bad})
        assert pre[h][0] is False
        pre = nd._batch_crypto([h], {h: ev})
        assert pre[h] == (True, h)
    finally:
        node_mod.crypto.randombytes, node_mod.time, node_mod.Node.device_crypto_threshold = orig_rb, orig_time, orig_thr
