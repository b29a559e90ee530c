"""CPU suite, part 3: the seeded input generator shared by fixtures, CPU baseline and GPU path."""
import numpy as np

from curve25519_amd import synth


def test_matches_the_c_generator(oracle):
    for shape, seed in (((5, 32), 1), ((4096, 32), synth.SEED_X25519_SK), ((3,), 7), ((1, 64), 9), ((0, 32), 3)):
        assert np.array_equal(synth.random_bytes(shape, seed), oracle.random_bytes(shape, seed)), shape


def test_streams_are_prefix_stable_and_distinct():
    a, b = synth.random_bytes((16, 32), 5), synth.random_bytes((64, 32), 5)
    assert np.array_equal(a, b[:16])
    sk, pk = synth.x25519_inputs(64)
    assert not np.array_equal(sk, pk)
    assert (pk[:, 31] >> 7).any(), "peer keys must exercise bit 255 (the reference does not mask it)"


def test_corruption_pattern():
    n = 4096
    sig, msg = synth.random_bytes((n, 64), 21), synth.random_bytes((n, 32), 22)
    bsig, bmsg, bad = synth.corrupt_for_verify(sig, msg)
    changed = (bsig != sig).any(axis=1) | (bmsg != msg).any(axis=1)
    assert np.array_equal(changed, bad)
    assert 20 <= int(bad.sum()) <= 120                       # ~1/64
    diff_bits = np.unpackbits(bsig ^ sig, axis=1).sum(axis=1) + np.unpackbits(bmsg ^ msg, axis=1).sum(axis=1)
    assert set(np.unique(diff_bits)) <= {0, 1}
    again = synth.corrupt_for_verify(sig, msg)
    assert np.array_equal(again[0], bsig) and np.array_equal(again[1], bmsg)
