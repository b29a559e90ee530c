"""Deterministic synthetic inputs for the BASELINE.json configs.

One splitmix64 stream per array, little-endian bytes; the same stream is implemented in C by
oracle/orc_sc.c:orc_fill_random so fixtures, the CPU baseline and the GPU path all see identical
bytes.  Pure numpy -- no dependency on the oracle or on the HIP library.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)

# fixed seeds of the benchmark configurations
SEED_X25519_SK = 0x5eed0001
SEED_X25519_PK = 0x5eed0002
SEED_ED_SK = 0x5eed0003
SEED_ED_MSG = 0x5eed0004
SEED_ED_CORRUPT = 0x5eed0005


def random_bytes(shape, seed: int) -> np.ndarray:
    """uint8 array of `shape` filled from splitmix64(seed)."""
    n = int(np.prod(shape))
    words = (n + 7) // 8
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + GOLDEN * np.arange(1, words + 1, dtype=np.uint64)
        z = s
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.astype("<u8").view(np.uint8)[:n].reshape(shape).copy()


RANK_SEED_STRIDE = 0x100     # bench.py: rank r of a world > 1 draws its own 2^20 elements from seed + 0x100 * r


def rank_seed_shift(rank: int, world: int) -> int:
    """The seed offset of rank `rank`'s inputs (weak scaling: every rank owns its own batch).  A world of one is rank 0
    of any world: the committed digests of tests/golden/digests.json["ranks"][r] belong to shift 0x100 * r."""
    return RANK_SEED_STRIDE * rank if world > 1 else 0


def x25519_inputs(n: int, seed_shift: int = 0):
    """(sk, pk): n x 32 uniform bytes each.  pk bit 255 is deliberately NOT masked."""
    return random_bytes((n, 32), SEED_X25519_SK + seed_shift), random_bytes((n, 32), SEED_X25519_PK + seed_shift)


def ed25519_inputs(n: int, msg_size: int = 32, seed_shift: int = 0):
    """(sk32, msg): secret seeds and fixed-length messages."""
    return random_bytes((n, 32), SEED_ED_SK + seed_shift), random_bytes((n, msg_size), SEED_ED_MSG + seed_shift)


def mixed_thirds(n: int):
    """Element i of a mixed batch (BASELINE.json configs[4]) is X25519 / sign / verify by contiguous thirds (the split
    SURVEY.md 8(d) allows; stated here so fixtures and ranks agree): [0, a) X25519, [a, b) sign, [b, n) verify."""
    a, b = n // 3, 2 * (n // 3)
    return (0, a), (a, b), (b, n)


def corrupt_for_verify(sig: np.ndarray, msg: np.ndarray):
    """Config 4's 1/64 sprinkle of bad entries: returns (sig', msg', expected_bad_mask).

    Entry i is corrupted when the low 6 bits of its control byte are zero; bits 6..7 choose what
    gets one bit flipped: 0/3 -> message, 1 -> R half of the signature, 2 -> S half.
    """
    n = sig.shape[0]
    ctl = random_bytes((n, 4), SEED_ED_CORRUPT)
    bad = (ctl[:, 0] & 63) == 0
    which = ctl[:, 0] >> 6
    pos = ctl[:, 1].astype(np.int64)
    bit = (np.uint8(1) << (ctl[:, 2] & 7)).astype(np.uint8)
    sig = sig.copy()
    msg = msg.copy()
    idx = np.nonzero(bad)[0]
    for i in idx:
        if which[i] == 1:
            sig[i, pos[i] % 32] ^= bit[i]
        elif which[i] == 2:
            # keep S's top 3 bits alone so the flipped S stays a different residue mod L
            sig[i, 32 + pos[i] % 31] ^= bit[i]
        else:
            msg[i, pos[i] % msg.shape[1]] ^= bit[i]
    return sig, msg, bad


def page_aligned(shape, dtype=np.uint8, like=None) -> np.ndarray:
    """A zeroed array in pages of its own (an anonymous mmap): what c25519_amd_host_register() wants -- page locking
    works on whole pages, so the buffer must not share one with anything else.  `like`: initial contents."""
    import mmap
    dtype = np.dtype(dtype)
    count = int(np.prod(shape))
    size = max(1, count * dtype.itemsize)
    size = (size + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE
    buf = mmap.mmap(-1, size)
    arr = np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)
    if like is not None:
        arr[...] = like
    return arr


def locked_bytes(arr: np.ndarray) -> int:
    """Length to pass to c25519_amd_host_register for an array made by page_aligned(): its whole pages."""
    import mmap
    return (arr.nbytes + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE
