"""CPU suite, part 4: the N > 1 path.  world_size-2, -3 and -8 gloo groups (8 = the node the scaling bench runs on: n = 8k + 5,
a 2^23-shaped mixed split with equal thirds, and more ranks than elements) run the sharding + gather layer of
curve25519_amd/sharded.py with an oracle-backed engine standing in for the HIP engine (tests may use the
oracle; the product never does).  Checks: contiguous shards, one gather to the root, uneven shard sizes,
empty shards, result == the unsharded oracle result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """CPU stand-in with the HipEngine interface (torch uint8 tensors in, torch tensors out)."""

    def __init__(self):
        from oracle_lib import Oracle
        self.o = Oracle()

    def x25519_shared(self, pk, sk):
        out, clamped = self.o.x25519_shared(pk.numpy(), sk.numpy())
        sk.copy_(torch.from_numpy(clamped))                     # the device engine clamps sk in place
        return torch.from_numpy(out)

    def ed25519_sign(self, priv, msg):
        return torch.from_numpy(self.o.ed25519_sign(priv.numpy(), msg.numpy()))

    def ed25519_verify(self, sig, pk, msg):
        return torch.from_numpy(self.o.ed25519_verify(sig.numpy(), pk.numpy(), msg.numpy())).reshape(-1, 1)


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from curve25519_amd import sharded, synth
        eng = OracleEngine()
        lo, hi = sharded.shard_bounds(n, world)[rank]
        sk, pk = synth.x25519_inputs(n)
        sk_l, pk_l = torch.from_numpy(sk[lo:hi].copy()), torch.from_numpy(pk[lo:hi].copy())
        shared = sharded.x25519_shared_sharded(eng, pk_l, sk_l, root=0)

        esk, msg = synth.ed25519_inputs(n)
        pub, priv = eng.o.ed25519_keypair(esk[lo:hi])
        sig = sharded.ed25519_sign_sharded(eng, torch.from_numpy(priv), torch.from_numpy(msg[lo:hi].copy()), root=0)
        sig_l = torch.from_numpy(eng.o.ed25519_sign(priv, msg[lo:hi]))
        ok = sharded.ed25519_verify_sharded(eng, sig_l, torch.from_numpy(pub), torch.from_numpy(msg[lo:hi].copy()), root=0)
        # the steady-state overlapped gather used by bench.py: 3 batches through 2 alternating buffers
        rows = max(hi - lo for lo, hi in sharded.shard_bounds(n, world))
        og = sharded.OverlappedGather(rows, 32, torch.device("cpu"), root=0)
        seen = []
        for stepno in range(3):
            buf = og.next_buffer()
            buf.zero_()
            buf[: hi - lo] = torch.from_numpy(eng.o.x25519_shared(pk[lo:hi], sk[lo:hi])[0]) if hi > lo else buf[:0]
            buf[:, 0] ^= stepno                                  # make every batch distinguishable
            og.submit()
            if stepno >= 1:                                        # previous buffer's gather overlaps this batch
                pass
        og.finish()
        if rank == 0:
            seen = [og.gathered(b).numpy().copy() for b in (0, 1)]
        # BASELINE.json configs[4]: mixed batch, contiguous thirds, every third sharded over the ranks
        (xa, xb), (sa, sb), (va, vb) = sharded.mixed_thirds(n)
        cut = lambda lo_, hi_: sharded.shard_bounds(hi_ - lo_, world)[rank]  # noqa: E731
        t = lambda a_: torch.from_numpy(np.ascontiguousarray(a_))  # noqa: E731
        fpub, fpriv = eng.o.ed25519_keypair(esk)
        fsig = eng.o.ed25519_sign(fpriv, msg)
        l0, h0 = cut(xa, xb); l1, h1 = cut(sa, sb); l2, h2 = cut(va, vb)
        mixed = sharded.mixed_sharded(eng, t(pk[xa + l0:xa + h0]), t(sk[xa + l0:xa + h0].copy()),
                                      t(fpriv[sa + l1:sa + h1]), t(msg[sa + l1:sa + h1]),
                                      t(fsig[va + l2:va + h2]), t(fpub[va + l2:va + h2]), t(msg[va + l2:va + h2]))
        if rank == 0:
            seen.append([m.numpy() for m in mixed])
        if rank == 0:
            q.put((shared.numpy(), sig.numpy(), ok.numpy(), seen, rows))
        else:
            assert shared is None and sig is None and ok is None
            q.put(None)
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n", [(2, 96), (2, 7), (3, 10), (2, 1), (8, 45), (8, 48), (8, 5)])
def test_sharded_matches_unsharded(world, n, oracle):
    from curve25519_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shared, sig, ok, seen, rows = next(r for r in results if r is not None)
    sk, pk = synth.x25519_inputs(n)
    exp_shared, _ = oracle.x25519_shared(pk, sk)
    assert np.array_equal(shared, exp_shared)
    esk, msg = synth.ed25519_inputs(n)
    pub, priv = oracle.ed25519_keypair(esk)
    assert np.array_equal(sig, oracle.ed25519_sign(priv, msg))
    assert ok.shape == (n, 1) and ok.all()
    # buffer 0 last carried batch 2, buffer 1 carried batch 1; rows are rank-major, zero padded to `rows`
    from curve25519_amd.sharded import shard_bounds
    (xa, xb), (sa, sb), (va, vb) = __import__("curve25519_amd.sharded", fromlist=["x"]).mixed_thirds(n)
    m_shared, m_sig, m_ok = seen[2]
    assert np.array_equal(m_shared, exp_shared[xa:xb])
    assert np.array_equal(m_sig, oracle.ed25519_sign(priv, msg)[sa:sb])
    assert m_ok.shape == (vb - va, 1) and m_ok.all()
    for b, stepno in ((0, 2), (1, 1)):
        got = seen[b].reshape(world, rows, 32)
        for r, (lo, hi) in enumerate(shard_bounds(n, world)):
            exp = exp_shared[lo:hi].copy()
            exp[:, 0] ^= stepno
            assert np.array_equal(got[r, : hi - lo], exp), (b, r)


def test_shard_bounds_cover_exactly():
    from curve25519_amd.sharded import shard_bounds
    for n in (0, 1, 7, 8, 1 << 20, (1 << 23) + 5):
        for world in (1, 2, 3, 8):
            b = shard_bounds(n, world)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
