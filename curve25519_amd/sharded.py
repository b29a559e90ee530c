"""Multi-GPU layer: one process per GPU, contiguous shards, one gather of the result slab to the root.

The path is embarrassingly parallel: element i of a batch never looks at element j,
the only shared data is the read-only 24 KiB base table, which every GPU regenerates for itself.  So
there is no data-path collective at all; the single exchange step is the `gather` of each rank's
output rows to the root that BASELINE.json's north_star names (RCCL over xGMI when the backend is
"nccl"; the same code runs over gloo on CPU tensors in the tests).

`engine` is any object with the batch methods used below.  The product engine is `HipEngine`
(device tensors, libcurve25519_amd.so); tests/ substitute an oracle-backed engine to exercise the
sharding and gather logic without a GPU.
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous index ranges [lo, hi) of a batch of n over `world` ranks (sizes differ by <= 1)."""
    return [(n * r // world, n * (r + 1) // world) for r in range(world)]


def gather_rows(local: torch.Tensor, root: int = 0, group=None, always: bool = False) -> Optional[torch.Tensor]:
    """Gather each rank's [m_r, w] rows to `root` in rank order (one collective; rows padded to the
    largest shard so that every peer sends the same count).  Returns the concatenation on root, None
    elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not always:
        return local
    m = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(m) for _ in range(world)]
    dist.all_gather(counts, m, group=group)           # tiny control message, not the data path
    counts = [int(c.item()) for c in counts]
    mmax = max(counts)
    send = local
    if local.shape[0] != mmax:
        send = torch.zeros((mmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    recv = [torch.empty_like(send) for _ in range(world)] if rank == root else None
    dist.gather(send, recv, dst=root, group=group)
    if rank != root:
        return None
    return torch.cat([r[:c] for r, c in zip(recv, counts)], dim=0)


class OverlappedGather:
    """The steady-state form of the one gather: equal-size shards, receive buffers allocated once, and the
    collective issued asynchronously so that it runs (on RCCL's stream, over xGMI) underneath the NEXT
    batch's kernels -- those are VALU-bound and leave HBM and the links idle.  Two result buffers alternate;
    a buffer is only reused after the gather that reads it has been waited for (stream-level wait)."""

    def __init__(self, rows: int, width: int, device, root: int = 0, group=None, dtype=torch.uint8):
        self.root, self.group = root, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.out = [torch.empty((rows, width), dtype=dtype, device=device) for _ in range(2)]
        # a gloo group (the launcher self-test, where several ranks share one GPU and RCCL would refuse the duplicate
        # device) cannot move device memory: the rows are staged through host copies; RCCL gathers device buffers
        self.host_staged = dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"
        wire = "cpu" if self.host_staged else device
        self.wire = [torch.empty((rows, width), dtype=dtype, device=wire) for _ in range(2)] if self.host_staged else self.out
        self.recv = [[torch.empty((rows, width), dtype=dtype, device=wire) for _ in range(self.world)]
                     for _ in range(2)] if self.rank == root else [None, None]
        self.work = [None, None]
        self.step = 0

    def next_buffer(self) -> torch.Tensor:
        """Result buffer for the next batch (waits, on the current stream, for its previous gather)."""
        b = self.step & 1
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        return self.out[b]

    def submit(self):
        """Start gathering the buffer handed out by the last next_buffer() call."""
        b = self.step & 1
        if self.host_staged:
            self.wire[b].copy_(self.out[b])
        self.work[b] = dist.gather(self.wire[b], self.recv[b], dst=self.root, group=self.group, async_op=True)
        self.step += 1

    def finish(self):
        for b in range(2):
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None

    def gathered(self, buffer_index: int) -> Optional[torch.Tensor]:
        """On root: the [world*rows, width] result of the given buffer's last gather."""
        return torch.cat(self.recv[buffer_index], dim=0) if self.rank == self.root else None


class HipEngine:
    """Device-resident batches on the current CUDA device, asynchronous on torch's current stream."""

    def __init__(self, device=None):
        from . import api
        self.api = api
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if device.index is None:                        # "cuda" -> the current device's ordinal, so that tensor.device
            device = torch.device("cuda", torch.cuda.current_device())   # comparisons in api.py are exact
        self.device = device

    def _empty(self, n, w, dtype=torch.uint8):
        return torch.empty((n, w), dtype=dtype, device=self.device)

    def x25519_shared(self, pk, sk):
        out = self._empty(pk.shape[0], 32)
        self.api.curve25519_dh_CreateSharedKey_dev(out, pk, sk)
        return out

    def ed25519_keypair(self, sk):
        pub, priv = self._empty(sk.shape[0], 32), self._empty(sk.shape[0], 64)
        self.api.ed25519_CreateKeyPair_dev(pub, priv, sk)
        return pub, priv

    def ed25519_sign(self, priv, msg):
        sig = self._empty(priv.shape[0], 64)
        self.api.ed25519_SignMessage_dev(sig, priv, msg)
        return sig

    def ed25519_verify(self, sig, pk, msg):
        ok = torch.empty((sig.shape[0], 1), dtype=torch.int32, device=self.device)
        self.api.ed25519_VerifySignature_dev(ok, sig, pk, msg)
        return ok

    def mixed(self, x_pk, x_sk, s_priv, s_msg, v_sig, v_pk, v_msg):
        """The three sub-batches of a mixed batch, each on a stream of its own (the library keeps a work scratch per
        stream, so they overlap on the device: one operation's last round of workgroups fills up with the next
        operation's -- 6.9 instead of 7.8 ms per 2^20, profiles/r03_split_streams.txt).  The side streams start behind
        the current stream's work and the current stream continues behind theirs."""
        if getattr(self, "_side", None) is None:
            self._side = [torch.cuda.Stream(self.device) for _ in range(3)]
        cur = torch.cuda.current_stream(self.device)
        calls = ((self.x25519_shared, (x_pk, x_sk)), (self.ed25519_sign, (s_priv, s_msg)),
                 (self.ed25519_verify, (v_sig, v_pk, v_msg)))
        outs = []
        for st, (fn, a) in zip(self._side, calls):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(fn(*a))
            for t in a + (outs[-1],):
                t.record_stream(st)                    # torch's allocator: these blocks are in use on st
        for st in self._side:
            cur.wait_stream(st)
        for o in outs:
            o.record_stream(cur)                       # allocated on a side stream, consumed on `cur` (the gathers)
        return tuple(outs)


def x25519_shared_sharded(engine, pk_local, sk_local, root: int = 0, group=None):
    """Each rank computes its shard; the shared secrets are gathered to `root`."""
    return gather_rows(engine.x25519_shared(pk_local, sk_local), root, group)


def ed25519_sign_sharded(engine, priv_local, msg_local, root: int = 0, group=None):
    return gather_rows(engine.ed25519_sign(priv_local, msg_local), root, group)


def ed25519_verify_sharded(engine, sig_local, pk_local, msg_local, root: int = 0, group=None):
    return gather_rows(engine.ed25519_verify(sig_local, pk_local, msg_local), root, group)


# ---- BASELINE.json configs[4]: a mixed X25519 + Ed25519 batch sharded over the ranks -----------------------
def mixed_thirds(n: int) -> Tuple[Tuple[int, int], Tuple[int, int], Tuple[int, int]]:
    """[0, a) X25519, [a, b) sign, [b, n) verify -- synth.mixed_thirds (one definition for fixtures and ranks)."""
    from .synth import mixed_thirds as _thirds
    return _thirds(n)


def mixed_sharded(engine, x_pk, x_sk, s_priv, s_msg, v_sig, v_pk, v_msg, root: int = 0, group=None):
    """Each rank holds its contiguous shard of the three sub-batches; results are gathered to `root` with one
    gather per output type (32-byte secrets, 64-byte signatures, int32 verdicts).  Returns the three
    gathered tensors on root, (None, None, None) elsewhere."""
    if hasattr(engine, "mixed"):                       # the HIP engine: one stream per sub-batch
        shared, sig, ok = engine.mixed(x_pk, x_sk, s_priv, s_msg, v_sig, v_pk, v_msg)
    else:
        shared = engine.x25519_shared(x_pk, x_sk)
        sig = engine.ed25519_sign(s_priv, s_msg)
        ok = engine.ed25519_verify(v_sig, v_pk, v_msg)
    return (gather_rows(shared, root, group), gather_rows(sig, root, group), gather_rows(ok, root, group))
