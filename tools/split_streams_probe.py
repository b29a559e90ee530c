#!/usr/bin/env python3
"""tools/split_streams_probe.py -- does a pass gain from being issued as K sub-batches on K streams (one operation's draining
round of workgroups overlapping the next sub-batch's kernels)?  Device-resident inputs, 2^20 elements, best / median of 7."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from curve25519_amd import api, synth  # noqa: E402

n = 1 << 20
dev = torch.device("cuda", 0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
sk, pk = (up(a) for a in synth.x25519_inputs(n))
esk, msg = synth.ed25519_inputs(n)
pub, priv = api.ed25519_CreateKeyPair(esk)
sig = api.ed25519_SignMessage(priv, msg)
pub, priv, sig, msg = up(pub), up(priv), up(sig), up(msg)
o32 = torch.empty((n, 32), dtype=torch.uint8, device=dev)
o64 = torch.empty((n, 64), dtype=torch.uint8, device=dev)
ok = torch.empty((n, 1), dtype=torch.int32, device=dev)
ops = {
    "x25519": lambda a, b: api.curve25519_dh_CreateSharedKey_dev(o32[a:b], pk[a:b], sk[a:b]),
    "sign": lambda a, b: api.ed25519_SignMessage_dev(o64[a:b], priv[a:b], msg[a:b]),
    "verify": lambda a, b: api.ed25519_VerifySignature_dev(ok[a:b], sig[a:b], pub[a:b], msg[a:b]),
}
streams = [torch.cuda.Stream(dev) for _ in range(4)]
for name, fn in ops.items():
    for k in (1, 2, 3, 4):
        cuts = [(n * j // k) // 1024 * 1024 for j in range(k)] + [n]
        ts = []
        for rep in range(9):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for j in range(k):
                with torch.cuda.stream(streams[j]):
                    fn(cuts[j], cuts[j + 1])
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        ts = sorted(ts[2:])
        print(f"{name:7s} {k} stream(s): min {ts[0] * 1e3:7.3f} ms  median {ts[len(ts) // 2] * 1e3:7.3f} ms", flush=True)
    if name == "verify":
        assert bool((ok == 1).all())
