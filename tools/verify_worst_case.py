#!/usr/bin/env python3
"""tools/verify_worst_case.py -- ed25519_VerifySignature_dev when an adversary chooses the inputs.  An off-curve public
key cannot take the lattice path: its element goes on the slow list and the reference's operation order runs for it
behind the walk.  Four batches against the all-valid one: a single garbage key (the slow kernel's latency, one lane),
one per 256 elements, every second key, every key (the walk's waves then find nothing to do and leave).

    python tools/verify_worst_case.py [--n 1048576]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from curve25519_amd import _lib, api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
n = ap.parse_args().n
dev = torch.device("cuda", 0)
esk, msg = synth.ed25519_inputs(n)
pub, priv = api.ed25519_CreateKeyPair(esk)
sig = api.ed25519_SignMessage(priv, msg)
L = _lib.load()


def off_curve_key():
    import vectors
    for k in synth.random_bytes((64, 32), 0x999):
        if vectors.ed_decode(int.from_bytes(k.tobytes(), "little") & (2**255 - 1), 0) is None:
            return k
    raise SystemExit("no off-curve key among 64 random strings?")


def timed(pk_np, label):
    d = [torch.from_numpy(x).to(dev) for x in (sig, pk_np, msg)]
    ok = torch.empty((n, 1), dtype=torch.int32, device=dev)
    run = lambda: api.ed25519_VerifySignature_dev(ok, d[0], d[1], d[2])
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"{label:46s} {ms:8.3f} ms per 2^{n.bit_length() - 1}  = {n / ms / 1e3:7.1f} M verifies/s   "
          f"accepted {int(ok.sum())}  reference-order elements {L.c25519_amd_verify_last_slow_elements()} of {n}")
    return ms


base = timed(pub, "all keys on the curve (lattice path)")
off = off_curve_key()
worst = base
for label, sel in (("ONE off-curve key in the batch", slice(12345, 12346)), ("one off-curve key per 256 elements", slice(128, None, 256)),
                   ("every second key off the curve", slice(0, None, 2)), ("every key off the curve", slice(0, None))):
    bad = pub.copy()
    bad[sel] = off
    worst = max(worst, timed(bad, label))
print(f"worst case / normal = {worst / base:.2f}")
