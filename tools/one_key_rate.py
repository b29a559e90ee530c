#!/usr/bin/env python3
"""tools/one_key_rate.py -- one ed25519_Verify_Init, many ed25519_Verify_Check calls of a few thousand pairs (the reference's
two-phase use, ed25519_verify.c:282-286): the comb a first call of >= 2^16 pairs built for the key is
remembered by the calling thread, and every later call with that context walks the two wide combs at any size.  Per size:
device-resident calls (ed25519_Verify_Check_dev, HIP events) and host-pointer calls (ed25519_Verify_Check_batch, wall clock),
with the remembered comb and with the comb path turned off (tunable ONE_KEY_WIDE = 0: the reference-order kernel, what every
call below 2^16 pairs ran before)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from curve25519_amd import _lib, api, synth  # noqa: E402

L = _lib.load()
dev = torch.device("cuda", 0)
big = 1 << 16
sk = synth.random_bytes((1, 32), 0x7b03)
pub, priv = api.ed25519_CreateKeyPair(sk)
ctx = api.ed25519_Verify_Init(pub)
msg = synth.random_bytes((big, 32), 0x7b04)
sig = api.ed25519_SignMessage(np.repeat(priv, big, axis=0), msg)
d_ctx, d_sig, d_msg = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (ctx[0], sig, msg))
d_ok = torch.empty(big, dtype=torch.int32, device=dev)
assert api.ed25519_Verify_Check(ctx[0], sig, msg).all()                 # builds and remembers the key's comb
assert L.c25519_amd_verify_check_last_wide() == 1


def dev_ms(n, reps=30):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    f = lambda: _lib.check(L.ed25519_Verify_Check_dev(p(d_ok), p(d_ctx), p(d_sig), p(d_msg), 32, n, st), "ed25519_Verify_Check_dev")  # noqa: E731
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    assert bool(d_ok[:n].all())
    return best


def host_ms(n, reps=30):
    f = lambda: api.ed25519_Verify_Check(ctx[0], sig[:n], msg[:n])  # noqa: E731
    for _ in range(5):
        f()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); ok = f(); best = min(best, (time.perf_counter() - t) * 1e3)
    assert ok.all()
    return best


print(f"# tools/one_key_rate.py on {torch.cuda.get_device_name(0)}: ed25519_Verify_Check with ONE context, 32-byte messages; ms per call | M pairs/s")
print(f"{'pairs':>8} {'remembered comb, _dev':>26} {'(one lane per pair)':>20} {'reference order, _dev':>26} {'x':>5} {'remembered comb, _batch':>28} {'reference order, _batch':>28} {'x':>5}")
for lg in (11, 12, 13, 14, 15, 16):
    n = 1 << lg
    w_dev, w_host = dev_ms(n), host_ms(n)
    assert L.c25519_amd_verify_check_last_wide() == 1, n
    with _lib.tunable("QUAD_MAX", 0):                                   # the combs walked by one lane per pair (what 2^15 pairs and more run)
        l_dev = dev_ms(n)
    with _lib.tunable("ONE_KEY_WIDE", 0):
        r_dev, r_host = dev_ms(n), host_ms(n)
    cell = lambda ms: f"{ms:8.3f} ms {n / ms / 1e3:9.1f} M/s"  # noqa: E731
    print(f"{'2^' + str(lg):>8} {cell(w_dev):>26} {l_dev:17.3f} ms {cell(r_dev):>26} {r_dev / w_dev:5.2f} {cell(w_host):>28} {cell(r_host):>28} {r_host / w_host:5.2f}")
