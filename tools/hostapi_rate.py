#!/usr/bin/env python3
"""tools/hostapi_rate.py -- throughput of the host-pointer (*_batch) entry points, the ones a C caller of the drop-in
API uses: pageable numpy arrays in, results out, PCIe + staging inclusive, against the device-resident (*_dev) rate.

    python tools/hostapi_rate.py [--n 1048576] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from curve25519_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--json", default=None)
args = ap.parse_args()
n = args.n
dev = torch.device("cuda", 0)
sk, pk = synth.x25519_inputs(n)
esk, msg = synth.ed25519_inputs(n)
api.curve25519_dh_CreateSharedKey(pk[:1024], sk[:1024])
pub, priv = api.ed25519_CreateKeyPair(esk)
sig = api.ed25519_SignMessage(priv, msg)


def host_rate(fn, reps=9):
    """median of `reps` individually timed calls: the boxes' host cores are shared, a mean carries the neighbours' bursts"""
    fn()
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2]


def host_rate_pair(fa, fb, reps=9):
    """the two calls alternating, median of each: for a ratio, both see the same minutes of the box"""
    fa(); fb()
    ta, tb = [], []
    for _ in range(reps):
        t = time.perf_counter(); fa(); ta.append(time.perf_counter() - t)
        t = time.perf_counter(); fb(); tb.append(time.perf_counter() - t)
    return sorted(ta)[len(ta) // 2], sorted(tb)[len(tb) // 2]


def dev_rate(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


d = {k: torch.from_numpy(v).to(dev) for k, v in dict(sk=sk, pk=pk, msg=msg, pub=pub, priv=priv, sig=sig).items()}
o32, o64 = torch.empty((n, 32), dtype=torch.uint8, device=dev), torch.empty((n, 64), dtype=torch.uint8, device=dev)
ok = torch.empty((n, 1), dtype=torch.int32, device=dev)
# the C entry points themselves, on caller-owned arrays that already exist (what a C program does); the numpy wrappers of
# curve25519_amd.api additionally allocate fresh result arrays (page faults) and copy sk, which is Python's cost
from curve25519_amd import _lib  # noqa: E402
L = _lib.load()
h32, h64, hok = np.zeros((n, 32), np.uint8), np.zeros((n, 64), np.uint8), np.zeros(n, np.int32)
P = lambda a: a.ctypes.data  # noqa: E731
# the same arrays page-locked by the caller (c25519_amd_host_register): *_batch then skips its staging copies
reg = {k: synth.page_aligned(v.shape, v.dtype, like=v)
       for k, v in dict(h32=h32, h64=h64, hok=hok, pk=pk, sk=sk, priv=priv, msg=msg, sig=sig, pub=pub).items()}
for v in reg.values():
    assert L.c25519_amd_host_register(P(v), synth.locked_bytes(v)) == 0
R = lambda k: reg[k].ctypes.data  # noqa: E731
rows = {}
for name, cfn, rfn, pfn, dfn, moved in (
        ("x25519", lambda: L.curve25519_dh_CreateSharedKey_batch(P(h32), P(pk), P(sk), n),
         lambda: L.curve25519_dh_CreateSharedKey_batch(R("h32"), R("pk"), R("sk"), n),
         lambda: api.curve25519_dh_CreateSharedKey(pk, sk),
         lambda: api.curve25519_dh_CreateSharedKey_dev(o32, d["pk"], d["sk"]), 128),
        ("sign", lambda: L.ed25519_SignMessage_batch(P(h64), P(priv), P(msg), 32, n),
         lambda: L.ed25519_SignMessage_batch(R("h64"), R("priv"), R("msg"), 32, n),
         lambda: api.ed25519_SignMessage(priv, msg),
         lambda: api.ed25519_SignMessage_dev(o64, d["priv"], d["msg"]), 160),
        ("verify", lambda: L.ed25519_VerifySignature_batch(P(hok), P(sig), P(pub), P(msg), 32, n),
         lambda: L.ed25519_VerifySignature_batch(R("hok"), R("sig"), R("pub"), R("msg"), 32, n),
         lambda: api.ed25519_VerifySignature(sig, pub, msg),
         lambda: api.ed25519_VerifySignature_dev(ok, d["sig"], d["pub"], d["msg"]), 132)):
    assert cfn() == 0 and rfn() == 0
    tc, tr, tp, td = host_rate(cfn), host_rate(rfn), host_rate(pfn), dev_rate(dfn)
    rows[name] = {"c_abi_ms": round(tc * 1e3, 3), "c_abi_Mops": round(n / tc / 1e6, 2),
                  "numpy_wrapper_ms": round(tp * 1e3, 3), "dev_ms": round(td * 1e3, 3),
                  "dev_Mops": round(n / td / 1e6, 2), "c_abi_over_dev": round(td / tc, 3),
                  "pcie_GBps": round(moved * n / tc / 1e9, 2), "registered_ms": round(tr * 1e3, 3),
                  "registered_Mops": round(n / tr / 1e6, 2), "registered_over_dev": round(td / tr, 3)}
    print(f"{name:7s} *_batch (host pointers) {tc * 1e3:8.2f} ms per 2^{int(np.log2(n))} = {n / tc / 1e6:7.1f} M ops/s "
          f"({moved * n / tc / 1e9:5.1f} GB/s moved, staging inclusive; numpy wrapper {tp * 1e3:7.2f} ms) | *_dev "
          f"{td * 1e3:7.2f} ms = {n / td / 1e6:7.1f} M ops/s | ratio {td / tc:.2f} | caller's arrays registered "
          f"{tr * 1e3:7.2f} ms = {n / tr / 1e6:7.1f} M ops/s, ratio {td / tr:.2f}")
# the multi-GPU C entry points over every device of the box (one here: the same shard / worker-thread / pinned-pipeline /
# gather / slab-download code as for eight), against the single-GPU *_batch call on the same pageable arrays
import ctypes as C  # noqa: E402
ndev = min(api.device_count(), 8)
m32, m64, mok = np.zeros((n, 32), np.uint8), np.zeros((n, 64), np.uint8), np.zeros(n, np.int32)
# four shapes of the same call: as shipped (a one-device handle skips the gather), the N > 1 path forced on the devices
# that exist (resident results, piece-wise grouped ncclGather, the root's drain and copy threads), the gather switched off
# (every device downloads its own rows: c25519_amd_multi_set_gather(h, 0)) -- and, on a one-GPU box, EIGHT VIRTUAL DEVICES
# on the one GPU (a device list naming it eight times: the 8-GPU code with device-to-device copies for the gather), both modes
shapes = [("as shipped", ndev, False, 1, "multi"), ("gather path forced", ndev, True, 1, "multi_forced_gather"),
          ("gather off", ndev, True, 0, "multi_gather_off")]
if ndev == 1:
    shapes += [("8 virtual, gather", 8, False, 1, "multi_virtual8"), ("8 virtual, no gather", 8, False, 0, "multi_virtual8_gather_off")]
for label, D, force, gather, key in shapes:
    mh = C.c_void_p()
    devs = list(range(ndev)) if D == ndev else [0] * D
    assert L.c25519_amd_multi_create(C.byref(mh), (C.c_int * D)(*devs), D) == 0
    _lib.set_tunable("MULTI_FORCE_GATHER", 1 if force else -1)
    assert L.c25519_amd_multi_set_gather(mh, gather) == 0
    for name, mfn, bfn in (("x25519", lambda: L.curve25519_dh_CreateSharedKey_multi(mh, P(m32), P(pk), P(sk), n),
                            lambda: L.curve25519_dh_CreateSharedKey_batch(P(h32), P(pk), P(sk), n)),
                           ("sign", lambda: L.ed25519_SignMessage_multi(mh, P(m64), P(priv), P(msg), 32, n),
                            lambda: L.ed25519_SignMessage_batch(P(h64), P(priv), P(msg), 32, n)),
                           ("verify", lambda: L.ed25519_VerifySignature_multi(mh, P(mok), P(sig), P(pub), P(msg), 32, n),
                            lambda: L.ed25519_VerifySignature_batch(P(hok), P(sig), P(pub), P(msg), 32, n))):
        m32[:] = 0; m64[:] = 0; mok[:] = -1
        assert mfn() == 0
        assert np.array_equal(m32 if name == "x25519" else m64 if name == "sign" else mok,
                              h32 if name == "x25519" else h64 if name == "sign" else hok), (label, name)
        tm, tb = host_rate_pair(mfn, bfn)                 # alternating calls, medians
        rows[name].update({"multi_devices": ndev, key + "_ms": round(tm * 1e3, 3), key + "_Mops": round(n / tm / 1e6, 2),
                           key + "_batch_ms_interleaved": round(tb * 1e3, 3), key + "_over_batch": round(tb / tm, 3),
                           key + "_helper_threads": int(L.c25519_amd_multi_helper_threads(mh))})
        print(f"{name:7s} *_multi over {D} device(s), {label:20s} {tm * 1e3:8.2f} ms = {n / tm / 1e6:7.1f} M ops/s | *_batch "
              f"alternating with it {tb * 1e3:8.2f} ms | ratio to *_batch {tb / tm:.2f} | helper threads "
              f"{L.c25519_amd_multi_helper_threads(mh)} of {L.c25519_amd_usable_cpus()} usable CPUs")
    L.c25519_amd_multi_destroy(mh)
_lib.set_tunable("MULTI_FORCE_GATHER", -1)
assert np.array_equal(hok, np.ones(n, np.int32)) and np.array_equal(reg["hok"], hok)
assert np.array_equal(reg["h32"], h32) and np.array_equal(reg["h64"], h64)
for v in reg.values():
    assert L.c25519_amd_host_unregister(P(v)) == 0
if args.json:
    json.dump({"n": n, "rows": rows}, open(args.json, "w"), indent=1)
