#!/usr/bin/env python3
"""Within-process interleaved A/B timing of engine builds (kernel variants compiled to different .so files).

    python tools/ab_bench.py build_a.so build_b.so lib.so@KEY=VAL,KEY2=VAL2 ...   [--ops x25519,sign,verify,keypair] [--rounds 5]

`lib.so@KEY=VAL` times a library under run-time knobs (the library's tunables, include/curve25519_amd.h:
c25519_amd_tunable_set -- INV_K, XF_SPLIT, BASE_COMB, ...; a C25519_AMD_ prefix is accepted and dropped): the knobs are
set in that library's own table right before its launches.

Each library is dlopen'ed privately; every round runs each (library, op) once at N = 2^20 with inputs
resident in HBM and reports min / median kernel time from HIP events on torch's current stream."""
import argparse
import ctypes as C
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from curve25519_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--ops", default="x25519,sign,verify")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--n", type=int, default=1 << 20)
args = ap.parse_args()
n = args.n
dev = torch.device("cuda", 0)
vp, sz = C.c_void_p, C.c_size_t
libs = []
envs = {}
for spec in args.libs:
    p, _, knobs = spec.partition("@")
    L = C.CDLL(os.path.abspath(p))
    L.curve25519_dh_CreateSharedKey_dev.argtypes = [vp, vp, vp, sz, vp]
    L.ed25519_CreateKeyPair_dev.argtypes = [vp, vp, vp, sz, vp]
    L.ed25519_SignMessage_dev.argtypes = [vp, vp, vp, sz, sz, vp]
    L.ed25519_VerifySignature_dev.argtypes = [vp, vp, vp, vp, sz, sz, vp]
    name = os.path.basename(p) + ("@" + knobs if knobs else "")
    envs[name] = dict(kv.split("=", 1) for kv in knobs.split(",")) if knobs else {}
    libs.append((name, L))
KNOBS = sorted({k for e in envs.values() for k in e})


LIB_OF = dict(libs)


def set_env(name):
    L = LIB_OF[name]
    if not hasattr(L, "c25519_amd_tunable_set"):
        assert not envs[name], f"{name}: this build has no tunable table"
        return
    L.c25519_amd_tunable_set.argtypes = [C.c_char_p, C.c_long]
    for k in KNOBS:
        L.c25519_amd_tunable_set(k.replace("C25519_AMD_", "").encode(), -1)
    for k, v in envs[name].items():
        assert L.c25519_amd_tunable_set(k.replace("C25519_AMD_", "").encode(), int(v)) == 0, (name, k)


sk_np, pk_np = synth.x25519_inputs(n)
esk_np, msg_np = synth.ed25519_inputs(n)
sk, pk = torch.from_numpy(sk_np).to(dev), torch.from_numpy(pk_np).to(dev)
esk, msg = torch.from_numpy(esk_np).to(dev), torch.from_numpy(msg_np).to(dev)
out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
pub = torch.empty((n, 32), dtype=torch.uint8, device=dev)
priv = torch.empty((n, 64), dtype=torch.uint8, device=dev)
sig = torch.empty((n, 64), dtype=torch.uint8, device=dev)
ok = torch.empty((n,), dtype=torch.int32, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


def run(L, op):
    if op == "x25519":
        rc = L.curve25519_dh_CreateSharedKey_dev(p(out), p(pk), p(sk), n, st())
    elif op == "keypair":
        rc = L.ed25519_CreateKeyPair_dev(p(pub), p(priv), p(esk), n, st())
    elif op == "sign":
        rc = L.ed25519_SignMessage_dev(p(sig), p(priv), p(msg), 32, n, st())
    elif op == "verify":
        rc = L.ed25519_VerifySignature_dev(p(ok), p(sig), p(pub), p(msg), 32, n, st())
    assert rc == 0, (op, rc)


libs[0][1].ed25519_CreateKeyPair_dev(p(pub), p(priv), p(esk), n, st())
libs[0][1].ed25519_SignMessage_dev(p(sig), p(priv), p(msg), 32, n, st())
torch.cuda.synchronize()
ops = args.ops.split(",")
BURST = 4
times = {(name, op): [] for name, _ in libs for op in ops}
ref_out = {}
for r in range(args.rounds + 1):
    for op in ops:
        for name, L in libs:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            set_env(name)
            run(L, op)                                  # warm-up: the chip needs 20-40 ms of load after an idle gap to reach
            torch.cuda.synchronize()                    # its sustained clock (profiles/r04_warmup_probe.txt): a signing pass
            w0 = time.perf_counter()                    # timed behind ONE warm-up launch reads 19 % slow
            while time.perf_counter() - w0 < 0.05:
                run(L, op)
                torch.cuda.synchronize()
            a.record()
            for _ in range(BURST):
                run(L, op)
            b.record(); torch.cuda.synchronize()
            if r:
                times[(name, op)].append(a.elapsed_time(b) / BURST)
            res = {"x25519": out, "keypair": pub, "sign": sig, "verify": ok}[op]
            h = hash(res.cpu().numpy().tobytes())
            assert os.environ.get("AB_ALLOW_DIFF") or ref_out.setdefault(op, h) == h, f"{name} {op}: output differs between builds"
for op in ops:
    for name, _ in libs:
        t = times[(name, op)]
        print(f"{op:8s} {name:44s} min {min(t):8.3f} ms  median {statistics.median(t):8.3f} ms  -> {n / min(t) / 1e3:9.1f} Mops/s")
if "verify" in ops:
    print("verify all ok:", bool(ok.all().item()))
