#!/usr/bin/env python3
"""tools/scratch/warmup_probe.py OP -- how many untimed launches does a timed burst need in front of it?  After an idle gap
(a host-side sleep) run W warm-up launches, then time 4 launches with HIP events; W = 0, 1, 5, 20, 100, 400."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from curve25519_amd import synth
op = sys.argv[1] if len(sys.argv) > 1 else "sign"
n = 1 << 20
L = C.CDLL(os.path.abspath("curve25519_amd/libcurve25519_amd.so"))
vp, sz = C.c_void_p, C.c_size_t
L.curve25519_dh_CreateSharedKey_dev.argtypes = [vp, vp, vp, sz, vp]
L.ed25519_CreateKeyPair_dev.argtypes = [vp, vp, vp, sz, vp]
L.ed25519_SignMessage_dev.argtypes = [vp, vp, vp, sz, sz, vp]
L.ed25519_VerifySignature_dev.argtypes = [vp, vp, vp, vp, sz, sz, vp]
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(a).to(dev)
sk, pk = synth.x25519_inputs(n); esk, msg = synth.ed25519_inputs(n, 32)
sk, pk, esk, msg = t(sk), t(pk), t(esk), t(msg)
out = torch.empty((n, 32), dtype=torch.uint8, device=dev); pub = torch.empty_like(out)
priv = torch.empty((n, 64), dtype=torch.uint8, device=dev); sig = torch.empty_like(priv)
ok = torch.empty((n,), dtype=torch.int32, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda x: C.c_void_p(x.data_ptr())
L.ed25519_CreateKeyPair_dev(p(pub), p(priv), p(esk), n, st()); L.ed25519_SignMessage_dev(p(sig), p(priv), p(msg), 32, n, st())
def run():
    if op == "sign": rc = L.ed25519_SignMessage_dev(p(sig), p(priv), p(msg), 32, n, st())
    elif op == "keypair": rc = L.ed25519_CreateKeyPair_dev(p(pub), p(priv), p(esk), n, st())
    elif op == "x25519": rc = L.curve25519_dh_CreateSharedKey_dev(p(out), p(pk), p(sk), n, st())
    else: rc = L.ed25519_VerifySignature_dev(p(ok), p(sig), p(pub), p(msg), 32, n, st())
    assert rc == 0
torch.cuda.synchronize()
for idle in (0.2, 0.0):
    for W in (0, 1, 5, 20, 100, 400):
        res = []
        for rep in range(3):
            torch.cuda.synchronize(); time.sleep(idle)
            for _ in range(W): run()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4): run()
            b.record(); torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 4)
        print(f"{op:8s} idle {idle:.1f} s, {W:4d} warm-up launches: " + "  ".join(f"{x:7.3f}" for x in res) + " ms per launch")
