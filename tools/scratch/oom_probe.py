"""tools/scratch/oom_probe.py -- the device is (almost) full when a call needs its work scratch: the call must fail with a message and
leave the library usable.  Every pointer and size is real (no out-of-bounds launch if an allocation unexpectedly succeeds)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from curve25519_amd import _lib, api, synth
L = _lib.load()
dev = torch.device("cuda", 0)
n = 1 << 20
esk, msg = synth.ed25519_inputs(n)
pub, priv = api.ed25519_CreateKeyPair(esk[:4096])
d_sig = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
d_pk = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
d_msg = torch.from_numpy(msg).to(dev)
d_ok = torch.zeros((n, 1), dtype=torch.int32, device=dev)
free, total = torch.cuda.mem_get_info()
print(f"free {free / 2**30:.1f} GiB of {total / 2**30:.1f}", flush=True)
hog = []
chunk = 8 << 30
while torch.cuda.mem_get_info()[0] > (2 << 30) + chunk:
    hog.append(torch.empty(chunk, dtype=torch.uint8, device=dev))
while torch.cuda.mem_get_info()[0] > (1 << 30) + (256 << 20):
    hog.append(torch.empty(256 << 20, dtype=torch.uint8, device=dev))
print(f"free now {torch.cuda.mem_get_info()[0] / 2**30:.2f} GiB; verification of 2^20 needs {L.ed25519_VerifySignature_scratch_bytes(n) / 2**30:.2f} GiB of scratch", flush=True)
P = lambda t: C.c_void_p(t.data_ptr())
rc = L.ed25519_VerifySignature_dev(P(d_ok), P(d_sig), P(d_pk), P(d_msg), 32, n, None)
print("verify_dev on a full device: rc =", rc, "|", L.c25519_amd_last_error().decode()[:200], flush=True)
sig = np.zeros((n, 64), np.uint8); pk = np.zeros((n, 32), np.uint8); ok = np.zeros(n, np.int32)
rc2 = L.ed25519_VerifySignature_batch(ok.ctypes.data, sig.ctypes.data, pk.ctypes.data, msg.ctypes.data, 32, n)
print("verify_batch on a full device: rc =", rc2, "|", L.c25519_amd_last_error().decode()[:200], flush=True)
del hog
torch.cuda.empty_cache()
torch.cuda.synchronize()
s = api.ed25519_SignMessage(priv, msg[:4096])
print("after the failures:", bool(api.ed25519_VerifySignature(s, pub, msg[:4096]).all()), "and rc was nonzero:", rc != 0 and rc2 != 0, flush=True)
