import sys,time; sys.path.insert(0,".")
import numpy as np
from curve25519_amd import api, synth
n=1<<20
sk,pk=synth.x25519_inputs(n)
esk,msg=synth.ed25519_inputs(n)
api.curve25519_dh_CreateSharedKey(pk[:1024],sk[:1024])
for name,fn in (("x25519", lambda: api.curve25519_dh_CreateSharedKey(pk,sk)),):
    fn(); t=time.perf_counter(); fn(); fn(); dt=(time.perf_counter()-t)/2
    print(name, "host-buffer API: %.2f ms per 2^20 -> %.1f M ops/s (PCIe + staging inclusive)"%(dt*1e3, n/dt/1e6))
pub,priv=api.ed25519_CreateKeyPair(esk)
sig=api.ed25519_SignMessage(priv,msg)
for name,fn in (("sign", lambda: api.ed25519_SignMessage(priv,msg)),("verify", lambda: api.ed25519_VerifySignature(sig,pub,msg))):
    fn(); t=time.perf_counter(); fn(); fn(); dt=(time.perf_counter()-t)/2
    print(name, "host-buffer API: %.2f ms per 2^20 -> %.1f M ops/s"%(dt*1e3, n/dt/1e6))
