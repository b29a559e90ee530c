import ctypes as C, sys, time, os
import numpy as np
def run(path):
    L=C.CDLL(os.path.abspath(path))
    buf=lambda n,f=0:(C.c_ubyte*n)(*([f]*n))
    sk,pk,sh=buf(32,7),buf(32,9),buf(32)
    esk,pub,priv,sig,msg=buf(32,3),buf(32),buf(64),buf(64),buf(32,5)
    L.ed25519_CreateKeyPair(pub,priv,None,esk); L.ed25519_SignMessage(sig,priv,None,msg,32)
    L.ed25519_VerifySignature.restype=C.c_int
    ops={"shared":lambda:L.curve25519_dh_CreateSharedKey(sh,pk,sk),"public":lambda:L.curve25519_dh_CalculatePublicKey(pk,sk),
         "keypair":lambda:L.ed25519_CreateKeyPair(pub,priv,None,esk),"sign":lambda:L.ed25519_SignMessage(sig,priv,None,msg,32),
         "verify":lambda:L.ed25519_VerifySignature(sig,pub,msg,32)}
    out={}
    for k,f in ops.items():
        for _ in range(50): f()
        t=time.perf_counter()
        for _ in range(400): f()
        out[k]=(time.perf_counter()-t)/400*1e6
    return out
res={p:[] for p in sys.argv[1:]}
for r in range(3):
    for p in sys.argv[1:]:
        res[p].append(run(p))
for p,rs in res.items():
    print(os.path.basename(p), {k: round(min(r[k] for r in rs),1) for k in rs[0]})
