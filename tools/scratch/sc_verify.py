import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from curve25519_amd import api, synth
sk, msg = synth.random_bytes((1, 32), 1), synth.random_bytes((1, 32), 2)
pub, priv = api.ed25519_CreateKeyPair(sk)
sig = api.ed25519_SignMessage(priv, msg)
for _ in range(300):
    assert api.ed25519_VerifySignature(sig, pub, msg)[0] == 1
