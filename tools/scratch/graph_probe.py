#!/usr/bin/env python3
"""tools/scratch/graph_probe.py -- can a warm *_dev call be captured into a HIP graph, and what does replaying it save?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from curve25519_amd import api, synth
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(a).to(dev)
for n in (8192, 16384, 65536):
    esk, msg = synth.ed25519_inputs(n, 32)
    pub_h, priv_h = api.ed25519_CreateKeyPair(esk)
    sig_h = api.ed25519_SignMessage(priv_h, msg)
    priv, msgd, pub, sigd = t(priv_h), t(msg), t(pub_h), t(sig_h)
    sig = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    ok = torch.empty((n, 1), dtype=torch.int32, device=dev)
    s = torch.cuda.Stream(dev)
    for name, call, res, exp in (("sign", lambda: api.ed25519_SignMessage_dev(sig, priv, msgd), sig, sig_h),
                                 ("verify", lambda: api.ed25519_VerifySignature_dev(ok, sigd, pub, msgd), ok, np.ones((n, 1), np.int32))):
        with torch.cuda.stream(s):
            for _ in range(3): call()
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                call()
            res.zero_(); torch.cuda.synchronize()
            g.replay(); torch.cuda.synchronize()
            good = np.array_equal(res.cpu().numpy(), exp)
        except Exception as e:
            print(f"n={n} {name}: capture failed: {e!r}"[:300]); torch.cuda.synchronize(); continue
        def timeit(fn, reps=200):
            for _ in range(20): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
        with torch.cuda.stream(s):
            direct = timeit(call)
        replay = timeit(g.replay)
        print(f"n={n:6d} {name:6s}: captured, replay correct: {good}; back-to-back direct calls {direct:7.1f} us each, graph replays {replay:7.1f} us each")
