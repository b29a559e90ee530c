#!/usr/bin/env python3
"""tools/mid_batch_kernels.py -- which kernels a *_dev call of n device-resident elements launches and how long each runs, for
the sizes between the per-wave and the chip-filling shapes.  Two modes:

    rocprofv3 --kernel-trace -d DIR -o mid -- python tools/mid_batch_kernels.py run [--exps 12,14] [--ops x25519,sign,keypair,verify]
    python tools/mid_batch_kernels.py report DIR/.../mid_results.db      # per (kernel, grid): launches, average / minimum us

`run` warms every (op, n) for ~30 ms, then issues 24 calls; `report` drops the first half of every kernel's launches."""
import os
import sqlite3
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(argv):
    import argparse
    import numpy as np
    import torch
    from curve25519_amd import api, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--exps", default="12,14")
    ap.add_argument("--ops", default="x25519,public_fast,keypair,sign,verify")
    a = ap.parse_args(argv)
    exps = [int(e) for e in a.exps.split(",")]
    N = 1 << max(exps)
    dev = torch.device("cuda", 0)
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    sk, pk = (up(x) for x in synth.x25519_inputs(N))
    esk, msg = synth.ed25519_inputs(N)
    pub, priv = api.ed25519_CreateKeyPair(esk)
    sig = api.ed25519_SignMessage(priv, msg)
    desk, dpriv, dmsg, dsig, dpub = up(esk), up(priv), up(msg), up(sig), up(pub)
    o32, o64, p32, p64 = (torch.empty((N, w), dtype=torch.uint8, device=dev) for w in (32, 64, 32, 64))
    ok = torch.empty((N, 1), dtype=torch.int32, device=dev)
    ops = {
        "x25519": lambda n: api.curve25519_dh_CreateSharedKey_dev(o32[:n], pk[:n], sk[:n]),
        "public_fast": lambda n: api.curve25519_dh_CalculatePublicKey_dev(o32[:n], sk[:n], fast=True),
        "keypair": lambda n: api.ed25519_CreateKeyPair_dev(p32[:n], p64[:n], desk[:n]),
        "sign": lambda n: api.ed25519_SignMessage_dev(o64[:n], dpriv[:n], dmsg[:n]),
        "verify": lambda n: api.ed25519_VerifySignature_dev(ok[:n], dsig[:n], dpub[:n], dmsg[:n]),
    }
    for name in a.ops.split(","):
        for e in exps:
            n = 1 << e
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.03:
                ops[name](n)
                torch.cuda.synchronize()
            for _ in range(24):
                ops[name](n)
            torch.cuda.synchronize()


def report(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, grid_x, workgroup_x, start, end - start from kernels order by start").fetchall()
    groups, order = {}, []
    for name, grid, wg, _start, dur in rows:
        key = (name.split("(")[0][:64], grid, wg)
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(dur / 1e3)
    print(f"# {path}\n{'kernel':<66}{'lanes x wg':>16}{'launches':>10}{'avg us':>10}{'min us':>10}")
    for key in order:
        d = groups[key]
        d = d[len(d) // 2:]
        print(f"{key[0]:<66}{str(key[1]) + ' x ' + str(key[2]):>16}{len(groups[key]):>10}{sum(d) / len(d):>10.1f}{min(d):>10.1f}")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run(sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "run" else sys.argv[1:])
