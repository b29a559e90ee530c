#!/usr/bin/env python3
"""tools/host_feed_rate.py -- what the *_multi entry points need from the HOST on eight GPUs, against what this box's CPUs and one
PCIe link deliver.  Per pass: bytes a caller's arrays move per operation, that times the
device-resident rate of one GPU (profiles/rNN_bench.json) = GB/s one device's pipeline asks for, x 8; beside it what was measured
here: one link's DMA rate (pinned, each direction and both at once) and what N copy threads move from pageable to pinned memory
(the staging copies of host_pipeline.hpp: SharedCopyPool has 12 threads on a 16-CPU quota).  The verdict column is arithmetic on
those, not a measurement of eight GPUs: nobody has had eight."""
import glob
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = 256 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")


def rate(f, reps=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return reps * n / (time.perf_counter() - t) / 1e9


h2d = rate(lambda: d.copy_(h, non_blocking=True))
d2h = rate(lambda: h.copy_(d, non_blocking=True))
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2):
        h2.copy_(d2, non_blocking=True)


bidir = rate(both)
src = np.ones(n, np.uint8)
dst = h.numpy()


def threaded(k, reps=4):
    cuts = [n * i // k for i in range(k + 1)]
    def work(i):
        for _ in range(reps):
            np.copyto(dst[cuts[i]:cuts[i + 1]], src[cuts[i]:cuts[i + 1]])
    ts = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    t = time.perf_counter()
    [x.start() for x in ts]; [x.join() for x in ts]
    return reps * n / (time.perf_counter() - t) / 1e9


threaded(4)
copies = {k: threaded(k) for k in (1, 2, 4, 8, 12, 16)}
cpus = len(os.sched_getaffinity(0))
bench = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")))[-1]
line = json.loads([l for l in open(bench) if l.startswith("{")][-1])
resident = {"x25519": line["value"], "verify": line["roofline"]["verify"]["value"], "sign": line["roofline"]["sign"]["value"]}
moved = {"x25519": (64, 64), "verify": (128, 4), "sign": (96, 64)}          # bytes in / out per operation with 32-byte messages
print(f"# tools/host_feed_rate.py on {torch.cuda.get_device_name(0)}, {cpus} usable CPUs; resident rates: profiles/{os.path.basename(bench)}")
print(f"one PCIe link, pinned memory: H2D {h2d:.1f} GB/s, D2H {d2h:.1f} GB/s, both directions at once {bidir:.1f} GB/s each")
print("copy threads, pageable -> pinned (numpy.copyto on slices of a 256 MiB array): " + ", ".join(f"{k}: {v:.1f} GB/s" for k, v in copies.items()))
pool = copies[12]
print(f"{'pass':8} {'B in/out per op':>16} {'resident M/s':>13} {'one device GB/s in/out':>24} {'x 8 GB/s':>10}  what binds on eight devices")
for wl, (bi, bo) in moved.items():
    r = resident[wl]
    gi, go = bi * r / 1e9, bo * r / 1e9
    link = min(1.0, h2d / gi if gi else 1.0, d2h / go if go else 1.0)
    host = min(1.0, pool / (8 * (gi + go) * link))
    what = []
    if link < 0.999:
        what.append(f"the link: {link:.2f} x resident on every device, page-locked or not")
    if host < 0.999:
        what.append(f"12 copy threads ({pool:.0f} GB/s) feed pageable arrays at {host:.2f} x of that -- page-lock the arrays")
    if not what:
        what.append("neither (link and copy pool have room)")
    print(f"{wl:8} {f'{bi} / {bo}':>16} {r / 1e6:13.1f} {f'{gi:.1f} / {go:.1f}':>24} {8 * (gi + go):10.1f}  " + "; ".join(what))
