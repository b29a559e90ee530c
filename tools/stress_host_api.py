#!/usr/bin/env python3
"""tools/stress_host_api.py -- soak of the host-pointer (*_batch) pipeline: several host threads, each with its own
streams / buffer sets / helper threads, hammer the three passes with batch sizes that exercise every shape of
run_batch (single piece, exactly 8 pieces, more pieces than buffer sets via C25519_AMD_BATCH_PIECES, ragged tails),
pageable and page-locked arguments mixed; every result is compared with the bytes the device-pointer path gave for the
same inputs.  In the mix: the device-pointer calls on a stream of the thread's own (three passes back to back), blinded signatures with one context, two-phase verification under one key (per-wave kernels, the
reference-order kernel and the two wide combs with the thread's remembered key comb), and a `*_multi` handle of three virtual
devices per thread (workers, gather streams, the process-wide copy threads shared by every thread's handle).
    python tools/stress_host_api.py [--threads 4] [--iters 40]
"""
import argparse
import ctypes as C
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from curve25519_amd import _lib, api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=4)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--n", type=int, default=(1 << 19) + 12345)
ap.add_argument("--churn", type=int, default=1, help="generations of threads: every generation starts --threads fresh threads and joins them (no "
                                                        "c25519_amd_thread_release in generations > 1: the thread-exit path frees)")
ap.add_argument("--only", type=int, default=-1, help="every call is this operation (0 X25519, 1 sign, 2 verify, 3 blinded sign, 4 one-key verify, 5 *_multi, 6 *_dev on the thread's own stream)")
args = ap.parse_args()
torch.cuda.init()                        # (in the main thread, before the workers: op 6 hands torch's streams and tensors to *_dev)
L = _lib.load()
N = args.n
sk, pk = synth.x25519_inputs(N)
esk, msg = synth.ed25519_inputs(N)
shared, sk_clamped = api.curve25519_dh_CreateSharedKey(pk, sk)
pub, priv = api.ed25519_CreateKeyPair(esk)
sig = api.ed25519_SignMessage(priv, msg)
bsig, bmsg, bad = synth.corrupt_for_verify(sig, msg)
want_ok = (~bad).astype(np.int32)
P = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
errors = []
# one key, many signatures (two-phase verification): the same private key over the first 2^17 + 77 messages
ONE = (1 << 17) + 77
one_sig = api.ed25519_SignMessage(np.ascontiguousarray(np.broadcast_to(priv[5], (ONE, 64))), msg[:ONE])
one_bsig, one_bmsg, one_bad = synth.corrupt_for_verify(one_sig, msg[:ONE])
one_ok = (~one_bad).astype(np.int32)
one_ctx = api.ed25519_Verify_Init(pub[5:6])[0]
bctx = np.zeros(192, np.uint8)
L.ed25519_Blinding_Init.restype = C.c_void_p
assert L.ed25519_Blinding_Init(P(bctx), b"stress", 6) == bctx.ctypes.data


def worker(tid):
    rng = np.random.default_rng(1000 + tid)
    locked = []
    handle = C.c_void_p()
    stream = None
    try:
        dev = (C.c_int * 3)(0, 0, 0)
        assert L.c25519_amd_multi_create(C.byref(handle), dev, 3) == 0
        for it in range(args.iters):
            n = int(rng.choice([1, 2, 64, 65, 255, 1025, 4097, 1 << 16, 1 << 17, (1 << 17) + 1, 200003, 1 << 18, N]))
            lo = int(rng.integers(0, N - n + 1))
            sl = slice(lo, lo + n)
            op = it % 7 if args.only < 0 else args.only
            detail = ""
            lock = bool(rng.integers(0, 2))
            if op == 0:
                out, s2, p2 = np.zeros((n, 32), np.uint8), sk[sl].copy(), np.ascontiguousarray(pk[sl])
                if not lock and rng.integers(0, 2):                # host arrays at odd byte addresses: nothing may assume alignment
                    def odd(a):
                        raw = np.empty(a.size + 3, np.uint8)
                        v = raw[3:].reshape(a.shape)
                        v[...] = a
                        return v
                    out, s2, p2 = odd(out), odd(s2), odd(p2)
                if lock:
                    out, s2 = synth.page_aligned((n, 32)), synth.page_aligned((n, 32), like=s2)
                    for a in (out, s2):
                        assert L.c25519_amd_host_register(P(a), synth.locked_bytes(a)) == 0
                        locked.append(a)
                assert L.curve25519_dh_CreateSharedKey_batch(P(out), P(p2), P(s2), n) == 0
                good = np.array_equal(out, shared[sl]) and np.array_equal(s2, sk_clamped[sl])
            elif op == 1:
                out = np.zeros((n, 64), np.uint8)
                pr, m = np.ascontiguousarray(priv[sl]), np.ascontiguousarray(msg[sl])
                if lock:
                    pr = synth.page_aligned(pr.shape, like=pr)
                    assert L.c25519_amd_host_register(P(pr), synth.locked_bytes(pr)) == 0
                    locked.append(pr)
                assert L.ed25519_SignMessage_batch(P(out), P(pr), P(m), m.shape[1], n) == 0
                good = np.array_equal(out, sig[sl])
            elif op == 3:                                          # blinded signatures, one context for every thread
                out = np.zeros((n, 64), np.uint8)
                pr, m = np.ascontiguousarray(priv[sl]), np.ascontiguousarray(msg[sl])
                assert L.ed25519_SignMessage_blinded_batch(P(out), P(pr), P(bctx), P(m), m.shape[1], n) == 0
                good = np.array_equal(out, sig[sl])
            elif op == 4:                                          # two-phase verification under one key
                n = min(n, ONE)
                lo = int(rng.integers(0, ONE - n + 1))
                sl = slice(lo, lo + n)
                out = np.full(n, -7, np.int32)
                s2, m = np.ascontiguousarray(one_bsig[sl]), np.ascontiguousarray(one_bmsg[sl])
                assert L.ed25519_Verify_Check_batch(P(out), P(one_ctx), P(s2), P(m), m.shape[1], n) == 0
                good = np.array_equal(out, one_ok[sl])
            elif op == 5:                                          # the multi-device layer: three virtual devices, either gather mode
                mode = int(rng.integers(0, 2))
                assert L.c25519_amd_multi_set_gather(handle, mode) == 0
                out, s2, p2 = np.full((n, 32), 0xEE, np.uint8), sk[sl].copy(), np.ascontiguousarray(pk[sl])
                assert L.curve25519_dh_CreateSharedKey_multi(handle, P(out), P(p2), P(s2), n) == 0
                good = np.array_equal(out, shared[sl]) and np.array_equal(s2, sk_clamped[sl])
                if not good:
                    badrows = np.nonzero((out != shared[sl]).any(axis=1))[0]
                    detail = f"x25519 gather={mode} bad rows {len(badrows)}: {badrows[:4]}..{badrows[-2:]} first bytes {out[badrows[0], :4] if len(badrows) else None} sk ok {np.array_equal(s2, sk_clamped[sl])}"
                if good:
                    o2 = np.full((n, 64), 0xEE, np.uint8)
                    pr, m = np.ascontiguousarray(priv[sl]), np.ascontiguousarray(msg[sl])
                    assert L.ed25519_SignMessage_multi(handle, P(o2), P(pr), P(m), m.shape[1], n) == 0
                    good = np.array_equal(o2, sig[sl])
                    if not good:
                        badrows = np.nonzero((o2 != sig[sl]).any(axis=1))[0]
                        detail = f"sign gather={mode} bad rows {len(badrows)}: {badrows[:4]}..{badrows[-2:]} first bytes {o2[badrows[0], :4]}"
            elif op == 6:                                          # device pointers, this thread's own stream, three passes back to back
                if stream is None:
                    stream = torch.cuda.Stream()
                with torch.cuda.stream(stream):
                    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", non_blocking=False)  # noqa: E731
                    d_pk, d_sk = up(pk[sl]), up(sk[sl])
                    d_out = torch.full((n, 32), 0xEE, dtype=torch.uint8, device="cuda")
                    api.curve25519_dh_CreateSharedKey_dev(d_out, d_pk, d_sk)
                    d_priv, d_msg = up(priv[sl]), up(msg[sl])
                    d_sig = torch.full((n, 64), 0xEE, dtype=torch.uint8, device="cuda")
                    api.ed25519_SignMessage_dev(d_sig, d_priv, d_msg)
                    d_bsig, d_pub, d_bmsg = up(bsig[sl]), up(pub[sl]), up(bmsg[sl])
                    d_ok = torch.full((n, 1), -7, dtype=torch.int32, device="cuda")
                    api.ed25519_VerifySignature_dev(d_ok, d_bsig, d_pub, d_bmsg)
                    stream.synchronize()
                    good = (np.array_equal(d_out.cpu().numpy(), shared[sl]) and np.array_equal(d_sk.cpu().numpy(), sk_clamped[sl])
                            and np.array_equal(d_sig.cpu().numpy(), sig[sl]) and np.array_equal(d_ok.cpu().numpy()[:, 0], want_ok[sl]))
                    if not good:
                        detail = f"dev: x {np.array_equal(d_out.cpu().numpy(), shared[sl])} sign {np.array_equal(d_sig.cpu().numpy(), sig[sl])} verify {np.array_equal(d_ok.cpu().numpy()[:, 0], want_ok[sl])}"
            else:
                out = np.full(n, -7, np.int32)
                s2, p2, m = np.ascontiguousarray(bsig[sl]), np.ascontiguousarray(pub[sl]), np.ascontiguousarray(bmsg[sl])
                if lock:
                    out = synth.page_aligned(n, np.int32, like=out)
                    assert L.c25519_amd_host_register(P(out), synth.locked_bytes(out)) == 0
                    locked.append(out)
                assert L.ed25519_VerifySignature_batch(P(out), P(s2), P(p2), P(m), m.shape[1], n) == 0
                good = np.array_equal(out, want_ok[sl])
            while locked:
                assert L.c25519_amd_host_unregister(P(locked.pop())) == 0
            if not good:
                errors.append((tid, it, op, n, lo, lock, detail))
                return
        if args.churn == 1 or tid % 2:
            L.c25519_amd_thread_release()
    except Exception as e:  # noqa: BLE001
        errors.append((tid, repr(e), L.c25519_amd_last_error()))
    finally:
        if handle:
            L.c25519_amd_multi_destroy(handle)


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20


marks = []
for gen in range(args.churn):
    threads = [threading.Thread(target=worker, args=(gen * args.threads + t,)) for t in range(args.threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                             # (torch caches freed blocks per stream: not this library's memory)
    marks.append((torch.cuda.mem_get_info()[0] / 2**20, rss_mb()))
    if errors:
        break
if args.churn > 1:
    k = min(2, len(marks) - 1)                           # (the first generations warm the allocators' caches)
    dev_growth, rss_growth = marks[k][0] - marks[-1][0], marks[-1][1] - marks[k][1]
    print(f"churn: {len(marks)} generations of {args.threads} threads; device memory in use grew {dev_growth:.0f} MiB, host RSS {rss_growth:.0f} MiB "
          f"between generation {k + 1} and the last")
    print("  per generation (device MiB free, host RSS MiB):", " ".join(f"{a:.0f}/{b:.0f}" for a, b in marks))
    # (host RSS is reported, not judged: Python / numpy / torch keep per-thread arenas; the library's own per-thread state is
    # measured without them by tools/scratch/churn_threads.c -- flat over 480 threads)
    if dev_growth > 256:
        errors.append(("device memory leak", dev_growth))
print("errors:", errors)
print("stress ok" if not errors else "STRESS FAILED", f"({args.threads} threads x {args.iters} calls, pieces={os.environ.get('C25519_AMD_BATCH_PIECES', '8')})")
sys.exit(1 if errors else 0)
