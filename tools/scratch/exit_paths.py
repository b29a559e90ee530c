#!/usr/bin/env python3
"""tools/scratch/exit_paths.py -- ways a process can END with the library's state alive: each case runs in a child process and must
exit 0 quickly (no hang in a static destructor, no abort from a HIP call after the runtime has gone)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PRE = f"""
import sys, ctypes as C, threading, numpy as np
sys.path.insert(0, {ROOT!r})
from curve25519_amd import _lib, api, synth
L = _lib.load()
n = (1 << 18) + 5
sk, pk = synth.x25519_inputs(n)
"""
CASES = {
    "plain exit after a pipelined call (helpers parked, no release)": "api.curve25519_dh_CreateSharedKey(pk, sk)",
    "sys.exit in the middle of the script": "api.curve25519_dh_CreateSharedKey(pk, sk); sys.exit(0)",
    "os._exit": "api.curve25519_dh_CreateSharedKey(pk, sk); import os; os._exit(0)",
    "a multi handle never destroyed": "h = C.c_void_p(); assert L.c25519_amd_multi_create(C.byref(h), (C.c_int * 3)(0, 0, 0), 3) == 0\n"
                                      "out = np.zeros((n, 32), np.uint8); s2 = sk.copy()\n"
                                      "assert L.curve25519_dh_CreateSharedKey_multi(h, out.ctypes.data, pk.ctypes.data, s2.ctypes.data, n) == 0",
    "a daemon thread parked after its calls": "ev = threading.Event()\n"
                                                "def w():\n    api.curve25519_dh_CreateSharedKey(pk, sk); ev.set(); threading.Event().wait()\n"
                                                "threading.Thread(target=w, daemon=True).start(); ev.wait()",
    "thread_release then more calls then exit": "api.curve25519_dh_CreateSharedKey(pk, sk); L.c25519_amd_thread_release(); api.curve25519_dh_CreateSharedKey(pk[:7], sk[:7])",
    "one-key verification state and blinding state alive": "esk, msg = synth.ed25519_inputs(1 << 16)\npub, priv = api.ed25519_CreateKeyPair(esk)\n"
        "sig = api.ed25519_SignMessage(np.ascontiguousarray(np.broadcast_to(priv[0], (1 << 16, 64))), msg)\n"
        "assert api.ed25519_Verify_Check(api.ed25519_Verify_Init(pub[:1])[0], sig, msg).all()",
}
bad = 0
# exit() while another thread is in the middle of a call: a C program (a Python daemon thread cannot show it -- the interpreter
# frees the arrays its call is still writing when it finalises, whatever the library does)
exe = "/tmp/exit_midcall"
subprocess.check_call(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "exit_midcall.c"), "-o", exe,
                       "-L", os.path.join(ROOT, "curve25519_amd"), "-lcurve25519_amd", "-lpthread",
                       "-Wl,-rpath," + os.path.join(ROOT, "curve25519_amd"), "-Wl,-rpath,/opt/rocm/lib"])
for mode in (0, 1):
    for us in (300, 1100, 2300, 4100, 7700, 13000):
        try:
            rc = subprocess.run([exe, str(mode), str(us)], capture_output=True, text=True, timeout=60).returncode
        except subprocess.TimeoutExpired:
            rc = "HANG"
        print(f"{'ok ' if rc == 0 else 'BAD'} rc={rc}  exit() {us} us after a completed call, inside a loop of {'*_multi' if mode else '*_batch'} calls on another thread")
        bad += rc != 0
for name, body in CASES.items():
    t0 = time.time()
    try:
        p = subprocess.run([sys.executable, "-c", PRE + body], capture_output=True, text=True, timeout=120)
        rc, err = p.returncode, p.stderr.strip().splitlines()[-1:] if p.returncode else ""
    except subprocess.TimeoutExpired:
        rc, err = "HANG", ""
    print(f"{'ok ' if rc == 0 else 'BAD'} rc={rc} {time.time() - t0:5.1f} s  {name} {err}")
    bad += rc != 0
sys.exit(1 if bad else 0)
