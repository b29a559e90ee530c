"""Host-side mirror of the reference's operator interface for the scalar-multiplication path.

Function names follow the reference's C API (include/curve25519_dh.h:34-48,
include/ed25519_signature.h:40-93); each takes N-element contiguous arrays and calls the matching
`*_batch` (host memory, numpy) or `*_dev` (device memory, torch CUDA tensors on the current stream)
entry point of libcurve25519_amd.so.  Nothing here computes: no numpy/torch arithmetic, no CPU
fallback -- without the HIP library and a gfx950 device every call raises EngineError.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EngineError  # noqa: F401  (re-export)


def _np(a, width, name):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 1 and a.size == width:
        a = a.reshape(1, width)
    if a.ndim != 2 or a.shape[1] != width:
        raise ValueError(f"{name} must have shape (n, {width}), got {a.shape}")
    return a


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _msgs(msg, n):
    msg = np.ascontiguousarray(msg, dtype=np.uint8)
    if n == 0:
        return msg.reshape(0, 0), 0
    msg = msg.reshape(n, -1)
    return msg, msg.shape[1]


# ---- host-memory (numpy) interface ---------------------------------------------------------------

def curve25519_dh_CreateSharedKey(pk, sk):
    """n x curve25519_dh_CreateSharedKey.  Returns (shared, clamped_sk); inputs are not modified
    (the C function clamps in place -- the clamped copy is returned instead)."""
    pk = _np(pk, 32, "pk")
    sk = np.array(_np(sk, 32, "sk"), copy=True)
    if pk.shape[0] != sk.shape[0]:
        raise ValueError("pk and sk must have the same number of rows")
    out = np.empty_like(pk)
    L = _lib.load()
    _lib.check(L.curve25519_dh_CreateSharedKey_batch(_ptr(out), _ptr(pk), _ptr(sk), pk.shape[0]),
               "curve25519_dh_CreateSharedKey_batch")
    return out, sk


def curve25519_dh_CalculatePublicKey(sk, fast=False):
    """n x curve25519_dh_CalculatePublicKey (or _fast).  Returns (pk, clamped_sk)."""
    sk = np.array(_np(sk, 32, "sk"), copy=True)
    out = np.empty_like(sk)
    L = _lib.load()
    fn = L.curve25519_dh_CalculatePublicKey_fast_batch if fast else L.curve25519_dh_CalculatePublicKey_batch
    _lib.check(fn(_ptr(out), _ptr(sk), sk.shape[0]), "curve25519_dh_CalculatePublicKey_batch")
    return out, sk


def ed25519_CreateKeyPair(sk):
    """n x ed25519_CreateKeyPair(blinding=NULL).  Returns (pub[n,32], priv[n,64])."""
    sk = _np(sk, 32, "sk")
    n = sk.shape[0]
    pub = np.empty((n, 32), np.uint8)
    priv = np.empty((n, 64), np.uint8)
    _lib.check(_lib.load().ed25519_CreateKeyPair_batch(_ptr(pub), _ptr(priv), _ptr(sk), n), "ed25519_CreateKeyPair_batch")
    return pub, priv


def ed25519_SignMessage(priv, msg):
    """n x ed25519_SignMessage(blinding=NULL) over fixed-length messages msg[n, msg_size]."""
    priv = _np(priv, 64, "priv")
    n = priv.shape[0]
    msg, msg_size = _msgs(msg, n)
    sig = np.empty((n, 64), np.uint8)
    _lib.check(_lib.load().ed25519_SignMessage_batch(_ptr(sig), _ptr(priv), _ptr(msg), msg_size, n), "ed25519_SignMessage_batch")
    return sig


def _ragged(messages):
    """list of bytes-like -> (concatenated uint8 array, uint64 offsets[n+1])"""
    lens = np.fromiter((len(m) for m in messages), dtype=np.uint64, count=len(messages))
    offsets = np.zeros(len(messages) + 1, np.uint64)
    np.cumsum(lens, out=offsets[1:])
    flat = np.frombuffer(b"".join(bytes(m) for m in messages), np.uint8) if len(messages) else np.zeros(0, np.uint8)
    return np.ascontiguousarray(flat), offsets


def ed25519_SignMessage_ragged(priv, messages):
    """n x ed25519_SignMessage with per-element message lengths (`messages`: sequence of bytes-like)."""
    priv = _np(priv, 64, "priv")
    n = priv.shape[0]
    if len(messages) != n:
        raise ValueError("one message per private key")
    flat, offsets = _ragged(messages)
    sig = np.empty((n, 64), np.uint8)
    _lib.check(_lib.load().ed25519_SignMessage_ragged_batch(_ptr(sig), _ptr(priv), _ptr(flat), _ptr(offsets), n),
               "ed25519_SignMessage_ragged_batch")
    return sig


def ed25519_VerifySignature_ragged(sig, pk, messages):
    sig = _np(sig, 64, "sig")
    pk = _np(pk, 32, "pk")
    n = sig.shape[0]
    if len(messages) != n or pk.shape[0] != n:
        raise ValueError("one message and one key per signature")
    flat, offsets = _ragged(messages)
    ok = np.empty(n, np.int32)
    _lib.check(_lib.load().ed25519_VerifySignature_ragged_batch(_ptr(ok), _ptr(sig), _ptr(pk), _ptr(flat), _ptr(offsets), n),
               "ed25519_VerifySignature_ragged_batch")
    return ok


def ed25519_VerifySignature(sig, pk, msg):
    """n x ed25519_VerifySignature.  Returns int32[n] of 1 (valid) / 0 (invalid)."""
    sig = _np(sig, 64, "sig")
    pk = _np(pk, 32, "pk")
    n = sig.shape[0]
    if pk.shape[0] != n:
        raise ValueError("sig and pk must have the same number of rows")
    msg, msg_size = _msgs(msg, n)
    ok = np.empty(n, np.int32)
    _lib.check(_lib.load().ed25519_VerifySignature_batch(_ptr(ok), _ptr(sig), _ptr(pk), _ptr(msg), msg_size, n),
               "ed25519_VerifySignature_batch")
    return ok


def ed25519_Verify_Init(pk):
    """n x ed25519_Verify_Init: per-key contexts, uint8[n, 2080] (pk || 16 rows x 4 canonical elements)."""
    pk = _np(pk, 32, "pk")
    ctx = np.empty((pk.shape[0], 2080), np.uint8)
    _lib.check(_lib.load().ed25519_Verify_Init_batch(_ptr(ctx), _ptr(pk), pk.shape[0]), "ed25519_Verify_Init_batch")
    return ctx


def ed25519_Verify_Check(ctx, sig, msg):
    """One key (a 2080-byte context), n (signature, message) pairs -> int32[n] verdicts."""
    ctx = np.ascontiguousarray(ctx, dtype=np.uint8).reshape(-1)
    if ctx.size != 2080:
        raise ValueError("ctx must be one 2080-byte context")
    sig = _np(sig, 64, "sig")
    n = sig.shape[0]
    msg, msg_size = _msgs(msg, n)
    ok = np.empty(n, np.int32)
    _lib.check(_lib.load().ed25519_Verify_Check_batch(_ptr(ok), _ptr(ctx), _ptr(sig), _ptr(msg), msg_size, n),
               "ed25519_Verify_Check_batch")
    return ok


def base_folding8_table():
    """(256, 3, 32) uint8: the device-generated 8-fold base table in the reference's PA_POINT row order."""
    out = np.empty((256, 3, 32), np.uint8)
    _lib.check(_lib.load().c25519_amd_base_table(_ptr(out)), "c25519_amd_base_table")
    return out


def device_count() -> int:
    return int(_lib.load().c25519_amd_device_count())


# ---- device-memory (torch CUDA tensors) interface --------------------------------------------------
# torch is only plumbing here: it owns the HBM buffers and the stream.  Every tensor is checked (CUDA, dtype,
# contiguous, shape, same row count, same device) before its data_ptr() is handed to the C ABI, and the call runs
# with that device current -- a short or strided tensor is a ValueError here, not an out-of-bounds access there.

def _check(t, width, name, n=None, dtype=None, device=None):
    import torch
    dtype = torch.uint8 if dtype is None else dtype
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous {dtype} CUDA tensor")
    if t.dim() != 2 or (width is not None and t.shape[1] != width):
        raise ValueError(f"{name} must have shape (n, {width}), got {tuple(t.shape)}")
    if n is not None and t.shape[0] != n:
        raise ValueError(f"{name} must have {n} rows like the other arguments, got {t.shape[0]}")
    if device is not None and t.device != device:
        raise ValueError(f"{name} is on {t.device}, the other arguments on {device}")
    return C.c_void_p(t.data_ptr())


class _on:
    """`with _on(t):` = torch.cuda.device(t.device) + the current stream of that device as a void*."""

    def __init__(self, t):
        import torch
        self.ctx = torch.cuda.device(t.device)

    def __enter__(self):
        import torch
        self.ctx.__enter__()
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def __exit__(self, *a):
        return self.ctx.__exit__(*a)


def curve25519_dh_CreateSharedKey_dev(shared, pk, sk):
    """In-place device form: writes `shared`, clamps `sk`; asynchronous on torch's current stream."""
    n, d = pk.shape[0], pk.device
    args = (_check(shared, 32, "shared", n, device=d), _check(pk, 32, "pk"), _check(sk, 32, "sk", n, device=d))
    with _on(pk) as st:
        _lib.check(_lib.load().curve25519_dh_CreateSharedKey_dev(*args, n, st), "curve25519_dh_CreateSharedKey_dev")


def curve25519_dh_CalculatePublicKey_dev(pk, sk, fast=False):
    L = _lib.load()
    fn = L.curve25519_dh_CalculatePublicKey_fast_dev if fast else L.curve25519_dh_CalculatePublicKey_dev
    n, d = sk.shape[0], sk.device
    args = (_check(pk, 32, "pk", n, device=d), _check(sk, 32, "sk"))
    with _on(sk) as st:
        _lib.check(fn(*args, n, st), "curve25519_dh_CalculatePublicKey_dev")


def ed25519_CreateKeyPair_dev(pub, priv, sk):
    n, d = sk.shape[0], sk.device
    args = (_check(pub, 32, "pub", n, device=d), _check(priv, 64, "priv", n, device=d), _check(sk, 32, "sk"))
    with _on(sk) as st:
        _lib.check(_lib.load().ed25519_CreateKeyPair_dev(*args, n, st), "ed25519_CreateKeyPair_dev")


def ed25519_SignMessage_dev(sig, priv, msg):
    n, d = priv.shape[0], priv.device
    args = (_check(sig, 64, "sig", n, device=d), _check(priv, 64, "priv"), _check(msg, None, "msg", n, device=d))
    with _on(priv) as st:
        _lib.check(_lib.load().ed25519_SignMessage_dev(*args, msg.shape[1], n, st), "ed25519_SignMessage_dev")


def ed25519_VerifySignature_dev(verdict, sig, pk, msg):
    import torch
    n, d = sig.shape[0], sig.device
    args = (_check(verdict, 1, "verdict", n, dtype=torch.int32, device=d), _check(sig, 64, "sig"),
            _check(pk, 32, "pk", n, device=d), _check(msg, None, "msg", n, device=d))
    with _on(sig) as st:
        _lib.check(_lib.load().ed25519_VerifySignature_dev(*args, msg.shape[1], n, st), "ed25519_VerifySignature_dev")
