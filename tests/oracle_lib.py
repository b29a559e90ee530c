"""ctypes bindings for the parity checker (oracle/) -- TEST INFRASTRUCTURE ONLY.

`Oracle`   : the repo's own CPU restatement (oracle/liborc25519.so), travels everywhere.
`Reference`: the real reference built from /root/reference by `make -C oracle ref`
             (oracle/_ref/libcurve25519_ref.so); present in the build container and, as a prebuilt
             .so, on the GPU box.  Never imported by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_SO = os.path.join(ORC_DIR, "liborc25519.so")
REF_SO = os.path.join(ORC_DIR, "_ref", "libcurve25519_ref.so")
REF_ASM_SO = os.path.join(os.path.dirname(REF_SO), "libcurve25519_ref_asm.so")   # the x86-64 assembly back-end (make -C oracle ref-asm)

u8p = C.POINTER(C.c_uint8)


def _p(a):
    return a.ctypes.data_as(u8p)


def _rows(a, n):
    """reshape to (n, row_len); an empty batch keeps whatever row length it has (or 0)."""
    if n == 0:
        return a.reshape(0, a.shape[1] if a.ndim == 2 else 0)
    return a.reshape(n, -1)


def build_oracle(force=False):
    srcs = [os.path.join(ORC_DIR, f) for f in os.listdir(ORC_DIR) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(ORC_SO)) or any(os.path.getmtime(s) > os.path.getmtime(ORC_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORC_DIR, "liborc25519.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/source") and not os.path.exists(REF_SO):
        subprocess.check_call(["make", "-C", ORC_DIR, "ref"], stdout=subprocess.DEVNULL)


class Oracle:
    """Batch interface over oracle/liborc25519.so; arrays are numpy uint8, shape (n, k)."""

    def __init__(self):
        build_oracle()
        self.lib = C.CDLL(ORC_SO)
        L = self.lib
        L.orc_x25519_shared_batch.argtypes = [u8p, u8p, u8p, C.c_size_t, C.c_int]
        L.orc_x25519_public_batch.argtypes = [u8p, u8p, C.c_size_t, C.c_int, C.c_int]
        L.orc_ed25519_keypair_batch.argtypes = [u8p, u8p, u8p, C.c_size_t, C.c_int]
        L.orc_ed25519_sign_batch.argtypes = [u8p, u8p, u8p, C.c_size_t, C.c_size_t, C.c_int]
        L.orc_ed25519_verify_batch.argtypes = [C.POINTER(C.c_int32), u8p, u8p, u8p, C.c_size_t, C.c_size_t, C.c_int]
        L.orc_fill_random.argtypes = [u8p, C.c_size_t, C.c_uint64]
        L.orc_base_folding8.restype = C.c_void_p
        L.orc_sha512_init.argtypes = [C.c_void_p]
        L.orc_sha512_update.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_sha512_final.argtypes = [C.c_void_p, u8p]
        L.orc_fold8.argtypes = [u8p, u8p]
        L.orc_fold4.argtypes = [u8p, u8p]

    def random_bytes(self, shape, seed):
        a = np.empty(shape, dtype=np.uint8)
        self.lib.orc_fill_random(_p(a), a.size, seed)
        return a

    def x25519_shared(self, pk, sk, threads=1):
        """returns (shared, clamped_sk); inputs are not modified."""
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        sk = np.array(sk, dtype=np.uint8, copy=True, order="C")
        out = np.empty_like(pk)
        self.lib.orc_x25519_shared_batch(_p(out), _p(pk), _p(sk), pk.shape[0], threads)
        return out, sk

    def x25519_public(self, sk, fast=False, threads=1):
        sk = np.array(sk, dtype=np.uint8, copy=True, order="C")
        out = np.empty_like(sk)
        self.lib.orc_x25519_public_batch(_p(out), _p(sk), sk.shape[0], int(fast), threads)
        return out, sk

    def ed25519_keypair(self, sk, threads=1):
        sk = np.ascontiguousarray(sk, dtype=np.uint8)
        n = sk.shape[0]
        pub = np.empty((n, 32), np.uint8)
        priv = np.empty((n, 64), np.uint8)
        self.lib.orc_ed25519_keypair_batch(_p(pub), _p(priv), _p(sk), n, threads)
        return pub, priv

    def ed25519_sign(self, priv, msg, threads=1):
        priv = np.ascontiguousarray(priv, dtype=np.uint8)
        msg = np.ascontiguousarray(msg, dtype=np.uint8)
        n = priv.shape[0]
        msg = _rows(msg, n)
        sig = np.empty((n, 64), np.uint8)
        self.lib.orc_ed25519_sign_batch(_p(sig), _p(priv), _p(msg), msg.shape[1], n, threads)
        return sig

    def ed25519_verify(self, sig, pk, msg, threads=1):
        sig = np.ascontiguousarray(sig, dtype=np.uint8)
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        n = sig.shape[0]
        msg = _rows(np.ascontiguousarray(msg, dtype=np.uint8), n)
        ok = np.empty(n, np.int32)
        self.lib.orc_ed25519_verify_batch(ok.ctypes.data_as(C.POINTER(C.c_int32)), _p(sig), _p(pk), _p(msg),
                                          msg.shape[1], n, threads)
        return ok

    def sha512(self, data: bytes) -> bytes:
        ctx = C.create_string_buffer(64 + 128 + 8 + 16)
        out = np.empty(64, np.uint8)
        self.lib.orc_sha512_init(ctx)
        self.lib.orc_sha512_update(ctx, data, len(data))
        self.lib.orc_sha512_final(ctx, _p(out))
        return out.tobytes()

    def fold8(self, k: bytes):
        kk = np.frombuffer(k, np.uint8).copy()
        out = np.empty(32, np.uint8)
        self.lib.orc_fold8(_p(out), _p(kk))
        return out

    def fold4(self, k: bytes):
        kk = np.frombuffer(k, np.uint8).copy()
        out = np.empty(64, np.uint8)
        self.lib.orc_fold4(_p(out), _p(kk))
        return out

    def ed25519_verify_point(self, sig, pk, msg):
        """enc(T), T = s*B + h*(-A): what Verify_Check compares with enc(R).  uint8[n, 32]."""
        sig = np.ascontiguousarray(sig, dtype=np.uint8)
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        n = sig.shape[0]
        msg = _rows(np.ascontiguousarray(msg, dtype=np.uint8), n)
        self.lib.orc_ed25519_verify_point.argtypes = [u8p, u8p, u8p, u8p, C.c_size_t]
        out = np.empty((n, 32), np.uint8)
        for i in range(n):
            self.lib.orc_ed25519_verify_point(_p(out[i]), _p(sig[i]), _p(pk[i]), _p(msg[i]), msg.shape[1])
        return out

    def verify_init_table(self, pk):
        """ed25519_Verify_Init for n keys: the 16-row 4-fold tables as Python ints mod p, shape [n][16][4]."""
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        self.lib.orc_ed25519_verify_init.argtypes = [C.c_void_p, u8p]
        P = 2**255 - 19
        out = []
        for i in range(pk.shape[0]):
            ctx = np.zeros(32 + 16 * 4 * 32, np.uint8)
            self.lib.orc_ed25519_verify_init(ctx.ctypes.data, _p(pk[i]))
            assert ctx[:32].tobytes() == pk[i].tobytes()
            rows = ctx[32:].reshape(16, 4, 32)
            out.append([[int.from_bytes(rows[r, f].tobytes(), "little") % P for f in range(4)] for r in range(16)])
        return out

    def base_table(self):
        """(256, 3, 32) uint8: canonical (Y+X, Y-X, 2dT) rows of the 8-fold table."""
        ptr = self.lib.orc_base_folding8()
        return np.ctypeslib.as_array(C.cast(ptr, u8p), shape=(256 * 96,)).reshape(256, 3, 32).copy()


class Reference:
    """Single-call loops over the real reference library (portable-C back-end; asm=True: its x86-64 assembly back-end)."""

    @staticmethod
    def available(asm=False):
        return os.path.exists(REF_ASM_SO if asm else REF_SO)

    def __init__(self, asm=False):
        self.so = REF_ASM_SO if asm else REF_SO
        self.lib = C.CDLL(self.so)
        L = self.lib
        L.curve25519_dh_CreateSharedKey.argtypes = [u8p, u8p, u8p]
        L.curve25519_dh_CalculatePublicKey.argtypes = [u8p, u8p]
        L.curve25519_dh_CalculatePublicKey_fast.argtypes = [u8p, u8p]
        L.ed25519_CreateKeyPair.argtypes = [u8p, u8p, C.c_void_p, u8p]
        L.ed25519_SignMessage.argtypes = [u8p, u8p, C.c_void_p, u8p, C.c_size_t]
        L.ed25519_VerifySignature.argtypes = [u8p, u8p, u8p, C.c_size_t]
        L.ed25519_VerifySignature.restype = C.c_int
        L.ecp_8Folds.argtypes = [u8p, u8p]
        L.ecp_4Folds.argtypes = [u8p, u8p]

    def x25519_shared(self, pk, sk):
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        sk = np.array(sk, dtype=np.uint8, copy=True, order="C")
        out = np.empty_like(pk)
        for i in range(pk.shape[0]):
            self.lib.curve25519_dh_CreateSharedKey(_p(out[i]), _p(pk[i]), _p(sk[i]))
        return out, sk

    def x25519_public(self, sk, fast=False):
        sk = np.array(sk, dtype=np.uint8, copy=True, order="C")
        out = np.empty_like(sk)
        f = self.lib.curve25519_dh_CalculatePublicKey_fast if fast else self.lib.curve25519_dh_CalculatePublicKey
        for i in range(sk.shape[0]):
            f(_p(out[i]), _p(sk[i]))
        return out, sk

    def ed25519_keypair(self, sk):
        sk = np.ascontiguousarray(sk, dtype=np.uint8)
        n = sk.shape[0]
        pub = np.empty((n, 32), np.uint8)
        priv = np.empty((n, 64), np.uint8)
        for i in range(n):
            self.lib.ed25519_CreateKeyPair(_p(pub[i]), _p(priv[i]), None, _p(sk[i]))
        return pub, priv

    def ed25519_sign(self, priv, msg):
        priv = np.ascontiguousarray(priv, dtype=np.uint8)
        n = priv.shape[0]
        msg = _rows(np.ascontiguousarray(msg, dtype=np.uint8), n)
        sig = np.empty((n, 64), np.uint8)
        for i in range(n):
            self.lib.ed25519_SignMessage(_p(sig[i]), _p(priv[i]), None, _p(msg[i]), msg.shape[1])
        return sig

    def ed25519_verify(self, sig, pk, msg):
        sig = np.ascontiguousarray(sig, dtype=np.uint8)
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        n = sig.shape[0]
        msg = _rows(np.ascontiguousarray(msg, dtype=np.uint8), n)
        ok = np.empty(n, np.int32)
        for i in range(n):
            ok[i] = self.lib.ed25519_VerifySignature(_p(sig[i]), _p(pk[i]), _p(msg[i]), msg.shape[1])
        return ok

    def verify_init_table(self, pk):
        """ed25519_Verify_Init(NULL, pk): the reference's own q_table, values reduced mod p, [n][16][4]."""
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        self.lib.ed25519_Verify_Init.argtypes = [C.c_void_p, u8p]
        self.lib.ed25519_Verify_Init.restype = C.c_void_p
        P = 2**255 - 19
        out = []
        for i in range(pk.shape[0]):
            ctx = np.zeros(2080, np.uint8)
            assert self.lib.ed25519_Verify_Init(ctx.ctypes.data, _p(pk[i])) == ctx.ctypes.data
            rows = ctx[32:].reshape(16, 4, 32)
            out.append([[int.from_bytes(rows[r, f].tobytes(), "little") % P for f in range(4)] for r in range(16)])
        return out

    # threaded drivers (C thread pool in liborc25519.so calling into the dlopen'ed reference)
    def _drv(self):
        build_oracle()
        d = C.CDLL(ORC_SO)
        d.orc_ref_x25519_shared_batch.argtypes = [C.c_char_p, u8p, u8p, u8p, C.c_size_t, C.c_int]
        d.orc_ref_ed25519_sign_batch.argtypes = [C.c_char_p, u8p, u8p, u8p, C.c_size_t, C.c_size_t, C.c_int]
        d.orc_ref_ed25519_verify_batch.argtypes = [C.c_char_p, C.POINTER(C.c_int32), u8p, u8p, u8p, C.c_size_t,
                                                   C.c_size_t, C.c_int]
        return d

    def x25519_shared_threaded(self, pk, sk, threads):
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        sk = np.array(sk, dtype=np.uint8, copy=True, order="C")
        out = np.empty_like(pk)
        rc = self._drv().orc_ref_x25519_shared_batch(self.so.encode(), _p(out), _p(pk), _p(sk), pk.shape[0], threads)
        assert rc == 0
        return out, sk

    def ed25519_sign_threaded(self, priv, msg, threads):
        priv = np.ascontiguousarray(priv, dtype=np.uint8)
        n = priv.shape[0]
        msg = _rows(np.ascontiguousarray(msg, dtype=np.uint8), n)
        sig = np.empty((n, 64), np.uint8)
        assert self._drv().orc_ref_ed25519_sign_batch(self.so.encode(), _p(sig), _p(priv), _p(msg), msg.shape[1], n, threads) == 0
        return sig

    def ed25519_verify_threaded(self, sig, pk, msg, threads):
        sig = np.ascontiguousarray(sig, dtype=np.uint8)
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
        n = sig.shape[0]
        msg = _rows(np.ascontiguousarray(msg, dtype=np.uint8), n)
        ok = np.empty(n, np.int32)
        assert self._drv().orc_ref_ed25519_verify_batch(self.so.encode(), ok.ctypes.data_as(C.POINTER(C.c_int32)), _p(sig),
                                                        _p(pk), _p(msg), msg.shape[1], n, threads) == 0
        return ok

    def base_table(self):
        t = (C.c_uint8 * (256 * 96)).in_dll(self.lib, "_w_base_folding8")
        return np.frombuffer(bytes(t), np.uint8).reshape(256, 3, 32).copy()
