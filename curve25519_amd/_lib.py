"""ctypes view of the C-ABI library (include/curve25519_amd.h).  Fails loudly: there is no Python or
CPU implementation behind these calls -- if libcurve25519_amd.so is missing or no gfx950 device is
usable, the caller gets an exception, never a silently slower path."""
import ctypes as C
import os

from . import build as _build

_vp, _sz = C.c_void_p, C.c_size_t

# name -> argtypes ; every function returns int (0 = ok) unless listed in _RESTYPE
SIGNATURES = {
    "curve25519_dh_CreateSharedKey_batch": [_vp, _vp, _vp, _sz],
    "curve25519_dh_CreateSharedKey_dev": [_vp, _vp, _vp, _sz, _vp],
    "curve25519_dh_CalculatePublicKey_batch": [_vp, _vp, _sz],
    "curve25519_dh_CalculatePublicKey_dev": [_vp, _vp, _sz, _vp],
    "curve25519_dh_CalculatePublicKey_fast_batch": [_vp, _vp, _sz],
    "curve25519_dh_CalculatePublicKey_fast_dev": [_vp, _vp, _sz, _vp],
    "ed25519_CreateKeyPair_batch": [_vp, _vp, _vp, _sz],
    "ed25519_CreateKeyPair_dev": [_vp, _vp, _vp, _sz, _vp],
    "ed25519_SignMessage_batch": [_vp, _vp, _vp, _sz, _sz],
    "ed25519_SignMessage_dev": [_vp, _vp, _vp, _sz, _sz, _vp],
    "ed25519_SignMessage_ragged_batch": [_vp, _vp, _vp, _vp, _sz],
    "ed25519_SignMessage_ragged_dev": [_vp, _vp, _vp, _vp, _sz, _vp],
    "ed25519_VerifySignature_ragged_batch": [_vp, _vp, _vp, _vp, _vp, _sz],
    "ed25519_VerifySignature_ragged_dev": [_vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "ed25519_VerifySignature_batch": [_vp, _vp, _vp, _vp, _sz, _sz],
    "ed25519_VerifySignature_dev": [_vp, _vp, _vp, _vp, _sz, _sz, _vp],
    "ed25519_VerifySignature_scratch_bytes": [_sz],
    "c25519_amd_verify_last_slow_elements": [],
    "c25519_amd_verify_check_last_wide": [],
    "c25519_amd_host_register": [_vp, _sz],
    "c25519_amd_host_unregister": [_vp],
    "ed25519_Verify_Init_batch": [_vp, _vp, _sz],
    "ed25519_Verify_Init_dev": [_vp, _vp, _sz, _vp],
    "ed25519_Verify_Check_batch": [_vp, _vp, _vp, _vp, _sz, _sz],
    "ed25519_Verify_Check_dev": [_vp, _vp, _vp, _vp, _sz, _sz, _vp],
    "ed25519_Blinding_Init_dev": [_vp, _vp, _sz, _vp],
    "ed25519_CreateKeyPair_blinded_batch": [_vp, _vp, _vp, _vp, _sz],
    "ed25519_CreateKeyPair_blinded_dev": [_vp, _vp, _vp, _vp, _sz, _vp],
    "ed25519_SignMessage_blinded_batch": [_vp, _vp, _vp, _vp, _sz, _sz],
    "ed25519_SignMessage_blinded_dev": [_vp, _vp, _vp, _vp, _sz, _sz, _vp],
    "c25519_amd_multi_create": [_vp, _vp, C.c_int],
    "c25519_amd_multi_destroy": [_vp],
    "c25519_amd_multi_device_count": [_vp],
    "c25519_amd_multi_set_gather": [_vp, C.c_int],
    "c25519_amd_multi_helper_threads": [_vp],
    "c25519_amd_tunable_set": [C.c_char_p, C.c_long],
    "c25519_amd_tunable_get": [C.c_char_p],
    "c25519_amd_usable_cpus": [],
    "curve25519_dh_CreateSharedKey_multi": [_vp, _vp, _vp, _vp, _sz],
    "ed25519_SignMessage_multi": [_vp, _vp, _vp, _vp, _sz, _sz],
    "ed25519_VerifySignature_multi": [_vp, _vp, _vp, _vp, _vp, _sz, _sz],
    "c25519_amd_base_table": [_vp],
    "c25519_amd_sc_selftest": [_vp, _vp, _vp, _sz, C.c_int],
    "c25519_amd_fold_selftest": [_vp, _vp, _sz],
    "c25519_amd_thread_release": [],
    "c25519_amd_fe_selftest": [_vp, _vp, _vp, _sz, C.c_int],
    "c25519_amd_verify_point_dev": [_vp, _vp, _vp, _vp, _sz, _sz, _vp],
    "c25519_amd_device_count": [],
    "c25519_amd_set_device": [C.c_int],
    "c25519_amd_version": [],
    "c25519_amd_last_error": [],
    # the reference's own single-call API (include/curve25519_dh.h, include/ed25519_signature.h)
    "curve25519_dh_CalculatePublicKey": [_vp, _vp],
    "curve25519_dh_CalculatePublicKey_fast": [_vp, _vp],
    "curve25519_dh_CreateSharedKey": [_vp, _vp, _vp],
    "ed25519_CreateKeyPair": [_vp, _vp, _vp, _vp],
    "ed25519_SignMessage": [_vp, _vp, _vp, _vp, _sz],
    "ed25519_Blinding_Init": [_vp, _vp, _sz],
    "ed25519_Blinding_Finish": [_vp],
    "ed25519_VerifySignature": [_vp, _vp, _vp, _sz],
    "ed25519_Verify_Init": [_vp, _vp],
    "ed25519_Verify_Check": [_vp, _vp, _vp, _sz],
    "ed25519_Verify_Finish": [_vp],
}
_RESTYPE = {
    "ed25519_VerifySignature_scratch_bytes": _sz,
    "c25519_amd_verify_last_slow_elements": C.c_long,
    "c25519_amd_verify_check_last_wide": C.c_long,
    "c25519_amd_tunable_get": C.c_long,
    "c25519_amd_version": C.c_char_p,
    "c25519_amd_last_error": C.c_char_p,
    "ed25519_Blinding_Init": _vp,
    "ed25519_Verify_Init": _vp,
    "curve25519_dh_CalculatePublicKey": None,
    "curve25519_dh_CalculatePublicKey_fast": None,
    "curve25519_dh_CreateSharedKey": None,
    "ed25519_CreateKeyPair": None,
    "ed25519_SignMessage": None,
    "ed25519_Blinding_Finish": None,
    "ed25519_Verify_Finish": None,
    "c25519_amd_thread_release": None,
    "c25519_amd_multi_destroy": None,
}

_lib = None


class EngineError(RuntimeError):
    pass


def library_path() -> str:
    return _build.LIB


def load(build_if_missing: bool = True):
    """dlopen the in-tree library, building it first when hipcc is available and it is stale."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if build_if_missing and _build.is_stale() and (os.path.exists("/opt/rocm/bin/hipcc")):
        _build.build()
    if not os.path.exists(path):
        raise EngineError(f"{path} is missing: run `python -m curve25519_amd.build` (needs hipcc). "
                          "There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError here = the ABI lost a symbol: let it surface
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = lib
    return lib


def set_tunable(name: str, value: int):
    """A tuning / A-B knob of the library (include/curve25519_amd.h: c25519_amd_tunable_set); value < 0 = built-in choice."""
    check(load().c25519_amd_tunable_set(name.encode(), int(value)), f"c25519_amd_tunable_set({name})")


class tunable:
    """with tunable("COOP_MAX", 0): ...  -- the knob for the duration of the block, its previous value afterwards"""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = load().c25519_amd_tunable_get(self.name.encode())
        set_tunable(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_tunable(self.name, self.prev)
        return False


def check(rc: int, what: str):
    if rc != 0:
        msg = load().c25519_amd_last_error().decode(errors="replace")
        raise EngineError(f"{what} failed (rc={rc}): {msg}")
