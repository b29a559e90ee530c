"""The two field representations the product did NOT choose (tools/ubench/field_forms.cuh: saturated 8 x 32 and
9 x 28.33-bit limbs) must be correct for the GPU A/B that times them (tools/ubench/field_ab.hip,
profiles/r02_field_ab.txt) to mean anything: field operations against Python big integers on the adversarial
patterns of tests/vectors.py, and a whole X25519 through each form's ladder step against the RFC 7748 / edge-key
vectors of the reference (tests/golden/kat.json)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import vectors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "host_emul")
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))


@pytest.fixture(scope="module")
def forms():
    lib = os.path.join(HERE, "libc25519_forms.so")
    srcs = [os.path.join(HERE, "forms.cpp"), os.path.join(HERE, "valu_model.h"),
            os.path.join(ROOT, "tools", "ubench", "field_forms.cuh"), os.path.join(ROOT, "curve25519_amd", "csrc", "fe25519.cuh")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas",
                               "-Wno-unused-function", "-include", os.path.join(HERE, "valu_model.h"),
                               "-I", os.path.join(ROOT, "curve25519_amd", "csrc"), "-I", os.path.join(ROOT, "tools", "ubench"),
                               os.path.join(HERE, "forms.cpp"), "-o", lib])
    L = C.CDLL(lib)
    L.forms_fe_op.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    L.forms_x25519.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return L


@pytest.mark.parametrize("form", [8, 9])
def test_alternative_field_forms_against_big_integers(forms, form):
    P = vectors.P
    pairs, a, b = vectors.field_cases()
    n = len(pairs)
    ops = {0: lambda x, y: x * y, 1: lambda x, y: x * x, 2: lambda x, y: x + y, 3: lambda x, y: x - y,
           4: lambda x, y: x + 121665 * y, 5: lambda x, y: pow(x, P - 2, P)}
    for op, f in ops.items():
        out = np.empty((n, 32), np.uint8)
        forms.forms_fe_op(out.ctypes.data, a.ctypes.data, b.ctypes.data, n, op, form)
        for i, (x, y) in enumerate(pairs):
            assert int.from_bytes(out[i].tobytes(), "little") == f(x, y) % P, (form, op, hex(x), hex(y))


@pytest.mark.parametrize("form", [8, 9])
def test_alternative_field_forms_run_x25519(forms, form):
    recs = KAT["x25519"]
    pk = np.concatenate([np.frombuffer(bytes.fromhex(r["pk"]), np.uint8).reshape(1, -1) for r in recs]).copy()
    sk = np.concatenate([np.frombuffer(bytes.fromhex(r["sk"]), np.uint8).reshape(1, -1) for r in recs]).copy()
    out = np.empty_like(pk)
    forms.forms_x25519(out.ctypes.data, pk.ctypes.data, sk.ctypes.data, len(recs), form)
    for i, r in enumerate(recs):
        assert out[i].tobytes().hex() == r["shared"], (form, r["name"])
