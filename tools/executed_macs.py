#!/usr/bin/env python3
"""tools/executed_macs.py [--write profiles/rNN_executed_macs.json] -- v_mad_u64_u32 instructions the kernels EXECUTE per
operation, counted instead of estimated: the device source (curve25519_amd/csrc/*.cuh) compiled for the host against the C
model of the gfx950 primitives (tests/host_emul/valu_model.h), where every v_mad_u64_u32 goes through one counted function,
run one lane at a time on seeded inputs of the benchmark's distribution.  What a kernel adds around the per-lane code is
composed here the way engine.hip launches it:
  * the shared inversion (k_batch_invert, K = 16 elements per lane at 2^20): (one fe_invert + 3 (K - 1) products) / K per
    element plus the product(s) that apply 1/Z;
  * verification's walk starts at the first digit of the wave's LONGEST element: the points kernel sorts the elements so
    that waves start at digit 32 (elements above 32 collect at the back) -- the count runs every element from digit
    max(own, 32).
bench.py reads the committed JSON (roofline.valu.executed_macs_per_op); tests/test_bench_contract.py re-counts on the CPU
and compares.  No GPU needed."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "host_emul"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

INV_K = 16            # engine.hip: inversion_k(2^20)
WAVE_TOP = 32         # engine.hip: FastScratch::order -- waves of elements whose scalars start at digit 32 or below


def count(sample=64, base_comb=1):
    """base_comb: 1 = the wide 13 x 20 fixed-base comb (shipped default, engine.hip: base_comb_wide), 0 = the 8 x 32 LDS comb"""
    import build as emul_build
    from curve25519_amd import synth
    lib = C.CDLL(emul_build.build())
    vp, sz = C.c_void_p, C.c_size_t
    lib.emul_mad_count_take.restype = C.c_ulonglong
    p = lambda a: a.ctypes.data  # noqa: E731
    take = lambda: int(lib.emul_mad_count_take())  # noqa: E731
    n = sample
    lib.emul_set_base_comb(base_comb)

    def fe_op(op):                                     # one field operation on random operands
        a, b, out = synth.random_bytes((n, 32), 11), synth.random_bytes((n, 32), 12), np.empty((n, 32), np.uint8)
        lib.emul_fe_op.argtypes = [vp, vp, vp, sz, C.c_int]
        take()
        lib.emul_fe_op(p(out), p(a), p(b), n, op)
        return take() / n

    mul, sq, inv = fe_op(0), fe_op(1), fe_op(4)
    shared_inv = (inv + 3 * (INV_K - 1) * mul) / INV_K   # Montgomery's trick: prefix products, one inversion, two products back per element

    sk, pk = synth.x25519_inputs(n)
    out = np.empty((n, 32), np.uint8)
    lib.emul_x25519.argtypes = [vp, vp, vp, sz]
    take()
    lib.emul_x25519(p(out), p(pk), p(sk), n)
    x_all = take() / n                                   # ladder + a private inversion + the last product
    ladder = x_all - inv - mul
    x25519 = ladder + shared_inv + mul

    esk, msg = synth.ed25519_inputs(n)
    pub, priv = np.empty((n, 32), np.uint8), np.empty((n, 64), np.uint8)
    lib.emul_ed25519_keypair.argtypes = [vp, vp, vp, vp, sz]
    lib.emul_ed25519_keypair(p(pub), p(priv), None, p(esk), 1)      # (the first call builds the model's base tables)
    take()
    lib.emul_ed25519_keypair(p(pub), p(priv), None, p(esk), n)
    keypair = take() / n - inv + shared_inv              # the private inversion replaced by the shared one
    sig = np.empty((n, 64), np.uint8)
    lib.emul_ed25519_sign.argtypes = [vp, vp, vp, vp, sz, sz]
    take()
    lib.emul_ed25519_sign(p(sig), p(priv), None, p(msg), 32, n)
    sign = take() / n - inv + shared_inv

    ok, slow = np.empty(n, np.int32), np.empty(n, np.int32)
    lib.emul_ed25519_verify_fast_at.argtypes = [vp, vp, vp, vp, vp, sz, sz, C.c_int]
    lib.emul_ed25519_verify_fast_at(p(ok), p(slow), p(sig), p(pub), p(msg), 32, 1, WAVE_TOP)     # (builds the walk's comb table)
    take()
    lib.emul_ed25519_verify_fast_at(p(ok), p(slow), p(sig), p(pub), p(msg), 32, n, WAVE_TOP)
    verify = take() / n
    assert ok.all() and not slow.any()
    lib.emul_set_base_comb(0)
    return {"per_op": {"x25519": round(x25519), "sign": round(sign), "verify": round(verify)},
            "detail": {"fe_mul": mul, "fe_sq": sq, "fe_invert": inv, "x25519_ladder": round(ladder, 1),
                       "shared_inversion_per_element_K16": round(shared_inv, 1), "ed25519_keypair": round(keypair),
                       "fixed_base_comb": "13 x 20, four tables through L2" if base_comb else "8 x 32, eight tables in LDS",
                       "verify_wave_top_digit": WAVE_TOP, "sample_elements": n},
            "method": "device source on the C model of the gfx950 primitives, v_mad_u64_u32 counted (tools/executed_macs.py)"}


if __name__ == "__main__":
    r = count(base_comb=0 if "--lds-comb" in sys.argv else 1)
    print(json.dumps(r, indent=1))
    if "--write" in sys.argv:
        with open(sys.argv[sys.argv.index("--write") + 1], "w") as f:
            json.dump(r, f, indent=1)
