"""bench.py's roofline object and the issue model behind `valu.issue_model_frac`, checked on the CPU from the committed
profiles: the arithmetic the one-line JSON contract promises (achieved = algorithmic bytes / kernel time, frac = achieved /
peak, the VALU sub-object's fractions) and tools/cycle_probe.py's class-cost models of a ladder step."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_issue_models_of_a_ladder_step():
    cp = load(os.path.join(ROOT, "tools", "cycle_probe.py"), "c25519_cycle_probe")
    mad, half, vop2 = cp.STEP_MAD, cp.STEP_HALF, cp.STEP_FULL
    assert (mad, half, vop2) == (739, 188, 319)                      # profiles/r04_isa_mix.txt, the ladder's loop
    n4 = mad + half
    assert cp.model_cycles("nominal") == 4 * n4 + 2 * vop2 == 4346
    assert abs(cp.model_cycles("measured_vop2_paired") - (4.26 * n4 + 2.13 * vop2)) < 1e-6
    # the floor: the VOP2 work first fills the 0.26-cycle bubble every 4-cycle-class instruction leaves at four waves
    floor = 4.26 * n4 + max(0.0, 2.13 * vop2 - 0.26 * n4)
    assert abs(cp.model_cycles("floor") - floor) < 1e-6 and 4380 < floor < 4395


def test_roofline_object_from_the_committed_measurements():
    bench = load(os.path.join(ROOT, "bench.py"), "c25519_bench")
    n, ms = 1 << 20, 7.8
    probe = bench.issue_model(n, live=False)
    assert probe and "committed measurement" in probe["source"]
    r = bench.roofline_for("x25519", n, ms, probe)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["algorithmic_bytes_per_launch"] == 96 * n                 # pk + sk in, shared key out
    assert abs(r["achieved"] - 96 * n / (ms * 1e-3) / 1e9) < 1e-2 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-6
    assert r["kernel"] == "k_x25519_ladder + k_batch_invert<FinishX25519>"
    v = r["valu"]
    assert abs(v["achieved"] * 1e12 - v["algorithmic_macs_per_op"] * n / (ms * 1e-3)) / (v["achieved"] * 1e12) < 1e-3
    assert abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-3
    assert abs(v["frac_executed"] / v["frac"] - v["executed_macs_per_op"] / v["algorithmic_macs_per_op"]) < 1e-3
    assert 0.9 < v["issue_model_frac"] <= 1.02 and 4300 < v["simd_cycles_per_ladder_step"] < 5000
    assert v["ladder_step_instructions"] == {"v_mad_u64_u32": 739, "other_4_cycle_class": 188, "vop2": 319}
    # a batch the one-launch kernel runs names that kernel
    assert bench.roofline_for("x25519", 1 << 15, 0.7)["kernel"] == "k_x25519_fused"
    # traffic comes from the committed PMC passes of the same kernels, per pass
    with open(os.path.join(ROOT, "profiles", "r04_pmc.json")) as f:
        assert json.load(f)
    assert r["traffic"] is None or 2e8 < r["traffic"] < 4e8
