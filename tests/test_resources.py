"""Register / scratch budgets of the hot kernels, from the compiler's own remarks (hipcc -Rpass-analysis=kernel-resource-usage,
cross-compiled for gfx950 -- no GPU needed).  A spill in one of these kernels does not fail any parity test, it just
costs time on the device; round 2 shipped with several that nobody had asked the compiler about.  The same
parse prints profiles/rNN_resource_usage.txt (tools/resource_usage.py)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def usage():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    import resource_usage
    return {k["pretty"]: k for k in resource_usage.compile_remarks()}


# kernel (demangled prefix) -> the most registers it may allocate per lane (VGPR + AGPR): what its occupancy target allows
HOT = {
    "k_x25519_fused<false, 512>": 128, "k_x25519_fused<true, 512>": 128,
    "k_x25519_fused<false, 256>": 128, "k_x25519_fused<false, 128>": 128, "k_x25519_fused<false, 64>": 128,   # small batches
    "k_x25519_fused<true, 256>": 128, "k_x25519_fused<true, 128>": 128, "k_x25519_fused<true, 64>": 128,
    "k_x25519_ladder<false>": 128, "k_x25519_ladder<true>": 128,                                          # full batches
    "k_ed25519_verify_fast_scalars": 128, "k_ed25519_verify_fast_points": 168, "k_ed25519_verify_fast_walk": 256,
    "k_ed25519_verify_slow": 256,
    # the fixed-base kernels, <BLIND, WIDE>: the wide comb read through L2 (shipped) and the 8 x 32 comb staged in LDS
    "k_ed25519_sign_mult<false, true>": 128, "k_ed25519_sign_mult<true, true>": 128,
    "k_ed25519_sign_mult<false, false>": 128, "k_ed25519_sign_mult<true, false>": 128,
    "k_ed25519_keypair_mult<false, true>": 128, "k_ed25519_keypair_mult<true, true>": 128,
    "k_ed25519_keypair_mult<false, false>": 128, "k_ed25519_keypair_mult<true, false>": 128,
    "k_x25519_public_fast_mult<true>": 128, "k_x25519_public_fast_mult<false>": 128, "k_ed25519_sign_finish": 128,
    "k_ed25519_verify_check_shared": 168, "k_ed25519_verify_check<c25519::QTableLimbs>": 168,
}


def test_no_kernel_of_a_hot_pass_touches_scratch(usage):
    seen = 0
    for name, k in usage.items():
        hot = name in HOT or name.startswith("k_batch_invert<")
        if not hot:
            continue
        seen += 1
        assert k.get("scratch", 0) == 0, f"{name}: {k.get('scratch')} bytes of scratch per lane ({k.get('vgpr_spill')} VGPRs spilled)"
        if name in HOT:
            assert k["vgpr"] + k.get("agpr", 0) <= HOT[name], f"{name}: {k['vgpr']} + {k.get('agpr', 0)} registers, budget {HOT[name]}"
    assert set(HOT) <= set(usage), sorted(set(HOT) - set(usage))
    assert seen >= len(HOT) + 15                  # every listed kernel and the inversion's instantiations were found


def test_every_kernel_is_spill_free(usage):
    """... and nothing else in the library spills either (self-test hooks, table generation, the rare slow path)."""
    bad = {n: k["scratch"] for n, k in usage.items() if k.get("scratch", 0)}
    assert not bad, bad
