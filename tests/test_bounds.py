"""CPU suite, part 5: worst-case limb bounds of the device field arithmetic.  tools/fe_bounds.py replays every
formula of the kernels on per-limb upper bounds and asserts that no 32-bit operand, 64-bit column or biased
subtraction can overflow for ANY input -- the random parity tests cannot show that."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fe_bound_contract_holds():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fe_bounds.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all bounds hold" in out.stdout


def test_checker_detects_a_violation():
    """The checker must actually bite: a multiplication whose operands are two subtractions deep overflows."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fe_bounds as fb
    R = fb.reduced_fixpoint()
    deep = fb.add(fb.sub(R, R), fb.sub(R, R))          # beta 6 on both sides
    try:
        fb.mul(deep, deep, "too deep")
    except fb.Bad:
        return
    raise AssertionError("bound checker accepted an overflowing multiplication")
