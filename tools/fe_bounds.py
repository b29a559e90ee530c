#!/usr/bin/env python3
"""Worst-case limb-bound checker for curve25519_amd/csrc/fe25519.cuh and every formula built on it.

The device field code keeps ten unsaturated limbs (radix 2^25.5) in 32-bit registers and 64-bit column
accumulators; it is only correct if no 32-bit operand, no 64-bit column and no biased subtraction ever
overflows / goes negative -- for ALL inputs, not just the random ones the parity tests throw at it.  This
script replays the formulas of x25519.cuh / ge25519.cuh / engine.hip on per-limb upper bounds (interval
arithmetic with exact Python integers) and asserts every such condition, starting from the worst inputs the
contract allows ("reduced" = whatever a mul/sqr carry chain can emit).  Run: python tools/fe_bounds.py
"""
M26, M25 = (1 << 26) - 1, (1 << 25) - 1
W = [26, 25] * 5
MASK = [M26, M25] * 5
P2 = [0x7FFFFDA] + [0x3FFFFFE, 0x7FFFFFE] * 4 + [0x3FFFFFE]
U32, U64 = (1 << 32) - 1, (1 << 64) - 1


class Bad(AssertionError):
    pass


def need(cond, what):
    if not cond:
        raise Bad(what)


def carry64(h, where):
    h = list(h)
    for i in range(10):
        need(h[i] <= U64, f"{where}: column {i} overflows 64 bits ({h[i].bit_length()} bits)")
    for i in range(9):
        h[i + 1] += h[i] >> W[i]
        need(h[i + 1] <= U64, f"{where}: carry into column {i+1} overflows")
        h[i] = min(h[i], MASK[i])
    c = h[9] >> 25
    h[9] = min(h[9], M25)
    h[0] += c * 19
    need(h[0] <= U64, f"{where}: wrap overflows")
    h[1] += h[0] >> 26
    h[0] = min(h[0], M26)
    need(all(x <= U32 for x in h), f"{where}: limb exceeds 32 bits after carry")
    return h


def mul(a, b, where="mul"):
    b19 = [19 * x for x in b]
    a2 = [2 * x for x in a]
    need(all(x <= U32 for x in b19[1:]), f"{where}: 19*b overflows 32 bits (max beta_b {max(b[i] / (1 << W[i]) for i in range(10)):.2f})")
    need(all(a2[i] <= U32 for i in range(1, 10, 2)), f"{where}: 2*a overflows 32 bits")
    h = [0] * 10
    for k in range(10):
        for i in range(10):
            j = (k - i) % 10
            x = a2[i] if (i & 1 and j & 1) else a[i]
            y = b19[j] if i > k else b[j]
            h[k] += x * y
    return carry64(h, where)


def sqr_columns(a, where):
    f2 = [2 * x for x in a]
    f19 = [19 * x for x in a]
    f38 = [38 * x for x in a]
    need(all(x <= U32 for x in f2), f"{where}: 2*a overflows")
    need(all(f19[j] <= U32 for j in range(5, 10)), f"{where}: 19*a overflows")
    need(all(f38[j] <= U32 for j in (5, 7, 9)), f"{where}: 38*a overflows (beta {a[9] / (1 << 25):.2f})")
    h = [0] * 10
    for k in range(10):
        for i in range(10):
            j = (k - i) % 10
            if i > j:
                continue
            wrap = i + j >= 10
            odd2 = i & 1 and j & 1
            if wrap and j & 1:
                x = f2[i] if (i < j and i & 1) else a[i]
                y = f38[j]
            else:
                x = f2[i] if i < j else a[i]
                y = f19[j] if wrap else (f2[j] if odd2 else a[j])
            h[k] += x * y
    return h


def sqr(a, where="sqr"):
    return carry64(sqr_columns(a, where), where)


def sqr_sub(a, m, where="sqr_sub"):
    h = sqr_columns(a, where)
    for i in range(10):
        need(m[i] <= 2 * P2[i], f"{where}: 4p bias smaller than subtrahend limb {i}")
        h[i] += 2 * P2[i]                       # upper bound of 4p - m is 4p
    return carry64(h, where)


def sqr2_add_sub(a, p, m, where="sqr2_add_sub"):
    h = sqr_columns(a, where)
    for i in range(10):
        need(m[i] <= P2[i], f"{where}: 2p bias smaller than subtrahend limb {i}")
        need(p[i] + P2[i] <= U32, f"{where}: 32-bit addend overflows")
        h[i] = 2 * h[i] + p[i] + P2[i]
    return carry64(h, where)


def chained_small_mul(mulitplicand, c, addend, where):
    """fe_mul121665_add / fe_mul_small: one MAD per limb, the previous limb's carry added to the 32-bit addend."""
    l, carry = [0] * 10, 0
    for i in range(10):
        t = addend[i] + carry
        need(t <= U32, f"{where}: addend + carry overflows 32 bits at limb {i}")
        h = mulitplicand[i] * c + t
        need(h <= U64, f"{where}: limb {i} overflows 64 bits")
        carry = h >> W[i]
        need(carry <= U32, f"{where}: carry out of limb {i} exceeds 32 bits")
        l[i] = min(h, MASK[i])
    t = l[0] + 19 * carry
    need(19 * carry <= U32 and t <= U32, f"{where}: wrap overflows 32 bits")
    l[1] += t >> 26
    l[0] = min(t, M26)
    need(all(x <= U32 for x in l), f"{where}: limb exceeds 32 bits")
    return l


def mul121665_add(a, b, where="mul121665_add"):
    return chained_small_mul(b, 121665, a, where)


def add(a, b, where="add"):
    r = [x + y for x, y in zip(a, b)]
    need(all(x <= U32 for x in r), f"{where}: 32-bit overflow")
    return r


def sub(a, b, where="sub"):
    need(all(b[i] <= P2[i] for i in range(10)), f"{where}: subtrahend limb exceeds 2p (would go negative)")
    r = [a[i] + P2[i] for i in range(10)]
    need(all(x <= U32 for x in r), f"{where}: 32-bit overflow")
    return r


def neg(a, where="neg"):
    need(all(a[i] <= P2[i] for i in range(10)), f"{where}: operand exceeds 2p")
    return list(P2)


def carry32(a, where="carry32"):
    h = list(a)
    for i in range(9):
        h[i + 1] += h[i] >> W[i]
        need(h[i + 1] <= U32, f"{where}: 32-bit overflow")
        h[i] = min(h[i], MASK[i])
    c = h[9] >> 25
    h[9] = min(h[9], M25)
    h[0] += 19 * c
    need(h[0] <= U32, f"{where}: 32-bit overflow")
    h[1] += h[0] >> 26
    h[0] = min(h[0], M26)
    return h


def to_words(a, where="to_words"):
    h = list(a)
    for _ in range(2):
        for i in range(9):
            h[i + 1] += h[i] >> W[i]
            need(h[i + 1] <= U32, f"{where}: 32-bit overflow")
            h[i] = min(h[i], MASK[i])
        c = h[9] >> 25
        h[9] = min(h[9], M25)
        h[0] += 19 * c
        need(h[0] <= U32, f"{where}: 32-bit overflow")
    need(h[0] <= M26 + 19, f"{where}: limb 0 not within 2^26+19 after two passes ({h[0]})")
    need(all(h[i] <= MASK[i] for i in range(1, 10)), f"{where}: limbs not strictly reduced after two passes")


def select(a, b):
    return [max(x, y) for x, y in zip(a, b)]


FROM_WORDS = [M26 + 19, M25, M26, M25, M26, M25, M26, M25, M26, M25]
ONE = [1] + [0] * 9
CANON = list(MASK)


def reduced_fixpoint():
    """Largest limbs any carry chain can emit, iterated until stable."""
    red = list(FROM_WORDS)
    big = [U32 // 19 if True else 0 for _ in range(10)]
    del big
    for _ in range(4):
        worst_in_a = [5 * (1 << W[i]) for i in range(10)]
        worst_in_b = [int(3.3 * (1 << W[i])) for i in range(10)]
        cands = [mul(worst_in_a, worst_in_b, "contract mul"), sqr(worst_in_b, "contract sqr"),
                 carry32([50 * (1 << W[i]) for i in range(10)], "contract carry32"),
                 mul121665_add(worst_in_a, worst_in_a, "contract a24")]
        new = [max([red[i]] + [c[i] for c in cands]) for i in range(10)]
        if new == red:
            break
        red = new
    return red


def beta(a):
    return max(a[i] / (1 << W[i]) for i in range(10))


def chain250(x, R):
    """fe_chain250 / fe_invert / fe_pow2523: every intermediate is a mul/sqr output, i.e. reduced."""
    x2 = sqr(x, "chain x2")
    t = sqr(sqr(x2, "c"), "c")
    x9 = mul(t, x, "chain x9")
    x11 = mul(x9, x2, "chain x11")
    for v in (x2, x9, x11):
        need(all(v[i] <= R[i] for i in range(10)), "chain output not reduced")
    # the remaining steps only combine reduced values
    mul(R, R, "chain mul")
    sqr(R, "chain sqr")
    mul(R, x, "chain final mul by x")
    return R


def main():
    R = reduced_fixpoint()
    print(f"reduced limb bound: beta <= {beta(R):.6f}  (limb1 <= 2^25 + {R[1] - (1 << 25)})")
    need(all(R[i] <= P2[i] for i in range(10)), "reduced value exceeds 2p: fe_sub bias too small")

    # ---- x25519.cuh ----
    X1 = FROM_WORDS
    SX, SZ, DX, DZ = R, select(R, ONE), R, R

    def ladder_step(SX, SZ, DX, DZ):
        A = sub(SX, SZ, "ladder A")
        B = add(SX, SZ)
        C = sub(DX, DZ, "ladder C")
        Dp = add(DX, DZ)
        P, M = select(Dp, B), select(C, A)
        A = mul(A, Dp, "ladder A*D")
        B = mul(C, B, "ladder C*B")
        C = add(A, B)
        B = sub(A, B, "ladder A-B")
        nSX = sqr(C, "ladder x3")
        A = sqr(B, "ladder (A-B)^2")
        nSZ = mul(A, X1, "ladder z3")
        nSZ9 = chained_small_mul(A, 9, [0] * 10, "ladder z3, base point u = 9 (fe_mul_small)")
        need(all(nSZ9[i] <= R[i] for i in range(10)), "fe_mul_small output not reduced")
        A = sqr(P, "ladder AA")
        B = sqr(M, "ladder BB")
        nDX = mul(A, B, "ladder x4")
        B = sub(A, B, "ladder E")
        A = mul121665_add(A, B, "ladder a24")
        nDZ = mul(B, A, "ladder z4")
        return nSX, nSZ, nDX, nDZ

    out = ladder_step(SX, SZ, DX, DZ)
    for v in out:
        need(all(v[i] <= R[i] for i in range(10)), "ladder output not reduced: loop invariant broken")
    chain250(R, R)
    to_words(mul(R, R, "x25519 final"), "x25519 to_words")
    print("x25519 ladder, inversion, encoding: ok")

    # ---- ge25519.cuh ----
    def ge_double(X, Y, Z):
        A, B = sqr(X, "dbl A"), sqr(Y, "dbl B")
        Hn = add(A, B)
        G = sub(B, A, "dbl G")
        t = add(X, Y)
        E = sqr_sub(t, Hn, "dbl E")
        Fn = sqr2_add_sub(Z, A, B, "dbl Fn")
        return mul(E, Fn, "dbl X"), mul(G, Hn, "dbl Y"), mul(G, Fn, "dbl Z"), mul(E, Hn, "dbl T")

    def ge_add(X, Y, Z, T, ypx, ymx, t2d, z2):
        a = mul(sub(Y, X, "add Y-X"), ymx, "add a")
        b = mul(add(Y, X), ypx, "add b")
        c = mul(T, t2d, "add c")
        d = add(Z, Z) if z2 is None else mul(Z, z2, "add d")
        e, h = sub(b, a, "add e"), add(b, a)
        f, g = sub(d, c, "add f"), add(d, c)
        if z2 is None:      # ge_add_pa operand order
            return mul(f, e, "add X"), mul(g, h, "add Y"), mul(f, g, "add Z"), mul(e, h, "add T")
        return mul(e, f, "add X"), mul(g, h, "add Y"), mul(f, g, "add Z"), mul(e, h, "add T")

    for v in ge_double(R, R, R) + ge_add(R, R, R, R, R, R, R, None) + ge_add(R, R, R, R, R, R, R, R):
        need(all(v[i] <= R[i] for i in range(10)), "point op output not reduced")
    # ge_to_pe / ge_from_pa / ge_from_pe
    for v in (carry32(add(R, R)), carry32(sub(R, R, "to_pe ymx")), mul(R, CANON, "to_pe t2d")):
        need(all(v[i] <= R[i] for i in range(10)), "ge_to_pe output not reduced")
    carry32(sub(R, R, "from_pa x"))
    # ge_calc_x
    u = carry32(sub(sqr(FROM_WORDS, "calc y^2"), ONE, "calc u"))
    v = add(mul(R, CANON, "calc d*y^2"), ONE)
    b = sqr(v, "calc v^2")
    a = mul(mul(u, b, "calc u v^2"), v, "calc u v^3")
    b = mul(a, sqr(b, "calc v^4"), "calc u v^7")
    chain250(b, R)
    X = mul(R, a, "calc x")
    to_words(sub(mul(sqr(X, "calc x^2"), v, "calc v x^2"), u, "calc check"), "calc check to_words")
    to_words(X, "calc x to_words")
    carry32(select(neg(X, "calc neg"), X))
    # table generation kernel and the public_fast tail
    carry32(add(CANON, CANON))
    carry32(sub(CANON, CANON, "table B ymx"))
    to_words(add(R, R), "table row to_words")
    to_words(sub(R, R, "table row ymx"), "table row to_words")
    num, den = add(R, R), carry32(sub(R, R, "fast den"))
    chain250(den, R)
    to_words(mul(num, R, "fast u"), "fast to_words")
    print("edwards double / add / decompress / table build / encodings: ok")

    # ---- verify_fast.cuh / blinding (lanes.cuh): rows negated on the fly, the on-curve check, the neutral-element test ----
    t2d_neg = select(neg(R, "pe_cond_neg t2d"), R)           # beta 2 where the row was negated
    for v in ge_add(R, R, R, R, R, R, t2d_neg, R):           # ge_add_pe with a conditionally negated row
        need(all(v[i] <= R[i] for i in range(10)), "add with a negated row: output not reduced")

    def ge_add_pe_row(X, Y, Z, T, ypx, ymx, t2d, z2):        # verify_fast.cuh: the streamed addition's operand order
        a = mul(sub(Y, X, "row add Y-X"), ymx, "row add a")
        b = mul(add(Y, X), ypx, "row add b")
        e, h = sub(b, a, "row add e"), add(b, a)
        c, d = mul(T, t2d, "row add c"), mul(Z, z2, "row add d")
        f, g = sub(d, c, "row add f"), add(d, c)
        return mul(e, f, "row add X"), mul(e, h, "row add T"), mul(g, f, "row add Z"), mul(g, h, "row add Y")

    def ge_add_pa_lds(X, Y, Z, T, ypx, ymx, t2d):            # ... and the streamed affine addition from the LDS table
        a = mul(sub(Y, X, "lds add Y-X"), ymx, "lds add a")
        b = mul(add(Y, X), ypx, "lds add b")
        e, h = sub(b, a, "lds add e"), add(b, a)
        c, d = mul(T, t2d, "lds add c"), add(Z, Z)
        f, g = sub(d, c, "lds add f"), add(d, c)
        return mul(f, e, "lds add X"), mul(e, h, "lds add T"), mul(f, g, "lds add Z"), mul(g, h, "lds add Y")

    for v in ge_add_pe_row(R, R, R, R, R, R, t2d_neg, R) + ge_add_pa_lds(R, R, R, R, CANON, CANON, CANON):
        need(all(v[i] <= R[i] for i in range(10)), "streamed addition: output not reduced")
    # packed table rows (fe_pack_words): what ge_store_pe_row packs -- fe_carry32 outputs and one product -- stays below
    # 2^256 as a positional sum, and what fe_from_words hands back is a legal second operand again
    POS = [0, 26, 51, 77, 102, 128, 153, 179, 204, 230]
    for v, what in ((carry32(add(R, R)), "row Y+X"), (carry32(sub(R, R, "row Y-X")), "row Y-X"), (mul(R, CANON, "row 2dT"), "row 2dT"),
                    (carry32(add(R, R)), "row 2Z")):
        need(sum(x << p for x, p in zip(v, POS)) < 1 << 256, f"packed {what} does not fit 256 bits")
    for v in ge_add_pe_row(R, R, R, R, FROM_WORDS, FROM_WORDS, select(neg(FROM_WORDS, "row neg"), FROM_WORDS), FROM_WORDS):
        need(all(v[i] <= R[i] for i in range(10)), "addition of an unpacked row: output not reduced")
    mul(t2d_neg, CANON, "from_pe T of a negated row")        # ge_from_pe: t2d is the FIRST operand (beta <= 5)
    to_words(add(mul(sqr(R, "check x^2"), add(R, ONE), "check v x^2"), R), "calc check c + u")      # ge_calc_x_checked
    to_words(sub(R, R, "neutral Y - Z"), "neutral test to_words")
    carry32(neg(R, "negate X / T of R and Q"))
    # ge_from_pa with the blinding context's random Z: all four coordinates times zr (reduced from words), Z = carry32(2 zr)
    for v in (mul(carry32(sub(R, R, "blind x")), FROM_WORDS, "blind X*zr"), mul(R, FROM_WORDS, "blind T*zr"),
              carry32(add(FROM_WORDS, FROM_WORDS))):
        need(all(v[i] <= R[i] for i in range(10)), "blinded start not reduced")
    # ---- ge25519.cuh signed comb (ge_base_mult): rows leave LDS negated when their column's top digit is -1 ----
    for v in ge_add(R, R, R, R, R, R, t2d_neg, None):        # ge_add_pa with a conditionally negated affine row
        need(all(v[i] <= R[i] for i in range(10)), "signed comb: add with a negated affine row: output not reduced")
    mul(t2d_neg, CANON, "signed comb: ge_from_pa T of a negated first row")
    carry32(neg(CANON, "signed comb table generation: -B's 2dxy"))
    print("lattice verification walk, signed comb and blinded base walk: ok")
    print("all bounds hold")


if __name__ == "__main__":
    main()
