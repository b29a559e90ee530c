#!/usr/bin/env python3
"""Generates the committed golden fixtures from the REAL reference (oracle/_ref, built from
/root/reference by `make -C oracle ref`).  Runs only in the build container; the fixtures it writes
are data (inputs + the reference's outputs), never reference source.

    python tests/golden/gen_golden.py            # kat.json, random_1024.npz, digests.json
    python tests/golden/gen_golden.py --no-big   # skip the 2^20 digests (minutes of CPU)
    python tests/golden/gen_golden.py --degenerate-only   # only degenerate_verify.npz (seconds)
    python tests/golden/gen_golden.py --ranks-only        # only digests.json["ranks"]: bench.py's ranks 0..7 at 2^20

Known-answer inputs come from RFC 7748 5.2, RFC 8032 7.1 and from the reference's own tests
(test/curve25519_test.c:412-445, test/openssl_test.c:20,97,138); the expected outputs are whatever the
reference computes -- where an RFC states an expected value the script asserts the reference agrees
(it does, except for RFC 7748 vector 2, whose peer key has bit 255 set: the reference does not mask
that bit).
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from curve25519_amd import synth  # noqa: E402
from oracle_lib import Reference  # noqa: E402

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
h2b = bytes.fromhex


def le32(v):
    return int(v).to_bytes(32, "little")


def arr(b):
    return np.frombuffer(b, np.uint8).reshape(1, -1)


def kat(ref):
    out = {"x25519": [], "ed25519": [], "ed25519_verify": [], "folds": [], "sha512": []}

    def x(name, sk, pk, expect=None):
        shared, clamped = ref.x25519_shared(arr(pk), arr(sk))
        rec = {"name": name, "sk": sk.hex(), "pk": pk.hex(), "shared": shared[0].tobytes().hex(),
               "sk_clamped": clamped[0].tobytes().hex()}
        if expect is not None:
            assert rec["shared"] == expect, (name, rec["shared"], expect)
        out["x25519"].append(rec)

    # RFC 7748 5.2
    x("rfc7748-1", h2b("a546e36bf0527c9d3b16154b82465edd62144c0ac1fc5a18506a2244ba449ac4"),
      h2b("e6db6867583030db3594c1a424b15f7c726624ec26b3353b10a903a6d0ab1c4c"),
      "c3da55379de9c6908e94ea4df28d084f32eccf03491c71f754b4075577a28552")
    k2 = h2b("4b66e9d4d1b4673c5ad22691957d6af5c11b6421e0ea01d42ca4169e7918ba0d")
    u2 = h2b("e5210f12786811d3f4b7959d0538ae2c31dbe7106fc03c3efc4cd549c715a493")
    x("rfc7748-2-unmasked(reference behaviour)", k2, u2)
    x("rfc7748-2-masked", k2, u2[:31] + bytes([u2[31] & 0x7f]),
      "95cbde9476e8907d7aade45cb4b873f88b595a68799fa152e6f8f7647aac7957")
    # edge-case peer keys with sk = 0x42 * 32
    sk42 = bytes([0x42]) * 32
    for name, v in [("pk=0", 0), ("pk=1", 1), ("pk=p-1", P - 1), ("pk=p", P), ("pk=p+1", P + 1),
                    ("pk=2^255-1", 2**255 - 1), ("pk=2^256-1", 2**256 - 1), ("pk=9", 9), ("pk=p+9", P + 9),
                    ("pk=2^255+9", 2**255 + 9), ("pk=2", 2), ("pk=2^255", 2**255)]:
        x(name, sk42, le32(v))
    # extreme secret keys (clamping decides the effective scalar)
    for name, sk in [("sk=0", bytes(32)), ("sk=ff", b"\xff" * 32), ("sk=01..", bytes(range(1, 33)))]:
        x(name, sk, le32(9))
    # reference test/curve25519_test.c:435-445 (Alice / Bruce) and test/openssl_test.c:20,97
    alice = h2b("03ac674216f3e15c761ee1a5e255f067953623c8b388b4459e13f978d7c846f4")
    bruce = h2b("88d4266fd4e6338d13b845fcf289579d209c897823b9217da3e161936f031589")
    apk, _ = ref.x25519_public(arr(alice))
    bpk, _ = ref.x25519_public(arr(bruce))
    x("alice*bruce_pk", alice, bpk[0].tobytes())
    x("bruce*alice_pk", bruce, apk[0].tobytes())
    assert out["x25519"][-1]["shared"] == out["x25519"][-2]["shared"]
    x("openssl_test", bytes(range(0x00, 0x20)), bytes(range(0x20, 0x40)))
    out["x25519_public"] = []
    for name, sk in [("alice", alice), ("bruce", bruce), ("sk42", sk42), ("openssl_test", bytes(range(32)))]:
        pk, cl = ref.x25519_public(arr(sk))
        pkf, _ = ref.x25519_public(arr(sk), fast=True)
        assert np.array_equal(pk, pkf)
        out["x25519_public"].append({"name": name, "sk": sk.hex(), "pk": pk[0].tobytes().hex(),
                                     "sk_clamped": cl[0].tobytes().hex()})

    def ed(name, sk, msg, pk_expect=None, sig_expect=None):
        pub, priv = ref.ed25519_keypair(arr(sk))
        if len(msg):
            sig = ref.ed25519_sign(priv, arr(msg))
            ok = ref.ed25519_verify(sig, pub, arr(msg))
        else:
            sig = ref.ed25519_sign(priv, np.zeros((1, 0), np.uint8))
            ok = ref.ed25519_verify(sig, pub, np.zeros((1, 0), np.uint8))
        rec = {"name": name, "sk": sk.hex(), "msg": msg.hex(), "pk": pub[0].tobytes().hex(),
               "priv": priv[0].tobytes().hex(), "sig": sig[0].tobytes().hex(), "verify": int(ok[0])}
        if pk_expect:
            assert rec["pk"] == pk_expect, (name, rec["pk"])
        if sig_expect:
            assert rec["sig"] == sig_expect, (name, rec["sig"])
        assert rec["verify"] == 1
        out["ed25519"].append(rec)
        return rec

    # RFC 8032 7.1 TEST 1-3 (TEST 2 is also test/curve25519_test.c:412-424)
    ed("rfc8032-test1", h2b("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60"), b"",
       "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a",
       "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46b"
       "d25bf5f0595bbe24655141438e7a100b")
    t2 = ed("rfc8032-test2", h2b("4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb"), h2b("72"),
            "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c",
            "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c"
            "387b2eaeb4302aeeb00d291612bb0c00")
    ed("rfc8032-test3", h2b("c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7"), h2b("af82"),
       "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025",
       "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac18ff9b538d16f290ae67f760984dc659"
       "4a7c15e9716ed28dc027beceea1ec40a")
    ed("openssl_test", bytes(range(0x00, 0x20)), bytes(range(0x40, 0x60)),
       "03a107bff3ce10be1d70dd18e74bc09967e4d6309ba50d5f1ddc8664125531b8")
    # message lengths around the SHA-512 block boundaries (prefix is 32 or 64 bytes)
    sk_len = h2b("000102030405060708090a0b0c0d0e0f101112131415161718191a1b1c1d1e1f")
    for n in (1, 7, 8, 31, 47, 48, 55, 63, 64, 79, 80, 95, 96, 111, 112, 127, 128, 129, 200, 255, 256, 257):
        ed(f"len{n}", sk_len, bytes((i * 7 + 3) & 0xff for i in range(n)))

    # negative / quirky verdicts
    pk, sig, msg = h2b(t2["pk"]), h2b(t2["sig"]), h2b("72")

    def v(name, sig_, pk_, msg_):
        ok = ref.ed25519_verify(arr(sig_), arr(pk_), arr(msg_) if len(msg_) else np.zeros((1, 0), np.uint8))
        out["ed25519_verify"].append({"name": name, "sig": sig_.hex(), "pk": pk_.hex(), "msg": msg_.hex(),
                                      "verify": int(ok[0])})
        return int(ok[0])

    assert v("valid", sig, pk, msg) == 1
    assert v("flipped-msg", sig, pk, h2b("73")) == 0
    assert v("flipped-R-bit", bytes([sig[0] ^ 1]) + sig[1:], pk, msg) == 0
    assert v("flipped-S-bit", sig[:32] + bytes([sig[32] ^ 1]) + sig[33:], pk, msg) == 0
    s_plus_l = (int.from_bytes(sig[32:], "little") + L).to_bytes(32, "little")
    assert v("S+L (non-canonical S is accepted by the reference)", sig[:32] + s_plus_l, pk, msg) == 1
    v("pk y=2 (off-curve, not rejected up front)", sig, le32(2), msg)
    v("pk=0", sig, bytes(32), msg)
    v("pk=ff", sig, b"\xff" * 32, msg)
    v("pk y=1 (neutral / small order)", sig, le32(1), msg)
    v("pk non-canonical y = p+3", sig, le32(P + 3), msg)
    v("sig=0", bytes(64), pk, msg)
    v("sig=ff", b"\xff" * 64, pk, msg)
    v("wrong-key", sig, h2b(out["ed25519"][0]["pk"]), msg)
    v("empty-msg-vs-test2-sig", sig, pk, b"")

    # fold recodings (curve25519_utils.c:125-153) on seeded scalars
    from oracle_lib import _p
    for t in range(16):
        k = synth.random_bytes((32,), 0xF01D + t)
        f8, f4 = np.empty(32, np.uint8), np.empty(64, np.uint8)
        ref.lib.ecp_8Folds(_p(f8), _p(k))
        ref.lib.ecp_4Folds(_p(f4), _p(k))
        out["folds"].append({"k": k.tobytes().hex(), "fold8": f8.tobytes().hex(), "fold4": f4.tobytes().hex()})

    # SHA-512 KATs the reference's self-test pins (test/curve25519_selftest.c:131-141)
    out["sha512"] = [{"msg": "abc".encode().hex(), "digest": hashlib.sha512(b"abc").hexdigest()},
                     {"msg_repeat": ["61", 1000000], "digest": hashlib.sha512(b"a" * 1000000).hexdigest()}]
    out["base_folding8_sha256"] = hashlib.sha256(ref.base_table().tobytes()).hexdigest()
    return out


# ---- seeded batches: digests of the reference's outputs -----------------------------------------------

def _chunk_x(args):
    lo, hi, n, shift = args
    ref = Reference()
    sk, pk = synth.x25519_inputs(n, seed_shift=shift)
    out, cl = ref.x25519_shared(pk[lo:hi], sk[lo:hi])
    return out, cl


def _chunk_ed(args):
    lo, hi, n, shift = args
    ref = Reference()
    sk, msg = synth.ed25519_inputs(n, seed_shift=shift)
    pub, priv = ref.ed25519_keypair(sk[lo:hi])
    sig = ref.ed25519_sign(priv, msg[lo:hi])
    return pub, priv, sig


def _chunk_v(args):
    lo, hi, sig, pub, msg = args
    ref = Reference()
    return ref.ed25519_verify(sig, pub, msg)


def seeded(n, pool, chunks=64, seed_shift=0):
    """Digests (and the arrays) of the reference's outputs on the seeded batch of n; seed_shift = the rank offset of
    bench.py's weak-scaling inputs (synth.rank_seed_shift)."""
    bounds = [(n * i // chunks, n * (i + 1) // chunks, n, seed_shift) for i in range(chunks)]
    xs = pool.map(_chunk_x, bounds)
    shared = np.concatenate([a for a, _ in xs])
    clamped = np.concatenate([b for _, b in xs])
    eds = pool.map(_chunk_ed, bounds)
    pub = np.concatenate([a for a, _, _ in eds])
    priv = np.concatenate([b for _, b, _ in eds])
    sig = np.concatenate([c for _, _, c in eds])
    _, msg = synth.ed25519_inputs(n, seed_shift=seed_shift)
    bsig, bmsg, bad = synth.corrupt_for_verify(sig, msg)
    ok = np.concatenate(pool.map(_chunk_v, [(lo, hi, bsig[lo:hi], pub[lo:hi], bmsg[lo:hi]) for lo, hi, _, _ in bounds]))
    assert np.array_equal(ok == 0, bad), "corrupted entries must be exactly the rejected ones"
    d = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    return {"n": n, "x25519_shared": d(shared), "x25519_sk_clamped": d(clamped), "ed25519_pub": d(pub),
            "ed25519_priv": d(priv), "ed25519_sig": d(sig), "ed25519_verdicts": d(ok.astype("<i4")),
            "verify_rejected": int(bad.sum())}, dict(shared=shared, clamped=clamped, pub=pub, priv=priv, sig=sig,
                                                     bsig=bsig, bmsg=bmsg, ok=ok.astype(np.int32))


PREFIXES = (1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20)


def rank_digests(n, world, pool):
    """digests.json["ranks"]: what bench.py's rank r (its own n elements from seed + 0x100 * r) must produce -- every
    pass's output buffer, and the three slices of the mixed workload (BASELINE.json configs[4]: contiguous thirds) --
    so that a bench line on any number of GPUs attests bit-exactness of what it timed.  The seeded streams are
    positional (element i's bytes do not depend on n), so the outputs of a batch of m < n are the first m rows of the
    batch of n: "prefix" holds the digests of the power-of-two batches bench.py's tests run at."""
    d = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    by_rank = []
    for r in range(world):
        shift = synth.RANK_SEED_STRIDE * r
        rec, full = seeded(n, pool, chunks=256, seed_shift=shift)
        rec["rank"], rec["seed_shift"] = r, shift
        rec["prefix"] = {}
        for m in PREFIXES:
            if m > n:
                continue
            (x0, x1), (s0, s1), (v0, v1) = synth.mixed_thirds(m)
            ok = full["ok"][:m].astype("<i4")
            rec["prefix"][str(m)] = {
                "x25519_shared": d(full["shared"][:m]), "x25519_sk_clamped": d(full["clamped"][:m]),
                "ed25519_pub": d(full["pub"][:m]), "ed25519_priv": d(full["priv"][:m]), "ed25519_sig": d(full["sig"][:m]),
                "ed25519_verdicts": d(ok), "verify_rejected": int((ok == 0).sum()),
                "mixed_thirds": {"x25519_shared": d(full["shared"][x0:x1]), "ed25519_sig": d(full["sig"][s0:s1]),
                                 "ed25519_verdicts": d(ok[v0:v1]), "verify_rejected": int((ok[v0:v1] == 0).sum())}}
        assert all(rec["prefix"][str(n)][k] == rec[k] for k in ("x25519_shared", "ed25519_sig", "ed25519_verdicts"))
        by_rank.append(rec)
        print(f"rank {r}: {rec['x25519_shared'][:16]}.. rejected {rec['verify_rejected']}", flush=True)
    return {"n": n, "seed_stride": synth.RANK_SEED_STRIDE, "prefixes": [m for m in PREFIXES if m <= n], "by_rank": by_rank}


def degenerate(ref):
    """degenerate_verify.npz: the inputs tests/vectors.py builds -- small-order keys and R's in every encoding the
    reference decodes, S in {0, L, 2L, 15L, 1}, mixed-order keys with small-order R, S and S + L -- with the verdicts of
    the REAL reference (ed25519_verify.c:179-197, :287-313).  Both back-ends of the reference must agree on them."""
    import vectors
    sig, pk, msg, label = vectors.degenerate_signature_cases()
    verdict = ref.ed25519_verify(sig, pk, msg).astype(np.int32)
    if Reference.available(asm=True):
        assert np.array_equal(verdict, Reference(asm=True).ed25519_verify(sig, pk, msg)), "portable C and asm64 disagree"
    assert 0 < verdict.sum() < len(verdict)
    np.savez_compressed(os.path.join(HERE, "degenerate_verify.npz"), sig=sig, pk=pk, msg=msg, label=label, verdict=verdict)
    print(f"wrote degenerate_verify.npz: {len(verdict)} cases, {int(verdict.sum())} accepted by the reference; by class "
          + ", ".join(f"{c}: {int(verdict[label == c].sum())}/{int((label == c).sum())}" for c in sorted(set(label.tolist()))))


def main():
    assert Reference.available(), "build oracle/_ref first: make -C oracle ref"
    ref = Reference()
    if "--ranks-only" in sys.argv:           # only digests.json["ranks"] (8 x 2^20 through the reference: ~10 min on 8 cores)
        path = os.path.join(HERE, "digests.json")
        with open(path) as f:
            dig = json.load(f)
        with mp.Pool(os.cpu_count()) as pool:
            dig["ranks"] = rank_digests(1 << 20, 8, pool)
        assert all(dig["ranks"]["by_rank"][0][k] == v for k, v in dig[str(1 << 20)].items()), "rank 0 = the world-1 batch"
        with open(path, "w") as f:
            json.dump(dig, f, indent=1)
        print("wrote digests.json[ranks]")
        return
    degenerate(ref)
    if "--degenerate-only" in sys.argv:
        return
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat(ref), f, indent=1)
    print("wrote kat.json")
    with mp.Pool(os.cpu_count()) as pool:
        dig = {}
        d1024, full = seeded(1024, pool, chunks=8)
        sk, pk = synth.x25519_inputs(1024)
        esk, msg = synth.ed25519_inputs(1024)
        np.savez_compressed(os.path.join(HERE, "random_1024.npz"), x_sk=sk, x_pk=pk, x_shared=full["shared"],
                            x_sk_clamped=full["clamped"], ed_sk=esk, ed_msg=msg, ed_pub=full["pub"],
                            ed_priv=full["priv"], ed_sig=full["sig"], v_sig=full["bsig"], v_msg=full["bmsg"],
                            v_ok=full["ok"])
        print("wrote random_1024.npz")
        dig["1024"] = d1024
        dig["4096"], _ = seeded(4096, pool)
        if "--no-big" not in sys.argv:
            dig["ranks"] = rank_digests(1 << 20, 8, pool)
            dig[str(1 << 20)] = {k: v for k, v in dig["ranks"]["by_rank"][0].items()
                                 if k not in ("rank", "seed_shift", "prefix")}
        with open(os.path.join(HERE, "digests.json"), "w") as f:
            json.dump(dig, f, indent=1)
        print("wrote digests.json")


if __name__ == "__main__":
    main()
