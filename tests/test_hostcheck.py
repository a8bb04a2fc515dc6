"""Host build of the DEVICE arithmetic headers (elastic_elgamal_amd/csrc/*.cuh) with -DEG_BOUNDCHECK,
checked against the oracle.  Test-only: proves the limb-bound discipline on every executed path and
lets the device math be validated without a GPU.  The product never loads this library."""
import ctypes as C
import os
import random
import subprocess
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent / "hostcheck"
L = 2**252 + 27742317777372353535851937790883648493
P = 2**255 - 19


@pytest.fixture(scope="module")
def hc():
    lib = HERE / "libhostcheck.so"
    srcs = [HERE / "hostcheck.cpp"] + list((HERE.parent.parent / "elastic_elgamal_amd" / "csrc").glob("*.cuh"))
    if not lib.exists() or any(s.stat().st_mtime > lib.stat().st_mtime for s in srcs):
        subprocess.check_call(
            ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DEG_BOUNDCHECK", "-fsanitize=undefined",
             "-fno-sanitize-recover=undefined", *os.environ.get("EG_HOSTCHECK_FLAGS", "").split(),
             "-o", str(lib), str(HERE / "hostcheck.cpp")]
        )
    return C.CDLL(str(lib))


def _b(n=32):
    return C.create_string_buffer(n)


@pytest.fixture(params=[5, 6, 7])
def teeth(hc, request):
    """The comb shape of the per-ballot tables (ge25519.cuh: Teeth<T>): 5 x 51 and 6 x 43 are the shapes the product instantiates (a plan
    picks one, host_plan.hpp: plan_teeth), 7 x 37 the third the templates allow."""
    hc.hc_set_teeth(request.param)
    yield request.param
    hc.hc_set_teeth(6)


def test_field_ops(hc):
    rnd = random.Random(1)
    edge = [0, 1, 2, P - 1, P - 2, 2**255 - 20, 2**254, (1 << 255) - 1, 19, 2**29 - 1, 2**57 - 1, 2**227, 2**228 - 1]
    vals = edge + [rnd.randrange(2**255) for _ in range(100)]
    for a in vals:
        out = _b()
        hc.hc_fe_roundtrip(a.to_bytes(32, "little"), out)
        assert int.from_bytes(out.raw, "little") == a % P
    for a in vals:
        for b in vals[:14]:
            out = _b(160)
            hc.hc_fe_ops(a.to_bytes(32, "little"), b.to_bytes(32, "little"), out)
            r = [int.from_bytes(out.raw[32 * i : 32 * i + 32], "little") for i in range(5)]
            assert r[0] == a * b % P
            assert r[1] == a * a % P
            assert r[2] == pow(a, P - 2, P)
            assert r[3] == (a + b) % P
            assert r[4] == (a - b) % P


def test_packed_field_elements(hc):
    """fe_pack8 / fe_unpack8 (the 128-byte table entries): any class-1 limb vector - limbs at their maxima, limb 1 a hair above, values
    just below and above p and 2^255 - comes back as the same field element, with every limb inside its bounds (asserted in the build)."""
    rnd = random.Random(12)
    width = [29 if i % 3 == 0 else 28 for i in range(9)]
    top = [(1 << w) - 1 for w in width]
    cases = [[0] * 9, top, [top[0], (1 << 28) + 2500] + top[2:], [1] + [0] * 8, [0] * 8 + [(1 << 28) - 1], [(1 << 29) - 19] + top[1:],
             [0] * 8 + [1 << 28]]
    for _ in range(300):
        cases.append([rnd.randrange(1 << w) for w in width])
    offs = [(85 * i + 2) // 3 for i in range(9)]
    assert offs == [0, 29, 57, 85, 114, 142, 170, 199, 227]
    for limbs in cases:
        out = _b()
        arr = (C.c_uint32 * 9)(*limbs)
        assert hc.hc_fe_pack_roundtrip(arr, out) == 1, limbs
        assert int.from_bytes(out.raw, "little") == sum(l << o for l, o in zip(limbs, offs)) % P


def test_scalar_ops(hc):
    rnd = random.Random(2)
    for _ in range(300):
        w = rnd.getrandbits(512).to_bytes(64, "little")
        out = _b()
        hc.hc_sc_from_wide(w, out)
        assert int.from_bytes(out.raw, "little") == int.from_bytes(w, "little") % L
    for w in [b"\xff" * 64, b"\x00" * 64, (L).to_bytes(64, "little"), (L - 1).to_bytes(64, "little"), (2**512 - 1 - 12345).to_bytes(64, "little")]:
        out = _b()
        hc.hc_sc_from_wide(w, out)
        assert int.from_bytes(out.raw, "little") == int.from_bytes(w, "little") % L
    vals = [0, 1, L - 1, L - 2, 2**252, 2**252 - 1] + [rnd.randrange(L) for _ in range(40)]
    for a in vals:
        out = _b()
        hc.hc_sc_neg(a.to_bytes(32, "little"), out)
        assert int.from_bytes(out.raw, "little") == (-a) % L
        for b in vals[:10]:
            c = rnd.randrange(L)
            hc.hc_sc_muladd(a.to_bytes(32, "little"), b.to_bytes(32, "little"), c.to_bytes(32, "little"), out)
            assert int.from_bytes(out.raw, "little") == (a * b + c) % L
    for a in vals[:16]:
        out = _b()
        hc.hc_sc_invert(a.to_bytes(32, "little"), out)
        assert int.from_bytes(out.raw, "little") == pow(a, L - 2, L)
    assert hc.hc_sc_is_canonical((L - 1).to_bytes(32, "little")) == 1
    assert hc.hc_sc_is_canonical(L.to_bytes(32, "little")) == 0
    assert hc.hc_sc_is_canonical((L + 1).to_bytes(32, "little")) == 0
    assert hc.hc_sc_is_canonical(b"\xff" * 32) == 0
    assert hc.hc_sc_is_canonical(b"\x00" * 32) == 1


def test_ristretto_codec_and_group(hc, oracle):
    rnd = random.Random(3)
    g = oracle.const_bytes(4)
    pts = [b"\x00" * 32, g] + [oracle.point_mul_generator(rnd.randrange(L).to_bytes(32, "little")) for _ in range(40)]
    for p in pts:
        out = _b()
        assert hc.hc_point_roundtrip(p, out) == 1 and out.raw == p
    # invalid encodings agree with the oracle
    bad = [b"\xff" * 32, (P).to_bytes(32, "little"), (1).to_bytes(32, "little"), (P - 1).to_bytes(32, "little")]
    bad += [rnd.getrandbits(255).to_bytes(32, "little") for _ in range(200)]
    for e in bad:
        out = _b()
        ok = hc.hc_point_roundtrip(e, out)
        want = oracle.point_roundtrip(e)
        assert bool(ok) == (want is not None)
        if want is not None:
            assert out.raw == want
    # prepared points (eg_points_prepare_device): the packed affine form round-trips, an undecodable encoding is prepared as the identity
    import ctypes as C
    for p in pts:
        out, prep = _b(), C.create_string_buffer(96)
        assert hc.hc_prepared_roundtrip(p, out, prep) == 1 and out.raw == p
    for e in bad[:40]:
        out, prep = _b(), C.create_string_buffer(96)
        if not hc.hc_prepared_roundtrip(e, out, prep):
            assert out.raw == b"\x00" * 32 and prep.raw == b"\x00" * 32 + b"\x01" + b"\x00" * 63
    for a, b in zip(pts[:20], pts[20:40]):
        out = _b()
        assert hc.hc_point_add(a, b, 0, out) == 1 and out.raw == oracle.point_add(a, b)
        assert hc.hc_point_add(a, b, 1, out) == 1 and out.raw == oracle.point_add(a, b, sub=True)
    # doubling through the unified addition (P + P) and adding the identity
    out = _b()
    assert hc.hc_point_add(g, g, 0, out) == 1 and out.raw == oracle.point_mul_generator((2).to_bytes(32, "little"))
    assert hc.hc_point_add(g, b"\x00" * 32, 0, out) == 1 and out.raw == g
    assert hc.hc_point_add(g, g, 1, out) == 1 and out.raw == b"\x00" * 32


def test_double_mul_generator(hc, oracle, teeth):
    rnd = random.Random(4)
    cases = []
    edge_scalars = [0, 1, 8, 16, L - 1, 2**252, 2**252 - 1, 0x8888888888888888888888888888888888888888888888888888888888888888 % L,
                    0x0777777777777777777777777777777777777777777777777777777777777777]
    pts = [oracle.point_mul_generator(rnd.randrange(L).to_bytes(32, "little")) for _ in range(6)] + [b"\x00" * 32]
    for k in edge_scalars:
        cases.append((k, pts[0], rnd.randrange(L)))
        cases.append((rnd.randrange(L), pts[1], k))
    for _ in range(24):
        cases.append((rnd.randrange(L), rnd.choice(pts), rnd.randrange(L)))
    for k, p, r in cases:
        out = _b()
        kb, rb = k.to_bytes(32, "little"), r.to_bytes(32, "little")
        assert hc.hc_double_mul_generator(kb, p, rb, out) == 1
        assert out.raw == oracle.point_double_mul_generator(kb, p, rb), (k, r)
        assert hc.hc_double_mul_generator_teeth(kb, p, rb, out) == 1      # signed comb over the base's table, every shape
        assert out.raw == oracle.point_double_mul_generator(kb, p, rb), (k, r)
        assert hc.hc_double_mul_generator_halved(kb, p, rb, out) == 1     # halved scalars + doubled encoder
        assert out.raw == oracle.point_double_mul_generator(kb, p, rb), (k, r)


def test_comb_widths(hc, oracle):
    """ge_fixed_mul_add takes the window width from the table: every width the engine may use gives the oracle's [r]G, also for
    scalars whose digits sit on the extremes of the signed windows."""
    rnd = random.Random(46)
    for bits in (8, 15, 20, 22, 24, 26):
        half = 1 << (bits - 1)
        edge = [0, 1, half - 1, half, half + 1, (1 << bits) - 1, L - 1, 2**252, sum(half << (bits * w) for w in range(254 // bits)) % L]
        for r in edge + [rnd.randrange(L) for _ in range(6)]:
            out = _b()
            rb = r.to_bytes(32, "little")
            hc.hc_mul_generator_bits(bits, rb, out)
            assert out.raw == oracle.point_mul_generator(rb), (bits, r)


def test_shared_chain_multi_mul(hc, oracle, teeth):
    """ge_teeth_mul_multi (Straus over teeth tables, one doubling chain for all terms) against the oracle's multi_mul."""
    rnd = random.Random(44)
    pts = [oracle.point_mul_generator(rnd.randrange(L).to_bytes(32, "little")) for _ in range(9)] + [b"\x00" * 32]
    edge = [0, 1, 2, L - 1, L - 2, 2**252, 2**252 + 2**251 + 12345, 8]      # incl. halved odd scalars (> l) and even ones
    for n in (1, 2, 3, 7):
        for trial in range(6):
            ks = [edge[(trial + i) % len(edge)] if trial < 3 else rnd.randrange(L) for i in range(n)]
            ps = [pts[(trial * 3 + i) % len(pts)] for i in range(n)]
            r = rnd.randrange(L)
            out = _b()
            kb = b"".join(k.to_bytes(32, "little") for k in ks)
            assert hc.hc_multi_mul_teeth(n, kb, b"".join(ps), r.to_bytes(32, "little"), out) == 1
            want = oracle.point_multi_mul(kb + r.to_bytes(32, "little"), b"".join(ps) + oracle.const_bytes(4))
            assert out.raw == want, (n, trial)


def test_sum_of_tables(hc, oracle, teeth):
    """ge_teeth_tables_sum: the comb table of a sum of bases, made from the members' tables without doublings, holds the same
    curve points as a table built from the sum, and a product over it equals the oracle's (incl. the identity as a member, equal
    members, a member and its negative: sums that hit the identity)."""
    rnd = random.Random(45)
    pts = [oracle.point_mul_generator(rnd.randrange(L).to_bytes(32, "little")) for _ in range(9)] + [b"\x00" * 32]
    neg0 = oracle.point_mul_generator((L - 1).to_bytes(32, "little"))
    g1 = oracle.point_mul_generator((1).to_bytes(32, "little"))
    sets = [[pts[0]], pts[:2], pts[:3], pts[:5], pts[2:9], [pts[9], pts[1]], [pts[3], pts[3], pts[3]], [g1, neg0], [g1, neg0, pts[4]],
            pts[:9] + pts[:7]]
    edge = [0, 1, 2, L - 1, 2**252 + 2**251 + 12345, 8]
    for si, ps in enumerate(sets):
        for trial in range(3):
            k = edge[(si + trial) % len(edge)] if trial < 2 else rnd.randrange(L)
            r = rnd.randrange(L)
            out = _b()
            kb, rb = k.to_bytes(32, "little"), r.to_bytes(32, "little")
            assert hc.hc_sum_table_mul(len(ps), b"".join(ps), kb, rb, out) == 1, (si, trial)
            want = oracle.point_multi_mul(kb * len(ps) + rb, b"".join(ps) + oracle.const_bytes(4))
            assert out.raw == want, (si, trial)


def test_sum_of_tables_group_by_group(hc, oracle, teeth):
    """Ring-group walk (ge_teeth_sum_accumulate + ge_teeth_tables_sum over the accumulator): the table of a sum whose members' tables
    exist one group at a time holds the same points as the table summed from all members at once, for every group size, including
    members that cancel (identity partial sums inside and across groups)."""
    rnd = random.Random(46)
    pts = [oracle.point_mul_generator(rnd.randrange(L).to_bytes(32, "little")) for _ in range(9)] + [b"\x00" * 32]
    neg0 = oracle.point_mul_generator((L - 1).to_bytes(32, "little"))
    g1 = oracle.point_mul_generator((1).to_bytes(32, "little"))
    sets = [[pts[0]], pts[:2], pts[:5], pts[:9], [g1, neg0], [g1, neg0, pts[4], pts[9]], [pts[3], pts[3], pts[3], pts[3]], pts[:9] + pts[:7]]
    for si, ps in enumerate(sets):
        for group in (1, 2, 3, 4, len(ps)):
            k, r = (rnd.randrange(L) if (si + group) % 3 else [0, 1, L - 1][group % 3]), rnd.randrange(L)
            out = _b()
            kb, rb = k.to_bytes(32, "little"), r.to_bytes(32, "little")
            assert hc.hc_sum_table_grouped(len(ps), group, b"".join(ps), kb, rb, out) == 1, (si, group)
            want = oracle.point_multi_mul(kb * len(ps) + rb, b"".join(ps) + oracle.const_bytes(4))
            assert out.raw == want, (si, group)


def test_merlin(hc, oracle):
    out = _b(64)
    hc.hc_merlin.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_uint64, C.c_char_p, C.c_char_p]
    # upstream KAT prefix
    hc.hc_merlin(b"test protocol", b"some label", b"some data", 9, None, 0, b"challenge", out)
    m = oracle.Merlin(b"test protocol"); m.append(b"some label", b"some data")
    assert out.raw == m.challenge(b"challenge", 64)
    rnd = random.Random(5)
    for n in [0, 1, 31, 32, 64, 100, 165, 166, 167, 400]:
        msg = bytes(rnd.getrandbits(8) for _ in range(n))
        pos = hc.hc_merlin(b"encrypted_choice_ranges", b"enc", msg, n, b"i", 7, b"c", out)
        m = oracle.Merlin(b"encrypted_choice_ranges"); m.append(b"enc", msg); m.append_u64(b"i", 7)
        assert pos == m.pos
        assert out.raw == m.challenge(b"c", 64)


def test_doubled_encoder(hc, oracle):
    """encode(2P) with the rational inverse square root (one inversion) equals the plain encoder, including the
    degenerate points where N = 0 (identity, and points with X = 0 or Y = 0 up to torsion)."""
    rnd = random.Random(6)
    pts = [b"\x00" * 32, oracle.const_bytes(4)] + [oracle.point_mul_generator(rnd.randrange(L).to_bytes(32, "little")) for _ in range(200)]
    for p in pts:
        out = _b()
        assert hc.hc_double_encode(p, out) == 1, p.hex()
        assert out.raw == oracle.point_add(p, p)


def test_bench_work_model_matches_the_code(hc):
    """bench.py prices a ballot with per-building-block (fe_mul, fe_sq) counts; they must be the counts of the shipped code."""
    import ast
    names = ["decode", "direct_table", "direct_mul", "comb", "encode", "base_table", "base_mul", "enc_batch_each", "enc_batch_inversion",
             "multi_first", "multi_extra", "sum_table_first", "sum_table_extra", "comb_wide"]
    src = (HERE.parent.parent / "bench.py").read_text()
    tree = ast.parse(src)
    ops = next(ast.literal_eval(n.value) for n in tree.body
               if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "OPS")
    for t in (5, 6):                                   # the table-backed blocks depend on the comb shape of the plan
        hc.hc_set_teeth(t)
        out = (C.c_ulonglong * 28)()
        hc.hc_op_counts(out)
        got = {n: (out[2 * i], out[2 * i + 1]) for i, n in enumerate(names)}
        assert ops[t] == got, t
    hc.hc_set_teeth(6)
