"""The product's pure-host C++ (plan builders + flattening + index check of csrc/host_plan.hpp, native wire ingest of
csrc/wire_json.hpp) compiled with -fsanitize=address,undefined and driven through every plan shape and through fuzzed JSON.
The sanitizer build runs in a child process (ASan must be loaded first), so a finding shows up as a non-zero exit code."""
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent / "hostcheck"
ROOT = HERE.parent.parent


@pytest.fixture(scope="module")
def lib():
    so = HERE / "libplancheck.so"
    srcs = [HERE / "plancheck.cpp", ROOT / "elastic_elgamal_amd" / "csrc" / "host_plan.hpp", ROOT / "elastic_elgamal_amd" / "csrc" / "wire_json.hpp",
            ROOT / "elastic_elgamal_amd" / "csrc" / "plan.h"]
    if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=all", "-o", str(so), str(HERE / "plancheck.cpp")])
    return so


def _run(lib, body: str):
    asan = subprocess.check_output(["g++", "-print-file-name=libasan.so"], text=True).strip()
    code = textwrap.dedent(f"""
        import ctypes as C, json, random, sys
        sys.path.insert(0, {str(ROOT)!r})
        L = C.CDLL({str(lib)!r})
        L.pc_qv_size.restype = C.c_ulonglong
    """) + textwrap.dedent(body)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env={"LD_PRELOAD": asan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=99", "PATH": "/usr/bin:/bin"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    return r.stdout


def test_every_plan_shape_builds_flattens_and_checks(lib):
    out = _run(lib, """
        total = 0
        for n in list(range(1, 41)) + [64, 150, 1000, 4000]:
            for kind in (0, 1):
                j = L.pc_plan(kind, n, C.c_ulonglong(0)); assert j == 4 * n + (4 if kind == 0 else 0), (kind, n, j); total += 1
        # the ring-group walk with every group size (v - 1 rings per group; 0 = every table at once): same equations, consistent indices
        for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 17, 33, 150):
            for kind in (0, 1):
                for rg in (0, 1, 2, 3, 4, 5, 8, n - 1, n, n + 1):
                    if rg < 0: continue
                    j = L.pc_plan(kind, n, C.c_ulonglong(rg + 1)); assert j == 4 * n + (4 if kind == 0 else 0), (kind, n, rg, j); total += 1
        for n, credits in [(1, 1), (2, 4), (3, 9), (5, 15), (5, 20), (5, 25), (4, 100), (8, 36), (12, 200), (16, 64), (20, 1000), (3, 10000), (256, 100000)]:
            assert L.pc_plan(2, n, C.c_ulonglong(credits)) > 0, (n, credits); total += 1
        for ub in list(range(2, 300)) + [1000, 65536, 777777, 1000000]:
            assert L.pc_plan(5, 0, C.c_ulonglong(ub)) > 0, ub; total += 1
        for n in (1, 2, 5, 9, 40, 1000):
            assert L.pc_plan(6, n, C.c_ulonglong(0)) == 2 * n + 2; total += 1
        assert L.pc_plan(3, 0, C.c_ulonglong(0)) == 2 and L.pc_plan(4, 0, C.c_ulonglong(0)) == 4 and L.pc_plan(7, 3, C.c_ulonglong(0)) == 2
        buf = C.create_string_buffer(256)
        for ub, want in ((5, "0..5"), (21, "3 * 0..7 + 0..3"), (100, "20 * 0..5 + 4 * 0..5 + 0..4"), (12345678, None)):
            assert L.pc_range(C.c_ulonglong(ub), buf, 256) > 0
            if want: assert buf.value.decode() == want, buf.value
        print("plans", total)
    """)
    assert "plans" in out


def test_index_check_catches_terms_over_another_groups_tables(lib):
    """in a grouped plan (ring-group walk) a table-backed term names a table SLOT; a term bent onto the base of another
    group that sits in the same slot passes every range check and would silently read a different ring's table.  check_flat_plan now
    derives which group's tables are resident at each stage and refuses it (and a term that names another slot of its own group)."""
    out = _run(lib, """
        why = C.create_string_buffer(200)
        seen = set()
        for n, single, rg in ((5, 1, 2), (5, 1, 1), (16, 0, 4), (9, 1, 4), (40, 0, 8), (6, 1, 3)):
            assert L.pc_plan_mutated(n, single, rg, 0, why, 200) == 0, (n, single, rg, why.value)
            for mutation in (1, 2):
                r = L.pc_plan_mutated(n, single, rg, mutation, why, 200)
                if rg == 1 and mutation == 2:
                    assert r in (1, -2)          # one ring per group: R and B are its only two slots
                assert r == 1 or (r == -2 and rg == 1), (n, single, rg, mutation, r, why.value)
                if r == 1: seen.add(why.value.decode())
        print(sorted(seen))
    """)
    assert "not resident" in out and "another base's table slot" in out, out


def test_wire_packer_survives_fuzzed_json(lib):
    out = _run(lib, """
        from elastic_elgamal_amd import serde, ingest
        gold = json.load(open({0!r}))
        c, q = gold["encrypted-choice"], gold["qv-ballot"]
        rnd = random.Random(5)
        texts = []
        base = json.dumps([c, c, c])
        qbase = json.dumps(q, indent=1)
        for t, src in ((0, base), (1, qbase)):
            for _ in range(1500):
                b = bytearray(src.encode())
                for _ in range(rnd.randrange(1, 6)):
                    if len(b) < 2: break
                    op = rnd.randrange(4)
                    i = rnd.randrange(len(b))
                    if op == 0: b[i] = rnd.randrange(256)
                    elif op == 1: del b[i : i + rnd.randrange(1, 60)]
                    elif op == 2: b[i:i] = bytes(rnd.choice(b'{{}}[]",:\\\\ =') for _ in range(rnd.randrange(1, 5)))
                    else: b = b[: rnd.randrange(len(b))]
                texts.append((t, bytes(b)))
        ok = bad = 0
        qsize = L.pc_qv_size(5, C.c_ulonglong(15))
        for t, b in texts:
            packed = C.create_string_buffer(16 * max(736, qsize)); st = (C.c_uint32 * 16)()
            if t == 0: n = L.pc_pack_choice(5, 1, b, C.c_size_t(len(b)), 3, packed, st, C.c_size_t(16))
            else: n = L.pc_pack_qv(5, C.c_ulonglong(15), b, C.c_size_t(len(b)), 5, packed, st, C.c_size_t(16))
            assert n != -7, ("the parallel and the sequential splitter disagree on", b)
            assert n != -8, ("the pooled and the spawned packer disagree on", b)
            for window in (1, 7, 64, 700, 1500, 5000, 1 << 20):
                assert L.pc_split_windows(b, C.c_size_t(len(b)), C.c_size_t(window), 1 + window % 4) == 1, ("streaming splitter differs", window, b)
            if n < 0: bad += 1
            else: ok += sum(1 for k in range(n) if st[k] == 0)
        print("fuzzed", len(texts), "unsplittable", bad, "objects still accepted", ok)
        # and the untouched texts pack to the golden bytes under the sanitizers too
        packed = C.create_string_buffer(3 * 736); st = (C.c_uint32 * 3)()
        assert L.pc_pack_choice(5, 1, base.encode(), C.c_size_t(len(base)), 3, packed, st, C.c_size_t(3)) == 3 and list(st) == [0, 0, 0]
        assert packed.raw[:736] == serde.pack_encrypted_choice(c)
    """.format(str(ROOT / "tests" / "golden" / "snapshots_serde.json")))
    assert "fuzzed 3000" in out


def test_streaming_splitter_agrees_with_the_one_shot_splitter(lib):
    """eg_verify_json_begin / _feed / _end below the C ABI (csrc/wire_json.hpp: StreamSplitter), under ASan + UBSan: texts (arrays, sequences,
    one object per line; escaped quotes, brackets and backslashes inside strings, nested junk; truncated, unbalanced and garbage
    variants) fed in pieces cut at random offsets, at EVERY offset of a short text, one byte at a time and as one piece: the values
    emitted are byte for byte those of split_objects on the whole text, and a text is refused exactly when split_objects refuses it."""
    out = _run(lib, r"""
        L.pc_stream_split.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_size_t, C.c_size_t]
        rnd = random.Random(5)
        def obj(i):
            tricky = ['a"b', 'x\\\\', 'br{ack}[et]s', '\\\\"', 'u\\u0041', '', ']}', '\\\\\\"{']
            return json.dumps({"k": i, "s": tricky[i % len(tricky)], "n": {"a": [1, {"b": "}"}], "c": "["}, "pad": "x" * rnd.randrange(0, 40)})
        objs = [obj(i) for i in range(40)]
        texts = ["[" + ", ".join(objs) + "]", " [\n" + ",\n".join(objs) + "\n]  \n", "\n".join(objs), "".join(objs), " ".join(objs) + "\n",
                 "[]", "", "  ", "[ ]", "[" + objs[0] + "]", objs[0]]
        bad = ["[" + ", ".join(objs[:5]) + ",]", "[" + ", ".join(objs[:5]), ", ".join(objs[:5]), "[" + " ".join(objs[:5]) + "]", objs[0][:-1],
               "[" + objs[0] + "] x", "[" + objs[0] + "]]", "x" + objs[0], "[[" + objs[0] + "]]", objs[0] + "}", "[" + objs[0] + ",," + objs[1] + "]",
               objs[0] + "\\" + objs[1], '{"a": "unterminated', "[1, 2]", "]"]
        total = refused = 0
        def run(t, cuts, threads=3, window=64, maxv=5):
            b = t.encode()
            arr = (C.c_size_t * max(len(cuts), 1))(*cuts)
            return L.pc_stream_split(b, len(b), arr, len(cuts), threads, window, maxv)
        for t in texts + bad:
            n = len(t.encode())
            want = run(t, [])
            assert want != -7 and want != -9, t[:60]
            assert (want >= 0) == (t in texts), (t[:60], want)
            plans = [[], list(range(1, n)), sorted(rnd.sample(range(n + 1), min(n + 1, 7)))] + [sorted(rnd.randrange(n + 1) for _ in range(rnd.randrange(1, 30))) for _ in range(25)]
            if n <= 400:
                plans += [[k] for k in range(n + 1)] + [[k, k] for k in range(0, n + 1, 3)]
            for cuts in plans:
                for threads, window, maxv in ((3, 64, 5), (1, 1 << 20, 1 << 20), (4, 7, 1)):
                    got = run(t, cuts, threads, window, maxv)
                    assert got == want, (t[:60], cuts[:10], threads, window, maxv, got, want)
                    total += 1; refused += got < 0
        # ballots of the reference's own shape, many pieces
        snaps = json.loads(open(@@SNAP@@).read())
        c = json.dumps(snaps["encrypted-choice"]); q = json.dumps(snaps["qv-ballot"])
        big = "[" + ",".join([c, q] * 300) + "]"
        n = len(big)
        for k in range(12):
            cuts = sorted(rnd.randrange(n + 1) for _ in range(rnd.choice((1, 10, 200, 3000))))
            assert run(big, cuts, 4, 5000, 50) == 600
            total += 1
        print("stream split cases", total, "refused", refused)
    """.replace("@@SNAP@@", repr(str(ROOT / "tests" / "golden" / "snapshots_serde.json"))))
    assert "stream split cases" in out and "refused 0" not in out
    assert int(out.split("stream split cases")[1].split()[0]) > 5000, out


def test_worker_pool_under_thread_sanitizer(tmp_path):
    """The parser's worker pool (egwire::WorkerPool: split passes and packing of every window of eg_verify_*_json) in a
    -fsanitize=thread build: 3000 objects cut and packed on 8 threads, windows of 5 kB, five times over; any data race fails the run."""
    exe = tmp_path / "poolcheck"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-o", str(exe),
                           str(HERE / "poolcheck.cpp"), str(HERE / "plancheck.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env={"TSAN_OPTIONS": "halt_on_error=1 exitcode=66", "PATH": "/usr/bin:/bin"})
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-6000:]
    assert r.stdout.count("n=3000 split=1") == 5


@pytest.mark.parametrize("kind", ["single", "multi", "qv"])
def test_native_object_path_agrees_with_the_oracle_on_objects(lib, kind):
    """The object path BELOW the C ABI (csrc/wire_json.hpp: resolve_*_objects, what eg_verify_*_json runs for ballots whose shape is not
    the election's): the library's JSON entry point with its two GPU services answered by the oracle must give, for the scenario
    table and for 80 randomly reshaped / tampered objects, the verdicts of oracle/objects.c (the reference's verify() on objects:
    deserialisation order, check_options_count, every LenMismatch check in the reference's order) - under ASan + UBSan."""
    out = _run(lib, f"""
        kind = {kind!r}
        sys.path.insert(0, {str(ROOT / 'tests')!r})
        import copy
        from oracle import oracle
        from elastic_elgamal_amd import ingest, serde
        from ingest_cases import BAD_POINT, BAD_SCALAR, choice_cases, flip, qv_cases
        from test_ingest_cpu import _choice_object_verdict, _qv_object_verdict
        golden = json.loads(open({str(ROOT / 'tests' / 'golden' / 'snapshots_ristretto.json')!r}).read())
        import base64
        s = golden["public_key_b64"]; pk = base64.urlsafe_b64decode(s + "=" * (-len(s) % 4))
        Lq = 2**252 + 27742317777372353535851937790883648493
        rnd = random.Random(11)
        n = 3
        if kind == "qv":
            credits = 9
            op = oracle.QvParams(pk, n, credits)
            packed = op.generate_batch(12, 0, 8); sz = len(packed) // 8
            objs = [ingest.unpack_qv_ballot(packed[i * sz:(i + 1) * sz], n, credits) for i in range(8)]
            cases, verdict = qv_cases(objs), _qv_object_verdict
        else:
            single = kind == "single"
            op = oracle.ChoiceParams(pk, n, single)
            packed = op.generate_batch(11, 0, 8, n_selected=0 if single else 2); sz = len(packed) // 8
            objs = [serde.unpack_encrypted_choice(packed[i * sz:(i + 1) * sz], n, single) for i in range(8)]
            cases, verdict = choice_cases(objs, single), _choice_object_verdict
        fuzzed = []
        for k in range(80):
            o = copy.deepcopy(objs[k % 8])
            for _ in range(rnd.randrange(1, 4)):
                lists = []
                def walk(x):
                    if isinstance(x, dict):
                        for v in x.values(): walk(v)
                    elif isinstance(x, list):
                        lists.append(x)
                        for v in x: walk(v)
                walk(o)
                target = rnd.choice(lists); action = rnd.randrange(4)
                if action == 0 and target: target.pop(rnd.randrange(len(target)))
                elif action == 1 and target: target.append(copy.deepcopy(rnd.choice(target)))
                elif action == 2 and target and isinstance(target[0], str):
                    i = rnd.randrange(len(target)); target[i] = rnd.choice([flip(target[i]), BAD_SCALAR])
                elif target and isinstance(target[0], dict) and "random_element" in target[0]:
                    rnd.choice(target)["blinded_element"] = BAD_POINT
            fuzzed.append(o)
        batch = [c[1] for c in cases] + fuzzed + [{{"choices": "junk"}}]
        want = [verdict(op, o) for o in batch]
        CHECK = C.CFUNCTYPE(C.c_int, C.c_size_t, C.c_char_p, C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte))
        VERIFY = C.CFUNCTYPE(C.c_int, C.c_size_t, C.POINTER(C.c_ubyte), C.c_size_t, C.POINTER(C.c_uint32))
        def check(cnt, kinds, data, ok):
            raw = bytes(data[: 32 * cnt])
            for i in range(cnt):
                item = raw[32 * i: 32 * i + 32]
                ok[i] = (oracle.point_roundtrip(item) is not None) if kinds[i: i + 1] == b"P" else int.from_bytes(item, "little") < Lq
            return 0
        def verify(cnt, packed_ptr, stride, status):
            st = op.verify_batch(bytes(packed_ptr[: cnt * stride]))
            for i in range(cnt): status[i] = st[i]
            return 0
        text = json.dumps(batch).encode()
        status = (C.c_uint32 * (len(batch) + 1))()
        if kind == "qv":
            got_n = L.pc_resolve_qv(n, C.c_ulonglong(credits), text, C.c_size_t(len(text)), CHECK(check), VERIFY(verify), status, C.c_size_t(len(batch)))
        else:
            got_n = L.pc_resolve_choice(n, int(single), text, C.c_size_t(len(text)), CHECK(check), VERIFY(verify), status, C.c_size_t(len(batch)))
        assert got_n == len(batch), got_n
        got = list(status[: got_n])
        bad = [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w]
        assert not bad, bad[:5]
        assert [c[2] for c in cases] == got[: len(cases)]
        print("kinds", len({{w & 0xFF for w in want}}))
    """)
    assert int(out.split()[-1]) >= 4
