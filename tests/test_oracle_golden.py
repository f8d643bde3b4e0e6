"""The CPU restatement (oracle/red_oracle.c) against every golden vector: the known answers the
reference's own tests assert, and the outputs of the reference itself on the config DFAs."""
import numpy as np
import pytest

import oracle as O
from golden_util import (CONFIG_DFAS, STYLES, expect_of, kat_items, load_dfa, load_kat,
                         load_omnibus, load_vectors, unb64)


def _run(cpu, verb, style, lead, text):
    return getattr(cpu, verb)(text, style, lead)


def test_kat_matcher_cpp():
    n = 0
    for name, fmt, blob, calls in kat_items():
        cpu = O.CpuOracle(blob)
        for c in calls:
            got = _run(cpu, c["verb"], c["style"], c["lead"], unb64(c["text"]))
            exp = c["expect"]
            if isinstance(exp, int):
                assert got == exp, (name, fmt, c)
            else:
                for g, e in zip(got, exp):
                    assert e is None or g == e, (name, fmt, c, got)
            n += 1
    assert n > 400


def test_zero_header_rejected():
    # test/serializer.cpp:20-24
    z = load_kat()["zero_header"]
    assert O.check_header(b"\0" * z["size"]) == "Serialized DFA: bad magic number"


def test_header_messages():
    blob = bytearray(load_dfa("err"))
    assert O.check_header(bytes(blob)) is None
    assert O.check_header(bytes(blob[:100])) == "Serialized DFA: header too short"
    b = bytearray(blob); b[5] = 9
    assert O.check_header(bytes(b)) == "Serialized DFA: unrecognized version"
    b = bytearray(blob); b[-1] ^= 1
    assert O.check_header(bytes(b)) == "serialized DFA: checksum mismatch"
    b = bytearray(blob); b[8:12] = b[8:12][::-1]
    assert O.check_header(bytes(b)) == "serialized DFA: foreign endian-ness"
    with pytest.raises(ValueError):
        O.CpuOracle(bytes(b))


def test_fnv64_kat():
    # test/fnv.cpp:11-21
    k = load_kat()["fnv64"]
    text = unb64(k["text"]) + b"\0"
    for n, hexval in k["answers"]:
        assert O.fnv1a64(text[:n]) == int(hexval, 16)


def test_omnibus_table():
    rows, blobs = load_omnibus()
    assert len(rows) == 160
    checked = 0
    for r in rows:
        for fmt, key in r["fmt"].items():
            if key in ("limit", "parse", "err"):
                continue
            cpu = O.CpuOracle(blobs[key].tobytes())
            text = unb64(r["text"]).split(b"\0")[0]
            res = cpu.check(text, "full", True)
            oc = cpu.match(text, "full", True)
            assert res == oc[0] and (res == 1) == r["match"], (r, fmt)
            checked += 1
    assert checked > 500


@pytest.mark.parametrize("name", CONFIG_DFAS)
def test_reference_vectors(name):
    vec = load_vectors(name)
    cpu = O.CpuOracle(load_dfa(name))
    data, offsets = vec["data"], vec["offsets"]
    for verb in ("check", "match", "scan", "search"):
        for si, sty in enumerate(STYLES, start=1):
            for lead in (0, 1):
                er, es, ee = expect_of(vec, verb, si, lead)
                r, s, e = cpu.batch(verb, sty, lead, data, offsets=offsets, threads=2)
                assert np.array_equal(r, er), (name, verb, sty, lead)
                if es is not None:
                    assert np.array_equal(s, es) and np.array_equal(e, ee), (name, verb, sty)


def test_collect_kat_and_vectors():
    """Red::collect (lib/Red.cpp:103-116): the known answer of test/red.cpp:190-221 and the
    reference's outputs on the mixed input set."""
    import json
    import os
    from golden_util import GOLD
    kat = json.load(open(os.path.join(GOLD, "collect_kat.json")))
    cpu = O.CpuOracle(unb64(kat["reda"]))
    got, k = cpu.collect(unb64(kat["text"]))
    assert got == [tuple(x) for x in kat["expect"]] and k == 5
    vec = np.load(os.path.join(GOLD, "collect_vectors.npz"))
    cap = int(vec["cap"][0])
    for name in ("newyork4", "num3", "newyork_loose"):
        cpu = O.CpuOracle(vec[name + "_blob"].tobytes())
        counts, res, st, en = cpu.collect_batch(vec["data"], cap, offsets=vec["offsets"])
        assert np.array_equal(counts, vec[name + "_counts"])
        assert np.array_equal(res, vec[name + "_res"])
        assert np.array_equal(st, vec[name + "_start"]) and np.array_equal(en, vec[name + "_end"])


def test_match_all_and_stateful_kat_and_vectors():
    """matchAll (include/Matcher.h:711-766) and StatefulMatcher (include/Matcher.h:770-792):
    the known answers of test/matcher.cpp:695-745,800-818 in every format, and the reference's
    outputs on the mixed input set (cap 8 is overflowed by some inputs on purpose)."""
    import json
    import os
    from golden_util import GOLD
    for k in json.load(open(os.path.join(GOLD, "matchall_kat.json"))):
        got, cnt = O.CpuOracle(unb64(k["reda"])).match_all(unb64(k["text"]))
        assert cnt == k["count"] and got == [tuple(x) for x in k["expect"]], k["src"]
    for k in json.load(open(os.path.join(GOLD, "stateful_kat.json"))):
        cpu = O.CpuOracle(unb64(k["reda"]))
        assert cpu.stateful(b"")[0] == k["initial"]
        fin, per = cpu.stateful(unb64(k["text"]))
        assert fin == k["final"] and per.tolist() == k["per_byte"]
    vec = np.load(os.path.join(GOLD, "matchall_stateful_vectors.npz"))
    cap = int(vec["cap"][0])
    data, offsets = vec["data"], vec["offsets"]
    for name in ("set5", "loose2", "num3", "newyork", "err", "uri", "log100", "syn256"):
        cpu = O.CpuOracle(vec[name + "_blob"].tobytes())
        for lead in (1, 0):
            c, r, s, e = cpu.match_all_batch(data, cap, do_leader=lead, offsets=offsets)
            key = "%s_lead%d_" % (name, lead)
            assert np.array_equal(c, vec[key + "counts"]) and np.array_equal(r, vec[key + "res"])
            assert np.array_equal(s, vec[key + "start"]) and np.array_equal(e, vec[key + "end"])
        # stateful: whole inputs, then the same inputs cut in two chunks with the state carried
        state = np.full(len(offsets) - 1, O.STATE_INITIAL, dtype=np.uint32)
        assert np.array_equal(cpu.advance_batch(data, state, offsets=offsets),
                              vec[name + "_sm_final"])
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        cut = (lens // 3).astype(np.uint64)
        mid = offsets[:-1] + cut
        per = vec[name + "_sm_per_byte"]
        state = np.full(len(lens), O.STATE_INITIAL, dtype=np.uint32)
        offs_a = np.stack([offsets[:-1], mid], axis=1)
        ra = np.array([cpu_adv(cpu, data, state, i, int(offs_a[i, 0]), int(offs_a[i, 1]))
                       for i in range(len(lens))], dtype=np.int32)
        exp_a = np.where(cut > 0, per[np.maximum(mid.astype(np.int64) - 1, 0)],
                         vec[name + "_sm_initial"][0])
        assert np.array_equal(ra, exp_a), name
        rb = np.array([cpu_adv(cpu, data, state, i, int(mid[i]), int(offsets[i + 1]))
                       for i in range(len(lens))], dtype=np.int32)
        assert np.array_equal(rb, vec[name + "_sm_final"]), name


def cpu_adv(cpu, data, state, i, lo, hi):
    st = state[i:i + 1]
    return int(cpu.advance_batch(data[lo:hi], st, offsets=np.array([0, hi - lo], dtype=np.uint64))[0])


def test_split_lines_rule():
    """lib/Util.cpp:109-130: lines end at a newline, the newline is not part of the line, and a
    trailing fragment without one is not a line - offsets form vs the loop form."""
    rng = np.random.default_rng(5)
    for text in (b"", b"\n", b"abc", b"abc\n", b"a\n\nbc\nrest", b"\n\n\n",
                 bytes(rng.choice(np.frombuffer(b"ab\n", dtype=np.uint8), 500))):
        offs = O.split_lines(text)
        lines = [text[int(offs[k]):int(offs[k + 1]) - 1] for k in range(len(offs) - 1)]
        assert lines == O.split_lines_loop(text)
        assert int(offs[-1]) == (text.rfind(b"\n") + 1)


def _fnv_lines(out, ooff):
    return np.array([O.fnv1a64(out[int(ooff[i]):int(ooff[i + 1])].tobytes())
                     for i in range(len(ooff) - 1)], dtype=np.uint64)


def test_replace_kat_and_vectors():
    """replace (include/Matcher.h:643-706): the known answers of test/matcher.cpp:648-691 in
    every format, and the reference's outputs (counts, lengths, FNV-1a-64 per rewritten line)
    for every style x doLeader x (replacement, max) case on the mixed input set."""
    import json
    import os
    from golden_util import GOLD
    for k in json.load(open(os.path.join(GOLD, "replace_kat.json"))):
        cpu = O.CpuOracle(unb64(k["reda"]))
        got = cpu.replace(k["text"].encode(), k["repl"].encode(), k["style"], True, k["max"])
        assert got == (k["count"], k["expect"].encode()), k
    vec = np.load(os.path.join(GOLD, "replace_vectors.npz"))
    data, offsets = vec["data"], vec["offsets"]
    ins = [data[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(len(offsets) - 1)]
    for name in ("num3", "abc", "err"):
        cpu = O.CpuOracle(vec[name + "_blob"].tobytes())
        for ci in range(3):
            repl, mx = str(vec["case_repl"][ci]).encode(), int(vec["case_max"][ci])
            for si in (1, 3, 4, 5):
                for lead in (0, 1):
                    key = "%s_c%d_%d_%d_" % (name, ci, si, lead)
                    res = [cpu.replace(t, repl, si, lead, mx) for t in ins]
                    assert [r[0] for r in res] == vec[key + "counts"].tolist(), key
                    assert [O.fnv1a64(r[1]) for r in res] == vec[key + "fnv"].tolist(), key
