#!/usr/bin/env python3
"""gen_golden.py - makes tests/golden/ from the REAL reference.  TEST INFRASTRUCTURE.

Run in the build container only (needs /root/reference and oracle/_ref/libredref.so, built by
``make -C oracle ref``).  Everything written is DATA: serialized DFA blobs the reference's own
compiler produced, input byte strings, and the outputs the reference's matcher returned for
them - plus the expected values the reference's own gtest files assert, transcribed as data.
No reference source text is stored.

Outputs (all under tests/golden/):
  dfas/<name>.reda[.xz]       blobs of the BASELINE config DFAs (reference compiler output)
  kat_matcher.json            known answers asserted by test/matcher.cpp, test/executable.cpp,
                              test/serializer.cpp, test/fnv.cpp (+ the blobs, per format)
  omnibus.json(+.npz)         the 160-row {regex,text,shouldMatch} table of
                              test/omnibus.cpp:244-407, with per-format blobs
  vectors_<name>.npz          reference outputs for every verb x style x doLeader on a mixed
                              set of crafted + random inputs, per config DFA
"""
from __future__ import annotations

import base64
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import oracle as O  # noqa: E402
from one_amd import workloads as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REFTEST = "/root/reference/quol/red/test"
FORMATS = [("auto", O.FMT_AUTO), ("1", O.FMT_1), ("2", O.FMT_2), ("4", O.FMT_4)]
STY = ["instant", "first", "tangent", "last", "full"]


def b64(b: bytes) -> str:
    return base64.b64encode(b).decode()


def compile_all_formats(patterns):
    out = {}
    for name, fmt in FORMATS:
        try:
            out[name] = O.ref_compile(patterns, fmt)
        except O.RefError as e:
            out[name] = ("limit" if e.code == -3 else "parse" if e.code == -4 else "err", str(e))
    return out


# ------------------------------------------------------------------------------------------
# Known answers asserted by the reference's own tests (values transcribed from the EXPECT_*
# lines; the generator re-runs each through the reference and refuses to write on mismatch).
# call = (verb, style, doLeader, input, expected) ; expected = result | (result,start,end)
# with None = "not asserted by the reference test".  Run-time-style overloads use
# doLeader=true (lib/Matcher.cpp:53-62).
# ------------------------------------------------------------------------------------------
IC, LS = O.F_IGNORE_CASE, O.F_LOOSE_START
RAW, AUTO = O.LANG_RAW, O.LANG_AUTO
NUM3 = [("[0-9]+", 1, 0), ("[0-9]+a", 2, 0), ("[0-9]+abcd", 3, 0)]


def _verify(style, rows):
    calls = []
    for text, exp in rows:
        calls += [("check", style, True, text, exp[0]), ("match", style, True, text, exp),
                  ("search", style, True, text, exp)]
    return calls


def _searchrows(style, rows):
    calls = []
    for text, exp in rows:
        calls += [("scan", style, True, text, exp[0]), ("search", style, True, text, exp)]
    return calls


KAT = [
    dict(name="case", src="test/matcher.cpp:22-42", patterns=[("abc", 1, 0)],
         calls=[("check", "full", False, "abc", 1), ("check", "full", False, "aBc", 0),
                ("check", "full", False, "ABC", 0), ("check", "full", False, "xyz", 0)]),
    dict(name="case_ignore", src="test/matcher.cpp:22-42", patterns=[("abc", 1, IC)],
         calls=[("check", "full", False, "abc", 1), ("check", "full", False, "aBc", 1),
                ("check", "full", False, "ABC", 1), ("check", "full", False, "xyz", 0)]),
    dict(name="endmarks", src="test/matcher.cpp:45-58", patterns=[("abe", 1, 0), ("ace", 2, 0)],
         calls=[("check", "full", False, "abe", 1), ("check", "full", False, "ace", 2),
                ("check", "full", False, "abc", 0), ("check", "full", False, "age", 0)]),
    dict(name="start_degenerate", src="test/matcher.cpp:61-77", patterns=[("a*", 1, 0)],
         calls=[("search", "last", True, "123a", (1, 3, 4)),
                ("search", "last", True, "123az", (1, 4, 4))]),
    dict(name="start_intended", src="test/matcher.cpp:79-93", patterns=[(".*[a-z]+", 1, 0)],
         calls=[("search", "last", True, "123a", (1, 3, 4)),
                ("search", "last", True, "123az!", (1, 3, 5))]),
    dict(name="startEnd", src="test/matcher.cpp:97-125",
         patterns=[("[^a]*ab*c", 1, 0), ("[^d]*dummy", 2, 0)],
         calls=[("match", "last", True, "abbc", (1, 0, 4)),
                ("match", "last", True, "xabbc", (1, 1, 5)),
                ("match", "last", True, "xabbcx", (1, 1, 5)),
                ("match", "last", True, "xyzabbcxyz", (1, 3, 7))]),
    dict(name="matchTangent", src="test/matcher.cpp:129-146", patterns=[("[0-9]+", 1, 0)],
         calls=[("match", "instant", True, "0123456789abcdef", (1, None, 1)),
                ("match", "first", True, "0123456789abcdef", (1, None, 10)),
                ("match", "tangent", True, "0123456789abcdef", (1, None, 10))]),
    dict(name="matchLast", src="test/matcher.cpp:149-163",
         patterns=[("New", 1, LS), ("New York", 2, LS), ("York", 3, LS)],
         calls=[("match", "last", True, "I love New York.", (2, 7, 15))]),
    dict(name="verifyInstant", src="test/matcher.cpp:166-213", patterns=NUM3,
         calls=_verify("instant", [("1", (1, 0, 1)), ("123", (1, 0, 1)), ("123abcd", (1, 0, 1)),
                                   ("123abcde", (1, 0, 1))])),
    dict(name="verifyFirst", src="test/matcher.cpp:216-263", patterns=NUM3,
         calls=_verify("first", [("1", (1, 0, 1)), ("123", (1, 0, 3)), ("123abcd", (1, 0, 3)),
                                 ("123XYZ", (1, 0, 3))])),
    dict(name="verifyTangent", src="test/matcher.cpp:266-313", patterns=NUM3,
         calls=_verify("tangent", [("1", (1, 0, 1)), ("123", (1, 0, 3)), ("123abcd", (2, 0, 4)),
                                   ("123XYZ", (1, 0, 3))])),
    dict(name="verifyLast", src="test/matcher.cpp:316-363", patterns=NUM3,
         calls=_verify("last", [("1", (1, 0, 1)), ("123", (1, 0, 3)), ("123abcd", (3, 0, 7)),
                                ("123abcde", (3, 0, 7))])),
    dict(name="verifyFull", src="test/matcher.cpp:366-413", patterns=NUM3,
         calls=_verify("full", [("1", (1, 0, 1)), ("123", (1, 0, 3)), ("123abcd", (3, 0, 7)),
                                ("123abcde", (0, 0, 0))])),
    dict(name="searchInstant", src="test/matcher.cpp:417-460", patterns=NUM3,
         calls=_searchrows("instant", [(".,_1", (1, 3, 4)), (".,_123", (1, 3, 4)),
                                       (".,_123abcd", (1, 3, 4)), (".,_123abcde", (1, 3, 4))])),
    dict(name="searchFirst", src="test/matcher.cpp:463-506", patterns=NUM3,
         calls=_searchrows("first", [(".,_1", (1, 3, 4)), (".,_123", (1, 3, 6)),
                                     (".,_123abcd", (1, 3, 6)), (".,_123XYZ", (1, 3, 6))])),
    dict(name="searchTangent", src="test/matcher.cpp:509-552", patterns=NUM3,
         calls=_searchrows("tangent", [(".,_1", (1, 3, 4)), (".,_123", (1, 3, 6)),
                                       (".,_123abcd", (2, 3, 7)), (".,_123XYZ", (1, 3, 6))])),
    dict(name="searchLast", src="test/matcher.cpp:555-598", patterns=NUM3,
         calls=_searchrows("last", [(".,_1", (1, 3, 4)), (".,_123", (1, 3, 6)),
                                    (".,_123abcd", (3, 3, 10)), (".,_123abcde", (3, 3, 10))])),
    dict(name="searchFull", src="test/matcher.cpp:601-644", patterns=NUM3,
         calls=_searchrows("full", [(".,_1", (1, 3, 4)), (".,_123", (1, 3, 6)),
                                    (".,_123abcd", (3, 3, 10)), (".,_123abcde", (0, 0, 0))])),
    dict(name="check_overloads", src="test/matcher.cpp:753-772; test/executable.cpp:69-81",
         patterns=[("ab*c", 1, 0, AUTO), ("ca*b", 2, 0, AUTO)],
         calls=[("check", "full", True, "bca", 0), ("check", "full", True, "bac", 1),
                ("check", "full", True, "cab", 2),
                ("match", "full", True, "bca", (0, None, 0)),
                ("match", "full", True, "bac", (1, None, 3)),
                ("match", "full", True, "cab", (2, None, 3))]),
    dict(name="exec_memory", src="test/executable.cpp:26-64; test/serializer.cpp:29-53",
         patterns=[("ab*c", 1, 0, AUTO)],
         calls=[("match", "full", True, "abbc", (1, None, None))]),
    # SURVEY.md section 8(a) a-M / a-N quirks, verified there against the reference
    dict(name="quirk_start_after_end", src="SURVEY.md a-M quirk 1", patterns=[(".*error", 1, 0)],
         calls=[("match", "last", False, "an error: foo e", (1, 14, 8)),
                ("match", "tangent", False, "an error: foo e", (1, 3, 8))]),
    dict(name="quirk_accepting_initial", src="SURVEY.md a-M quirk 2", patterns=[("a*", 1, 0)],
         calls=[("match", "last", False, "", (1, 0, 0)), ("match", "last", False, "b", (0, 0, 0)),
                ("match", "instant", False, "b", (0, 0, 0))]),
    dict(name="quirk_accepting_deadend", src="SURVEY.md a-M quirk 3", patterns=[("abc.*", 1, 0)],
         calls=[("match", "last", False, "abcdefgh", (1, 0, 8)),
                ("match", "full", False, "abcdefgh", (1, 0, 8))]),
    dict(name="quirk_scan_leader_skip", src="SURVEY.md a-N quirk", patterns=[("aab", 1, 0)],
         calls=[("scan", "instant", True, "aaab", 0), ("scan", "instant", False, "aaab", 1),
                ("search", "instant", True, "aaab", (1, 1, 4))]),
]

FNV64_KAT = dict(src="test/fnv.cpp:11-21", text=b64(b"chongo was here!\n"),
                 answers=[[0, "cbf29ce484222325"], [1, "af63de4c8601eff2"],
                          [2, "08a25607b54a22ae"], [17, "46810940eff5f915"],
                          [6, "e150688c8217b8fd"]])


def run_call(ref, verb, style, lead, text: bytes):
    if verb in ("check", "scan"):
        return getattr(ref, verb)(text, style, lead)
    return getattr(ref, verb)(text, style, lead)


def agrees(got, exp) -> bool:
    if isinstance(exp, int):
        return got == exp
    return all(e is None or g == e for g, e in zip(got, exp))


def gen_kat():
    cases = []
    for k in KAT:
        blobs = compile_all_formats(k["patterns"])
        case = dict(name=k["name"], src=k["src"], blobs={}, calls=[])
        for fname, blob in blobs.items():
            if isinstance(blob, tuple):
                case["blobs"][fname] = {"error": blob[0]}
                continue
            case["blobs"][fname] = {"reda": b64(blob)}
            ref, cpu = O.Reference(blob), O.CpuOracle(blob)
            for verb, style, lead, text, exp in k["calls"]:
                tb = text.encode("latin-1")
                got = run_call(ref, verb, style, lead, tb)
                assert agrees(got, exp), (k["name"], fname, verb, style, text, got, exp)
                got2 = run_call(cpu, verb, style, lead, tb)
                assert got == got2, ("oracle!=ref", k["name"], fname, verb, style, text)
        for verb, style, lead, text, exp in k["calls"]:
            case["calls"].append(dict(verb=verb, style=style, lead=lead,
                                      text=b64(text.encode("latin-1")),
                                      expect=exp if isinstance(exp, int) else list(exp)))
        cases.append(case)
    zero = dict(src="test/serializer.cpp:20-24", size=1024, expect_bad=True)
    assert O.ref_check_header(b"\0" * 1024) is not None
    with open(os.path.join(GOLD, "kat_matcher.json"), "w") as f:
        json.dump(dict(cases=cases, zero_header=zero, fnv64=FNV64_KAT), f, indent=0)
    print("kat_matcher.json:", len(cases), "cases")


# ------------------------------------------------------------------------------------------
_REC = re.compile(r'Rec\{')


def c_unescape(s: str) -> bytes:
    out = bytearray()
    i = 0
    simple = {"n": 10, "t": 9, "r": 13, "\\": 92, '"': 34, "'": 39, "0": 0, "a": 7, "b": 8,
              "f": 12, "v": 11, "?": 63}
    while i < len(s):
        c = s[i]
        if c != "\\":
            out += c.encode("latin-1")
            i += 1
            continue
        i += 1
        c = s[i]
        if c == "x":
            j = i + 1
            while j < len(s) and s[j] in "0123456789abcdefABCDEF":
                j += 1
            out.append(int(s[i + 1:j], 16) & 0xFF)
            i = j
        elif c in "01234567":
            j = i
            while j < len(s) and j < i + 3 and s[j] in "01234567":
                j += 1
            out.append(int(s[i:j], 8) & 0xFF)
            i = j
        else:
            out.append(simple[c])
            i += 1
    return bytes(out)


def parse_omnibus_rows():
    """Reads the {regex,text,shouldMatch} DATA rows of test/omnibus.cpp (testRecs[])."""
    src = open(os.path.join(REFTEST, "omnibus.cpp"), encoding="latin-1").read()
    beg = src.index("Rec testRecs[] = {")
    end = src.index("};", beg)
    body = src[beg:end]
    rows = []
    tok = re.compile(r'"((?:[^"\\]|\\.)*)"|\b(nullptr|true|false)\b|(Rec\{)|(\})|(,)')
    cur = None
    for m in tok.finditer(body[body.index("{") + 1:]):
        if m.group(3):
            cur = []
            sep = False
        elif m.group(4):
            if cur is not None:
                rows.append(cur)
                cur = None
        elif cur is not None:
            if m.group(5):
                sep = True
            elif m.group(1) is not None:
                # adjacent string literals (no comma between) concatenate
                if cur and isinstance(cur[-1], list) and not sep:
                    cur[-1].append(m.group(1))
                else:
                    cur.append([m.group(1)])
                sep = False
            else:
                cur.append(m.group(2))
                sep = False
    out = []
    for r in rows:
        regex = c_unescape("".join(r[0]))
        text = None if r[1] == "nullptr" else c_unescape("".join(r[1]))
        out.append((regex, text, r[2] == "true"))
    return out


def gen_omnibus():
    rows = parse_omnibus_rows()
    assert len(rows) == 160, len(rows)  # 160 Rec{} rows in testRecs[]
    blobs = {}
    meta = []
    for i, (regex, text, should) in enumerate(rows):
        rec = dict(regex=b64(regex), text=None if text is None else b64(text), match=should,
                   fmt={})
        for fname, fmt in FORMATS:
            try:
                blob = O.ref_compile([(regex, 1, 0, O.LANG_AUTO)], fmt)
            except O.RefError as e:
                kind = {-3: "limit", -4: "parse"}.get(e.code, "err")
                rec["fmt"][fname] = kind
                if kind == "parse":
                    assert text is None, (i, regex)
                continue
            assert text is not None, (i, regex)
            ref, cpu = O.Reference(blob), O.CpuOracle(blob)
            # the test calls check(rex, const char*, styFull): C-string input, doLeader=true
            t = text.split(b"\0")[0]
            r1 = ref.check(t, "full", True)
            r2 = ref.match(t, "full", True)
            assert r1 == r2[0] and (r1 == 1) == should, (i, regex, t, r1, r2)
            assert cpu.check(t, "full", True) == r1 and cpu.match(t, "full", True) == r2
            key = "r%03d_%s" % (i, fname)
            blobs[key] = np.frombuffer(blob, dtype=np.uint8)
            rec["fmt"][fname] = key
        meta.append(rec)
    with open(os.path.join(GOLD, "omnibus.json"), "w") as f:
        json.dump(dict(src="test/omnibus.cpp:244-407,506-555", rows=meta), f, indent=0)
    np.savez_compressed(os.path.join(GOLD, "omnibus_blobs.npz"), **blobs)
    print("omnibus:", len(meta), "rows,", len(blobs), "blobs")


# ------------------------------------------------------------------------------------------
def config_dfas():
    d = {}
    d["err"] = O.ref_compile([("error", 1, 0)])
    d["uri"] = O.ref_compile([(W.URI_REGEX, 1, O.F_LOOSE_START)])
    d["log100"] = O.ref_compile(W.log100_patterns())
    d["syn256"] = O.ref_syn_dfa(256, 42)
    # BASELINE configs[4]'s "~4 K-state DFA": dense random, through the reference's minimizer +
    # serializer like SYN-256 (4,097 states x 256 classes, fmtDirect4, 4.2 MB: stored xz-compressed)
    d["syn4k"] = O.ref_syn_dfa(4096, 5)
    d["num3"] = O.ref_compile(NUM3)
    d["newyork"] = O.ref_compile([("New", 1, LS), ("New York", 2, LS), ("York", 3, LS)])
    d["aab"] = O.ref_compile([("aab", 1, 0)])
    d["dotstar_err"] = O.ref_compile([(".*error", 1, 0)])
    d["uri_v6"] = O.ref_compile([(W.URI_V6_REGEX, 1, O.F_LOOSE_START | O.F_IGNORE_CASE)])
    d["uri_user"] = O.ref_compile([(W.URI_USER_REGEX, 1, O.F_LOOSE_START)])
    return d


def inputs_for(name: str, rng: np.random.Generator):
    """A mixed bag of crafted and random inputs (list of bytes)."""
    ins = [b"", b"a", b"e", b"error", b"erro", b"errorx", b"xerror", b"an error: foo e",
           b"aaab", b"aab", b"aabaab", b"123abcd", b".,_123abcde", b"I love New York.",
           b"New", b"New York", b"York", b"0", b"0123456789abcdef", b"\0\0", b"\xff" * 5]
    heads = W.log100_heads()
    ins += heads[:10] + [h[:-3] for h in heads[10:20]] + [b"x" + h for h in heads[20:24]]
    ins += [W.URI_PLANT, b"see " + W.URI_PLANT, b"http://1.2.3.4", b"ftp://a.bc/",
            b"http://a.b", b"xxhttps://a.io:80/?#", b"HTTP://A.COM"]
    if name == "uri_user":
        ins += [W.URI_USER_PLANT, b"at " + W.URI_USER_PLANT + b"x", b"git://u@h.io", b"http://@a.bc"]
    if name == "uri_v6":  # added with that DFA; the older sets keep their recorded inputs
        ins += [W.URI_V6_PLANT, b"go " + W.URI_V6_PLANT + b"tail", b"ssh://[::1]:22/",
                b"Gopher://U@[1:2:3:4:5:6:7:8]", b"s3://bucket.name.io/k?v#f", b"nfs://[1::]"]
    for _ in range(700):
        n = int(rng.integers(0, 200))
        ins.append(W.ALPHABET47[rng.integers(0, 47, n)].tobytes())
    for _ in range(300):
        n = int(rng.integers(0, 300))
        ins.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    for _ in range(300):  # planted positives at random places in noise
        pre = W.ALPHABET47[rng.integers(0, 47, int(rng.integers(0, 40)))].tobytes()
        post = W.ALPHABET47[rng.integers(0, 47, int(rng.integers(0, 40)))].tobytes()
        mid = [b"error", W.URI_PLANT, heads[int(rng.integers(0, 100))], b"New York",
               b"%d" % rng.integers(0, 10**6), b"123abcd", b"aab"][int(rng.integers(0, 7))]
        ins.append(pre + mid + post)
    return ins


def gen_vectors(only=None):
    dfas = config_dfas()
    for name, blob in dfas.items():
        if only and name not in only:
            continue
        if len(blob) > (1 << 20):
            import lzma
            with open(os.path.join(GOLD, "dfas", name + ".reda.xz"), "wb") as f:
                f.write(lzma.compress(blob, preset=6))
        else:
            with open(os.path.join(GOLD, "dfas", name + ".reda"), "wb") as f:
                f.write(blob)
        info = O.CpuOracle(blob).info
        print("dfa %-12s %8d B  %s" % (name, len(blob), info))
        rng = np.random.default_rng(0xC0FFEE)
        ins = inputs_for(name, rng)
        data = np.frombuffer(b"".join(ins), dtype=np.uint8)
        offsets = np.zeros(len(ins) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([len(x) for x in ins])
        ref, cpu = O.Reference(blob), O.CpuOracle(blob)
        arrays = dict(data=data, offsets=offsets)
        for verb in ("check", "match", "scan", "search"):
            for si, sty in enumerate(STY, start=1):
                for lead in (0, 1):
                    r, s, e = ref.batch(verb, sty, lead, data, offsets=offsets)
                    r2, s2, e2 = cpu.batch(verb, sty, lead, data, offsets=offsets)
                    assert (r == r2).all() and (s == s2).all() and (e == e2).all(), \
                        ("oracle!=ref", name, verb, sty, lead)
                    key = "%s_%d_%d" % (verb, si, lead)
                    arrays[key + "_res"] = r
                    if verb in ("match", "search"):
                        arrays[key + "_start"] = s
                        arrays[key + "_end"] = e
        np.savez_compressed(os.path.join(GOLD, "vectors_%s.npz" % name), **arrays)


def gen_collect():
    """Red::collect: the known answer of test/red.cpp:190-221 + reference outputs on the mixed
    input set for two multi-pattern DFAs."""
    pats = [("new york", 1, 0), ("new", 2, 0), ("york", 3, 0), ("[0-9]+", 4, 0)]
    blob = O.ref_compile(pats)
    text = b"in new york12345, a new 6789 york city"
    expect = [(1, 3, 11), (4, 11, 16), (2, 20, 23), (4, 24, 28), (3, 29, 33)]
    got, k = O.ref_collect(blob, text)
    assert got == expect and k == 5, got
    assert O.ref_collect(blob, b"new york")[1] == 1
    out = dict(kat=dict(src="test/red.cpp:190-221", reda=b64(blob), text=b64(text),
                        expect=[list(x) for x in expect]), sets={})
    rng = np.random.default_rng(0xC011EC7)
    ins = inputs_for("collect", rng)[:1200]
    ins += [text, b"new york", b"", b"newnewnew", b"12 34 56 new york york new 7"]
    cap = 16
    arrays = {}
    for name, b in (("newyork4", blob), ("num3", O.ref_compile(NUM3)),
                    ("newyork_loose", O.ref_compile([("New", 1, LS), ("New York", 2, LS),
                                                     ("York", 3, LS)]))):
        cpu = O.CpuOracle(b)
        counts = np.zeros(len(ins), dtype=np.uint64)
        res = np.zeros((len(ins), cap), dtype=np.int32)
        st = np.zeros((len(ins), cap), dtype=np.uint64)
        en = np.zeros((len(ins), cap), dtype=np.uint64)
        for i, t in enumerate(ins):
            got, k = O.ref_collect(b, t, cap)
            assert (got, k) == cpu.collect(t, cap), (name, t)
            counts[i] = k
            for j, (r, s_, e_) in enumerate(got):
                res[i, j], st[i, j], en[i, j] = r, s_, e_
        arrays[name + "_blob"] = np.frombuffer(b, dtype=np.uint8)
        arrays[name + "_counts"], arrays[name + "_res"] = counts, res
        arrays[name + "_start"], arrays[name + "_end"] = st, en
    data = np.frombuffer(b"".join(ins), dtype=np.uint8)
    offsets = np.zeros(len(ins) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(x) for x in ins])
    arrays["data"], arrays["offsets"] = data, offsets
    arrays["cap"] = np.array([cap])
    np.savez_compressed(os.path.join(GOLD, "collect_vectors.npz"), **arrays)
    with open(os.path.join(GOLD, "collect_kat.json"), "w") as f:
        json.dump(out["kat"], f)
    print("collect: kat + %d inputs x 3 dfas" % len(ins))


SET5 = [("0", 1, 0), ("0123", 2, 0), ("[0-2]+", 3, 0), ("[3-9]+", 4, 0), ("012345", 5, 0)]


def gen_matchall_stateful():
    """matchAll (include/Matcher.h:711-766) and StatefulMatcher (include/Matcher.h:770-792):
    the known answers of test/matcher.cpp:695-745,800-818 and test/red.cpp:176-187, run through
    the real reference in all four formats, + reference outputs on the mixed input set."""
    kats = []
    exp5 = [(1, 0, 1), (3, 0, 3), (2, 0, 4), (5, 0, 6)]
    expl = [(1, None, 3), (2, None, 6), (2, None, 9), (1, None, 12), (2, None, 14), (1, None, 15)]
    for fmt, blob in compile_all_formats(SET5).items():
        if isinstance(blob, tuple):
            continue
        got, k = O.Reference(blob).match_all(b"0123456789")
        assert k == 4 and got == exp5, got
        kats.append(dict(src="test/matcher.cpp:695-723 (and test/red.cpp:176-187)", fmt=fmt,
                         reda=b64(blob), text=b64(b"0123456789"), count=4,
                         expect=[list(x) for x in got]))
    for fmt, blob in compile_all_formats([("a+", 1, LS), ("b+", 2, LS)]).items():
        if isinstance(blob, tuple):
            continue
        got, k = O.Reference(blob).match_all(b".aa..b.bb..abba.")
        assert k == 6 and all(g[0] == e[0] and g[2] == e[2] for g, e in zip(got, expl)), got
        kats.append(dict(src="test/matcher.cpp:725-745 (start_ not asserted there; the "
                             "reference's own output is recorded)", fmt=fmt, reda=b64(blob),
                         text=b64(b".aa..b.bb..abba."), count=6, expect=[list(x) for x in got]))
    skat = []
    for fmt, blob in compile_all_formats([("ale+", 1, 0), ("ale*x", 2, 0)]).items():
        if isinstance(blob, tuple):
            continue
        ini, fin, per = O.Reference(blob).stateful(b"aleex")
        assert ini == 0 and fin == 2 and per.tolist() == [0, 0, 1, 1, 2]
        skat.append(dict(src="test/matcher.cpp:800-818", fmt=fmt, reda=b64(blob),
                         text=b64(b"aleex"), initial=0, per_byte=per.tolist(), final=2))
    # blobs for the C++ mirror test (tests/cpp/matcher_cabi_test.cpp)
    for name, pats in (("set5", SET5), ("loose2", [("a+", 1, LS), ("b+", 2, LS)]),
                       ("ale", [("ale+", 1, 0), ("ale*x", 2, 0)]),
                       ("num3defg", [("[0-9]+", 1, 0), ("[0-9]+d", 2, 0), ("[0-9]+defg", 3, 0)]),
                       ("newyork4", [("new york", 1, 0), ("new", 2, 0), ("york", 3, 0),
                                     ("[0-9]+", 4, 0)])):
        with open(os.path.join(GOLD, "dfas", name + ".reda"), "wb") as f:
            f.write(O.ref_compile(pats))
    with open(os.path.join(GOLD, "matchall_kat.json"), "w") as f:
        json.dump(kats, f)
    with open(os.path.join(GOLD, "stateful_kat.json"), "w") as f:
        json.dump(skat, f)

    rng = np.random.default_rng(0xA11)
    ins = inputs_for("matchall", rng)[:1100]
    ins += [b"0123456789", b".aa..b.bb..abba.", b"aleex", b"", b"0", b"00123", b"9876543210" * 30]
    data = np.frombuffer(b"".join(ins), dtype=np.uint8)
    offsets = np.zeros(len(ins) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(x) for x in ins])
    cap = 8  # small on purpose: some inputs overflow it, counts keep counting
    dfas = config_dfas()
    sets = dict(set5=O.ref_compile(SET5), loose2=O.ref_compile([("a+", 1, LS), ("b+", 2, LS)]),
                num3=dfas["num3"], newyork=dfas["newyork"], err=dfas["err"], uri=dfas["uri"],
                log100=dfas["log100"], syn256=dfas["syn256"])
    arrays = dict(data=data, offsets=offsets, cap=np.array([cap]))
    for name, b in sets.items():
        ref, cpu = O.Reference(b), O.CpuOracle(b)
        counts = np.zeros(len(ins), dtype=np.uint64)
        res = np.zeros((len(ins), cap), dtype=np.int32)
        st = np.zeros((len(ins), cap), dtype=np.uint64)
        en = np.zeros((len(ins), cap), dtype=np.uint64)
        for i, t in enumerate(ins):
            got, k = ref.match_all(t, cap)
            assert (got, k) == cpu.match_all(t, True, cap), (name, t)
            counts[i] = k
            for j, (r, s_, e_) in enumerate(got):
                res[i, j], st[i, j], en[i, j] = r, s_, e_
        c2, r2, s2, e2 = cpu.match_all_batch(data, cap, do_leader=True, offsets=offsets)
        assert (c2 == counts).all() and (r2 == res).all() and (s2 == st).all() and (e2 == en).all()
        arrays[name + "_blob"] = np.frombuffer(b, dtype=np.uint8)
        arrays[name + "_lead1_counts"], arrays[name + "_lead1_res"] = counts, res
        arrays[name + "_lead1_start"], arrays[name + "_lead1_end"] = st, en
        # doLeader = false has no public entry in the reference (lib/Matcher.cpp:101 fixes it
        # to true); where the DFA has no leader the two are the same walk, which is asserted,
        # and for the DFAs with one the restatement's output is recorded.
        c0, r0, s0, e0 = cpu.match_all_batch(data, cap, do_leader=False, offsets=offsets)
        if cpu.info["leaderLen"] == 0:
            assert (c0 == counts).all() and (r0 == res).all() and (e0 == en).all()
        arrays[name + "_lead0_counts"], arrays[name + "_lead0_res"] = c0, r0
        arrays[name + "_lead0_start"], arrays[name + "_lead0_end"] = s0, e0
        # StatefulMatcher: every advance() of a fresh matcher over every input
        per = np.zeros(len(data), dtype=np.int32)
        fin = np.zeros(len(ins), dtype=np.int32)
        ini = np.zeros(1, dtype=np.int32)
        for i, t in enumerate(ins):
            a, z, pb = ref.stateful(t)
            z2, pb2 = cpu.stateful(t)
            assert z == z2 and (pb == pb2).all(), (name, t)
            ini[0] = a
            fin[i] = z
            per[int(offsets[i]):int(offsets[i + 1])] = pb
        arrays[name + "_sm_initial"], arrays[name + "_sm_final"] = ini, fin
        arrays[name + "_sm_per_byte"] = per
    np.savez_compressed(os.path.join(GOLD, "matchall_stateful_vectors.npz"), **arrays)
    print("matchAll/stateful: %d + %d kats, %d inputs x %d dfas" %
          (len(kats), len(skat), len(ins), len(sets)))


def gen_replace():
    """replace (include/Matcher.h:643-706): the known answers of test/matcher.cpp:648-691 through
    the real reference in every format, + reference outputs on the mixed input set for every
    style x doLeader, two replacement strings and two max counts."""
    kats = []
    rows1 = [("fooac", "bar", 9999, "last", 1, "foobar"), ("fooacz", "bar", 9999, "last", 1, "foobarz"),
             ("xacyabbcz", ",", 9999, "tangent", 2, "x,y,z")]
    rows2 = [("#123defg!", "xyz", 1, "instant", 1, "#xyz23defg!"),
             ("#123defg!", "xyz", 9999, "instant", 3, "#xyzxyzxyzdefg!"),
             ("#123defg!", "xyz", 9999, "first", 1, "#xyzdefg!"),
             ("#123defg!", "xyz", 9999, "tangent", 1, "#xyzefg!"),
             ("#123defg!", "xyz", 9999, "last", 1, "#xyz!"),
             ("#123defg!", "xyz", 9999, "full", 0, "#123defg!"),
             ("#123defg", "xyz", 9999, "full", 1, "#xyz")]
    for src, pats, rows in (("test/matcher.cpp:648-664", [("ab*c", 1, 0)], rows1),
                            ("test/matcher.cpp:667-691",
                             [("[0-9]+", 1, 0), ("[0-9]+d", 2, 0), ("[0-9]+defg", 3, 0)], rows2)):
        for fmt, blob in compile_all_formats(pats).items():
            if isinstance(blob, tuple):
                continue
            ref = O.Reference(blob)
            for text, repl, mx, sty, cnt, exp in rows:
                got = ref.replace(text.encode(), repl.encode(), sty, True, mx)
                assert got == (cnt, exp.encode()), (src, fmt, text, sty, got)
                kats.append(dict(src=src, fmt=fmt, reda=b64(blob), text=text, repl=repl, max=mx,
                                 style=sty, count=cnt, expect=exp))
    with open(os.path.join(GOLD, "replace_kat.json"), "w") as f:
        json.dump(kats, f)
    rng = np.random.default_rng(0x4E91)
    ins = inputs_for("replace", rng)[:420]
    ins += [b"fooac", b"xacyabbcz", b"#123defg!", b"#123defg", b"", b"0", b"00", b"a1b22c333"]
    data = np.frombuffer(b"".join(ins), dtype=np.uint8)
    offsets = np.zeros(len(ins) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(x) for x in ins])
    dfas = config_dfas()
    sets = dict(num3=dfas["num3"], newyork=dfas["newyork"], err=dfas["err"], uri=dfas["uri"],
                abc=O.ref_compile([("ab*c", 1, 0)]))
    arrays = dict(data=data, offsets=offsets)
    cases = [(b"<>", 1 << 62), (b"", 2), (b"a much longer replacement", 1)]
    for name, b in sets.items():
        ref, cpu = O.Reference(b), O.CpuOracle(b)
        arrays[name + "_blob"] = np.frombuffer(b, dtype=np.uint8)
        for ci, (repl, mx) in enumerate(cases):
            for si, sty in enumerate(STY, start=1):
                for lead in (0, 1):
                    counts = np.zeros(len(ins), dtype=np.uint64)
                    outs = []
                    for i, t in enumerate(ins):
                        k, o = ref.replace(t, repl, sty, lead, mx)
                        assert (k, o) == cpu.replace(t, repl, sty, lead, mx), (name, sty, lead, t)
                        counts[i] = k
                        outs.append(o)
                    ooff = np.zeros(len(ins) + 1, dtype=np.uint64)
                    ooff[1:] = np.cumsum([len(o) for o in outs])
                    key = "%s_c%d_%d_%d_" % (name, ci, si, lead)
                    arrays[key + "counts"] = counts.astype(np.uint16)
                    arrays[key + "ooff"] = ooff.astype(np.uint32)
                    # the rewritten lines themselves as one FNV-1a-64 per line (the full bytes
                    # of 150 result sets would be 13 MB); num3's first case is kept whole
                    arrays[key + "fnv"] = np.array([O.fnv1a64(o) for o in outs], dtype=np.uint64)
                    if name == "num3" and ci == 0:
                        arrays[key + "out"] = np.frombuffer(b"".join(outs), dtype=np.uint8)
    arrays["case_repl"] = np.array([c[0].decode() for c in cases])
    arrays["case_max"] = np.array([c[1] for c in cases], dtype=np.uint64)
    np.savez_compressed(os.path.join(GOLD, "replace_vectors.npz"), **arrays)
    print("replace: %d kats, %d inputs x %d dfas x %d cases" % (len(kats), len(ins), len(sets), len(cases)))


def main():
    os.makedirs(os.path.join(GOLD, "dfas"), exist_ok=True)
    only = sys.argv[1:]
    steps = dict(kat=gen_kat, omnibus=gen_omnibus, vectors=gen_vectors, collect=gen_collect,
                 matchall=gen_matchall_stateful, replace=gen_replace)
    for name, fn in steps.items():
        if not only or name in only:
            fn()
    for arg in only:  # "vectors:NAME[,NAME]" regenerates the vectors of those DFAs only
        if arg.startswith("vectors:"):
            gen_vectors(set(arg.split(":", 1)[1].split(",")))


if __name__ == "__main__":
    main()
