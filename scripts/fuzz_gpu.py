"""Differential fuzz on the GPU: random DFAs x random line shapes x every verb / style / leader x
random placement and kernel-selection flags, against the CPU oracle.  Not part of the pytest
suite (it is open-ended); a failure prints the case and exits non-zero.
usage: fuzz_gpu.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import one_amd
import oracle as O
from oracle.reda_writer import random_dfa
from golden_util import load_dfa, CONFIG_DFAS

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
HOT_BIAS = len(sys.argv) > 3 and sys.argv[3] in ("hot", "cls")  # big DFAs, block-multiple strides
CLS_BIAS = len(sys.argv) > 3 and sys.argv[3] == "cls"           # ... of the class-table size
LISTS = len(sys.argv) > 3 and sys.argv[3] == "lists"            # only the record-list verbs
BLOCKS = len(sys.argv) > 3 and sys.argv[3] == "blocks"          # >= 4096 lines, check / match /
                                                                # match_all: the block kernels
rng = np.random.default_rng(seed)
ALPHA = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789 ./:-_=&?%@[]", dtype=np.uint8)


def make_dfa():
    if CLS_BIAS:
        n = int(rng.choice([257, 300, 700, 1300, 2000]))
        c = int(rng.choice([2, 7, 25, 40, 100, 127]))
        s = int(rng.integers(0, 1 << 30))
        dead = float(rng.choice([0.0, 0.002, 0.02]))
        return "rnd(%d,%d,%d,dead=%.3f)" % (n, c, s, dead), random_dfa(n, c, s, dead_frac=dead, accept_frac=0.2)
    if HOT_BIAS:
        k = int(rng.integers(0, 4))
        if k == 0:
            name = ["uri_v6", "uri_user", "log100"][int(rng.integers(0, 3))]
            return name, load_dfa(name)
        n = int(rng.choice([300, 800, 2500]))
        c = int(rng.choice([7, 30, 64]))
        s = int(rng.integers(0, 1 << 30))
        dead = float(rng.choice([0.0, 0.002, 0.02]))
        return "rnd(%d,%d,%d,dead=%.3f)" % (n, c, s, dead), random_dfa(n, c, s, dead_frac=dead, accept_frac=0.2)
    k = int(rng.integers(0, 10))
    if k < 4:
        name = CONFIG_DFAS[int(rng.integers(0, len(CONFIG_DFAS)))]
        return name, load_dfa(name)
    n = int(rng.choice([2, 5, 17, 100, 255, 256, 257, 300, 800, 2500]))
    c = int(rng.choice([1, 2, 7, 30, 64, 128, 256]))
    dead = float(rng.choice([0.0, 0.0, 0.01, 0.05, 0.3, 0.9]))  # 0.9: sparse rows (kind 7) when big
    acc = float(rng.choice([0.0, 0.05, 0.2, 0.9]))
    s = int(rng.integers(0, 1 << 30))
    return "rnd(%d,%d,%d,dead=%.2f,acc=%.2f)" % (n, c, s, dead, acc), \
        random_dfa(n, c, s, dead_frac=dead, accept_frac=acc)


def make_lines():
    total_cap = 600_000
    if BLOCKS:
        if rng.random() < 0.6:
            L = int(rng.choice([32, 33, 48, 63, 64, 65, 80, 100, 127, 129]))
            n = int(rng.integers(4096, 6000))
            return dict(stride=L, n=n), n * L
        n = int(rng.choice([4096, 5000, 9000]))
        lens = rng.integers(0, 140, n) if rng.random() < 0.7 else rng.geometric(1 / 30, n) - 1
        lens = np.minimum(lens, 300).astype(np.int64)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        return dict(offsets=off), int(off[-1])
    if rng.random() < 0.5:
        L = int(rng.choice([64, 128, 192, 256, 1024, 4096] if HOT_BIAS else
                           [0, 1, 15, 16, 17, 48, 63, 64, 65, 128, 192, 256, 1000, 4096]))
        n = int(rng.integers(1, max(2, min(40000, total_cap // max(L, 1)))))
        return dict(stride=L, n=n), n * L
    n = int(rng.choice([1, 2, 100, 3000, 20000]))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        lens = rng.integers(0, 200, n)
    elif kind == 1:
        lens = rng.geometric(1 / 40, n) - 1
    else:
        lens = np.where(rng.random(n) < 0.5, 0, rng.integers(0, 70, n))
    if rng.random() < 0.2:
        lens[int(rng.integers(0, n))] = int(rng.integers(3000, 9000))
    lens = np.minimum(lens, max(1, total_cap // n)).astype(np.int64)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    return dict(offsets=off), int(off[-1])


def make_data(nbytes, blob_name):
    k = int(rng.integers(0, 4))
    if k == 0:
        d = rng.integers(0, 256, nbytes, dtype=np.uint8)
    elif k == 1:
        d = ALPHA[rng.integers(0, len(ALPHA), nbytes)]
    elif k == 2:
        d = rng.integers(0, 4, nbytes, dtype=np.uint8) + 97
    else:
        d = ALPHA[rng.integers(0, len(ALPHA), nbytes)].copy()
        plants = [b"error", b"https://ab-c.example.com:8080/p?q=1#f ", b"New York", b"123abcd", b"aab",
                  b"ERROR [net] conn failed code=7", b"ssh://u@h.example.org:22/x "]
        for _ in range(max(1, nbytes // 300)):
            p = np.frombuffer(plants[int(rng.integers(0, len(plants)))], dtype=np.uint8)
            if nbytes > len(p):
                at = int(rng.integers(0, nbytes - len(p)))
                d[at:at + len(p)] = p
    return np.ascontiguousarray(d)


t0 = time.time()
stats = {}
for case in range(cases):
    name, blob = make_dfa()
    shape, nbytes = make_lines()
    data = make_data(nbytes, name)
    flags = {}
    for f, p in ((("force_generic", 0.05), ("no_bucketing", 0.3), ("force_stream", 0.5)) if CLS_BIAS else
                 (("force_generic", 0.05), ("force_global", 0.0), ("force_hot", 0.8),
                  ("no_bucketing", 0.3), ("force_stream", 0.7), ("force_chunking", 0.4)) if HOT_BIAS else
                 (("force_generic", 0.15), ("force_global", 0.15), ("force_hot", 0.25),
                  ("no_bucketing", 0.2), ("force_stream", 0.3), ("force_chunking", 0.3))):
        if rng.random() < p:
            flags[f] = True
    if rng.random() < 0.25:
        flags["force_early"] = True        # k_early for match over any LDS-resident table
    if rng.random() < 0.3:
        flags["stream_chains"] = int(rng.choice([2, 4]))  # k_stream / k_stream4 whatever the size
    if rng.random() < 0.3 and not CLS_BIAS:
        flags["lds_table_max"] = int(rng.choice([8 * 256, 40 * 256, 100 * 256, 20000, 70000]))
    try:
        exe = one_amd.Executable(blob, **flags)
    except one_amd.RedExcept as e:
        print("create refused:", name, flags, e)
        continue
    cpu = O.CpuOracle(blob)
    desc = (case, name, {k: (v if not hasattr(v, "shape") else "offsets[%d]" % (len(v) - 1)) for k, v in shape.items()}, flags, exe.info["table_kind"])
    verbs = (["match", "check", "advance", "match", "check"] if HOT_BIAS else
             ["match", "check", "scan", "search", "advance", "match_all", "collect", "replace"])
    if BLOCKS:
        verbs = ["match", "check", "match_all", "match", "check"]
    if LISTS:
        verbs = ["match_all", "collect", "match_all"]
    for verb in (rng.choice(verbs, 3) if HOT_BIAS or LISTS or BLOCKS else rng.choice(verbs, 3, replace=False)):
        sty = int(rng.choice([4, 5, 4, 5, 1, 2, 3])) if HOT_BIAS else int(rng.integers(1, 6))
        lead = int(rng.integers(0, 2))
        kw = dict(shape)
        try:
            if verb in ("match", "search"):
                fn = one_amd.match_batch if verb == "match" else one_amd.search_batch
                got = fn(exe, data, sty, lead, **kw)
                exp = cpu.batch(verb, sty, lead, data, threads=4, **kw)
                ok = all(np.array_equal(g, e) for g, e in zip(got, exp))
            elif verb in ("check", "scan"):
                if verb == "scan" and nbytes > 150_000:
                    continue  # O(n*m) on loose-start DFAs
                fn = one_amd.check_batch if verb == "check" else one_amd.scan_batch
                got = fn(exe, data, sty, lead, **kw)
                exp = cpu.batch(verb, sty, lead, data, threads=4, **kw)[0]
                ok = np.array_equal(got, exp)
            elif verb == "advance":
                n = kw["n"] if "n" in kw else len(kw["offsets"]) - 1
                st = np.full(n, one_amd.STATE_INITIAL, dtype=np.uint32)
                ost = np.full(n, O.STATE_INITIAL, dtype=np.uint32)
                got = one_amd.advance_batch(exe, data, st, **kw)
                exp = cpu.advance_batch(data, ost, **kw)
                ok = np.array_equal(got, exp)
            elif verb in ("match_all", "collect"):
                if nbytes > 150_000:
                    continue
                cap = 4
                if verb == "match_all":
                    got = one_amd.match_all_batch(exe, data, cap, bool(lead), **kw)
                    exp = cpu.match_all_batch(data, cap, do_leader=bool(lead), **kw)
                else:
                    got = one_amd.collect_batch(exe, data, cap, **kw)
                    exp = cpu.collect_batch(data, cap, **kw)
                ok = np.array_equal(got[0], exp[0])
                m = np.arange(cap)[None, :] < np.minimum(exp[0], cap).astype(np.int64)[:, None]
                ok = ok and all(np.array_equal(g[m], e[m]) for g, e in zip(got[1:], exp[1:]))
            else:
                if nbytes > 60_000:
                    continue
                repl = [b"", b"#", b"<<>>"][int(rng.integers(0, 3))]
                mx = int(rng.choice([0, 1, 3, 1 << 40]))
                counts, ooff, out = one_amd.replace_batch(exe, data, repl, sty, bool(lead), mx, **kw)
                n = len(counts)
                ok = True
                for i in rng.integers(0, n, min(n, 40)):
                    if "offsets" in kw:
                        t = data[int(kw["offsets"][i]):int(kw["offsets"][i + 1])].tobytes()
                    else:
                        t = data[i * kw["stride"]:(i + 1) * kw["stride"]].tobytes()
                    k, o = cpu.replace(t, repl, sty, bool(lead), mx)
                    if k != int(counts[i]) or o != out[int(ooff[i]):int(ooff[i + 1])].tobytes():
                        ok = False
            kern = one_amd.last_kernel()
            stats[kern] = stats.get(kern, 0) + 1
            if not ok:
                print("MISMATCH", desc, verb, sty, lead, kern)
                np.savez("gpurun_out/fuzz_fail.npz", blob=np.frombuffer(blob, dtype=np.uint8), data=data,
                         **{k: v for k, v in shape.items() if hasattr(v, "shape")})
                sys.exit(1)
        except one_amd.RedExcept as e:
            print("ERROR", desc, verb, sty, lead, e)
            sys.exit(1)
    if case % 25 == 0:
        print("case", case, "%.0fs" % (time.time() - t0), flush=True)
print("fuzz ok:", cases, "cases; kernels:", dict(sorted(stats.items())))
