#!/usr/bin/env python3
"""lab: the host form of match_text / split_lines / match_batch many times over small inputs,
every answer checked (an intermittent wrong answer was seen once in the C++ mirror test)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import one_amd
from golden_util import load_dfa
exe = one_amd.Executable(load_dfa("newyork"))
text = np.frombuffer(b"New\nnothing here\nI love New York.\n\nNew York", dtype=np.uint8)
want_off = [0, 4, 17, 34, 35]
want_res = [1, 0, 2, 0]
rng = np.random.default_rng(5)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20000):
    if it % 7 == 0:   # other traffic in between, of changing size
        n = int(rng.integers(1, 3000))
        one_amd.match_batch(exe, rng.integers(32, 127, n * 64, dtype=np.uint8), 4, 1, stride=64)
    offs, found, r, s, e = one_amd.match_text(exe, text, 4, 1)
    if found != 4 or offs.tolist() != want_off or r.tolist() != want_res:
        bad += 1
        print("iteration", it, "found", found, "offs", offs.tolist(), "res", r.tolist(), flush=True)
        if bad > 5:
            break
print("done, wrong answers:", bad)
