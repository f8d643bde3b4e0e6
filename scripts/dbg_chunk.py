import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, one_amd, oracle as O
from one_amd import workloads as W
from golden_util import load_dfa
blob = load_dfa("uri"); cpu = O.CpuOracle(blob)
n, L = 5, 65536
data = W.alphabet_bytes(n * L, 61).copy()
p = np.frombuffer(W.URI_PLANT, dtype=np.uint8)
for k in range(0, data.size - 200, 777): data[k:k + len(p)] = p
for kw in ({}, {"no_chunking": True}):
    exe = one_amd.Executable(blob, **kw)
    r, s, e = one_amd.match_batch(exe, data, 4, 0, stride=L, n=n)
    print(kw, one_amd.last_kernel(), r, s, e)
print(cpu.batch("match", 4, 0, data, stride=L, n=n))
