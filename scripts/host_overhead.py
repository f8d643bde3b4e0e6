"""Host-side cost of one redgpu_match_batch_dev call from Python (tiny batches, 1..3 streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, one_amd
from one_amd import _lib
blob = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "dfas", "syn256.reda"), "rb").read()
exe = one_amd.Executable(blob)
n, L = 2048, 64
data = torch.zeros(n * L, dtype=torch.uint8, device="cuda")
res = torch.empty(n, dtype=torch.int32, device="cuda")
st = torch.empty(n, dtype=torch.int64, device="cuda")
en = torch.empty(n, dtype=torch.int64, device="cuda")
fn = _lib.lib().redgpu_match_batch_dev
for ns in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    calls = [(exe._h, 4, 0, data.data_ptr(), None, L, n, res.data_ptr(), st.data_ptr(), en.data_ptr(),
              streams[i % ns].cuda_stream) for i in range(6)]
    for i in range(200):
        fn(*calls[i % 6])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 3000
    for i in range(K):
        fn(*calls[i % 6])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("streams %d: host issue %.2f us/call, incl. drain %.2f us/call" % (ns, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
