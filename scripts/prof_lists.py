"""A fixed sequence of matchAll launches for rocprofv3 (scripts/profile_lists.sh):
SYN-256, 2^20 lines x 64 B of random bytes, cap = $CAP (default 4), 20 launches."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, one_amd
from one_amd import _lib
from golden_util import load_dfa
l = _lib.lib()
exe = one_amd.Executable(load_dfa(os.environ.get("DFA", "syn256")))
cap = int(os.environ.get("CAP", "4"))
n, L = 1 << 20, 64
torch.manual_seed(1)
data = torch.randint(0, 256, (n * L,), dtype=torch.uint8, device="cuda")
cnt = torch.empty(n, dtype=torch.int64, device="cuda")
r = torch.empty(max(1, n * cap), dtype=torch.int32, device="cuda")
s = torch.empty(max(1, n * cap), dtype=torch.int64, device="cuda")
e = torch.empty(max(1, n * cap), dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    rc = l.redgpu_match_all_batch_dev(exe._h, 1, data.data_ptr(), None, L, n, cap, cnt.data_ptr(),
                                      r.data_ptr(), s.data_ptr(), e.data_ptr(), st)
    assert rc == 0
torch.cuda.synchronize()
print("matchAll cap", cap, "mean records/line", cnt.float().mean().item(), one_amd.last_kernel())
