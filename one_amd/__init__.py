"""one_amd - MI355X-native DFA match execution for RED (zezax/one quol/red).

Only the hot path lives here: `csrc/` (gfx950 HIP kernels + the C-ABI of include/redgpu.h),
`matcher.py` (host-side mirror of the reference's Executable / Style / check / match / scan),
`sharding.py` (one process per GPU, contiguous shards, RCCL result gather) and `workloads.py`
(synthetic inputs of the BASELINE configs).  Importing the package does not load the HIP
extension; the first call does, and fails loudly if it is not built.
"""
from .matcher import (Executable, Group, Style, RedExcept, RedExceptApi, RedExceptExec,  # noqa: F401
                      RedExceptLimit, RedExceptHip, check, check_batch, check_header, match,
                      match_batch, match_batches, check_batches, batch_descs, scan, scan_batch, search, search_batch, collect,
                      collect_batch, match_all, match_all_batch, advance_batch,
                      StatefulMatcher, STATE_INITIAL, split_lines, match_text, replace, replace_batch, last_kernel, styInstant, styFirst,
                      styTangent, styLast, styFull)

__all__ = ["Executable", "Group", "Style", "check", "match", "scan", "check_batch", "match_batch",
           "scan_batch", "search", "search_batch", "collect", "collect_batch", "check_header",
           "match_all", "match_all_batch", "advance_batch", "StatefulMatcher", "STATE_INITIAL",
           "split_lines", "match_text", "replace", "replace_batch", "last_kernel", "match_batches",
           "check_batches", "batch_descs"]
