"""ctypes binding of the C-ABI (include/redgpu.h) - the only way Python reaches the kernels."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libredgpu.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "redgpu.h")

OK, EAPI, EEXEC, ELIMIT, EHIP, ERCCL = 0, -1, -2, -3, -5, -6
VERB_CHECK, VERB_MATCH, VERB_SCAN, VERB_SEARCH = 0, 1, 2, 3
GATHER_PEER, GATHER_RCCL = 0, 1
DEVICE_CURRENT, DEVICE_NONE = -1, -2
F_FORCE_GENERIC, F_FORCE_GLOBAL, F_FORCE_HOT, F_NO_BUCKETING, F_FORCE_STREAM = 1, 2, 4, 8, 16
F_NO_CHUNKING, F_FORCE_CHUNKING = 32, 64
F_STREAM_CHAINS_2, F_STREAM_CHAINS_4, F_FORCE_EARLY, F_FORCE_LEAN, F_LEAN_CHAINS_4 = 128, 256, 512, 1024, 2048
F_FORCE_PIECES = 4096


class Opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("lds_table_max", C.c_uint32), ("flags", C.c_uint32),
                ("reserved", C.c_uint32 * 5)]


class Info(C.Structure):
    _fields_ = [("format", C.c_uint32), ("n_classes", C.c_uint32), ("leader_len", C.c_uint32),
                ("states_total", C.c_uint32), ("states_used", C.c_uint32),
                ("n_pure_dead", C.c_uint32), ("first_accept", C.c_uint32),
                ("table_kind", C.c_uint32), ("table_bytes", C.c_uint64),
                ("max_result", C.c_int32), ("device", C.c_int32), ("checksum", C.c_uint32),
                ("fast_path", C.c_uint32), ("n_hot", C.c_uint32), ("hot_lo", C.c_uint32),
                ("hot_coverage_ppm", C.c_uint32), ("early_death", C.c_uint32),
                ("forgetful", C.c_uint32), ("image_refs", C.c_uint32),
                ("suffix_closed", C.c_uint32)]


class BatchDesc(C.Structure):  # redgpu_batch
    _fields_ = [("data", C.c_void_p), ("offsets", C.c_void_p), ("stride", C.c_uint64),
                ("n", C.c_uint64), ("result", C.c_void_p), ("start", C.c_void_p),
                ("end", C.c_void_p)]


def build(force: bool = False) -> str:
    """Compile libredgpu.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-s", "-C", os.path.join(HERE, "csrc"), "clean"], check=True)
    subprocess.run(["make", "-s", "-j4", "-C", os.path.join(HERE, "csrc")], check=True)
    return LIB_PATH


_lib = None


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 (SONAME
    libamdhip64.so.7, the same SONAME libredgpu.so needs): when torch is installed, map that
    copy first so the kernels, torch tensors and torch streams all live in ONE runtime.  Loading
    the system runtime first and torch's second leaves the process with two, and the second
    one finds no GPU."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> C.CDLL:
    """Loads the HIP extension.  Fails loudly when it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "one_amd: %s is missing - build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C one_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        _share_torch_hip_runtime()
        l = C.CDLL(LIB_PATH)
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
        l.redgpu_version.restype = C.c_int
        l.redgpu_last_error.restype = C.c_char_p
        l.redgpu_last_kernel.restype = C.c_char_p
        l.redgpu_reda_check.restype = C.c_int
        l.redgpu_reda_check.argtypes = [vp, C.c_size_t, C.POINTER(C.c_char_p)]
        l.redgpu_dfa_create.restype = C.c_int
        l.redgpu_dfa_create.argtypes = [vp, C.c_size_t, C.POINTER(Opts), C.POINTER(vp)]
        l.redgpu_dfa_destroy.restype = None
        l.redgpu_dfa_destroy.argtypes = [vp]
        l.redgpu_dfa_info.restype = C.c_int
        l.redgpu_dfa_info.argtypes = [vp, C.POINTER(Info)]
        l.redgpu_dfa_serialized.restype = C.c_int
        l.redgpu_dfa_serialized.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        for name in ("redgpu_check_batch", "redgpu_scan_batch"):
            f = getattr(l, name)
            f.restype = C.c_int
            f.argtypes = [vp, i32, i32, vp, vp, u64, u64, vp]
        for name in ("redgpu_match_batch", "redgpu_search_batch"):
            f = getattr(l, name)
            f.restype = C.c_int
            f.argtypes = [vp, i32, i32, vp, vp, u64, u64, vp, vp, vp]
        for name in ("redgpu_check_batch_dev", "redgpu_scan_batch_dev"):
            f = getattr(l, name)
            f.restype = C.c_int
            f.argtypes = [vp, i32, i32, vp, vp, u64, u64, vp, vp]
        for name in ("redgpu_match_batch_dev", "redgpu_search_batch_dev"):
            f = getattr(l, name)
            f.restype = C.c_int
            f.argtypes = [vp, i32, i32, vp, vp, u64, u64, vp, vp, vp, vp]
        for name in ("redgpu_check_batches_dev", "redgpu_match_batches_dev"):
            f = getattr(l, name)
            f.restype = C.c_int
            f.argtypes = [vp, i32, i32, C.POINTER(BatchDesc), C.c_uint32, vp]
        l.redgpu_collect_batch.restype = C.c_int
        l.redgpu_collect_batch.argtypes = [vp, vp, vp, u64, u64, u64, vp, vp, vp, vp]
        l.redgpu_collect_batch_dev.restype = C.c_int
        l.redgpu_collect_batch_dev.argtypes = [vp, vp, vp, u64, u64, u64, vp, vp, vp, vp, vp]
        l.redgpu_match_all_batch.restype = C.c_int
        l.redgpu_match_all_batch.argtypes = [vp, i32, vp, vp, u64, u64, u64, vp, vp, vp, vp]
        l.redgpu_match_all_batch_dev.restype = C.c_int
        l.redgpu_match_all_batch_dev.argtypes = [vp, i32, vp, vp, u64, u64, u64, vp, vp, vp, vp,
                                                 vp]
        l.redgpu_advance_batch.restype = C.c_int
        l.redgpu_advance_batch.argtypes = [vp, vp, vp, u64, u64, vp, vp]
        l.redgpu_advance_batch_dev.restype = C.c_int
        l.redgpu_advance_batch_dev.argtypes = [vp, vp, vp, u64, u64, vp, vp, vp]
        l.redgpu_dfa_tune.restype = C.c_int
        l.redgpu_dfa_tune.argtypes = [vp, vp, vp, u64, u64]
        l.redgpu_dfa_tune_dev.restype = C.c_int
        l.redgpu_dfa_tune_dev.argtypes = [vp, vp, vp, u64, u64, vp]
        l.redgpu_replace_batch.restype = C.c_int
        l.redgpu_replace_batch.argtypes = [vp, i32, i32, vp, vp, u64, u64, vp, u64, u64, vp, vp,
                                           vp, u64]
        l.redgpu_replace_batch_dev.restype = C.c_int
        l.redgpu_replace_batch_dev.argtypes = [vp, i32, i32, vp, vp, u64, u64, vp, u64, u64, vp,
                                               vp, vp, u64, vp]
        l.redgpu_split_lines.restype = C.c_int
        l.redgpu_split_lines.argtypes = [vp, vp, u64, C.c_uint8, vp, u64, vp]
        l.redgpu_split_lines_dev.restype = C.c_int
        l.redgpu_split_lines_dev.argtypes = [vp, vp, u64, C.c_uint8, vp, u64, vp, vp]
        l.redgpu_check_text_dev.restype = C.c_int
        l.redgpu_check_text_dev.argtypes = [vp, C.c_int, C.c_int, vp, u64, C.c_uint8, vp, u64, vp,
                                            vp, vp]
        l.redgpu_match_text_dev.restype = C.c_int
        l.redgpu_match_text_dev.argtypes = [vp, C.c_int, C.c_int, vp, u64, C.c_uint8, vp, u64, vp,
                                            vp, vp, vp, vp]
        l.redgpu_match_text.restype = C.c_int
        l.redgpu_match_text.argtypes = [vp, C.c_int, C.c_int, vp, u64, C.c_uint8, vp, u64, vp,
                                        vp, vp, vp]
        l.redgpu_diag_read_dev.restype = C.c_int
        l.redgpu_diag_read_dev.argtypes = [vp, vp, u64, vp, vp]
        l.redgpu_diag_lds_dev.restype = C.c_int
        l.redgpu_diag_lds_dev.argtypes = [vp, C.c_uint32, vp, C.POINTER(u64), vp]
        l.redgpu_diag_lines_dev.restype = C.c_int
        l.redgpu_diag_lines_dev.argtypes = [vp, vp, u64, u64, vp, vp, vp, vp, vp]
        l.redgpu_diag_l2_dev.restype = C.c_int
        l.redgpu_diag_l2_dev.argtypes = [vp, vp, C.c_uint32, vp, C.POINTER(u64), vp]
        l.redgpu_diag_walked_dev.restype = C.c_int
        l.redgpu_diag_walked_dev.argtypes = [vp, i32, vp, vp, u64, u64, vp, vp]
        l.redgpu_host_register.restype = C.c_int
        l.redgpu_host_register.argtypes = [vp, C.c_size_t]
        l.redgpu_host_unregister.restype = C.c_int
        l.redgpu_host_unregister.argtypes = [vp]
        l.redgpu_thread_release.restype = None
        l.redgpu_scratch_entries.restype = u64
        l.redgpu_host_route_counts.restype = None
        l.redgpu_host_route_counts.argtypes = [vp]
        l.redgpu_group_create.restype = C.c_int
        l.redgpu_group_create.argtypes = [vp, C.c_size_t, C.POINTER(Opts), C.POINTER(C.c_int32),
                                          C.c_uint32, C.POINTER(vp)]
        l.redgpu_group_destroy.restype = None
        l.redgpu_group_destroy.argtypes = [vp]
        l.redgpu_group_size.restype = C.c_uint32
        l.redgpu_group_size.argtypes = [vp]
        l.redgpu_group_member.restype = vp
        l.redgpu_group_member.argtypes = [vp, C.c_uint32]
        l.redgpu_group_plan.restype = C.c_int
        l.redgpu_group_plan.argtypes = [vp, vp, u64, u64, vp]
        l.redgpu_group_batch.restype = C.c_int
        l.redgpu_group_batch.argtypes = [vp, i32, i32, i32, vp, vp, u64, u64, vp, vp, vp]
        l.redgpu_group_batch_dev.restype = C.c_int
        l.redgpu_group_batch_dev.argtypes = [vp, i32, i32, i32, vp, vp, u64, vp, vp, vp, vp, i32,
                                             vp, vp]
        l.redgpu_records_bytes.restype = u64
        l.redgpu_records_bytes.argtypes = [u64, i32, i32, i32]
        l.redgpu_records_pack_dev.restype = C.c_int
        l.redgpu_records_pack_dev.argtypes = [C.c_int32, vp, vp, vp, u64, i32, i32, vp, vp]
        l.redgpu_records_unpack_dev.restype = C.c_int
        l.redgpu_records_unpack_dev.argtypes = [C.c_int32, vp, u64, i32, i32, vp, vp, vp, vp]
        _lib = l
    return _lib


def declared_symbols() -> list[str]:
    """Every function include/redgpu.h declares (parsed from the header text)."""
    import re
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(redgpu_[a-z_]+)\s*\(", text)))
