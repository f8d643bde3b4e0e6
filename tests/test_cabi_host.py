"""CPU-side checks of the product's host logic: the C-ABI library loads, exports every symbol
include/redgpu.h declares, validates blobs with the reference's messages, and repacks them
(host-only handles: no HIP call, no compute)."""
import ctypes as C
import os

import numpy as np
import pytest

import one_amd
from one_amd import _lib
from golden_util import CONFIG_DFAS, load_dfa, load_kat, load_omnibus


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()


def test_library_exports_every_declared_symbol():
    names = _lib.declared_symbols()
    assert len(names) >= 14
    l = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(l, n), n
    assert _lib.lib().redgpu_version() >= 100


def test_header_validation_messages_match_reference():
    # messages of lib/Serializer.cpp:270-298, verbatim
    blob = bytearray(load_dfa("err"))
    assert one_amd.check_header(bytes(blob)) is None
    assert one_amd.check_header(bytes(blob[:100])) == "Serialized DFA: header too short"
    assert one_amd.check_header(b"\0" * 1024) == "Serialized DFA: bad magic number"  # serializer.cpp:20-24
    b = bytearray(blob); b[6] = 1
    assert one_amd.check_header(bytes(b)) == "Serialized DFA: unrecognized version"
    b = bytearray(blob); b[300] ^= 0x40
    assert one_amd.check_header(bytes(b)) == "serialized DFA: checksum mismatch"
    b = bytearray(blob); b[8:12] = b[8:12][::-1]
    assert one_amd.check_header(bytes(b)) == "serialized DFA: foreign endian-ness"


def test_create_rejects_bad_blobs_with_api_error():
    with pytest.raises(one_amd.RedExceptApi):
        one_amd.Executable(b"", device="none")
    with pytest.raises(one_amd.RedExceptApi, match="bad magic"):
        one_amd.Executable(b"\0" * 1024, device="none")
    with pytest.raises(one_amd.RedExceptApi, match="not REDA|bad magic|too short"):
        one_amd.Executable(b"this is not a REDA file", device="none")  # test/red.cpp:128-130


def test_create_rejects_out_of_range_offsets():
    """A blob whose checksum is right but whose transitions point outside the table must be
    refused (the reference would chase the pointer)."""
    import oracle as O
    blob = bytearray(load_dfa("err"))
    blob[-1] = 0xEE  # last transition entry of the last row
    import struct
    struct.pack_into("<I", blob, 8, O._orc().oracle_calc_checksum(bytes(blob), len(blob)))
    assert one_amd.check_header(bytes(blob)) is None
    with pytest.raises(one_amd.RedExceptApi, match="out of range"):
        one_amd.Executable(bytes(blob), device="none")


@pytest.mark.parametrize("name", CONFIG_DFAS)
def test_host_repack_info(name):
    import oracle as O
    blob = load_dfa(name)
    exe = one_amd.Executable(blob, device="none")
    info, ref = exe.info, O.CpuOracle(blob).info
    assert info["format"] == ref["fmt"]
    assert info["n_classes"] == ref["maxChar"] + 1
    assert info["leader_len"] == ref["leaderLen"]
    assert info["states_total"] == ref["stateCnt"]
    assert 0 < info["states_used"] <= info["states_total"]
    assert info["n_pure_dead"] <= info["first_accept"] <= info["states_used"]
    assert exe.serialized() == blob  # our own copy, byte for byte
    # copy semantics (test/red.cpp:55-78): scrambling the source after creation is harmless
    src = bytearray(blob)
    exe2 = one_amd.Executable(src, device="none")
    for i in range(len(src)):
        src[i] = 0x5A
    assert exe2.serialized() == blob


def test_table_placement():
    assert one_amd.Executable(load_dfa("syn256"), device="none").info["table_kind"] == 1
    assert one_amd.Executable(load_dfa("uri"), device="none").info["table_kind"] == 1
    # too big for LDS whole, but a real regex set with locality: hot rows in LDS (kind 6), one
    # contiguous range of device states that straddles first_accept
    for name in ("log100", "uri_v6"):
        # (left to itself the anchored signature set takes the sparse LDS form, below)
        i = one_amd.Executable(load_dfa(name), device="none", force_hot=(name == "log100")).info
        assert i["table_kind"] == 6 and 8 <= i["n_hot"] <= 254 and i["hot_coverage_ppm"] > 990000
        assert i["n_pure_dead"] <= i["hot_lo"] <= i["first_accept"] <= i["hot_lo"] + i["n_hot"]
        assert i["hot_lo"] + i["n_hot"] <= i["states_used"]
        small = one_amd.Executable(load_dfa(name), device="none", lds_table_max=16 * 256).info
        assert small["table_kind"] == 6 and small["n_hot"] == 16
    assert one_amd.Executable(load_dfa("log100"), device="none",
                              force_global=True).info["table_kind"] == 4
    # 94 % of LOG-100's transitions lead to the dead state and its walks die early: the whole
    # DFA goes to LDS in row-displacement form (252 KB class table -> 36 KB)
    i = one_amd.Executable(load_dfa("log100"), device="none").info
    assert i["table_kind"] == 7 and i["n_hot"] == 0 and i["table_bytes"] < 48 * 1024
    # the anchored signature set dies within a few bytes of arbitrary text; the loose-start
    # URI regex never does
    assert one_amd.Executable(load_dfa("log100"), device="none").info["early_death"] == 1
    assert one_amd.Executable(load_dfa("uri_v6"), device="none").info["early_death"] == 0
    # a dense random DFA has no hot set worth LDS: the whole table stays in L2
    from oracle.reda_writer import random_dfa
    i = one_amd.Executable(random_dfa(2000, 256, 1), device="none").info
    assert i["table_kind"] == 4 and i["n_hot"] == 0
    assert one_amd.Executable(load_dfa("uri"), device="none",
                              force_global=True).info["table_kind"] == 4
    # a class table that fits LDS whole stays there (only observed visits - redgpu_dfa_tune -
    # may move such a DFA to hot rows)
    i = one_amd.Executable(load_dfa("log100"), device="none", lds_table_max=400000).info
    assert i["table_kind"] == 3
    assert one_amd.Executable(load_dfa("uri_user"), device="none").info["table_kind"] == 3
    from oracle.reda_writer import random_dfa as _rd
    assert one_amd.Executable(_rd(700, 64, 4), device="none").info["table_kind"] == 3
    assert one_amd.Executable(_rd(270, 256, 4), device="none").info["table_kind"] == 2


def test_host_only_handle_refuses_compute():
    exe = one_amd.Executable(load_dfa("err"), device="none")
    with pytest.raises(one_amd.RedExceptApi, match="no device image"):
        one_amd.check_batch(exe, b"error", one_amd.styFull, offsets=[0, 5])


def test_all_golden_blobs_repack():
    rows, blobs = load_omnibus()
    n = 0
    for key in blobs.files:
        exe = one_amd.Executable(blobs[key].tobytes(), device="none")
        assert exe.info["states_used"] >= 1
        n += 1
    assert n == 566


def test_loader_cache_shares_images():
    """Same blob + device + build options -> one image (redgpu_info.image_refs); different
    options or different bytes -> their own; the image goes with its last handle."""
    blob = load_dfa("uri")
    a = one_amd.Executable(blob, device="none")
    assert a.info["image_refs"] == 1
    b = one_amd.Executable(bytearray(blob), device="none")
    assert a.info["image_refs"] == 2 and b.info["image_refs"] == 2
    c = one_amd.Executable(blob, device="none", force_global=True)      # other build options
    d = one_amd.Executable(blob, device="none", force_generic=True)     # handle-level flag only
    assert c.info["image_refs"] == 1 and c.info["table_kind"] == 4
    assert d.info["image_refs"] == 3 and a.info["table_kind"] == 1
    e = one_amd.Executable(load_dfa("err"), device="none")
    assert e.info["image_refs"] == 1
    b.close()
    d.close()
    assert a.info["image_refs"] == 1
    assert a.serialized() == blob
    a.close()
    f = one_amd.Executable(blob, device="none")                          # rebuilt from scratch
    assert f.info["image_refs"] == 1 and f.serialized() == blob


def test_suffix_closed_flag_host_side():
    """redgpu_info.suffix_closed (L = SIGMA* L, decided by language inclusion on the repacked DFA):
    set for the loose-start DFAs, clear for anchored and random ones."""
    for name, want in (("uri", 1), ("newyork", 1), ("dotstar_err", 1), ("uri_v6", 1), ("err", 0),
                       ("aab", 0), ("num3", 0), ("syn256", 0), ("log100", 0)):
        assert one_amd.Executable(load_dfa(name), device="none").info["suffix_closed"] == want, name
