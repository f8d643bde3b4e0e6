#!/usr/bin/env python3
"""Run by test_gpu_parity.py::test_text_pipeline_in_parts in a process of its own with
REDGPU_TEXT_PART_MB=1 (the library reads the variable once): redgpu_match_text[_dev] with the
text split in parts on a side stream while the caller's stream matches the part before
(redgpu.cpp textPipeline) - against oracle.split_lines + CpuOracle, as test_match_text_one_call."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import torch  # noqa: E402

import one_amd  # noqa: E402
import oracle as O  # noqa: E402
from golden_util import load_dfa  # noqa: E402
from one_amd import workloads as W  # noqa: E402

assert os.environ.get("REDGPU_TEXT_PART_MB") == "1"
rng = np.random.default_rng(61)
plant = np.frombuffer(W.URI_PLANT, dtype=np.uint8)
for name in ("uri", "uri_user", "uri_v6", "syn256"):
    blob = load_dfa(name)
    exe, cpu = one_amd.Executable(blob), O.CpuOracle(blob)
    for n_bytes, p_nl in ((5_300_001, 0.01), (2_097_152, 0.03), (3_000_000, 0.0005)):
        text = (W.random_bytes if name == "syn256" else W.alphabet_bytes)(n_bytes, 7 + n_bytes % 97).copy()
        text[text == 0x0A] = 0x20
        text[rng.random(n_bytes) < p_nl] = 0x0A
        if name != "syn256":
            for at in range(0, n_bytes - 100, 499):
                text[at:at + len(plant)] = plant
        exp = O.split_lines(text)
        n = len(exp) - 1
        keep = text[:int(exp[-1])]
        compact = keep[keep != 0x0A]
        coffs = exp - np.arange(n + 1, dtype=np.uint64)
        er, es, ee = cpu.batch("match", 4, 0, compact, offsets=coffs, threads=4)
        offs, found, r, s, e = one_amd.match_text(exe, text, 4, 0)
        assert found == n and np.array_equal(offs, exp), (name, n_bytes)
        assert np.array_equal(r, er) and np.array_equal(s, es) and np.array_equal(e, ee), (name, n_bytes)
        dev = torch.from_numpy(text).cuda()
        for cap in (n + 3, max(1, n // 2)):
            for _ in range(2):   # twice: the second call's splits wait for the first call's matches
                doffs, dcnt, dr, ds, de = one_amd.match_text(exe, dev, 4, 0, cap=cap)
            torch.cuda.synchronize()
            got = min(n, cap)
            assert int(dcnt.item()) == n
            assert np.array_equal(doffs[:got + 1].cpu().numpy().astype(np.uint64), exp[:got + 1])
            assert np.array_equal(dr[:got].cpu().numpy(), er[:got]), (name, n_bytes, cap)
            assert np.array_equal(ds[:got].cpu().numpy().astype(np.uint64), es[:got])
            assert np.array_equal(de[:got].cpu().numpy().astype(np.uint64), ee[:got])
        assert one_amd.last_kernel().startswith("k_ragged"), one_amd.last_kernel()
print("text pipeline ok")
