#!/usr/bin/env python3
"""bench.py - GB/s of input scanned by the MI355X DFA match-execution path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path (match<styLast,false> = "matchLong", full Outcome:
result + start + end) over one batch of synthetic input already resident in HBM.
Workload at every N = BASELINE.json configs[1], per GPU: the 256-state / 256-class DFA
(tests/golden/dfas/syn256.reda, produced by the reference's minimizer + serializer) over
2^20 lines x 64 B = 64 MiB of uniform random bytes; >= 5 distinct input buffers are rotated so
that no step finds its input in the 256 MiB Infinity Cache.  N > 1: one process per GPU,
weak scaling (every rank scans its own shard, no data-path collective); the final step's
per-line Outcomes are gathered to rank 0 over RCCL (compact wire records, widened on rank 0)
once, at the end, INSIDE the timed region - "RCCL over xGMI only for the final result gather".
(A gather after every step cannot scale for 64-byte lines: 3 bytes of results per 64 bytes of
input is 1/21 of the scan rate per GPU, well above what an xGMI link carries - DESIGN.md.)

Prints ONE JSON line on rank 0 (see the driver's contract) with `roofline` (dominant kernel
vs the 8 TB/s HBM peak, timed with events on the launch stream) and `cpu_baseline` (the
reference's CPU matcher on this box's host cores, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--lines", type=int, default=1 << 20)
    ap.add_argument("--line-len", type=int, default=64)
    ap.add_argument("--dfa", default="syn256", choices=["syn256", "uri"])
    ap.add_argument("--buffers", type=int, default=6)
    ap.add_argument("--no-start", action="store_true", help="outputs result + end only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the steps are issued on round-robin (independent batches "
                         "overlap: one step's ramp-up hides under the previous step's tail)")
    return ap.parse_args()


def make_inputs(args, rank, torch, W):
    """Distinct per-rank, per-buffer synthetic batches, generated on the host with the same
    counter-based generator the tests use, then made resident in HBM."""
    import numpy as np
    bufs, hosts = [], []
    for k in range(args.buffers):
        seed = 42 + 1000 * rank + k
        if args.dfa == "syn256":
            h = W.fixed_lines(args.lines, args.line_len, seed, alphabet=False)
        else:
            h = W.fixed_lines(args.lines, args.line_len, seed, plant=W.URI_PLANT)
        t = torch.from_numpy(h).cuda()
        bufs.append(t)
        hosts.append(h if k == 0 else None)
    return bufs, hosts[0]


def cpu_baseline(args, blob, host_batch):
    """The reference's own CPU matcher (oracle/_ref, compiled from the reference sources) if
    its prebuilt .so travelled with the repo, else the C restatement; all host cores, one
    contiguous shard per thread as tools/thr_red.cpp:86-91 does.  Bounded to ~cpu_seconds."""
    import oracle
    try:
        cores = len(os.sched_getaffinity(0))  # the box's CPU share, not the machine's
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    if oracle.have_ref():
        eng, kind = oracle.Reference(blob), "reference"
    else:
        eng, kind = oracle.CpuOracle(blob), "port"
    n, L = args.lines, args.line_len
    eng.batch("match", "last", 0, host_batch[: 4096 * L], stride=L, n=4096, threads=cores)
    t0 = time.perf_counter()
    passes = 0
    while True:
        eng.batch("match", "last", 0, host_batch, stride=L, n=n, threads=cores)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds or passes >= 200:
            break
    gbs = passes * n * L / dt / 1e9
    # single-thread figure for context
    t1 = time.perf_counter()
    m = min(n, 1 << 18)
    eng.batch("match", "last", 0, host_batch[: m * L], stride=L, n=m, threads=1)
    one = m * L / (time.perf_counter() - t1) / 1e9
    return {"value": round(gbs, 3), "unit": "GB/s", "cores": cores, "kind": kind,
            "sample": "%d passes over the full %d x %d B batch, match<styLast,false>, "
                      "%d threads" % (passes, n, L, cores),
            "single_thread_GBps": round(one, 3)}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD process
        # (nothing here has touched the GPU yet) and leave with its exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    import numpy as np
    import torch
    import one_amd
    from one_amd import workloads as W

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    backend = os.environ.get("BENCH_BACKEND", "nccl")  # "gloo": single-GPU rehearsal of N > 1
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    with open(os.path.join(ROOT, "tests", "golden", "dfas", args.dfa + ".reda"), "rb") as f:
        blob = f.read()
    exe = one_amd.Executable(blob, device=local_rank)
    info = exe.info
    n, L = args.lines, args.line_len
    want_start = not args.no_start

    bufs, host0 = make_inputs(args, rank, torch, W)
    nout = 3
    outs = [(torch.empty(n, dtype=torch.int32, device="cuda"),
             torch.empty(n, dtype=torch.int64, device="cuda") if want_start else None,
             torch.empty(n, dtype=torch.int64, device="cuda")) for _ in range(nout)]

    streams = [torch.cuda.Stream() for _ in range(args.streams)] if args.streams > 1 else None
    nout = max(nout, 2 * args.streams)
    while len(outs) < nout:
        outs.append((torch.empty(n, dtype=torch.int32, device="cuda"),
                     torch.empty(n, dtype=torch.int64, device="cuda") if want_start else None,
                     torch.empty(n, dtype=torch.int64, device="cuda")))

    # The timed loop calls the C-ABI entry point directly (redgpu_match_batch_dev) with argument
    # tuples built once: at ~20 us per kernel the Python conveniences of one_amd.match_batch
    # (tensor checks, allocation) would make the loop host-bound.
    from one_amd import _lib
    fn = _lib.lib().redgpu_match_batch_dev
    cur_stream = torch.cuda.current_stream().cuda_stream
    period = len(bufs) * nout * max(1, args.streams)
    calls = []
    for i in range(period):
        r_, s_, e_ = outs[i % nout]
        st = streams[i % len(streams)].cuda_stream if streams else cur_stream
        calls.append((exe._h, int(one_amd.styLast), 0, bufs[i % len(bufs)].data_ptr(), None, L, n,
                      r_.data_ptr(), s_.data_ptr() if s_ is not None else None, e_.data_ptr(),
                      st))

    def step(i):
        rc = fn(*calls[i % period])
        if rc != 0:
            raise RuntimeError(_lib.lib().redgpu_last_error().decode())
        return outs[i % nout]

    # ---- correctness gate: buffer 0 bit-exact against the CPU oracle ------------------------
    import oracle
    r, s, e = one_amd.match_batch(exe, bufs[0], one_amd.styLast, False, stride=L, n=n,
                                  want_start=want_start, out=outs[0])
    torch.cuda.synchronize()
    try:
        vthreads = max(1, min(len(os.sched_getaffinity(0)) // max(1, world), 32))
    except AttributeError:
        vthreads = 8
    er, es, ee = oracle.CpuOracle(blob).batch("match", "last", 0, host0, stride=L, n=n,
                                              threads=vthreads)
    bit_exact = bool(np.array_equal(r.cpu().numpy(), er) and
                     np.array_equal(e.cpu().numpy().astype(np.uint64), ee) and
                     (not want_start or np.array_equal(s.cpu().numpy().astype(np.uint64), es)))
    kernel_name = one_amd.last_kernel()

    # ---- multi-GPU result gather (compact wire form, widened on rank 0) ---------------------
    gather = None
    if world > 1:
        from one_amd import sharding
        gather = sharding.FinalGather(info["max_result"], L, want_start,
                                      via_host=(backend != "nccl"), equal_counts=True)

    if gather:
        # communicator set-up (RCCL connects lazily on the first collective) is not a step:
        # one untimed gather of the correctness-gate outputs, whatever --warmup is
        gather.push(outs[0])
        gather.flush()
    for i in range(args.warmup):
        o = step(i)
        if gather:
            gather.push(o)
    if streams is not None:
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    if gather and args.warmup:
        gather.flush()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        o = step(i)
        if gather:
            gather.push(o)
    if streams is not None:
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    ev1.record()
    if gather:
        gather.flush()  # the final result gather: inside the timed region
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    region_ms = ev0.elapsed_time(ev1)

    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- dominant kernel's average launch duration, HIP events on its launch stream -----------
    # One stream, launches back to back, one event before the first and one after the last:
    # (elapsed / launches) is the kernel's duration plus the ~1 us dependent-launch gap, and is
    # what rocprofv3 --kernel-trace --stats reports for the same command with --streams 1
    # (profiles/r01_kernel_stats_1stream.csv).  Per-launch event PAIRS are kept as a second
    # figure; they include ~4 us of launch latency each.
    torch.cuda.synchronize()
    nk = max(20, min(args.steps, 200))
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    single = [c[:-1] + (cur_stream,) for c in calls]
    for i in range(10):
        fn(*single[i % period])
    k0.record()
    for i in range(nk):
        fn(*single[i % period])
    k1.record()
    torch.cuda.synchronize()
    k_serial_ms = k0.elapsed_time(k1) / nk
    kms = []
    for i in range(min(args.steps, 50)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(*single[i % period])
        b.record()
        kms.append((a, b))
    torch.cuda.synchronize()
    per_launch = sorted(a.elapsed_time(b) for a, b in kms)
    k_avg_ms = sum(per_launch) / len(per_launch)
    k_med_ms = per_launch[len(per_launch) // 2]
    back_to_back_ms = region_ms / args.steps

    # ---- read-bandwidth calibration in the same session -----------------------------
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    rd = _lib.lib().redgpu_diag_read_dev
    nb = n * L
    for i in range(5):
        rd(exe._h, bufs[i % len(bufs)].data_ptr(), nb, sink.data_ptr(), cur_stream)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for i in range(60):
        rd(exe._h, bufs[i % len(bufs)].data_ptr(), nb, sink.data_ptr(), cur_stream)
    c1.record()
    torch.cuda.synchronize()
    read_ceiling = 60 * nb / (c0.elapsed_time(c1) * 1e-3) / 1e9

    bytes_per_step = n * L
    out_bytes = n * (4 + 8 + (8 if want_start else 0))
    value = world * args.steps * bytes_per_step / elapsed / 1e9
    # roofline: algorithmic input bytes per launch / the kernel's average launch duration
    # (k_serial_ms above; it still contains the ~1 us gap between dependent launches).
    kernel_ms = k_serial_ms
    achieved = bytes_per_step / (kernel_ms * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if world == 1 and args.dfa == "syn256" and want_start and os.path.exists(pmc_path):
        # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same kernel and
        # workload (FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE; scripts/profile_gpu.sh)
        traffic = json.load(open(pmc_path)).get("hbm_traffic_bytes_per_launch")
    line = {
        "metric": "GB/s input scanned (and Minput/s) for fixed DFA",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: %s DFA (%d states used / %d classes, REDA fmtDirect%d), "
                        "%d lines x %d B per GPU, match<styLast,false> -> %s" %
                        (args.dfa, info["states_used"], info["n_classes"], info["format"], n, L,
                         "result+start+end" if want_start else "result+end"),
            "lines_per_gpu": n, "line_len": L, "rotating_input_buffers": len(bufs),
            "streams": args.streams,
            "sharding": "contiguous shard per GPU, no data-path collective" +
                        ("; one final RCCL gather of the last step's Outcomes to rank 0"
                         if world > 1 else ""),
        },
        "minputs_per_s": round(world * args.steps * n / elapsed / 1e6, 1),
        "bit_exact": bit_exact,
        "kernel": kernel_name,
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_per_step,
            "output_bytes_per_launch": out_bytes,
            "kernel_ms": round(k_serial_ms, 5),
            "kernel_ms_how": "HIP events around %d back-to-back launches on one stream" % nk,
            "kernel_ms_event_pair_avg": round(k_avg_ms, 5),
            "kernel_ms_event_pair_median": round(k_med_ms, 5),
            "timed_region_ms_per_step": round(back_to_back_ms, 5),
            "total_traffic_GBps": round((bytes_per_step + out_bytes) / (kernel_ms * 1e-3) / 1e9, 1),
            "measured_read_ceiling_GBps": round(read_ceiling, 1),
            "measured_read_ceiling_how": "k_diag_read: 60 back-to-back streaming reads of the "
                                         "same %d-byte input buffers, one stream" % nb,
            "lds_roof_GBps": 4400.0,
            "lds_roof_note": "one ds_read_u8 per byte at 7.0 LDS cycles per 64-lane gather "
                             "(SQ_LDS_IDX_ACTIVE/SQ_INSTS_LDS), measured 15 us per 64 Mi lookups",
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args, blob, host0)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()
    if not bit_exact:
        raise SystemExit("bench: GPU output differs from the oracle - number is invalid")


if __name__ == "__main__":
    main()
