#!/usr/bin/env python3
"""bench.py - GB/s of input scanned by the MI355X DFA match-execution path.

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4}] [--dfa NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch of synthetic input already resident in HBM.
Workloads are BASELINE.json's configs, all through the C-ABI (include/redgpu.h):

  --config 1 (default at N = 1)  SYN-256 (256 states / 256 classes) or --dfa uri (URI-D on text),
             2^20 lines x 64 B per GPU, match<styLast,false> -> result + start + end.
  --config 2 (default at N > 1)  the same DFAs, 2^21 lines x 4 KiB = 8 GiB per GPU (one GPU's
             shard of the 16 M-line batch), full Outcome; N > 1: EVERY step's Outcomes are
             gathered to rank 0 over RCCL (compact records, pipelined against the next scans).
  --config 3 LOG-100 (100 signatures -> one DFA), 2^23 ragged lines of 32..256 B,
             matchLong = match<styLast,true>; the bytes the walk actually reads are reported
             beside the bytes of the lines (half of the lines die in their first bytes).
  --config 4 SYN-4K (4,096 states / 256 classes, 2 MiB table in L2) or --dfa uri_v6,
             65,536 inputs x 64 KiB = 4 GiB.

Prints ONE JSON line on rank 0: the driver's contract plus `roofline` (dominant kernel: algorithmic
bytes per launch / its average launch duration, HIP events on the launch stream, against the
8 TB/s HBM peak; every other number in it is measured by THIS run - the read ceiling and the
LDS gather roof by calibration kernels - except `traffic`, which is null unless a PMC summary
of the same command is on disk, and then says where it comes from) and `cpu_baseline` (the
reference's own CPU matcher on this box's host cores, rank 0, N = 1 only).  After the timed
region the last output of every stream is checked against the CPU oracle (bit-exact or the
run is invalid).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GLOOKUPS = 269.5  # same guide: L2 ~34.5 TB/s aggregate = 269.5 G 128-byte requests / s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[k]; default 1 at N = 1, 2 at N > 1")
    ap.add_argument("--dfa", default=None,
                    help="syn256 | uri (configs 1, 2), log100 (3), syn4k | uri_v6 (4)")
    ap.add_argument("--lines", type=int, default=None, help="override the config's line count")
    ap.add_argument("--buffers", type=int, default=None)
    ap.add_argument("--no-start", action="store_true", help="outputs result + end only")
    ap.add_argument("--rehearse-gather", action="store_true",
                    help="one GPU: run the N > 1 loop (every step's records packed, gathered over "
                         "a ONE-rank RCCL group and widened on the root) to price the root's share")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tune", action="store_true",
                    help="hot-row DFAs: keep the static choice of LDS-resident rows (default: "
                         "redgpu_dfa_tune on a 4 MiB sample of the input, outside the timed region)")
    ap.add_argument("--no-calibration", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the steps are issued on round-robin (1 = one stream; more "
                         "is a diagnostic, not the default: overlap of consecutive batches is the "
                         "library's job, see --per-call)")
    ap.add_argument("--per-call", type=int, default=None,
                    help="steps handed to the library per call of redgpu_match_batches_dev (the "
                         "caller's loop over its inputs, tools/bench.cpp:60-71, given to the "
                         "library whole); default min(steps, 32) for config 1, 1 elsewhere. "
                         "1 = one redgpu_match_batch_dev call per step")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
class Workload:
    """One config: device-resident rotating inputs, C-ABI call tuples, checker, CPU sample."""

    def __init__(self, args, rank, torch, W, one_amd, oracle):
        self.args, self.rank, self.torch, self.W, self.one_amd, self.oracle = \
            args, rank, torch, W, one_amd, oracle
        c = args.config
        defaults = {1: "syn256", 2: "syn256", 3: "log100", 4: "syn4k"}
        self.dfa_name = args.dfa or defaults[c]
        from tests.golden_util import load_dfa
        self.blob = load_dfa(self.dfa_name)
        self.exe = one_amd.Executable(self.blob, device=torch.cuda.current_device())
        self.info = self.exe.info
        self.want_start = not args.no_start
        self.style, self.lead = int(one_amd.styLast), 0
        self.offsets_dev = None
        self.ragged = False
        self.text = self.dfa_name not in ("syn256", "syn4k")
        if c == 1:
            self.n, self.L = args.lines or (1 << 20), 64
            self.nbuf = args.buffers or 6
            self.label = "configs[1]"
        elif c == 2:
            self.n, self.L = args.lines or (1 << 21), 4096
            self.nbuf = args.buffers or 2
            self.label = "configs[2] (one GPU's shard of 16 M lines x 4 KiB)"
        elif c == 3:
            self.n, self.L = args.lines or (1 << 23), 0
            self.nbuf = 1
            self.lead = 1
            self.ragged = True
            self.label = "configs[3]"
        else:
            self.n, self.L = args.lines or (1 << 16), 1 << 16
            self.nbuf = args.buffers or 2
            self.label = "configs[4]"
        self._build_inputs()
        self._expected = {}
        self.tuned = None
        if not args.no_tune and self.info["table_kind"] == 6 and not self.ragged:
            # REDGPU_TAB_HOT_ROWS: re-rank the LDS-resident rows by a sample of the input (outside
            # the timed region, like the upload; redgpu_dfa_tune)
            # (the sample is walked as 16 KiB pieces: the visit histogram wants lanes, but a walk
            # that restarts every KiB over-counts the states around the initial one)
            sample = min(self.n * self.L, 4 << 20)
            piece = min(self.L, 16384)
            self.info = self.exe.tune(self.bufs[0][:sample], stride=piece, n=sample // piece)
            self.tuned = "hot rows re-ranked on the first %d bytes of buffer 0" % sample

    # -- inputs ----------------------------------------------------------------------------
    def _build_inputs(self):
        torch, W, a = self.torch, self.W, self.args
        c, n, L = a.config, self.n, self.L
        self.bufs, self.hosts = [], []
        if c == 1:
            for k in range(self.nbuf):
                seed = 42 + 1000 * self.rank + k
                h = (W.fixed_lines(n, L, seed, plant=W.URI_PLANT) if self.text else
                     W.fixed_lines(n, L, seed, alphabet=False))
                self.bufs.append(torch.from_numpy(h).cuda())
                self.hosts.append(h)
            self.in_bytes = n * L
        elif c in (2, 4):
            # generated on the device (8 / 4 GiB per buffer); the checker pulls samples back
            alpha = torch.from_numpy(W.ALPHABET47.copy()).cuda()
            plant = torch.from_numpy(__import__("numpy").frombuffer(W.URI_PLANT, dtype="uint8").copy()).cuda()
            for k in range(self.nbuf):
                g = torch.Generator(device="cuda").manual_seed(3 + 1000 * self.rank + k)
                buf = torch.empty(n * L, dtype=torch.uint8, device="cuda")
                piece = 1 << 28
                for lo in range(0, n * L, piece):
                    hi = min(n * L, lo + piece)
                    v = torch.randint(0, 256, (hi - lo,), generator=g, device="cuda",
                                      dtype=torch.uint8)
                    buf[lo:hi] = alpha[(v % 47).long()] if self.text else v
                    del v
                if self.text:  # a URL in every 8th line, as configs[1]'s text
                    rows = buf.view(n, L)[::8]
                    at = 1000 if L > 2000 else 8
                    rows[:, at:at + plant.numel()] = plant
                self.bufs.append(buf)
            self.in_bytes = n * L
        else:
            data, offsets = W.ragged_lines(n, 32, 256, 4 + self.rank, heads=W.log100_heads(),
                                           head_every=2)
            self.hosts.append((data, offsets))
            self.bufs.append(torch.from_numpy(data).cuda())
            self.offsets_dev = torch.from_numpy(offsets.astype("int64")).cuda()
            self.in_bytes = int(offsets[-1])
        nout = max(3, 2 * (a.streams or 1), a.per_call or 1)
        # (zeroed, not just reserved: every page of every output set has been written once before
        # any clock starts, whichever sets the W warm-up steps happen to use)
        self.outs = [(torch.zeros(n, dtype=torch.int32, device="cuda"),
                      torch.zeros(n, dtype=torch.int64, device="cuda") if self.want_start else None,
                      torch.zeros(n, dtype=torch.int64, device="cuda")) for _ in range(nout)]
        self.out_bytes = n * (4 + 8 + (8 if self.want_start else 0)) + (8 * n if self.ragged else 0)

    # -- one step --------------------------------------------------------------------------
    def call_tuple(self, i, stream):
        r, s, e = self.outs[i % len(self.outs)]
        buf = self.bufs[i % len(self.bufs)]
        return (self.exe._h, self.style, self.lead, buf.data_ptr(),
                self.offsets_dev.data_ptr() if self.ragged else None,
                0 if self.ragged else self.L, self.n, r.data_ptr(),
                s.data_ptr() if s is not None else None, e.data_ptr(), stream)

    # -- checker: device outputs of (buffer b, output set o) against the CPU oracle ----------
    def verify(self, b, o):
        import numpy as np
        torch = self.torch
        cpu = self.oracle.CpuOracle(self.blob)
        r, s, e = self.outs[o]
        try:
            threads = max(1, min(len(os.sched_getaffinity(0)), 32))
        except AttributeError:
            threads = 8

        def same(idx, er, es, ee):
            rr = r if idx is None else r[idx]
            ok = np.array_equal(rr.cpu().numpy(), er)
            ok = ok and np.array_equal((e if idx is None else e[idx]).cpu().numpy().astype(np.uint64), ee)
            if self.want_start:
                ok = ok and np.array_equal((s if idx is None else s[idx]).cpu().numpy().astype(np.uint64), es)
            return bool(ok)

        c = self.args.config
        if c == 1:
            if b not in self._expected:  # one oracle pass per input buffer, however many outputs
                self._expected[b] = cpu.batch("match", "last", self.lead, self.hosts[b],
                                              stride=self.L, n=self.n, threads=threads)
            er, es, ee = self._expected[b]
            return same(None, er, es, ee), int((er > 0).sum())
        if c == 3:
            data, offsets = self.hosts[0]
            ok, hits = True, 0
            m = min(1 << 16, self.n)
            for lo in (0, self.n - m):
                so = offsets[lo:lo + m + 1]
                sub = data[int(so[0]):int(so[-1])]
                er, es, ee = cpu.batch("match", "last", self.lead, sub, offsets=so - so[0],
                                       threads=threads)
                idx = torch.arange(lo, lo + m, device="cuda")
                ok = ok and same(idx, er, es, ee)
                hits += int((er > 0).sum())
            return ok, hits
        # 2, 4: lines spread over the batch
        m = min(self.n, 1 << (10 if c == 2 else 6))
        idx = torch.arange(0, self.n, self.n // m, device="cuda")[:m]
        sample = self.bufs[b].view(self.n, self.L)[idx].contiguous().cpu().numpy().reshape(-1)
        er, es, ee = cpu.batch("match", "last", self.lead, sample, stride=self.L, n=len(idx),
                               threads=threads)
        return same(idx, er, es, ee), int((er > 0).sum())

    def verify_note(self):
        return {1: "every line of every batch of the last call (per-call mode) / of the last output of "
                   "every stream vs the CPU oracle",
                2: "1024 lines spread over the last output of every stream vs the CPU oracle",
                3: "first and last 65,536 lines of the last output vs the CPU oracle",
                4: "64 inputs spread over the last output of every stream vs the CPU oracle"}[
                    self.args.config]

    # -- the reference's CPU matcher on a bounded sample of the same workload -------------------
    def cpu_sample(self):
        import numpy as np
        c = self.args.config
        if c == 1:
            return dict(data=self.hosts[0], stride=self.L, n=self.n), \
                "the full %d x %d B batch" % (self.n, self.L)
        if c == 3:
            data, offsets = self.hosts[0]
            m = min(self.n, 1 << 21)
            so = offsets[:m + 1]
            return dict(data=data[:int(so[-1])], offsets=so), \
                "the first %d lines (%d bytes) of the batch" % (m, int(so[-1]))
        m = min(self.n, (1 << 15) if c == 2 else (1 << 10))
        idx = self.torch.arange(0, self.n, self.n // m, device="cuda")[:m]
        sample = self.bufs[0].view(self.n, self.L)[idx].contiguous().cpu().numpy().reshape(-1)
        return dict(data=sample, stride=self.L, n=m), \
            "%d lines x %d B spread over buffer 0" % (m, self.L)


def cpu_baseline(args, wl, oracle):
    """The reference's own CPU matcher (oracle/_ref, compiled from the reference sources) if its
    prebuilt .so travelled with the repo, else the C restatement; all host cores of this box's
    share, one contiguous shard per thread as tools/thr_red.cpp:86-91 does.  ~cpu_seconds."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    if oracle.have_ref():
        eng, kind = oracle.Reference(wl.blob), "reference"
    else:
        eng, kind = oracle.CpuOracle(wl.blob), "port"
    kw, what = wl.cpu_sample()
    data = kw.pop("data")
    nbytes = int(kw["offsets"][-1]) if "offsets" in kw else kw["stride"] * kw["n"]
    eng.batch("match", "last", wl.lead, data, threads=cores, **kw)  # warm
    t0 = time.perf_counter()
    passes = 0
    while True:
        eng.batch("match", "last", wl.lead, data, threads=cores, **kw)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds or passes >= 200:
            break
    gbs = passes * nbytes / dt / 1e9
    t1 = time.perf_counter()
    eng.batch("match", "last", wl.lead, data, threads=1, **kw)
    one = nbytes / (time.perf_counter() - t1) / 1e9
    model, phys, logical = cpu_identity()
    out = {"value": round(gbs, 3), "unit": "GB/s", "cores": cores, "kind": kind,
           "sample": "%d passes over %s, match<styLast,%s>, %d threads" %
                     (passes, what, "true" if wl.lead else "false", cores),
           "single_thread_GBps": round(one, 3),
           "cpu_model": model, "physical_cores_of_host": phys, "logical_cpus_of_host": logical,
           "threads_used": cores,
           "speedup_over_one_thread": round(gbs / one, 1) if one else None,
           "scaling_note": ("threads are pinned to nothing and capped at this process's CPU affinity "
                            "(min(affinity, 64)); the reference's matcher is one dependent L1 load per "
                            "byte (Proxy.h:143-147), so SMT siblings share a core's load port and add "
                            "little, each pass spawns and joins its threads (a 64 MiB pass lasts "
                            "~16 ms at this rate), and the box's other tenants share the sockets - "
                            "which is why N threads give less than N x one thread here although the "
                            "matcher itself is lock-free (doc/Performance.md:81-84)")}
    return out


def cpu_identity():
    """CPU model name, physical cores and logical CPUs of the host (from /proc/cpuinfo)."""
    model, phys, logical = "unknown", None, os.cpu_count()
    try:
        cores = set()
        cur_phys = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model == "unknown":
                model = v
            elif k == "physical id":
                cur_phys = v
            elif k == "core id":
                cores.add((cur_phys, v))
        phys = len(cores) or None
    except OSError:
        pass
    return model, phys, logical


# ------------------------------------------------------------------------------------------------
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
    from one_amd import _lib
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
    rehearse = bool(getattr(args, "rehearse_gather", False)) and world == 1
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    elif rehearse:
        import tempfile
        import torch.distributed as dist
        rdv = os.path.join(tempfile.mkdtemp(prefix="bench_rdv_"), "rdv")
        dist.init_process_group("nccl", init_method="file://" + rdv, rank=0, world_size=1,
                                device_id=torch.device("cuda", local_rank))
    multi = world > 1 or rehearse

    if args.config is None:
        args.config = 2 if multi else 1
    if args.steps is None:
        args.steps = {1: 300, 2: 20, 3: 40, 4: 5}[args.config]
    if args.warmup is None:
        args.warmup = {1: 30, 2: 3, 3: 5, 4: 1}[args.config]
    if args.per_call is None:
        args.per_call = min(max(1, args.steps), 32) if (args.config == 1 and not multi) else 1
    if multi or args.streams > 1:
        args.per_call = 1
    per_call = args.per_call

    import oracle  # checker and CPU baseline only - never inside a timed region
    wl = Workload(args, rank, torch, W, one_amd, oracle)
    info, n = wl.info, wl.n

    import ctypes as C
    fn = _lib.lib().redgpu_match_batch_dev
    fn_many = _lib.lib().redgpu_match_batches_dev
    cur_stream = torch.cuda.current_stream().cuda_stream
    streams = [torch.cuda.Stream() for _ in range(args.streams)] if args.streams > 1 else None
    period = len(wl.bufs) * len(wl.outs) * max(1, args.streams)
    calls = [wl.call_tuple(i, streams[i % len(streams)].cuda_stream if streams else cur_stream)
             for i in range(period)]
    # the same steps as redgpu_batch descriptors: step i = descs[i % period]; a call of the
    # multi-batch entry point takes a window of consecutive steps (the array is doubled so that a
    # window never wraps)
    descs = (_lib.BatchDesc * (2 * period))()
    for i in range(2 * period):
        c = calls[i % period]
        descs[i] = _lib.BatchDesc(c[3], c[4], c[5], c[6], c[7], c[8], c[9])
    dsz = C.sizeof(_lib.BatchDesc)
    last_on_stream = {}
    last_call = []

    def step(i):
        rc = fn(*calls[i % period])
        if rc != 0:
            raise RuntimeError(_lib.lib().redgpu_last_error().decode())
        last_on_stream[i % max(1, args.streams)] = (i % len(wl.bufs), i % len(wl.outs))
        return wl.outs[i % len(wl.outs)]

    wins = {}

    def window(i, cnt):
        """descriptors of steps i .. i + cnt - 1 (built once: outside the timed region)"""
        key = (i % period, cnt)
        if key not in wins:
            wins[key] = (_lib.BatchDesc * cnt).from_buffer(descs, key[0] * dsz)
        return wins[key]

    def steps_call(i, cnt):
        """steps i .. i + cnt - 1 in ONE call of redgpu_match_batches_dev on the current stream"""
        rc = fn_many(wl.exe._h, wl.style, wl.lead, window(i, cnt), cnt, cur_stream)
        if rc != 0:
            raise RuntimeError(_lib.lib().redgpu_last_error().decode())
        last_call[:] = [((i + k) % len(wl.bufs), (i + k) % len(wl.outs)) for k in range(cnt)]

    plans = {}
    host_trace = []

    def plan_of(count, per=None):
        """the multi-batch calls of `count` steps: (descriptor window, steps in it), built once"""
        per = per or per_call
        if (count, per) not in plans:
            plans[(count, per)] = [(window(i, min(per, count - i)), min(per, count - i), i)
                                   for i in range(0, count, per)]
        return plans[(count, per)]

    # The warm-up steps go out two per call: the host side of a launch (HIP's launch path, the
    # library's) costs ~3 us when it has just run and 20-50 us when it has not (scripts/lab/
    # host_issue.py: 3.1 us back to back, 7-11 us after 50 ms, 32-48 us after 0.5 s), and W steps
    # in one call would leave the timed call as the second launch of the process.
    warm_per = 2 if per_call > 1 else None

    def run_steps(count, per=None):
        """`count` steps, the way this run issues them; returns the last step's outputs"""
        if per_call > 1:
            h, sty, lead = wl.exe._h, wl.style, wl.lead
            ta = time.perf_counter()
            for win, cnt, i in plan_of(count, per):
                if fn_many(h, sty, lead, win, cnt, cur_stream) != 0:
                    raise RuntimeError(_lib.lib().redgpu_last_error().decode())
            host_trace[:] = [ta, time.perf_counter()]
            if count:
                last_call[:] = [((i + k) % len(wl.bufs), (i + k) % len(wl.outs)) for k in range(cnt)]
            return wl.outs[(count - 1) % len(wl.outs)]
        if per_call <= 1:
            o = None
            for i in range(count):
                o = step(i)
                if gather:
                    gather.push(o)
            return o
        i = 0
        while i < count:
            cnt = min(per_call, count - i)
            steps_call(i, cnt)
            i += cnt
        return wl.outs[(count - 1) % len(wl.outs)]

    # ---- multi-GPU: every step's Outcomes gathered to rank 0 (compact records, pipelined) -------
    gather = None
    if multi:
        from one_amd import sharding
        max_len = wl.L if not wl.ragged else 256
        gather = sharding.StepGather(n, info["max_result"], max_len, wl.want_start, depth=2,
                                     via_host=(backend != "nccl"),
                                     device="cuda" if backend == "nccl" else "cpu")
        # RCCL connects lazily on the first collective: one untimed gather, whatever --warmup is
        gather.push(step(0))
        gather.flush()

    # rank 0 alone, untimed: the same per-GPU workload on ONE GPU, so that the weak-scaling
    # reference of a multi-GPU line is this workload and not another config
    solo = None
    if multi:
        if rank == 0:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = max(3, min(args.steps, 10))
            for i in range(k):
                step(i)
            torch.cuda.synchronize()
            solo = k * wl.in_bytes / (time.perf_counter() - t0) / 1e9
        dist.barrier()

    if per_call > 1:  # the timed loop's descriptor windows, before any clock starts
        plan_of(args.steps)
        plan_of(args.warmup, warm_per)
    # (the two events exist - torch creates the HIP event at its first record() - before any clock
    # starts: creating them inside the timed region cost ~50 us of a 0.4 ms region)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    run_steps(args.warmup, warm_per)
    ev1.record()
    if streams is not None:
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    if gather:
        gather.flush()
    torch.cuda.synchronize()
    ev1.query()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    # the synchronize above may have put this core to sleep: a core that wakes into the timed
    # region issued the one launch of configs[1] in 113 us instead of 25 (one run in fifteen).
    # 300 us of spinning on the clock - no GPU work, nothing of a step - and it is awake.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 300e-6:
        pass

    ev0.record()           # (asynchronous: the idle GPU stamps it at once; its host cost is not a step's)
    t0 = time.perf_counter()
    run_steps(args.steps)
    if streams is not None:
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    ev1.record()
    t_issued = time.perf_counter()
    if gather:
        gather.flush()  # the gathers still in flight: inside the timed region
    else:
        # poll the closing event instead of sleeping in the synchronize below: a blocking wait
        # returns tens of microseconds after the GPU is done, which on a 0.35 ms region is not noise
        while not ev1.query():
            pass
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    region_ms = ev0.elapsed_time(ev1)
    kernel_name = one_amd.last_kernel()

    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- bit-exact gate: the last output each stream produced inside the timed loop -----------
    bit_exact, matches = True, 0
    checked = []
    for sidx, (b, o) in sorted(last_on_stream.items()):
        ok, hits = wl.verify(b, o)
        bit_exact = bit_exact and ok
        matches += hits
        checked.append({"stream": sidx, "buffer": b, "ok": ok})
    for k, (b, o) in enumerate(last_call):  # every batch of the last multi-batch call
        ok, hits = wl.verify(b, o)
        bit_exact = bit_exact and ok
        matches += hits
        checked.append({"batch_of_last_call": k, "buffer": b, "ok": ok})
    if dist:
        t = torch.tensor([1 if bit_exact else 0], dtype=torch.int64,
                         device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        bit_exact = bool(t.item())

    # ---- what the root gathered for the LAST step = every rank's own last output ---------------
    # (a checksum of checksums: each rank sums its result / start / end arrays, the root sums the
    # matching slices of the gathered arrays; its own slice is also compared element for element)
    gathered_ok = None
    if gather is not None and args.steps > 0:
        cdev = "cuda" if backend == "nccl" else "cpu"
        lr, ls, le = wl.outs[(args.steps - 1) % len(wl.outs)]
        mine = torch.stack([lr.to(torch.int64).sum(), le.sum(),
                            ls.sum() if ls is not None else torch.zeros((), dtype=torch.int64,
                                                                        device=lr.device)]).to(cdev)
        sums = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(sums, mine)
        if rank == 0:
            gr, gs, ge = gather.last
            gathered_ok = gr.numel() == world * n
            for k in range(world if gathered_ok else 0):
                sl = slice(k * n, (k + 1) * n)
                got = [int(gr[sl].to(torch.int64).sum()), int(ge[sl].sum()),
                       int(gs[sl].sum()) if gs is not None else 0]
                gathered_ok = gathered_ok and got == [int(x) for x in sums[k].tolist()]
            gathered_ok = gathered_ok and bool((gr[:n].to(lr.device) == lr).all()) and \
                bool((ge[:n].to(le.device) == le).all())
            bit_exact = bit_exact and gathered_ok

    # ---- dominant kernel's average launch duration: HIP events on its launch stream --------------
    # One stream, launches back to back, one event before the first and one after the last:
    # elapsed / launches = the kernel's duration plus the ~1.5 us gap between dependent launches -
    # what `rocprofv3 --kernel-trace --stats` reports for the same command.
    torch.cuda.synchronize()
    nk = max(5, min(args.steps, 200))
    single = [c[:-1] + (cur_stream,) for c in calls]
    for i in range(min(10, nk)):
        fn(*single[i % period])
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0.record()
    for i in range(nk):
        fn(*single[i % period])
    k1.record()
    torch.cuda.synchronize()
    kernel_ms_single = k0.elapsed_time(k1) / nk
    single_kernel_name = one_amd.last_kernel()
    kernel_ms = kernel_ms_single
    launches_timed = nk
    if per_call > 1:
        # the timed path's own launch: per_call batches per launch of k_stream_multi
        cnt = min(per_call, max(1, args.steps))
        reps = max(3, min(20, 200 // cnt))
        for r in range(2):
            steps_call(r * cnt, cnt)
        k0.record()
        for r in range(reps):
            steps_call(r * cnt, cnt)
        k1.record()
        torch.cuda.synchronize()
        kernel_ms = k0.elapsed_time(k1) / reps
        launches_timed = reps
        kernel_name = one_amd.last_kernel()
    batches_per_launch = min(per_call, max(1, args.steps)) if per_call > 1 else 1

    # ---- calibrations, same session -------------------------------------------------
    calib = {}
    walked = None
    if rank == 0 and not args.no_calibration:
        l = _lib.lib()
        sink = torch.zeros(4, dtype=torch.int32, device="cuda")
        # (a) streaming-read ceiling: one launch over >= 2 GiB (a 64 MiB launch is mostly ramp)
        scratch = wl.bufs[0] if wl.bufs[0].numel() >= (1 << 31) else \
            torch.empty(1 << 31, dtype=torch.uint8, device="cuda").random_(0, 256)
        nb = scratch.numel() & ~15
        for _ in range(2):
            l.redgpu_diag_read_dev(wl.exe._h, scratch.data_ptr(), nb, sink.data_ptr(), cur_stream)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(5):
            l.redgpu_diag_read_dev(wl.exe._h, scratch.data_ptr(), nb, sink.data_ptr(), cur_stream)
        c1.record()
        torch.cuda.synchronize()
        calib["read_ceiling_GBps"] = round(5 * nb / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
        calib["read_ceiling_how"] = "k_diag_read: 5 launches, each one streaming pass over %d bytes" % nb
        if scratch is not wl.bufs[0]:
            del scratch
        # (b) the same read at the batch's own size (what one launch of this workload can see)
        if wl.in_bytes < (1 << 31) and not wl.ragged:
            for i in range(3):
                l.redgpu_diag_read_dev(wl.exe._h, wl.bufs[i % len(wl.bufs)].data_ptr(),
                                       wl.in_bytes, sink.data_ptr(), cur_stream)
            c0.record()
            for i in range(30):
                l.redgpu_diag_read_dev(wl.exe._h, wl.bufs[i % len(wl.bufs)].data_ptr(),
                                       wl.in_bytes, sink.data_ptr(), cur_stream)
            c1.record()
            torch.cuda.synchronize()
            calib["read_at_batch_size_GBps"] = round(30 * wl.in_bytes / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
        # (c) LDS gather roof of a one-lookup-per-byte walk: the walk without its memory side
        import ctypes as C
        lookups = C.c_uint64(0)
        for _ in range(2):
            l.redgpu_diag_lds_dev(wl.exe._h, 256, sink.data_ptr(), C.byref(lookups), cur_stream)
        c0.record()
        for _ in range(5):
            l.redgpu_diag_lds_dev(wl.exe._h, 256, sink.data_ptr(), C.byref(lookups), cur_stream)
        c1.record()
        torch.cuda.synchronize()
        calib["lds_gather_roof_GBps"] = round(5 * lookups.value / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
        calib["lds_gather_roof_how"] = ("k_diag_lds: %d dependent ds_read_u8 lookups of a 64 KB table per "
                                        "launch on uniformly random bytes, 4 chains/lane x 512 lanes/CU, "
                                        "no input traffic; 1 lookup = 1 input byte" % lookups.value)
        # (c2) the same kernel on 16 x the batch in ONE launch: what it sustains once a launch's head
        # (table staging, first blocks arriving) and tail (the last lines' dependent steps) are
        # amortised - the gap to `achieved` is what a 64 MiB launch loses to them
        if args.config == 1 and wl.in_bytes <= (1 << 27):
            big_n = n * 16
            big = torch.empty(big_n * wl.L, dtype=torch.uint8, device="cuda").random_(0, 256)
            br = torch.empty(big_n, dtype=torch.int32, device="cuda")
            bs = torch.empty(big_n, dtype=torch.int64, device="cuda") if wl.want_start else None
            be = torch.empty(big_n, dtype=torch.int64, device="cuda")
            targs = (wl.exe._h, wl.style, wl.lead, big.data_ptr(), None, wl.L, big_n, br.data_ptr(),
                     bs.data_ptr() if bs is not None else None, be.data_ptr(), cur_stream)
            for _ in range(2):
                fn(*targs)
            c0.record()
            for _ in range(5):
                fn(*targs)
            c1.record()
            torch.cuda.synchronize()
            calib["kernel_at_16x_batch_GBps"] = round(5 * big_n * wl.L / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
            calib["kernel_at_16x_batch_how"] = ("the same entry point, %d lines x %d B of random bytes in one "
                                                "launch (not the workload: shows the launch's head + tail)" %
                                                (big_n, wl.L))
            del big, br, bs, be
        # (c2b) the memory side of this shape alone: 64 B read + 20 B written per line, requested and
        # stored the way the walk does, no table, no steps - the HBM roof of configs[1]'s shape
        if args.config == 1 and wl.L == 64 and wl.want_start:
            big_n = n * 16
            big = torch.empty(big_n * wl.L, dtype=torch.uint8, device="cuda").random_(0, 256)
            br = torch.empty(big_n, dtype=torch.int32, device="cuda")
            bs = torch.empty(big_n, dtype=torch.int64, device="cuda")
            be = torch.empty(big_n, dtype=torch.int64, device="cuda")
            margs = (wl.exe._h, big.data_ptr(), big_n, 64, br.data_ptr(), bs.data_ptr(), be.data_ptr(),
                     sink.data_ptr(), cur_stream)
            for _ in range(2):
                l.redgpu_diag_lines_dev(*margs)
            c0.record()
            for _ in range(5):
                l.redgpu_diag_lines_dev(*margs)
            c1.record()
            torch.cuda.synchronize()
            calib["lines64_memory_roof_GBps"] = round(5 * big_n * wl.L / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
            calib["lines64_memory_roof_how"] = ("k_diag_lines: %d lines x 64 B read as the walk requests them + "
                                                "an int32 and two uint64 stored per line, nothing else; GB/s of "
                                                "INPUT (x 84/64 = HBM traffic)" % big_n)
            del big, br, bs, be
        # (c2c) long lines: what HBM gives the request pattern itself - one cache line per lane and
        # request, the lanes of a wave a whole line length apart - reads only
        if args.config in (2, 4) and not wl.ragged and wl.L % 128 == 0:
            margs = (wl.exe._h, wl.bufs[0].data_ptr(), n, wl.L, None, None, None, sink.data_ptr(), cur_stream)
            for _ in range(2):
                l.redgpu_diag_lines_dev(*margs)
            c0.record()
            for _ in range(3):
                l.redgpu_diag_lines_dev(*margs)
            c1.record()
            torch.cuda.synchronize()
            calib["long_line_pattern_roof_GBps"] = round(3 * (n // 1024 * 1024) * wl.L / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
            calib["long_line_pattern_roof_how"] = ("k_diag_long: the batch's own bytes requested as the walk "
                                                   "requests them (128 B per lane and request, 2 lines per lane, "
                                                   "lanes %d B apart), no table, no steps, no stores" % wl.L)
        # (c3) the north star's literal outputs - result code and end offset, no start: the same
        # batch through the same entry point with start = NULL (12 B written per line, not 20)
        if wl.want_start and not wl.ragged and args.config in (1, 2):
            t = list(wl.call_tuple(0, cur_stream))
            t[8] = None
            reps = 20 if args.config == 1 else 5
            for _ in range(2):
                fn(*t)
            c0.record()
            for _ in range(reps):
                fn(*t)
            c1.record()
            torch.cuda.synchronize()
            calib["result_end_only_GBps"] = round(reps * wl.in_bytes / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
            calib["result_end_only_how"] = ("the same batch and entry point with start = NULL (result + end, "
                                            "the outputs north_star names), %d launches back to back on one "
                                            "stream; `value` keeps the full Outcome" % reps)
        # (c5) the launch-bound shape under a HIP graph: 20 launches of the timed call captured once
        # on one stream (the _dev entry points are stream-ordered, nothing in them synchronises or
        # allocates on this path) and replayed - what the ~1.5 us gap between dependent launches costs
        if args.config == 1 and not wl.ragged:
            try:
                gs = torch.cuda.Stream()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.stream(gs):
                    for i in range(3):
                        fn(*wl.call_tuple(i, gs.cuda_stream))
                gs.synchronize()
                with torch.cuda.graph(graph, stream=gs):
                    for i in range(20):
                        rc = fn(*wl.call_tuple(i, torch.cuda.current_stream().cuda_stream))
                        if rc != 0:
                            raise RuntimeError(l.redgpu_last_error().decode())
                graph.replay()
                torch.cuda.synchronize()
                c0.record()
                for _ in range(5):
                    graph.replay()
                c1.record()
                torch.cuda.synchronize()
                calib["graph_replay_GBps"] = round(100 * wl.in_bytes / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
                calib["graph_replay_how"] = ("20 launches captured in one HIP graph on one stream, replayed "
                                             "5 times (same entry point, rotating buffers)")
                del graph
            except Exception as ex:  # capture is evidence, not the product path
                calib["graph_replay_GBps"] = None
                calib["graph_replay_error"] = str(ex)[:200]
        # (c4) the path's other verb on the same batch: scan<styInstant,false> ("does the line contain
        # a match") - for a loose-start DFA (redgpu_info.suffix_closed) it runs on the same kernels
        if not wl.ragged and args.config in (1, 2) and info.get("suffix_closed"):
            sres = torch.empty(n, dtype=torch.int32, device="cuda")
            sargs = (wl.exe._h, int(one_amd.styInstant), 0, wl.bufs[0].data_ptr(), None, wl.L, n,
                     sres.data_ptr(), cur_stream)
            reps = 20 if args.config == 1 else 5
            for _ in range(2):
                l.redgpu_scan_batch_dev(*sargs)
            c0.record()
            for _ in range(reps):
                l.redgpu_scan_batch_dev(*sargs)
            c1.record()
            torch.cuda.synchronize()
            calib["scan_instant_GBps"] = round(reps * wl.in_bytes / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
            calib["scan_instant_kernel"] = one_amd.last_kernel()
            # scan<styInstant> is non-zero exactly where match<styLast> is (single-result DFA)
            l.redgpu_match_batch_dev(*wl.call_tuple(0, cur_stream))
            torch.cuda.synchronize()
            calib["scan_instant_agrees_with_match"] = bool(((sres != 0) == (wl.outs[0][0] != 0)).all())
        # (c6) the L2 gather roof, measured: dependent 2-byte gathers over a 2 MiB table in L2, no
        # input side (replaces the literal derived from L2's nominal 34.5 TB/s)
        if info["table_kind"] in (4, 5):
            ltab = torch.randint(0, 65536, (1 << 20,), device="cuda", dtype=torch.int32).to(torch.int16)
            for _ in range(2):
                l.redgpu_diag_l2_dev(wl.exe._h, ltab.data_ptr(), 4096, sink.data_ptr(), C.byref(lookups), cur_stream)
            c0.record()
            for _ in range(3):
                l.redgpu_diag_l2_dev(wl.exe._h, ltab.data_ptr(), 4096, sink.data_ptr(), C.byref(lookups), cur_stream)
            c1.record()
            torch.cuda.synchronize()
            calib["l2_gather_roof_Glookups"] = round(3 * lookups.value / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
            calib["l2_gather_roof_how"] = ("k_diag_l2: %d dependent 2-byte gathers per launch over a 2 MiB table "
                                           "resident in L2, 2 chains/lane x 2048 lanes/CU, no input traffic; "
                                           "1 gather = 1 input byte of a table-in-L2 walk" % lookups.value)
        # (d) bytes the walk actually reads (early-exit DFAs)
        if info["early_death"] or wl.ragged:
            w = torch.zeros(1, dtype=torch.int64, device="cuda")
            l.redgpu_diag_walked_dev(wl.exe._h, wl.lead, wl.bufs[0].data_ptr(),
                                     wl.offsets_dev.data_ptr() if wl.ragged else None,
                                     0 if wl.ragged else wl.L, n, w.data_ptr(), cur_stream)
            torch.cuda.synchronize()
            walked = int(w.item())

    # ---- the line ---------------------------------------------------------------------------------
    value = world * args.steps * wl.in_bytes / elapsed / 1e9
    l2_bound = info["table_kind"] in (4, 5)  # REDGPU_TAB_GLOBAL_*: one L2 gather per byte
    algo_bytes = (walked if walked is not None else wl.in_bytes) * batches_per_launch
    # kernel_ms = the measured back-to-back loop of the timed path's own launches (it counts the
    # ~1.5 us gap between dependent launches with every kernel); a HIP-graph replay of the
    # single-batch launches is reported beside it (graph_replay_GBps), never folded in
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    for tag in ("r03", "r02"):
        pmc_path = os.path.join(ROOT, "profiles", "%s_pmc_config%d_%s.json" % (tag, args.config, wl.dfa_name))
        if world == 1 and os.path.exists(pmc_path):
            pj = json.load(open(pmc_path))
            if pj.get("batches_per_launch", 1) != batches_per_launch:
                continue
            traffic = pj.get("hbm_traffic_bytes_per_launch")
            traffic_src = ("%s: separate rocprofv3 --pmc passes of this command (FETCH_SIZE doubled "
                           "per the gfx950 note + WRITE_SIZE), NOT measured by this run" %
                           os.path.relpath(pmc_path, ROOT))
            break
    # the L2-bound config is priced against the gather roof measured in this run (k_diag_l2); the
    # figure derived from the guide's nominal L2 bandwidth is kept beside it
    l2_peak = calib.get("l2_gather_roof_Glookups") or L2_PEAK_GLOOKUPS
    peak = l2_peak if l2_bound else HBM_PEAK_GBS
    roofline = {
        "bound": "l2-gather" if l2_bound else "hbm",
        "achieved": round(achieved, 1),
        "peak": peak,
        "unit": "Glookups/s" if l2_bound else "GB/s",
        "frac": round(achieved / peak, 4),
        "traffic": traffic,
        "kernel": kernel_name,
        "batches_per_launch": batches_per_launch,
        "algorithmic_bytes_per_launch": algo_bytes,
        "input_bytes_per_launch": wl.in_bytes * batches_per_launch,
        "output_bytes_per_launch": (wl.out_bytes - (8 * n if wl.ragged else 0)) * batches_per_launch,
        "kernel_ms": round(kernel_ms, 5),
        "kernel_ms_how": "HIP events around %d back-to-back launches of %s on one stream (includes "
                         "the ~1.5 us dependent-launch gap)" % (launches_timed, kernel_name),
        "timed_region_ms_per_step": round(region_ms / args.steps, 5),
        "timed_region_host_issue_ms": round((t_issued - t0) * 1e3, 5),
        "timed_region_library_calls_ms": (round((host_trace[1] - host_trace[0]) * 1e3, 5)
                                          if len(host_trace) == 2 else None),
    }
    # the timed region by its own HIP events (recorded on the launch stream in front of the first
    # and behind the last of the K steps): the same bytes over that time.  `achieved` / `frac`
    # above stay on kernel_ms - many launches back to back, later in the run, at sustained clocks.
    if region_ms > 0:
        a_tr = (walked if walked is not None else wl.in_bytes) * args.steps / (region_ms * 1e-3) / 1e9
        roofline["timed_region"] = {"ms": round(region_ms, 5), "achieved": round(a_tr, 1),
                                    "frac": round(a_tr / peak, 4),
                                    "how": "HIP events around the %d timed steps" % args.steps}
    if batches_per_launch > 1:
        # the same workload one batch per launch (redgpu_match_batch_dev per step): what a caller
        # that cannot hand over several batches at once gets
        a1 = (walked if walked is not None else wl.in_bytes) / (kernel_ms_single * 1e-3) / 1e9
        roofline["single_batch_launch"] = {
            "kernel": single_kernel_name, "kernel_ms": round(kernel_ms_single, 5),
            "achieved": round(a1, 1), "frac": round(a1 / peak, 4),
            "how": "HIP events around %d back-to-back single-batch launches on one stream" % nk}
    # "also report total-traffic GB/s = sum(L + out [+ 8]) / t" - what the launch must
    # move at the least: the lines, their offsets (ragged) and the Outcome fields it writes
    roofline["total_traffic_GBps"] = round((wl.in_bytes + wl.out_bytes) * batches_per_launch /
                                           (kernel_ms * 1e-3) / 1e9, 1)
    roofline["total_traffic_frac"] = round(roofline["total_traffic_GBps"] / HBM_PEAK_GBS, 4)
    if traffic_src:
        roofline["traffic_source"] = traffic_src
    if l2_bound:
        roofline["hbm_frac"] = round(wl.in_bytes * batches_per_launch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roofline["peak_source"] = ("l2_gather_roof_Glookups, measured in this run" if
                                   calib.get("l2_gather_roof_Glookups") else "nominal")
        roofline["l2_nominal_request_roof_Glookups"] = L2_PEAK_GLOOKUPS
        roofline["bound_note"] = ("one dependent gather of the 2 MiB class table per input byte; it "
                                  "lives in L2 (4 MiB per XCD): the walk cannot issue more gathers "
                                  "than L2 serves (nominally 34.5 TB/s / 128 B = 269.5 G requests/s)")
    if walked is not None:
        roofline["bytes_walked_per_launch"] = walked
        roofline["walked_frac_of_input"] = round(walked / max(1, wl.in_bytes), 4)
        roofline["scanned_GBps"] = round(wl.in_bytes * batches_per_launch / (kernel_ms * 1e-3) / 1e9, 1)
        roofline["algorithmic_note"] = ("achieved counts the bytes the reference's loop consumes "
                                        "(k_walked); scanned_GBps counts whole lines")
    roofline.update(calib)
    shape = ("%d ragged lines of 32..256 B (%.2f GiB)" % (n, wl.in_bytes / 2**30) if wl.ragged else
             "%d lines x %d B" % (n, wl.L))
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
            "workload": "%s: %s DFA (%d states used / %d classes, REDA fmtDirect%d), %s per GPU, "
                        "match<styLast,%s> -> %s" %
                        (wl.label, wl.dfa_name, info["states_used"], info["n_classes"],
                         info["format"], shape, "true" if wl.lead else "false",
                         "result+start+end" if wl.want_start else "result+end"),
            "input": "text (47-character alphabet, planted matches)" if wl.text else "uniformly random bytes",
            "lines_per_gpu": n, "line_len": wl.L if not wl.ragged else "32..256",
            "rotating_input_buffers": len(wl.bufs),
            "hot_rows": wl.tuned or ("static choice" if info["table_kind"] == 6 else None),
            "working_set_bytes": len(wl.bufs) * wl.in_bytes,
            "streams": args.streams,
            "steps_per_call": per_call,
            "value_is": ("steps issued round-robin on %d HIP streams: up to %d independent batches in "
                         "flight" % (args.streams, args.streams)) if args.streams > 1 else
                        ("one stream; the steps handed to redgpu_match_batches_dev %d at a time (one "
                         "launch per call: table staged once, tiles handed out across the batches)"
                         % per_call) if per_call > 1 else
                        "steps back to back on one stream, one redgpu_match_batch_dev call each",
            "sharding": "contiguous shard per GPU, no data-path collective" +
                        ("; EVERY step's Outcomes gathered to rank 0 over %s as compact records "
                         "(pipelined, 2 in flight)" % ("RCCL" if backend == "nccl" else "gloo")
                         if multi else "") +
                        (" - REHEARSAL on one GPU: a one-rank RCCL group, so the pack / collective "
                         "call / widen-on-root cost of a step is in the number but no bytes cross "
                         "xGMI" if rehearse else ""),
        },
        "minputs_per_s": round(world * args.steps * n / elapsed / 1e6, 1),
        "bit_exact": bit_exact,
        "bit_exact_how": wl.verify_note(),
        "checked": checked,
        "matches_in_checked": matches,
        "kernel": kernel_name,
        "roofline": roofline,
    }
    if solo is not None:
        line["single_gpu_same_workload_GBps"] = round(solo, 1)
        line["single_gpu_note"] = ("rank 0 alone, untimed phase, same per-GPU workload, no gather: the "
                                   "weak-scaling reference for this line")
    if gather is not None:
        line["gathered_steps"] = gather.finished
        line["gathered_matches_ranks"] = gathered_ok
        line["gathered_check"] = ("last step: per-rank sums of result / start / end vs the sums of the "
                                  "matching slices on the root; the root's own slice element for element")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args, wl, oracle)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()
    if not bit_exact:
        raise SystemExit("bench: GPU output differs from the oracle - number is invalid")


if __name__ == "__main__":
    main()
