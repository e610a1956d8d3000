#!/usr/bin/env python3
"""Headline benchmark: 6-stage biquad (DF1) cascade, 1 M streams x 4096-sample blocks per GPU.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  One "step" = one fz_run_block launch = one
4096-sample block of every stream of this rank (state carried from step to step), input frames
already resident in HBM.  Rank 0 prints ONE JSON line.

  value      whole-job Msamples/s = streams(all ranks) * 4096 * K / max-over-ranks wall time
  roofline   dominant kernel fz_block_kernel: algorithmic bytes per launch / its average launch
             duration = HIP-event time over the K back-to-back launches of the timed region / K,
             events recorded on the launch stream (torch's current stream, which run_block uses);
             peak = 8000 GB/s (HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md);
             traffic = PMC bytes of THIS kernel symbol on THIS workload (profiles/pmc_traffic.json,
             collected by tools/profile_round.sh), null when that symbol was not profiled;
             sustained = the same launches back to back for >= 2 s (power-managed clocks settle), with
             sustained.board = package power / power cap / shader clock sampled through rocm-smi meanwhile
             (this workload runs at the board's power cap: profiles/r03/power_and_clocks.txt)
  cpu_baseline  the compiled scalar oracle (one closure per stream, one call per sample: what
             the reference's compile()-callable does) timed on this box's host cores on a
             bounded sample of the same workload, rank 0, N == 1 only (buffers pre-touched, threads
             pinned); the oracle also checks the GPU output of >= 1024 random streams bit for bit.
  config2/3/4   the other single-GPU BASELINE configs at full size in the same run (rank 0, N == 1):
             library-default and tuned plan, roofline fraction, oracle parity on >= 1024 random streams.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# every run measures for itself: plans persisted by an earlier process on this board (plans.txt of the kernel cache) would
# turn "library_default" into "whatever was tuned last time"
os.environ.setdefault("FLOWZ_HIP_NO_PLAN_CACHE", "1")

HBM_PEAK_GBS = 8000.0
PARITY_STREAMS = 1024


# ---- CPU baseline --------------------------------------------------------------------------------------------------
def cpu_topology():
    """(logical cpus this process may use, one logical cpu per physical core among them, cgroup CPU quota in cores or None)."""
    avail = sorted(os.sched_getaffinity(0))
    core_of = {}
    try:
        cpu = phys = core = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k = k.strip()
            if k == "processor":
                cpu, phys, core = int(v), 0, None
            elif k == "physical id":
                phys = int(v)
            elif k == "core id":
                core = int(v)
                core_of[cpu] = (phys, core)
    except OSError:
        pass
    seen, one_per_core = set(), []
    for c in avail:
        key = core_of.get(c, ("?", c))
        if key not in seen:
            seen.add(key)
            one_per_core.append(c)
    # a container may be allowed fewer CPU-seconds per second than it sees CPUs (cgroup v2 cpu.max / v1 cfs quota): more
    # runnable threads than that only get throttled
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return avail, one_per_core, quota


def _cpu_threads_run(coracle, coefs, cpus, per_thread, n_samples, seed, reps=1, soa=False):
    """One pinned thread per entry of `cpus`; every thread generates and first-touches its own buffers, then all
    start together.  Returns (wall seconds of the timed part, outputs of thread 0)."""
    import threading

    import numpy as np

    n = len(cpus)
    gate = threading.Barrier(n)
    t_begin, t_end, outs, errs = [0.0] * n, [0.0] * n, [None] * n, []

    def work(i):
        try:
            try:
                os.sched_setaffinity(0, {cpus[i]})           # pid 0 = the calling thread
            except OSError:
                pass
            if soa:                                          # "Mode B": time-major SoA frames [T, streams]
                x = coracle.synth_fill(seed, i * per_thread, per_thread, n_samples)
            else:                                            # one contiguous buffer per stream (the CPU-friendly layout)
                x = coracle.synth_fill(seed, i * per_thread, per_thread, n_samples, stream_major=True)
            y = np.empty_like(x)
            y.fill(0.0)                                      # pages mapped by the thread that will write them
            gate.wait()
            t_begin[i] = time.perf_counter()
            for _ in range(reps):
                if soa:
                    coracle.df1_cascade_soa(coefs, x, out=y)
                else:
                    coracle.df1_cascade(coefs, x, stream_major=True, out=y)
            t_end[i] = time.perf_counter()
            outs[i] = y if i == 0 else None
        except Exception as e:                               # pragma: no cover
            errs.append(e)
            try:
                gate.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    return max(t_end) - min(t_begin), outs[0]


def cpu_baseline(n_samples, seed, coefs):
    """Time the compiled oracle on the host cores (SURVEY 8d Mode A, plus Mode B next to it)."""
    import numpy as np

    from oracle import coracle

    logical, physical, quota = cpu_topology()
    # single-thread calibration (pre-touched buffers): sizes the sample and is the per-thread yardstick
    x0 = coracle.synth_fill(seed, 0, 64, n_samples, stream_major=True)
    y0 = np.zeros_like(x0)
    y0.fill(0.0)
    coracle.df1_cascade(coefs, x0, stream_major=True, out=y0)
    t0 = time.perf_counter()
    coracle.df1_cascade(coefs, x0, stream_major=True, out=y0)
    rate1 = 64 * n_samples / (time.perf_counter() - t0)
    per_thread = 1024                                        # streams per thread (32 MiB of frames in + out), passed `reps` times
    reps = max(1, int(round(2.5 * rate1 / (per_thread * n_samples))))               # ~2.5 s per thread
    # How many threads can this box actually RUN?  Containers often see every CPU of the host but are granted far fewer
    # CPU-seconds per second (cgroup quota, or an over-committed VM): runnable threads beyond that are only throttled
    # and depress the per-thread figure.  Read the quota where it is published, and probe the scaling in any case:
    # short runs with 1, 2, 4 ... threads (one per physical core) until the aggregate rate stops growing.
    probe_reps = max(1, int(round(0.3 * rate1 / (per_thread * n_samples))))
    scaling, n_eff, best_rate, n = {}, 1, 0.0, 1
    cap = len(physical) if quota is None else max(1, min(len(physical), int(quota)))
    while n <= cap:
        wall, _ = _cpu_threads_run(coracle, coefs, physical[:n], per_thread, n_samples, seed, reps=probe_reps)
        rate = n * per_thread * n_samples * probe_reps / wall
        scaling[n] = round(rate / 1e6, 1)
        if rate < best_rate * 1.15:
            break
        n_eff, best_rate = n, rate
        if n == cap:
            break
        n = min(cap, n * 2)
    plans = [(f"{n_eff}_pinned_threads_on_distinct_physical_cores", physical[:n_eff])]
    if n_eff < len(physical):
        plans.append(("one_thread_per_physical_core", physical))
    elif len(logical) > len(physical):
        plans.append(("one_thread_per_logical_cpu", logical))
    runs = {}
    for label, cpus in plans:
        r = max(1, int(reps * min(1.0, n_eff / len(cpus))))                          # bounded: ~2.5 s wall either way
        wall, _ = _cpu_threads_run(coracle, coefs, cpus, per_thread, n_samples, seed, reps=r)
        tot = len(cpus) * per_thread * n_samples * r
        runs[label] = {"threads": len(cpus), "Msamples_per_s": round(tot / wall / 1e6, 1),
                       "Msamples_per_s_per_thread": round(tot / wall / 1e6 / len(cpus), 3), "wall_s": round(wall, 2),
                       "streams": len(cpus) * per_thread, "passes": r}
    best = max(runs, key=lambda k: runs[k]["Msamples_per_s"])
    b = runs[best]
    # `cores` = the cores the reported run could actually OCCUPY: its threads, capped by the CPU-seconds per second the cgroup
    # grants (256 logical CPUs under a quota of 16 are 16 cores' worth of work however many threads are runnable); the per-core
    # rate is the aggregate over those, so that it stays comparable with the single-thread calibration
    cores = b["threads"] if quota is None else max(1, min(b["threads"], int(round(quota))))
    base = {"value": b["Msamples_per_s"], "unit": "Msamples/s", "cores": cores, "kind": "port",
            "Msamples_per_s_per_core": round(b["Msamples_per_s"] / cores, 3), "threads": b["threads"],
            "single_thread_calibration_Msamples_per_s": round(rate1 / 1e6, 3),
            "physical_cores": len(physical), "logical_cpus": len(logical), "cgroup_cpu_quota": quota, "thread_scaling_probe_Msamples_per_s": scaling,
            "threads_pinned": True, "runs": runs,
            "sample": f"{b['streams']} streams x {n_samples} samples x {b['passes']} passes ({best}: {b['threads']} pinned threads on {cores} cores' worth of CPU time), 6-stage DF1 cascade, "
                      f"scalar closure per stream, one call per sample (oracle/flowz_oracle.c, gcc -O3 -ffp-contract=off), "
                      f"buffers allocated and first-touched by their thread before the timed region, {b['wall_s']:.2f} s wall"}
    # "Mode B" (SURVEY 8d): the same closures vectorised ACROSS streams by the compiler (SoA state, avx2/avx512
    # clones) -- a CPU stronger than the reference's scalar closure, reported next to it
    try:
        vec_streams, reps_v = 1024, 12                         # per thread: 16 MiB of frames
        xv = coracle.synth_fill(seed, 0, 256, n_samples)
        ok_vec = bool(np.array_equal(coracle.df1_cascade_soa(coefs, xv).view(np.uint32), coracle.df1_cascade(coefs, xv).view(np.uint32)))
        cpus = dict(plans)[best]
        reps = max(1, int(reps_v * min(1.0, n_eff / len(cpus))))
        wall_v, _ = _cpu_threads_run(coracle, coefs, cpus, vec_streams, n_samples, seed, reps=reps, soa=True)
        base["vectorised_across_streams"] = {
            "value": round(len(cpus) * vec_streams * reps * n_samples / wall_v / 1e6, 1), "unit": "Msamples/s",
            "cores": len(cpus) if quota is None else max(1, min(len(cpus), int(round(quota)))), "threads": len(cpus),
            "bitwise_equal_to_scalar": ok_vec,
            "note": "same arithmetic per stream, SoA state, compiler-vectorised (stronger than the reference's scalar closure)"}
    except Exception as e:                                  # never let the extra figure break the bench line
        base["vectorised_across_streams"] = {"error": str(e)[:200]}
    return base


# ---- GPU side helpers ----------------------------------------------------------------------------------------------
def frames(torch, dev, ns, T, w, tile):
    return torch.empty((ns // tile, T, tile, w) if tile else (T, ns, w), dtype=torch.float32, device=dev)


def pick_tile(ns, tile):
    return tile if tile and ns % tile == 0 and tile < ns else 0


def event_ms(torch, fn, reps):
    """HIP events on the launch stream around `reps` back-to-back launches -> ms per launch."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


class PowerSampler:
    """Board power and shader clock (rocm-smi) sampled in a thread while the sustained leg runs: the 6-biquad cascade at 1 M streams
    sits at the package power cap and the firmware lowers the clock until it fits (profiles/r03/power_and_clocks.txt), so a run's
    number is also a statement about the board.  Best effort: None when rocm-smi is missing or prints something else."""

    def __init__(self, device_index=0, period=0.3):
        import shutil
        import threading
        self.exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        self.dev, self.period, self.rows, self._stop = device_index, period, [], False
        self.thread = threading.Thread(target=self._run, daemon=True) if self.exe else None

    def _run(self):
        import re
        import subprocess
        while not self._stop:
            try:
                o = subprocess.run([self.exe, "-d", str(self.dev), "--showpower", "--showclocks", "--showmaxpower", "--json"],
                                   capture_output=True, text=True, timeout=5).stdout
                c = next(iter(json.loads(o).values()))
                watts = [float(v) for k, v in c.items() if "Power (W)" in k and "Max" not in k]
                cap = [float(v) for k, v in c.items() if "Max" in k and "Power (W)" in k]
                sclk = [int(re.search(r"(\d+)", v).group(1)) for k, v in c.items() if k.startswith("sclk clock speed")]
                if watts and sclk:
                    self.rows.append((time.perf_counter(), watts[0], cap[0] if cap else None, sclk[0]))
            except Exception:  # noqa: BLE001 -- a sampler must never take the bench down
                pass
            time.sleep(self.period)

    def start(self):
        if self.thread:
            self.thread.start()
        self.t0 = time.perf_counter()

    def stop(self, settle=0.7):
        self._stop = True
        if self.thread:
            self.thread.join(timeout=6)
        rows = [r for r in self.rows if r[0] - self.t0 >= settle] or self.rows
        if not rows:
            return None
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        watts, cap = med([r[1] for r in rows]), rows[0][2]
        return {"package_W": watts, "cap_W": cap, "at_power_cap": bool(cap and watts >= 0.98 * cap), "sclk_MHz": med([r[3] for r in rows]),
                "samples": len(rows), "source": "rocm-smi --showpower --showclocks, median over the sustained leg"}


def b_alg_of(prog, ns, T):
    return ns * (4 * T * (prog.n_in + prog.n_out) + 8 * prog.n_state + 4 * prog.n_param)


def gather_streams(torch, y, ids, tile):
    """[T, len(ids), w] numpy of the streams `ids` (local indices) of a frame tensor."""
    idt = torch.as_tensor(ids, device=y.device, dtype=torch.long)
    if tile:
        return y[idt // tile, :, idt % tile, :].permute(1, 0, 2).contiguous().cpu().numpy()
    return y[:, idt].contiguous().cpu().numpy()


def sample_ids(ns, k, seed):
    import numpy as np

    rng = np.random.default_rng(seed)
    return np.unique(np.concatenate([[0, 1, 63, 64, ns - 1], rng.integers(0, ns, k)]))


def ndiff_bits(a, b):
    import numpy as np

    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    return int((a.view(np.uint32) != b.view(np.uint32)).sum())


def parity_string(nd, n_streams, T):
    return f"bitwise-equal on {n_streams} random streams x {T} samples" if nd == 0 else f"MISMATCH {nd} samples"


def traffic_of(kernel, workload_key):
    """PMC HBM bytes per launch of exactly this kernel symbol on this workload, or None."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return None
    return json.load(open(tpath)).get(f"{kernel}|{workload_key}")


def measure_config(torch, F, prog, x, y, state, params, ns, T, tile, steps, workload_key, do_tune=True):
    """library default and (optionally) the tuned plan of one workload: ms per launch, GB/s, fraction of peak."""
    b = b_alg_of(prog, ns, T)
    out = {}

    def run(v):
        prog.run_block(x, state=state, params=params, out=y, variant=v)

    def timed(label, v):
        for _ in range(3):
            run(v)
        torch.cuda.synchronize()
        ms = event_ms(torch, lambda: run(v), steps)
        k = prog.kernel_name(v if v is not None else prog.plan(ns, tile), ns, T, tile)
        out[label] = {"kernel": k, "avg_launch_ms": round(ms, 4), "Msamples_per_s": round(ns * T / ms / 1e3, 1),
                      "achieved_GBs": round(b / ms / 1e6, 1), "frac": round(b / ms / 1e6 / HBM_PEAK_GBS, 4),
                      "traffic": traffic_of(k, workload_key)}

    # what a caller who never tunes gets: no variant.  The first big launch of a shape measures the candidates at hand by itself
    # (the warm-up launches above the timed ones; FLOWZ_HIP_AUTOTUNE=0 / --no-autotune: the static choice)
    timed("library_default", None)
    tv = None
    if do_tune:
        tv, _ = prog.tune(x, state=state, params=params, out=y)
        timed("tuned", tv)
    out["algorithmic_bytes_per_launch"] = b
    out["steps"] = steps
    best = max((k for k in ("library_default", "tuned") if k in out), key=lambda k: out[k]["frac"])
    out["frac"] = out[best]["frac"]
    out["best_plan"] = best
    return out, tv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=1 << 20, help="streams PER GPU (weak scaling)")
    ap.add_argument("--samples", type=int, default=4096, help="samples per block")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --streams per GPU (config 5: 1 M per GPU); strong: --streams-total divided over the GPUs")
    ap.add_argument("--streams-total", type=int, default=1 << 23, help="total streams of --scaling strong (SURVEY 8d config 5: 8 M)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl == RCCL over xGMI (default); gloo rehearses the N > 1 path when all ranks share one GPU")
    ap.add_argument("--lanes", type=int, default=0, help="streams per lane (0 = auto)")
    ap.add_argument("--unroll", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0,
                    help="frame layout of the headline workload: 0 = plain time-major [t][stream] (SURVEY 8d's device layout; since round "
                         "3 also the fastest: CU-wide workgroups in lockstep, XCD-wide synchronised), else streams per frame tile "
                         "(stream-tiled layout [tile][t][stream])")
    ap.add_argument("--secondary-tile", type=int, default=8192,
                    help="streams per frame tile of the secondary configs and of the tiled-layout leg (1-wire frames; 0 = time-major)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config2", action="store_true", help="skip the 65 536-stream measurement (BASELINE configs[1])")
    ap.add_argument("--no-config34", action="store_true", help="skip BASELINE configs[2] and [3] (4-parallel sum, oscillator chain)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s back-to-back run")
    ap.add_argument("--only", default="", help="profiling aid: run ONLY this secondary config (config2|config2h|config2q|config3|config3f|config4) "
                                               "with the forced / default variant and print its object")
    ap.add_argument("--no-autotune", action="store_true",
                    help="do not try the alternative kernel variants during warm-up (the pool's boxes differ by a few %%)")
    ap.add_argument("--no-layout-legs", action="store_true",
                    help="skip the same workload on the two contract layouts: plain time-major frames [t][stream] (SURVEY 8d) and "
                         "stream-major buffers [stream][t] (the reference's calling convention, test/benchmark.cpp:137-147)")
    args = ap.parse_args()
    if args.no_autotune:
        os.environ["FLOWZ_HIP_AUTOTUNE"] = "0"               # library_default = the static choice, nothing measured on first use

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without a launcher: become `torch.distributed.run` with one rank per GPU
        import subprocess
        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import numpy as np
    import torch
    import torch.distributed as dist

    from zignal_amd import dist as zdist
    from zignal_amd import flowz as F
    from zignal_amd import workloads as W

    SEED = W.SEED
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the flow-graph evaluator has no CPU path")
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % n_dev)                # (gloo rehearsal: several ranks may share device 0)
    dev = torch.device("cuda", local_rank % n_dev)
    distributed = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    stats_dev = dev if args.dist_backend == "nccl" else None

    T = args.samples
    total = args.streams_total if args.scaling == "strong" else args.streams * world
    begin, end = zdist.shard_range(total, rank, world)       # this rank's global stream ids
    ns = end - begin
    prog = F.compile(F.from_sexpr(W.df1_cascade(6)))
    forced = bool(args.lanes or args.unroll or args.block or args.flags)
    variant = F.make_variant(args.lanes, args.unroll, args.block, args.flags) if forced else None
    tile = pick_tile(ns, args.tile)
    lay = f"tile{tile}" if tile else "timemajor"

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- secondary configs (rank 0, N == 1): each frees its buffers before the next ---------------------------------
    def config2(ns2=65536):
        t2 = pick_tile(ns2, args.secondary_tile)
        x2, y2 = frames(torch, dev, ns2, T, 1, t2), frames(torch, dev, ns2, T, 1, t2)
        st2 = torch.zeros((prog.n_state, ns2), dtype=torch.float32, device=dev)
        F.synth_fill(x2, SEED)
        key = f"cascade6_{ns2}x{T}_" + (f"tile{t2}" if t2 else "timemajor")
        if args.only:
            for _ in range(20):
                prog.run_block(x2, state=st2, out=y2, variant=variant)
            ms = event_ms(torch, lambda: prog.run_block(x2, state=st2, out=y2, variant=variant), 200)
            return {"kernel": prog.kernel_name(variant, ns2, T, t2), "avg_launch_ms": round(ms, 4)}
        for _ in range(20):
            prog.run_block(x2, state=st2, out=y2)
        res, tv = measure_config(torch, F, prog, x2, y2, st2, None, ns2, T, t2, 200, key, do_tune=not args.no_autotune)
        st2.zero_()
        prog.run_block(x2, state=st2, out=y2, variant=tv)
        ids = sample_ids(ns2, PARITY_STREAMS, 12)
        from oracle import coracle, flowz_oracle as O
        want = coracle.df1_cascade([W.STABLE] * 6, O.synth_input(SEED, ids, T))
        res["parity"] = parity_string(ndiff_bits(gather_streams(torch, y2, ids, t2), want), len(ids), T)
        res["workload"] = (f"6-stage DF1 cascade, {ns2} streams x {T}-sample block " + ("(BASELINE configs[1]), " if ns2 == 65536 else "(configs[1] with fewer streams than lanes), ")
                           + (f"tiled:{t2}" if t2 else "time-major"))
        # compatibility with round 1's keys: the best plan's figures at top level
        res.update({k: res[res["best_plan"]][k] for k in ("avg_launch_ms", "Msamples_per_s", "achieved_GBs", "kernel")})
        return res

    def config3(fanout):
        g = W.par4_sum_fanout() if fanout else W.par4_sum()
        p3 = F.compile(F.from_sexpr(g))
        # the 4-wire sum on stream tiles (its rows are wide: tiles stream best), the 1-wire fan-out variant on plain time-major
        # frames like the headline (CU-wide lockstep workgroups, XCD-wide synchronised: 0.80 against 0.77-0.78 on tiles)
        t3 = 0 if fanout else pick_tile(ns3, p3.recommended_tile_streams())
        x3, y3 = frames(torch, dev, ns3, T, p3.n_in, t3), frames(torch, dev, ns3, T, 1, t3)
        st3 = torch.zeros((p3.n_state, ns3), dtype=torch.float32, device=dev)
        F.synth_fill(x3, SEED)
        key = ("par4f_" if fanout else "par4_") + f"{ns3}x{T}_" + (f"tile{t3}" if t3 else "timemajor")
        if args.only:
            for _ in range(3):
                p3.run_block(x3, state=st3, out=y3, variant=variant)
            ms = event_ms(torch, lambda: p3.run_block(x3, state=st3, out=y3, variant=variant), 5)
            return {"kernel": p3.kernel_name(variant, ns3, T, t3), "avg_launch_ms": round(ms, 4)}
        res, tv = measure_config(torch, F, p3, x3, y3, st3, None, ns3, T, t3, 10, key, do_tune=not args.no_autotune)
        st3.zero_()
        p3.run_block(x3, state=st3, out=y3, variant=tv)
        ids = sample_ids(ns3, PARITY_STREAMS, 13)
        from oracle import coracle, flowz_oracle as O
        xh = O.synth_input(SEED, ids, T, n_wires=p3.n_in)
        nd_in = ndiff_bits(gather_streams(torch, x3, ids, t3), xh)
        want = coracle.par4_sum(W.PAR4_SETS, xh, fanout=fanout)
        nd = ndiff_bits(gather_streams(torch, y3, ids, t3), want)
        res["parity"] = parity_string(nd + nd_in, len(ids), T)
        res["workload"] = (("(_1,_1,_1,_1) |= " if fanout else "") + f"(bq|bq|bq|bq) |= (_1+_2+_3+_4), {p3.n_in} input wire(s), {ns3} streams x {T}-sample "
                           f"block (BASELINE configs[2]), " + (f"tiled:{t3}" if t3 else "time-major"))
        return res

    def config4():
        p4 = F.compile(F.from_sexpr(W.osc_chain(6)))
        t4 = pick_tile(ns3, args.secondary_tile)
        x4, y4 = frames(torch, dev, ns3, T, 1, t4), frames(torch, dev, ns3, T, 1, t4)
        st4 = torch.zeros((p4.n_state, ns3), dtype=torch.float32, device=dev)
        P = W.osc_chain_params(SEED + 1, np.arange(ns3))
        pd = torch.from_numpy(P).to(dev)
        x4.zero_()                                           # a dirac at t = 0 on every stream
        (x4[:, 0] if t4 else x4[0]).fill_(1.0)
        key = f"osc6_{ns3}x{T}_" + (f"tile{t4}" if t4 else "timemajor")
        if args.only:
            for _ in range(3):
                p4.run_block(x4, state=st4, params=pd, out=y4, variant=variant)
            ms = event_ms(torch, lambda: p4.run_block(x4, state=st4, params=pd, out=y4, variant=variant), 10)
            return {"kernel": p4.kernel_name(variant, ns3, T, t4), "avg_launch_ms": round(ms, 4)}
        res, tv = measure_config(torch, F, p4, x4, y4, st4, pd, ns3, T, t4, 20, key, do_tune=not args.no_autotune)
        st4.zero_()
        p4.run_block(x4, state=st4, params=pd, out=y4, variant=tv)
        ids = sample_ids(ns3, PARITY_STREAMS, 14)
        from oracle import coracle
        xh = np.zeros((T, len(ids), 1), np.float32)
        xh[0] = 1.0
        want = coracle.osc_chain(np.ascontiguousarray(P[:, ids]), xh)
        res["parity"] = parity_string(ndiff_bits(gather_streams(torch, y4, ids, t4), want), len(ids), T)
        res["workload"] = (f"resonator oscillator -> 6 x DF1, 31 per-stream coefficients, dirac drive, {ns3} streams x {T}-sample block "
                           f"(BASELINE configs[3]), " + (f"tiled:{t4}" if t4 else "time-major"))
        return res

    def frame_layout_leg(tl):
        """The headline workload on the OTHER frame layout -- stream tiles when the headline runs on plain time-major frames
        [t][stream] (SURVEY 8d's device layout, the default), time-major frames when it was asked to run on tiles: library default and tuned."""
        tl = pick_tile(ns, tl)
        x2, y2 = frames(torch, dev, ns, T, 1, tl), frames(torch, dev, ns, T, 1, tl)
        st2 = torch.zeros((prog.n_state, ns), dtype=torch.float32, device=dev)
        F.synth_fill(x2, SEED)
        key = f"cascade6_{ns}x{T}_" + (f"tile{tl}" if tl else "timemajor")
        res, tv = measure_config(torch, F, prog, x2, y2, st2, None, ns, T, tl, max(5, args.steps // 2), key, do_tune=not args.no_autotune)
        st2.zero_()
        prog.run_block(x2, state=st2, out=y2, variant=tv)
        ids = sample_ids(ns, PARITY_STREAMS, 15)
        from oracle import coracle, flowz_oracle as O
        want = coracle.df1_cascade([W.STABLE] * 6, O.synth_input(SEED, ids, T))
        res["parity"] = parity_string(ndiff_bits(gather_streams(torch, y2, ids, tl), want), len(ids), T)
        res["workload"] = (f"6-stage DF1 cascade, {ns} streams x {T}-sample block, " +
                           (f"stream-tiled frames [tile][t][{tl} streams]" if tl else "plain time-major frames [t][stream] (SURVEY 8d)"))
        return res

    def stream_major_leg():
        """The headline workload on stream-major buffers [stream][t] -- one contiguous sample buffer per closure, the reference's
        calling convention (test/benchmark.cpp:137-147): fz_run_block_stream_major, library default and the best of SM_CANDIDATES."""
        xf = frames(torch, dev, ns, T, 1, tile)
        F.synth_fill(xf, SEED)
        xs = torch.empty((ns, T, 1), dtype=torch.float32, device=dev)
        F.frames_to_stream_major(xf, out=xs)
        del xf
        ys = torch.empty((ns, T, 1), dtype=torch.float32, device=dev)
        st2 = torch.zeros((prog.n_state, ns), dtype=torch.float32, device=dev)
        b = b_alg_of(prog, ns, T)
        reps = max(5, args.steps // 2)
        SMF = F.C.FZ_VF_STREAM_MAJOR

        def one(v):
            vv = None if v is None else F.make_variant(v[0], v[1], v[2], v[3])
            for _ in range(2):
                prog.run_block_stream_major(xs, state=st2, out=ys, variant=vv)
            torch.cuda.synchronize()
            ms = event_ms(torch, lambda: prog.run_block_stream_major(xs, state=st2, out=ys, variant=vv), reps)
            q = v or (0, 0, 0, 0)
            k = prog.kernel_name(F.make_variant(q[0], q[1], q[2], q[3] | SMF), ns, T)
            return {"kernel": k, "avg_launch_ms": round(ms, 4), "Msamples_per_s": round(ns * T / ms / 1e3, 1),
                    "achieved_GBs": round(b / ms / 1e6, 1), "frac": round(b / ms / 1e6 / HBM_PEAK_GBS, 4),
                    "traffic": traffic_of(k, f"cascade6_{ns}x{T}_streammajor")}

        res = {"library_default": one(None)}
        best_v = None
        if not args.no_autotune:
            cands = {}
            for v in W.SM_CANDIDATES:
                try:
                    cands[v] = one(v)
                except F.FlowzError:
                    pass
            best_v = min(cands, key=lambda v: cands[v]["avg_launch_ms"])
            # (as in fz_program_tune: a candidate replaces the default only when it wins by more than the scatter of repeats)
            if cands[best_v]["avg_launch_ms"] > 0.985 * res["library_default"]["avg_launch_ms"]:
                best_v = (0, 0, 0, 0)
            res["tuned"] = one(best_v)
            res["candidates_ms"] = {f"{v[0]},{v[1]},{v[2]},{v[3]}": c["avg_launch_ms"] for v, c in cands.items()}
        st2.zero_()
        prog.run_block_stream_major(xs, state=st2, out=ys, variant=None if best_v is None else F.make_variant(*best_v))
        ids = sample_ids(ns, PARITY_STREAMS, 16)
        idt = torch.as_tensor(ids, device=dev, dtype=torch.long)
        got = ys[idt].permute(1, 0, 2).contiguous().cpu().numpy()
        from oracle import coracle, flowz_oracle as O
        want = coracle.df1_cascade([W.STABLE] * 6, O.synth_input(SEED, ids, T))
        res["parity"] = parity_string(ndiff_bits(got, want), len(ids), T)
        res["algorithmic_bytes_per_launch"] = b
        best = max((k for k in ("library_default", "tuned") if k in res), key=lambda k: res[k]["frac"])
        res["frac"], res["best_plan"] = res[best]["frac"], best
        res["workload"] = (f"6-stage DF1 cascade, {ns} streams x {T}-sample block, stream-major buffers [stream][t] "
                           f"(the reference's calling convention, test/benchmark.cpp:137-147), no layout pass")
        return res

    ns3 = args.streams
    if args.only:
        fn = {"config2": config2, "config2h": lambda: config2(32768), "config2q": lambda: config2(16384), "config3": lambda: config3(False), "config3f": lambda: config3(True),
              "config4": config4, "timemajor": lambda: frame_layout_leg(0), "tiled": lambda: frame_layout_leg(args.secondary_tile),
              "streammajor": stream_major_leg}[args.only]
        print(json.dumps({args.only: fn()}), flush=True)
        return

    # ---- the headline workload ------------------------------------------------------------------------------------------
    x, y = frames(torch, dev, ns, T, 1, tile), frames(torch, dev, ns, T, 1, tile)
    state = torch.zeros((prog.n_state, ns), dtype=torch.float32, device=dev)
    F.synth_fill(x, SEED, stream0=begin)
    torch.cuda.synchronize()

    # plan selection (warm-up, untimed): when no variant is forced, let the library measure its candidate
    # variants for this shape on THIS board (fz_program_tune, the FFTW_MEASURE of this library; which one
    # wins differs from board to board) -- later launches without a variant use the winner
    tuned = None
    if not args.no_autotune and not forced:
        variant, _ = prog.tune(x, state=state, out=y)
        tuned = prog.kernel_name(variant, ns, T, tile)
        state.zero_()

    # first block from zero state: kept for the parity check (>= 1024 random streams across all tiles)
    prog.run_block(x, state=state, out=y, variant=variant)
    torch.cuda.synchronize()
    par_ids = sample_ids(ns, PARITY_STREAMS, 11) if rank == 0 else None
    first_block = gather_streams(torch, y, par_ids, tile) if rank == 0 else None
    for _ in range(max(args.warmup - 1, 0)):
        prog.run_block(x, state=state, out=y, variant=variant)

    # HIP events on the launch stream around the K back-to-back launches (one pair: an event per
    # launch costs tens of microseconds of GPU time each, visible on sub-millisecond kernels)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        prog.run_block(x, state=state, out=y, variant=variant)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    kern_avg_s = ev0.elapsed_time(ev1) / args.steps / 1e3

    checksum = zdist.bits_checksum(y[:, -1] if tile else y[-1])      # last time step of every stream: exact, shard-independent
    stats = zdist.reduce_stats(wall, float(ns) * T * args.steps, checksum, device=stats_dev)
    b_alg = b_alg_of(prog, ns, T)

    # the same launches back to back for >= 2 s: whatever the power management does to the clocks has happened by then
    sustained = None
    if rank == 0 and world == 1 and not args.no_sustained:
        # batches of launches (one HIP-event pair each) until >= 2 s of GPU time have gone by
        batch = max(args.steps, int(math.ceil(0.25 / max(kern_avg_s, 1e-6))))
        n_sus, ms_tot = 0, 0.0
        sampler = PowerSampler(dev.index or 0)
        sampler.start()
        while ms_tot < 2000.0 and n_sus < 4_000_000:
            ms_tot += event_ms(torch, lambda: prog.run_block(x, state=state, out=y, variant=variant), batch) * batch
            n_sus += batch
        ms_sus = ms_tot / n_sus
        sustained = {"launches": n_sus, "seconds": round(ms_tot / 1e3, 3), "avg_launch_ms": round(ms_sus, 4),
                     "achieved_GBs": round(b_alg / ms_sus / 1e6, 1), "frac": round(b_alg / ms_sus / 1e6 / HBM_PEAK_GBS, 4),
                     "board": sampler.stop()}

    # copy-kernel yardstick (same bytes in + out), rank 0 only
    copy_gbs = None
    if rank == 0:
        F.copy_probe(x, y)
        torch.cuda.synchronize()
        copy_gbs = 2.0 * x.numel() * 4 / (event_ms(torch, lambda: F.copy_probe(x, y), 3) / 1e3) / 1e9

    secondary = {}
    if rank == 0 and world == 1:
        del x, y, state
        torch.cuda.empty_cache()
        if not args.no_layout_legs:
            if tile:
                secondary["time_major_layout"] = frame_layout_leg(0)
            else:
                secondary["tiled_layout"] = frame_layout_leg(args.secondary_tile)
            torch.cuda.empty_cache()
            secondary["stream_major_layout"] = stream_major_leg()
            torch.cuda.empty_cache()
        if not args.no_config2 and ns != 65536:
            secondary["config2_65536_streams"] = config2()
            torch.cuda.empty_cache()
            secondary["cascade6_32768_streams"] = config2(32768)    # below one wave per SIMD: the wave-split kernel (two parts)
            torch.cuda.empty_cache()
            secondary["cascade6_16384_streams"] = config2(16384)    # a quarter of a wave per SIMD: three parts
            torch.cuda.empty_cache()
        if not args.no_config34:
            secondary["config3_par4_sum"] = config3(False)
            torch.cuda.empty_cache()
            secondary["config3_par4_sum_fanout"] = config3(True)
            torch.cuda.empty_cache()
            secondary["config4_osc_chain"] = config4()
            torch.cuda.empty_cache()

    if rank == 0:
        achieved = b_alg / kern_avg_s / 1e9
        kname = prog.kernel_name(variant if variant is not None else prog.plan(ns, tile), ns, T, tile)
        traffic = traffic_of(kname, f"cascade6_{ns}x{T}_{lay}")
        line = {
            "metric": "Msamples/sec/GPU + achieved HBM GB/s, 6-biquad cascade, 1M streams",
            "value": round(stats["samples"] / stats["seconds"] / 1e6, 1),
            "unit": "Msamples/s",
            "per_gpu": round(stats["samples"] / stats["seconds"] / 1e6 / world, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(stats["seconds"] / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"6-stage DF1 biquad cascade (flowz fwd|=bwd x6), {ns} streams/GPU x {T}-sample block, "
                                   f"uniform stable coefficients, "
                                   + (f"stream-tiled frames [tile][t][{tile} streams]" if tile else "time-major frames [t][stream]"),
                       "layout": f"tiled:{tile}" if tile else "time-major",
                       "streams_per_gpu": ns, "block_samples": T, "streams_total": total,
                       "parallelism": f"stream-sharded x{world}, no data-path collective"
                                      + (f" (statistics reduced over {args.dist_backend})" if distributed else ""),
                       "kernel_variant": {"streams_per_lane": args.lanes, "unroll": args.unroll,
                                          "block_threads": args.block, "flags": args.flags,
                                          "autotuned": tuned}},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_kernel": kname if traffic is not None else None,
                         "kernel": kname, "algorithmic_bytes_per_launch": b_alg,
                         "avg_launch_ms": round(kern_avg_s * 1e3, 4),
                         "measured_copy_GBs": round(copy_gbs, 1) if copy_gbs else None,
                         "frac_of_measured_copy": round(achieved / copy_gbs, 4) if copy_gbs else None},
            "checksum": stats["checksum"],
        }
        if sustained is not None:
            line["roofline"]["sustained"] = sustained
        line.update(secondary)
        if world == 1 and not args.no_cpu_baseline:
            from oracle import coracle, flowz_oracle as O
            coefs = [W.STABLE] * 6
            want = coracle.df1_cascade(coefs, O.synth_input(SEED, par_ids + begin, T))
            line["parity"] = parity_string(ndiff_bits(first_block, want), len(par_ids), T)
            line["cpu_baseline"] = cpu_baseline(T, SEED, coefs)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
