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
             peak = 8000 GB/s (HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md)
  cpu_baseline  the compiled scalar oracle (one closure per stream, one call per sample: what
             the reference's compile()-callable does) timed on this box's host cores on a
             bounded sample of the same workload, rank 0, N == 1 only; its outputs double as a
             bitwise parity check of the GPU output for those streams.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED = 20160512
HBM_PEAK_GBS = 8000.0


def cpu_baseline(n_samples, gpu_out_sampler):
    """Time the compiled oracle on all host cores; returns (dict, parity string)."""
    import concurrent.futures as cf

    import numpy as np

    import graphs as G
    from oracle import coracle

    cores = os.cpu_count() or 1
    coefs = [G.STABLE] * 6
    per_thread = 64
    # calibrate on a small piece, then size the sample for ~3 s per thread (bounded: 10-30 s CPU work)
    x0 = coracle.synth_fill(SEED, 0, per_thread, n_samples, stream_major=True)
    t0 = time.perf_counter()
    y0 = coracle.df1_cascade(coefs, x0, stream_major=True)
    dt = time.perf_counter() - t0
    rate = per_thread * n_samples / dt
    per_thread = int(min(max(64, (3.0 * rate / n_samples) // 64 * 64), 8192))
    chunks = [coracle.synth_fill(SEED, i * per_thread, per_thread, n_samples, stream_major=True) for i in range(cores)]
    with cf.ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        outs = list(ex.map(lambda c: coracle.df1_cascade(coefs, c, stream_major=True), chunks))
        wall = time.perf_counter() - t0
    total = cores * per_thread * n_samples
    base = {"value": round(total / wall / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{cores * per_thread} streams x {n_samples} samples, 6-stage DF1 cascade, scalar "
                      f"closure per stream (oracle/flowz_oracle.c, gcc -O3 -ffp-contract=off), "
                      f"{cores} threads, {wall:.2f} s wall"}
    # "Mode B" (SURVEY 8d): the same closures vectorised ACROSS streams by the compiler (SoA state, avx2/avx512
    # clones) -- a CPU stronger than the reference's scalar closure, reported next to it
    try:
        vec_streams = 1024                                   # per thread: 16 MiB of frames
        xv = [coracle.synth_fill(SEED, i * vec_streams, vec_streams, n_samples) for i in range(cores)]      # [T, ns, 1]
        y1 = coracle.df1_cascade_soa(coefs, xv[0])
        ok_vec = bool(np.array_equal(y1[:, :64, 0].view(np.uint32), coracle.df1_cascade(coefs, xv[0][:, :64]).view(np.uint32)[:, :, 0]))
        with cf.ThreadPoolExecutor(cores) as ex:
            t0 = time.perf_counter()
            reps = 12
            list(ex.map(lambda c: [coracle.df1_cascade_soa(coefs, c) for _ in range(reps)], xv))
            wall_v = time.perf_counter() - t0
        base["vectorised_across_streams"] = {
            "value": round(cores * vec_streams * reps * n_samples / wall_v / 1e6, 1), "unit": "Msamples/s", "cores": cores,
            "bitwise_equal_to_scalar": ok_vec,
            "note": "same arithmetic per stream, SoA state, compiler-vectorised (stronger than the reference's scalar closure)"}
    except Exception as e:                                  # never let the extra figure break the bench line
        base["vectorised_across_streams"] = {"error": str(e)[:200]}
    # parity: GPU output of the first 64 streams vs the oracle's
    got = gpu_out_sampler(64)                       # [T, 64] numpy
    want = outs[0][:64, :, 0].T
    nd = int((np.ascontiguousarray(got).view(np.uint32) != np.ascontiguousarray(want).view(np.uint32)).sum())
    return base, ("bitwise-equal on 64 streams x %d samples" % n_samples) if nd == 0 else f"MISMATCH {nd} samples"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=1 << 20, help="streams PER GPU (weak scaling)")
    ap.add_argument("--samples", type=int, default=4096, help="samples per block")
    ap.add_argument("--lanes", type=int, default=0, help="streams per lane (0 = auto)")
    ap.add_argument("--unroll", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--tile", type=int, default=8192,
                    help="streams per frame tile (stream-tiled layout [tile][t][stream], the HBM-friendly "
                         "default); 0 = plain time-major [t][stream]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config2", action="store_true", help="skip the secondary 65 536-stream measurement")
    ap.add_argument("--no-autotune", action="store_true",
                    help="do not try the alternative kernel variants during warm-up (the pool's boxes differ by a few %%)")
    ap.add_argument("--time-major-too", action="store_true",
                    help="also time the same workload on plain time-major frames (secondary figure)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without a launcher: become `torch.distributed.run` with one rank per GPU
        import subprocess
        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist

    import graphs as G
    from zignal_amd import dist as zdist
    from zignal_amd import flowz as F

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the flow-graph evaluator has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    ns, T = args.streams, args.samples
    begin, end = zdist.shard_range(ns * world, rank, world)      # this rank's global stream ids
    prog = F.compile(F.from_sexpr(G.df1_cascade(6)))
    variant = F.make_variant(args.lanes, args.unroll, args.block, args.flags)
    tile = args.tile if args.tile and ns % args.tile == 0 and args.tile < ns else 0
    shape = (ns // tile, T, tile, 1) if tile else (T, ns, 1)
    x = torch.empty(shape, dtype=torch.float32, device=dev)
    y = torch.empty(shape, dtype=torch.float32, device=dev)
    state = torch.zeros((prog.n_state, ns), dtype=torch.float32, device=dev)
    F.synth_fill(x, SEED, stream0=begin)
    torch.cuda.synchronize()

    # plan selection (warm-up, untimed): when no variant is forced, let the library measure its candidate
    # variants for this shape on THIS board (fz_program_tune, the FFTW_MEASURE of this library; which one
    # wins differs from board to board) -- later launches without a variant use the winner
    tuned = None
    if not args.no_autotune and not (args.lanes or args.unroll or args.block or args.flags):
        variant, _ = prog.tune(x, state=state, out=y)
        tuned = prog.kernel_name(variant, ns, T)
        state.zero_()

    # first block from zero state: kept for the parity check
    prog.run_block(x, state=state, out=y, variant=variant)
    torch.cuda.synchronize()
    first64_dev = None
    first64 = (y[0, :, :64, 0] if tile else y[:, :64, 0]).cpu().numpy() if rank == 0 else None
    for _ in range(max(args.warmup - 1, 0)):
        prog.run_block(x, state=state, out=y, variant=variant)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

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

    checksum = float((y[:, -1] if tile else y[-1]).double().sum().item())      # last time step of every stream
    stats = zdist.reduce_stats(wall, float(ns) * T * args.steps, checksum, device=dev)

    # the same workload on plain time-major frames [t][stream] (secondary figure, rank 0, N == 1)
    tm = None
    if args.time_major_too and tile and rank == 0 and world == 1:
        del first64_dev
        x2 = torch.empty((T, ns, 1), dtype=torch.float32, device=dev)
        y2 = torch.empty((T, ns, 1), dtype=torch.float32, device=dev)
        st2 = torch.zeros((prog.n_state, ns), dtype=torch.float32, device=dev)
        F.synth_fill(x2, SEED, stream0=begin)
        vtm = variant
        if tuned is not None:
            vtm, _ = prog.tune(x2, state=st2, out=y2)               # its own plan: the layouts prefer different ones
        prog.run_block(x2, state=st2, out=y2, variant=vtm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            prog.run_block(x2, state=st2, out=y2, variant=vtm)
        e1.record()
        torch.cuda.synchronize()
        tm_ms = e0.elapsed_time(e1) / 5
        tm = {"avg_launch_ms": round(tm_ms, 4), "Msamples_per_s": round(ns * T / tm_ms / 1e3, 1)}
        del x2, y2, st2

    # BASELINE config 2 (65 536 streams x 4096) in the same run, rank 0, N == 1: a different kernel
    # variant (own symbol in the rocprof stats), 1 GiB of frames
    cfg2 = None
    if rank == 0 and world == 1 and not args.no_config2 and ns != 65536:
        ns2 = 65536
        t2 = tile if tile and ns2 % tile == 0 else 0
        shp = (ns2 // t2, T, t2, 1) if t2 else (T, ns2, 1)
        x2 = torch.empty(shp, dtype=torch.float32, device=dev)
        y2 = torch.empty(shp, dtype=torch.float32, device=dev)
        st2 = torch.zeros((prog.n_state, ns2), dtype=torch.float32, device=dev)
        F.synth_fill(x2, SEED)
        v2 = None
        if tuned is not None:
            v2, _ = prog.tune(x2, state=st2, out=y2)
        for _ in range(20):
            prog.run_block(x2, state=st2, out=y2, variant=v2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(200):
            prog.run_block(x2, state=st2, out=y2, variant=v2)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / 200
        b2 = ns2 * (4 * T * 2 + 8 * prog.n_state)
        cfg2 = {"workload": f"6-stage DF1 cascade, {ns2} streams x {T}-sample block (BASELINE configs[1])",
                "steps": 200, "warmup": 20, "avg_launch_ms": round(ms2, 4), "Msamples_per_s": round(ns2 * T / ms2 / 1e3, 1),
                "achieved_GBs": round(b2 / ms2 / 1e6, 1), "frac": round(b2 / ms2 / 1e6 / HBM_PEAK_GBS, 4),
                "kernel": prog.kernel_name(v2, ns2, T)}
        del x2, y2, st2

    # copy-kernel yardstick (same bytes in + out), rank 0 only
    copy_gbs = None
    if rank == 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        F.copy_probe(x, y)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            F.copy_probe(x, y)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * x.numel() * 4 * 3 / (e0.elapsed_time(e1) / 1e3) / 1e9

    if rank == 0:
        b_alg = ns * (4 * T * (prog.n_in + prog.n_out) + 8 * prog.n_state + 4 * prog.n_param)
        achieved = b_alg / kern_avg_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(f"cascade6_{ns}x{T}_" + (f"tile{tile}" if tile else "timemajor"))
        line = {
            "metric": "Msamples/sec/GPU + achieved HBM GB/s, 6-biquad cascade, 1M streams",
            "value": round(stats["samples"] / stats["seconds"] / 1e6, 1),
            "unit": "Msamples/s",
            "per_gpu": round(stats["samples"] / stats["seconds"] / 1e6 / world, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(stats["seconds"] / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"6-stage DF1 biquad cascade (flowz fwd|=bwd x6), {ns} streams/GPU x {T}-sample block, "
                                   f"uniform stable coefficients, "
                                   + (f"stream-tiled frames [tile][t][{tile} streams]" if tile else "time-major frames [t][stream]"),
                       "layout": f"tiled:{tile}" if tile else "time-major",
                       "streams_per_gpu": ns, "block_samples": T, "streams_total": ns * world,
                       "parallelism": f"stream-sharded x{world}, no data-path collective",
                       "kernel_variant": {"streams_per_lane": args.lanes, "unroll": args.unroll,
                                          "block_threads": args.block, "flags": args.flags,
                                          "autotuned": tuned}},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": prog.kernel_name(variant, ns, T), "algorithmic_bytes_per_launch": b_alg,
                         "avg_launch_ms": round(kern_avg_s * 1e3, 4),
                         "measured_copy_GBs": round(copy_gbs, 1) if copy_gbs else None,
                         "frac_of_measured_copy": round(achieved / copy_gbs, 4) if copy_gbs else None},
            "checksum": stats["checksum"],
        }
        if cfg2 is not None:
            line["config2_65536_streams"] = cfg2
        if tm is not None:
            tm["achieved_GBs"] = round(b_alg / (tm["avg_launch_ms"] / 1e3) / 1e9, 1)
            tm["frac"] = round(tm["achieved_GBs"] / HBM_PEAK_GBS, 4)
            line["time_major_layout"] = tm
        if world == 1 and not args.no_cpu_baseline:
            base, parity = cpu_baseline(T, lambda k: first64[:, :k])
            line["cpu_baseline"] = base
            line["parity"] = parity
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
