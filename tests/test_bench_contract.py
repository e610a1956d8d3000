"""bench.py prints ONE JSON line with the driver's contract keys (small workload, real GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract_small_workload():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--streams", "16384", "--samples", "512"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel"].startswith("fz_block_kernel_p")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["parity"].startswith("bitwise-equal")
    assert d["roofline"]["sustained"]["seconds"] >= 1.9 and d["roofline"]["sustained"]["frac"] > 0
    board = d["roofline"]["sustained"]["board"]        # rocm-smi during the sustained leg (best effort: None without rocm-smi)
    assert board is None or (board["package_W"] > 0 and board["sclk_MHz"] > 0 and board["samples"] >= 1)
    assert "traffic_kernel" in d["roofline"]
    for k in ("config2_65536_streams", "cascade6_32768_streams", "cascade6_16384_streams", "config3_par4_sum", "config3_par4_sum_fanout", "config4_osc_chain"):
        assert d[k]["parity"].startswith("bitwise-equal"), (k, d[k]["parity"])
        assert d[k]["library_default"]["frac"] > 0 and d[k]["tuned"]["kernel"].startswith("fz_block_kernel_p")
    assert d["config"]["layout"] == "time-major"                       # SURVEY 8d's device layout is the headline's (round 3)
    for k in ("tiled_layout", "stream_major_layout"):                  # the other frame layout and the reference's calling convention, same workload
        assert d[k]["parity"].startswith("bitwise-equal"), (k, d[k]["parity"])
        assert d[k]["library_default"]["frac"] > 0 and d[k]["tuned"]["kernel"].startswith("fz_block_kernel_p") and 0 < d[k]["frac"] < 1
    assert c["Msamples_per_s_per_core"] > 0 and c["physical_cores"] >= 1 and c["logical_cpus"] >= c["physical_cores"]
    assert c["cores"] <= c["threads"] and (c["cgroup_cpu_quota"] is None or c["cores"] <= max(1, round(c["cgroup_cpu_quota"])))
    assert 0.5 < c["Msamples_per_s_per_core"] * c["cores"] / c["value"] < 2.0
    assert abs(d["value"] - 16384 * 512 * 3 / (d["ms_per_step"] * 3 / 1e3) / 1e6) / d["value"] < 1e-2


def _run_bench(args, nproc=1, timeout=900, launcher=False):
    if nproc > 1 or launcher:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_rehearse_the_sharded_path():
    """The N > 1 path of bench.py (launcher env, shard_range, per-rank generator offset, barrier, max-over-ranks time,
    the statistics all-reduce) with 2 ranks that share device 0 and reduce over gloo: the integer checksum of the two
    shards must equal a single-process run over the union of the global stream ids.  Weak and strong scaling modes."""
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-autotune", "--no-config2", "--no-config34", "--no-sustained",
              "--no-layout-legs"]
    one = _run_bench(["--gpus", "1", "--streams", "262144"] + common)
    weak = _run_bench(["--gpus", "2", "--streams", "131072", "--dist-backend", "gloo"] + common, nproc=2)
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak"
    assert weak["config"]["streams_total"] == 262144 and weak["config"]["streams_per_gpu"] == 131072
    assert weak["checksum"] == one["checksum"] and isinstance(weak["checksum"], int)
    assert abs(weak["value"] - 262144 * 4096 * 2 / (weak["ms_per_step"] * 2 / 1e3) / 1e6) / weak["value"] < 1e-2
    strong = _run_bench(["--gpus", "2", "--scaling", "strong", "--streams-total", "262144", "--dist-backend", "gloo"] + common, nproc=2)
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong" and strong["config"]["streams_total"] == 262144
    assert strong["checksum"] == one["checksum"]


@pytest.mark.gpu
def test_bench_one_rank_reduces_its_statistics_over_rccl():
    """The `nccl` (= RCCL) branch of bench.py on real hardware: one rank under torch.distributed.run, process group on the GPU,
    the three statistics reduced by RCCL all-reduces on device tensors (zignal_amd/dist.py).  Same checksum and stream count
    as the plain single-process run."""
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-autotune", "--no-config2", "--no-config34", "--no-sustained",
              "--no-layout-legs"]
    one = _run_bench(["--gpus", "1", "--streams", "131072"] + common)
    rccl = _run_bench(["--gpus", "1", "--streams", "131072", "--dist-backend", "nccl"] + common, launcher=True)
    assert rccl["n_gpus"] == 1 and rccl["config"]["streams_total"] == 131072
    assert "statistics reduced over nccl" in rccl["config"]["parallelism"]
    assert "reduced over" not in one["config"]["parallelism"]
    assert rccl["checksum"] == one["checksum"] and isinstance(rccl["checksum"], int)
    assert abs(rccl["value"] - 131072 * 4096 * 2 / (rccl["ms_per_step"] * 2 / 1e3) / 1e6) / rccl["value"] < 1e-2
