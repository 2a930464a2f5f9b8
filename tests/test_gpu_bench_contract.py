"""The one JSON line `bench.py` prints: the keys the driver and the judge read (a short run on a
small batch; the `cpu_baseline` leg is skipped here, it is a 15 s oracle run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema():
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2",
         "--num-envs", "4096", "--no-cpu-baseline", "--min-time", "1"],
        capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac"):
        assert key in r, key
    assert r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # value = env-steps of the timed window / its wall time; the window is whole K-step blocks and
    # lasts at least --min-time (5 s by default, 1 s here) however small K is
    assert d["timed_steps"] % 6 == 0 and d["timed_steps"] >= 6
    assert d["timed_s"] >= 0.9, d["timed_s"]
    assert abs(d["ms_per_step"] * 1e-3 * d["timed_steps"] - d["timed_s"]) < 1e-6
    assert abs(d["value"] - 4096 * d["timed_steps"] / d["timed_s"]) / d["value"] < 1e-6
    assert d["roofline"]["launches"] == d["timed_steps"]
    assert d["config"]["params"] == {"precision": 1}
    # the GPU legs run before the CPU baseline and their wall time is stated (the driver's busy sampler)
    assert d["gpu_active_s"] >= d["timed_s"] and 0 < d["gpu_kernel_s_timed_region"] <= d["timed_s"]
    # issued flops (PMC) and the algorithmic count of the instrumented restatement, side by side
    assert r["flops_algorithmic"] > 1e4 and 0 < r["frac_useful"] < 1
    # the reference benchmark's own (async) loop beside the sync value: two batches of num_envs / 2 in flight
    am = d["async_mode"]
    assert am["batch_size"] == 2048 and am["batches_in_flight"] == 2 and am["value"] > 0
    assert abs(am["value"] - am["batch_size"] * am["steps"] / (am["ms_per_step"] * 1e-3 * am["steps"])) / am["value"] < 1e-6


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no torchrun around it starts 2 ranks (one process per GPU, the driver's own
    launch line) and rank 0 prints ONE line with n_gpus = 2 and the per-rank rates; on this 1-GPU box the two ranks
    share device 0 over gloo (plumbing only).  With the RCCL backend more ranks than visible GPUs is refused."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5",
         "--warmup", "2", "--num-envs", "4096", "--no-cpu-baseline", "--min-time", "0.5"],
        capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["num_envs_per_gpu"] == 4096
    assert [r["rank"] for r in d["per_rank"]] == [0, 1]
    assert abs(d["value"] - 2 * 4096 * d["timed_steps"] / d["timed_s"]) / d["value"] < 1e-6
    assert d["timed_s"] >= max(r["timed_s"] for r in d["per_rank"]) - 1e-9  # MAX over ranks
    assert "cpu_baseline" not in d and "async_mode" not in d  # rank 0 at N = 1 only
    import torch

    if torch.cuda.device_count() < 2:
        bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                             capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
        assert bad.returncode != 0 and "GPU(s) visible" in bad.stderr


def test_bench_rccl_code_path_with_one_rank():
    """The multi-rank code path over RCCL -- process group on the device, barriers around the timed region, MAX and
    gather of the ranks' times on device tensors, the optional obs all-gather on the pool's stream -- with ONE rank
    (`--force-process-group`): two RCCL ranks cannot share this box's single GPU, and the 8-GPU run is the driver's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-process-group", "--allgather",
         "--backend", "nccl", "--steps", "5", "--warmup", "2", "--num-envs", "4096", "--no-cpu-baseline",
         "--only-timed", "--min-time", "0.5"],
        capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and [r["rank"] for r in d["per_rank"]] == [0]
    assert abs(d["value"] - 4096 * d["timed_steps"] / d["timed_s"]) / d["value"] < 1e-6


def test_kernel_timing_modes_agree():
    """epa_set_timing: 1 = an event pair per launch, 2 = one pair around the window (what bench.py
    uses); both count every launch and give the same duration up to the inter-launch gaps."""
    import numpy as np
    import torch

    from envpool_amd.core.device_pool import DevicePool

    n = 16384
    pool = DevicePool("HalfCheetah", n, seed=0, max_episode_steps=1000)
    act = torch.rand((n, 6), device="cuda", dtype=torch.float64) * 2 - 1
    pool.send_device(None)
    pool.recv_device()
    res = {}
    for mode in (1, 2, 1, 2):
        pool.set_timing(mode)
        for _ in range(30):
            pool.send_device(act.data_ptr())
            pool.recv_device()
        ms, launches = pool.kernel_time_ms()
        pool.set_timing(0)
        assert launches == 30
        res.setdefault(mode, []).append(ms)
    pool.set_timing(2)  # an empty window reports nothing
    assert pool.kernel_time_ms() == (0.0, 0)
    pool.set_timing(0)
    a, b = min(res[1]), min(res[2])
    assert 0.02 < a < 5.0 and 0.02 < b < 5.0  # (a sanity range: ~0.05 ms on an MI355X)
    assert abs(a - b) / a < 0.25, res
