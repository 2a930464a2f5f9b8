"""bench.py — headline benchmark: env-steps/s of HalfCheetah-v4 at
num_envs=65536 (per GPU) with uniform random actions, auto-reset on.

One "step" = one batched `step()` of all envs of this rank (frame_skip=5
mj_steps each) with actions already resident in HBM (epa_send_device /
epa_recv_device: no PCIe in the timed region; DESIGN.md quotes the
PCIe-inclusive numpy-API rate).  Ranks are independent shards of the env-id
range (weak scaling, no collective on the data path).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# half-width of the task's action space (reference specs: humanoid.h:78, humanoid_standup.h:71 [-0.4, 0.4];
# pusher.h:61 [-2, 2]; every other gym-MuJoCo task here [-1, 1]): the reference benchmark draws
# `action_space.sample()` (benchmark/test_envpool.py), i.e. uniform in these bounds
ACTION_HI = {"Humanoid": 0.4, "HumanoidStandup": 0.4, "Pusher": 2.0}


def cpu_baseline(task="HalfCheetah", target_s=12.0, action_hi=1.0):
    """CPU baseline on a bounded sample of the same workload, all host cores.

    Where oracle/_ref/libref_mujoco.so travelled (built in the container that holds
    /root/reference): the reference's OWN AsyncEnvPool thread pool and task wrapper
    (envpool/core/async_envpool.h, envpool/mujoco/gym/*.h compiled in place), num_threads = cores - 1,
    sync Send / Recv loop -- over the oracle/mjcpu fp64 engine, because mj_step itself lives in
    un-vendored MuJoCo 3.6.0 ("our fp64 CPU restatement inside the reference
    threadpool on all cores").  The engine does the arithmetic, so kind stays "port".
    Otherwise: the plain port, envs spread over the cores with OpenMP."""
    import ctypes
    import subprocess

    # one OpenMP thread per core, pinned: unpinned teams gave 2.9e5 .. 5.5e5 run to run
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port"], check=True)
    from oracle import orc
    from oracle.orc import Oracle

    probe = Oracle(task, 1, seed=0, max_episode_steps=1000)
    probe.lib.mjcpu_num_threads.restype = ctypes.c_int
    cores = int(probe.lib.mjcpu_num_threads())
    num_envs = 64 * cores
    rng = np.random.default_rng(0)

    def leg(o, budget_s):
        o.reset()
        act = rng.uniform(-action_hi, action_hi, size=(num_envs, o.action_elems))
        o.time_steps(5, act)  # warm caches / leave the reset steps behind
        t = o.time_steps(20, act)
        steps = min(20000, max(5, int(budget_s / max(t / 20, 1e-6))))
        t = o.time_steps(steps, act)
        return num_envs * steps / t, steps, t

    threadpool = orc.have_ref_mujoco()
    # the same engine in a bare OpenMP loop over the envs (no queues): what the cores can do at best
    omp_value, omp_steps, omp_t = leg(Oracle(task, num_envs, seed=0, max_episode_steps=1000),
                                      target_s / 3 if threadpool else target_s)
    omp = {"value": omp_value, "unit": "env-steps/s", "runtime": "OpenMP loop over the envs",
           "sample": f"{num_envs} envs x {omp_steps} steps, {cores} OpenMP threads, {omp_t:.1f}s"}
    if not threadpool:
        return {**omp, "cores": cores, "kind": "port",
                "sample": "oracle/mjcpu fp64 engine, plain port; " + omp["sample"] +
                          " (reference mj_step not runnable: MuJoCo 3.6.0 un-vendored)"}
    # cores - 1 workers + the driving thread (it polls Recv): with `cores` workers the pool ran oversubscribed by one
    workers = max(1, cores - 1)
    value, steps, t = leg(Oracle(task, num_envs, seed=0, max_episode_steps=1000, kind="reference_mujoco",
                                 num_threads=workers), 2 * target_s / 3)
    return {
        "value": value,
        "unit": "env-steps/s",
        "cores": cores,
        "kind": "port",
        "runtime": "reference AsyncEnvPool + reference task wrapper",
        "sample": f"oracle/mjcpu fp64 engine inside the reference's own AsyncEnvPool threadpool + task wrapper "
                  f"(oracle/_ref/libref_mujoco.so), num_threads={workers} + the driving thread, sync Send/Recv; "
                  f"{num_envs} envs x {steps} steps, {t:.1f}s (reference mj_step not runnable: MuJoCo 3.6.0 "
                  f"un-vendored; moodycamel's semaphore is a shim that polls 400000 times (~10 ms) before it blocks)",
        "openmp_port": omp,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--num-envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--task", default="HalfCheetah")
    ap.add_argument("--precision", default="fp64", choices=["fp32", "fp64"],
                    help="fp32 exists for --task Ant only (the planar families are fp64 only since round 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only-timed", action="store_true",
                    help="skip the reset / numpy-API / async legs: under rocprofv3 their launches (half-size async "
                         "batches of the SAME kernel) would pollute the per-kernel means (tools/profile_bench.sh)")
    ap.add_argument("--action-scale", type=float, default=None,
                    help="actions are uniform in [-s, s]; default: the task's action-space bound")
    ap.add_argument("--min-time", type=float, default=5.0,
                    help="the timed region repeats the K-step block until it lasts at least this many "
                         "seconds (same repeat count on every rank); 0 = exactly K steps")
    ap.add_argument("--param", action="append", default=[], metavar="KEY=VALUE",
                    help="extra pool parameter (A/B switches such as sort_by_cost=0)")
    ap.add_argument("--no-bind", action="store_true",
                    help="leave the rank's host threads to the scheduler (default: each rank runs on the CPUs of its "
                         "GPU's NUMA node, envpool_amd.bind_host_to_device -- what the reference's "
                         "benchmark/numa_test.sh does with numactl)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (nccl = RCCL; gloo only to exercise the "
                         "multi-process path on a box with fewer GPUs than ranks)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="join the process group and run the multi-rank code path (barriers, MAX / gather of the "
                         "times, --allgather) even with ONE rank: exercises the RCCL calls on a 1-GPU box")
    ap.add_argument("--allgather", action="store_true",
                    help="also all-gather the obs batch over RCCL every step (optional "
                         "exchange of SURVEY §8e; off by default: the path needs no collective)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` launched directly (no torchrun around it): start the N ranks here -- one
        # process per GPU, the same launch the driver uses (python -m torch.distributed.run --nnodes=1
        # --nproc-per-node N --master-addr 127.0.0.1), like the reference's benchmark/numa_test.sh:15-21 starts one
        # process per NUMA node -- and let rank 0 print the one JSON line.
        if args.backend == "nccl" and args.gpus > torch.cuda.device_count():
            raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible (one rank per "
                             f"GPU over RCCL); --backend gloo lets ranks share a GPU to exercise the plumbing")
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(torchrun --nproc-per-node {args.gpus}) or run `python bench.py --gpus {args.gpus}` directly")
    ngpu = torch.cuda.device_count()
    if args.backend == "nccl" and world > 1 and local_rank >= ngpu:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ngpu} GPU(s) visible")
    dev_index = local_rank % max(ngpu, 1)  # gloo test mode may share a GPU between ranks
    use_pg = world > 1 or args.force_process_group
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (--force-process-group without a launcher: any free port)
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    from envpool_amd.core.affinity import bind_host_to_device
    from envpool_amd.core.device_pool import DevicePool

    # one process per GPU, on the CPUs of that GPU's NUMA node (doorbells, completion signals and pinned memory are
    # local from there); the CPU-baseline leg gets the process's original CPUs back
    cpus_before = os.sched_getaffinity(0)
    host_binding = {"node": None, "cpus": 0, "bound": False} if args.no_bind else bind_host_to_device(dev_index)

    n = args.num_envs
    if args.precision == "fp32" and args.task != "Ant":
        raise SystemExit("--precision fp32: only the Ant kernel has an fp32 arithmetic mode")
    params = {"precision": 1 if args.precision == "fp64" else 0}
    for kv in args.param:
        key, val = kv.split("=", 1)
        params[key] = float(val)
    pool = DevicePool(args.task, n, seed=0, max_episode_steps=1000, device=dev_index,
                      env_id_offset=rank * n, params=params)
    adim = int(np.prod(pool.action_shape))
    # ring of 16 pre-generated action batches, Philox seed 1234
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    ahi = args.action_scale if args.action_scale is not None else ACTION_HI.get(args.task, 1.0)
    ring = [(torch.rand((n, adim), generator=gen, device=dev, dtype=torch.float64) * 2 - 1) * ahi
            for _ in range(16)]
    torch.cuda.synchronize()

    obs_index = [k for k, _, _ in pool.state_keys].index("obs")
    obs_shape = pool.state_keys[obs_index][2]
    pool_stream = torch.cuda.ExternalStream(pool.stream, device=dev)
    gathered = None
    if args.allgather and use_pg:
        gathered = torch.empty((world * n, *obs_shape), device=dev, dtype=torch.float64)

    def step(i):
        pool.send_device(ring[i % 16].data_ptr())
        ptrs, k = pool.recv_device()  # outputs stay on the device
        if gathered is not None:
            from envpool_amd.torch_interop import _DevArray

            local = torch.as_tensor(_DevArray(ptrs[obs_index], (k, *obs_shape), np.float64),
                                    device=dev)
            with torch.cuda.stream(pool_stream):  # ordered after the step kernel
                dist.all_gather_into_tensor(gathered, local)

    pool.send_device(None)  # reset all (first step of every env is a reset anyway)
    pool.recv_device()
    for i in range(args.warmup):
        step(i)
    pool.synchronize()
    torch.cuda.synchronize()
    # The timed region is `repeats` back-to-back blocks of exactly K steps, long enough to last
    # --min-time seconds: a 20-step HalfCheetah block is 4.5 ms, too short for any outside clock or
    # busy sampler to see.  The count comes from one untimed calibration block, MAX over ranks.
    repeats = 1
    if args.min_time > 0:
        tc = time.perf_counter()
        for i in range(args.steps):
            step(i)
        pool.synchronize()
        torch.cuda.synchronize()
        tc = time.perf_counter() - tc
        if use_pg:
            t = torch.tensor([tc], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tc = float(t.item())
        repeats = max(1, min(int(np.ceil(args.min_time / max(tc, 1e-6))), max(1, 200000 // max(args.steps, 1))))
    timed_steps = repeats * args.steps
    if use_pg:
        dist.barrier()
    # HIP events on the pool's stream around the whole timed region (first launch .. after the last):
    # kernel_ms = that / launches.  (An event pair around EVERY launch keeps consecutive step
    # kernels ~12 us apart on this runtime -- 5 % of a 0.23 ms step; tools/bench_families.py still
    # times per launch.)
    pool.set_timing(2)
    t0 = time.perf_counter()
    for i in range(timed_steps):
        step(i)
    kernel_ms, launches = pool.kernel_time_ms()  # records the closing event, waits for the stream
    pool.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pool.set_timing(False)
    per_rank_s = [elapsed]
    if use_pg:
        dist.barrier()
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)  # each rank's own wall time of the timed region (reported per rank)
        per_rank_s = [float(x.item()) for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    reset_ms = None
    numpy_api = None
    gpu_legs_s = elapsed  # wall time of the legs that keep the GPU busy (for the driver's busy sampler)
    if world == 1 and not args.only_timed:
        # (a) one step in which EVERY env resets (mt19937 draws + state init + reset frame):
        #     the spike an episode boundary costs, outside the steady-state figure above
        pool.set_timing(True)
        for _ in range(3):
            pool.send_device(None)
            pool.recv_device()
        pool.synchronize()
        reset_ms, _ = pool.kernel_time_ms()
        pool.set_timing(False)
        # (b) T_numpy_api: the reference-compatible host path -- numpy actions in
        #     (H2D), every state key out as numpy (D2H) -- PCIe inclusive; never `value`
        ids = np.arange(rank * n, rank * n + n, dtype=np.int32)
        rng = np.random.default_rng(0)
        hact = [rng.uniform(-ahi, ahi, size=(n, adim)) for _ in range(4)]
        for i in range(30):  # (the first host-path steps allocate the pinned result blocks, the staging slots and start the copy helpers)
            pool.send(ids, hact[i % 4])
            pool.recv()
        k_np = max(5, min(args.steps, 200))
        t1 = time.perf_counter()
        for i in range(k_np):
            pool.send(ids, hact[i % 4])
            pool.recv()
        dt_np = time.perf_counter() - t1
        gpu_legs_s += dt_np
        numpy_api = {"value": n * k_np / dt_np, "unit": "env-steps/s", "ms_per_step": 1e3 * dt_np / k_np,
                     "steps": k_np, "note": "send(numpy) + recv() -> numpy, PCIe inclusive"}

    async_mode = None
    if world == 1 and n % 2 == 0 and not args.only_timed:
        # (c) the reference benchmark's OWN loop (benchmark/test_envpool.py:94-105): async mode,
        #     `recv()` then `send(action, env_id)` of batch_size rows with num_envs / batch_size batches
        #     in flight -- here on the device path, batch_size = num_envs / 2.  Successive batches run
        #     on different compute streams and fill each other's tails.  Reported
        #     next to `value` (the sync step() of SURVEY 8d), never as it.
        del pool
        b = n // 2
        apool = DevicePool(args.task, n, batch_size=b, seed=0, max_episode_steps=1000, device=dev_index,
                           env_id_offset=rank * n, params=params)
        aids = torch.arange(rank * n, rank * n + n, device=dev, dtype=torch.int32)
        torch.cuda.synchronize()
        for j in range(2):
            apool.send_device(None, b, aids[j * b:].data_ptr())

        def cycle(i):
            ptrs, k = apool.recv_device()
            apool.send_device(ring[i % 16].data_ptr(), k, ptrs[0])  # ptrs[0]: the batch's info:env_id

        for i in range(16):
            cycle(i)
        apool.synchronize()
        ta = time.perf_counter()
        for i in range(8):
            cycle(i)
        apool.synchronize()
        ta = time.perf_counter() - ta
        k_async = max(8, int(np.ceil(2.0 / max(ta / 8, 1e-6))))
        ta = time.perf_counter()
        for i in range(k_async):
            cycle(i)
        apool.synchronize()
        ta = time.perf_counter() - ta
        gpu_legs_s += ta
        async_mode = {"value": b * k_async / ta, "unit": "env-steps/s", "batch_size": b, "batches_in_flight": 2,
                      "steps": k_async, "ms_per_step": 1e3 * ta / k_async,
                      "note": "async recv_device -> send_device loop of batch_size rows (the reference benchmark's "
                              "loop, benchmark/test_envpool.py:94-105), device path"}
        del apool

    if rank == 0:
        total_env_steps = n * world * timed_steps
        value = total_env_steps / elapsed
        # algorithmic bytes / env-step: action+ids
        # in, every state key out, persistent fp64 state read + written
        # Humanoid: 8 + 136 in, 26 + 376 x 8 + 72 out, 2 x 72 persistent doubles
        # Pusher: 8 + 56 in, 26 + 23 x 8 + 24 out, 2 x 34 persistent doubles (q 11, v 9, warm 9, lag 5)
        alg_bytes = {"HalfCheetah": 708, "Ant": 1132, "Walker2d": 692, "Hopper": 476,
                     "Humanoid": 4402, "HumanoidStandup": 4362, "Pusher": 842}[args.task]
        frame_skip = {"HalfCheetah": 5, "Ant": 5, "Walker2d": 4, "Hopper": 4,
                      "Humanoid": 5, "HumanoidStandup": 5, "Pusher": 5}[args.task]
        achieved_gbs = alg_bytes * n / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # HBM traffic and flop counts come from the committed rocprofv3 PMC passes of
        # this same command (tools/profile_bench.sh -> profiles/pmc.json): PMC
        # collection needs its own rocprofv3 runs and cannot happen inside the bench.
        hum_quad = params.get("hum_layout", 1) != 0  # one env per lane quad (default)
        # HalfCheetah / Walker2d in fp64 run on the lane-group kernel (planar_layout 2 or 4, default 2)
        lg_layout = int(params.get("planar_layout", 0)) or (2 if n > 16384 else 4)  # the pool's own rule
        lg = (args.task in ("HalfCheetah", "Walker2d") and args.precision == "fp64" and lg_layout > 1
              and params.get("frame_stack", 1) == 1)
        if args.task == "Hopper":  # a group of ONE lane (default) unless planar_layout = 1
            lg = int(params.get("planar_layout", 0)) != 1
            lg_layout = 1
        kbase = ("AntStepKernel" if args.task == "Ant" else
                 ("Humanoid4StepKernel" if hum_quad else "HumanoidStepKernel")
                 if args.task.startswith("Humanoid") else
                 "PusherStepKernel" if args.task == "Pusher" else "CheetahStepKernel")
        fp64_only = args.task.startswith("Humanoid") or args.task == "Pusher"
        kname = kbase + ("<double>" if args.precision == "fp64" or fp64_only else "<float>")
        if lg:
            kbase = "PlanarLgStepKernel"
            # waves per SIMD the build aims at: "planar_waves" only exists for 4 lanes per env (with 1 or 2 lanes
            # per env LDS allows one wave and PlanarLgLaunch ignores the key)
            lg_waves = int(params.get("planar_waves", 1)) if lg_layout == 4 else 1
            kname = f"PlanarLgStepKernel<{lg_layout},{lg_waves}>"
        if args.task in ("Walker2d", "Hopper"):
            kname += f"[{args.task}]"
        if args.task == "HumanoidStandup":
            kname += "[Standup]"
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "pmc.json")) as f:
                pmc = json.load(f).get(f"{kname}@{n}", {})
        except OSError:
            pass
        traffic = None
        valu = None
        stale = None
        if pmc:
            # PMC counts belong to the build they were collected on: the entry carries a hash of the
            # kernel's sources + Makefile (tools/kernel_sources.py); a changed kernel is not priced with them
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from kernel_sources import source_hash
            stale = pmc.get("src_hash") != source_hash(kname)
        if pmc and pmc.get("num_envs") == n and not stale:
            traffic = pmc["traffic_bytes_per_launch"]
            peak_tf = 78.6 if args.precision == "fp64" or fp64_only else 157.3
            tf = pmc["flops_per_launch"] / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
            valu = {"achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": tf / peak_tf,
                    "flops_per_env_step": pmc["flops_per_env_step"],
                    "flops_source": pmc.get("source")}
        hbm = {"achieved": achieved_gbs, "peak": 8000.0, "unit": "GB/s",
               "frac": achieved_gbs / 8000.0,
               "algorithmic_bytes_per_env_step": alg_bytes,
               "note": "algorithmic bytes / kernel time, the fraction BASELINE.md asks for"}
        humanoid = args.task.startswith("Humanoid") and not hum_quad
        if humanoid or valu is None:
            # The one-env-per-lane Humanoid kernel streams its workspace through HBM (DESIGN.md
            # K3c): HBM binds.  Without PMC flop counts for this exact configuration only the HBM
            # figure can be stated.
            roof = {"bound": "hbm", **{k: hbm[k] for k in ("achieved", "peak", "unit", "frac")},
                    "traffic": traffic, "valu": valu}
        else:
            # the mj_step kernels are fp VALU-issue bound (~65 flop per algorithmic byte, far
            # above machine balance): that roofline is the headline object, HBM is secondary
            roof = {"bound": "valu", **{k: valu[k] for k in ("achieved", "peak", "unit", "frac")},
                    "flops_per_env_step": valu["flops_per_env_step"],
                    "flops_source": valu["flops_source"], "traffic": traffic, "hbm": hbm}
        # algorithmic flops per env-step from the instrumented fp64 restatement (tools/count_flops.py:
        # oracle/mjcpu compiled with an operation-counting scalar, per stage M1-M9) next to the ISSUED
        # figure above (wave instructions x 64 from PMC, which also counts frozen lanes, replicated
        # work and partly filled waves)
        try:
            with open(os.path.join(ROOT, "profiles", "flops_algorithmic.json")) as f:
                alg = json.load(f).get(args.task)
        except OSError:
            alg = None
        if alg and kernel_ms > 0:
            peak_tf = 78.6 if args.precision == "fp64" or fp64_only else 157.3
            useful_tf = alg["flops_per_env_step"] * n / (kernel_ms * 1e-3) / 1e12
            roof["flops_algorithmic"] = alg["flops_per_env_step"]
            roof["frac_useful"] = useful_tf / peak_tf
            roof["flops_algorithmic_source"] = (
                "profiles/flops_algorithmic.json: oracle/mjcpu (general 3-D dense restatement) with an "
                "operation-counting scalar, mean over a random-action rollout; the planar kernels exploit "
                "the 2-D structure and can issue fewer")
        # What of the issued work an env NEEDS: a wave executes every Newton trip until its slowest env has converged
        # (the solver share of a wave's cycles is paid max-over-the-wave's-envs times), so
        #   frac_necessary = frac x ((1 - solver_share) + solver_share x trips_needed / trips_executed),
        # shares from the stage timers of the diagnostic builds, trip counts from the kernels' own counters
        # (profiles/necessary_work.json names the source files).  And how busy the fp64 pipe is: every arithmetic
        # wave instruction (an FMA once) holds a SIMD's issue slot for 4 cycles.
        try:
            with open(os.path.join(ROOT, "profiles", "necessary_work.json")) as f:
                nw = json.load(f).get(f"{kname}@{n}")
        except OSError:
            nw = None
        if nw and valu is not None and roof.get("bound") == "valu":
            share = nw["solver_share"]
            roof["frac_necessary"] = roof["frac"] * ((1.0 - share) + share * nw["trips_needed"] / nw["trips_executed"])
            roof["frac_necessary_source"] = nw["source"]
        if pmc and not stale and pmc.get("arith_wave_insts_per_launch") and kernel_ms > 0:
            simds, clock_hz = 1024, 2.4e9  # 256 CUs x 4 SIMDs at the peak clock the 78.6 TFLOP/s figure assumes
            roof["fp64_issue_slot_util" if args.precision == "fp64" or fp64_only else "fp32_issue_slot_util"] = (
                pmc["arith_wave_insts_per_launch"] * 4.0 / (simds * kernel_ms * 1e-3 * clock_hz))
            if pmc.get("valu_wave_insts_per_launch"):
                roof["arith_share_of_valu_insts"] = pmc["arith_wave_insts_per_launch"] / pmc["valu_wave_insts_per_launch"]
        if stale:
            roof["stale"] = True  # profiles/pmc.json holds counts of an older build of this kernel: not used
        roof.update({"kernel": kbase, "kernel_ms": kernel_ms, "launches": launches,
                     "kernel_ms_method": "HIP events on the pool's stream before the first and after the last "
                                         "launch of the timed region, / launches",
                     "algorithmic_bytes_per_env_step": alg_bytes,
                     "traffic_note": "HBM bytes per launch from the committed rocprofv3 PMC passes of "
                                     "this command (profiles/pmc.json; FETCH_SIZE x2 + WRITE_SIZE)"})
        out = {
            "metric": f"env steps/sec (raw FPS) at num_envs={n}, {args.task}-v4",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / timed_steps,
            "timed_steps": timed_steps,  # = steps x repeats (see --min-time); value = env-steps of ALL of them / timed_s
            "timed_s": elapsed,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.task}-v4 num_envs={n} per GPU, frame_skip={frame_skip}, "
                            f"actions uniform in the action space [-{ahi:g}, {ahi:g}] resident in HBM, auto-reset on",
                "num_envs_per_gpu": n,
                "frames_per_sec": value * frame_skip,
                "sharding": f"env ids sharded over {world} GPU(s), no collective",
                "params": params,  # every engine key the pool was created with, --param overrides included
                "host_binding": host_binding,  # the rank's CPUs: its GPU's NUMA node unless --no-bind
            },
            "roofline": roof,
            # wall seconds of the GPU legs (timed sync region + numpy-API leg + async leg), which run
            # BEFORE the cpu_baseline leg; kernel time inside the timed region = launches x kernel_ms
            "gpu_active_s": gpu_legs_s,
            "gpu_kernel_s_timed_region": kernel_ms * launches * 1e-3,
        }
        # every rank's own rate over its own wall time; `value` = all env-steps / the SLOWEST rank's time
        out["per_rank"] = [{"rank": r, "device": r % max(ngpu, 1), "timed_s": ts,
                            "env_steps_per_s": n * timed_steps / ts} for r, ts in enumerate(per_rank_s)]
        if world == 1:
            out["reset_step_ms"] = reset_ms
            out["numpy_api"] = numpy_api
            out["async_mode"] = async_mode
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            os.sched_setaffinity(0, cpus_before)  # every host core the process was given
            out["cpu_baseline"] = cpu_baseline(args.task, action_hi=ahi)
        print(json.dumps(out))
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
