"""Per-family kernel throughput vs the roofline that binds it.

For every env family on the hot path: device-resident random actions, K timed
steps of all N envs through epa_send_device/epa_recv_device, kernel time from
HIP events on the pool's stream.  HBM-streaming families (classic_control,
toy_text, Atari post-process) report achieved algorithmic GB/s against the
8 TB/s HBM peak; the MuJoCo kernels are VALU-bound and report env-steps/s.
Prints one JSON line per family and a markdown table (-> profiles/).
"""
import argparse
import json
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# family, params, max_episode_steps, action kind, algorithmic bytes / env-step
FAMILIES = [
    ("CartPole", {}, 500, ("int", 2), 128),
    ("Pendulum", {"version": 1}, 200, ("float", 2.0), 92),
    ("MountainCar", {}, 200, ("int", 3), 96),
    ("MountainCarContinuous", {}, 999, ("float", 1.0), 96),
    ("Acrobot", {}, 500, ("int", 3), 144),
    ("Catch", {"height": 10, "width": 5}, 0, ("int", 3), 12 + 26 + 200 + 2 * 9),
    ("FrozenLake", {"size": 8}, 200, ("int", 4), 72),
    ("Taxi", {}, 200, ("int", 6), 72),
    ("NChain", {}, 1000, ("int", 2), 72),
    ("CliffWalking", {"is_slippery": 1}, 0, ("int", 4), 76),
    ("Blackjack", {}, 0, ("int", 2), 84),
    # gym-MuJoCo pendulums (mj_pendulum.hip.h): 16 in + 26 + obs out + 2 x (3 nv doubles + 5) state
    ("InvertedPendulum", {}, 1000, ("float64", 3.0), 16 + 58 + 106),
    ("InvertedDoublePendulum", {}, 1000, ("float64", 1.0), 16 + 114 + 172),
    ("Reacher", {}, 50, ("float64x2", 1.0), 24 + 26 + 88 + 16 + 2 * (12 * 8 + 16 + 5)),
    ("Swimmer", {}, 1000, ("float64x2", 1.0), 24 + 26 + 64 + 56 + 2 * (15 * 8 + 5)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--big", type=int, default=1 << 22, help="also run at this N (HBM-resident)")
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps after the reset (e.g. 700: past the first "
                    "exhaustion of every env's 624 generator words, where the block-wise mt19937 of rounds 1-4 paid its "
                    "twists)")
    ap.add_argument("--param", action="append", default=[], metavar="KEY=VALUE",
                    help="extra engine key for every pool (A/B switches such as mt_tile=1)")
    ap.add_argument("--families", default="", help="comma-separated subset (default: all)")
    ap.add_argument("--no-atari", action="store_true")
    ap.add_argument("--atari-sizes", default="1024,16384")
    ap.add_argument("--plan-out", default="", help="write the ordered list of (kernel tag, family, N, launches) "
                    "this run dispatched: tools/summarize_families.py splits a rocprofv3 trace by it")
    args = ap.parse_args()
    import torch

    from envpool_amd.core.device_pool import DevicePool
    from envpool_amd.atari import AtariPostProcess

    dev = torch.device("cuda", 0)
    rows = []
    plan = []
    want = [f for f in args.families.split(",") if f]
    sizes = [args.num_envs] + ([args.big] if args.big > 0 else [])
    for n in sizes:
        for fam, params, max_steps, (kind, p), alg in FAMILIES:
            if want and fam not in want:
                continue
            extra = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in args.param}
            pool = DevicePool(fam, n, seed=0, max_episode_steps=max_steps, params={**params, **extra})
            if kind == "int":
                ring = [torch.randint(0, p, (n,), device=dev, dtype=torch.int32) for _ in range(8)]
            elif kind == "float64x2":
                ring = [(torch.rand((n, 2), device=dev, dtype=torch.float64) * 2 - 1) * p
                        for _ in range(8)]
            elif kind == "float64":
                ring = [(torch.rand((n, 1), device=dev, dtype=torch.float64) * 2 - 1) * p
                        for _ in range(8)]
            else:
                ring = [(torch.rand((n, 1), device=dev) * 2 - 1) * p for _ in range(8)]
            torch.cuda.synchronize()
            pool.send_device(None)
            pool.recv_device()
            for i in range(args.warmup):
                pool.step_device(ring[i % 8].data_ptr())  # send + recv in one library call
            pool.synchronize()
            pool.set_timing(True)
            for i in range(args.steps):
                pool.step_device(ring[i % 8].data_ptr())  # send + recv in one library call
            ms, launches = pool.kernel_time_ms()
            gbs = alg * n / (ms * 1e-3) / 1e9
            # what a caller of the device path gets per step, gaps between the launches included: one event pair
            # around the whole window (timing mode 2), with step_device and with the send_device + recv_device pair
            pool.set_timing(2)
            for i in range(args.steps):
                pool.step_device(ring[i % 8].data_ptr())
            step_ms, _ = pool.kernel_time_ms()
            pool.set_timing(2)
            for i in range(args.steps):
                pool.send_device(ring[i % 8].data_ptr())
                pool.recv_device()
            pair_ms, _ = pool.kernel_time_ms()
            pool.set_timing(False)
            rec = {"family": fam, "num_envs": n, "kernel_us": ms * 1e3, "launches": launches,
                   "env_steps_per_s": n / (ms * 1e-3), "algorithmic_bytes": alg,
                   "achieved_GBps": gbs, "hbm_frac": gbs / 8000.0,
                   "step_us_step_device": step_ms * 1e3, "step_us_send_recv": pair_ms * 1e3,
                   "env_steps_per_s_step_device": n / (step_ms * 1e-3)}
            rows.append(rec)
            print(json.dumps(rec))
            # step-kernel launches of this configuration, in dispatch order: 1 reset + 10 warm-up + the timed ones
            plan.append({"family": fam, "num_envs": n, "skip": 1 + args.warmup, "timed": args.steps, "algorithmic_bytes": alg,
                         "hip_event_us": ms * 1e3})
            pool.close()
    # Atari post-process (K4): frames resident on the device
    for n in ([] if args.no_atari else [int(x) for x in args.atari_sizes.split(",") if x]):
        post = AtariPostProcess(n)
        frames = torch.randint(0, 256, (n, 2, 210, 160), device=dev, dtype=torch.uint8)
        obs = torch.empty((n, 4, 84, 84), device=dev, dtype=torch.uint8)
        torch.cuda.synchronize()
        stream = torch.cuda.ExternalStream(post.stream, device=dev)
        for _ in range(5):
            post.push_device(frames.data_ptr(), obs.data_ptr(), n)
        stream.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            t0.record()
            for _ in range(50):
                post.push_device(frames.data_ptr(), obs.data_ptr(), n)
            t1.record()
        stream.synchronize()
        ms = t0.elapsed_time(t1) / 50
        alg = 2 * 33600 + 3 * 7056 + 7056 + 4 * 7056
        gbs = alg * n / (ms * 1e-3) / 1e9
        rec = {"family": "AtariPostProcess", "num_envs": n, "kernel_us": ms * 1e3, "launches": 50,
               "env_steps_per_s": n / (ms * 1e-3), "algorithmic_bytes": alg,
               "achieved_GBps": gbs, "hbm_frac": gbs / 8000.0}
        rows.append(rec)
        print(json.dumps(rec))
        plan.append({"family": "AtariPostProcess", "num_envs": n, "skip": 5, "timed": 50, "algorithmic_bytes": alg,
                     "hip_event_us": ms * 1e3})
    if args.plan_out:
        with open(args.plan_out, "w") as f:
            json.dump(plan, f, indent=1)
    # kernel us: HIP event pair around every launch; step us: one event pair around the whole window of launches (what a
    # caller of the device path gets per step, gaps included) with epa_step_device / with epa_send_device + epa_recv_device
    print("\n| family | N | kernel us | env-steps/s | alg B/step | GB/s | frac of 8 TB/s | step us (step_device) | step us (send + recv) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['family']} | {r['num_envs']} | {r['kernel_us']:.1f} | {r['env_steps_per_s']:.3g} | "
              f"{r['algorithmic_bytes']} | {r['achieved_GBps']:.0f} | {r['hbm_frac']:.3f} | "
              f"{r.get('step_us_step_device', float('nan')):.1f} | {r.get('step_us_send_recv', float('nan')):.1f} |")


if __name__ == "__main__":
    main()
