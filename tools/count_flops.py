"""Algorithmic flop count per env-step from the instrumented fp64 restatement (BASELINE.md
section 3, SURVEY.md section 8d: "counted_flops taken from the instrumented restatement, not
estimated").

oracle/_build/libmjc_count.so is oracle/mjcpu compiled UNCHANGED with `double` renamed to an
operation-counting class (oracle/flopcount/count_real.h).  This tool runs a free-running rollout
of the plain port (random actions uniform in the action space, auto-reset on, `--warmup` steps
discarded), replays every env-step of the sampled envs through the counting build from the
port's own state, checks that both produce the same bits, and writes the per-stage (M1-M9)
counts to profiles/flops_algorithmic.json -- which bench.py reads for `roofline.flops_algorithmic`
and `roofline.frac_useful`.

    make -C oracle port count && python tools/count_flops.py
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.orc import Oracle  # noqa: E402

NSTAGE, NKIND = 12, 6
KINDS = ["add", "mul", "div", "sqrt", "trans", "cmp"]
STAGES = ["outside", "M1 kinematics+comPos", "M2 crb", "M3 collision", "M4 makeConstraint",
          "M5 fwdVelocity", "M6 fwdActuation", "M7 fwdAcceleration", "M8 fwdConstraint (solver)",
          "M9 integrator", "rnePostConstraint", "-"]
# task: (frame_skip, action half-width, envs, steps, warmup)
TASKS = {
    "HalfCheetah": (5, 1.0, 64, 120, 150),
    "Walker2d": (4, 1.0, 64, 120, 150),
    "Hopper": (4, 1.0, 64, 120, 150),
    "Ant": (5, 1.0, 32, 60, 100),
    "Pusher": (5, 2.0, 32, 60, 20),
    "Humanoid": (5, 0.4, 12, 30, 20),
    "HumanoidStandup": (5, 0.4, 12, 30, 60),
}


def load():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libmjc_count.so"))
    lib.mjc_count_create.restype = ctypes.c_void_p
    lib.mjc_count_create.argtypes = [ctypes.c_char_p]
    vp = ctypes.c_void_p
    lib.mjc_count_dims.argtypes = [vp, vp]
    lib.mjc_count_env_step.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.mjc_count_destroy.argtypes = [vp]
    return lib


def count(lib, task, seed=0):
    frame_skip, amp, n, steps, warmup = TASKS[task]
    orc = Oracle(task, n, seed=seed, max_episode_steps=1000)
    h = lib.mjc_count_create(task.encode())
    dims = np.zeros(4, np.int32)
    lib.mjc_count_dims(h, dims.ctypes.data)
    nq, nv, nu, _ = (int(x) for x in dims)
    rng = np.random.default_rng(seed)
    orc.reset()
    for _ in range(warmup):
        orc.step(rng.uniform(-amp, amp, size=(n, nu)))
    counts = np.zeros(NSTAGE * NKIND, np.uint64)
    stats = np.zeros(4)
    samples = mismatches = resets = 0
    for _ in range(steps):
        st = orc.get_state()
        act = rng.uniform(-amp, amp, size=(n, nu))
        out = orc.step(act)
        nxt = orc.get_state()
        live = out["elapsed_step"].ravel() > 0  # a reset row ran no mj_step: not an env-step sample
        for e in range(n):
            if not live[e]:
                resets += 1
                continue
            q = st[e, :nq].copy()
            v = st[e, nq:nq + nv].copy()
            w = st[e, nq + nv:nq + 2 * nv].copy()
            a = np.ascontiguousarray(act[e])
            lib.mjc_count_env_step(h, q.ctypes.data, v.ctypes.data, w.ctypes.data, a.ctypes.data,
                                   frame_skip, 0, counts.ctypes.data, stats.ctypes.data)
            samples += 1
            same = (np.array_equal(q.view(np.uint64), nxt[e, :nq].view(np.uint64))
                    and np.array_equal(v.view(np.uint64), nxt[e, nq:nq + nv].view(np.uint64))
                    and np.array_equal(w.view(np.uint64), nxt[e, nq + nv:nq + 2 * nv].view(np.uint64)))
            mismatches += not same
    lib.mjc_count_destroy(h)
    c = counts.reshape(NSTAGE, NKIND).astype(np.float64) / samples
    flops = c[:, :5].sum(axis=1)
    fwd = stats[3]
    res = {
        "task": task, "frame_skip": frame_skip, "samples_env_steps": samples,
        "reset_rows_skipped": resets, "counted_build_equals_port_bitwise": mismatches == 0,
        "flops_per_env_step": float(flops.sum()),
        "flops_per_mj_step": float(flops.sum() / frame_skip),
        "by_kind_per_env_step": {k: float(c[:, i].sum()) for i, k in enumerate(KINDS)},
        "by_stage_per_env_step": {STAGES[s]: float(flops[s]) for s in range(NSTAGE) if flops[s] > 0},
        "mean_nefc_last_forward": float(stats[0] / fwd), "mean_ncon_last_forward": float(stats[1] / fwd),
        "mean_solver_iterations_last_forward": float(stats[2] / fwd),
        "definition": "add + mul + div + sqrt + transcendental calls executed on mjtNum by oracle/mjcpu "
                      "(dense restatement; Newton run to 1e-12 scaled gradient, i.e. more iterations than "
                      "MuJoCo's 1e-8); comparisons / min / max / fabs listed under cmp, not counted",
    }
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", nargs="*", default=list(TASKS))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "flops_algorithmic.json"))
    args = ap.parse_args()
    lib = load()
    out = {}
    if os.path.exists(args.out):
        out = json.load(open(args.out))
    for t in args.tasks:
        r = count(lib, t)
        out[t] = r
        print(f"{t}: {r['flops_per_env_step']:.0f} flops/env-step ({r['flops_per_mj_step']:.0f}/mj_step), "
              f"solver {r['mean_solver_iterations_last_forward']:.2f} it, nefc {r['mean_nefc_last_forward']:.1f}, "
              f"bitwise == port: {r['counted_build_equals_port_bitwise']}, n={r['samples_env_steps']}")
        for k, v in r["by_stage_per_env_step"].items():
            print(f"    {k:32s} {v:12.0f}  {100 * v / r['flops_per_env_step']:5.1f} %")
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
