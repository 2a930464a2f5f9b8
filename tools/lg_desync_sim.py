"""HOST EXPERIMENT: what would a wave of 32 HalfCheetah envs cost if its lane groups did not wait for each other at every
mj_step?  Replays per-env solver counts (tools/lg_desync/rollout_host.cpp: the kernel's own source on the host) through
(a) the shipped schedule -- five forward passes per env-step, each as long as its slowest env -- and (b) a
phase-decoupled wave loop: a group that has converged goes on to its next mj_step when at least `theta` groups wait
for a set-up phase (or nobody is solving).  Costs in cycles from the stage timers (profiles/r4c_lg_stage_timers.txt).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rollout(envs=2048, warm=150, steps=40, seed=1):
    so = "/tmp/liblg_rollout.so"
    src = os.path.join(ROOT, "tools", "lg_desync", "rollout_host.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    out = np.zeros((steps, envs, 5, 14), np.int16)
    L.lg_rollout(envs, warm, steps, seed, out.ctypes.data_as(ctypes.c_void_p))
    return out


# cycles (per wave): set-up of a forward pass (kinematics, smooth forces, constraint rows), one pass over the rows +
# stop test, factor / solve / M products of a trip, one line-search evaluation, integration of an mj_step, the rest
# of a chunk (loads, stores, bookkeeping)
COST = dict(S=5950.0, R=2650.0, F=1340.0, E=2000.0, I=900.0, X=4200.0)


def cost_sync(d, c=COST):
    """d: [32, 5, 14] counts of one wave's envs for one env-step -> cycles of the shipped schedule"""
    tot = c["X"]
    for s in range(5):
        rows, trips, ev = d[:, s, 0], d[:, s, 1], d[:, s, 2:]
        n_it = rows.max()
        tot += c["S"] + c["I"]
        for it in range(n_it):
            live_rows = rows > it
            if not live_rows.any():
                break
            tot += c["R"]
            live = trips > it
            if live.any():
                tot += c["F"] + c["E"] * ev[live, it].max()
    return tot


def cost_desync(d, theta, c=COST, fuse_integrate=True):
    """phase-decoupled loop.  Per group: sub (0..5), phase 0 = waits for set-up, 1 = solving, 2 = done."""
    n = d.shape[0]
    sub = np.zeros(n, int)
    phase = np.zeros(n, int)
    it = np.zeros(n, int)  # iterations done inside the current forward pass
    tot = c["X"]
    nset = 0
    nloop = 0
    while True:
        need = phase == 0
        solving = phase == 1
        if not need.any() and not solving.any():
            break
        if need.any() and (not solving.any() or need.sum() >= theta):
            tot += c["S"] + c["I"]  # integrate the previous mj_step + set-up of the next
            nset += 1
            phase[need] = 1
            it[need] = 0
            solving = phase == 1
        # one loop trip: rows pass + stop test for the solving groups
        nloop += 1
        tot += c["R"]
        idx = np.nonzero(solving)[0]
        rows = d[idx, sub[idx], 0]
        trips = d[idx, sub[idx], 1]
        cont = trips > it[idx]  # goes on into factor + line search
        if cont.any():
            ev = d[idx[cont], sub[idx[cont]], 2 + np.minimum(it[idx[cont]], 11)]
            tot += c["F"] + c["E"] * ev.max()
        it[idx] += 1
        # finished: stop test said so (rows == it after this pass) or exact termination in the line search
        # (rows == trips: no further pass over the rows)
        fin = (rows <= it[idx])
        f = idx[fin]
        sub[f] += 1
        phase[f] = np.where(sub[f] >= 5, 2, 0)
        sub[f] = np.minimum(sub[f], 4)
    tot += c["I"]  # the last integration
    return tot, nset, nloop


def main():
    d = rollout()
    steps, envs = d.shape[:2]
    print("per env and env-step: rows passes %.2f, trips %.2f, evals %.2f" % (
        d[..., 0].sum(-1).mean(), d[..., 1].sum(-1).mean(), d[..., 2:].sum((-1, -2)).mean()))
    waves = envs // 32
    sync, des = [], {th: [] for th in (1, 2, 4, 8, 12, 16, 24, 32)}
    wave_trips, own_trips = [], []
    for t in range(steps):
        for w in range(waves):
            dd = d[t, 32 * w:32 * w + 32]
            sync.append(cost_sync(dd))
            wave_trips.append(sum(dd[:, s, 1].max() for s in range(5)))
            own_trips.append(dd[:, :, 1].sum(1).max())
            for th in des:
                des[th].append(cost_desync(dd, th))
    sync = np.array(sync)
    print("wave trips per env-step (sum of max) %.2f, max of own sums %.2f" % (np.mean(wave_trips), np.mean(own_trips)))
    print("shipped schedule: %.0f cycles per chunk" % sync.mean())
    for th, v in des.items():
        v = np.array(v)
        print("theta %2d: %.0f cycles (%.3f of shipped), set-ups %.2f, loop trips %.2f" % (
            th, v[:, 0].mean(), v[:, 0].mean() / sync.mean(), v[:, 1].mean(), v[:, 2].mean()))




def chunk_costs():
    d = rollout()
    steps, envs = d.shape[:2]
    for per in (32, 24, 16, 8):
        c = []
        for t in range(0, steps, 4):
            for w in range(envs // per):
                c.append(cost_sync(d[t, per * w:per * w + per]))
        c = np.array(c)
        print("per %2d: mean %.0f std %.0f p5 %.0f p95 %.0f max %.0f  (per env %.0f)" % (
            per, c.mean(), c.std(), np.percentile(c, 5), np.percentile(c, 95), c.max(), c.mean() / per))


if len(sys.argv) > 1 and sys.argv[1] == "chunks":
    chunk_costs()

if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def ls_variants():
    """the shipped schedule's cost per chunk under solver knobs compiled into the host rollout (-D flags)"""
    src = os.path.join(ROOT, "tools", "lg_desync", "rollout_host.cpp")
    # EPA_LG_LS_MAX: evaluations of a line search at most (24 = the exact search of rounds 1-4, 1 = the product since
    # round 5); EPA_LG_LS_RTOL: |phi'| <= rtol |phi'(0)| ends a search
    for flags in (["-DEPA_LG_LS_MAX=24"], ["-DEPA_LG_LS_MAX=3"], ["-DEPA_LG_LS_MAX=2"], ["-DEPA_LG_LS_MAX=1"],
                  ["-DEPA_LG_LS_MAX=24", "-DEPA_LG_LS_RTOL=1e-6"], ["-DEPA_LG_LS_MAX=24", "-DEPA_LG_LS_RTOL=1e-3"]):
        so = "/tmp/liblg_rollout_v%d.so" % abs(hash(tuple(flags)))  # (dlopen caches by path)
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", so] + flags, check=True)
        L = ctypes.CDLL(so)
        envs, warm, steps = 2048, 150, 24
        d = np.zeros((steps, envs, 5, 14), np.int16)
        L.lg_rollout(envs, warm, steps, 1, d.ctypes.data_as(ctypes.c_void_p))
        del L
        c = np.mean([cost_sync(d[t, 32 * w:32 * w + 32]) for t in range(steps) for w in range(envs // 32)])
        wt = np.mean([sum(d[t, 32 * w:32 * w + 32, s, 1].max() for s in range(5)) for t in range(steps) for w in range(envs // 32)])
        we = np.mean([sum(d[t, 32 * w:32 * w + 32, s, 2 + k].max() for s in range(5) for k in range(12))
                      for t in range(steps) for w in range(envs // 32)])
        print("%-44s per env: trips %.2f evals %.2f | wave: trips %.2f evals %.2f | model cycles per chunk %.0f" % (
            " ".join(flags), d[..., 1].sum(-1).mean(), d[..., 2:].sum((-1, -2)).mean(), wt, we, c), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "ls":
    ls_variants()
