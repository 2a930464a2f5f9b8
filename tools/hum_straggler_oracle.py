"""HOST EXPERIMENT behind DESIGN K3d "stragglers": how many envs would a "park the env whose PGS needs more than S
sweeps" rule move out of the wave?  Per-env sweep counts of the LAST forward pass of an env-step (one of 20: RK4 x
frame_skip 5) from oracle/mjcpu, random actions as in bench.py, steady state (steps 100..).  A straggler list only
pays if P(sweeps > S) is small for an S well under the cap of 50; an env is parked for the rest of its env-step by
ANY of its 20 forward passes, so the parked share is 1 - (1 - p)^20 if passes were independent and >= p if the same
envs straggle in every pass (lag-1 agreement printed).
usage: tools/hum_straggler_oracle.py [HumanoidStandup|Humanoid] [envs] [steps]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, '/root/repo')
sys.path.insert(0, '/root/repo/tests')
from mj_util import _H  # noqa: E402
from oracle.orc import Oracle  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "HumanoidStandup"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
o = Oracle(task, N, seed=3, max_episode_steps=1000)
L = o.lib
inner = ctypes.cast(o.h, ctypes.POINTER(_H)).contents.h
vp = ctypes.c_void_p
L.mjcpu_raw_get.argtypes = [vp, ctypes.c_int, vp, vp, vp]
sc = np.zeros(256)
L.mjcpu_model_scalars.argtypes = [vp, vp]
L.mjcpu_model_scalars(inner, sc.ctypes.data)
nq, nv = int(sc[0]), int(sc[1])
o.reset()
rng = np.random.default_rng(0)
its = []
for t in range(T):
    o.step(rng.uniform(-0.4, 0.4, size=(N, 17)))
    if t >= 100:
        i = []
        for e in range(N):
            q, v, m = np.zeros(nq), np.zeros(nv), np.zeros(16)
            L.mjcpu_raw_get(inner, e, q.ctypes.data, v.ctypes.data, m.ctypes.data)
            i.append(int(m[4]))
        its.append(i)
its = np.array(its)  # [steps, envs]
print(task, "envs", N, "env-steps", its.shape[0], ": sweeps of the last forward pass: mean %.1f median %d p75 %d p90 %d; at the cap %.3f"
      % (its.mean(), np.median(its), np.percentile(its, 75), np.percentile(its, 90), (its >= 50).mean()))
for S in (2, 4, 8, 12, 16, 24, 32):
    p = (its > S).mean()
    a, b = its[:-1] > S, its[1:] > S
    stay = (a & b).sum() / max(1, a.sum())
    print("S = %2d: P(sweeps > S) = %.3f; still > S one env-step later %.2f; parked share of an env-step: "
          "%.2f (independent passes) .. >= %.2f (same envs every pass); mean sweeps of a wave of 16 capped at S: %.1f"
          % (S, p, stay, 1 - (1 - p) ** 20, p, np.minimum(its, S).reshape(its.shape[0], -1, 16).max(-1).mean()))
print("mean over waves of 16 envs of the wave's max: %.1f (sorted by the previous step's count: %.1f)" % (
    its.reshape(its.shape[0], -1, 16).max(-1).mean(),
    np.mean([np.take_along_axis(its[t], np.argsort(-its[t - 1], kind="stable"), 0).reshape(-1, 16).max(-1).mean()
             for t in range(1, its.shape[0])])))
