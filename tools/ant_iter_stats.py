"""Newton-iteration / contact-class statistics of the quad-layout Ant kernel (diagnostic, GPU).
state slot `time` = own Newton iterations + 1e4 * (iterations the wave executed
+ 1e3 * sphere classes the wave visited), all summed over the 20 forward passes of a step."""
import sys
import numpy as np
sys.path.insert(0, ".")
from envpool_amd.core.device_pool import DevicePool

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
pool = DevicePool("Ant", n, seed=0, max_episode_steps=1000, params={"precision": 1})
ids = np.arange(n, dtype=np.int32)
pool.reset(ids); pool.recv()
rng = np.random.default_rng(1234)
hist = []
for t in range(70):
    act = rng.uniform(-1, 1, size=(n, 8))
    pool.send(ids, act); out = pool.recv_dict()
    if t >= 30:
        c = pool.get_state()[:, 43]
        w = np.floor(c / 1e4)
        it_env = c - 1e4 * w
        cls = np.floor(w / 1e3)
        it_wave = w - 1e3 * cls
        hist.append((it_env, it_wave, cls, out["elapsed_step"].ravel().copy()))
it_env = np.array([h[0] for h in hist]); it_wave = np.array([h[1] for h in hist]); cls = np.array([h[2] for h in hist])
el = np.array([h[3] for h in hist])
live = el > 0
print(f"envs {n}; per env-step (20 forward passes): own Newton iterations mean {it_env[live].mean():.1f} "
      f"p50 {np.median(it_env[live]):.0f} p90 {np.percentile(it_env[live],90):.0f} p99 {np.percentile(it_env[live],99):.0f} max {it_env.max():.0f}")
wv = it_wave.reshape(len(hist), -1, 16)[:, :, 0]; cv = cls.reshape(len(hist), -1, 16)[:, :, 0]
print(f"wave-level: iterations executed mean {wv.mean():.1f} p50 {np.median(wv):.0f} p99 {np.percentile(wv,99):.0f} max {wv.max():.0f}; "
      f"sphere classes visited per pass mean {cv.mean()/20:.2f} p99 {np.percentile(cv,99)/20:.2f} max {cv.max()/20:.2f}")
print(f"resets per step: {(~live).mean():.4f}")
# persistence and what grouping by last step's count would buy (per-step totals: a lower bound)
a, b = it_env[:-1], it_env[1:]
cc = np.corrcoef(a.ravel(), b.ravel())[0, 1]
now = b.reshape(len(b), -1, 16).max(axis=2).mean()
order = np.argsort(a, axis=1, kind="stable")
srt = np.take_along_axis(b, order, axis=1).reshape(len(b), -1, 16).max(axis=2)
print(f"corr(iters[t], iters[t+1]) = {cc:.3f}; mean over waves of max-over-16-envs: now {now:.1f}, "
      f"grouped by last step's count {srt.mean():.1f} (max wave {srt.max():.0f}), env mean {b.mean():.1f}")
