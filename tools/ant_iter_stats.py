"""Newton-iteration / contact-class statistics of the quad-layout Ant kernel (diagnostic, GPU).
state slot `time` = own Newton iterations + 1e4 * (iterations the wave executed
+ 1e3 * sphere classes the wave visited), all summed over the 20 forward passes of a step."""
import sys
import numpy as np
sys.path.insert(0, ".")
from envpool_amd.core.device_pool import DevicePool

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
pool = DevicePool("Ant", n, seed=0, max_episode_steps=1000, params={"precision": 1})  # needs the EPA_WAVE_TRACE build for durations
ids = np.arange(n, dtype=np.int32)
pool.reset(ids); pool.recv()
rng = np.random.default_rng(1234)
hist = []
for t in range(70):
    act = rng.uniform(-1, 1, size=(n, 8))
    pool.send(ids, act); out = pool.recv_dict()
    if t >= 30:
        c = pool.get_state()[:, 43]
        kc = np.floor(c / 1e10)
        c = c - 1e10 * kc
        w = np.floor(c / 1e4)
        it_env = c - 1e4 * w
        cls = np.floor(w / 1e3)
        it_wave = w - 1e3 * cls
        hist.append((it_env, it_wave, cls, out["elapsed_step"].ravel().copy(), kc))
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

kc = np.array([h[4] for h in hist]).reshape(len(hist), -1, 16)[:, :, 0]  # kilo-clocks per wave
print(f"wave duration (kilo core clocks): mean {kc.mean():.0f} p10 {np.percentile(kc,10):.0f} p50 {np.median(kc):.0f} "
      f"p90 {np.percentile(kc,90):.0f} p99 {np.percentile(kc,99):.0f} max {kc.max():.0f}")
print(f"corr(duration, iterations executed) {np.corrcoef(kc.ravel(), wv.ravel())[0,1]:.3f}; "
      f"corr(duration, classes visited) {np.corrcoef(kc.ravel(), cv.ravel())[0,1]:.3f}")
A = np.stack([np.ones(kc.size), wv.ravel(), cv.ravel()], 1)
coef, *_ = np.linalg.lstsq(A, kc.ravel(), rcond=None)
print(f"fit: kclocks = {coef[0]:.0f} + {coef[1]:.2f} * iterations + {coef[2]:.2f} * class visits; residual rms {np.std(kc.ravel() - A @ coef):.0f}")
print(f"wave persistence: corr(duration[t], duration[t+1]) {np.corrcoef(kc[:-1].ravel(), kc[1:].ravel())[0,1]:.3f}; "
      f"corr(classes[t], classes[t+1]) {np.corrcoef(cv[:-1].ravel(), cv[1:].ravel())[0,1]:.3f}")
tot = kc.sum(axis=1).mean()
print(f"sum of wave durations / 1024 SIMDs = {tot/1024:.0f} kclocks; longest wave {kc.max(axis=1).mean():.0f} kclocks")
