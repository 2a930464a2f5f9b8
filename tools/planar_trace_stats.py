"""Per-wave durations of the last planar (HalfCheetah / Walker2d / Hopper) launch
(EPA_PLANAR_TRACE dump).  usage: python tools/planar_trace_stats.py <file>"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 6)
t = t[t[:, 1] > 0]
w0, w1, c0, c1, it, hw = t.T
dur = (w1 - w0) / 100.0
start = (w0 - w0.min()) / 100.0
print(f"waves {len(t)}  launch span {(w1.max() - w0.min()) / 100.0:.0f} us  start spread p99 {np.percentile(start, 99):.1f} us")
print(f"wave duration us: mean {dur.mean():.1f} p10 {np.percentile(dur,10):.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f}")
print(f"effective core clock {((c1 - c0).sum() / (w1 - w0).sum()) * 100:.0f} MHz")
print(f"Newton iterations executed per wave and env-step: mean {it.mean():.1f} p50 {np.median(it):.0f} p99 {np.percentile(it,99):.0f} max {it.max()}")
A = np.stack([np.ones(len(t)), it], 1)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
print(f"fit: duration = {coef[0]:.1f} us + {coef[1]:.2f} us x iterations; corr {np.corrcoef(dur, it)[0,1]:.3f}; residual rms {np.std(dur - A @ coef):.1f} us")
