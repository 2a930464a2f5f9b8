"""Per-wave schedule of the last Ant launch (EPA_ANT_TRACE dump): durations, effective clock,
per-SIMD occupancy timeline.  usage: python tools/ant_trace_stats.py <file>"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 6)
t = t[t[:, 1] > 0]
# the buffer keeps stale entries of waves whose first quad was resetting in the last launch:
# keep the last launch only (entries after the last gap of > 100 us between wave starts)
o = np.argsort(t[:, 0])
t = t[o]
gaps = np.where(np.diff(t[:, 0]) > 100 * 100)[0]
if len(gaps):
    t = t[gaps[-1] + 1:]
w0, w1, c0, c1, slot, hw = t.T
span = (w1.max() - w0.min()) / 100.0  # us (100 MHz wall clock)
dur = (w1 - w0) / 100.0
print(f"waves {len(t)}  launch span {span:.0f} us  wave duration us: mean {dur.mean():.0f} p50 {np.median(dur):.0f} p90 {np.percentile(dur,90):.0f} max {dur.max():.0f}")
print(f"effective core clock: {((c1 - c0).sum() / (w1 - w0).sum()) * 100:.0f} MHz")
print(f"sum of durations / 1024 SIMDs = {dur.sum()/1024:.0f} us")
# HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ... ; XCC from slot order unknown
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((se * 2 + sh) * 16 + cu) * 4 + simd
start = (w0 - w0.min()) / 100.0
print("distinct (se,sh,cu,simd) ids:", len(np.unique(key)), " (several XCDs share ids)")
order = np.argsort(w0)
print("first 8 started slots:", slot[order][:8], " last 8 started:", slot[order][-8:])
late = start > 0.25 * span
print(f"waves starting after 25% of the span: {late.sum()} ; their mean duration {dur[late].mean():.0f} us; early mean {dur[~late].mean():.0f} us")
idle_tail = (w1.max() - w1) / 100.0
print(f"time between a wave's end and launch end: mean {idle_tail.mean():.0f} us (of waves ending last on their SIMD this is idle time)")
# concurrency profile
ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([start + dur, -np.ones_like(start)], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
for frac in (0.1, 0.25, 0.5, 0.75, 0.9):
    i = np.searchsorted(ev[:, 0], frac * span)
    print(f"  waves in flight at {int(frac*100)}% of the span: {int(conc[min(i, len(conc)-1)])}")

# greedy list schedule of the same durations on 1024 SIMDs in slot order, and in LPT order
import heapq
def sched(d):
    h = [0.0] * 1024
    heapq.heapify(h)
    for x in d:
        heapq.heappush(h, heapq.heappop(h) + x)
    return max(h)
by_slot = dur[np.argsort(slot)]
print(f"list schedule on 1024 SIMDs: in slot order {sched(by_slot):.0f} us, longest-first {sched(np.sort(dur)[::-1]):.0f} us")
# start times by slot rank: is dispatch in slot order?
r = np.argsort(np.argsort(slot))
print("corr(start time, slot rank) =", round(float(np.corrcoef(start, r)[0, 1]), 3))
q = np.percentile(start, [0, 10, 25, 50, 75, 90, 100])
print("start time percentiles (us):", np.round(q))
for lo in range(0, len(t), 256):
    sel = (r >= lo) & (r < lo + 256)
    print(f"  slots rank {lo:5d}..{lo+255:5d}: start mean {start[sel].mean():6.0f} us, duration mean {dur[sel].mean():5.0f} us")
