"""Timeline of the host path's sync step from a rocprofv3 kernel + memory-copy trace (no counters):
    cd /tmp; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o t -- python tools/numpy_step_ab.py HalfCheetah 65536 32768 6
    python tools/numpy_step_timeline.py DIR
prints, for a few steady-state steps, when each kernel / copy of the step started and ended relative to the step's first activity."""
import csv
import glob
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# steps: split at gaps; take a window in the middle of the run
mid = len(ev) * 2 // 3
t0 = None
shown = 0
for s, e, name in ev[mid:mid + 60]:
    if t0 is None or s - last_end > 30000 and ("Step" in name or "H2D" in name.upper() or "HOST_TO" in name.upper()):
        t0 = s
        shown += 1
        if shown > 4:
            break
        print("--- step")
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  {name}")
    last_end = e
