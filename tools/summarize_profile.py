"""Summarise tools/profile_bench.sh output (kernel-trace stats + PMC passes)
for the dominant step kernel into a small markdown table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KEY = ("StepKernel", "AtariPostKernel")


def short(name):
    for k, model in (("CheetahStepKernel<float, 0>", ""), ("CheetahStepKernel<double, 0>", ""),
                     ("CheetahStepKernel<float, 1>", "[Walker2d]"),
                     ("CheetahStepKernel<double, 1>", "[Walker2d]"),
                     ("CheetahStepKernel<float, 2>", "[Walker2d-v5]"),
                     ("CheetahStepKernel<double, 2>", "[Walker2d-v5]"),
                     ("CheetahStepKernel<float, 3>", "[Hopper]"),
                     ("CheetahStepKernel<double, 3>", "[Hopper]")):
        if k in name:  # planar kernel, second template argument = PlanarModelId
            return k.split(",")[0] + ">" + model
    for k, v in (("AntStepKernel<float, false>", "AntStepKernel<float>"),
                 ("AntStepKernel<double, false>", "AntStepKernel<double>"),
                 ("AntStepKernel<float, true>", "AntStepKernel<float>[v5 cfrc_ext]"),
                 ("AntStepKernel<double, true>", "AntStepKernel<double>[v5 cfrc_ext]")):
        if k in name:
            return v
    if "Humanoid4StepKernel" in name:  # one env per lane quad (mujoco_humanoid4.hip)
        return "Humanoid4StepKernel<double>" + ("[Standup]" if "StandupMP" in name else "")
    if "HumanoidStepKernel" in name:
        return "HumanoidStepKernel<double>" + ("[Standup]" if "StandupMP" in name else "")
    if "PusherStepKernel" in name:
        return "PusherStepKernel<double>"
    for k in ("PendStepKernel", "ReacherStepKernel", "SwimmerStepKernel", "ClassicStepKernel",
              "ToyStepKernel", "AtariPostKernel"):
        if k in name:
            return k
    return name[:60]


print(f"# rocprofv3 summary: {out}\n")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel-trace --stats (dominant kernels)\n")
    print("| kernel | calls | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in KEY):
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | "
                  f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {r['Percentage']} |")
    print()
# the bench's timed window inside the trace: bench.py launches 1 reset + W warm-up + K timed steps (then
# reset / numpy-API legs unless --only-timed); tools/profile_bench.sh runs it with --steps 200 --warmup 300, so that the
# window holds the de-synchronised steady state the bench line measures (right after the common reset every env of the
# episodic tasks is upright and cheap: Hopper 0.22 ms per launch over the first 200 steps, 0.35 ms in steady state)
W, K = int(os.environ.get("EPA_PROF_WARMUP", 300)), int(os.environ.get("EPA_PROF_STEPS", 200))
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if any(k in name for k in KEY) and "GetState" not in name:
            rows[short(name)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for k, v in rows.items():
        v.sort()
        if len(v) < 1 + W + K:
            continue
        win = v[1 + W:1 + W + K]
        dur = [e - b for b, e in win]
        gap = [win[i + 1][0] - win[i][1] for i in range(len(win) - 1)]
        print(f"## timed window of the bench inside this trace: {k}\n")
        print(f"launches {1 + W + 1}..{1 + W + K} of {len(v)}: avg {sum(dur)/len(dur)/1e3:.1f} us "
              f"(min {min(dur)/1e3:.1f}, max {max(dur)/1e3:.1f}); gap to the next launch avg "
              f"{sum(gap)/len(gap)/1e3:.1f} us; (last end - first start) / {K} = "
              f"{(win[-1][1] - win[0][0])/K/1e3:.1f} us\n")
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if any(k in name for k in KEY):
            agg[short(name)][r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
for k, ctrs in agg.items():
    print(f"## PMC per launch (mean over launches): {k}\n")
    print("| counter | mean | launches |")
    print("|---|---|---|")
    for c in sorted(ctrs):
        v = [x for _, x in sorted(ctrs[c])]  # dispatch order
        # the bench's timed window (the launches after the reset + warm-up ones) when the run has it,
        # else everything but the first launch (reset path)
        vv = v[1 + W:1 + W + K] if len(v) >= 1 + W + K else (v[1:] if len(v) > 5 else v)
        print(f"| {c} | {sum(vv)/len(vv):.4g} | {len(vv)} |")
    print()
