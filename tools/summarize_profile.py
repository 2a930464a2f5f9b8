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
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if any(k in name for k in KEY):
            agg[short(name)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, ctrs in agg.items():
    print(f"## PMC per launch (mean over launches): {k}\n")
    print("| counter | mean | launches |")
    print("|---|---|---|")
    for c in sorted(ctrs):
        v = ctrs[c]
        # drop the first launch (reset path) from the mean when there are many
        vv = v[1:] if len(v) > 5 else v
        print(f"| {c} | {sum(vv)/len(vv):.4g} | {len(v)} |")
    print()
