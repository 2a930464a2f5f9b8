"""Summarise tools/profile_families.sh: rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes over
tools/bench_families.py for the HBM-streaming kernels (K1 ClassicStepKernel, K2 ToyStepKernel, K4
AtariPostKernel).  One process dispatches several configurations of the same kernel (N = 65536, then 4 M):
the dispatches of a kernel are split in order by the plan bench_families.py wrote (--plan-out).

HBM bytes per launch = 1024 x (2 x FETCH_SIZE + WRITE_SIZE): both counters are in KB, and on gfx950 FETCH_SIZE
counts half the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM section).
usage: python tools/summarize_families.py <dir> [--json out.json]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
KERNEL_OF = {
    "CartPole": "ClassicStepKernel<0>", "Pendulum": "ClassicStepKernel<1>", "MountainCar": "ClassicStepKernel<2>",
    "MountainCarContinuous": "ClassicStepKernel<3>", "Acrobot": "ClassicStepKernel<4>",
    "Catch": "ToyStepKernel<0>", "FrozenLake": "ToyStepKernel<1>", "Taxi": "ToyStepKernel<2>",
    "NChain": "ToyStepKernel<3>", "CliffWalking": "ToyStepKernel<4>", "Blackjack": "ToyStepKernel<5>",
    "AtariPostProcess": "AtariPostKernel",
}


def tag(name):
    for k in ("ClassicStepKernel<", "ToyStepKernel<"):
        i = name.find(k)
        if i >= 0:
            return name[i:name.index(">", i) + 1]
    if "AtariPostKernel" in name:
        return "AtariPostKernel"
    return None


plan = json.load(open(os.path.join(out, "plan.json")))


def split(seq):
    """{kernel tag: ordered per-dispatch values} -> {(family, N): values of the timed launches}"""
    pos = defaultdict(int)
    res = {}
    for p in plan:
        k = KERNEL_OF[p["family"]]
        v = seq.get(k, [])
        a = pos[k] + p["skip"]
        res[(p["family"], p["num_envs"])] = v[a:a + p["timed"]]
        pos[k] = a + p["timed"]
    return res


dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = []
    for r in csv.DictReader(open(f)):
        t = tag(r.get("Kernel_Name", ""))
        if t:
            rows.append((int(r["Start_Timestamp"]), t, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for _, t, d in sorted(rows):
        dur[t].append(d)
ctr = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    seq = defaultdict(list)
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        rows = []
        for r in csv.DictReader(open(f)):
            t = tag(r.get("Kernel_Name", ""))
            if t and r["Counter_Name"] == c:
                rows.append((int(r.get("Dispatch_Id", 0)), t, float(r["Counter_Value"])))
        for _, t, v in sorted(rows):
            seq[t].append(v)
    ctr[c] = split(seq)
durs = split(dur)

print(f"# rocprofv3 summary of the HBM-streaming kernels: {out}\n")
print("kernel trace and each PMC counter are separate rocprofv3 runs of the same command "
      "(`tools/profile_families.sh`); per-launch means over the timed launches of each configuration.\n")
print("| family | kernel | N | launches | rocprof avg us (min, max) | HIP-event us | algorithmic MB | FETCH_SIZE KB | "
      "WRITE_SIZE KB | HBM MB = 2 x fetch + write | traffic / algorithmic | algorithmic GB/s (frac of 8 TB/s) | "
      "HBM GB/s |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
res = {}
for p in plan:
    key = (p["family"], p["num_envs"])
    d = durs.get(key, [])
    if not d:
        continue
    us = sum(d) / len(d) / 1e3
    alg = p["algorithmic_bytes"] * p["num_envs"]
    fe = ctr["FETCH_SIZE"].get(key, [])
    wr = ctr["WRITE_SIZE"].get(key, [])
    fe_kb = sum(fe) / len(fe) if fe else None
    wr_kb = sum(wr) / len(wr) if wr else None
    hbm = 1024.0 * (2 * fe_kb + wr_kb) if fe and wr else None
    gbs = alg / (us * 1e-6) / 1e9
    print(f"| {p['family']} | {KERNEL_OF[p['family']]} | {p['num_envs']} | {len(d)} | {us:.1f} ({min(d)/1e3:.1f}, "
          f"{max(d)/1e3:.1f}) | {p['hip_event_us']:.1f} | {alg/1e6:.2f} | "
          f"{'%.0f' % fe_kb if fe else '-'} | {'%.0f' % wr_kb if wr else '-'} | "
          f"{'%.2f' % (hbm/1e6) if hbm else '-'} | {'%.2f' % (hbm/alg) if hbm else '-'} | "
          f"{gbs:.0f} ({gbs/8000:.3f}) | {'%.0f' % (hbm/(us*1e-6)/1e9) if hbm else '-'} |")
    res[f"{KERNEL_OF[p['family']]}[{p['family']}]@{p['num_envs']}"] = {
        "rocprof_avg_us": us, "hip_event_us": p["hip_event_us"], "launches": len(d),
        "algorithmic_bytes_per_launch": alg, "fetch_size_kb": fe_kb, "write_size_kb": wr_kb,
        "traffic_bytes_per_launch": hbm, "traffic_over_algorithmic": hbm / alg if hbm else None,
        "algorithmic_GBps": gbs, "hbm_frac": gbs / 8000.0, "num_envs": p["num_envs"],
        "note": "gfx950: FETCH_SIZE counts half the bytes of a coalesced stream -> x2"}
if json_out:
    json.dump(res, open(json_out, "w"), indent=1)
