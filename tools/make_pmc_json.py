"""Collects the PMC passes of tools/profile_bench.sh (gpurun_out/prof_<tag>/summary.md)
into profiles/pmc.json, which bench.py reads for `roofline.traffic` and the
VALU figures.  usage: python tools/make_pmc_json.py <round-tag-prefix> [num_envs]

Conventions (MI355X_MICROARCH.md, HBM / rocprofv3 section):
  FETCH_SIZE, WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts half the bytes of a
  coalesced stream -> x2.  flops = 64 lanes x (2 FMA + ADD + MUL + TRANS) wave-level
  instructions of the arithmetic type (exec mask ignored -> an upper bound).
"""
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_sources import source_hash  # noqa: E402

prefix = sys.argv[1] if len(sys.argv) > 1 else "r1f"
num_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
# a prefix that names one pass exactly means that pass only (prof_r5z_cheetah_lg2 must not pull in ..._lg2_32k)
exact = os.path.join(root, "gpurun_out", f"prof_{prefix}", "summary.md")
paths = [exact] if os.path.exists(exact) else sorted(glob.glob(os.path.join(root, "gpurun_out", f"prof_{prefix}*", "summary.md")))
for path in paths:
    text = open(path).read()
    # the kernel of the bench's timed window (since round 6 a trace also holds the few launches of OTHER kernel
    # variants that the library's load-time self-test makes), and ITS counter section only
    w = re.search(r"## timed window of the bench inside this trace: (.+)", text)
    sections = re.split(r"## PMC per launch \(mean over launches\): ", text)[1:]
    if not w or not sections:
        continue
    kernel = w.group(1).strip()
    mine = [sec for sec in sections if sec.split("\n", 1)[0].strip() == kernel]
    if not mine:
        continue
    ctr = {k: float(v) for k, v in re.findall(r"\| (\w+) \| ([0-9.e+-]+) \| \d+ \|", mine[0])}
    tr = re.search(r"\| %s \| (\d+) \| ([0-9.]+) \|" % re.escape(kernel), text)
    # the bench's timed window inside the trace, when the summary has it (steady state; the all-launch average of
    # --stats includes the cheap steps right after the common reset)
    win = re.search(r"launches \d+\.\.\d+ of \d+: avg ([0-9.]+) us", text)
    lg = re.search(r"PlanarLgStepKernel<(\d), (\d), (\d)>", kernel)  # <lanes per env, model, waves per SIMD>
    if lg:  # canonical name (bench.py builds the same): all lane-group kernels are fp64
        model = {"0": "", "1": "[Walker2d]", "2": "[Walker2d-v5]", "3": "[Hopper]"}[lg.group(2)]
        kernel = f"PlanarLgStepKernel<{lg.group(1)},{lg.group(3)}>{model}"
    f64 = "<double>" in kernel or lg is not None
    sfx = "F64" if f64 else "F32"
    flops = 64.0 * (2 * ctr.get(f"SQ_INSTS_VALU_FMA_{sfx}", 0) + ctr.get(f"SQ_INSTS_VALU_ADD_{sfx}", 0) +
                    ctr.get(f"SQ_INSTS_VALU_MUL_{sfx}", 0) + ctr.get(f"SQ_INSTS_VALU_TRANS_{sfx}", 0))
    quad = "Ant" in kernel or "Humanoid4" in kernel  # one env per lane quad: 16 envs per wave
    per_wave = 64 // int(lg.group(1)) if lg else (16 if quad else 64)
    waves = (num_envs + per_wave - 1) // per_wave
    out[f"{kernel}@{num_envs}"] = {
        "fetch_size_kb": ctr.get("FETCH_SIZE"),
        "write_size_kb": ctr.get("WRITE_SIZE"),
        "traffic_bytes_per_launch": 1024.0 * (2 * ctr.get("FETCH_SIZE", 0) + ctr.get("WRITE_SIZE", 0)),
        "flops_per_launch": flops,
        "flops_per_env_step": flops / num_envs,
        # wave-level arithmetic instructions of the kernel's type (an FMA counts once): x 4 cycles each = the issue
        # slots they occupy on a SIMD (bench.py: roofline.fp64_issue_slot_util)
        "arith_wave_insts_per_launch": (ctr.get(f"SQ_INSTS_VALU_FMA_{sfx}", 0) + ctr.get(f"SQ_INSTS_VALU_ADD_{sfx}", 0) +
                                        ctr.get(f"SQ_INSTS_VALU_MUL_{sfx}", 0) + ctr.get(f"SQ_INSTS_VALU_TRANS_{sfx}", 0)),
        "valu_wave_insts_per_launch": ctr.get("SQ_INSTS_VALU", 0),
        "valu_insts_per_wave": ctr.get("SQ_INSTS_VALU", 0) / waves,
        "lds_insts_per_wave": ctr.get("SQ_INSTS_LDS", 0) / waves,
        "wait_frac_of_wave_cycles": ctr.get("SQ_WAIT_ANY", 0) / max(ctr.get("SQ_WAVE_CYCLES", 1), 1),
        "rocprof_avg_us": float(win.group(1)) if win else (float(tr.group(2)) if tr else None),
        "rocprof_avg_us_all_launches": float(tr.group(2)) if tr else None,
        "note": "gfx950: FETCH_SIZE counts half the bytes of a coalesced stream -> x2",
        "source": f"profiles/{os.path.basename(os.path.dirname(path)).replace('prof_', '')}_summary.md",
        "num_envs": num_envs,
        "src_hash": source_hash(kernel),  # the kernel's sources + Makefile at the time of the profile
    }
dst = os.path.join(root, "profiles", "pmc.json")
try:
    merged = json.load(open(dst))
except OSError:
    merged = {}
merged.update(out)  # entries of other kernels / configurations stay
json.dump(merged, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
