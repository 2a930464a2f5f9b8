"""Copies what a measurement pass (tools/gpu_runs/gpu_r<N>z.sh; tag r4z, r5z, ...) left under gpurun_out/ into profiles/
(summaries, kernel stats, bench lines, logs) and prints the numbers DESIGN.md section 5 quotes:
per kernel the rocprof average duration, issued flops per env-step and fraction, algorithmic flops and useful
fraction, HBM traffic vs algorithmic bytes.  Run after `python tools/make_pmc_json.py <tag> <num_envs>` for every size.

    python tools/collect_pass.py r5z       # copy + print
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6z"


def copy(src, dst):
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))


def main():
    for d in sorted(glob.glob(os.path.join(G, f"prof_{TAG}_*"))):
        tag = os.path.basename(d)[len("prof_"):]
        copy(os.path.join(d, "summary.md"), f"{tag}_summary.md")
        copy(os.path.join(d, "trace", "t_kernel_stats.csv"), f"{tag}_kernel_stats.csv")
    copy(os.path.join(G, TAG, "bench_default.json"), f"{TAG}_bench_default.json")
    copy(os.path.join(G, TAG, "bench.jsonl"), f"{TAG}_bench.jsonl")
    copy(os.path.join(G, TAG, "numpy_api.jsonl"), f"{TAG}_numpy_api.jsonl")
    copy(os.path.join(G, TAG, "gpu_tests.log"), f"{TAG}_gpu_tests.log")
    copy(os.path.join(G, TAG, "bench_families.md"), f"{TAG}_bench_families.md")
    copy(os.path.join(G, TAG, "probe_gpu_box.log"), f"{TAG}_probe_gpu_box.log")
    pmc = json.load(open(os.path.join(P, "pmc.json")))
    alg = json.load(open(os.path.join(P, "flops_algorithmic.json")))
    alg_bytes = {"HalfCheetah": 708, "Ant": 1132, "Walker2d": 692, "Hopper": 476, "Humanoid": 4402,
                 "HumanoidStandup": 4362, "Pusher": 842}
    rows = [("PlanarLgStepKernel<2,1>@65536", "HalfCheetah"), ("PlanarLgStepKernel<4,1>@8192", "HalfCheetah"),
            ("PlanarLgStepKernel<2,1>[Walker2d]@65536", "Walker2d"), ("PlanarLgStepKernel<1,1>[Hopper]@65536", "Hopper"),
            ("CheetahStepKernel<double>[Hopper]@65536", "Hopper"), ("CheetahStepKernel<double>@65536", "HalfCheetah"),
            ("AntStepKernel<double>@32768", "Ant"), ("AntStepKernel<double>@65536", "Ant"),
            ("AntStepKernel<float>@65536", "Ant"), ("PusherStepKernel<double>@65536", "Pusher"),
            ("Humanoid4StepKernel<double>@65536", "Humanoid"), ("Humanoid4StepKernel<double>[Standup]@65536", "HumanoidStandup")]
    for key, task in rows:
        e = pmc.get(key)
        if not e or TAG not in e.get("source", ""):
            print(f"{key}: no {TAG} entry")
            continue
        n, us = e["num_envs"], e["rocprof_avg_us"]
        peak = 157.3e12 if "<float>" in key else 78.6e12
        iss = e["flops_per_env_step"]
        a = alg[task]["flops_per_env_step"]
        t = us * 1e-6
        print(f"{key}: {us:.1f} us | issued {iss:.3g} -> {iss * n / t / peak:.3f} | algorithmic {a:.3g} -> "
              f"{a * n / t / peak:.3f} | issued/alg {iss / a:.2f} | traffic {e['traffic_bytes_per_launch'] / 1e6:.1f} MB / "
              f"{alg_bytes[task] * n / 1e6:.1f} MB = {e['traffic_bytes_per_launch'] / (alg_bytes[task] * n):.2f}x | "
              f"wait {e['wait_frac_of_wave_cycles']:.2f} | env-steps/s at the rocprof duration {n / t:.3g}")
    for f in (f"{TAG}_bench_default.json", f"{TAG}_bench.jsonl"):
        path = os.path.join(P, f)
        if not os.path.exists(path):
            continue
        for line in open(path):
            if not line.startswith("{"):
                continue
            d = json.loads(line)
            extra = ""
            if "cpu_baseline" in d:
                c = d["cpu_baseline"]
                extra = f" | cpu threadpool {c['value']:.3g} openmp {c.get('openmp_port', {}).get('value', 0):.3g} on {c['cores']}"
            print(f"{d['metric']} {d['dtype']} {d['config']['params']}: {d['value']:.4g} kernel_ms {d['roofline']['kernel_ms']:.4f}"
                  f" async {(d.get('async_mode') or {}).get('value', 0):.4g} numpy {(d.get('numpy_api') or {}).get('value', 0):.4g}{extra}")


if __name__ == "__main__":
    main()
