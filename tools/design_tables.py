"""Prints the two measurement tables of DESIGN.md section 5 from the tracked evidence: profiles/pmc.json (rocprofv3 window
duration, issued flops, HBM traffic), profiles/flops_algorithmic.json and the bench lines of a pass (profiles/<tag>_bench*.json*).
    python tools/design_tables.py r5z"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6z"
ALG_BYTES = {"HalfCheetah": 708, "Ant": 1132, "Walker2d": 692, "Hopper": 476, "Humanoid": 4402, "HumanoidStandup": 4362,
             "Pusher": 842}
ROWS = [
    ("PlanarLgStepKernel<2,1>@65536", "HalfCheetah", "`PlanarLgStepKernel<2>` HalfCheetah @65536 (headline)",
     "planar + arrow structure (fewer), × 14.4 executed Newton trips where an env needs 6.7, torso work replicated on 2 lanes (more)"),
    ("PlanarLgStepKernel<4,1>@8192", "HalfCheetah", "`PlanarLgStepKernel<4>` HalfCheetah @8192 (config 3)",
     "4 lanes per env replicate the leg state, half-filled waves count as full"),
    ("PlanarLgStepKernel<2,1>[Walker2d]@65536", "Walker2d", "`PlanarLgStepKernel<2>` Walker2d @65536", "as HalfCheetah"),
    ("PlanarLgStepKernel<1,1>[Hopper]@65536", "Hopper", "`PlanarLgStepKernel<1>` Hopper @65536",
     "no replication at all (one lane = the robot), planar 6-dof form; max over 64 envs per trip"),
    ("AntStepKernel<double>@32768", "Ant", "`AntStepKernel<double>` @32768 (config-4 shard)",
     "3.2 executed trips per pass vs 1.7, torso replicated ×4; traffic: the chunk's state goes through HBM between the 5 units of its env-step"),
    ("PusherStepKernel<double>@65536", "Pusher", "`PusherStepKernel` @65536", "welded bodies merged, rows built only when touching"),
    ("Humanoid4StepKernel<double>@65536", "Humanoid", "`Humanoid4StepKernel` @65536", "trunk replicated ×4, wave-level sweep counts"),
    ("Humanoid4StepKernel<double>[Standup]@65536", "HumanoidStandup", "`Humanoid4StepKernel[Standup]` @65536",
     "48 sweeps per wave where the mean env needs 18 (K3d)"),
]


def main():
    pmc = json.load(open(os.path.join(P, "pmc.json")))
    alg = json.load(open(os.path.join(P, "flops_algorithmic.json")))
    try:
        nw = json.load(open(os.path.join(P, "necessary_work.json")))
    except OSError:
        nw = {}
    print("| kernel @N | µs | issued flops / env-step → frac | → frac_necessary | fp64 issue slots busy | algorithmic → frac_useful | issued / algorithmic, why | traffic / algorithmic bytes |")
    print("|---|---|---|---|---|---|---|---|")
    for key, task, label, why in ROWS:
        e = pmc[key]
        n, us = e["num_envs"], e["rocprof_avg_us"]
        t = us * 1e-6
        iss, a = e["flops_per_env_step"], alg[task]["flops_per_env_step"]
        tr, ab = e["traffic_bytes_per_launch"], ALG_BYTES[task] * n
        trs = f"{tr / 1e9:.1f} GB / {ab / 1e6:.0f} MB = **{tr / ab:.0f}×**" if tr > 1e9 else f"{tr / 1e6:.1f} / {ab / 1e6:.1f} MB = {tr / ab:.2f}×"
        frac = iss * n / t / 78.6e12
        w = nw.get(key)
        nec = f"{frac * ((1 - w['solver_share']) + w['solver_share'] * w['trips_needed'] / w['trips_executed']):.3f}" if w else "—"
        slots = f"{e['arith_wave_insts_per_launch'] * 4 / (1024 * t * 2.4e9):.2f}" if e.get("arith_wave_insts_per_launch") else "—"
        print(f"| {label} | {us:.1f} | {iss:.3g} → {frac:.3f} | {nec} | {slots} | {a:.3g} → {a * n / t / 78.6e12:.3f} | {iss / a:.2f}: {why} | {trs} |")
    print()
    lines = {}
    d = json.load(open(os.path.join(P, f"{TAG}_bench_default.json")))
    lines[("HalfCheetah", 65536, "")] = d
    for l in open(os.path.join(P, f"{TAG}_bench.jsonl")):
        x = json.loads(l)
        task = x["metric"].split(",")[-1].strip().split("-")[0]
        par = x["config"]["params"]
        sfx = "K3" if par.get("planar_layout") == 1.0 else ("fp32" if x["dtype"] == "f32" else "")
        lines[(task, x["config"]["num_envs_per_gpu"], sfx)] = x

    def v(task, n, sfx=""):
        x = lines[(task, n, sfx)]
        return f"{x['value']:.3g} ({x['roofline']['kernel_ms']:.3f})"

    def asy(task, n):
        return f"{lines[(task, n, '')]['async_mode']['value']:.3g}"

    h = lines[("HalfCheetah", 65536, "")]
    print("| Configuration | env-steps/s (kernel ms per launch) |")
    print("|---|---|")
    print(f"| **HalfCheetah-v4 N=65536** (headline, sync `step()`) | **{h['value']:.3g}** ({h['roofline']['kernel_ms']:.3f}) |")
    print(f"| same, async: two 32768-env batches in flight (`async_mode`) | **{asy('HalfCheetah', 65536)}** |")
    print(f"| same, numpy API (PCIe inclusive, `numpy_api` of the line) | {h['numpy_api']['value']:.3g} |")
    print(f"| HalfCheetah-v4 N=131072 / 32768 / 8192 (BASELINE config 3, fp64) | {v('HalfCheetah', 131072)} / {v('HalfCheetah', 32768)} / **{v('HalfCheetah', 8192)}** |")
    print(f"| HalfCheetah-v4 / Hopper-v4 N=65536 on K3 (`planar_layout=1`, exact line search) | {v('HalfCheetah', 65536, 'K3')} / {v('Hopper', 65536, 'K3')} |")
    print(f"| Walker2d-v4 / Hopper-v4 N=65536 | {v('Walker2d', 65536)} / **{v('Hopper', 65536)}** |")
    print(f"| Ant-v4 N=32768 (config 4's shard) / 65536; fp32 at 65536 | **{v('Ant', 32768)}** / {v('Ant', 65536)}; {v('Ant', 65536, 'fp32')} |")
    print(f"| Humanoid-v4 / HumanoidStandup-v4 / Pusher-v4 N=65536 | {v('Humanoid', 65536)} / {v('HumanoidStandup', 65536)} / {v('Pusher', 65536)} |")
    print(f"| Humanoid-v4 / HumanoidStandup-v4 / Walker2d-v4 / Ant-v4 (32768), async: two half batches in flight | {asy('Humanoid', 65536)} / {asy('HumanoidStandup', 65536)} / {asy('Walker2d', 65536)} / {asy('Ant', 32768)} |")
    c = h.get("cpu_baseline") or {}
    if c:
        print(f"\ncpu_baseline: reference threadpool {c['value']:.3g} on {c['cores']} threads; openmp_port {c.get('openmp_port', {}).get('value', 0):.3g}")


if __name__ == "__main__":
    if "--write" in sys.argv:  # replace what stands between the two markers of DESIGN.md
        import contextlib
        import io

        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            main()
        t1, t2 = buf.getvalue().split("\n\n")[:2]
        path = os.path.join(ROOT, "DESIGN.md")
        doc = open(path).read()
        for name, body in (("issued", t1), ("throughput", t2)):
            a, b = f"<!-- table:{name}:begin -->\n", f"<!-- table:{name}:end -->"
            i, j = doc.index(a) + len(a), doc.index(b)
            doc = doc[:i] + body.strip() + "\n" + doc[j:]
        open(path, "w").write(doc)
    else:
        main()
