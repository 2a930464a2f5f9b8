"""Build guard for the MuJoCo kernels (no GPU needed): ROCm 7.2 / gfx950 builds of these
one-wave-per-SIMD kernels with HEAVY SGPR spilling have returned wrong results (mj_cheetah with
170 SGPR spills, run-to-run different; mujoco_pusher with the model as a kernel argument, ~250
spills: joint-limit forces wrong on 75 % of the envs while the same source was right in an
isolated harness -- see envpool_amd/csrc/Makefile).  The GPU parity tests catch a
bad build; this test catches the precondition already on the build box, from the code object's
own metadata."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "envpool_amd", "lib", "libenvpool_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"
# every model is a compile-time constant now (gen_mj_consts.cpp); the largest count among the
# one-env-per-lane kernels is 48 (ClassicStepKernel<4>), the known-bad builds had 170 and ~250
MAX_SGPR_SPILLS = 64
# Humanoid4StepKernel: 290 (Humanoid) / 570 (Standup, 24 register rows) SGPRs -- kernel arguments and
# literals set up before the step loop and needed again by the observation / reward epilogue -- are
# parked in VGPR lanes ACROSS the loop: the v_writelane sit before it; Humanoid reads them back after
# it, Standup also reads ~110 of them back inside the row build (never written there).  Both are
# checked against the oracle to 1e-13 on the GPU.  Bounded separately so that growth is noticed.
MAX_SGPR_SPILLS_BY_KERNEL = {"Humanoid4StepKernel": 600}


def _kernel_metadata():
    if not os.path.exists(LIB):
        pytest.skip("libenvpool_amd.so not built")
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("ROCm llvm tools not installed")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(LIB, tmp)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "libenvpool_amd.so"],
                       cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "hipv4" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=tmp,
                                   check=True, capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                sgpr = re.search(r"\.sgpr_spill_count:\s+(\d+)", block)
                scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
                if name and sgpr:
                    out[name.group(1)] = (int(sgpr.group(1)), int(scratch.group(1)) if scratch else 0)
    return out


def test_step_kernels_do_not_spill_sgprs_heavily():
    meta = {k: v for k, v in _kernel_metadata().items() if "StepKernel" in k}
    assert len(meta) >= 30, sorted(meta)  # classic 5, toy 6, planar 8, Ant 4, chains 4, Humanoid 4, Pusher 2
    worst = max(meta.items(), key=lambda kv: kv[1][0])
    print("most SGPR spills:", worst)
    def limit(name):
        for key, lim in MAX_SGPR_SPILLS_BY_KERNEL.items():
            if key in name:
                return lim
        return MAX_SGPR_SPILLS

    bad = {k: v for k, v in meta.items() if v[0] > limit(k)}
    assert not bad, bad


# ---- the product never reaches the checker (oracle/ is test infrastructure) ----
def test_product_python_never_touches_oracle():
    pkg = os.path.join(ROOT, "envpool_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            for n, line in enumerate(src.splitlines(), 1):
                code = line.split("#", 1)[0]
                if re.search(r"\boracle\b", code) and not code.lstrip().startswith(('"', "'")):
                    bad.append(f"{os.path.relpath(os.path.join(dirpath, f), ROOT)}:{n}: {line.strip()}")
    assert bad == [], bad


def test_product_library_does_not_link_the_oracle():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    dyn = subprocess.run([f"{LLVM}/llvm-readelf", "--dynamic", LIB], capture_output=True, text=True, check=True).stdout
    needed = re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", dyn)
    assert needed and not [n for n in needed if "oracle" in n or "mjcpu" in n or "ref_driver" in n], needed
    syms = subprocess.run([f"{LLVM}/llvm-readelf", "--dyn-syms", "-W", LIB], capture_output=True, text=True, check=True).stdout
    assert not re.findall(r"\b(mjcpu_\w+|oracle_\w+)\b", syms)


def test_missing_library_fails_loudly(monkeypatch):
    from envpool_amd.core import native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", os.path.join(ROOT, "envpool_amd", "lib", "does_not_exist.so"))
    with pytest.raises(RuntimeError):
        native.lib()
