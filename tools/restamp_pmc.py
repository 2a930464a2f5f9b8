"""Re-stamps profiles/pmc.json entries of ONE translation unit with the hash of its sources at HEAD -- allowed only when
the machine code is the same: the .text section of the TU's gfx950 code object built from the commit the counts were
collected on must equal the one built from the working tree (edits that only touch diagnostic macros, host code or
comments of a kernel source).  usage: tools/restamp_pmc.py <tu.hip> <commit the profile ran on> <kernel base name>"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_sources import base_name, source_hash  # noqa: E402

_MJ = ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-spill-sgpr-to-vgpr=false"]
FLAGS = {"mujoco_planar_lg.hip": _MJ, "mujoco_gym.hip": _MJ, "mujoco_pusher.hip": _MJ,
         "mujoco_ant.hip": _MJ + ["-fno-slp-vectorize"]}
LLVM = "/opt/rocm/lib/llvm/bin"


def text_sha(csrc, tu):
    with tempfile.TemporaryDirectory() as t:
        co, elf, txt = (os.path.join(t, x) for x in ("a.co", "a.elf", "a.text"))
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + FLAGS.get(tu, []) +
                       ["--cuda-device-only", "-c", tu, "-o", co], cwd=csrc, check=True, capture_output=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={co}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"], check=True)
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.text", elf, txt], check=True)
        return hashlib.sha256(open(txt, "rb").read()).hexdigest()


def main():
    tu, commit, base = sys.argv[1], sys.argv[2], sys.argv[3]
    with tempfile.TemporaryDirectory() as old:
        tar = subprocess.run(["git", "archive", commit, "envpool_amd/csrc", "include"], cwd=ROOT, check=True, capture_output=True)
        subprocess.run(["tar", "-x", "-C", old], input=tar.stdout, check=True)
        os.makedirs(os.path.join(old, "envpool_amd", "csrc", "build"), exist_ok=True)
        for f in os.listdir(os.path.join(ROOT, "envpool_amd", "csrc", "build")):
            if f.endswith(".inc"):  # generated model constants (their generators are hashed with the kernel sources)
                subprocess.run(["cp", os.path.join(ROOT, "envpool_amd", "csrc", "build", f),
                                os.path.join(old, "envpool_amd", "csrc", "build", f)], check=True)
        a = text_sha(os.path.join(old, "envpool_amd", "csrc"), tu)
    b = text_sha(os.path.join(ROOT, "envpool_amd", "csrc"), tu)
    print(f"{tu}: .text sha256 at {commit}: {a[:16]}, working tree: {b[:16]}")
    if a != b:
        sys.exit("machine code differs: re-profile instead")
    p = os.path.join(ROOT, "profiles", "pmc.json")
    pmc = json.load(open(p))
    n = 0
    for k, v in pmc.items():
        if base_name(k) == base and v.get("src_hash") != source_hash(k):
            v["src_hash"] = source_hash(k)
            v["restamped"] = f"sources changed after the profile ({commit[:7]}) without changing the machine code (.text {a[:16]})"
            n += 1
    json.dump(pmc, open(p, "w"), indent=1)
    print(f"restamped {n} entries of {base}")


if __name__ == "__main__":
    main()
