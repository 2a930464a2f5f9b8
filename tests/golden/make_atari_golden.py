"""Generates tests/golden/atari_<case>.npz from oracle/_ref/libref_atari.so, i.e. from the
reference's OWN envpool/atari/atari_env.h compiled in place over the synthetic console
(tests/synth_ale) and the cv::resize restatement (oracle/atari/atari_post.c).  Run in the build
container (needs /root/reference):  make -C oracle ref && python tests/golden/make_atari_golden.py
Per step the file keeps every scalar key, CRC32 of each obs row and of each RAM row, and the
full observation of a few steps."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atari_cases as ac  # noqa: E402
from oracle.orc import Oracle  # noqa: E402

only = sys.argv[1:]  # e.g. `make_atari_golden.py config5_n1024`: regenerate just that fixture
for name, (_, n, seed, max_steps, steps) in {**ac.CASES, **ac.BIG_CASES}.items():
    if only and name not in only:
        continue
    c = ac.config(name)
    orc = Oracle("Atari", n, seed=seed, max_episode_steps=max_steps, extra=ac.extra(c),
                 kind="reference_atari", num_threads=2)
    assert orc.action_dtype == np.int32
    acts = ac.actions(name, ac.num_actions(c))
    out = {k: [] for k in ac.SCALARS}
    obs_crc, ram_crc, full = [], [], {}
    b = orc.reset()
    for t in range(steps + 1):
        for k in ac.SCALARS:
            out[k].append(b[k].ravel().copy())
        obs_crc.append(ac.crc_rows(b["obs"]))
        ram_crc.append(ac.crc_rows(b["info:ram"]))
        if name in ac.CASES and t in (0, 1, 17, steps // 2, steps):
            full[f"obs_{t}"] = b["obs"].copy()
        if t < steps:
            b = orc.step(acts[t])
    path = os.path.join(ROOT, "tests", "golden", f"atari_{name}.npz")
    np.savez_compressed(path, actions=acts, obs_crc=np.array(obs_crc), ram_crc=np.array(ram_crc),
                        **{k.replace(":", "__"): np.array(v) for k, v in out.items()}, **full)
    d = np.array(out["done"])
    print(name, "steps", steps, "episodes ended", int(d.sum()), "trunc", int(np.array(out["trunc"]).sum()),
          "rewards", float(np.abs(np.array(out["reward"])).sum()), os.path.getsize(path), "bytes")
