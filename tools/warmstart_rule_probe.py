"""Sizes the one engine rule SURVEY Appendix A could only tag "[M: exact place of the warm-start
copy]": is qacc_warmstart saved by every mj_fwdConstraint (RK4 stages 2-4 warm-start from the
previous stage; oracle/mjcpu default, warmstart_rule 0) or once per mj_step (rule 1)?

Converged Newton does not depend on the start; the 50-sweep PGS of Humanoid / HumanoidStandup
does.  For each task: a free-running rollout under rule 0 gives the visited states; at every step
the SAME state and action are also stepped once under rule 1 (teacher-forced), and the two
observations are compared.  Also reported: free-running separation of the two rules.

    python tools/warmstart_rule_probe.py > profiles/r4_warmstart_rule_probe.log
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.orc import Oracle  # noqa: E402
from tests.mj_util import mj_extra  # noqa: E402


def probe(task, n=16, steps=200, amp=0.4):
    a = Oracle(task, n, seed=1, max_episode_steps=1000, extra=mj_extra(task, warmstart_rule=0))
    b = Oracle(task, n, seed=1, max_episode_steps=1000, extra=mj_extra(task, warmstart_rule=1))
    c = Oracle(task, n, seed=1, max_episode_steps=1000, extra=mj_extra(task, warmstart_rule=1))
    a.reset(), b.reset(), c.reset()
    rng = np.random.default_rng(0)
    tf_abs, tf_rel, free = [], [], []
    for t in range(steps):
        act = rng.uniform(-amp, amp, size=(n, a.action_elems))
        b.set_state(a.get_state())
        ra, rb, rc = a.step(act), b.step(act), c.step(act)
        live = (ra["elapsed_step"].ravel() > 0) & (rb["elapsed_step"].ravel() > 0)
        d = np.abs(ra["obs"] - rb["obs"])[live]
        tf_abs.append(d.max(initial=0.0))
        tf_rel.append((d / (1e-6 + np.abs(ra["obs"][live]))).max(initial=0.0))
        livec = live & (rc["elapsed_step"].ravel() > 0)
        free.append(np.abs(ra["obs"] - rc["obs"])[livec].max(initial=0.0))
    tf_abs, tf_rel, free = map(np.asarray, (tf_abs, tf_rel, free))
    print(f"{task}: teacher-forced one env-step, rule 0 vs rule 1, {n} envs x {steps} steps: "
          f"max |d obs| {tf_abs.max():.3e} (median over steps {np.median(tf_abs):.3e}), "
          f"max rel {tf_rel.max():.3e}")
    print(f"{task}: free-running separation max |d obs| after 10 / 50 / {steps} steps: "
          f"{free[:10].max():.3e} / {free[:50].max():.3e} / {free.max():.3e}")


if __name__ == "__main__":
    for task, amp in (("Humanoid", 0.4), ("HumanoidStandup", 0.4), ("HalfCheetah", 1.0), ("Ant", 1.0),
                      ("Walker2d", 1.0), ("Hopper", 1.0)):
        probe(task, amp=amp)
