"""Atari end to end (A1 host emulator loop, A2 palette / max-pool / resize / stack kernel,
A3 WriteState) against the reference's own atari_env.h.

Oracle: oracle/_ref/libref_atari.so = envpool/atari/atari_env.h compiled in place over the
synthetic console (ALE un-vendored) and the cv::resize restatement, and the fixtures
tests/golden/atari_*.npz generated from it (tests/golden/make_atari_golden.py).  Everything is
integer / byte work: the bar is bit-exact."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import atari_cases as ac  # noqa: E402
from atari_util import adapter_path, plugin_path, register_synthetic_ids  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


PLUGINS = {"synth_plugin": plugin_path, "ale_adapter_over_shim": adapter_path}


def make_pool(name, batch_size=0, num_threads=3, plugin="synth_plugin"):
    from envpool_amd.atari import AtariDevicePool

    c = ac.config(name)
    _, n, seed, max_steps, _ = ac.case(name)
    conf = {k: c[k] for k in ac.KEYS if k != "rom"}
    conf.update(num_envs=n, task=ac.ROMS[c["rom"]], base_path="/synthetic",
                emulator_lib=PLUGINS[plugin](), num_threads=num_threads)
    return AtariDevicePool(conf, batch_size=batch_size, seed=seed, max_episode_steps=max_steps)


@pytest.mark.parametrize("plugin", list(PLUGINS))
@pytest.mark.parametrize("name", list(ac.CASES))
def test_atari_env_matches_reference_fixtures(name, plugin):
    """Both emulator plugins: the synthetic console behind the plugin ABI directly, and the real-ALE
    adapter source (integration/ale_adapter/ale_adapter.cc: getScreen().getArray(), getRAM().array(),
    theOSystem->colourPalette(), Logger::setMode ...) compiled against the ALE-API shim."""
    g = np.load(os.path.join(GOLDEN, f"atari_{name}.npz"))
    _, n, seed, max_steps, steps = ac.CASES[name]
    c = ac.config(name)
    pool = make_pool(name, plugin=plugin)
    assert pool.action_keys[-1][1] == np.int32
    planes = c["stack_num"] * (1 if c["gray_scale"] else 3)
    assert dict((k, s) for k, _, s in pool.state_keys)["obs"] == (planes, c["img_height"], c["img_width"])
    ids = np.arange(n, dtype=np.int32)
    ref = None
    from oracle import orc
    if orc.have_ref_atari():  # the reference itself, when it travelled with the snapshot
        ref = orc.Oracle("Atari", n, seed=seed, max_episode_steps=max_steps, extra=ac.extra(c),
                         kind="reference_atari", num_threads=2)
    pool.reset(ids)
    a = pool.recv_dict()
    b = ref.reset() if ref else None
    acts = g["actions"]
    for t in range(steps + 1):
        for k in ac.SCALARS:
            np.testing.assert_array_equal(a[k].ravel(), g[k.replace(":", "__")][t], err_msg=f"{name} {k} @ {t}")
        np.testing.assert_array_equal(ac.crc_rows(a["obs"]), g["obs_crc"][t], err_msg=f"{name} obs crc @ {t}")
        np.testing.assert_array_equal(ac.crc_rows(a["info:ram"]), g["ram_crc"][t], err_msg=f"{name} ram @ {t}")
        if f"obs_{t}" in g.files:
            np.testing.assert_array_equal(a["obs"].reshape(n, -1), g[f"obs_{t}"])
        if ref:
            np.testing.assert_array_equal(a["obs"].reshape(n, -1), b["obs"], err_msg=f"{name} obs @ {t}")
            np.testing.assert_array_equal(a["discount"].ravel(), b["discount"].ravel())
        if t < steps:
            pool.send(ids, acts[t])
            a = pool.recv_dict()
            b = ref.step(acts[t]) if ref else None
    pool.close()


def test_atari_env_at_baseline_config5_size():
    """BASELINE.json config 5 (Atari Pong-v5, num_envs=1024) through the whole pool -- host emulator workers,
    H2D of 2 x 1024 raw screens, AtariPostKernel over 1024 blocks, chunked D2H -- against the reference's
    atari_env.h rollout of the same 1024 envs (fixture: scalars + per-row CRC32 of obs and RAM for every step,
    incl. the step where all 1024 episodes are truncated and auto-reset; live reference library where present)."""
    name = "config5_n1024"
    g = np.load(os.path.join(GOLDEN, f"atari_{name}.npz"))
    _, n, seed, max_steps, steps = ac.case(name)
    assert n == 1024
    pool = make_pool(name, num_threads=8)
    ids = np.arange(n, dtype=np.int32)
    ref = None
    from oracle import orc
    if orc.have_ref_atari():
        ref = orc.Oracle("Atari", n, seed=seed, max_episode_steps=max_steps, extra=ac.extra(ac.config(name)),
                         kind="reference_atari", num_threads=4)
    pool.reset(ids)
    a = pool.recv_dict()
    b = ref.reset() if ref else None
    acts = g["actions"]
    ended = 0
    for t in range(steps + 1):
        for k in ac.SCALARS:
            np.testing.assert_array_equal(a[k].ravel(), g[k.replace(":", "__")][t], err_msg=f"{k} @ {t}")
        np.testing.assert_array_equal(ac.crc_rows(a["obs"]), g["obs_crc"][t], err_msg=f"obs crc @ {t}")
        np.testing.assert_array_equal(ac.crc_rows(a["info:ram"]), g["ram_crc"][t], err_msg=f"ram @ {t}")
        if ref:
            np.testing.assert_array_equal(a["obs"].reshape(n, -1), b["obs"], err_msg=f"obs @ {t}")
        ended += int(a["done"].sum())
        if t < steps:
            pool.send(ids, acts[t])
            a = pool.recv_dict()
            b = ref.step(acts[t]) if ref else None
    assert ended >= n  # every env finished an episode and was auto-reset inside the window
    pool.close()


def test_atari_async_mode_streams_match_sync():
    """batch_size < num_envs: rows arrive first come first served, but every env's own
    sequence of outputs is the one of the sync run (actions are a function of env and step)."""
    name = "default"
    _, n, _, _, _ = ac.CASES[name]
    steps = 40

    def act_of(e, t):
        return np.int32((7 * e + 3 * t) % 6)

    sync = make_pool(name)
    ids = np.arange(n, dtype=np.int32)
    sync.reset(ids)
    seq = {e: [] for e in range(n)}
    a = sync.recv_dict()
    for t in range(steps):
        for e in range(n):
            seq[e].append((int(a["elapsed_step"][e]), float(a["reward"][e]), bool(a["done"][e]),
                           ac.crc_rows(a["obs"][e:e + 1])[0]))
        sync.send(ids, np.array([act_of(e, t) for e in range(n)], dtype=np.int32))
        a = sync.recv_dict()
    sync.close()
    pool = make_pool(name, batch_size=2, num_threads=4)
    pool.reset(ids)
    count = {e: 0 for e in range(n)}
    while min(count.values()) < steps - 1:
        b = pool.recv_dict()
        assert b["obs"].shape[0] == 2
        eids = b["info:env_id"].ravel()
        for r, e in enumerate(eids):
            t = count[int(e)]
            if t < steps:
                got = (int(b["elapsed_step"][r]), float(b["reward"][r]), bool(b["done"][r]),
                       ac.crc_rows(b["obs"][r:r + 1])[0])
                assert got == seq[int(e)][t], (int(e), t, got, seq[int(e)][t])
            count[int(e)] += 1
        pool.send(eids.astype(np.int32),
                  np.array([act_of(int(e), count[int(e)] - 1) for e in eids], dtype=np.int32))
    pool.close()


def test_atari_python_api_and_errors():
    import envpool_amd as envpool

    register_synthetic_ids()
    env = envpool.make("SynthFire-v5", env_type="gymnasium", num_envs=4, seed=1,
                       emulator_lib=plugin_path(), base_path="/synthetic", episodic_life=True)
    assert env.action_space.n == 6 and env.observation_space.shape == (4, 84, 84)
    obs, info = env.reset()
    assert obs.shape == (4, 4, 84, 84) and obs.dtype == np.uint8
    assert set(info) >= {"lives", "reward", "terminated", "ram", "env_id", "elapsed_step"}
    assert (obs[:, 0] == obs[:, 3]).all()  # a reset shows the frame in every stack slot
    for _ in range(5):
        obs, rew, term, trunc, info = env.step(np.array([2, 3, 0, 1], dtype=np.int32))
    assert rew.dtype == np.float32 and info["ram"].shape == (4, 128)
    env.close()
    with pytest.raises(ValueError, match="emulator plugin"):
        envpool.make("Pong-v5", env_type="gymnasium", num_envs=2)
    with pytest.raises(ValueError, match="cannot load ROM"):
        envpool.make("Pong-v5", env_type="gymnasium", num_envs=2, emulator_lib=plugin_path())
    with pytest.raises(RuntimeError, match="host"):
        make_pool("default").send_device(None)
