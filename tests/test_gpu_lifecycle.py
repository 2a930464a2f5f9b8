"""Pool lifecycle on the GPU: building, stepping and destroying pools of every family over and over must give the device
memory back (the reference's pools are plain host objects whose destructor joins the workers, async_envpool.h:99-116;
ours own device state, result blocks, scratch, streams and events -- a leak of any of them ends a long training job)."""
import gc

import numpy as np
import pytest

import envpool_amd as envpool

pytestmark = pytest.mark.gpu

TASKS = ["CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "FrozenLake-v1", "Taxi-v3", "Blackjack-v1",
         "Catch-v0", "HalfCheetah-v4", "Walker2d-v4", "Hopper-v4", "Ant-v4", "Humanoid-v4", "Pusher-v4",
         "InvertedDoublePendulum-v4", "Swimmer-v4"]


MUJOCO = {"HalfCheetah", "Walker2d", "Hopper", "Ant", "Humanoid", "Pusher", "InvertedDoublePendulum", "Swimmer"}


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def _exercise(task, num_envs, async_mode):
    kw = dict(num_envs=num_envs, seed=1, max_episode_steps=7)
    if async_mode:
        kw["batch_size"] = num_envs // 2
    env = envpool.make_gym(task, **kw)
    space = env.action_space
    rng = np.random.default_rng(0)

    def act(k):
        if hasattr(space, "n"):
            return rng.integers(0, space.n, k)
        return rng.uniform(space.low, space.high, (k, *space.shape))

    if async_mode:
        env.async_reset()
        for _ in range(12):
            obs, rew, term, trunc, info = env.recv()
            env.send(act(len(info["env_id"])), info["env_id"])
    else:
        env.reset()
        for _ in range(9):  # crosses max_episode_steps: the auto-reset path allocates nothing either
            env.step(act(num_envs))
    env.close()
    del env
    gc.collect()


@pytest.mark.parametrize("task", TASKS)
def test_pools_give_their_memory_back(task):
    # sized so that one pool's state is well above the slack below (the classic / toy pools hold ~100 B per env)
    num_envs = 2048 if task.split("-")[0] in MUJOCO else 65536
    for mode in (False, True):  # first round: one-time allocations (module load, self-test, pinned-block cache)
        _exercise(task, num_envs, mode)
    base = _free_bytes()
    lows = []
    for rep in range(6):
        _exercise(task, num_envs, rep % 2 == 1)
        lows.append(_free_bytes())
    # a leak of one pool's state (>= 1 MB at this size for every family but the smallest) would show as a steady fall;
    # the allocator's own granularity (2 MB fragments) is the slack
    assert base - min(lows) <= 4 << 20, (task, base, lows)
    assert lows[-1] >= lows[1] - (2 << 20), (task, lows)


def test_many_small_pools_in_a_row():
    """300 create / reset / step / destroy cycles of a small pool (streams, events and the pinned ring are per pool)."""
    before = None
    for i in range(300):
        env = envpool.make_gym("CartPole-v1", num_envs=16, seed=i)
        env.reset()
        env.step(np.zeros(16, dtype=np.int32))
        env.close()
        if i == 20:
            before = _free_bytes()
    gc.collect()
    assert before - _free_bytes() <= 4 << 20
