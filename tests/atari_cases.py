"""Shared Atari parity cases: config (reference key order of AtariEnvFns::DefaultConfig,
atari_env.h:52-63), ROM variant of the synthetic console, seeds, horizon."""
import zlib

import numpy as np

ROMS = ("synth_fire", "synth_nofire", "synth_fire_short")
KEYS = ("stack_num", "frame_skip", "noop_max", "zero_discount_on_life_loss", "episodic_life",
        "reward_clip", "use_fire_reset", "img_height", "img_width", "rom", "mode", "difficulty",
        "full_action_space", "repeat_action_probability", "use_inter_area_resize", "gray_scale")
DEFAULT = dict(stack_num=4, frame_skip=4, noop_max=30, zero_discount_on_life_loss=0,
               episodic_life=0, reward_clip=0, use_fire_reset=1, img_height=84, img_width=84,
               rom=0, mode=-1, difficulty=-1, full_action_space=0,
               repeat_action_probability=0.0, use_inter_area_resize=1, gray_scale=1)
# name -> (config overrides, num_envs, seed, max_episode_steps, steps)
CASES = {
    "default": ({}, 6, 11, 27000, 160),
    "episodic_clip": (dict(episodic_life=1, reward_clip=1, zero_discount_on_life_loss=1, rom=1,
                           noop_max=5), 6, 3, 27000, 200),
    "rgb_linear_short": (dict(gray_scale=0, use_inter_area_resize=0, stack_num=2, rom=2,
                              noop_max=3), 5, 7, 40, 120),
    "skip3_sticky_full": (dict(frame_skip=3, repeat_action_probability=0.25, full_action_space=1,
                               mode=1, difficulty=1, use_fire_reset=0), 5, 21, 27000, 120),
    "skip1_small": (dict(frame_skip=1, img_height=64, img_width=96, stack_num=3, rom=2), 4, 5, 50, 150),
}
# BASELINE.json config 5 ("Atari Pong-v5 num_envs=1024"): the default Pong-like configuration at the full
# batch size.  Kept apart from CASES (which every small-case test iterates): its fixture holds scalars and
# per-row CRCs only, no full observations.
BIG_CASES = {
    "config5_n1024": ({}, 1024, 13, 40, 48),
}


def case(name):
    return CASES[name] if name in CASES else BIG_CASES[name]


def config(name):
    c = dict(DEFAULT)
    c.update(case(name)[0])
    return c


def extra(c):
    """oracle/_ref/libref_atari.so: orc_create(extra[])"""
    return [float(c[k]) for k in KEYS]


def actions(name, num_actions):
    _, n, seed, _, steps = case(name)
    rng = np.random.default_rng(1000 + seed)
    return rng.integers(0, num_actions, size=(steps, n)).astype(np.int32)


def num_actions(c):
    if c["full_action_space"]:
        return 18
    return 6 if c["rom"] != 1 else 3


def crc_rows(a):
    a = np.ascontiguousarray(a)
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(a.shape[0])], dtype=np.uint32)


SCALARS = ("info:env_id", "elapsed_step", "done", "reward", "discount", "step_type", "trunc",
           "info:lives", "info:reward", "info:terminated")
