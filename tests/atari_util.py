"""Test helper: builds the synthetic-console emulator plugin (tests/synth_ale) and returns its
path.  ALE and its ROMs are not available offline; the product loads this plugin through the
same ABI it would load the real ALE adapter with (include/envpool_amd_emulator.h)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "synth_ale", "plugin.cc")
HDR = os.path.join(HERE, "synth_ale", "synth_ale.h")
SO = os.path.join(HERE, "synth_ale", "libsynth_ale.so")
ROMS = ("synth_fire", "synth_nofire", "synth_fire_short")


def plugin_path() -> str:
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", SRC, "-o", SO], check=True)
    return SO


def adapter_path() -> str:
    """integration/ale_adapter/ale_adapter.cc (the real-ALE plugin source) compiled against the ALE-API
    shim of oracle/ref_shims_atari: the same console behind the API a deployment would use."""
    integ = os.path.join(os.path.dirname(HERE), "integration")
    subprocess.run(["make", "-s", "-C", integ, "adapter"], check=True)
    return os.path.join(integ, "_build", "libepa_ale_over_shim.so")


def register_synthetic_ids() -> None:
    """`SynthFire-v5` etc.: the ids the reference would derive from ROM files of these names."""
    from envpool_amd.registration import list_all_envs, register

    have = set(list_all_envs())
    for game in ROMS:
        name = "".join(g.capitalize() for g in game.split("_")) + "-v5"
        if name not in have:
            register(task_id=name, import_path="envpool_amd.atari", spec_cls="AtariEnvSpec",
                     dm_cls="AtariDMEnvPool", gymnasium_cls="AtariGymnasiumEnvPool",
                     task=game, max_episode_steps=27000)
