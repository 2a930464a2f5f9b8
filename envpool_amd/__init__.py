"""envpool_amd: MI355X-native batched-step engine behind envpool's public API.

    import envpool_amd as envpool
    env = envpool.make("HalfCheetah-v4", env_type="gymnasium", num_envs=65536)

mirrors `envpool/__init__.py` of the reference for the hot-path env families
(classic_control, toy_text, gym-MuJoCo HalfCheetah/Ant).
"""

from . import entry  # noqa: F401  (registers the envs)
from .core.affinity import bind_host_to_device, device_local_cpus, device_numa_node
from .registration import (
    list_all_envs,
    make,
    make_dm,
    make_gym,
    make_gymnasium,
    make_spec,
    register,
)

__version__ = "0.1.0"

__all__ = [
    "register",
    "make",
    "make_dm",
    "make_gym",
    "make_gymnasium",
    "make_spec",
    "list_all_envs",
    "bind_host_to_device",
    "device_local_cpus",
    "device_numa_node",
]
