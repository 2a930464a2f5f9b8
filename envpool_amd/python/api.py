"""py_env: wrap a native (spec, pool) class pair into the three public classes
(mirror of envpool/python/api.py:22-41)."""

from .dm_envpool import DMEnvPoolMeta
from .env_spec import EnvSpecMeta
from .gymnasium_envpool import GymnasiumEnvPoolMeta


def py_env(envspec: type, envpool: type) -> tuple[type, type, type]:
    # strip the leading "_" the native layer adds (py_envpool.h:304,321)
    spec_name = envspec.__name__[1:]
    pool_name = envpool.__name__[1:]
    return (
        EnvSpecMeta(spec_name, (envspec,), {}),
        DMEnvPoolMeta(pool_name.replace("EnvPool", "DMEnvPool"), (envpool,), {}),
        GymnasiumEnvPoolMeta(pool_name.replace("EnvPool", "GymnasiumEnvPool"), (envpool,), {}),
    )
