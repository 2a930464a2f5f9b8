"""Import every family's registration (mirror of envpool/entry.py)."""
import envpool_amd.atari.registration  # noqa: F401
import envpool_amd.classic_control.registration  # noqa: F401
import envpool_amd.mujoco.gym.registration  # noqa: F401
import envpool_amd.toy_text.registration  # noqa: F401
