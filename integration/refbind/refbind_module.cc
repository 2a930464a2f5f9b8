// REFERENCE-SIDE BINDING, compiled in place against /root/reference (never copied):
// the reference's own pybind11 host shim -- PyEnvSpec / PyEnvPool / REGISTER from
// envpool/core/py_envpool.h:100-332 -- instantiated with DeviceEnvPool<Spec>
// (device_envpool.h) where the reference uses AsyncEnvPool<Env>
// (e.g. envpool/classic_control/classic_control_envpool.cc,
// envpool/mujoco/gym/mujoco_envpool.cc).  The Spec types (config, state/action
// specs, key order) are the reference's: CartPoleEnvSpec, PendulumEnvSpec,
// FrozenLakeEnvSpec, HalfCheetahEnvSpec, AntEnvSpec.
//
// The resulting module exposes _XxxEnvSpec / _XxxEnvPool with the attribute
// surface envpool/python/envpool.py:297-349 calls (_send / _recv / _reset /
// _spec / _state_keys / _action_keys ...).  tests/test_gpu_refbind.py checks it
// row-for-row against the ctypes path.
#include "envpool/core/py_envpool.h"

#include "envpool/classic_control/cartpole.h"
#include "envpool/classic_control/pendulum.h"
#include "envpool/mujoco/gym/ant.h"
#include "envpool/mujoco/gym/half_cheetah.h"
#include "envpool/toy_text/frozen_lake.h"

#include "device_envpool.h"

namespace eab = envpool_amd_binding;

// one small adapter per family: family name + the config keys the kernel reads,
// named exactly like the reference's DefaultConfig() keys
struct DeviceCartPolePool : eab::DeviceEnvPool<classic_control::CartPoleEnvSpec> {
  explicit DeviceCartPolePool(const classic_control::CartPoleEnvSpec& s)
      : DeviceEnvPool("CartPole", s, {}) {}
};
struct DevicePendulumPool : eab::DeviceEnvPool<classic_control::PendulumEnvSpec> {
  explicit DevicePendulumPool(const classic_control::PendulumEnvSpec& s)
      : DeviceEnvPool("Pendulum", s, {{"version", s.config["version"_]}}) {}
};
struct DeviceFrozenLakePool : eab::DeviceEnvPool<toy_text::FrozenLakeEnvSpec> {
  explicit DeviceFrozenLakePool(const toy_text::FrozenLakeEnvSpec& s)
      : DeviceEnvPool("FrozenLake", s, {{"size", s.config["size"_]}}) {}
};
struct DeviceHalfCheetahPool : eab::DeviceEnvPool<mujoco_gym::HalfCheetahEnvSpec> {
  explicit DeviceHalfCheetahPool(const mujoco_gym::HalfCheetahEnvSpec& s)
      : DeviceEnvPool(
            "HalfCheetah", s,
            {{"frame_skip", s.config["frame_skip"_]},
             {"frame_stack", s.config["frame_stack"_]},
             {"post_constraint", s.config["post_constraint"_]},
             {"ctrl_cost_weight", s.config["ctrl_cost_weight"_]},
             {"forward_reward_weight", s.config["forward_reward_weight"_]},
             {"reset_noise_scale", s.config["reset_noise_scale"_]},
             {"exclude_current_positions_from_observation",
              s.config["exclude_current_positions_from_observation"_]}}) {}
};
struct DeviceAntPool : eab::DeviceEnvPool<mujoco_gym::AntEnvSpec> {
  explicit DeviceAntPool(const mujoco_gym::AntEnvSpec& s)
      : DeviceEnvPool(
            "Ant", s,
            {{"frame_skip", s.config["frame_skip"_]},
             {"frame_stack", s.config["frame_stack"_]},
             {"post_constraint", s.config["post_constraint"_]},
             {"ctrl_cost_weight", s.config["ctrl_cost_weight"_]},
             {"contact_cost_weight", s.config["contact_cost_weight"_]},
             {"healthy_reward", s.config["healthy_reward"_]},
             {"healthy_z_min", s.config["healthy_z_min"_]},
             {"healthy_z_max", s.config["healthy_z_max"_]},
             {"contact_force_min", s.config["contact_force_min"_]},
             {"contact_force_max", s.config["contact_force_max"_]},
             {"reset_noise_scale", s.config["reset_noise_scale"_]},
             {"forward_reward_weight", s.config["forward_reward_weight"_]},
             {"terminate_when_unhealthy", s.config["terminate_when_unhealthy"_]},
             {"use_contact_force", s.config["use_contact_force"_]},
             {"legacy_healthy_reward", s.config["legacy_healthy_reward"_]},
             {"exclude_worldbody_contact_forces", s.config["exclude_worldbody_contact_forces"_]},
             {"exclude_current_positions_from_observation",
              s.config["exclude_current_positions_from_observation"_]}}) {}
};

// same naming as the reference's *_envpool.cc files
using CartPoleEnvSpec = PyEnvSpec<classic_control::CartPoleEnvSpec>;
using CartPoleEnvPool = PyEnvPool<DeviceCartPolePool>;
using PendulumEnvSpec = PyEnvSpec<classic_control::PendulumEnvSpec>;
using PendulumEnvPool = PyEnvPool<DevicePendulumPool>;
using FrozenLakeEnvSpec = PyEnvSpec<toy_text::FrozenLakeEnvSpec>;
using FrozenLakeEnvPool = PyEnvPool<DeviceFrozenLakePool>;
using GymHalfCheetahEnvSpec = PyEnvSpec<mujoco_gym::HalfCheetahEnvSpec>;
using GymHalfCheetahEnvPool = PyEnvPool<DeviceHalfCheetahPool>;
using GymAntEnvSpec = PyEnvSpec<mujoco_gym::AntEnvSpec>;
using GymAntEnvPool = PyEnvPool<DeviceAntPool>;

PYBIND11_MODULE(refbind, m) {
  m.doc() = "the reference's pybind11 shim over libenvpool_amd.so (test fixture)";
  REGISTER(m, CartPoleEnvSpec, CartPoleEnvPool)
  REGISTER(m, PendulumEnvSpec, PendulumEnvPool)
  REGISTER(m, FrozenLakeEnvSpec, FrozenLakeEnvPool)
  REGISTER(m, GymHalfCheetahEnvSpec, GymHalfCheetahEnvPool)
  REGISTER(m, GymAntEnvSpec, GymAntEnvPool)
}
