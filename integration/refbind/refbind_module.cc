// REFERENCE-SIDE BINDING, compiled in place against /root/reference (never copied):
// the reference's own pybind11 host shim -- PyEnvSpec / PyEnvPool / REGISTER from
// envpool/core/py_envpool.h:100-332 -- instantiated with DeviceEnvPool<Spec>
// (device_envpool.h) where the reference uses AsyncEnvPool<Env>
// (e.g. envpool/classic_control/classic_control_envpool.cc,
// envpool/mujoco/gym/mujoco_envpool.cc).  The Spec types (config, state/action
// specs, key order) are the reference's, for every family libenvpool_amd has a kernel for:
// classic_control (5), toy_text (6) and the gym-MuJoCo tasks (11).
//
// The resulting module exposes _XxxEnvSpec / _XxxEnvPool with the attribute
// surface envpool/python/envpool.py:297-349 calls (_send / _recv / _reset /
// _spec / _state_keys / _action_keys ...).  tests/test_gpu_refbind.py checks it
// row-for-row against the ctypes path.
#include "envpool/core/py_envpool.h"

#include "envpool/classic_control/acrobot.h"
#include "envpool/classic_control/cartpole.h"
#include "envpool/classic_control/mountain_car.h"
#include "envpool/classic_control/mountain_car_continuous.h"
#include "envpool/classic_control/pendulum.h"
#include "envpool/mujoco/gym/ant.h"
#include "envpool/mujoco/gym/half_cheetah.h"
#include "envpool/mujoco/gym/hopper.h"
#include "envpool/mujoco/gym/humanoid.h"
#include "envpool/mujoco/gym/humanoid_standup.h"
#include "envpool/mujoco/gym/inverted_double_pendulum.h"
#include "envpool/mujoco/gym/inverted_pendulum.h"
#include "envpool/mujoco/gym/pusher.h"
#include "envpool/mujoco/gym/reacher.h"
#include "envpool/mujoco/gym/swimmer.h"
#include "envpool/mujoco/gym/walker2d.h"
#include "envpool/toy_text/blackjack.h"
#include "envpool/toy_text/catch.h"
#include "envpool/toy_text/cliffwalking.h"
#include "envpool/toy_text/frozen_lake.h"
#include "envpool/toy_text/nchain.h"
#include "envpool/toy_text/taxi.h"

#include "device_envpool.h"

namespace eab = envpool_amd_binding;

// One adapter per family: the kernel's family name; the kernel parameters are the Spec's own
// numeric config entries under the reference's DefaultConfig() key names (NumericParams).
#define EAB_DEVICE_POOL(Name, SpecT, family)                           \
  struct Name : eab::DeviceEnvPool<SpecT> {                            \
    explicit Name(const SpecT& s)                                      \
        : DeviceEnvPool(family, s, eab::NumericParams(s.config)) {}    \
  };

// the two model variants a family ships are told apart by xml_file (mujoco_env.h:50-58)
inline eab::Params XmlVariant(const std::string& xml_file, const char* v4, const char* v5) {
  if (xml_file == v5) return {{"xml_v5", 1.0}};
  if (xml_file == v4) return {{"xml_v5", 0.0}};
  throw std::invalid_argument("xml_file=" + xml_file + " has no compiled-in model");
}

EAB_DEVICE_POOL(DeviceCartPolePool, classic_control::CartPoleEnvSpec, "CartPole")
EAB_DEVICE_POOL(DevicePendulumPool, classic_control::PendulumEnvSpec, "Pendulum")
EAB_DEVICE_POOL(DeviceMountainCarPool, classic_control::MountainCarEnvSpec, "MountainCar")
EAB_DEVICE_POOL(DeviceMountainCarContinuousPool, classic_control::MountainCarContinuousEnvSpec,
                "MountainCarContinuous")
EAB_DEVICE_POOL(DeviceAcrobotPool, classic_control::AcrobotEnvSpec, "Acrobot")
EAB_DEVICE_POOL(DeviceCatchPool, toy_text::CatchEnvSpec, "Catch")
EAB_DEVICE_POOL(DeviceFrozenLakePool, toy_text::FrozenLakeEnvSpec, "FrozenLake")
EAB_DEVICE_POOL(DeviceTaxiPool, toy_text::TaxiEnvSpec, "Taxi")
EAB_DEVICE_POOL(DeviceNChainPool, toy_text::NChainEnvSpec, "NChain")
EAB_DEVICE_POOL(DeviceCliffWalkingPool, toy_text::CliffWalkingEnvSpec, "CliffWalking")
EAB_DEVICE_POOL(DeviceBlackjackPool, toy_text::BlackjackEnvSpec, "Blackjack")
EAB_DEVICE_POOL(DeviceHalfCheetahPool, mujoco_gym::HalfCheetahEnvSpec, "HalfCheetah")
EAB_DEVICE_POOL(DeviceAntPool, mujoco_gym::AntEnvSpec, "Ant")
EAB_DEVICE_POOL(DeviceHopperPool, mujoco_gym::HopperEnvSpec, "Hopper")
EAB_DEVICE_POOL(DeviceSwimmerPool, mujoco_gym::SwimmerEnvSpec, "Swimmer")
EAB_DEVICE_POOL(DeviceReacherPool, mujoco_gym::ReacherEnvSpec, "Reacher")
EAB_DEVICE_POOL(DeviceInvertedPendulumPool, mujoco_gym::InvertedPendulumEnvSpec, "InvertedPendulum")
EAB_DEVICE_POOL(DeviceInvertedDoublePendulumPool, mujoco_gym::InvertedDoublePendulumEnvSpec,
                "InvertedDoublePendulum")
EAB_DEVICE_POOL(DeviceHumanoidPool, mujoco_gym::HumanoidEnvSpec, "Humanoid")
EAB_DEVICE_POOL(DeviceHumanoidStandupPool, mujoco_gym::HumanoidStandupEnvSpec, "HumanoidStandup")
struct DeviceWalker2dPool : eab::DeviceEnvPool<mujoco_gym::Walker2dEnvSpec> {
  explicit DeviceWalker2dPool(const mujoco_gym::Walker2dEnvSpec& s)
      : DeviceEnvPool("Walker2d", s,
                      eab::NumericParams(s.config, XmlVariant(s.config["xml_file"_], "walker2d.xml",
                                                              "walker2d_v5.xml"))) {}
};
struct DevicePusherPool : eab::DeviceEnvPool<mujoco_gym::PusherEnvSpec> {
  explicit DevicePusherPool(const mujoco_gym::PusherEnvSpec& s)
      : DeviceEnvPool("Pusher", s,
                      eab::NumericParams(s.config, XmlVariant(s.config["xml_file"_], "pusher.xml",
                                                              "pusher_v5.xml"))) {}
};

// same naming as the reference's *_envpool.cc files (classic_control_envpool.cc,
// toy_text_envpool.cc, mujoco/gym/mujoco_envpool.cc)
#define EAB_REGISTER(m, Stem, SpecT, PoolT)       \
  using Stem##EnvSpec = PyEnvSpec<SpecT>;         \
  using Stem##EnvPool = PyEnvPool<PoolT>;         \
  REGISTER(m, Stem##EnvSpec, Stem##EnvPool)

PYBIND11_MODULE(refbind, m) {
  m.doc() = "the reference's pybind11 shim over libenvpool_amd.so (test fixture)";
  EAB_REGISTER(m, CartPole, classic_control::CartPoleEnvSpec, DeviceCartPolePool)
  EAB_REGISTER(m, Pendulum, classic_control::PendulumEnvSpec, DevicePendulumPool)
  EAB_REGISTER(m, MountainCar, classic_control::MountainCarEnvSpec, DeviceMountainCarPool)
  EAB_REGISTER(m, MountainCarContinuous, classic_control::MountainCarContinuousEnvSpec,
               DeviceMountainCarContinuousPool)
  EAB_REGISTER(m, Acrobot, classic_control::AcrobotEnvSpec, DeviceAcrobotPool)
  EAB_REGISTER(m, Catch, toy_text::CatchEnvSpec, DeviceCatchPool)
  EAB_REGISTER(m, FrozenLake, toy_text::FrozenLakeEnvSpec, DeviceFrozenLakePool)
  EAB_REGISTER(m, Taxi, toy_text::TaxiEnvSpec, DeviceTaxiPool)
  EAB_REGISTER(m, NChain, toy_text::NChainEnvSpec, DeviceNChainPool)
  EAB_REGISTER(m, CliffWalking, toy_text::CliffWalkingEnvSpec, DeviceCliffWalkingPool)
  EAB_REGISTER(m, Blackjack, toy_text::BlackjackEnvSpec, DeviceBlackjackPool)
  EAB_REGISTER(m, GymHalfCheetah, mujoco_gym::HalfCheetahEnvSpec, DeviceHalfCheetahPool)
  EAB_REGISTER(m, GymAnt, mujoco_gym::AntEnvSpec, DeviceAntPool)
  EAB_REGISTER(m, GymWalker2d, mujoco_gym::Walker2dEnvSpec, DeviceWalker2dPool)
  EAB_REGISTER(m, GymHopper, mujoco_gym::HopperEnvSpec, DeviceHopperPool)
  EAB_REGISTER(m, GymSwimmer, mujoco_gym::SwimmerEnvSpec, DeviceSwimmerPool)
  EAB_REGISTER(m, GymReacher, mujoco_gym::ReacherEnvSpec, DeviceReacherPool)
  EAB_REGISTER(m, GymPusher, mujoco_gym::PusherEnvSpec, DevicePusherPool)
  EAB_REGISTER(m, GymInvertedPendulum, mujoco_gym::InvertedPendulumEnvSpec,
               DeviceInvertedPendulumPool)
  EAB_REGISTER(m, GymInvertedDoublePendulum, mujoco_gym::InvertedDoublePendulumEnvSpec,
               DeviceInvertedDoublePendulumPool)
  EAB_REGISTER(m, GymHumanoid, mujoco_gym::HumanoidEnvSpec, DeviceHumanoidPool)
  EAB_REGISTER(m, GymHumanoidStandup, mujoco_gym::HumanoidStandupEnvSpec,
               DeviceHumanoidStandupPool)
}
