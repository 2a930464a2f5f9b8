// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// oracle/_ref/libref_mujoco.so: the reference's own gym-MuJoCo task wrappers, compiled in place
// from /root/reference (never copied), running inside the reference's own AsyncEnvPool:
//   envpool/mujoco/gym/mujoco_env.h:82-148      MujocoEnv (MujocoReset / MujocoStep)
//   envpool/mujoco/gym/half_cheetah.h:105-185   ant.h:135-278   walker2d.h   hopper.h
//   swimmer.h   reacher.h   pusher.h   inverted_pendulum.h   inverted_double_pendulum.h
//   humanoid.h:172-268   humanoid_standup.h
//   envpool/mujoco/frame_stack.h                FrameStackBuffer
//   envpool/core/async_envpool.h:42-238, env.h  thread pool, queues, EnvStep bookkeeping
// MuJoCo itself (3.6.0, un-vendored) is replaced by oracle/ref_shims_mujoco/mujoco.h whose
// functions forward to the restatement oracle/mjcpu (ref_mujoco_shim.cc).  kind =
// "reference_mujoco": reference wrapper + runtime over a ported engine.  What this pins:
// every line of oracle/mjcpu/tasks.c (reset draws through libstdc++'s real
// uniform_real_distribution / normal_distribution, rewards, healthy / termination rules,
// observation and info assembly, post_constraint, elapsed_step / done / trunc bookkeeping).
// What it does not pin: the engine arithmetic (both sides run the same oracle/mjcpu/engine.c).
//
// Task names and the positions of `extra` are those of mjcpu_create (oracle/mjcpu/tasks.c), so
// a test can run the same case through kind="port" and kind="reference_mujoco".
#include <cstdint>
#include <string>

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

#include "ref_driver_common.h"

namespace {

struct Extra_ {
  const double* e;
  int n;
  bool Has(int i) const { return e != nullptr && i < n; }
  double Get(int i, double d) const { return Has(i) ? e[i] : d; }
  bool Flag(int i, bool d) const { return Has(i) ? e[i] != 0 : d; }
};

// weights every task shares (positions 0-3 of `extra`); post_constraint follows the port's
// default (the v4 registration: gym/registration.py:93 passes version == "v5")
template <typename C>
void Basic(C& c, const Extra_& x) {
  if (x.Has(0)) c["frame_skip"_] = static_cast<int>(x.e[0]);
  c["post_constraint"_] = x.Flag(13, false);
  c["frame_stack"_] = static_cast<int>(x.Get(23, 1));
}

}  // namespace

extern "C" {

void* orc_create(const char* task, int num_envs, int seed, int max_episode_steps,
                 const double* extra, int n_extra, int num_threads) {
  std::string t(task);
  Extra_ x{extra, n_extra};
  // engine debugging switches of the port (no contact / limit / actuation / passive, integrator
  // and timestep overrides, no self-collision) have no counterpart in the reference's config
  for (int i : {4, 5, 6, 7, 18}) {
    if (x.Get(i, 0) != 0) return nullptr;
  }
  if (x.Get(8, -1) >= 0 || x.Get(9, 0) > 0 || x.Get(24, 0) != 0) return nullptr;
  using namespace mujoco_gym;  // NOLINT
  try {
    if (t == "HalfCheetah") {
      return new Ref<HalfCheetahEnvPool>(num_envs, seed, max_episode_steps, num_threads,
                                         [&](auto& c) {
        Basic(c, x);
        if (x.Has(1)) c["ctrl_cost_weight"_] = x.e[1];
        if (x.Has(2)) c["forward_reward_weight"_] = x.e[2];
        if (x.Has(3)) c["reset_noise_scale"_] = x.e[3];
      });
    }
    if (t == "Ant") {
      return new Ref<AntEnvPool>(num_envs, seed, max_episode_steps, num_threads, [&](auto& c) {
        Basic(c, x);
        if (x.Has(1)) c["ctrl_cost_weight"_] = x.e[1];
        if (x.Has(2)) c["forward_reward_weight"_] = x.e[2];
        if (x.Has(3)) c["reset_noise_scale"_] = x.e[3];
        c["use_contact_force"_] = x.Flag(12, false);
        c["exclude_worldbody_contact_forces"_] = x.Flag(14, false);
        if (x.Get(15, -1) >= 0) c["legacy_healthy_reward"_] = x.e[15] != 0;
      });
    }
    if (t == "Walker2d" || t == "Walker2dV5") {
      const bool v5 = t == "Walker2dV5";
      return new Ref<Walker2dEnvPool>(num_envs, seed, max_episode_steps, num_threads,
                                      [&](auto& c) {
        Basic(c, x);
        if (x.Has(1)) c["ctrl_cost_weight"_] = x.e[1];
        if (x.Has(2)) c["forward_reward_weight"_] = x.e[2];
        if (x.Has(3)) c["reset_noise_scale"_] = x.e[3];
        if (v5) c["xml_file"_] = std::string("walker2d_v5.xml");
        c["legacy_healthy_reward"_] = !v5;  // gym/registration.py:79-83
        if (x.Get(15, -1) >= 0) c["legacy_healthy_reward"_] = x.e[15] != 0;
      });
    }
    if (t == "Hopper") {
      return new Ref<HopperEnvPool>(num_envs, seed, max_episode_steps, num_threads,
                                    [&](auto& c) {
        Basic(c, x);
        if (x.Has(1)) c["ctrl_cost_weight"_] = x.e[1];
        if (x.Has(2)) c["forward_reward_weight"_] = x.e[2];
        if (x.Has(3)) c["reset_noise_scale"_] = x.e[3];
        if (x.Get(15, -1) >= 0) c["legacy_healthy_reward"_] = x.e[15] != 0;
      });
    }
    if (t == "Swimmer") {
      return new Ref<SwimmerEnvPool>(num_envs, seed, max_episode_steps, num_threads,
                                     [&](auto& c) {
        Basic(c, x);
        if (x.Has(1)) c["ctrl_cost_weight"_] = x.e[1];
        if (x.Has(2)) c["forward_reward_weight"_] = x.e[2];
        if (x.Has(3)) c["reset_noise_scale"_] = x.e[3];
      });
    }
    if (t == "Reacher") {
      return new Ref<ReacherEnvPool>(num_envs, seed, max_episode_steps, num_threads,
                                     [&](auto& c) {
        Basic(c, x);
        if (x.Has(1)) c["ctrl_cost_weight"_] = x.e[1];
        c["reward_after_step"_] = x.Flag(16, false);
        c["obs_include_z_distance"_] = x.Flag(17, true);
      });
    }
    if (t == "Pusher" || t == "PusherV5") {
      const bool v5 = t == "PusherV5";
      return new Ref<PusherEnvPool>(num_envs, seed, max_episode_steps, num_threads,
                                    [&](auto& c) {
        Basic(c, x);
        if (x.Has(1)) c["ctrl_cost_weight"_] = x.e[1];
        if (v5) c["xml_file"_] = std::string("pusher_v5.xml");
        c["reward_after_step"_] = x.Flag(16, false);
        if (x.Has(20)) c["dist_cost_weight"_] = x.e[20];
        if (x.Has(21)) c["near_cost_weight"_] = x.e[21];
        c["weighted_reward_info"_] = x.Flag(22, false);
      });
    }
    if (t == "InvertedPendulum") {
      return new Ref<InvertedPendulumEnvPool>(num_envs, seed, max_episode_steps, num_threads,
                                              [&](auto& c) {
        Basic(c, x);
        if (x.Has(3)) c["reset_noise_scale"_] = x.e[3];
        c["reward_if_not_terminated"_] = x.Flag(10, false);
      });
    }
    if (t == "InvertedDoublePendulum") {
      return new Ref<InvertedDoublePendulumEnvPool>(num_envs, seed, max_episode_steps,
                                                    num_threads, [&](auto& c) {
        Basic(c, x);
        if (x.Has(3)) c["reset_noise_scale"_] = x.e[3];
        c["reward_if_not_terminated"_] = x.Flag(10, false);
        c["constraint_obs_dim"_] = static_cast<int>(x.Get(11, 3));
      });
    }
    if (t == "Humanoid") {
      return new Ref<HumanoidEnvPool>(num_envs, seed, max_episode_steps, num_threads,
                                      [&](auto& c) {
        Basic(c, x);
        if (x.Has(1)) c["ctrl_cost_weight"_] = x.e[1];
        if (x.Has(2)) c["forward_reward_weight"_] = x.e[2];
        if (x.Has(3)) c["reset_noise_scale"_] = x.e[3];
        c["use_contact_force"_] = x.Flag(12, false);
        c["exclude_worldbody_observations"_] = x.Flag(14, false);
        c["exclude_root_actuator_forces"_] = x.Flag(19, false);
        if (x.Get(15, -1) >= 0) c["legacy_healthy_reward"_] = x.e[15] != 0;
      });
    }
    if (t == "HumanoidStandup") {
      return new Ref<HumanoidStandupEnvPool>(num_envs, seed, max_episode_steps, num_threads,
                                             [&](auto& c) {
        Basic(c, x);
        if (x.Has(1)) c["ctrl_cost_weight"_] = x.e[1];
        if (x.Has(2)) c["forward_reward_weight"_] = x.e[2];
        if (x.Has(3)) c["reset_noise_scale"_] = x.e[3];
        c["exclude_worldbody_observations"_] = x.Flag(14, false);
        c["exclude_root_actuator_forces"_] = x.Flag(19, false);
      });
    }
  } catch (const std::exception& e) {
    std::cerr << "orc_create(" << t << "): " << e.what() << std::endl;
  }
  return nullptr;
}

}  // extern "C"
