// Shared by the Humanoid translation units (mujoco_humanoid.hip built twice, mujoco_humanoid4.hip).
#ifndef ENVPOOL_AMD_CSRC_MUJOCO_HUMANOID_COMMON_H_
#define ENVPOOL_AMD_CSRC_MUJOCO_HUMANOID_COMMON_H_

#include "device_common.hip.h"
#include "engine.h"

namespace epa {

struct HumDev {
  double* ws;     // [ceil(N / 64)][Layout::total][64]: block b belongs to wave b of a launch
  double* state;  // [Layout::npersist][N]: what persists between steps, per env
  // quad kernel, cost-sorted scheduling: an env's solver cost of its last step, and the launch's
  // row order (slot -> row) that puts envs of similar cost into the same wave
  int* cost;  // [N]
  int* perm;  // [N], nullptr: rows in order
};

struct HumTask {
  int frame_skip, obs_skip;
  int terminate_when_unhealthy, legacy_healthy_reward;
  int use_contact_force, post_constraint, exclude_worldbody, exclude_root_actuator;
  double ctrl_cost_weight, forward_reward_weight, healthy_reward;
  double healthy_z_min, healthy_z_max, reset_noise_scale, dt;
  double contact_cost_weight, contact_cost_max;
  int debug;  // timing builds only ("hum_debug"): stages of the quad kernel switched off
};


// mujoco_humanoid4.hip: the one-env-per-lane-quad kernels (mj_hum4.hip.h)
void Hum4LaunchStep(hipStream_t st, bool standup, int blocks, HumDev dev, CommonDev cm, StepArgs a,
                    const double* act, OutPtrs out, HumTask task);
size_t Hum4WorkspaceBytes(int num_envs);
// fills dev.perm[0..k) with the rows 0..k-1 ordered by dev.cost of their envs (stable, deterministic)
void Hum4LaunchSort(hipStream_t st, HumDev dev, StepArgs a);

}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MUJOCO_HUMANOID_COMMON_H_
