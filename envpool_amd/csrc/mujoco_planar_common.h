// Shared between the two translation units of the planar gym robots: mujoco_gym.hip (one env per
// lane: HalfCheetah / Walker2d / Hopper, fp32 and fp64) and mujoco_planar_lg.hip (one env per lane
// group: HalfCheetah / Walker2d, fp64).  Device state, task parameters, the compile-time models.
#ifndef ENVPOOL_AMD_CSRC_MUJOCO_PLANAR_COMMON_H_
#define ENVPOOL_AMD_CSRC_MUJOCO_PLANAR_COMMON_H_

#include "device_common.hip.h"
#include "engine.h"
#include "mj_cheetah.hip.h"
#include "mj_cheetah_model.h"
#include "build/mj_cheetah_consts.inc"  // generated: kCheetahModelConst (gen_mj_consts.cpp)
#include "build/mj_walker_consts.inc"   // generated: kWalkerModelConst, kWalkerV5ModelConst, kHopperModelConst

namespace epa {
namespace planar {

using mj::CheetahModel;

struct CheetahDev {
  double* qpos;  // [9][N]
  double* qvel;  // [9][N]
  double* warm;  // [9][N]
  int* iters;              // Newton iterations of the last step (profiling)
  double* nsaved;          // normal_distribution::_M_saved
  unsigned char* navail;   // normal_distribution::_M_saved_available
  // diagnostic (EPA_PLANAR_TRACE=<file>): per wave of the last launch {wall clock begin,
  // end (100 MHz), core clock begin, end, Newton iterations the wave executed (sum over
  // mj_steps of the slowest lane's count), HW_ID}; nullptr otherwise
  long long* trace;
};

struct CheetahTask {
  int frame_skip;
  int frame_stack;  // always 1 for the kernels: TypedFrameStackBuffer (envpool/mujoco/frame_stack.h:74-146) is Pool::EnableObsStack
  int obs_skip;  // 1 if exclude_current_positions_from_observation
  double ctrl_cost_weight, forward_reward_weight, reset_noise_scale;
  double dt;     // frame_skip * timestep, computed in fp64 like the reference
  // Walker2d (walker2d.h:32-47) and Hopper (hopper.h:32-49; healthy_z_max unused there)
  double healthy_reward, healthy_z_min, healthy_z_max, healthy_angle_min,
      healthy_angle_max, velocity_min, velocity_max, healthy_state_min, healthy_state_max;
  int terminate_when_unhealthy, legacy_healthy_reward;
  int lanes;  // envs per wave of THIS launch (64 once the batch fills every SIMD; see Launch)
};

// compile-time model of the planar kernel instance (mj_cheetah.hip.h, PlanarModelId)
template <typename T, int kModel>
constexpr CheetahModel<T> PlanarModel() {
  if constexpr (kModel == mj::kPlanarCheetah) {
    return mj::CastCheetahModel<T>(kCheetahModelConst);
  } else if constexpr (kModel == mj::kPlanarWalker) {
    return mj::CastCheetahModel<T>(kWalkerModelConst);
  } else if constexpr (kModel == mj::kPlanarHopper) {
    return mj::CastCheetahModel<T>(kHopperModelConst);
  } else {
    return mj::CastCheetahModel<T>(kWalkerV5ModelConst);
  }
}

constexpr int kCheetahBlock = 64;

// longest-chunk-first dispatch state of the lane-group kernel (mujoco_planar_lg.hip): `d` = device
// block of PlanarLgOrderBytes(cap) bytes (zeroed), cap >= chunks of any launch that uses it;
// gen = launches that used it so far; use = this launch follows one of the same shape
struct LgOrder {
  unsigned* d{nullptr};
  int cap{0};
  int gen{0};
  int use{0};
};

}  // namespace planar

// mujoco_planar_lg.hip: one env per group of `kl` lanes -- 2 or 4 for HalfCheetah / Walker2d, 1 for the Hopper (and
// only it); `tab` = the device copy of mj::plg::BuildTable<kl>(model); model = mj::PlanarModelId; frame_stack must be 1;
// waves = 1 or 2: the kernel variant whose register allocation aims at that many waves per SIMD
// `wave_slots`: SIMDs of the device; `ticket` / `ticket_base`: the chunk queue's device counter and the
// host's copy of its value (one pair per stream that launches concurrently, see PlanarLgStepKernel)
// spread: partly filled waves while there are fewer full chunks than resident waves ("planar_spread")
// Returns whether the launch took part in the longest-first protocol of `order` (it skips it while chunks do not
// queue for waves): only then has generation order.gen + 2 been cleared, and only then may the host advance gen.
bool PlanarLgLaunch(hipStream_t st, int kl, int waves, int model, int wave_slots, bool spread,
                    const planar::CheetahDev& dev,
                    const CommonDev& cm, const StepArgs& a, const double* action, const OutPtrs& out,
                    const planar::CheetahTask& task, const double* tab, unsigned* ticket, unsigned* ticket_base,
                    const planar::LgOrder& order);
size_t PlanarLgOrderBytes(int cap);
// fills `tab` (host) for lane groups of `kl`; returns the number of doubles (<= kPlanarLgTabMax)
constexpr int kPlanarLgTabMax = 512;
int PlanarLgBuildTable(int kl, int model, double* tab);

}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MUJOCO_PLANAR_COMMON_H_
