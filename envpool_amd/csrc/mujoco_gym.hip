// K3 — gym-MuJoCo batched step kernels for the planar robots (HalfCheetah,
// Walker2d, Hopper; one env per thread, one wave per block).
//
// Replaces, for the whole batch in one launch:
//   MujocoEnv::{MujocoReset,MujocoStep}     envpool/mujoco/gym/mujoco_env.h:126-148
//   HalfCheetahEnvBase::{MujocoResetModel,Reset,Step,WriteState}
//                                           envpool/mujoco/gym/half_cheetah.h:105-185
//   Walker2dEnvBase::{...}                  envpool/mujoco/gym/walker2d.h:119-219
//   HopperEnvBase::{...}                    envpool/mujoco/gym/hopper.h:121-230
// including the `frame_skip x mj_step` physics (mj_cheetah.hip.h) and the
// runtime around it (async_envpool.h:118-132, env.h:184-256).
//
// Data layout (HBM): persistent state is SoA float64 — qpos[9][N], qvel[9][N],
// qacc_warmstart[9][N] (+ the saved value of the env's normal_distribution and
// the shared done/cur_step/mt19937 arrays).  Arithmetic runs in T = double, the
// reference's mjtNum (the kernel stays a template over T, but only the fp64
// instantiation is built since round 4: the fp32 mode met neither the 1e-5 bar nor
// the fp64 lane-group kernel's speed).  The root x is carried as a local offset per
// step.  Outputs are written in the reference dtypes (obs/info float64, reward float32).
//
// Not HBM-bound: ~708 algorithmic bytes vs ~5e4 flops per env-step
// => bound by fp64 VALU issue + wave divergence in the Newton iteration count.
#include "device_common.hip.h"
#include "engine.h"
#include "mj_cheetah.hip.h"
#include "mj_cheetah_model.h"
#include "mujoco_planar_common.h"  // CheetahDev, CheetahTask, PlanarModel<T, model>, the generated models

namespace epa {
namespace {

using mj::CheetahModel;
using mj::kNU;
using mj::kNV;
using planar::CheetahDev;
using planar::CheetahTask;
using planar::kCheetahBlock;
using planar::PlanarModel;


template <typename T, int kModel>
__global__ __launch_bounds__(kCheetahBlock) void CheetahStepKernel(
    CheetahDev dev, CommonDev cm, StepArgs a, const double* __restrict__ action,
    OutPtrs out, CheetahTask task, mj::SolverCfg<T> scfg) {
  // the MuJoCo model as compile-time constants (see gen_mj_consts.cpp)
  constexpr CheetahModel<T> m = PlanarModel<T, kModel>();
  constexpr bool kWalker = kModel != mj::kPlanarCheetah;  // Walker2d or Hopper: RK4, mirrored hinges
  constexpr bool kHopper = kModel == mj::kPlanarHopper;
  // the Hopper model only owns dofs 0..5 / motors 0..2 of the tree (ghost second leg)
  constexpr int kNVr = kHopper ? 6 : kNV, kNUr = kHopper ? 3 : kNU;
  // per-contact constants [slot][lane]; read back with a runtime slot index in the
  // solver passes (see DispatchBody) so they stay in LDS instead of VGPRs/scratch
  __shared__ T lds_buf[mj::kLdsSlots * kCheetahBlock];
  const int lane = threadIdx.x;
  const int n = cm.n;
  if (lane >= task.lanes) return;
  const int row = blockIdx.x * task.lanes + lane;
  if (row >= a.k) return;
  const int e = a.ids ? a.ids[row] - a.id_offset : row;
  bool done = cm.done[e] != 0;
  int cur = cm.cur_step[e];
  const bool reset = a.force_reset || done;  // async_envpool.h:127
  double qpos[kNV], qvel[kNV];
  float reward = 0.0f;
  double xv = 0.0, ctrl_cost = 0.0;
  if (reset) {
    // MujocoReset: mj_resetData + MujocoResetModel (half_cheetah.h:105-117);
    // the trailing mj_forward only refreshes qacc_warmstart, which the solver
    // re-derives (unique minimiser), so the warm start is simply cleared.
    cur = 0;
    done = false;
    Mt19937 g(cm, e);
    double saved = dev.nsaved[e];
    int avail = dev.navail[e];
    for (int i = kNVr; i < kNV; ++i) qpos[i] = qvel[i] = 0.0;  // ghost leg
    for (int i = 0; i < kNVr; ++i) {
      // init_qpos = qpos0: all zero but the rootz ref 1.25 (walker2d_envpool.xml:36, hopper :39)
      const double q0 = (kWalker && i == 1) ? 1.25 : 0.0;
      qpos[i] = q0 + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
    }
    for (int i = 0; i < kNVr; ++i) {
      if constexpr (kWalker) {  // walker2d.h:119-126, hopper.h:121-128: uniform noise on qvel too
        qvel[i] = 0.0 + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
      } else {
        qvel[i] = 0.0 + g.Normal(0.0, task.reset_noise_scale, &saved, &avail);
      }
    }
    g.Commit();
    dev.nsaved[e] = saved;
    dev.navail[e] = (unsigned char)avail;
    for (int i = 0; i < kNV; ++i) {
      dev.qpos[(size_t)i * n + e] = qpos[i];
      dev.qvel[(size_t)i * n + e] = qvel[i];
      dev.warm[(size_t)i * n + e] = 0.0;
    }
  } else {
    ++cur;
    T q[kNV], v[kNV], w[kNV], ctrl[kNU];
    mj::static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr double sg = mj::PlanarDofSign(kModel, i);  // hinge about -y: q' = -q
      qpos[i] = dev.qpos[(size_t)i * n + e];
      q[i] = (T)(sg * qpos[i]);
      v[i] = (T)(sg * dev.qvel[(size_t)i * n + e]);
      w[i] = (T)(sg * dev.warm[(size_t)i * n + e]);
    });
    const double x_before = qpos[0];
    q[0] = T(0);
    const double* act = action + (size_t)row * kNUr;
    mj::static_for<0, kNU>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < kNUr) {
        double ai = act[i];
        ctrl_cost += task.ctrl_cost_weight * ai * ai;  // half_cheetah.h:143-146
        // ctrllimited motors: MuJoCo clamps ctrl to ctrlrange [-1, 1]
        ctrl[i] = (T)(ai < -1.0 ? -1.0 : (ai > 1.0 ? 1.0 : ai));
      } else {
        ctrl[i] = T(0);
      }
    });
    auto lds = [&](int slot) -> T& { return lds_buf[slot * kCheetahBlock + lane]; };
    int iters = 0;
#ifdef EPA_WAVE_TRACE  // diagnostic build only (tools/build_trace_lib.sh): the fp64 kernel sits
    // at the 512-register limit and pays 4 % for carrying these values
    int wave_iters = 0;
    const long long c_begin = dev.trace ? clock64() : 0, w_begin = dev.trace ? wall_clock64() : 0;
#endif
    for (int s = 0; s < task.frame_skip; ++s) {  // mujoco_env.h:142-144
      int it;
      if constexpr (kWalker) {
        it = mj::PlanarStepRK4(m, scfg, q, v, w, ctrl, lds);
      } else {
        it = mj::CheetahStep(m, scfg, q, v, w, ctrl, lds);
      }
      iters += it;
#ifdef EPA_WAVE_TRACE
      if (dev.trace) {
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_xor(it, d);
          it = o > it ? o : it;
        }
        wave_iters += it;
      }
#endif
    }
    dev.iters[e] = iters;
#ifdef EPA_WAVE_TRACE
    if (dev.trace && lane == 0) {
      long long* tr = dev.trace + (size_t)blockIdx.x * 6;
      tr[0] = w_begin;
      tr[1] = wall_clock64();
      tr[2] = c_begin;
      tr[3] = clock64();
      tr[4] = wave_iters;
      tr[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
    }
#endif
    const double x_after = x_before + (double)q[0];
    xv = (x_after - x_before) / task.dt;  // half_cheetah.h:148-149
    qpos[0] = x_after;
    mj::static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr double sg = mj::PlanarDofSign(kModel, i);
      if constexpr (i > 0) qpos[i] = sg * (double)q[i];
      qvel[i] = sg * (double)v[i];
      dev.qpos[(size_t)i * n + e] = qpos[i];
      dev.qvel[(size_t)i * n + e] = qvel[i];
      dev.warm[(size_t)i * n + e] = sg * (double)w[i];
    });
    if constexpr (kWalker) {  // walker2d.h:162-177,181-190 / hopper.h:170-203
      bool healthy;
      if constexpr (kHopper) {
        healthy = !(qpos[2] <= task.healthy_angle_min || qpos[2] >= task.healthy_angle_max ||
                    qpos[1] <= task.healthy_z_min);
        for (int i = 2; i < kNVr; ++i) {
          if (qpos[i] <= task.healthy_state_min || qpos[i] >= task.healthy_state_max) healthy = false;
        }
        for (int i = 0; i < kNVr; ++i) {
          if (qvel[i] <= task.healthy_state_min || qvel[i] >= task.healthy_state_max) healthy = false;
        }
      } else {
        healthy = !(qpos[1] < task.healthy_z_min || qpos[1] > task.healthy_z_max ||
                    qpos[2] < task.healthy_angle_min || qpos[2] > task.healthy_angle_max);
      }
      bool give = healthy;
      if (task.legacy_healthy_reward) give = task.terminate_when_unhealthy || healthy;
      const double healthy_reward = give ? task.healthy_reward : 0.0;
      reward = static_cast<float>(xv * task.forward_reward_weight + healthy_reward - ctrl_cost);
      done = (task.terminate_when_unhealthy ? !healthy : false) ||
             cur >= a.max_episode_steps;
    } else {
      reward = static_cast<float>(xv * task.forward_reward_weight - ctrl_cost);
      done = cur >= a.max_episode_steps;  // ++elapsed_step_ >= max_episode_steps_
    }
  }
  cm.done[e] = done ? 1 : 0;
  cm.cur_step[e] = cur;
  // WriteState, half_cheetah.h:158-185
  const int nobs = 2 * kNVr - task.obs_skip;
  // (frame_stack > 1: the engine's generic ring, Pool::EnableObsStack -- the kernel always writes one frame per row)
  {
    double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * nobs;
    for (int i = task.obs_skip; i < kNVr; ++i) *(obs++) = qpos[i];
    for (int i = 0; i < kNVr; ++i) {
      double x = qvel[i];
      if constexpr (kWalker) {  // walker2d.h:210-215
        x = x < task.velocity_max ? x : task.velocity_max;  // std::min(vmax, x)
        x = x > task.velocity_min ? x : task.velocity_min;  // std::max(vmin, x)
      }
      *(obs++) = x;
    }
  }
  if constexpr (kWalker) {  // walker2d.h:218-219
    ((double*)out.p[kKeyEnv0 + 1])[row] = reset ? 0.0 : qpos[0];
    ((double*)out.p[kKeyEnv0 + 2])[row] = xv;
  } else {
    ((double*)out.p[kKeyEnv0 + 1])[row] = xv * task.forward_reward_weight;
    ((double*)out.p[kKeyEnv0 + 2])[row] = -ctrl_cost;
    ((double*)out.p[kKeyEnv0 + 3])[row] = reset ? 0.0 : qpos[0];
    ((double*)out.p[kKeyEnv0 + 4])[row] = xv;
  }
  WriteCommon(out, row, e + a.id_offset, cur, done, reward,
              a.max_episode_steps);
}

// flat state, same layout as oracle/mjcpu: qpos[9] qvel[9] warm[9] time xlag
// ylag done cur_step normal_saved normal_avail
// `nv`: dofs of the model (9, Hopper: 6 -- its ghost dofs are not part of the state)
__global__ void CheetahGetState(CheetahDev dev, CommonDev cm, const int* ids,
                                int k, double* out, int nv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i], n = cm.n;
  double* o = out + (size_t)i * (3 * nv + 7);
  for (int j = 0; j < nv; ++j) {
    o[j] = dev.qpos[(size_t)j * n + e];
    o[nv + j] = dev.qvel[(size_t)j * n + e];
    o[2 * nv + j] = dev.warm[(size_t)j * n + e];
  }
  double* t = o + 3 * nv;
  t[0] = dev.iters[e];  // (oracle: time) Newton iterations of the last step
  t[1] = 0;
  t[2] = 0;
  t[3] = cm.done[e];
  t[4] = cm.cur_step[e];
  t[5] = dev.nsaved[e];
  t[6] = dev.navail[e];
}
__global__ void CheetahSetState(CheetahDev dev, CommonDev cm, const int* ids,
                                int k, const double* in, int nv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i], n = cm.n;
  const double* o = in + (size_t)i * (3 * nv + 7);
  for (int j = 0; j < nv; ++j) {
    dev.qpos[(size_t)j * n + e] = o[j];
    dev.qvel[(size_t)j * n + e] = o[nv + j];
    dev.warm[(size_t)j * n + e] = o[2 * nv + j];
  }
  for (int j = nv; j < kNV; ++j) {  // ghost dofs stay at rest
    dev.qpos[(size_t)j * n + e] = 0.0;
    dev.qvel[(size_t)j * n + e] = 0.0;
    dev.warm[(size_t)j * n + e] = 0.0;
  }
  const double* t = o + 3 * nv;
  cm.done[e] = t[3] != 0.0;
  cm.cur_step[e] = (int)t[4];
  dev.nsaved[e] = t[5];
  dev.navail[e] = t[6] != 0.0;
}

std::vector<KeySpec> CheetahKeys(const Config& cfg, bool walker = false, bool hopper = false) {
  int no_pos = cfg.Get("exclude_current_positions_from_observation", 1) != 0;
  int fs = (int)cfg.Get("frame_stack", 1);
  // half_cheetah.h:44-62, hopper.h:51-63 (non-ENVPOOL_TEST build); StackSpec, frame_stack.h:42-71
  std::vector<int> oshape = {(hopper ? 12 : 18) - (no_pos ? 1 : 0)};
  if (fs > 1) oshape.insert(oshape.begin(), fs);
  if (walker) {  // walker2d.h:49-62
    return {{"obs", EPA_F64, oshape},
            {"info:x_position", EPA_F64, {}},
            {"info:x_velocity", EPA_F64, {}}};
  }
  return {{"obs", EPA_F64, oshape},
          {"info:reward_run", EPA_F64, {}},
          {"info:reward_ctrl", EPA_F64, {}},
          {"info:x_position", EPA_F64, {}},
          {"info:x_velocity", EPA_F64, {}}};
}

class CheetahPool : public Pool {
 public:
  bool ConcurrentSafe() const override { return true; }  // per-env state + the launch's own block only
  // model: mj::kPlanarCheetah / kPlanarWalker / kPlanarWalkerV5 / kPlanarHopper
  CheetahPool(const Config& cfg, int model)
      : Pool(cfg, CheetahKeys(cfg, model != mj::kPlanarCheetah, model == mj::kPlanarHopper),
             KeySpec{"action", EPA_F64, {model == mj::kPlanarHopper ? 3 : kNU}},
             /*needs_rng=*/true),
        model_id_(model) {
    const bool walker = model != mj::kPlanarCheetah;  // Walker2d or Hopper
    const bool hopper = model == mj::kPlanarHopper;
    task_.frame_stack = (int)cfg.Get("frame_stack", 1);
    if (task_.frame_stack < 1) {
      throw std::invalid_argument("frame_stack must be greater than 0");
    }
    // "precision": 1 = fp64 arithmetic, the reference's mjtNum = double.  The fp32 arithmetic mode
    // ("precision" = 0) of rounds 1-3 is gone: on a cond ~ 1e4 Newton system it missed the 1e-5 bar
    // (p99 |d obs| 1.6e-4) and, since the lane-group fp64 kernel, it was also the SLOWER one
    // (2.91e8 vs 3.04e8 env-steps/s at N = 65536) -- neither parity nor speed.
    if ((int)cfg.Get("precision", 1) != 1) {
      throw std::invalid_argument("\"precision\" = 0 (fp32 arithmetic) was removed for the planar MuJoCo "
                                  "families (HalfCheetah, Walker2d, Hopper); only 1 (fp64, the reference's "
                                  "mjtNum) is accepted");
    }
    // defaults: half_cheetah.h:33-43 / walker2d.h:32-47
    task_.frame_skip = (int)cfg.Get("frame_skip", walker ? 4 : 5);
    task_.obs_skip =
        cfg.Get("exclude_current_positions_from_observation", 1) != 0 ? 1 : 0;
    task_.ctrl_cost_weight = cfg.Get("ctrl_cost_weight", walker ? 0.001 : 0.1);
    task_.forward_reward_weight = cfg.Get("forward_reward_weight", 1.0);
    task_.reset_noise_scale = cfg.Get("reset_noise_scale", walker ? 0.005 : 0.1);
    task_.dt = task_.frame_skip * (walker ? kWalkerModelConst.timestep
                                          : kCheetahModelConst.timestep);
    task_.healthy_reward = cfg.Get("healthy_reward", 1.0);
    task_.healthy_z_min = cfg.Get("healthy_z_min", hopper ? 0.7 : 0.8);
    task_.healthy_z_max = cfg.Get("healthy_z_max", 2.0);
    task_.healthy_angle_min = cfg.Get("healthy_angle_min", hopper ? -0.2 : -1.0);
    task_.healthy_angle_max = cfg.Get("healthy_angle_max", hopper ? 0.2 : 1.0);
    task_.healthy_state_min = cfg.Get("healthy_state_min", -100.0);
    task_.healthy_state_max = cfg.Get("healthy_state_max", 100.0);
    task_.velocity_min = cfg.Get("velocity_min", -10.0);
    task_.velocity_max = cfg.Get("velocity_max", 10.0);
    task_.terminate_when_unhealthy = cfg.Get("terminate_when_unhealthy", 1) != 0;
    task_.legacy_healthy_reward = cfg.Get("legacy_healthy_reward", 1) != 0;
    size_t n = cfg.num_envs;
    EPA_HIP(hipMalloc(&dev_.qpos, sizeof(double) * kNV * n));
    EPA_HIP(hipMalloc(&dev_.qvel, sizeof(double) * kNV * n));
    EPA_HIP(hipMalloc(&dev_.warm, sizeof(double) * kNV * n));
    EPA_HIP(hipMalloc(&dev_.nsaved, sizeof(double) * n));
    EPA_HIP(hipMalloc(&dev_.navail, n));
    EPA_HIP(hipMalloc(&dev_.iters, sizeof(int) * n));
    trace_.Init("EPA_PLANAR_TRACE", (n + kCheetahBlock - 1) / kCheetahBlock, stream_);
    dev_.trace = trace_.d;
    spread_ = cfg.Get("planar_spread", 1) != 0;  // extension key, see Launch
    // "planar_layout" (extension key): 1 = one env per lane (CheetahStepKernel), 2 / 4 = one env per
    // group of 2 / 4 lanes (mujoco_planar_lg.hip; fp64, HalfCheetah / Walker2d, frame_stack 1),
    // 0 (default) = chosen HERE from the rows this pool normally has in flight (num_envs in sync mode,
    // min(num_envs, 4 x batch_size) in async mode, see below): 4 lanes per env while one round of 16-env chunks
    // covers the rows (16384 = 1024 SIMDs x 16), 2 above (round 5, profiles/r5aa_layout_sweep.txt: N = 16384
    // 2.17e8 with 4 lanes vs 1.84e8 with 2, N = 20480 1.57e8 vs 2.30e8; rounds 3-4 switched at 24576).  Fixed per pool, not per launch: the two
    // layouts sum the contact rows in different orders, and an env's bits must not depend on how
    // many other envs a particular send happens to carry.  It DOES depend on num_envs / batch_size of
    // the pool (to rounding: the two layouts agree with the oracle to 1e-9 each, tests/test_gpu_mujoco.py);
    // pass planar_layout explicitly where two pools of different shape must agree bit for bit.
    // Hopper (one leg): 0 (default) = the lane-group kernel with a group of ONE lane -- the lane's 6 local
    // dofs (torso + leg) are the whole robot: packed 6 x 6 instead of the ghost-leg 9 x 9 of CheetahStepKernel,
    // no scratch; 1 = CheetahStepKernel as for the others; 2 / 4 do not exist.
    layout_ = (int)cfg.Get("planar_layout", 0);
    if (layout_ != 0 && layout_ != 1 && layout_ != 2 && layout_ != 4) {
      throw std::invalid_argument("planar_layout must be 0, 1, 2 or 4");
    }
    if (hopper && layout_ > 1) throw std::invalid_argument("planar_layout: the Hopper takes 0 or 1");
    // frame_stack > 1: the generic ring of the engine (EnableObsStack; the kernels write one frame per row), so
    // that stacked and plain pools run the same step kernel and agree bit for bit (the in-kernel ring the
    // one-env-per-lane kernel had in rounds 1-3 went with the fp32 mode that was its last user)
    if (task_.frame_stack > 1) {
      EnableObsStack();
      task_.frame_stack = 1;
    }
    lg_ok_ = true;
    // (async mode: several batches are in flight on the pool's compute streams, so what fills the machine is
    // batch_size x streams, not one batch: 8 x 8192 of 65536 envs measured 2.83e8 with 2 lanes per env against
    // 2.06e8 with 4, profiles/archive/r3r_async_probe.jsonl; and partly filled waves -- made to occupy more SIMDs with ONE
    // small batch -- only take lanes from the other batches there)
    async_ = cfg.batch_size > 0 && cfg.batch_size < cfg.num_envs;
    if (layout_ == 0) {
      long long rows = cfg.num_envs;
      // (4 = the default number of compute streams; NOT the pool's own "compute_streams", so that an env's bits
      // do not depend on that knob: tests compare 1 stream with several bit for bit)
      if (async_) rows = std::min<long long>(cfg.num_envs, 4ll * cfg.batch_size);
      layout_ = rows > 16384 ? 2 : 4;
      if (hopper) layout_ = kLayoutHopperLg;
    }
    // register budget of the lane-group kernel: one wave per SIMD with all 512 registers (default;
    // measured faster at every batch size, profiles/archive/r3f_lane_group_sweep.txt) or two with 256 + spills
    lg_waves_ = (int)cfg.Get("planar_waves", 1) == 2 ? 2 : 1;
    // longest-first dispatch by the previous launch's timing: on by default for the RK4 models only.  With the
    // round-5 solver a HalfCheetah chunk's duration no longer says anything about the next one (host model:
    // correlation 0.06) and the bookkeeping costs more than the order gives: N = 65536 4.05e8 with, 4.14e8 without;
    // Walker2d 1.733e8 with, 1.720e8 without (profiles/r5ab_lpt_ab.txt)
    lpt_ = cfg.Get("planar_lpt", walker ? 1 : 0) != 0;
    if (lg_ok_) {
      for (int i = 0; i < 2; ++i) {
        if (hopper && i == 1) break;  // one table: a group of one lane
        std::vector<double> tab(kPlanarLgTabMax, 0.0);
        const int cnt = PlanarLgBuildTable(hopper ? 1 : (i == 0 ? 2 : 4), model_id_, tab.data());
        EPA_HIP(hipMalloc(&d_tab_[i], sizeof(double) * cnt));
        EPA_HIP(hipMemcpy(d_tab_[i], tab.data(), sizeof(double) * cnt, hipMemcpyHostToDevice));
      }
    }
    {
      hipDeviceProp_t prop;
      EPA_HIP(hipGetDeviceProperties(&prop, cfg.device));
      wave_slots_ = prop.multiProcessorCount * 4;  // one wave per SIMD, four SIMDs per CU
      if (wave_slots_ < 1) wave_slots_ = 1;
    }
    EPA_HIP(hipMemsetAsync(dev_.iters, 0, sizeof(int) * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.qpos, 0, sizeof(double) * kNV * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.qvel, 0, sizeof(double) * kNV * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.warm, 0, sizeof(double) * kNV * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.nsaved, 0, sizeof(double) * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.navail, 0, n, stream_));
    // Walker2d / Hopper terminate when unhealthy, each env at its own time: tiled generator words; the
    // HalfCheetah never terminates early (all envs draw in the same launch)
    if (walker) mt_tile_default_ = 16;
    // A whole-pool host-path step as two half launches (Pool::SendPipelined): pays where half the rows take about
    // half the time, i.e. with 2 lanes per env, where 65536 rows are two rounds of chunks (HalfCheetah 1.09e8 ->
    // 1.3e8, Walker2d 8.1e7 -> 9.2e7 env-steps/s through the numpy API); a Hopper launch (64 envs per wave) or a
    // Pusher launch is ONE round at that size, and two half launches take twice as long (profiles/r6g_numpy_step_ab.txt)
    if (layout_ == 2) pipeline_default_ = 32768;
    InitCommon();
  }
  ~CheetahPool() override {
    trace_.DumpAndFree();
    (void)hipFree(dev_.qpos);
    (void)hipFree(dev_.qvel);
    (void)hipFree(dev_.warm);
    (void)hipFree(dev_.nsaved);
    (void)hipFree(dev_.navail);
    (void)hipFree(dev_.iters);
    for (double* t : d_tab_) {
      if (t) (void)hipFree(t);
    }
    for (auto& kv : tickets_) (void)hipFree(kv.second.d);
    if (order_.d) (void)hipFree(order_.d);
  }
  int ModelNv() const { return model_id_ == mj::kPlanarHopper ? 6 : kNV; }
  int StateDim() const override { return 3 * ModelNv() + 7; }
  void GetState(const int* d_ids, int k, double* d_out) override {
    hipLaunchKernelGGL(CheetahGetState, dim3((k + 255) / 256), dim3(256), 0,
                       stream_, dev_, common_, d_ids, k, d_out, ModelNv());
  }
  void SetState(const int* d_ids, int k, const double* d_in) override {
    hipLaunchKernelGGL(CheetahSetState, dim3((k + 255) / 256), dim3(256), 0,
                       stream_, dev_, common_, d_ids, k, d_in, ModelNv());
  }

 protected:
  void Launch(const int* d_ids, int k, const void* d_action, bool force_reset,
              const OutPtrs& out) override {
    StepArgs a{d_ids, k, force_reset ? 1 : 0, cfg_.max_episode_steps,
               cfg_.env_id_offset};
    // A wave runs as long as its slowest lane and visits every end sphere that touches on ANY of its
    // lanes, and these kernels hold one wave per SIMD: a batch between 16 and 64 envs per SIMD is spread
    // over all SIMDs with 16 / 32 / 48 envs per wave (HalfCheetah N = 16384 .. 49152: +5 .. +8 %,
    // profiles/archive/r2ze_planar_spread.txt).  Not below 16 per SIMD: there partially filled waves measured
    // SLOWER (N = 8192 as 512 waves of 16: 0.26 ms against 0.21 ms as 128 full waves).
    // One env per lane group (mj_planar_lg.hip.h) wherever it applies.
    int layout = lg_ok_ && trace_.d == nullptr ? layout_ : 1;
    if (layout > 1) {
      Ticket& tk = tickets_[stream_];  // launches on different streams run concurrently: a queue each
      if (tk.d == nullptr) {
        EPA_HIP(hipMalloc(&tk.d, sizeof(unsigned)));
        EPA_HIP(hipMemsetAsync(tk.d, 0, sizeof(unsigned), stream_));
      }
      // longest-chunk-first dispatch: whole-pool batches only (the chunks are then the same envs
      // from launch to launch); "planar_lpt" = 0 switches it off (A/B)
      planar::LgOrder lo;
      const bool chain = lpt_ && d_ids == nullptr && !force_reset;
      const int shape = k * 16 + layout;  // (only its sameness from launch to launch matters)
      if (chain) {
        if (order_.d == nullptr) {
          order_.cap = (cfg_.num_envs + 3) / 4;  // the smallest chunk is a quarter wave of 4 lanes per env
          EPA_HIP(hipMalloc(&order_.d, PlanarLgOrderBytes(order_.cap)));
          EPA_HIP(hipMemsetAsync(order_.d, 0, PlanarLgOrderBytes(order_.cap), stream_));
          order_.gen = 0;
          order_shape_ = -1;
        }
        order_.use = (shape == order_shape_ && order_stream_ == stream_) ? 1 : 0;
        lo = order_;
      }
      const bool lg1 = layout == kLayoutHopperLg;  // lanes per env: 1 (Hopper), 2 or 4
      const bool filed =
          PlanarLgLaunch(stream_, lg1 ? 1 : layout, lg_waves_, model_id_, wave_slots_, spread_ && !async_, dev_, common_,
                         a, static_cast<const double*>(d_action), out, task_, d_tab_[(lg1 || layout == 2) ? 0 : 1], tk.d,
                         &tk.base, lo);
      // A generation is advanced only by a launch that filed its chunks into gen + 1 AND cleared gen + 2 (the
      // kernel skips both while chunks do not queue for waves): every increment comes with a clear, or a later
      // launch would file into a buffer that still holds an older launch's chunk lists (chunks dispatched twice /
      // never).  Any other launch breaks the chain of same-shape launches.
      if (chain && filed) {
        ++order_.gen;
        order_shape_ = shape;
        order_stream_ = stream_;
      } else {
        order_shape_ = -1;
      }
      return;
    }
    int lanes = kCheetahBlock;
    if (spread_ && trace_.d == nullptr && k >= 16 * wave_slots_) {
      lanes = ((k + wave_slots_ - 1) / wave_slots_ + 15) / 16 * 16;
      lanes = lanes > kCheetahBlock ? kCheetahBlock : lanes;
    }
    task_.lanes = lanes;
    int blocks = (k + lanes - 1) / lanes;
    const double* act = static_cast<const double*>(d_action);
    const mj::SolverCfg<double> sd{50, 1e-13};
#define EPA_LAUNCH_PLANAR(T, MODEL, SC)                                            \
  hipLaunchKernelGGL((CheetahStepKernel<T, MODEL>), dim3(blocks), dim3(kCheetahBlock), \
                     0, stream_, dev_, common_, a, act, out, task_, SC)
    switch (model_id_) {
      case mj::kPlanarCheetah: EPA_LAUNCH_PLANAR(double, mj::kPlanarCheetah, sd); break;
      case mj::kPlanarWalker: EPA_LAUNCH_PLANAR(double, mj::kPlanarWalker, sd); break;
      case mj::kPlanarWalkerV5: EPA_LAUNCH_PLANAR(double, mj::kPlanarWalkerV5, sd); break;
      default: EPA_LAUNCH_PLANAR(double, mj::kPlanarHopper, sd); break;
    }
#undef EPA_LAUNCH_PLANAR
  }

 private:
  CheetahDev dev_{};
  WaveTrace trace_;
  int model_id_;
  CheetahTask task_{};
  bool spread_{true};
  bool async_{false};
  int wave_slots_{1024};
  static constexpr int kLayoutHopperLg = 8;  // layout_: the Hopper on the lane-group kernel (group of one lane)
  int layout_{0};
  bool lg_ok_{false};
  int lg_waves_{1};
  double* d_tab_[2] = {nullptr, nullptr};
  struct Ticket {
    unsigned* d{nullptr};
    unsigned base{0};
  };
  std::map<hipStream_t, Ticket> tickets_;  // chunk queue of the lane-group kernel, per launch stream
  planar::LgOrder order_;
  int order_shape_{-1};
  hipStream_t order_stream_{nullptr};
  bool lpt_{true};
};

}  // namespace

bool DescribeAnt(const std::string& family, const Config& cfg,
                 std::vector<KeySpec>* state, KeySpec* action);
Pool* MakeAnt(const std::string& family, const Config& cfg);
bool DescribePendulum(const std::string& family, const Config& cfg,
                      std::vector<KeySpec>* state, KeySpec* action);
Pool* MakePendulum(const std::string& family, const Config& cfg);
bool DescribeHumanoid(const std::string& family, const Config& cfg,
                      std::vector<KeySpec>* state, KeySpec* action);
Pool* MakeHumanoid(const std::string& family, const Config& cfg);
bool DescribePusher(const std::string& family, const Config& cfg,
                    std::vector<KeySpec>* state, KeySpec* action);
Pool* MakePusher(const std::string& family, const Config& cfg);

bool DescribeMujoco(const std::string& family, const Config& cfg,
                    std::vector<KeySpec>* state, KeySpec* action) {
  if (family == "HalfCheetah" || family == "Walker2d" || family == "Hopper") {
    *state = CheetahKeys(cfg, family != "HalfCheetah", family == "Hopper");
    *action = KeySpec{"action", EPA_F64, {family == "Hopper" ? 3 : kNU}};
    return true;
  }
  if (DescribePendulum(family, cfg, state, action)) return true;
  if (DescribeHumanoid(family, cfg, state, action)) return true;
  if (DescribePusher(family, cfg, state, action)) return true;
  return DescribeAnt(family, cfg, state, action);
}

Pool* MakeMujoco(const std::string& family, const Config& cfg) {
  if (family == "HalfCheetah") return new CheetahPool(cfg, mj::kPlanarCheetah);
  if (family == "Hopper") return new CheetahPool(cfg, mj::kPlanarHopper);
  if (family == "Walker2d") {
    // "xml_v5" = 1: walker2d_v5.xml (gym/registration.py:79-83)
    return new CheetahPool(cfg, cfg.Get("xml_v5", 0) != 0 ? mj::kPlanarWalkerV5
                                                         : mj::kPlanarWalker);
  }
  if (Pool* p = MakePendulum(family, cfg)) return p;
  if (Pool* p = MakeHumanoid(family, cfg)) return p;
  if (Pool* p = MakePusher(family, cfg)) return p;
  return MakeAnt(family, cfg);
}

}  // namespace epa
