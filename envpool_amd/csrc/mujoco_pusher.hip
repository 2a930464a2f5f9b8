// K6 — gym-MuJoCo Pusher batched step kernel (one env per thread).
//
// Replaces, for the whole batch in one launch:
//   MujocoEnv::{MujocoReset,MujocoStep}   envpool/mujoco/gym/mujoco_env.h:126-148
//   PusherEnvBase::{MujocoResetModel,Reset,Step,GetDist,WriteState}
//                                         envpool/mujoco/gym/pusher.h:115-224
// with `frame_skip x mj_step` (Euler) from mj_pusher.hip.h.  Pusher-v2 / v4: pusher.xml;
// Pusher-v5: pusher_v5.xml (object without the sphere, cylinder density 0.01),
// reward_after_step and weighted_reward_info (gym/registration.py:68-73).  post_constraint
// (mj_rnePostConstraint after the steps) changes nothing that is observed.
//
// Persistent state (SoA fp64): qpos[11][N], qvel[11][N], qacc_warmstart[11][N] (entries 9,
// 10: the goal's slides -- constants of an episode, zero velocity) and lag[5][N] = xpos of
// tips_arm (3) and of the object (x, y) at the LAST forward evaluation: GetDist and the
// observation read data_->xpos, which mj_step leaves at the state before its integration
// (pusher.h:190-224).
#include "device_common.hip.h"
#include "engine.h"
#include "mj_pusher.hip.h"
#include "build/mj_pusher_consts.inc"  // generated: kPusherModelConst, kPusherV5ModelConst (gen_mj_consts.cpp)

namespace epa {
namespace {

namespace PU = mj::pusher;

struct PusherDev {
  double* qpos;  // [11][N]
  double* qvel;  // [11][N]
  double* warm;  // [11][N]
  double* lag;   // [5][N]
};

struct PusherTask {
  int frame_skip, reward_after_step, weighted_reward_info;
  double ctrl_cost_weight, dist_cost_weight, near_cost_weight, reset_qvel_scale;
  double cyl_x_min, cyl_x_max, cyl_y_min, cyl_y_max, cyl_dist_min;
  int lanes;  // envs per wave of this launch (mujoco_gym.hip: planar_spread)
};

constexpr int kPusherBlock = 64;
constexpr int kPusherStateDim = 3 * PU::kNQ + 7 + 5;

// The model is a compile-time constant (immediates / folded arithmetic): as a kernel argument
// its 170 doubles live in -- and spill out of -- SGPRs, and ROCm 7.2 builds of this kernel with
// ~250 SGPR spills returned wrong joint-limit forces (see the Makefile note on MJFLAGS).
template <bool kV5>
__global__ __launch_bounds__(kPusherBlock) void PusherStepKernel(
    PusherDev dev, CommonDev cm, StepArgs a, const double* __restrict__ action, OutPtrs out,
    PusherTask task, mj::SolverCfg<double> scfg) {
  constexpr PU::PusherModel<double> m = kV5 ? kPusherV5ModelConst : kPusherModelConst;
  // Jacobians of the constraint rows that touch somewhere in the wave, [slot][lane] (mj_pusher.hip.h, Row)
  __shared__ double row_lds[PU::kRowSlots * kPusherBlock];
  auto lds = [&](int slot) -> double& { return row_lds[slot * kPusherBlock + threadIdx.x]; };
  const int n = cm.n;
  if ((int)threadIdx.x >= task.lanes) return;
  const int row = blockIdx.x * task.lanes + threadIdx.x;
  if (row >= a.k) return;
  const int e = a.ids ? a.ids[row] - a.id_offset : row;
  bool done = cm.done[e] != 0;
  int cur = cm.cur_step[e];
  const bool reset = a.force_reset || done;  // async_envpool.h:127
  double q[PU::kNV], v[PU::kNV], w[PU::kNV], gy, gx;  // gy, gx: goal_slidey / goal_slidex
  PU::PusherLag<double> lag;
  float reward = 0.0f;
  // the reset WriteState stores the NEGATED +0.0 costs: -0.0 in all three keys (pusher.h:225-232)
  double info[3] = {-0.0, -0.0, -0.0};
  if (reset) {  // MujocoReset + MujocoResetModel, pusher.h:115-136
    cur = 0;
    done = false;
    Mt19937 g(cm, e);
    for (int i = 0; i < PU::kNL; ++i) q[i] = 0.0;  // init_qpos_ = qpos0
    for (;;) {  // qpos[nq - 4] = x, qpos[nq - 3] = y: the reference writes x into the y slide
      const double x = g.UniformReal(task.cyl_x_min, task.cyl_x_max);
      const double y = g.UniformReal(task.cyl_y_min, task.cyl_y_max);
      if (sqrt(x * x + y * y) > task.cyl_dist_min) {
        q[7] = x;
        q[8] = y;
        break;
      }
    }
    gy = gx = 0.0;
    for (int i = 0; i < PU::kNL; ++i) v[i] = 0.0 + g.UniformReal(-task.reset_qvel_scale, task.reset_qvel_scale);
    v[7] = v[8] = 0.0;
    g.Commit();
    for (int i = 0; i < PU::kNV; ++i) w[i] = 0.0;
    {  // mj_forward: xpos of the reset state (the warm start it leaves is re-derived anyway)
      double qacc[PU::kNV], M[PU::kNV * PU::kNV], f[PU::kNV], ww[PU::kNV];
      const double zero[PU::kNL] = {0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < PU::kNV; ++i) ww[i] = 0.0;
      PU::PusherForward(m, scfg, q, v, zero, ww, qacc, M, f, &lag, lds);
    }
  } else {
    ++cur;
    mj::static_for<0, PU::kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      q[i] = dev.qpos[(size_t)i * n + e];
      v[i] = dev.qvel[(size_t)i * n + e];
      w[i] = dev.warm[(size_t)i * n + e];
    });
    gy = dev.qpos[(size_t)9 * n + e];
    gx = dev.qpos[(size_t)10 * n + e];
    mj::static_for<0, 3>([&](auto ic) { lag.tips[decltype(ic)::value] = dev.lag[(size_t)decltype(ic)::value * n + e]; });
    lag.obj[0] = dev.lag[(size_t)3 * n + e];
    lag.obj[1] = dev.lag[(size_t)4 * n + e];
    const double goal[3] = {m.goal_pos[0] + gx, m.goal_pos[1] + gy, m.goal_pos[2]};
    auto dists = [&](double* near_cost, double* dist_cost) {  // GetDist, pusher.h:190-195
      const double ax = lag.obj[0] - lag.tips[0], ay = lag.obj[1] - lag.tips[1], az = m.obj_pos[2] - lag.tips[2];
      *near_cost = sqrt(ax * ax + ay * ay + az * az);
      const double bx = lag.obj[0] - goal[0], by = lag.obj[1] - goal[1], bz = m.obj_pos[2] - goal[2];
      *dist_cost = sqrt(bx * bx + by * by + bz * bz);
    };
    double near_cost = 0.0, dist_cost = 0.0;
    if (!task.reward_after_step) dists(&near_cost, &dist_cost);
    double act[PU::kNL], ctrl_cost = 0.0;
    mj::static_for<0, PU::kNL>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      act[i] = action[(size_t)row * PU::kNL + i];
      ctrl_cost += act[i] * act[i];  // pusher.h:171-175
    });
    for (int s = 0; s < task.frame_skip; ++s) PU::PusherStep(m, scfg, q, v, w, act, &lag, lds);
    if (task.reward_after_step) dists(&near_cost, &dist_cost);
    reward = static_cast<float>(-ctrl_cost * task.ctrl_cost_weight - dist_cost * task.dist_cost_weight -
                                near_cost * task.near_cost_weight);
    done = cur >= a.max_episode_steps;
    info[0] = -dist_cost * (task.weighted_reward_info ? task.dist_cost_weight : 1.0);
    info[1] = -ctrl_cost * (task.weighted_reward_info ? task.ctrl_cost_weight : 1.0);
    info[2] = -near_cost * task.near_cost_weight;
  }
  mj::static_for<0, PU::kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    dev.qpos[(size_t)i * n + e] = q[i];
    dev.qvel[(size_t)i * n + e] = v[i];
    dev.warm[(size_t)i * n + e] = w[i];
  });
  if (reset) {
    for (int i = 9; i < PU::kNQ; ++i) {
      dev.qpos[(size_t)i * n + e] = 0.0;
      dev.qvel[(size_t)i * n + e] = 0.0;
      dev.warm[(size_t)i * n + e] = 0.0;
    }
  }
  mj::static_for<0, 3>([&](auto ic) { dev.lag[(size_t)decltype(ic)::value * n + e] = lag.tips[decltype(ic)::value]; });
  dev.lag[(size_t)3 * n + e] = lag.obj[0];
  dev.lag[(size_t)4 * n + e] = lag.obj[1];
  cm.done[e] = done ? 1 : 0;
  cm.cur_step[e] = cur;
  // WriteState, pusher.h:197-224
  double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * 23;
  for (int i = 0; i < 7; ++i) obs[i] = q[i];
  for (int i = 0; i < 7; ++i) obs[7 + i] = v[i];
  for (int i = 0; i < 3; ++i) obs[14 + i] = lag.tips[i];
  obs[17] = lag.obj[0];
  obs[18] = lag.obj[1];
  obs[19] = m.obj_pos[2];
  obs[20] = m.goal_pos[0] + gx;
  obs[21] = m.goal_pos[1] + gy;
  obs[22] = m.goal_pos[2];
  for (int i = 0; i < 3; ++i) ((double*)out.p[kKeyEnv0 + 1 + i])[row] = info[i];
  WriteCommon(out, row, e + a.id_offset, cur, done, reward, a.max_episode_steps);
}

// flat state like oracle/mjcpu: qpos[11] qvel[11] warm[11] time xlag ylag done cur_step 0 0
// + xpos of tips_arm (3) and object (x, y) of the last forward evaluation
__global__ void PusherGetState(PusherDev dev, CommonDev cm, const int* ids, int k, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i], n = cm.n;
  double* o = out + (size_t)i * kPusherStateDim;
  for (int j = 0; j < PU::kNQ; ++j) {
    o[j] = dev.qpos[(size_t)j * n + e];
    o[PU::kNQ + j] = dev.qvel[(size_t)j * n + e];
    o[2 * PU::kNQ + j] = dev.warm[(size_t)j * n + e];
  }
  double* t = o + 3 * PU::kNQ;
  t[0] = 0;
  t[1] = dev.lag[e];
  t[2] = dev.lag[(size_t)n + e];
  t[3] = cm.done[e];
  t[4] = cm.cur_step[e];
  t[5] = t[6] = 0;
  for (int j = 0; j < 5; ++j) t[7 + j] = dev.lag[(size_t)j * n + e];
}
__global__ void PusherSetState(PusherDev dev, CommonDev cm, const int* ids, int k, const double* in) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i], n = cm.n;
  const double* o = in + (size_t)i * kPusherStateDim;
  for (int j = 0; j < PU::kNQ; ++j) {
    dev.qpos[(size_t)j * n + e] = o[j];
    dev.qvel[(size_t)j * n + e] = o[PU::kNQ + j];
    dev.warm[(size_t)j * n + e] = o[2 * PU::kNQ + j];
  }
  const double* t = o + 3 * PU::kNQ;
  cm.done[e] = t[3] != 0.0;
  cm.cur_step[e] = (int)t[4];
  for (int j = 0; j < 5; ++j) dev.lag[(size_t)j * n + e] = t[7 + j];
}

std::vector<KeySpec> PusherKeys(const Config& cfg) {  // pusher.h:47-60
  return {{"obs", EPA_F64, StackedObsShape(cfg, 23)},
          {"info:reward_dist", EPA_F64, {}},
          {"info:reward_ctrl", EPA_F64, {}},
          {"info:reward_near", EPA_F64, {}}};
}

class PusherPool : public Pool {
 public:
  bool ConcurrentSafe() const override { return true; }  // per-env state + the launch's own block only
  explicit PusherPool(const Config& cfg)
      : Pool(cfg, PusherKeys(cfg), KeySpec{"action", EPA_F64, {PU::kNL}}, /*needs_rng=*/true) {
    EnableObsStack();
    v5_ = cfg.Get("xml_v5", 0) != 0;
    // defaults: pusher.h:33-46
    task_.frame_skip = (int)cfg.Get("frame_skip", 5);
    task_.reward_after_step = cfg.Get("reward_after_step", 0) != 0;
    task_.weighted_reward_info = cfg.Get("weighted_reward_info", 0) != 0;
    task_.ctrl_cost_weight = cfg.Get("ctrl_cost_weight", 0.1);
    task_.dist_cost_weight = cfg.Get("dist_cost_weight", 1.0);
    task_.near_cost_weight = cfg.Get("near_cost_weight", 0.5);
    task_.reset_qvel_scale = cfg.Get("reset_qvel_scale", 0.005);
    task_.cyl_x_min = cfg.Get("cylinder_x_min", -0.3);
    task_.cyl_x_max = cfg.Get("cylinder_x_max", 0.0);
    task_.cyl_y_min = cfg.Get("cylinder_y_min", -0.2);
    task_.cyl_y_max = cfg.Get("cylinder_y_max", 0.2);
    task_.cyl_dist_min = cfg.Get("cylinder_dist_min", 0.17);
    spread_ = cfg.Get("planar_spread", 1) != 0;
    {
      hipDeviceProp_t prop;
      EPA_HIP(hipGetDeviceProperties(&prop, cfg.device));
      wave_slots_ = prop.multiProcessorCount * 4 < 1 ? 1 : prop.multiProcessorCount * 4;
    }
    size_t n = cfg.num_envs;
    for (double** p : {&dev_.qpos, &dev_.qvel, &dev_.warm}) {
      EPA_HIP(hipMalloc(p, sizeof(double) * PU::kNQ * n));
      EPA_HIP(hipMemsetAsync(*p, 0, sizeof(double) * PU::kNQ * n, stream_));
    }
    EPA_HIP(hipMalloc(&dev_.lag, sizeof(double) * 5 * n));
    EPA_HIP(hipMemsetAsync(dev_.lag, 0, sizeof(double) * 5 * n, stream_));
    InitCommon();
  }
  ~PusherPool() override {
    (void)hipFree(dev_.qpos);
    (void)hipFree(dev_.qvel);
    (void)hipFree(dev_.warm);
    (void)hipFree(dev_.lag);
  }
  int StateDim() const override { return kPusherStateDim; }
  void GetState(const int* d_ids, int k, double* d_out) override {
    hipLaunchKernelGGL(PusherGetState, dim3((k + 255) / 256), dim3(256), 0, stream_, dev_,
                       common_, d_ids, k, d_out);
  }
  void SetState(const int* d_ids, int k, const double* d_in) override {
    hipLaunchKernelGGL(PusherSetState, dim3((k + 255) / 256), dim3(256), 0, stream_, dev_,
                       common_, d_ids, k, d_in);
  }

 protected:
  void Launch(const int* d_ids, int k, const void* d_action, bool force_reset,
              const OutPtrs& out) override {
    StepArgs a{d_ids, k, force_reset ? 1 : 0, cfg_.max_episode_steps, cfg_.env_id_offset};
    // one wave per SIMD, as the planar kernel: a batch of 32 .. 64 envs per SIMD runs with 32 / 48 envs
    // per wave on all SIMDs (see CheetahPool::Launch, "planar_spread"; measured N = 32768: +5 %, and
    // 16 per wave at N = 16384: -4 %, hence the floor of 32 here)
    int lanes = kPusherBlock;
    if (spread_ && k >= 32 * wave_slots_) {
      lanes = ((k + wave_slots_ - 1) / wave_slots_ + 15) / 16 * 16;
      lanes = lanes > kPusherBlock ? kPusherBlock : lanes;
    }
    task_.lanes = lanes;
    int blocks = (k + lanes - 1) / lanes;
    const mj::SolverCfg<double> sc{50, 1e-13};
    auto* kernel = v5_ ? PusherStepKernel<true> : PusherStepKernel<false>;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kPusherBlock), 0, stream_, dev_, common_, a,
                       static_cast<const double*>(d_action), out, task_, sc);
  }

 private:
  PusherDev dev_{};
  bool v5_{false};
  PusherTask task_{};
  bool spread_{true};
  int wave_slots_{1024};
};

}  // namespace

bool DescribePusher(const std::string& family, const Config& cfg, std::vector<KeySpec>* state,
                    KeySpec* action) {
  if (family != "Pusher") return false;
  *state = PusherKeys(cfg);
  *action = KeySpec{"action", EPA_F64, {PU::kNL}};
  return true;
}

Pool* MakePusher(const std::string& family, const Config& cfg) {
  if (family != "Pusher") return nullptr;
  return new PusherPool(cfg);
}

}  // namespace epa
