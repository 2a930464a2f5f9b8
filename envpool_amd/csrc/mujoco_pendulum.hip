// gym InvertedPendulum / InvertedDoublePendulum batched step kernel (one env
// per thread).  Replaces, for the whole batch in one launch:
//   MujocoEnv::{MujocoReset,MujocoStep}        envpool/mujoco/gym/mujoco_env.h:126-148
//   InvertedPendulumEnvBase::{MujocoResetModel,Reset,Step,WriteState}
//                                              envpool/mujoco/gym/inverted_pendulum.h:100-185
//   InvertedDoublePendulumEnvBase::{...}       envpool/mujoco/gym/inverted_double_pendulum.h:108-186
//   ReacherEnvBase::{...}                      envpool/mujoco/gym/reacher.h:112-221
//   SwimmerEnvBase::{...}                      envpool/mujoco/gym/swimmer.h:110-186
// with the `frame_skip x mj_step` (RK4) physics of mj_pendulum.hip.h.  No contacts
// (every geom has contype 0), joint limits only; state is 3 x nv doubles per env,
// so unlike the legged robots this kernel is HBM-streaming: 2 (3) dofs, ~2e3
// flops and ~250 algorithmic bytes per env-step.
#include "device_common.hip.h"
#include "engine.h"
#include "mj_pendulum.hip.h"
#include "mj_pendulum_model.h"
#include "build/mj_pendulum_consts.inc"  // generated: k{InvertedPendulum,InvertedDoublePendulum,Reacher,Swimmer}ModelConst

namespace epa {
namespace {

namespace P = mj::pend;

struct PendDev {
  double* qpos;  // [nv][N]
  double* qvel;  // [nv][N]
  double* warm;  // [nv][N]
  double* nsaved;          // normal_distribution::_M_saved (double pendulum qvel noise)
  unsigned char* navail;   // normal_distribution::_M_saved_available
};

struct PendTask {
  int frame_skip;
  int reward_if_not_terminated;
  int constraint_obs_dim;  // double pendulum: 3 (v2/v4) or 1 (v5)
  double healthy_reward, healthy_z_min, healthy_z_max, reset_noise_scale;
  double observation_min, observation_max;
};

constexpr int kPendBlock = 256;

template <int NL>
__global__ __launch_bounds__(kPendBlock) void PendStepKernel(
    PendDev dev, CommonDev cm, StepArgs a, const double* __restrict__ action, OutPtrs out,
    PendTask task, mj::SolverCfg<double> scfg) {
  // the model is a compile-time constant (gen_mj_consts.cpp): as a kernel argument it lived in --
  // and spilled out of -- SGPRs (124 spills for NL = 2; see the Makefile note and tests/test_build_guard.py)
  constexpr P::PendModel<double, NL, P::kBaseCart> m = [] {
    if constexpr (NL == 1) return kInvertedPendulumModelConst;
    else return kInvertedDoublePendulumModelConst;
  }();
  constexpr int NV = NL + 1;
  const int n = cm.n;
  const int row = blockIdx.x * kPendBlock + threadIdx.x;
  if (row >= a.k) return;
  const int e = a.ids ? a.ids[row] - a.id_offset : row;
  bool done = cm.done[e] != 0;
  int cur = cm.cur_step[e];
  const bool reset = a.force_reset || done;  // async_envpool.h:127
  double q[NV], v[NV], w[NV];
  P::PendAux<double, NL> aux{};
  float reward = 0.0f;
  if (reset) {
    // MujocoReset: mj_resetData, MujocoResetModel, mj_forward (mujoco_env.h:126-131)
    cur = 0;
    done = false;
    Mt19937 g(cm, e);
    double saved = dev.nsaved[e];
    int avail = dev.navail[e];
    for (int i = 0; i < NV; ++i) {
      q[i] = 0.0 + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
    }
    for (int i = 0; i < NV; ++i) {
      if constexpr (NL == 1) {  // inverted_pendulum.h:100-107: uniform
        v[i] = 0.0 + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
      } else {  // inverted_double_pendulum.h:117-124: normal
        v[i] = 0.0 + g.Normal(0.0, task.reset_noise_scale, &saved, &avail);
      }
      w[i] = 0.0;
    }
    g.Commit();
    dev.nsaved[e] = saved;
    dev.navail[e] = (unsigned char)avail;
    double qacc[NV];
    const double zero[NV] = {0};
    P::PendForward(m, scfg, q, v, zero, w, qacc, aux);  // ctrl = 0 after mj_resetData
  } else {
    ++cur;
    mj::static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      q[i] = dev.qpos[(size_t)i * n + e];
      v[i] = dev.qvel[(size_t)i * n + e];
      w[i] = dev.warm[(size_t)i * n + e];
    });
    double act[NV] = {0};  // the only motor drives the slider (dof 0)
    act[0] = action[row];
    for (int s = 0; s < task.frame_skip; ++s) {  // mujoco_env.h:142-144
      P::PendStepRK4(m, scfg, q, v, w, act, aux);
    }
    bool terminated;
    if constexpr (NL == 1) {  // inverted_pendulum.h:137-148,151-167
      bool healthy = !(q[1] < task.healthy_z_min || q[1] > task.healthy_z_max);
      mj::static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        healthy = healthy && isfinite(q[i]) && isfinite(v[i]);
      });
      terminated = !healthy;
      reward = task.reward_if_not_terminated ? static_cast<float>(!terminated) : 1.0f;
    } else {  // inverted_double_pendulum.h:126-151; site_xpos of the last RK4 stage
      const double x = aux.tip_x, y = aux.tip_z;
      const double dist_penalty = 0.01 * x * x + (y - 2) * (y - 2);
      const double vel_penalty = 1e-3 * v[1] * v[1] + 5e-3 * v[2] * v[2];
      terminated = !(y > task.healthy_z_max);
      const double alive_bonus = task.reward_if_not_terminated
                                     ? task.healthy_reward * static_cast<int>(!terminated)
                                     : task.healthy_reward;
      reward = static_cast<float>(alive_bonus - dist_penalty - vel_penalty);
    }
    done = terminated || cur >= a.max_episode_steps;
  }
  mj::static_for<0, NV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    dev.qpos[(size_t)i * n + e] = q[i];
    dev.qvel[(size_t)i * n + e] = v[i];
    dev.warm[(size_t)i * n + e] = w[i];
  });
  cm.done[e] = done ? 1 : 0;
  cm.cur_step[e] = cur;
  if constexpr (NL == 1) {  // WriteState, inverted_pendulum.h:169-180
    double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * 4;
    obs[0] = q[0];
    obs[1] = q[1];
    obs[2] = v[0];
    obs[3] = v[1];
  } else {  // inverted_double_pendulum.h:153-182
    const int nobs = 8 + task.constraint_obs_dim;
    double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * nobs;
    auto clip = [&](double x) {
      x = task.observation_max < x ? task.observation_max : x;  // std::min(max_, x)
      x = task.observation_min > x ? task.observation_min : x;  // std::max(min_, x)
      return x;
    };
    obs[0] = q[0];
    obs[1] = sin(q[1]);
    obs[2] = sin(q[2]);
    obs[3] = cos(q[1]);
    obs[4] = cos(q[2]);
    mj::static_for<0, NV>([&](auto ic) { obs[5 + decltype(ic)::value] = clip(v[decltype(ic)::value]); });
    mj::static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (i < task.constraint_obs_dim) obs[8 + i] = clip(aux.qfrc_constraint[i]);
    });
  }
  WriteCommon(out, row, e + a.id_offset, cur, done, reward, a.max_episode_steps);
}

// flat state, same layout as oracle/mjcpu: qpos[nv] qvel[nv] warm[nv] time xlag
// ylag done cur_step normal_saved normal_avail
template <int NV>
__global__ void PendGetState(PendDev dev, CommonDev cm, const int* ids, int k, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i], n = cm.n;
  double* o = out + (size_t)i * (3 * NV + 7);
  for (int j = 0; j < NV; ++j) {
    o[j] = dev.qpos[(size_t)j * n + e];
    o[NV + j] = dev.qvel[(size_t)j * n + e];
    o[2 * NV + j] = dev.warm[(size_t)j * n + e];
  }
  double* t = o + 3 * NV;
  t[0] = t[1] = t[2] = 0;
  t[3] = cm.done[e];
  t[4] = cm.cur_step[e];
  t[5] = dev.nsaved[e];
  t[6] = dev.navail[e];
}
template <int NV>
__global__ void PendSetState(PendDev dev, CommonDev cm, const int* ids, int k, const double* in) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i], n = cm.n;
  const double* o = in + (size_t)i * (3 * NV + 7);
  for (int j = 0; j < NV; ++j) {
    dev.qpos[(size_t)j * n + e] = o[j];
    dev.qvel[(size_t)j * n + e] = o[NV + j];
    dev.warm[(size_t)j * n + e] = o[2 * NV + j];
  }
  const double* t = o + 3 * NV;
  cm.done[e] = t[3] != 0.0;
  cm.cur_step[e] = (int)t[4];
  dev.nsaved[e] = t[5];
  dev.navail[e] = t[6] != 0.0;
}

// ---- Reacher ------------------------------------------------------------------
// qpos = [joint0, joint1, target_x, target_y]; the target never moves (see
// BuildReacher), so the dynamics only see the two arm hinges.  The reference reads
// body positions from mjData (reacher.h:175-181): after an RK4 mj_step those are
// the fingertip position of the LAST stage's evaluation, kept in `lag`.
struct ReacherDev {
  double* qpos;  // [4][N]
  double* qvel;  // [4][N]
  double* warm;  // [4][N]
  double* lag;   // [2][N] fingertip xpos (x, y) as left by the last forward evaluation
};

struct ReacherTask {
  int frame_skip, reward_after_step, obs_include_z;
  double ctrl_cost_weight, dist_cost_weight;
  double reset_qpos_scale, reset_qvel_scale, reset_goal_scale;
};

__global__ __launch_bounds__(kPendBlock) void ReacherStepKernel(
    ReacherDev dev, CommonDev cm, StepArgs a, const double* __restrict__ action, OutPtrs out,
    ReacherTask task, mj::SolverCfg<double> scfg) {
  constexpr P::PendModel<double, 2, P::kBaseFixed> m = kReacherModelConst;
  const int n = cm.n;
  const int row = blockIdx.x * kPendBlock + threadIdx.x;
  if (row >= a.k) return;
  const int e = a.ids ? a.ids[row] - a.id_offset : row;
  bool done = cm.done[e] != 0;
  int cur = cm.cur_step[e];
  const bool reset = a.force_reset || done;  // async_envpool.h:127
  double q[2], v[2], w[2], tx, ty, fx, fy;  // arm state, target qpos, lagged fingertip
  P::PendAux<double, 2> aux{};
  float reward = 0.0f;
  double dist_cost = 0.0, ctrl_cost = 0.0;
  if (reset) {  // MujocoReset + MujocoResetModel, reacher.h:112-132
    cur = 0;
    done = false;
    Mt19937 g(cm, e);
    for (int i = 0; i < 2; ++i) q[i] = 0.0 + g.UniformReal(-task.reset_qpos_scale, task.reset_qpos_scale);
    for (;;) {  // goal: rejection sampling inside the disc of radius reset_goal_scale
      const double x = g.UniformReal(-task.reset_goal_scale, task.reset_goal_scale);
      const double y = g.UniformReal(-task.reset_goal_scale, task.reset_goal_scale);
      if (sqrt(x * x + y * y) < task.reset_goal_scale) {
        tx = x;
        ty = y;
        break;
      }
    }
    for (int i = 0; i < 2; ++i) v[i] = 0.0 + g.UniformReal(-task.reset_qvel_scale, task.reset_qvel_scale);
    g.Commit();
    w[0] = w[1] = 0.0;
    double qacc[2];
    const double zero[2] = {0.0, 0.0};
    P::PendForward(m, scfg, q, v, zero, w, qacc, aux);  // mj_forward: xpos, warm start
    fx = aux.tip_x;
    fy = -aux.tip_z;  // the kernel's z axis is -y of the model (mj_pendulum.hip.h)
    dev.qpos[(size_t)2 * n + e] = tx;
    dev.qpos[(size_t)3 * n + e] = ty;
    dev.qvel[(size_t)2 * n + e] = 0.0;
    dev.qvel[(size_t)3 * n + e] = 0.0;
    dev.warm[(size_t)2 * n + e] = 0.0;
    dev.warm[(size_t)3 * n + e] = 0.0;
  } else {
    ++cur;
    for (int i = 0; i < 2; ++i) {
      q[i] = dev.qpos[(size_t)i * n + e];
      v[i] = dev.qvel[(size_t)i * n + e];
      w[i] = dev.warm[(size_t)i * n + e];
    }
    tx = dev.qpos[(size_t)2 * n + e];
    ty = dev.qpos[(size_t)3 * n + e];
    fx = dev.lag[e];
    fy = dev.lag[(size_t)n + e];
    // target xpos = body pos (.1, -.1) + (qpos - ref (.1, -.1)) = (qpos_x, qpos_y);
    // fingertip and target share z = .01, so dist[2] is exactly 0
    double dx = fx - tx, dy = fy - ty;  // GetDist before the step (reacher.h:155-158)
    const double act[2] = {action[(size_t)row * 2], action[(size_t)row * 2 + 1]};
    for (int s = 0; s < task.frame_skip; ++s) P::PendStepRK4(m, scfg, q, v, w, act, aux);
    fx = aux.tip_x;
    fy = -aux.tip_z;
    if (task.reward_after_step) {
      dx = fx - tx;
      dy = fy - ty;
    }
    dist_cost = task.dist_cost_weight * sqrt(dx * dx + dy * dy + 0.0 * 0.0);
    for (int i = 0; i < 2; ++i) ctrl_cost += task.ctrl_cost_weight * act[i] * act[i];
    reward = static_cast<float>(-dist_cost - ctrl_cost);
    done = cur >= a.max_episode_steps;
  }
  for (int i = 0; i < 2; ++i) {
    dev.qpos[(size_t)i * n + e] = q[i];
    dev.qvel[(size_t)i * n + e] = v[i];
    dev.warm[(size_t)i * n + e] = w[i];
  }
  dev.lag[e] = fx;
  dev.lag[(size_t)n + e] = fy;
  cm.done[e] = done ? 1 : 0;
  cm.cur_step[e] = cur;
  // WriteState, reacher.h:184-221
  const int nobs = task.obs_include_z ? 11 : 10;
  double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * nobs;
  obs[0] = cos(q[0]);
  obs[1] = cos(q[1]);
  obs[2] = sin(q[0]);
  obs[3] = sin(q[1]);
  obs[4] = tx;
  obs[5] = ty;
  obs[6] = v[0];
  obs[7] = v[1];
  obs[8] = fx - tx;
  obs[9] = fy - ty;
  if (task.obs_include_z) obs[10] = 0.0;
  ((double*)out.p[kKeyEnv0 + 1])[row] = -dist_cost;
  ((double*)out.p[kKeyEnv0 + 2])[row] = -ctrl_cost;
  WriteCommon(out, row, e + a.id_offset, cur, done, reward, a.max_episode_steps);
}

// flat state like oracle/mjcpu: qpos[4] qvel[4] warm[4] time xlag ylag done cur_step 0 0
__global__ void ReacherGetState(ReacherDev dev, CommonDev cm, const int* ids, int k, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i], n = cm.n;
  double* o = out + (size_t)i * 19;
  for (int j = 0; j < 4; ++j) {
    o[j] = dev.qpos[(size_t)j * n + e];
    o[4 + j] = dev.qvel[(size_t)j * n + e];
    o[8 + j] = dev.warm[(size_t)j * n + e];
  }
  o[12] = 0;
  o[13] = dev.lag[e];
  o[14] = dev.lag[(size_t)n + e];
  o[15] = cm.done[e];
  o[16] = cm.cur_step[e];
  o[17] = o[18] = 0;
}
__global__ void ReacherSetState(ReacherDev dev, CommonDev cm, const int* ids, int k,
                                const double* in) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i], n = cm.n;
  const double* o = in + (size_t)i * 19;
  for (int j = 0; j < 4; ++j) {
    dev.qpos[(size_t)j * n + e] = o[j];
    dev.qvel[(size_t)j * n + e] = o[4 + j];
    dev.warm[(size_t)j * n + e] = o[8 + j];
  }
  dev.lag[e] = o[13];
  dev.lag[(size_t)n + e] = o[14];
  cm.done[e] = o[15] != 0.0;
  cm.cur_step[e] = (int)o[16];
}

std::vector<KeySpec> ReacherKeys(const Config& cfg) {  // reacher.h:44-60
  int nobs = cfg.Get("obs_include_z_distance", 1) != 0 ? 11 : 10;
  return {{"obs", EPA_F64, StackedObsShape(cfg, nobs)},
          {"info:reward_dist", EPA_F64, {}},
          {"info:reward_ctrl", EPA_F64, {}}};
}

class ReacherPool : public Pool {
 public:
  bool ConcurrentSafe() const override { return true; }  // per-env state + the launch's own block only
  explicit ReacherPool(const Config& cfg)
      : Pool(cfg, ReacherKeys(cfg), KeySpec{"action", EPA_F64, {2}}, /*needs_rng=*/true) {
    EnableObsStack();
    model_ = P::BuildReacher();
    // defaults: reacher.h:32-43
    task_.frame_skip = (int)cfg.Get("frame_skip", 2);
    task_.reward_after_step = cfg.Get("reward_after_step", 0) != 0;
    task_.obs_include_z = cfg.Get("obs_include_z_distance", 1) != 0;
    task_.ctrl_cost_weight = cfg.Get("ctrl_cost_weight", 1.0);
    task_.dist_cost_weight = cfg.Get("dist_cost_weight", 1.0);
    task_.reset_qpos_scale = cfg.Get("reset_qpos_scale", 0.1);
    task_.reset_qvel_scale = cfg.Get("reset_qvel_scale", 0.005);
    task_.reset_goal_scale = cfg.Get("reset_goal_scale", 0.2);
    size_t n = cfg.num_envs;
    for (double** p : {&dev_.qpos, &dev_.qvel, &dev_.warm}) {
      EPA_HIP(hipMalloc(p, sizeof(double) * 4 * n));
      EPA_HIP(hipMemsetAsync(*p, 0, sizeof(double) * 4 * n, stream_));
    }
    EPA_HIP(hipMalloc(&dev_.lag, sizeof(double) * 2 * n));
    EPA_HIP(hipMemsetAsync(dev_.lag, 0, sizeof(double) * 2 * n, stream_));
    InitCommon();
  }
  ~ReacherPool() override {
    (void)hipFree(dev_.qpos);
    (void)hipFree(dev_.qvel);
    (void)hipFree(dev_.warm);
    (void)hipFree(dev_.lag);
  }
  int StateDim() const override { return 19; }
  void GetState(const int* d_ids, int k, double* d_out) override {
    hipLaunchKernelGGL(ReacherGetState, dim3((k + 255) / 256), dim3(256), 0, stream_, dev_,
                       common_, d_ids, k, d_out);
  }
  void SetState(const int* d_ids, int k, const double* d_in) override {
    hipLaunchKernelGGL(ReacherSetState, dim3((k + 255) / 256), dim3(256), 0, stream_, dev_,
                       common_, d_ids, k, d_in);
  }

 protected:
  void Launch(const int* d_ids, int k, const void* d_action, bool force_reset,
              const OutPtrs& out) override {
    StepArgs a{d_ids, k, force_reset ? 1 : 0, cfg_.max_episode_steps, cfg_.env_id_offset};
    int blocks = (k + kPendBlock - 1) / kPendBlock;
    const mj::SolverCfg<double> sc{50, 1e-13};
    hipLaunchKernelGGL(ReacherStepKernel, dim3(blocks), dim3(kPendBlock), 0, stream_, dev_,
                       common_, a, static_cast<const double*>(d_action), out, task_, sc);
  }

 private:
  ReacherDev dev_{};
  P::PendModel<double, 2, P::kBaseFixed> model_{};
  ReacherTask task_{};
};

// ---- Swimmer -------------------------------------------------------------------
// qpos = [slider1 x, slider2 y, free_body_rot, motor1_rot, motor2_rot]; the kernel's
// plane is (x, z = -y), so the y slide changes sign on load / store.
struct SwimmerTask {
  int frame_skip, obs_skip;
  double ctrl_cost_weight, forward_reward_weight, reset_noise_scale, dt;
};

__global__ __launch_bounds__(kPendBlock) void SwimmerStepKernel(
    PendDev dev, CommonDev cm, StepArgs a, const double* __restrict__ action, OutPtrs out,
    SwimmerTask task, mj::SolverCfg<double> scfg) {
  constexpr P::PendModel<double, 3, P::kBaseFree> m = kSwimmerModelConst;
  constexpr int NV = 5;
  const int n = cm.n;
  const int row = blockIdx.x * kPendBlock + threadIdx.x;
  if (row >= a.k) return;
  const int e = a.ids ? a.ids[row] - a.id_offset : row;
  bool done = cm.done[e] != 0;
  int cur = cm.cur_step[e];
  const bool reset = a.force_reset || done;  // async_envpool.h:127
  double q[NV], v[NV], w[NV];  // kernel coordinates (y mirrored)
  P::PendAux<double, 3> aux{};
  float reward = 0.0f;
  double info[7] = {0, 0, 0, 0, 0, 0, 0};
  if (reset) {  // MujocoReset + MujocoResetModel, swimmer.h:110-121
    cur = 0;
    done = false;
    Mt19937 g(cm, e);
    for (int i = 0; i < NV; ++i) q[i] = 0.0 + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
    for (int i = 0; i < NV; ++i) v[i] = 0.0 + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
    g.Commit();
    q[1] = -q[1];
    v[1] = -v[1];
    for (int i = 0; i < NV; ++i) w[i] = 0.0;
    double qacc[NV];
    const double zero[NV] = {0, 0, 0, 0, 0};
    P::PendForward(m, scfg, q, v, zero, w, qacc, aux);  // mj_forward (warm start)
    info[4] = sqrt(0.0);  // WriteState(0, 0, 0, 0, 0, 0, true): swimmer.h:127
    info[1] = -0.0;       // `-ctrl_cost` of +0.0 (swimmer.h:172)
  } else {
    ++cur;
    mj::static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr double sg = i == 1 ? -1.0 : 1.0;
      q[i] = sg * dev.qpos[(size_t)i * n + e];
      v[i] = sg * dev.qvel[(size_t)i * n + e];
      w[i] = sg * dev.warm[(size_t)i * n + e];
    });
    const double x_before = q[0], y_before = -q[1];
    const double a0 = action[(size_t)row * 2], a1 = action[(size_t)row * 2 + 1];
    const double act[NV] = {0, 0, 0, a0, a1};  // motors on motor1_rot / motor2_rot
    for (int s = 0; s < task.frame_skip; ++s) P::PendStepRK4(m, scfg, q, v, w, act, aux);
    const double x_after = q[0], y_after = -q[1];
    const double ctrl_cost = task.ctrl_cost_weight * a0 * a0 + task.ctrl_cost_weight * a1 * a1;
    const double xv = (x_after - x_before) / task.dt, yv = (y_after - y_before) / task.dt;
    reward = static_cast<float>(xv * task.forward_reward_weight - ctrl_cost);
    done = cur >= a.max_episode_steps;
    info[0] = xv * task.forward_reward_weight;
    info[1] = -ctrl_cost;
    info[2] = x_after;
    info[3] = y_after;
    info[4] = sqrt(x_after * x_after + y_after * y_after);
    info[5] = xv;
    info[6] = yv;
  }
  double qm[NV], vm[NV];  // model coordinates
  mj::static_for<0, NV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr double sg = i == 1 ? -1.0 : 1.0;
    qm[i] = sg * q[i];
    vm[i] = sg * v[i];
    dev.qpos[(size_t)i * n + e] = qm[i];
    dev.qvel[(size_t)i * n + e] = vm[i];
    dev.warm[(size_t)i * n + e] = sg * w[i];
  });
  cm.done[e] = done ? 1 : 0;
  cm.cur_step[e] = cur;
  // WriteState, swimmer.h:155-186
  double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * (2 * NV - task.obs_skip);
  for (int i = task.obs_skip; i < NV; ++i) *(obs++) = qm[i];
  for (int i = 0; i < NV; ++i) *(obs++) = vm[i];
  for (int i = 0; i < 7; ++i) ((double*)out.p[kKeyEnv0 + 1 + i])[row] = info[i];
  WriteCommon(out, row, e + a.id_offset, cur, done, reward, a.max_episode_steps);
}

std::vector<KeySpec> SwimmerKeys(const Config& cfg) {  // swimmer.h:44-61
  int no_pos = cfg.Get("exclude_current_positions_from_observation", 1) != 0;
  std::vector<KeySpec> k = {{"obs", EPA_F64, StackedObsShape(cfg, no_pos ? 8 : 10)}};
  for (const char* name : {"info:reward_fwd", "info:reward_ctrl", "info:x_position",
                           "info:y_position", "info:distance_from_origin", "info:x_velocity",
                           "info:y_velocity"}) {
    k.push_back({name, EPA_F64, {}});
  }
  return k;
}

int PendObsDim(const Config& cfg, int nl) {
  if (nl == 1) return 4;  // inverted_pendulum.h:43-55
  int c = (int)cfg.Get("constraint_obs_dim", 3);
  if (c < 0 || c > 3) throw std::invalid_argument("constraint_obs_dim must be in [0, 3]");
  return 1 + 2 + 2 + 3 + c;  // inverted_double_pendulum.h:46-60
}

template <int NL>
class PendPool : public Pool {
 public:
  bool ConcurrentSafe() const override { return true; }  // per-env state + the launch's own block only
  static constexpr int NV = NL + 1;
  explicit PendPool(const Config& cfg)
      : Pool(cfg, {{"obs", EPA_F64, StackedObsShape(cfg, PendObsDim(cfg, NL))}},
             KeySpec{"action", EPA_F64, {1}}, /*needs_rng=*/true) {
    EnableObsStack();
    if constexpr (NL == 1) {
      model1_ = P::BuildInvertedPendulum();
    } else {
      model2_ = P::BuildInvertedDoublePendulum();
    }
    // defaults: inverted_pendulum.h:32-41 / inverted_double_pendulum.h:32-44
    task_.frame_skip = (int)cfg.Get("frame_skip", NL == 1 ? 2 : 5);
    task_.reward_if_not_terminated = cfg.Get("reward_if_not_terminated", 0) != 0;
    task_.constraint_obs_dim = NL == 1 ? 0 : (int)cfg.Get("constraint_obs_dim", 3);
    task_.healthy_reward = cfg.Get("healthy_reward", NL == 1 ? 1.0 : 10.0);
    task_.healthy_z_min = cfg.Get("healthy_z_min", -0.2);
    task_.healthy_z_max = cfg.Get("healthy_z_max", NL == 1 ? 0.2 : 1.0);
    task_.reset_noise_scale = cfg.Get("reset_noise_scale", NL == 1 ? 0.01 : 0.1);
    task_.observation_min = cfg.Get("observation_min", -10.0);
    task_.observation_max = cfg.Get("observation_max", 10.0);
    size_t n = cfg.num_envs;
    EPA_HIP(hipMalloc(&dev_.qpos, sizeof(double) * NV * n));
    EPA_HIP(hipMalloc(&dev_.qvel, sizeof(double) * NV * n));
    EPA_HIP(hipMalloc(&dev_.warm, sizeof(double) * NV * n));
    EPA_HIP(hipMalloc(&dev_.nsaved, sizeof(double) * n));
    EPA_HIP(hipMalloc(&dev_.navail, n));
    EPA_HIP(hipMemsetAsync(dev_.qpos, 0, sizeof(double) * NV * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.qvel, 0, sizeof(double) * NV * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.warm, 0, sizeof(double) * NV * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.nsaved, 0, sizeof(double) * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.navail, 0, n, stream_));
    mt_tile_default_ = 16;  // the inverted pendulums fall over at their own times
    InitCommon();
  }
  ~PendPool() override {
    (void)hipFree(dev_.qpos);
    (void)hipFree(dev_.qvel);
    (void)hipFree(dev_.warm);
    (void)hipFree(dev_.nsaved);
    (void)hipFree(dev_.navail);
  }
  int StateDim() const override { return 3 * NV + 7; }
  void GetState(const int* d_ids, int k, double* d_out) override {
    hipLaunchKernelGGL(PendGetState<NV>, dim3((k + 255) / 256), dim3(256), 0, stream_, dev_,
                       common_, d_ids, k, d_out);
  }
  void SetState(const int* d_ids, int k, const double* d_in) override {
    hipLaunchKernelGGL(PendSetState<NV>, dim3((k + 255) / 256), dim3(256), 0, stream_, dev_,
                       common_, d_ids, k, d_in);
  }

 protected:
  void Launch(const int* d_ids, int k, const void* d_action, bool force_reset,
              const OutPtrs& out) override {
    StepArgs a{d_ids, k, force_reset ? 1 : 0, cfg_.max_episode_steps, cfg_.env_id_offset};
    int blocks = (k + kPendBlock - 1) / kPendBlock;
    const mj::SolverCfg<double> sc{50, 1e-13};
    if constexpr (NL == 1) {
      hipLaunchKernelGGL(PendStepKernel<1>, dim3(blocks), dim3(kPendBlock), 0, stream_, dev_,
                         common_, a, static_cast<const double*>(d_action), out, task_, sc);
    } else {
      hipLaunchKernelGGL(PendStepKernel<2>, dim3(blocks), dim3(kPendBlock), 0, stream_, dev_,
                         common_, a, static_cast<const double*>(d_action), out, task_, sc);
    }
  }

 private:
  PendDev dev_{};
  P::PendModel<double, 1, P::kBaseCart> model1_{};
  P::PendModel<double, 2, P::kBaseCart> model2_{};
  PendTask task_{};
};

class SwimmerPool : public Pool {
 public:
  bool ConcurrentSafe() const override { return true; }  // per-env state + the launch's own block only
  static constexpr int NV = 5;
  explicit SwimmerPool(const Config& cfg)
      : Pool(cfg, SwimmerKeys(cfg), KeySpec{"action", EPA_F64, {2}}, /*needs_rng=*/true) {
    EnableObsStack();
    model_ = P::BuildSwimmer();
    // defaults: swimmer.h:32-42
    task_.frame_skip = (int)cfg.Get("frame_skip", 4);
    task_.obs_skip = cfg.Get("exclude_current_positions_from_observation", 1) != 0 ? 2 : 0;
    task_.ctrl_cost_weight = cfg.Get("ctrl_cost_weight", 1e-4);
    task_.forward_reward_weight = cfg.Get("forward_reward_weight", 1.0);
    task_.reset_noise_scale = cfg.Get("reset_noise_scale", 0.1);
    task_.dt = task_.frame_skip * model_.timestep;
    size_t n = cfg.num_envs;
    for (double** p : {&dev_.qpos, &dev_.qvel, &dev_.warm}) {
      EPA_HIP(hipMalloc(p, sizeof(double) * NV * n));
      EPA_HIP(hipMemsetAsync(*p, 0, sizeof(double) * NV * n, stream_));
    }
    EPA_HIP(hipMalloc(&dev_.nsaved, sizeof(double) * n));  // unused (uniform noise only);
    EPA_HIP(hipMalloc(&dev_.navail, n));                   // kept for the shared state layout
    EPA_HIP(hipMemsetAsync(dev_.nsaved, 0, sizeof(double) * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.navail, 0, n, stream_));
    InitCommon();
  }
  ~SwimmerPool() override {
    (void)hipFree(dev_.qpos);
    (void)hipFree(dev_.qvel);
    (void)hipFree(dev_.warm);
    (void)hipFree(dev_.nsaved);
    (void)hipFree(dev_.navail);
  }
  int StateDim() const override { return 3 * NV + 7; }
  void GetState(const int* d_ids, int k, double* d_out) override {
    hipLaunchKernelGGL(PendGetState<NV>, dim3((k + 255) / 256), dim3(256), 0, stream_, dev_,
                       common_, d_ids, k, d_out);
  }
  void SetState(const int* d_ids, int k, const double* d_in) override {
    hipLaunchKernelGGL(PendSetState<NV>, dim3((k + 255) / 256), dim3(256), 0, stream_, dev_,
                       common_, d_ids, k, d_in);
  }

 protected:
  void Launch(const int* d_ids, int k, const void* d_action, bool force_reset,
              const OutPtrs& out) override {
    StepArgs a{d_ids, k, force_reset ? 1 : 0, cfg_.max_episode_steps, cfg_.env_id_offset};
    int blocks = (k + kPendBlock - 1) / kPendBlock;
    const mj::SolverCfg<double> sc{50, 1e-13};
    hipLaunchKernelGGL(SwimmerStepKernel, dim3(blocks), dim3(kPendBlock), 0, stream_, dev_,
                       common_, a, static_cast<const double*>(d_action), out, task_, sc);
  }

 private:
  PendDev dev_{};
  P::PendModel<double, 3, P::kBaseFree> model_{};
  SwimmerTask task_{};
};

}  // namespace

bool DescribePendulum(const std::string& family, const Config& cfg,
                      std::vector<KeySpec>* state, KeySpec* action) {
  if (family == "Reacher" || family == "Swimmer") {
    *state = family == "Reacher" ? ReacherKeys(cfg) : SwimmerKeys(cfg);
    *action = KeySpec{"action", EPA_F64, {2}};
    return true;
  }
  int nl = family == "InvertedPendulum" ? 1 : (family == "InvertedDoublePendulum" ? 2 : 0);
  if (nl == 0) return false;
  *state = {{"obs", EPA_F64, StackedObsShape(cfg, PendObsDim(cfg, nl))}};
  *action = KeySpec{"action", EPA_F64, {1}};
  return true;
}

Pool* MakePendulum(const std::string& family, const Config& cfg) {
  if (family == "InvertedPendulum") return new PendPool<1>(cfg);
  if (family == "InvertedDoublePendulum") return new PendPool<2>(cfg);
  if (family == "Reacher") return new ReacherPool(cfg);
  if (family == "Swimmer") return new SwimmerPool(cfg);
  return nullptr;
}

}  // namespace epa
