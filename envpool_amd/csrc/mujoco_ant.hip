// K3b — gym-MuJoCo Ant batched step kernel (one env per thread, one wave/block).
//
// Replaces, for the whole batch in one launch:
//   MujocoEnv::{MujocoReset,MujocoStep}   envpool/mujoco/gym/mujoco_env.h:126-148
//   AntEnvBase::{MujocoResetModel,Reset,Step,IsHealthy,WriteState}
//                                         envpool/mujoco/gym/ant.h:135-278
// with `frame_skip x mj_step` (RK4: 4 forward passes each) from mj_ant.cuh.
// Ant-v4: use_contact_force=false (no cfrc_ext in obs, contact cost 0).
// Ant-v3: use_contact_force=true but post_constraint=false: MuJoCo 3 fills
//   cfrc_ext only in mj_rnePostConstraint, which the reference then never calls
//   (mujoco_env.h:145-147) => 84 zeros in the obs, contact cost 0.
// Ant-v5: use_contact_force + post_constraint: cfrc_ext of the last forward
//   evaluation (mj_ant.cuh, AntContactWrench), world body excluded.
//
// Persistent state (SoA fp64): qpos[15][N], qvel[14][N], qacc_warmstart[14][N],
// lag[2][N] = data_->xpos[torso].xy of the last forward pass (the reference
// reads the *lagged* torso position, ant.h:169-173 / SURVEY §7 H3), and the
// env's normal_distribution saved value.
#define EPA_SINCOS_MODE 2  // see mj_cheetah.cuh; Ant fp64, N=65536, 200-step bench: mode 2 7.0 M, mode 1 6.7 M, library 6.8 M env-steps/s
#include "device_common.cuh"
#include "engine.h"
#include "mj_ant.cuh"
#include "mj_ant_model.h"
#include "build/mj_ant_consts.inc"  // generated: kAntModelConst (gen_mj_consts.cpp)

namespace epa {
namespace {

namespace A = mj::ant;

struct AntDev {
  double* qpos;  // [15][N]
  double* qvel;  // [14][N]
  double* warm;  // [14][N]
  double* lag;   // [2][N]
  double* nsaved;
  unsigned char* navail;
  double* stack;  // [N][frame_stack * nobs] obs ring (frame_stack > 1 only)
  double* cfrc;   // [N][14][6] cfrc_ext accumulator (use_contact_force + post_constraint only)
};

constexpr int kAntMjBodies = 14;  // world + torso + 4 x (stub, leg, ankle) MuJoCo bodies

struct AntTask {
  int frame_skip, obs_skip, frame_stack;
  int terminate_when_unhealthy, legacy_healthy_reward;
  int use_contact_force, post_constraint, exclude_worldbody;
  double ctrl_cost_weight, forward_reward_weight, healthy_reward;
  double healthy_z_min, healthy_z_max, reset_noise_scale, dt;
  double contact_cost_weight, contact_force_min, contact_force_max;
};

constexpr int kAntBlock = 64;

// kWrench: the Ant-v5 variant that also evaluates cfrc_ext (separate instantiation
// so that Ant-v3/v4 do not pay registers for the extra pass)
template <typename T, bool kWrench>
__global__ __launch_bounds__(kAntBlock) void AntStepKernel(
    AntDev dev, CommonDev cm, StepArgs a, const double* __restrict__ action,
    OutPtrs out, AntTask task, mj::SolverCfg<T> scfg) {
  constexpr A::AntModel<T> m = A::CastAntModel<T>(kAntModelConst);
  // lane-private LDS block [slot][lane]: M and the contact geometry of the
  // current forward pass (mj_ant.cuh, AntPublish)
  __shared__ T lds_buf[A::kAntLdsSlots * kAntBlock];
  const int lane = threadIdx.x;
  const int n = cm.n;
  const int row = blockIdx.x * kAntBlock + threadIdx.x;
  if (row >= a.k) return;
  const int e = a.ids ? a.ids[row] - a.id_offset : row;
  bool done = cm.done[e] != 0;
  int cur = cm.cur_step[e];
  const bool reset = a.force_reset || done;
  double qpos[A::kNQ], qvel[A::kNV];
  float reward = 0.0f;
  double info[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (reset) {
    // MujocoReset (mujoco_env.h:126-131) + MujocoResetModel (ant.h:135-147)
    cur = 0;
    done = false;
    Mt19937 g(cm, e);
    double saved = dev.nsaved[e];
    int avail = dev.navail[e];
    const double init_qpos[A::kNQ] = {0, 0, 0.75, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < A::kNQ; ++i) {
      qpos[i] = init_qpos[i] +
                g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
    }
    for (int i = 0; i < A::kNV; ++i) {
      qvel[i] = 0.0 + g.Normal(0.0, task.reset_noise_scale, &saved, &avail);
    }
    g.Commit();
    dev.nsaved[e] = saved;
    dev.navail[e] = (unsigned char)avail;
    // mj_forward leaves xpos[torso] = qpos[0:3]; the warm start is re-derived by the
    // solver (unique minimiser) so it is simply cleared.  MuJoCo >= 3.1.4 no longer
    // normalises the free-joint quaternion of qpos in place (mj_kinematics works on a
    // normalised copy), so the reset observation carries the raw init_qpos + noise
    // quaternion; the first mj_step's mj_integratePos leaves a unit quaternion.
    for (int i = 0; i < A::kNQ; ++i) dev.qpos[(size_t)i * n + e] = qpos[i];
    for (int i = 0; i < A::kNV; ++i) {
      dev.qvel[(size_t)i * n + e] = qvel[i];
      dev.warm[(size_t)i * n + e] = 0.0;
    }
    dev.lag[e] = qpos[0];
    dev.lag[(size_t)n + e] = qpos[1];
    info[6] = 0.0;  // sqrt(0)
  } else {
    ++cur;
    T q[A::kNQ], v[A::kNV], w[A::kNV], ctrl[A::kNU];
    mj::static_for<0, A::kNQ>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      q[i] = (T)dev.qpos[(size_t)i * n + e];
    });
    mj::static_for<0, A::kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      v[i] = (T)dev.qvel[(size_t)i * n + e];
      w[i] = (T)dev.warm[(size_t)i * n + e];
    });
    const double x_before = dev.lag[e], y_before = dev.lag[(size_t)n + e];
    const double* act = action + (size_t)row * A::kNU;
    double ctrl_cost = 0.0;
    mj::static_for<0, A::kNU>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      double ai = act[i];
      ctrl_cost += task.ctrl_cost_weight * ai * ai;  // ant.h:176-179
      ctrl[i] = (T)(ai < -1.0 ? -1.0 : (ai > 1.0 ? 1.0 : ai));
    });
    T lagx = T(0), lagy = T(0);
    auto lds = [&](int slot) -> T& { return lds_buf[slot * kAntBlock + lane]; };
    // mj_rnePostConstraint after the last mj_step (mujoco_env.h:145-147)
    const bool wrench = kWrench;
    double* cf = wrench ? dev.cfrc + (size_t)e * kAntMjBodies * 6 : nullptr;
    if (wrench) {
      for (int i = 0; i < kAntMjBodies * 6; ++i) cf[i] = 0.0;
    }
    auto sink = [&](int g, A::Vec3<T> tq, A::Vec3<T> f) {
      // MuJoCo body id of geom body g is 1 + g; the world body (0) gets -[tq; f]
      const double w6[6] = {(double)tq.x, (double)tq.y, (double)tq.z,
                            (double)f.x, (double)f.y, (double)f.z};
      for (int j = 0; j < 6; ++j) {
        cf[(1 + g) * 6 + j] += w6[j];
        cf[j] -= w6[j];
      }
    };
    for (int s = 0; s < task.frame_skip; ++s) {
      A::AntStep<kWrench>(m, scfg, q, v, w, ctrl, &lagx, &lagy, lds,
                          wrench && s == task.frame_skip - 1, sink);
    }
    const double x_after = (double)lagx, y_after = (double)lagy;
    bool healthy = true;  // IsHealthy, ant.h:214-229
    mj::static_for<0, A::kNQ>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      qpos[i] = (double)q[i];
      healthy = healthy && isfinite(qpos[i]);
      dev.qpos[(size_t)i * n + e] = qpos[i];
    });
    mj::static_for<0, A::kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      qvel[i] = (double)v[i];
      healthy = healthy && isfinite(qvel[i]);
      dev.qvel[(size_t)i * n + e] = qvel[i];
      dev.warm[(size_t)i * n + e] = (double)w[i];
    });
    if (qpos[2] < task.healthy_z_min || qpos[2] > task.healthy_z_max) healthy = false;
    dev.lag[e] = x_after;
    dev.lag[(size_t)n + e] = y_after;
    const double xv = (x_after - x_before) / task.dt;
    const double yv = (y_after - y_before) / task.dt;
    double contact_cost = 0.0;
    if (wrench) {  // ant.h:183-194 (without post_constraint cfrc_ext stays zero)
      for (int i = task.exclude_worldbody ? 6 : 0; i < kAntMjBodies * 6; ++i) {
        double x = cf[i];
        x = task.contact_force_max < x ? task.contact_force_max : x;  // std::min(max_, x)
        x = task.contact_force_min > x ? task.contact_force_min : x;  // std::max(min_, x)
        contact_cost += task.contact_cost_weight * x * x;
      }
    }
    bool give = healthy;
    if (task.legacy_healthy_reward) give = task.terminate_when_unhealthy || healthy;
    const double healthy_reward = give ? task.healthy_reward : 0.0;
    reward = static_cast<float>(xv * task.forward_reward_weight + healthy_reward -
                                ctrl_cost - contact_cost);
    done = (task.terminate_when_unhealthy ? !healthy : false) ||
           (cur >= a.max_episode_steps);
    info[0] = xv * task.forward_reward_weight;
    info[1] = -ctrl_cost;
    info[2] = -contact_cost;
    info[3] = healthy_reward;
    info[4] = x_after;
    info[5] = y_after;
    info[6] = sqrt(x_after * x_after + y_after * y_after);
    info[7] = xv;
    info[8] = yv;
  }
  cm.done[e] = done ? 1 : 0;
  cm.cur_step[e] = cur;
  // WriteState, ant.h:231-278
  const int cf0 = task.exclude_worldbody ? 6 : 0;
  const int ncf = task.use_contact_force ? kAntMjBodies * 6 - cf0 : 0;
  const int nobs = A::kNQ + A::kNV - task.obs_skip + ncf;
  const int S = task.frame_stack;
  double* obs0 = (double*)out.p[kKeyEnv0] + (size_t)row * nobs * S;
  double* newest = obs0 + (size_t)(S - 1) * nobs;
  {
    double* obs = newest;
    for (int i = task.obs_skip; i < A::kNQ; ++i) *(obs++) = qpos[i];
    for (int i = 0; i < A::kNV; ++i) *(obs++) = qvel[i];
    // ant.h:248-258; a reset (mj_resetData) and Ant-v3 leave cfrc_ext at zero
    const bool have = !reset && kWrench;
    const double* cfr = dev.cfrc + (size_t)e * kAntMjBodies * 6;
    for (int i = 0; i < ncf; ++i) {
      double x = have ? cfr[cf0 + i] : 0.0;
      x = x > task.contact_force_min ? x : task.contact_force_min;  // std::max(x, min_)
      x = x < task.contact_force_max ? x : task.contact_force_max;  // std::min(.., max_)
      *(obs++) = x;
    }
  }
  if (S > 1) {  // FrameStackBuffer::Commit, envpool/mujoco/frame_stack.h:109-135
    double* st = dev.stack + (size_t)e * S * nobs;
    if (reset) {
      for (int f = 0; f < S - 1; ++f) {
        for (int i = 0; i < nobs; ++i) obs0[f * nobs + i] = newest[i];
      }
      for (int j = 0; j < S * nobs; ++j) st[j] = obs0[j];
    } else {
      for (int j = 0; j < (S - 1) * nobs; ++j) {
        double x = st[j + nobs];
        st[j] = x;
        obs0[j] = x;
      }
      for (int i = 0; i < nobs; ++i) st[(S - 1) * nobs + i] = newest[i];
    }
  }
  for (int i = 0; i < 9; ++i) ((double*)out.p[kKeyEnv0 + 1 + i])[row] = info[i];
  WriteCommon(out, row, e + a.id_offset, cur, done, reward, a.max_episode_steps);
}

// flat state like oracle/mjcpu: qpos[15] qvel[14] warm[14] time xlag ylag done
// cur_step normal_saved normal_avail
constexpr int kAntStateDim = A::kNQ + 2 * A::kNV + 7;
__global__ void AntGetState(AntDev dev, CommonDev cm, const int* ids, int k, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i], n = cm.n;
  double* o = out + (size_t)i * kAntStateDim;
  for (int j = 0; j < A::kNQ; ++j) o[j] = dev.qpos[(size_t)j * n + e];
  for (int j = 0; j < A::kNV; ++j) {
    o[A::kNQ + j] = dev.qvel[(size_t)j * n + e];
    o[A::kNQ + A::kNV + j] = dev.warm[(size_t)j * n + e];
  }
  double* t = o + A::kNQ + 2 * A::kNV;
  t[0] = 0;
  t[1] = dev.lag[e];
  t[2] = dev.lag[(size_t)n + e];
  t[3] = cm.done[e];
  t[4] = cm.cur_step[e];
  t[5] = dev.nsaved[e];
  t[6] = dev.navail[e];
}
__global__ void AntSetState(AntDev dev, CommonDev cm, const int* ids, int k, const double* in) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i], n = cm.n;
  const double* o = in + (size_t)i * kAntStateDim;
  for (int j = 0; j < A::kNQ; ++j) dev.qpos[(size_t)j * n + e] = o[j];
  for (int j = 0; j < A::kNV; ++j) {
    dev.qvel[(size_t)j * n + e] = o[A::kNQ + j];
    dev.warm[(size_t)j * n + e] = o[A::kNQ + A::kNV + j];
  }
  const double* t = o + A::kNQ + 2 * A::kNV;
  dev.lag[e] = t[1];
  dev.lag[(size_t)n + e] = t[2];
  cm.done[e] = t[3] != 0.0;
  cm.cur_step[e] = (int)t[4];
  dev.nsaved[e] = t[5];
  dev.navail[e] = t[6] != 0.0;
}

std::vector<KeySpec> AntKeys(const Config& cfg) {
  int no_pos = cfg.Get("exclude_current_positions_from_observation", 1) != 0;
  int fs = (int)cfg.Get("frame_stack", 1);
  // ant.h:51-75 (obs 27/29 + 6 per body with use_contact_force); StackSpec, frame_stack.h:42-71
  int ncf = 0;
  if (cfg.Get("use_contact_force", 0) != 0) {
    ncf = 6 * (kAntMjBodies - (cfg.Get("exclude_worldbody_contact_forces", 0) != 0 ? 1 : 0));
  }
  std::vector<int> oshape = {(no_pos ? 27 : 29) + ncf};
  if (fs > 1) oshape.insert(oshape.begin(), fs);
  std::vector<KeySpec> k = {{"obs", EPA_F64, oshape}};
  for (const char* name :
       {"info:reward_forward", "info:reward_ctrl", "info:reward_contact",
        "info:reward_survive", "info:x_position", "info:y_position",
        "info:distance_from_origin", "info:x_velocity", "info:y_velocity"}) {
    k.push_back({name, EPA_F64, {}});
  }
  return k;
}

class AntPool : public Pool {
 public:
  explicit AntPool(const Config& cfg)
      : Pool(cfg, AntKeys(cfg), KeySpec{"action", EPA_F64, {A::kNU}}, true) {
    task_.frame_stack = (int)cfg.Get("frame_stack", 1);
    if (task_.frame_stack < 1) {
      throw std::invalid_argument("frame_stack must be greater than 0");
    }
    task_.use_contact_force = cfg.Get("use_contact_force", 0) != 0;
    task_.post_constraint = cfg.Get("post_constraint", 0) != 0;
    task_.exclude_worldbody = cfg.Get("exclude_worldbody_contact_forces", 0) != 0;
    task_.contact_cost_weight = cfg.Get("contact_cost_weight", 5e-4);
    task_.contact_force_min = cfg.Get("contact_force_min", -1.0);
    task_.contact_force_max = cfg.Get("contact_force_max", 1.0);
    fp64_ = (int)cfg.Get("precision", 1) == 1;
    model_ = A::BuildAntModel();
    task_.frame_skip = (int)cfg.Get("frame_skip", 5);
    task_.obs_skip = cfg.Get("exclude_current_positions_from_observation", 1) != 0 ? 2 : 0;
    task_.terminate_when_unhealthy = cfg.Get("terminate_when_unhealthy", 1) != 0;
    task_.legacy_healthy_reward = cfg.Get("legacy_healthy_reward", 1) != 0;
    task_.ctrl_cost_weight = cfg.Get("ctrl_cost_weight", 0.5);
    task_.forward_reward_weight = cfg.Get("forward_reward_weight", 1.0);
    task_.healthy_reward = cfg.Get("healthy_reward", 1.0);
    task_.healthy_z_min = cfg.Get("healthy_z_min", 0.2);
    task_.healthy_z_max = cfg.Get("healthy_z_max", 1.0);
    task_.reset_noise_scale = cfg.Get("reset_noise_scale", 0.1);
    task_.dt = task_.frame_skip * model_.timestep;
    size_t n = cfg.num_envs;
    EPA_HIP(hipMalloc(&dev_.qpos, sizeof(double) * A::kNQ * n));
    EPA_HIP(hipMalloc(&dev_.qvel, sizeof(double) * A::kNV * n));
    EPA_HIP(hipMalloc(&dev_.warm, sizeof(double) * A::kNV * n));
    EPA_HIP(hipMalloc(&dev_.lag, sizeof(double) * 2 * n));
    EPA_HIP(hipMalloc(&dev_.nsaved, sizeof(double) * n));
    EPA_HIP(hipMalloc(&dev_.navail, n));
    EPA_HIP(hipMemsetAsync(dev_.qpos, 0, sizeof(double) * A::kNQ * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.qvel, 0, sizeof(double) * A::kNV * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.warm, 0, sizeof(double) * A::kNV * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.lag, 0, sizeof(double) * 2 * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.nsaved, 0, sizeof(double) * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.navail, 0, n, stream_));
    if (task_.use_contact_force && task_.post_constraint) {
      EPA_HIP(hipMalloc(&dev_.cfrc, sizeof(double) * kAntMjBodies * 6 * n));
      EPA_HIP(hipMemsetAsync(dev_.cfrc, 0, sizeof(double) * kAntMjBodies * 6 * n, stream_));
    }
    if (task_.frame_stack > 1) {
      const int ncf = task_.use_contact_force
                          ? 6 * (kAntMjBodies - (task_.exclude_worldbody ? 1 : 0)) : 0;
      size_t sb = sizeof(double) * n * task_.frame_stack *
                  (A::kNQ + A::kNV - task_.obs_skip + ncf);
      EPA_HIP(hipMalloc(&dev_.stack, sb));
      EPA_HIP(hipMemsetAsync(dev_.stack, 0, sb, stream_));
    }
    InitCommon();
  }
  ~AntPool() override {
    (void)hipFree(dev_.qpos);
    (void)hipFree(dev_.qvel);
    (void)hipFree(dev_.warm);
    (void)hipFree(dev_.lag);
    (void)hipFree(dev_.nsaved);
    (void)hipFree(dev_.navail);
    if (dev_.stack) (void)hipFree(dev_.stack);
    if (dev_.cfrc) (void)hipFree(dev_.cfrc);
  }
  int StateDim() const override { return kAntStateDim; }
  void GetState(const int* d_ids, int k, double* d_out) override {
    hipLaunchKernelGGL(AntGetState, dim3((k + 255) / 256), dim3(256), 0, stream_, dev_,
                       common_, d_ids, k, d_out);
  }
  void SetState(const int* d_ids, int k, const double* d_in) override {
    hipLaunchKernelGGL(AntSetState, dim3((k + 255) / 256), dim3(256), 0, stream_, dev_,
                       common_, d_ids, k, d_in);
  }

 protected:
  void Launch(const int* d_ids, int k, const void* d_action, bool force_reset,
              const OutPtrs& out) override {
    StepArgs a{d_ids, k, force_reset ? 1 : 0, cfg_.max_episode_steps, cfg_.env_id_offset};
    int blocks = (k + kAntBlock - 1) / kAntBlock;
    const double* act = static_cast<const double*>(d_action);
    const mj::SolverCfg<double> sd{50, 1e-13};
    const mj::SolverCfg<float> sf{12, 1e-6f};
    const bool wrench = task_.use_contact_force && task_.post_constraint;
#define EPA_LAUNCH_ANT(T, W, SC)                                                          \
  hipLaunchKernelGGL((AntStepKernel<T, W>), dim3(blocks), dim3(kAntBlock), 0, stream_, dev_, \
                     common_, a, act, out, task_, SC)
    if (fp64_) {
      if (wrench) EPA_LAUNCH_ANT(double, true, sd); else EPA_LAUNCH_ANT(double, false, sd);
    } else {
      if (wrench) EPA_LAUNCH_ANT(float, true, sf); else EPA_LAUNCH_ANT(float, false, sf);
    }
#undef EPA_LAUNCH_ANT
  }

 private:
  AntDev dev_{};
  A::AntModel<double> model_;
  AntTask task_{};
  bool fp64_{true};
};

}  // namespace

bool DescribeAnt(const std::string& family, const Config& cfg,
                 std::vector<KeySpec>* state, KeySpec* action) {
  if (family != "Ant") return false;
  *state = AntKeys(cfg);
  *action = KeySpec{"action", EPA_F64, {A::kNU}};
  return true;
}

Pool* MakeAnt(const std::string& family, const Config& cfg) {
  if (family != "Ant") return nullptr;
  return new AntPool(cfg);
}

}  // namespace epa
