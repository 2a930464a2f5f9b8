// K3c — gym-MuJoCo Humanoid / HumanoidStandup batched step kernel (one env per lane, one
// wave per workgroup, per-env workspace in HBM: see mj_tree.hip.h).
//
// Replaces, for the whole batch in one launch:
//   MujocoEnv::{MujocoReset,MujocoStep}   envpool/mujoco/gym/mujoco_env.h:126-148
//   HumanoidEnvBase::{MujocoResetModel,Reset,Step,IsHealthy,GetMassCenter,WriteState}
//                                         envpool/mujoco/gym/humanoid.h:129-268
//   HumanoidStandupEnvBase::{...}         envpool/mujoco/gym/humanoid_standup.h:119-240
// with `frame_skip x mj_step` (RK4: 4 forward passes each, PGS 50 iterations) and, for the
// v5 ids, mj_rnePostConstraint's cfrc_ext.
// v3 / v4 ids register post_constraint=False (gym/registration.py:95-124): MuJoCo 3 fills
// cfrc_ext only in mj_rnePostConstraint, so their 84 cfrc_ext observations and the contact
// cost stay zero, exactly as in the reference.
//
// The observation's cinert / cvel / qfrc_actuator / cfrc_ext and the mass centre belong to
// the LAST forward evaluation (RK4 stage 4 of the last sub-step), like the mjData fields the
// reference reads; the mass centre "before" a step is therefore the lagged one of the
// previous step (persistent slot `lag`).
#define EPA_SINCOS_MODE 0  // see mj_cheetah.hip.h; Humanoid: mode 1 lets the scheduler interleave 17 joints (2.3 k VGPR spills, 1.94 -> 1.77 M env-steps/s)
#include <map>

#include "mujoco_humanoid_common.h"
#include "mj_tree.hip.h"
#include "build/mj_humanoid_consts.inc"  // generated: kHumanoidModelConst, kHumanoidStandupModelConst

namespace epa {

namespace {

namespace T = mj::tree;

#ifdef EPA_HUM_STANDUP_TU  // see the launchers below: one model per translation unit
struct StandupMP {
  static constexpr T::TreeModel kM = kHumanoidStandupModelConst;
};
#else
struct HumanoidMP {
  static constexpr T::TreeModel kM = kHumanoidModelConst;
};
#endif

constexpr int kHumBlock = 64;


// The step kernel of this file is the SUPERSEDED one-env-per-lane form ("hum_layout" = 0; the product kernel is
// mujoco_humanoid4.hip).  It is built only into the alternate library (`make EPA_ALT_KERNELS=1` ->
// libenvpool_amd_alt.so, ~90 s per model) that the cross-check tests load; the default library keeps this file's
// pool class, state accessors and layout constants and refuses hum_layout = 0.
#ifdef EPA_ALT_KERNELS
template <class MP, bool kStandup>
__global__ __launch_bounds__(kHumBlock) void HumanoidStepKernel(
    HumDev dev, CommonDev cm, StepArgs a, const double* __restrict__ action, OutPtrs out,
    HumTask task) {
  using E = T::Tree<MP>;
  constexpr T::TreeModel m = MP::kM;
  constexpr T::Layout L = E::kL;
  __shared__ double lds_ar[E::kArLds * kHumBlock];  // [entry][lane], see Tree::SolvePgsReg
  const int row = blockIdx.x * kHumBlock + threadIdx.x;
  if (row >= a.k) return;
  const int e = a.ids ? a.ids[row] - a.id_offset : row;
  const int n = cm.n;
  const T::Ws w{dev.ws + (size_t)blockIdx.x * L.total * 64, threadIdx.x};
  bool done = cm.done[e] != 0;
  int cur = cm.cur_step[e];
  const bool reset = a.force_reset || done;
  double ctrl_cost = 0.0;
  for (int i = 0; i < L.npersist; ++i) w(i) = dev.state[(size_t)i * n + e];
  if (reset) {
    // MujocoReset (mujoco_env.h:126-131) + MujocoResetModel (humanoid.h:129-141): one
    // uniform distribution for qpos and qvel; mj_resetData clears ctrl and the warm start
    cur = 0;
    done = false;
    Mt19937 g(cm, e);
    for (int i = 0; i < E::NQ; ++i) {
      // (runtime loop: the table lookup is wave-uniform)
      static constexpr T::TreeModel mm = MP::kM;
      w(L.qpos + i) = mm.qpos0[i] + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
    }
    for (int i = 0; i < E::NV; ++i) {
      w(L.qvel + i) = 0.0 + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
      w(L.warm + i) = 0.0;
    }
    for (int i = 0; i < E::NU; ++i) w(L.ctrl + i) = 0.0;
    g.Commit();
  } else {
    ++cur;
    const double* act = action + (size_t)row * E::NU;
    for (int i = 0; i < E::NU; ++i) {
      const double ai = act[i];
      ctrl_cost += task.ctrl_cost_weight * ai * ai;  // humanoid.h:171-174
      w(L.ctrl + i) = ai;                            // clamped to ctrlrange in mj_fwdActuation
    }
  }
  const double x_before = w(L.lag), y_before = w(L.lag + 1);
  // reset lanes: mj_forward once; stepping lanes: frame_skip x (4 RK stages).  Every lane runs
  // the wave's trip count with its own state writes predicated (no divergent control flow).
  const int nfwd = reset ? 1 : 4 * task.frame_skip;
  const int nmax = mj::WaveAny(!reset) ? 4 * task.frame_skip : 1;
  typename E::RowCount rows{0, 0, 0};
  for (int it = 0; it < nmax; ++it) {
    const bool live = it < nfwd;
    rows = E::Forward(w, live, lds_ar + threadIdx.x);
    E::RkAdvance(w.Fresh(), it & 3, live && !reset);
  }
  // mj_rnePostConstraint after the last mj_step (mujoco_env.h:145-147)
  const bool wrench = task.post_constraint != 0;
  if (wrench) E::ContactWrench(w, rows);
  double mx = 0.0, my = 0.0;  // GetMassCenter, humanoid.h:212-223
  mj::static_for<1, E::NB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    mx += m.body_mass[b] * w(L.xipos + 3 * b);
    my += m.body_mass[b] * w(L.xipos + 3 * b + 1);
  });
  mx /= m.total_mass;
  my /= m.total_mass;
  w(L.lag) = mx;
  w(L.lag + 1) = my;
  for (int i = 0; i < L.npersist; ++i) dev.state[(size_t)i * n + e] = w(i);
  const bool have_cfrc = wrench && !reset;  // a reset leaves mj_resetData's zeros
  float reward = 0.0f;
  double info[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (!reset) {
    double contact_cost = 0.0;
    if ((kStandup || task.use_contact_force) && have_cfrc) {  // humanoid.h:180-187
      for (int i = 0; i < 6 * E::NB; ++i) {
        const double x = w(L.cext + i);
        contact_cost += task.contact_cost_weight * x * x;
      }
      contact_cost = contact_cost < task.contact_cost_max ? contact_cost : task.contact_cost_max;
    }
    const double z = w(L.qpos + 2);
    if constexpr (kStandup) {  // humanoid_standup.h:160-185
      const double xv = z / m.timestep;
      reward = static_cast<float>(xv * task.forward_reward_weight + task.healthy_reward -
                                  ctrl_cost - contact_cost);
      done = cur >= a.max_episode_steps;
      info[0] = xv * task.forward_reward_weight;
      info[1] = -ctrl_cost;
      info[2] = task.healthy_reward;
      info[3] = -contact_cost;
    } else {
      const double xv = (mx - x_before) / task.dt, yv = (my - y_before) / task.dt;
      const bool healthy = task.healthy_z_min < z && z < task.healthy_z_max;
      bool give = healthy;
      if (task.legacy_healthy_reward) give = task.terminate_when_unhealthy || healthy;
      const double healthy_reward = give ? task.healthy_reward : 0.0;
      reward = static_cast<float>(xv * task.forward_reward_weight + healthy_reward - ctrl_cost -
                                  contact_cost);
      done = (task.terminate_when_unhealthy ? !healthy : false) || (cur >= a.max_episode_steps);
      info[0] = xv * task.forward_reward_weight;
      info[1] = -ctrl_cost;
      info[2] = healthy_reward;
      info[3] = -contact_cost;
      info[4] = mx;
      info[5] = my;
      info[6] = sqrt(mx * mx + my * my);
      info[7] = xv;
      info[8] = yv;
    }
  } else {
    // the reset WriteState stores `-ctrl_cost` / `-contact_cost` of +0.0: -0.0 (humanoid.h:272-274)
    info[1] = -0.0;
    info[3] = -0.0;
    if (kStandup) info[2] = task.healthy_reward;  // WriteState(0, 0, 0, 0): reward_alive is the constant
  }
  cm.done[e] = done ? 1 : 0;
  cm.cur_step[e] = cur;
  // WriteState, humanoid.h:225-268
  const int b0 = task.exclude_worldbody ? 1 : 0;
  const int a0 = task.exclude_root_actuator ? 6 : 0;
  const int nobs = (E::NQ - task.obs_skip) + E::NV + 22 * (E::NB - b0) + (E::NV - a0);
  double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * nobs;
  for (int i = task.obs_skip; i < E::NQ; ++i) *(obs++) = w(L.qpos + i);
  for (int i = 0; i < E::NV; ++i) *(obs++) = w(L.qvel + i);
  for (int i = 10 * b0; i < 10 * E::NB; ++i) *(obs++) = i < 10 ? 0.0 : w(L.cinert + i);
  for (int i = 6 * b0; i < 6 * E::NB; ++i) *(obs++) = w(L.cvel + i);
  for (int i = a0; i < E::NV; ++i) *(obs++) = w(L.act + i);
  for (int i = 6 * b0; i < 6 * E::NB; ++i) *(obs++) = have_cfrc ? w(L.cext + i) : 0.0;
  constexpr int ninfo = kStandup ? 4 : 9;
  for (int i = 0; i < ninfo; ++i) ((double*)out.p[kKeyEnv0 + 1 + i])[row] = info[i];
  WriteCommon(out, row, e + a.id_offset, cur, done, reward, a.max_episode_steps);
}

#endif  // EPA_ALT_KERNELS

// flat state like oracle/mjcpu: qpos[24] qvel[23] warm[23] time xlag ylag done cur_step
// normal_saved normal_avail (the last two unused: uniform noise only; xlag / ylag: the lagged
// mass centre)
template <class MP>
__global__ void HumGetState(HumDev dev, CommonDev cm, const int* ids, int k, double* out) {
  using E = T::Tree<MP>;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int e = ids[i], n = cm.n;
  constexpr int np = E::kL.npersist;  // qpos qvel warm lag[2], in this order
  double* o = out + (size_t)i * (np + 5);
  for (int j = 0; j < np - 2; ++j) o[j] = dev.state[(size_t)j * n + e];
  double* t = o + np - 2;
  t[0] = 0;
  t[1] = dev.state[(size_t)(np - 2) * n + e];
  t[2] = dev.state[(size_t)(np - 1) * n + e];
  t[3] = cm.done[e];
  t[4] = cm.cur_step[e];
  t[5] = 0;
  t[6] = 0;
}
template <class MP>
__global__ void HumSetState(HumDev dev, CommonDev cm, const int* ids, int k, const double* in) {
  using E = T::Tree<MP>;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int e = ids[i], n = cm.n;
  constexpr int np = E::kL.npersist;
  const double* o = in + (size_t)i * (np + 5);
  for (int j = 0; j < np - 2; ++j) dev.state[(size_t)j * n + e] = o[j];
  const double* t = o + np - 2;
  dev.state[(size_t)(np - 2) * n + e] = t[1];
  dev.state[(size_t)(np - 1) * n + e] = t[2];
  cm.done[e] = t[3] != 0.0;
  cm.cur_step[e] = (int)t[4];
}

}  // namespace

// The two models are compiled in separate translation units (the step kernel takes ~1.5 min
// each): this file is built twice, EPA_HUM_STANDUP_TU selects which model's kernels and
// launchers it emits; the pool class lives in the Humanoid unit.
#ifdef EPA_HUM_STANDUP_TU
using HumModelP = StandupMP;
#define EPA_HUM_FN(name) name##Standup
constexpr bool kThisStandup = true;
#else
using HumModelP = HumanoidMP;
#define EPA_HUM_FN(name) name##Humanoid
constexpr bool kThisStandup = false;
#endif
void EPA_HUM_FN(HumLaunchStep)(hipStream_t st, int blocks, HumDev dev, CommonDev cm, StepArgs a,
                               const double* act, OutPtrs out, HumTask task) {
#ifdef EPA_ALT_KERNELS
  hipLaunchKernelGGL((HumanoidStepKernel<HumModelP, kThisStandup>), dim3(blocks), dim3(kHumBlock), 0,
                     st, dev, cm, a, act, out, task);
#else
  (void)kThisStandup;
  throw std::runtime_error("the one-env-per-lane Humanoid kernel is not in this build (make EPA_ALT_KERNELS=1)");
#endif
}
void EPA_HUM_FN(HumLaunchGet)(hipStream_t st, int k, HumDev dev, CommonDev cm, const int* ids,
                              double* out) {
  hipLaunchKernelGGL(HumGetState<HumModelP>, dim3((k + 255) / 256), dim3(256), 0, st, dev, cm, ids,
                     k, out);
}
void EPA_HUM_FN(HumLaunchSet)(hipStream_t st, int k, HumDev dev, CommonDev cm, const int* ids,
                              const double* in) {
  hipLaunchKernelGGL(HumSetState<HumModelP>, dim3((k + 255) / 256), dim3(256), 0, st, dev, cm, ids,
                     k, in);
}
int EPA_HUM_FN(HumWorkspaceSlots)() { return T::Tree<HumModelP>::kL.total; }

#ifndef EPA_HUM_STANDUP_TU
void HumLaunchStepStandup(hipStream_t, int, HumDev, CommonDev, StepArgs, const double*, OutPtrs,
                          HumTask);
void HumLaunchGetStandup(hipStream_t, int, HumDev, CommonDev, const int*, double*);
void HumLaunchSetStandup(hipStream_t, int, HumDev, CommonDev, const int*, const double*);
int HumWorkspaceSlotsStandup();

namespace {

int HumObsDim(const Config& cfg) {
  // humanoid.h:50-60: 376 (378 with positions), minus the world body's cinert / cvel /
  // cfrc_ext (22) and the free joint's qfrc_actuator (6) for the v5 ids
  int n = cfg.Get("exclude_current_positions_from_observation", 1) != 0 ? 376 : 378;
  if (cfg.Get("exclude_worldbody_observations", 0) != 0) n -= 22;
  if (cfg.Get("exclude_root_actuator_forces", 0) != 0) n -= 6;
  return n;
}

std::vector<KeySpec> HumKeys(const Config& cfg, bool standup) {
  std::vector<KeySpec> k = {{"obs", EPA_F64, StackedObsShape(cfg, HumObsDim(cfg))}};
  if (standup) {  // humanoid_standup.h:62-65
    for (const char* name : {"info:reward_linup", "info:reward_quadctrl", "info:reward_alive",
                             "info:reward_impact"}) {
      k.push_back({name, EPA_F64, {}});
    }
  } else {  // humanoid.h:66-74
    for (const char* name :
         {"info:reward_linvel", "info:reward_quadctrl", "info:reward_alive", "info:reward_impact",
          "info:x_position", "info:y_position", "info:distance_from_origin", "info:x_velocity",
          "info:y_velocity"}) {
      k.push_back({name, EPA_F64, {}});
    }
  }
  return k;
}

class HumanoidPool : public Pool {
 public:
  HumanoidPool(const Config& cfg, bool standup)
      : Pool(cfg, HumKeys(cfg, standup), KeySpec{"action", EPA_F64, {kHumanoidModelConst.nu}},
             /*needs_rng=*/true),
        standup_(standup) {
    EnableObsStack();
    // defaults: humanoid.h:32-48, humanoid_standup.h:32-45
    task_.frame_skip = (int)cfg.Get("frame_skip", 5);
    task_.obs_skip = cfg.Get("exclude_current_positions_from_observation", 1) != 0 ? 2 : 0;
    task_.terminate_when_unhealthy = cfg.Get("terminate_when_unhealthy", 1) != 0;
    task_.legacy_healthy_reward = cfg.Get("legacy_healthy_reward", 1) != 0;
    task_.use_contact_force = cfg.Get("use_contact_force", 0) != 0;
    task_.post_constraint = cfg.Get("post_constraint", 1) != 0;
    task_.exclude_worldbody = cfg.Get("exclude_worldbody_observations", 0) != 0;
    task_.exclude_root_actuator = cfg.Get("exclude_root_actuator_forces", 0) != 0;
    task_.ctrl_cost_weight = cfg.Get("ctrl_cost_weight", 0.1);
    task_.forward_reward_weight = cfg.Get("forward_reward_weight", standup ? 1.0 : 1.25);
    task_.healthy_reward = cfg.Get("healthy_reward", standup ? 1.0 : 5.0);
    task_.healthy_z_min = cfg.Get("healthy_z_min", 1.0);
    task_.healthy_z_max = cfg.Get("healthy_z_max", 2.0);
    task_.reset_noise_scale = cfg.Get("reset_noise_scale", 1e-2);
    task_.contact_cost_weight = cfg.Get("contact_cost_weight", 5e-7);
    task_.contact_cost_max = cfg.Get("contact_cost_max", 10.0);
    task_.dt = task_.frame_skip * kHumanoidModelConst.timestep;
    // "hum_debug" switches stages of the quad kernel off (timing probes) or routes solver statistics
    // into the info keys: only in the diagnostic build (tools/build_trace_lib.sh, -DEPA_HUM_DEBUG);
    // the product library refuses it, so no measured figure can come from disabled physics
    task_.debug = (int)cfg.Get("hum_debug", 0);
#ifndef EPA_HUM_DEBUG
    if (task_.debug != 0) {
      throw std::invalid_argument("hum_debug needs the diagnostic build (tools/build_trace_lib.sh)");
    }
#endif
    // "hum_layout": 1 (default) one env per lane quad (mj_hum4.hip.h), 0 one env per lane with
    // the HBM workspace (mj_tree.hip.h; kept for A/B runs)
    quad_ = cfg.Get("hum_layout", 1) != 0;
#ifndef EPA_ALT_KERNELS
    if (!quad_) {
      throw std::invalid_argument("\"hum_layout\" = 0 (the superseded one-env-per-lane Humanoid kernel) is only built "
                                  "into the alternate library: make -C envpool_amd/csrc EPA_ALT_KERNELS=1 and load "
                                  "lib/libenvpool_amd_alt.so (ENVPOOL_AMD_LIB)");
    }
#endif
    // "hum_sort": 1 (default) waves are formed from envs of similar solver cost (Hum4SortKernel)
    sort_ = quad_ && cfg.Get("hum_sort", 1) != 0;
    EPA_HIP(hipMalloc(&dev_.cost, sizeof(int) * (size_t)cfg.num_envs));
    EPA_HIP(hipMemsetAsync(dev_.cost, 0, sizeof(int) * (size_t)cfg.num_envs, stream_));
    if (!quad_) {  // (the quad kernel's per-launch scratch is per compute stream: ScratchFor)
      const size_t blocks = ((size_t)cfg.num_envs + 63) / 64;
      ws_bytes_ = sizeof(double) * blocks * 64 * (size_t)Total();
      EPA_HIP(hipMalloc(&dev_.ws, ws_bytes_));
      EPA_HIP(hipMemsetAsync(dev_.ws, 0, ws_bytes_, stream_));
    }
    const size_t sb = sizeof(double) * (size_t)T::MakeLayout(kHumanoidModelConst).npersist * cfg.num_envs;
    EPA_HIP(hipMalloc(&dev_.state, sb));
    EPA_HIP(hipMemsetAsync(dev_.state, 0, sb, stream_));
    mt_tile_default_ = 16;  // unhealthy terminations: every env resets at its own time
    InitCommon();
  }
  ~HumanoidPool() override {
    if (!quad_) (void)hipFree(dev_.ws);
    for (auto& kv : scratch_) {
      (void)hipFree(kv.second.ws);
      (void)hipFree(kv.second.perm);
    }
    (void)hipFree(big_.ws);
    (void)hipFree(big_.perm);
    if (big_ev_) (void)hipEventDestroy(big_ev_);
    (void)hipFree(dev_.state);
    (void)hipFree(dev_.cost);
  }
  // The quad kernel keeps what persists per ENV (dev_.state, dev_.cost); its workspace and the cost-sort
  // permutation belong to a LAUNCH, so with one copy per compute stream batches of an async pool run
  // concurrently like every other family's (round 4; they used to share one copy and one stream).
  bool ConcurrentSafe() const override { return quad_; }
  int StateDim() const override { return kHumanoidModelConst.nq + 2 * kHumanoidModelConst.nv + 7; }
  void GetState(const int* d_ids, int k, double* d_out) override {
    (standup_ ? HumLaunchGetStandup : HumLaunchGetHumanoid)(stream_, k, dev_, common_, d_ids, d_out);
  }
  void SetState(const int* d_ids, int k, const double* d_in) override {
    (standup_ ? HumLaunchSetStandup : HumLaunchSetHumanoid)(stream_, k, dev_, common_, d_ids, d_in);
  }

 protected:
  void Launch(const int* d_ids, int k, const void* d_action, bool force_reset,
              const OutPtrs& out) override {
    StepArgs a{d_ids, k, force_reset ? 1 : 0, cfg_.max_episode_steps, cfg_.env_id_offset};
    if (quad_) {
      const Scratch& sc = ScratchFor(stream_, k);
      HumDev dev = dev_;
      dev.ws = sc.ws;
      dev.perm = nullptr;
      if (sort_ && k > 16) {
        dev.perm = sc.perm;
        Hum4LaunchSort(stream_, dev, a);
      }
      Hum4LaunchStep(stream_, standup_, (k + 15) / 16, dev, common_, a, static_cast<const double*>(d_action),
                     out, task_);
      if (big_used_) {
        EPA_HIP(hipEventRecord(big_ev_, stream_));
        big_used_ = false;
      }
      return;
    }
    const int blocks = (k + kHumBlock - 1) / kHumBlock;
    (standup_ ? HumLaunchStepStandup : HumLaunchStepHumanoid)(
        stream_, blocks, dev_, common_, a, static_cast<const double*>(d_action), out, task_);
  }

 private:
  int Total() const {
    return standup_ ? HumWorkspaceSlotsStandup() : HumWorkspaceSlotsHumanoid();
  }
  // per-launch scratch of the quad kernel on compute stream `st`: one copy per stream, sized ONCE for the pool's
  // usual launch (batch_size in async mode, else num_envs).  A bigger launch (the reset of all envs of an async pool
  // in one send) does not regrow it -- that took a hipFree, which waits for the whole device and stalled every
  // other compute stream mid-run -- but goes to ONE shared copy sized for num_envs, allocated at the first such
  // launch; launches that use it are chained by an event, whichever stream they run on.  Bound on the memory:
  // compute_streams x Hum4WorkspaceBytes(batch_size) + Hum4WorkspaceBytes(num_envs) (include/envpool_amd.h).
  struct Scratch {
    double* ws{nullptr};
    int* perm{nullptr};
    int rows{0};
  };
  void AllocScratch(Scratch& sc, int rows, hipStream_t st) {
    sc.rows = rows;
    const size_t bytes = Hum4WorkspaceBytes(sc.rows);
    EPA_HIP(hipMalloc(&sc.ws, bytes));
    EPA_HIP(hipMemsetAsync(sc.ws, 0, bytes, st));
    EPA_HIP(hipMalloc(&sc.perm, sizeof(int) * (size_t)sc.rows));
  }
  const Scratch& ScratchFor(hipStream_t st, int k) {
    const bool async = cfg_.batch_size > 0 && cfg_.batch_size < cfg_.num_envs;
    const int usual = async ? cfg_.batch_size : cfg_.num_envs;
    if (k <= usual) {
      Scratch& sc = scratch_[st];
      if (sc.ws == nullptr) AllocScratch(sc, usual, st);
      return sc;
    }
    if (big_.ws == nullptr) {
      AllocScratch(big_, cfg_.num_envs, st);
      EPA_HIP(hipEventCreateWithFlags(&big_ev_, hipEventDisableTiming));
    } else {
      EPA_HIP(hipStreamWaitEvent(st, big_ev_, 0));  // behind the previous launch that used the shared copy
    }
    big_used_ = true;
    return big_;
  }
  HumDev dev_{};
  HumTask task_{};
  size_t ws_bytes_{0};
  bool standup_;
  bool quad_{true}, sort_{true};
  std::map<hipStream_t, Scratch> scratch_;
  Scratch big_;  // shared by the launches that exceed the usual size
  hipEvent_t big_ev_{nullptr};
  bool big_used_{false};
};

}  // namespace

bool DescribeHumanoid(const std::string& family, const Config& cfg, std::vector<KeySpec>* state,
                      KeySpec* action) {
  if (family != "Humanoid" && family != "HumanoidStandup") return false;
  *state = HumKeys(cfg, family == "HumanoidStandup");
  *action = KeySpec{"action", EPA_F64, {kHumanoidModelConst.nu}};
  return true;
}

Pool* MakeHumanoid(const std::string& family, const Config& cfg) {
  if (family == "Humanoid") return new HumanoidPool(cfg, false);
  if (family == "HumanoidStandup") return new HumanoidPool(cfg, true);
  return nullptr;
}
#endif  // !EPA_HUM_STANDUP_TU

}  // namespace epa
