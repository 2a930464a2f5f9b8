// K3b — gym-MuJoCo Ant batched step kernel: ONE ENV PER LANE QUAD (one lane per
// leg, 16 envs per wavefront, one wave per block), see mj_ant4.hip.h.
//
// Replaces, for the whole batch in one launch:
//   MujocoEnv::{MujocoReset,MujocoStep}   envpool/mujoco/gym/mujoco_env.h:126-148
//   AntEnvBase::{MujocoResetModel,Reset,Step,IsHealthy,WriteState}
//                                         envpool/mujoco/gym/ant.h:135-278
// with `frame_skip x mj_step` (RK4: 4 forward passes each) from mj_ant4.hip.h.
// Ant-v4: use_contact_force=false (no cfrc_ext in obs, contact cost 0).
// Ant-v3: use_contact_force=true but post_constraint=false: MuJoCo 3 fills
//   cfrc_ext only in mj_rnePostConstraint, which the reference then never calls
//   (mujoco_env.h:145-147) => 84 zeros in the obs, contact cost 0.
// Ant-v5: use_contact_force + post_constraint: cfrc_ext of the last forward
//   evaluation (mj_ant4.hip.h, ContactWrench), world body excluded.
//
// Persistent state (SoA fp64): qpos[15][N], qvel[14][N], qacc_warmstart[14][N],
// lag[2][N] = data_->xpos[torso].xy of the last forward pass (the reference
// reads the *lagged* torso position, ant.h:169-173 / SURVEY §7 H3), and the
// env's normal_distribution saved value.  Lane l of a quad loads / stores the
// torso part (replicated) and the two dofs of leg l; lane 0 writes what is per env.
#define EPA_SINCOS_MODE 1
#include "device_common.hip.h"
#include "engine.h"
#if defined(EPA_ANT_TIMERS) && defined(__HIP_DEVICE_COMPILE__)
// diagnostic build (tools/build_ant_timers.sh): per-wave stage timers in LDS, summed into g_ant_timers when the wave
// leaves the queue; categories: mj_ant4.hip.h
namespace epa {
__shared__ unsigned long long ant_t_acc[16];  // [0] last tick, [1..8] cycles per category, [9..15] counts
__device__ unsigned long long g_ant_timers[16];
}  // namespace epa
#define EPA_ANT_TICK(K)                                   \
  do {                                                    \
    if (threadIdx.x == 0) {                               \
      const unsigned long long c_ = clock64();            \
      ::epa::ant_t_acc[1 + (K)] += c_ - ::epa::ant_t_acc[0]; \
      ::epa::ant_t_acc[0] = c_;                           \
    }                                                     \
  } while (0)
#define EPA_ANT_COUNT(K)                                  \
  do {                                                    \
    if (threadIdx.x == 0) ::epa::ant_t_acc[9 + (K)] += 1; \
  } while (0)
// classes per forward pass: [12] the wave's union (what the class loops run over), [13] the busiest lane's own,
// [14] the busiest QUAD's union (sum over its lanes' own sets, as a set)
#define EPA_ANT_CLASSES(sph, own)                                                              \
  do {                                                                                         \
    const int mine_ = __builtin_popcount((unsigned)(own));                                     \
    int mx_ = 0;                                                                               \
    for (int b_ = 1; b_ <= 7; ++b_) mx_ = __builtin_amdgcn_ballot_w64(mine_ >= b_) ? b_ : mx_; \
    if (threadIdx.x == 0) {                                                                    \
      ::epa::ant_t_acc[12] += __builtin_popcount(sph);                                         \
      ::epa::ant_t_acc[13] += mx_;                                                             \
    }                                                                                          \
  } while (0)
#elif defined(EPA_ANT_TIMERS)
namespace epa {
__device__ unsigned long long g_ant_timers[16];
}
#endif
#include "mj_ant4.hip.h"
#include "mj_ant_model.h"
#include "build/mj_ant_consts.inc"  // generated: kAntModelConst (gen_mj_consts.cpp)

namespace epa {
namespace {

namespace A = mj::ant;
namespace A4 = mj::ant4;

static_assert(A4::CheckLegSymmetry(kAntModelConst),
              "ant_envpool.xml no longer has the mirror structure mj_ant4.hip.h relies on");

struct AntDev {
  double* qpos;  // [15][N]
  double* qvel;  // [14][N]
  double* warm;  // [14][N]
  double* lag;   // [2][N]
  double* nsaved;
  unsigned char* navail;
  double* cost;  // [N] profiling: Newton iterations of the last step, see AntGetState
  // diagnostic (EPA_ANT_TRACE=<file>): per wave of the last launch {wall clock begin, end
  // (100 MHz), core clock begin, end, slot, HW_ID}; nullptr otherwise
  long long* trace;
};

constexpr int kAntMjBodies = 14;  // world + torso + 4 x (stub, leg, ankle) MuJoCo bodies

struct AntTask {
  int frame_skip, obs_skip;
  int terminate_when_unhealthy, legacy_healthy_reward;
  int use_contact_force, post_constraint, exclude_worldbody;
  double ctrl_cost_weight, forward_reward_weight, healthy_reward;
  double healthy_z_min, healthy_z_max, reset_noise_scale, dt;
  double contact_cost_weight, contact_force_min, contact_force_max;
};

constexpr int kAntBlock = 64;                 // one wavefront
constexpr int kAntEnvsPerBlock = kAntBlock / 4;

// waves per SIMD the register allocator targets: fp64 needs the whole 512-entry file
// (486 registers, no scratch); fp32 fits two waves (25 spilled registers) and gains 50 %
// from the second wave (profiles/archive/r2b: 1.94e7 -> 2.93e7 env-steps/s at N=32768)
template <typename T>
constexpr int kAntWavesPerEu = sizeof(T) == 4 ? 2 : 1;

// ---- the work queue -------------------------------------------------------------------------
// A launch is a queue of UNITS served by persistent waves (grid = the waves resident at once; tickets from one
// device counter).  A unit is (chunk of 16 rows, `sub` consecutive mj_steps of the env-step): with sub = 1 an
// env-step of a chunk is frame_skip units of ~1/5 of the work, and the state of the chunk's envs goes through HBM
// between them (43 numbers per env: nothing next to 4e5 flops per mj_step).  Why: a wave runs as long as the
// slowest of its 16 envs in every one of the 20 forward passes, a chunk's env-step lasts 0.5 ms on average and
// 1.1 ms at worst, and with whole env-steps as the unit of work (round 2-5: one block per chunk) the launch at
// N = 32768 -- two chunks per SIMD -- ended with a few long waves running alone: 1.45 ms where the sum of the wave
// durations / 1024 SIMDs is 1.04.  Units are handed out substep-major (every chunk's units of substep range 0,
// then range 1, ...) so that what is left at the end of the launch is short units.  Unit (c, j + 1) reads what
// (c, j) wrote, maybe on another XCD: (c, j) publishes `progress[c] = epoch + j + 1` behind a device-scope fence
// and the taker of (c, j + 1) -- a LATER ticket, so (c, j) is running or done: no deadlock, every ticket is taken
// with the atomic by a wave that is resident -- waits for it.
// Dealing the rows to chunks in descending order of the Newton iterations their env needed in its previous env-step
// (the cost-ordered dispatch of round 2: +7 % then, at a persistence of 0.64) was rebuilt on this queue and measured
// out: since the one-evaluation line search an env's trip count is noise from step to step (correlation 0.098,
// profiles/r6b_ant_iter_stats.txt), grouping by it leaves the wave maxima where they are (48.2 vs 48.3) and the
// sort plus the scattered state accesses cost 5 % (profiles/r6b_ant_queue_ab.md).  Not kept.
struct AntArgs {
  AntDev dev;
  CommonDev cm;
  StepArgs a;
  const double* action;
  OutPtrs out;
  AntTask task;
  unsigned* ticket;
  unsigned ticket_base;
  int nchunks;           // ceil(k / 16)
  int sub;               // mj_steps per unit
  int units_per_chunk;   // ceil(frame_skip / sub)
  unsigned char* rowflag;  // [k] 1: the row was a reset row of this launch (units after the first skip it)
  unsigned* progress;    // [nchunks] epoch + units of the chunk completed
  unsigned epoch;
  int max_iter;          // mj::SolverCfg
  double gtol;
};
using AntArgsK = const __attribute__((address_space(4))) AntArgs;
#if defined(__HIP_DEVICE_COMPILE__)
// the kernel's arguments re-read from the kernarg segment per unit (see mujoco_planar_lg.hip, KernArgs): as
// by-value parameters they would be loop invariants of the persistent loop held in SGPRs across the solver
__device__ __forceinline__ AntArgsK* AntKernArgs() {
  AntArgsK* p = (AntArgsK*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
#endif

// One unit: mj_steps [s0, s1) of the env-step of the rows at positions 16 ci .. 16 ci + 15.
// kWrench: the Ant-v5 variant that also evaluates cfrc_ext (separate instantiation
// so that Ant-v3/v4 do not pay registers for the extra pass)
template <typename T, bool kWrench>
__device__ __forceinline__ void AntUnit(int ci, int s0, int s1, T* lds_buf) {
#if defined(__HIP_DEVICE_COMPILE__)
  // The kernel arguments are read from the kernarg segment where they are used, before AND again after the physics
  // (486 of the 512 registers are its own, and SGPRs held across it spill): `ap` is refreshed behind a barrier the
  // optimiser cannot see through, and nothing per-row -- the row, its env, the output pointers -- stays live either.
  AntArgsK* ap = AntKernArgs();
#define task (ap->task)
#define out (ap->out)
// WriteState, ant.h:231-278: obs = qpos[skip:] ++ qvel ++ clamped cfrc_ext
#define cf0 (task.exclude_worldbody ? 6 : 0)
#define ncf (task.use_contact_force ? kAntMjBodies * 6 - cf0 : 0)
#define nobs (A::kNQ + A::kNV - task.obs_skip + ncf)
  constexpr A::AntModel<T> m = A::CastAntModel<T>(kAntModelConst);
  const mj::SolverCfg<T> scfg{ap->max_iter, (T)ap->gtol};
  const int lane = threadIdx.x;
  const int l = lane & 3;  // the leg this lane owns
  if (ci * kAntEnvsPerBlock + (lane >> 2) >= ap->a.k) return;  // whole quads leave together
  const bool first_unit = s0 == 0;
  T q[9], v[A4::kL], w[A4::kL], ctrl[2];
  {
  const AntDev dev = ap->dev;
  const CommonDev cm = ap->cm;
  const StepArgs a = ap->a;
  const int n = cm.n;
  const int pos = ci * kAntEnvsPerBlock + (lane >> 2);
  const bool last_unit = s1 >= task.frame_skip;
  const int row = pos;
  const int e = a.ids ? a.ids[row] - a.id_offset : row;
  bool reset;
  if (first_unit) {
    reset = a.force_reset || cm.done[e] != 0;
    if (!last_unit && l == 0) ap->rowflag[row] = reset ? 1 : 0;
  } else {
    reset = ap->rowflag[row] != 0;
  }
  if (reset) {
    if (!first_unit || l != 0) return;  // the RNG stream of an env is sequential: its first lane draws
    double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * nobs;
    double* obs_v = obs + A::kNQ - task.obs_skip;
    double* obs_c = obs_v + A::kNV;
    // MujocoReset (mujoco_env.h:126-131) + MujocoResetModel (ant.h:135-147)
    double qpos[A::kNQ], qvel[A::kNV];
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
    cm.done[e] = 0;
    cm.cur_step[e] = 0;
    for (int i = task.obs_skip; i < A::kNQ; ++i) obs[i - task.obs_skip] = qpos[i];
    for (int i = 0; i < A::kNV; ++i) obs_v[i] = qvel[i];
    // a reset (mj_resetData) leaves cfrc_ext at zero; ant.h:248-258 clamps it
    double z = 0.0;
    z = z > task.contact_force_min ? z : task.contact_force_min;
    z = z < task.contact_force_max ? z : task.contact_force_max;
    for (int i = 0; i < ncf; ++i) obs_c[i] = z;
    // WriteState(0.0, 0, ...): info[6] = sqrt(0); reward_ctrl / reward_contact are stored as
    // `-ctrl_cost` / `-contact_cost` of +0.0, i.e. -0.0 (ant.h:262-263)
    for (int i = 0; i < 9; ++i) ((double*)out.p[kKeyEnv0 + 1 + i])[row] = (i == 1 || i == 2) ? -0.0 : 0.0;
    WriteCommon(out, row, e + a.id_offset, 0, false, 0.0f, a.max_episode_steps);
    return;
  }
  // lane layout: q = torso pose (7) + hip, ankle of leg l; v, w likewise (6 + 2)
  mj::static_for<0, 7>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    q[i] = (T)dev.qpos[(size_t)i * n + e];
  });
  mj::static_for<0, 6>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v[i] = (T)dev.qvel[(size_t)i * n + e];
    w[i] = (T)dev.warm[(size_t)i * n + e];
  });
  mj::static_for<0, 2>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    q[7 + c] = (T)dev.qpos[(size_t)(7 + 2 * l + c) * n + e];
    v[6 + c] = (T)dev.qvel[(size_t)(6 + 2 * l + c) * n + e];
    w[6 + c] = (T)dev.warm[(size_t)(6 + 2 * l + c) * n + e];
  });
  const double* act0 = ap->action + (size_t)row * A::kNU;
  // ctrl[u] drives dof CtrlDof(u): hip_4 ankle_4 hip_1 ankle_1 hip_2 ... (ant_envpool.xml:85-94)
  mj::static_for<0, 2>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const double ai = act0[(2 + 2 * l + c) & 7];
    ctrl[c] = (T)(ai < -1.0 ? -1.0 : (ai > 1.0 ? 1.0 : ai));
  });
  }
  A4::Leg<T, bool> lg;
  lg.sx = (l == 0 || l == 3) ? T(1) : T(-1);
  lg.sy = l < 2 ? T(1) : T(-1);
  lg.sxy = lg.sx * lg.sy;
  lg.axs = (l & 1) ? T(1) : T(-1);
  lg.alo = (l == 0 || l == 3) ? m.lo[1] : m.lo[3];
  lg.ahi = (l == 0 || l == 3) ? m.hi[1] : m.hi[3];
  lg.first = l == 0;
  T lagx = T(0), lagy = T(0);
  auto lds = [&](int slot) -> T& { return lds_buf[A4::LdsOffset(slot, lane)]; };
  // mj_rnePostConstraint after the last mj_step (mujoco_env.h:145-147): cfrc_ext of the
  // lane's stub / leg / ankle bodies, and of the torso body on the first lane
  T cf[3][6], cft[6];
  if constexpr (kWrench) {
    for (int j = 0; j < 6; ++j) cft[j] = cf[0][j] = cf[1][j] = cf[2][j] = T(0);
  }
  T n_env = T(0);
  int n_wave = 0;
#ifdef EPA_WAVE_TRACE  // diagnostic build only (tools/build_trace_lib.sh)
  const long long t_begin = clock64();
  const long long w_begin = wall_clock64();
#endif
  EPA_ANT_TICK(0);
  for (int s = s0; s < s1; ++s) {
    A4::Step<unsigned, kWrench>(m, lg, scfg, q, v, w, ctrl, &lagx, &lagy, lds,
                                kWrench && s == task.frame_skip - 1, cf, cft, &n_env, &n_wave);
  }
  EPA_ANT_TICK(6);
  ap = AntKernArgs();
  const AntDev dev = ap->dev;
  const CommonDev cm = ap->cm;
  const StepArgs a = ap->a;
  const int n = cm.n;
  const bool last_unit = s1 >= task.frame_skip;
  int pos2 = ci * kAntEnvsPerBlock + (lane >> 2);
  asm volatile("" : "+v"(pos2));
  const int row = pos2;
  const int e = a.ids ? a.ids[row] - a.id_offset : row;
  double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * nobs;
  double* obs_v = obs + A::kNQ - task.obs_skip;
  double* obs_c = obs_v + A::kNV;
  const double* act = ap->action + (size_t)row * A::kNU;
  // new state; IsHealthy, ant.h:214-229
  bool healthy = true;
  mj::static_for<0, 2>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const double qq = (double)q[7 + c], vv = (double)v[6 + c];
    healthy = healthy && isfinite(qq) && isfinite(vv);
    dev.qpos[(size_t)(7 + 2 * l + c) * n + e] = qq;
    dev.qvel[(size_t)(6 + 2 * l + c) * n + e] = vv;
    dev.warm[(size_t)(6 + 2 * l + c) * n + e] = (double)w[6 + c];
    if (last_unit) {
      if (7 + 2 * l + c >= task.obs_skip) obs[7 + 2 * l + c - task.obs_skip] = qq;
      obs_v[6 + 2 * l + c] = vv;
    }
  });
  // this env's Newton iterations + 1e4 x (those its wave executed + 1e3 x sphere classes visited)
  // + 1e10 x thousands of core clocks the wave spent in the physics; summed over the units of the env-step
  double cost = (double)n_env + 1.0e4 * (double)(n_wave % 1000000);
#ifdef EPA_WAVE_TRACE
  cost += 1.0e10 * (double)((clock64() - t_begin) / 1000);
  if (dev.trace && lane == 0 && first_unit) {
    long long* tr = dev.trace + (size_t)ci * 6;
    tr[0] = w_begin;
    tr[1] = wall_clock64();
    tr[2] = t_begin;
    tr[3] = clock64();
    tr[4] = ci * kAntEnvsPerBlock;
    tr[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
  }
#endif
  if (!last_unit) {  // the state goes through HBM to the wave that takes the chunk's next unit
    if (l != 0) return;
    mj::static_for<0, 7>([&](auto ic) { dev.qpos[(size_t)decltype(ic)::value * n + e] = (double)q[decltype(ic)::value]; });
    mj::static_for<0, 6>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dev.qvel[(size_t)i * n + e] = (double)v[i];
      dev.warm[(size_t)i * n + e] = (double)w[i];
    });
    dev.cost[e] = (first_unit ? 0.0 : dev.cost[e]) + cost;
    return;
  }
  const double x_before = dev.lag[e], y_before = dev.lag[(size_t)n + e];
  const double x_after = (double)lagx, y_after = (double)lagy;
  mj::static_for<0, 7>([&](auto ic) { healthy = healthy && isfinite((double)q[decltype(ic)::value]); });
  mj::static_for<0, 6>([&](auto ic) { healthy = healthy && isfinite((double)v[decltype(ic)::value]); });
  healthy = mj::All4(healthy);
  const double z = (double)q[2];
  if (z < task.healthy_z_min || z > task.healthy_z_max) healthy = false;
  double contact_cost = 0.0;
  if constexpr (kWrench) {  // ant.h:183-194 and :248-258
    auto clampc = [&](double x) {  // cost: std::max(min_, std::min(max_, x))
      x = task.contact_force_max < x ? task.contact_force_max : x;
      return task.contact_force_min > x ? task.contact_force_min : x;
    };
    auto clampo = [&](double x) {  // obs: std::min(std::max(x, min_), max_)
      x = x > task.contact_force_min ? x : task.contact_force_min;
      return x < task.contact_force_max ? x : task.contact_force_max;
    };
    double part = 0.0;
    for (int j = 0; j < 6; ++j) {
      // the world body receives the opposite of every contact wrench
      const double mine = (double)cf[0][j] + (double)cf[1][j] + (double)cf[2][j] + (double)cft[j];
      const double world = -mj::Sum4(mine);
      for (int b = 0; b < 3; ++b) {
        const double x = (double)cf[b][j];
        part += clampc(x) * clampc(x);
        obs_c[(2 + 3 * l + b) * 6 + j - cf0] = clampo(x);
      }
      if (l == 0) {
        const double x = (double)cft[j];
        part += clampc(x) * clampc(x);
        obs_c[6 + j - cf0] = clampo(x);
        if (!task.exclude_worldbody) {
          part += clampc(world) * clampc(world);
          obs_c[j] = clampo(world);
        }
      }
    }
    contact_cost = task.contact_cost_weight * mj::Sum4(part);
  } else {
    // Ant-v3 (use_contact_force without post_constraint): cfrc_ext stays zero
    if (ncf > 0) {
      double zz = 0.0;
      zz = zz > task.contact_force_min ? zz : task.contact_force_min;
      zz = zz < task.contact_force_max ? zz : task.contact_force_max;
      for (int i = l; i < ncf; i += 4) obs_c[i] = zz;
    }
  }
  if (l != 0) return;
  // per-env outputs: the first lane of the quad
  mj::static_for<0, 7>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const double qq = (double)q[i];
    dev.qpos[(size_t)i * n + e] = qq;
    if (i >= task.obs_skip) obs[i - task.obs_skip] = qq;
  });
  mj::static_for<0, 6>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const double vv = (double)v[i];
    dev.qvel[(size_t)i * n + e] = vv;
    dev.warm[(size_t)i * n + e] = (double)w[i];
    obs_v[i] = vv;
  });
  dev.lag[e] = x_after;
  dev.lag[(size_t)n + e] = y_after;
  dev.cost[e] = (first_unit ? 0.0 : dev.cost[e]) + cost;
  const int cur = cm.cur_step[e] + 1;
  double ctrl_cost = 0.0;
  for (int i = 0; i < A::kNU; ++i) ctrl_cost += task.ctrl_cost_weight * act[i] * act[i];  // ant.h:176-179
  const double xv = (x_after - x_before) / task.dt;
  const double yv = (y_after - y_before) / task.dt;
  bool give = healthy;
  if (task.legacy_healthy_reward) give = task.terminate_when_unhealthy || healthy;
  const double healthy_reward = give ? task.healthy_reward : 0.0;
  const float reward = static_cast<float>(xv * task.forward_reward_weight + healthy_reward -
                                          ctrl_cost - contact_cost);
  const bool done = (task.terminate_when_unhealthy ? !healthy : false) || (cur >= a.max_episode_steps);
  const double info[9] = {xv * task.forward_reward_weight, -ctrl_cost, -contact_cost,
                          healthy_reward, x_after, y_after,
                          sqrt(x_after * x_after + y_after * y_after), xv, yv};
  cm.done[e] = done ? 1 : 0;
  cm.cur_step[e] = cur;
  for (int i = 0; i < 9; ++i) ((double*)out.p[kKeyEnv0 + 1 + i])[row] = info[i];
  WriteCommon(out, row, e + a.id_offset, cur, done, reward, a.max_episode_steps);
#undef task
#undef out
#undef cf0
#undef ncf
#undef nobs
#endif
}

// waves per SIMD the register allocator targets: fp64 needs the whole 512-entry file
// (486 registers, no scratch); fp32 fits two waves (25 spilled registers) and gains 50 %
// from the second wave (profiles/archive/r2b: 1.94e7 -> 2.93e7 env-steps/s at N=32768)
template <typename T, bool kWrench>
__global__ __launch_bounds__(kAntBlock)
__attribute__((amdgpu_waves_per_eu(kAntWavesPerEu<T>, kAntWavesPerEu<T>))) void AntStepKernel(AntArgs args) {
  // the wave's LDS block: quad-shared and lane-private slots (mj_ant4.hip.h, LdsOffset): local M,
  // the contact geometry and the contact constants of the current forward pass
  __shared__ T lds_buf[A4::kLdsElems];
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef EPA_ANT_TIMERS
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) ant_t_acc[i] = 0;
    ant_t_acc[0] = clock64();
  }
  const long long t_wave0 = clock64();
#endif
  for (;;) {
    AntArgsK* ap = AntKernArgs();
    unsigned t = 0;
    if (threadIdx.x == 0) t = atomicAdd(ap->ticket, 1u);
    const int u = (int)((unsigned)__builtin_amdgcn_readfirstlane(t) - ap->ticket_base);
    if (u >= ap->nchunks * ap->units_per_chunk) break;
    const int j = u / ap->nchunks, ci = u - j * ap->nchunks;  // substep-major
    if (j > 0) {  // the chunk's previous unit: a lower ticket, so it is running or done
      const unsigned want = ap->epoch + (unsigned)j;
      if (threadIdx.x == 0) {
        while ((int)(__hip_atomic_load(&ap->progress[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
          __builtin_amdgcn_s_sleep(16);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    EPA_ANT_TICK(7);
    EPA_ANT_COUNT(2);
    const int s0 = j * ap->sub;
    const int s1 = s0 + ap->sub;
    AntUnit<T, kWrench>(ci, s0, s1, lds_buf);
    ap = AntKernArgs();
    if (j + 1 < ap->units_per_chunk) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the wave's state stores, before the flag
      if (threadIdx.x == 0) {
        __hip_atomic_store(&ap->progress[ci], ap->epoch + (unsigned)j + 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    EPA_ANT_TICK(0);
  }
#ifdef EPA_ANT_TIMERS
  if (threadIdx.x == 0) {
    for (int i = 0; i < 11; ++i) atomicAdd(&g_ant_timers[i], ant_t_acc[1 + i]);
    atomicAdd(&g_ant_timers[11], (unsigned long long)(clock64() - t_wave0));  // the wave's life
    atomicAdd(&g_ant_timers[12], 1ull);                                        // waves
    atomicAdd(&g_ant_timers[13], ant_t_acc[12]);
    atomicAdd(&g_ant_timers[14], ant_t_acc[13]);
  }
#endif
#endif
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
  t[0] = dev.cost[e];  // (oracle: time) profiling counters of the last step
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
  // ant.h:51-75 (obs 27/29 + 6 per body with use_contact_force); StackSpec, frame_stack.h:42-71
  int ncf = 0;
  if (cfg.Get("use_contact_force", 0) != 0) {
    ncf = 6 * (kAntMjBodies - (cfg.Get("exclude_worldbody_contact_forces", 0) != 0 ? 1 : 0));
  }
  std::vector<KeySpec> k = {{"obs", EPA_F64, StackedObsShape(cfg, (no_pos ? 27 : 29) + ncf)}};
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
  bool ConcurrentSafe() const override { return true; }  // per-env state + the launch's own block only
  explicit AntPool(const Config& cfg)
      : Pool(cfg, AntKeys(cfg), KeySpec{"action", EPA_F64, {A::kNU}}, true) {
    // (a unit queue re-reads the action rows per unit: uploaded, not read in place -- engine.h; and with several
    // batches in flight the long Ant kernels overlap their downloads anyway: direct batches measured 5-9 % slower
    // there, profiles/r6m_async_numpy_*.jsonl)
    direct_default_ = (cfg.batch_size > 0 && cfg.batch_size < cfg.num_envs) ? 0 : 1;
    task_.use_contact_force = cfg.Get("use_contact_force", 0) != 0;
    task_.post_constraint = cfg.Get("post_constraint", 0) != 0;
    task_.exclude_worldbody = cfg.Get("exclude_worldbody_contact_forces", 0) != 0;
    task_.contact_cost_weight = cfg.Get("contact_cost_weight", 5e-4);
    task_.contact_force_min = cfg.Get("contact_force_min", -1.0);
    task_.contact_force_max = cfg.Get("contact_force_max", 1.0);
    fp64_ = (int)cfg.Get("precision", 1) == 1;
    sub_ = (int)cfg.Get("ant_sub", 1);
    {
      hipDeviceProp_t prop;
      EPA_HIP(hipGetDeviceProperties(&prop, cfg.device));
      wave_slots_ = prop.multiProcessorCount * 4;
    }
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
    if (sub_ < 1 || sub_ > task_.frame_skip) sub_ = task_.frame_skip;
    size_t n = cfg.num_envs;
    EPA_HIP(hipMalloc(&dev_.qpos, sizeof(double) * A::kNQ * n));
    EPA_HIP(hipMalloc(&dev_.qvel, sizeof(double) * A::kNV * n));
    EPA_HIP(hipMalloc(&dev_.warm, sizeof(double) * A::kNV * n));
    EPA_HIP(hipMalloc(&dev_.lag, sizeof(double) * 2 * n));
    EPA_HIP(hipMalloc(&dev_.nsaved, sizeof(double) * n));
    EPA_HIP(hipMalloc(&dev_.navail, n));
    EPA_HIP(hipMalloc(&dev_.cost, sizeof(double) * n));
    EPA_HIP(hipMemsetAsync(dev_.cost, 0, sizeof(double) * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.qpos, 0, sizeof(double) * A::kNQ * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.qvel, 0, sizeof(double) * A::kNV * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.warm, 0, sizeof(double) * A::kNV * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.lag, 0, sizeof(double) * 2 * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.nsaved, 0, sizeof(double) * n, stream_));
    EPA_HIP(hipMemsetAsync(dev_.navail, 0, n, stream_));
    trace_.Init("EPA_ANT_TRACE", (n + kAntEnvsPerBlock - 1) / kAntEnvsPerBlock, stream_);
    dev_.trace = trace_.d;
    mt_tile_default_ = 16;  // unhealthy terminations: every env resets at its own time
    InitCommon();
    EnableObsStack();  // frame_stack > 1: generic ring (envpool/mujoco/frame_stack.h:74-146)
  }
  ~AntPool() override {
    (void)hipFree(dev_.qpos);
    (void)hipFree(dev_.qvel);
    (void)hipFree(dev_.warm);
    (void)hipFree(dev_.lag);
    (void)hipFree(dev_.nsaved);
    (void)hipFree(dev_.navail);
    trace_.DumpAndFree();
    (void)hipFree(dev_.cost);
    for (auto& kv : queues_) {
      (void)hipFree(kv.second.ticket);
      (void)hipFree(kv.second.rowflag);
      (void)hipFree(kv.second.progress);
    }
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
    // launches on different streams run concurrently (async mode): a queue and its scratch each
    Queue& qu = queues_[stream_];
    if (qu.ticket == nullptr) {
      const size_t n = (size_t)cfg_.num_envs, nch = (n + kAntEnvsPerBlock - 1) / kAntEnvsPerBlock;
      EPA_HIP(hipMalloc(&qu.ticket, sizeof(unsigned)));
      EPA_HIP(hipMalloc(&qu.rowflag, n));
      EPA_HIP(hipMalloc(&qu.progress, sizeof(unsigned) * nch));
      EPA_HIP(hipMemsetAsync(qu.ticket, 0, sizeof(unsigned), stream_));
      EPA_HIP(hipMemsetAsync(qu.progress, 0, sizeof(unsigned) * nch, stream_));
    }
    AntArgs args{};
    args.dev = dev_;
    args.cm = common_;
    args.a = a;
    args.action = static_cast<const double*>(d_action);
    args.out = out;
    args.task = task_;
    args.nchunks = (k + kAntEnvsPerBlock - 1) / kAntEnvsPerBlock;
    // a reset launch has nothing to balance: one unit per chunk, rows in order
    args.sub = force_reset ? task_.frame_skip : sub_;
    args.units_per_chunk = (task_.frame_skip + args.sub - 1) / args.sub;
    args.rowflag = qu.rowflag;
    args.progress = qu.progress;
    args.epoch = qu.epoch;
    qu.epoch += (unsigned)args.units_per_chunk + 1u;
    args.max_iter = fp64_ ? 50 : 12;
    args.gtol = fp64_ ? 1e-13 : 1e-6;
    const int resident = wave_slots_ * (fp64_ ? 1 : 2);
    const int units = args.nchunks * args.units_per_chunk;
    const int blocks = units < resident ? units : resident;
    args.ticket = qu.ticket;
    args.ticket_base = qu.base;
    qu.base += (unsigned)(units + blocks);  // every wave takes one ticket past the end
    const bool wrench = task_.use_contact_force && task_.post_constraint;
#define EPA_LAUNCH_ANT(T, W) \
  hipLaunchKernelGGL((AntStepKernel<T, W>), dim3(blocks), dim3(kAntBlock), 0, stream_, args)
    if (fp64_) {
      if (wrench) EPA_LAUNCH_ANT(double, true); else EPA_LAUNCH_ANT(double, false);
    } else {
      if (wrench) EPA_LAUNCH_ANT(float, true); else EPA_LAUNCH_ANT(float, false);
    }
#undef EPA_LAUNCH_ANT
  }

 private:
  struct Queue {
    unsigned* ticket{nullptr};
    unsigned base{0};
    unsigned char* rowflag{nullptr};
    unsigned* progress{nullptr};
    unsigned epoch{1};
  };
  std::map<hipStream_t, Queue> queues_;
  int wave_slots_{1024};  // SIMDs of the device
  int sub_{1};            // "ant_sub": mj_steps per unit of the work queue
  AntDev dev_{};
  WaveTrace trace_;
  A::AntModel<double> model_;
  AntTask task_{};
  bool fp64_{true};
};

}  // namespace

}  // namespace epa
#ifdef EPA_ANT_TIMERS
// diagnostic build only: read (and clear) the stage timers
extern "C" int epa_debug_ant_timers(unsigned long long* out16, int clear) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(epa::g_ant_timers), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (clear) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(epa::g_ant_timers), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
namespace epa {

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
