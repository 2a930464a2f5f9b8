// K3' — batched step kernel of the planar legged gym robots with one env per LANE GROUP
// (mj_planar_lg.hip.h), fp64: HalfCheetah / Walker2d on 2 or 4 adjacent lanes, the one-legged Hopper on ONE.
//
// Replaces, for the whole batch in one launch, exactly what CheetahStepKernel (mujoco_gym.hip)
// replaces -- MujocoEnv::{MujocoReset,MujocoStep} (envpool/mujoco/gym/mujoco_env.h:126-148),
// HalfCheetahEnvBase::{MujocoResetModel,Reset,Step,WriteState} (gym/half_cheetah.h:105-185),
// Walker2dEnvBase::{...} (gym/walker2d.h:119-219), HopperEnvBase::{...} (gym/hopper.h:121-230) and the runtime around them
// (async_envpool.h:118-132, env.h:184-256) -- on the same device state (SoA float64 qpos / qvel /
// qacc_warmstart [9][N]), so a pool can switch between the two layouts from one launch to the next.
//
// Lane c of a group loads the torso state (replicated) and its own leg's; the group's first lane
// does the env-level work (mt19937 reset draws, reward, bookkeeping, torso outputs), every leg's
// first lane writes that leg's state and observation entries.
#include "mj_planar_lg.hip.h"
#include "mujoco_planar_common.h"

namespace epa {
namespace {

using mj::CheetahModel;
using mj::kNU;
using mj::kNV;
using planar::CheetahDev;
using planar::CheetahTask;
using planar::PlanarModel;
namespace plg = mj::plg;

constexpr int kBlock = 64;

// (an LDS pointer typed as such: through a generic pointer the optimiser-opaque Refresh() below turns
// every table read into a flat_load, which takes the long way round to the LDS)
using LdsConstDouble = const __attribute__((address_space(3))) double;
template <int KL>
struct DevCx {
  LdsConstDouble* tab;  // the wave's LDS copy of the table at this lane's column: tab[id * KL]
  double* lds;        // the wave's LDS block at this lane: lds[slot * 64]
  __device__ __forceinline__ double C(int id) const { return tab[id * KL]; }
  __device__ __forceinline__ double& Lds(int slot) { return lds[slot * kBlock]; }
  // per-lane addressing (a lane visits its own touching slots): LDS slot `s * mul + add`, table entry `base + idx`
  __device__ __forceinline__ double LdsL(unsigned s, int mul, int add) const { return lds[(s * mul + add) * kBlock]; }
  __device__ __forceinline__ void LdsLStore(unsigned s, int mul, int add, double x, bool on) {
    if (on) lds[(s * mul + add) * kBlock] = x;
  }
  __device__ __forceinline__ double CL(int base, unsigned idx) const { return tab[(base + idx) * KL]; }
  // the table pointer becomes opaque to the optimiser: what is read through it afterwards cannot be
  // hoisted above this point (out of the mj_step loop, into registers that then spill).  The
  // constants are read from LDS where they are used (~90 ds_read_b64 per forward pass, ~100 cycles
  // each, off the VALU); from global memory every one of them was an L2 round trip.
  __device__ __forceinline__ void Refresh() { asm volatile("" : "+v"(tab)); }
#ifdef EPA_LG_TIMERS  // diagnostic build only (mj_planar_lg.hip.h: EPA_LG_TICK)
  long long t_last{0};
  long long acc[6]{0, 0, 0, 0, 0, 0};
  unsigned cnt[3]{0, 0, 0};
  template <int K>
  __device__ __forceinline__ void TickEnd() {
    const long long now = clock64();
    acc[K] += now - t_last;
    t_last = now;
  }
  template <int K>
  __device__ __forceinline__ void Count() { ++cnt[K]; }
#endif
};
#ifdef EPA_LG_TIMERS
// [0..4] cycles per category, [5..7] trip counters, [8] chunks, [9] cycles of whole chunks, [10] category 5
// (the integration of an mj_step: what follows the forward pass)
__device__ unsigned long long g_lg_timers[16];
#endif
#ifdef EPA_LG_SCHED_TRACE
// diagnostic build only (tools/lg_sched_trace.py): who ran which chunk when.  [0] = records filed, then
// {wave (= block), chunk, start, end} per chunk in 100 MHz wall-clock ticks (s_memrealtime)
constexpr int kSchedCap = 1 << 15;
__device__ unsigned long long g_lg_sched[1 + 4 * kSchedCap];
// [0] ticks between the top of a chunk and the start of its stepping branch (= the RESET branch of the lanes whose
// env resets, which the wave executes first), [1] chunks counted, [2] chunks in which some env reset
__device__ unsigned long long g_lg_reset[4];
#endif

// Everything the kernel is given, as ONE by-value argument: it sits at offset 0 of the kernarg
// segment and is read from there (scalar loads) where it is needed, see KernArgs().
struct LgArgs {
  CheetahDev dev;
  CommonDev cm;
  StepArgs a;
  const double* action;
  OutPtrs out;
  CheetahTask task;
  plg::SolverCfgLg<double> scfg;
  const double* tab;
  unsigned* ticket;
  unsigned ticket_base;
  int nchunks;
  int per;  // envs per chunk: 64 / KL, or fewer (partly filled waves) while that still gives every SIMD a chunk
  // longest-chunk-first dispatch (see PlanarLgStepKernel): three generations of
  // {cnt[kLptBuckets], sum, n, list[kLptBuckets][lpt_cap]}; this launch reads generation lpt_gen % 3
  // (if lpt_use), fills (lpt_gen + 1) % 3 and clears (lpt_gen + 2) % 3.  nullptr: off.
  unsigned* lpt;
  int lpt_cap;
  int lpt_gen;
  int lpt_use;
};
constexpr int kLptBuckets = 16;
constexpr int kLptHead = kLptBuckets + 4;  // cnt[16], sum (2 words), n, pad
__host__ __device__ constexpr size_t LptGenWords(int cap) { return (size_t)kLptHead + (size_t)kLptBuckets * cap; }
using LgArgsK = const __attribute__((address_space(4))) LgArgs;
// The kernel's arguments, re-read from the kernarg segment through a pointer that is opaque to the
// optimiser.  With the arguments as ordinary by-value parameters the persistent loop below makes
// every one of them (48 output pointers, 20 task scalars, ...) a loop invariant that is loaded once
// and then kept in SGPRs across the whole solver: 90 SGPR spills.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ LgArgsK* KernArgs() {
  LgArgsK* p = (LgArgsK*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
#endif

// One chunk of 64 / KL envs (rows chunk * 64 / KL ...) on this wave.
template <int KL, int kModel>
__device__ __forceinline__ void StepChunk(int chunk, const double* tab_lds, double* lds_buf) {
#if defined(__HIP_DEVICE_COMPILE__)  // (address-space-4 reads: nothing for hipcc's host pass to compile)
  LgArgsK* ap = KernArgs();
  const CheetahDev dev = ap->dev;
  const CommonDev cm = ap->cm;
  const StepArgs a = ap->a;
  const double* __restrict__ action = ap->action;
  const plg::SolverCfgLg<double> scfg = ap->scfg;
#define task (ap->task)
#define out (ap->out)
  using G = plg::Grp<KL>;
  constexpr CheetahModel<double> m = PlanarModel<double, kModel>();
  constexpr bool kWalker = kModel != mj::kPlanarCheetah;  // Walker2d or Hopper: RK4, mirrored hinges
  constexpr bool kHopper = kModel == mj::kPlanarHopper;   // KL = 1: dofs 0..5 / motors 0..2 of the 9-dof tree
  static_assert(kHopper == (KL == 1), "a group of one lane is the one-legged model, and only it");
  constexpr int kNVr = kHopper ? 6 : kNV, kNUr = kHopper ? 3 : kNU;
  const int lane = threadIdx.x;
  const int n = cm.n;
  const int c = lane & (KL - 1);  // lane coordinate in the group
  const int leg = G::Leg(c);
  const bool first = c == 0;                  // env-level work
  const bool leg_first = G::Par(c) == 0;      // leg-level outputs
  const int row = chunk * ap->per + (lane / KL);
  if (lane / KL >= ap->per || row >= a.k) return;
  const int e = a.ids ? a.ids[row] - a.id_offset : row;
  bool done = cm.done[e] != 0;
  int cur = cm.cur_step[e];
  const bool reset = a.force_reset || done;  // async_envpool.h:127
  const int nobs = 2 * kNVr - task.obs_skip;
  double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * nobs;
  float reward = 0.0f;
  double xv = 0.0, ctrl_cost = 0.0, xpos = 0.0, qz = 0.0, qang = 0.0;
#ifdef EPA_LG_SCHED_TRACE
  const unsigned long long tr0 = wall_clock64();
  const bool any_reset = __builtin_amdgcn_ballot_w64(reset) != 0;
#endif
  if (reset) {
    // MujocoReset: mj_resetData + MujocoResetModel (half_cheetah.h:105-117, walker2d.h:119-126);
    // the env's RNG stream is sequential: the group's first lane draws everything and writes the
    // whole state and observation row (the warm start is cleared, see CheetahStepKernel)
    cur = 0;
    done = false;
    if (first) {
      Mt19937 g(cm, e);
      double saved = dev.nsaved[e];
      int avail = dev.navail[e];
      double qpos[kNV], qvel[kNV];
      for (int i = kNVr; i < kNV; ++i) qpos[i] = qvel[i] = 0.0;  // the dofs the Hopper does not have
      for (int i = 0; i < kNVr; ++i) {
        const double q0 = (kWalker && i == 1) ? 1.25 : 0.0;  // rootz ref (walker2d_envpool.xml:36, hopper :39)
        qpos[i] = q0 + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
      }
      for (int i = 0; i < kNVr; ++i) {
        if constexpr (kWalker) {
          qvel[i] = 0.0 + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
        } else {
          qvel[i] = 0.0 + g.Normal(0.0, task.reset_noise_scale, &saved, &avail);
        }
      }
      g.Commit();
      dev.nsaved[e] = saved;
      dev.navail[e] = (unsigned char)avail;
      double* o = obs;
      for (int i = 0; i < kNV; ++i) {
        dev.qpos[(size_t)i * n + e] = qpos[i];
        dev.qvel[(size_t)i * n + e] = qvel[i];
        dev.warm[(size_t)i * n + e] = 0.0;
        if (i >= task.obs_skip && i < kNVr) *(o++) = qpos[i];
      }
      for (int i = 0; i < kNVr; ++i) {
        double x = qvel[i];
        if constexpr (kWalker) {  // walker2d.h:210-215
          x = x < task.velocity_max ? x : task.velocity_max;
          x = x > task.velocity_min ? x : task.velocity_min;
        }
        *(o++) = x;
      }
    }
  } else {
#ifdef EPA_LG_SCHED_TRACE
    const unsigned long long ts0 = wall_clock64();
    if (lane == __builtin_amdgcn_readfirstlane(lane)) {
      atomicAdd(&g_lg_reset[0], ts0 - tr0);
      atomicAdd(&g_lg_reset[1], 1ull);
      if (any_reset) atomicAdd(&g_lg_reset[2], 1ull);
    }
#endif
    ++cur;
    double q[plg::kLV], v[plg::kLV], w[plg::kLV], ctrl[3];
    double x_before = 0.0;
    mj::static_for<0, plg::kLV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int gi = i < 3 ? i : 3 * leg + i;  // global dof
      const double sg = (kWalker && i >= 3) ? -1.0 : 1.0;  // hinge about -y: q' = -q (PlanarDofSign)
      const double qq = dev.qpos[(size_t)gi * n + e];
      if constexpr (i == 0) x_before = qq;
      q[i] = sg * qq;
      v[i] = sg * dev.qvel[(size_t)gi * n + e];
      w[i] = sg * dev.warm[(size_t)gi * n + e];
    });
    q[0] = 0.0;  // the root x is carried as a local offset per env-step
    const double* act = action + (size_t)row * kNUr;
    mj::static_for<0, 3>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const double ai = act[3 * leg + i];
      // ctrllimited motors: MuJoCo clamps ctrl to ctrlrange [-1, 1]
      ctrl[i] = ai < -1.0 ? -1.0 : (ai > 1.0 ? 1.0 : ai);
    });
    if (first) {  // half_cheetah.h:143-146: summed in the reference's order
      for (int i = 0; i < kNUr; ++i) ctrl_cost += task.ctrl_cost_weight * act[i] * act[i];
    }
    DevCx<KL> cx{(LdsConstDouble*)tab_lds + c, lds_buf + lane};
#ifdef EPA_LG_TIMERS
    cx.t_last = clock64();
    const long long t_chunk0 = cx.t_last;
#endif
    double iters = 0.0;
    for (int s = 0; s < task.frame_skip; ++s) {  // mujoco_env.h:142-144
      if constexpr (kWalker) {
        iters += plg::StepRK4<KL>(m, scfg, cx, q, v, w, ctrl);
      } else {
        iters += plg::StepEuler<KL>(m, scfg, cx, q, v, w, ctrl);
      }
    }
#ifdef EPA_LG_TIMERS
    cx.template TickEnd<0>();
    if (__builtin_amdgcn_readfirstlane(lane) == lane) {  // the wave's first active lane
      for (int i = 0; i < 5; ++i) atomicAdd(&g_lg_timers[i], (unsigned long long)cx.acc[i]);
      atomicAdd(&g_lg_timers[10], (unsigned long long)cx.acc[5]);
      for (int i = 0; i < 3; ++i) atomicAdd(&g_lg_timers[5 + i], (unsigned long long)cx.cnt[i]);
      atomicAdd(&g_lg_timers[8], 1ull);
      atomicAdd(&g_lg_timers[9], (unsigned long long)(clock64() - t_chunk0));
    }
#endif
#ifdef EPA_LG_SCHED_TRACE
    if (lane == __builtin_amdgcn_readfirstlane(lane)) atomicAdd(&g_lg_reset[3], wall_clock64() - ts0);  // the mj_steps
#endif
    const double x_after = x_before + q[0];
    xv = (x_after - x_before) / task.dt;  // half_cheetah.h:148-149
    xpos = x_after;
    qz = q[1];
    qang = q[2];
    const int no = kNVr - task.obs_skip;  // position entries of the observation
    bool state_ok = true;  // Hopper: every qpos[2:] and qvel inside (healthy_state_min, healthy_state_max)
    mj::static_for<0, plg::kLV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const bool mine = i < 3 ? first : leg_first;
      const int gi = i < 3 ? i : 3 * leg + i;
      const double sg = (kWalker && i >= 3) ? -1.0 : 1.0;
      const double qq = i == 0 ? x_after : sg * q[i];
      const double vv = sg * v[i];
      if constexpr (kHopper) {  // hopper.h:192-203 (the lane holds the whole state)
        if (i >= 2 && (qq <= task.healthy_state_min || qq >= task.healthy_state_max)) state_ok = false;
        if (vv <= task.healthy_state_min || vv >= task.healthy_state_max) state_ok = false;
      }
      if (mine) {
        dev.qpos[(size_t)gi * n + e] = qq;
        dev.qvel[(size_t)gi * n + e] = vv;
        dev.warm[(size_t)gi * n + e] = sg * w[i];
        if (gi >= task.obs_skip) obs[gi - task.obs_skip] = qq;
        double x = vv;
        if constexpr (kWalker) {  // walker2d.h:210-215
          x = x < task.velocity_max ? x : task.velocity_max;
          x = x > task.velocity_min ? x : task.velocity_min;
        }
        obs[no + gi] = x;
      }
    });
    if (first) dev.iters[e] = (int)iters;
    if constexpr (kWalker) {  // walker2d.h:162-177,181-190 / hopper.h:170-203
      bool healthy;
      if constexpr (kHopper) {
        healthy = !(qang <= task.healthy_angle_min || qang >= task.healthy_angle_max || qz <= task.healthy_z_min) &&
                  state_ok;
      } else {
        healthy = !(qz < task.healthy_z_min || qz > task.healthy_z_max ||
                    qang < task.healthy_angle_min || qang > task.healthy_angle_max);
      }
      bool give = healthy;
      if (task.legacy_healthy_reward) give = task.terminate_when_unhealthy || healthy;
      const double healthy_reward = give ? task.healthy_reward : 0.0;
      reward = static_cast<float>(xv * task.forward_reward_weight + healthy_reward - ctrl_cost);
      done = (task.terminate_when_unhealthy ? !healthy : false) || cur >= a.max_episode_steps;
    } else {
      reward = static_cast<float>(xv * task.forward_reward_weight - ctrl_cost);
      done = cur >= a.max_episode_steps;  // ++elapsed_step_ >= max_episode_steps_
    }
  }
  if (!first) return;
  cm.done[e] = done ? 1 : 0;
  cm.cur_step[e] = cur;
  if constexpr (kWalker) {  // walker2d.h:218-219
    ((double*)out.p[kKeyEnv0 + 1])[row] = reset ? 0.0 : xpos;
    ((double*)out.p[kKeyEnv0 + 2])[row] = xv;
  } else {  // half_cheetah.h:158-185
    ((double*)out.p[kKeyEnv0 + 1])[row] = xv * task.forward_reward_weight;
    ((double*)out.p[kKeyEnv0 + 2])[row] = -ctrl_cost;
    ((double*)out.p[kKeyEnv0 + 3])[row] = reset ? 0.0 : xpos;
    ((double*)out.p[kKeyEnv0 + 4])[row] = xv;
  }
  const OutPtrs outv = out;
  WriteCommon(outv, row, e + a.id_offset, cur, done, reward, a.max_episode_steps);
#undef task
#undef out
#endif
}

// PERSISTENT waves over a dynamic queue of chunks.  A wave runs as long as its slowest env and
// as its busiest lane has touching slots, so chunks differ a lot in duration (mean wave 80 us,
// slowest 143 us at N = 32768 in round 3: profiles/archive/r3e_*): with one chunk per wave the launch lasts as long as
// its slowest wave.  Here the grid is the number of waves that are resident at once, wave i starts
// with chunk i and then takes the next chunk off an atomic ticket counter until none is left --
// whoever finishes early does the extra work.  `ticket_base`: the counter's value before this
// launch; every wave makes exactly one failing fetch, so a launch advances the counter by exactly
// its number of chunks and the host never has to reset it.
// LONGEST CHUNK FIRST.  With about two chunks per resident wave (N = 65536: 2048 chunks, 1024 waves)
// the order of the queue decides the makespan: a slow chunk (many touching end spheres) taken last
// keeps one SIMD busy while 1023 idle.  A chunk's duration is strongly correlated from one env-step
// to the next (contact states persist), so every chunk times itself (s_memtime) and files itself
// into one of 16 duration buckets (relative to the previous launch's mean) for the NEXT launch,
// which serves the buckets slowest first.  Only the order of dispatch changes, never which envs
// share a wave: results are bit-identical with and without it.  Used for whole-pool batches
// (ids == nullptr) that follow a launch of the same shape; everything else runs in index order.
// W: waves per SIMD the register allocation aims at (512 registers per SIMD lane: 1 -> 512, 2 -> 256)
template <int KL, int kModel, int W>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(W, W))) void PlanarLgStepKernel(LgArgs args) {
  __shared__ double lds_buf[plg::LdsSlots<KL>() * kBlock];
  __shared__ double tab_lds[plg::Tab<KL>::kSize];
  {
    const double* __restrict__ tab = args.tab;
    for (int i = threadIdx.x; i < plg::Tab<KL>::kSize; i += kBlock) tab_lds[i] = tab[i];
  }
  __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
  if (args.lpt != nullptr && blockIdx.x == 0 && threadIdx.x < kLptHead) {  // generation + 2: cleared for the next launch to fill
    args.lpt[LptGenWords(args.lpt_cap) * ((args.lpt_gen + 2) % 3) + threadIdx.x] = 0u;
  }
  int tk = blockIdx.x;  // position in the queue: the first one is the wave's index, then tickets
  for (;;) {
    LgArgsK* ap = KernArgs();
    int chunk = tk;
    unsigned mean = 0;
    if (ap->lpt != nullptr) {
      const unsigned* cur = ap->lpt + LptGenWords(ap->lpt_cap) * (ap->lpt_gen % 3);
      if (ap->lpt_use) {
        int rem = tk;
        for (int b = kLptBuckets - 1; b >= 0; --b) {  // slowest bucket first
          const int c = (int)cur[b];
          if (rem < c) {
            chunk = (int)cur[kLptHead + (size_t)b * ap->lpt_cap + rem];
            break;
          }
          rem -= c;
        }
        const unsigned long long sum = ((unsigned long long)cur[kLptBuckets + 1] << 32) | cur[kLptBuckets];
        mean = cur[kLptBuckets + 2] ? (unsigned)(sum / cur[kLptBuckets + 2]) : 0u;
      }
    }
    chunk = __builtin_amdgcn_readfirstlane(chunk);
    const long long t0 = clock64();
#ifdef EPA_LG_SCHED_TRACE
    const unsigned long long w0 = wall_clock64();
#endif
    StepChunk<KL, kModel>(chunk, tab_lds, lds_buf);
#ifdef EPA_LG_SCHED_TRACE
    if (threadIdx.x == 0) {
      const unsigned long long w1 = wall_clock64();
      const unsigned long long i = atomicAdd(&g_lg_sched[0], 1ull);
      if (i < (unsigned long long)kSchedCap) {
        g_lg_sched[1 + 4 * i] = blockIdx.x;
        g_lg_sched[2 + 4 * i] = (unsigned long long)chunk;
        g_lg_sched[3 + 4 * i] = w0;
        g_lg_sched[4 + 4 * i] = w1;
      }
    }
#endif
    ap = KernArgs();
    unsigned t = 0;
    if (threadIdx.x == 0) {
      if (ap->lpt != nullptr) {  // file this chunk for the next launch
        unsigned* nxt = ap->lpt + LptGenWords(ap->lpt_cap) * ((ap->lpt_gen + 1) % 3);
        const unsigned d = (unsigned)(clock64() - t0);
        int b = mean ? (int)(((unsigned long long)d * 8u) / mean) : (int)(d >> 14);  // no mean yet: ~16k-cycle steps
        b = b > kLptBuckets - 1 ? kLptBuckets - 1 : b;
        const unsigned pos = atomicAdd(&nxt[b], 1u);
        nxt[kLptHead + (size_t)b * ap->lpt_cap + pos] = (unsigned)chunk;
        atomicAdd(reinterpret_cast<unsigned long long*>(&nxt[kLptBuckets]), (unsigned long long)d);
        atomicAdd(&nxt[kLptBuckets + 2], 1u);
      }
      t = atomicAdd(ap->ticket, 1u);
    }
    t = __builtin_amdgcn_readfirstlane(t);
    tk = (int)gridDim.x + (int)(t - ap->ticket_base);
    if (tk >= ap->nchunks) break;
  }
#endif
}

// returns whether the launch took part in the longest-first protocol (filed its chunks for generation gen + 1 and
// cleared generation gen + 2): the host advances `gen` only then
template <int KL, int W>
bool LaunchKl(hipStream_t st, int model, int wave_slots, bool spread, const CheetahDev& dev, const CommonDev& cm,
              const StepArgs& a, const double* action, const OutPtrs& out, const CheetahTask& task,
              const double* tab, unsigned* ticket, unsigned* ticket_base, const planar::LgOrder& lo) {
  // waves resident at once: W per SIMD by registers (LDS allows no more than one at KL = 2)
  const int resident = wave_slots * ((KL <= 2 || W == 1) ? 1 : 2);
  // A wave runs as long as its slowest env (and its busiest lane): while
  // there are fewer full chunks than resident waves, smaller chunks (partly filled waves) on more
  // SIMDs are faster (N = 8192 at 4 lanes per env: 512 waves of 16 envs 95 us, 1024 waves of 8 envs
  // 90 us; N = 12288 as 12 per wave: 96 -> 93 us; profiles/archive/r3l_lane_group_spread_ab.txt).  Half, three
  // quarters or all of a wave: quarter-filled waves measured SLOWER (N = 2048: 87 -> 112 us).
  const int full = kBlock / KL, q = full / 4;
  int per = ((a.k + resident - 1) / resident + q - 1) / q * q;
  per = per < full / 2 ? full / 2 : (per > full ? full : per);
  if (!spread) per = full;
  const int nchunks = (a.k + per - 1) / per;
  const int blocks = nchunks < resident ? nchunks : resident;
  const unsigned base = *ticket_base;
  *ticket_base = base + (unsigned)nchunks;  // see PlanarLgStepKernel
  // the order of dispatch only matters when chunks queue for waves: with one chunk per resident wave or fewer the
  // self-timing and the bucket lists are pure overhead (N = 8192: +1.5 % without, profiles/r5j_*)
  unsigned* const lpt = nchunks > resident ? lo.d : nullptr;
  const LgArgs args{dev, cm, a, action, out, task, plg::SolverCfgLg<double>{50, 1e-13}, tab, ticket, base, nchunks,
                    per, lpt, lo.cap, lo.gen, lo.use};
  if constexpr (KL == 1) {  // a group of one lane: the one-legged model
    if (model != mj::kPlanarHopper) throw std::invalid_argument("lane group of 1: the Hopper only");
    hipLaunchKernelGGL((PlanarLgStepKernel<1, mj::kPlanarHopper, W>), dim3(blocks), dim3(kBlock), 0, st, args);
  } else {
    switch (model) {
      case mj::kPlanarCheetah:
        hipLaunchKernelGGL((PlanarLgStepKernel<KL, mj::kPlanarCheetah, W>), dim3(blocks), dim3(kBlock), 0, st, args);
        break;
      case mj::kPlanarWalker:
        hipLaunchKernelGGL((PlanarLgStepKernel<KL, mj::kPlanarWalker, W>), dim3(blocks), dim3(kBlock), 0, st, args);
        break;
      case mj::kPlanarWalkerV5:
        hipLaunchKernelGGL((PlanarLgStepKernel<KL, mj::kPlanarWalkerV5, W>), dim3(blocks), dim3(kBlock), 0, st, args);
        break;
      default:
        throw std::invalid_argument("lane groups of 2 / 4: HalfCheetah and Walker2d");
    }
  }
  return lpt != nullptr;
}

}  // namespace

bool PlanarLgLaunch(hipStream_t st, int kl, int waves, int model, int wave_slots, bool spread,
                    const planar::CheetahDev& dev,
                    const CommonDev& cm, const StepArgs& a, const double* action, const OutPtrs& out,
                    const planar::CheetahTask& task, const double* tab, unsigned* ticket, unsigned* ticket_base,
                    const planar::LgOrder& lo) {
#define EPA_LG(KL, W) \
  LaunchKl<KL, W>(st, model, wave_slots, spread, dev, cm, a, action, out, task, tab, ticket, ticket_base, lo)
  if (kl == 1) return EPA_LG(1, 1);
  // LDS (38 KB per wave) allows one wave per SIMD whatever the register budget: no <2, *, 2> build
  if (kl == 2) return EPA_LG(2, 1);
  if (waves == 1) return EPA_LG(4, 1);
  return EPA_LG(4, 2);
#undef EPA_LG
}

#ifdef EPA_LG_TIMERS
}  // namespace epa
// diagnostic build only: read (and clear) the stage timers
extern "C" int epa_debug_lg_timers(unsigned long long* out16, int clear) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(epa::g_lg_timers), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (clear) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(epa::g_lg_timers), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
namespace epa {
#endif

#ifdef EPA_LG_SCHED_TRACE
}  // namespace epa
// diagnostic build only: copy out (and clear) the chunk schedule records; returns the number filed
extern "C" long long epa_debug_lg_sched(unsigned long long* out, int max_records, int clear) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  unsigned long long n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(epa::g_lg_sched), sizeof(n)) != hipSuccess) return -1;
  unsigned long long k = n < (unsigned long long)epa::kSchedCap ? n : (unsigned long long)epa::kSchedCap;
  if (k > (unsigned long long)max_records) k = max_records;
  if (out != nullptr && k > 0 &&
      hipMemcpyFromSymbol(out, HIP_SYMBOL(epa::g_lg_sched), sizeof(unsigned long long) * 4 * k, sizeof(unsigned long long)) != hipSuccess) {
    return -1;
  }
  if (out != nullptr && max_records >= 4) {  // the reset-branch counters ride in front: out[-4 .. -1] is not possible,
    // so they are appended after the records the caller asked for
    unsigned long long r[4];
    if (hipMemcpyFromSymbol(r, HIP_SYMBOL(epa::g_lg_reset), sizeof(r)) != hipSuccess) return -1;
    for (int i = 0; i < 4; ++i) out[4 * (size_t)max_records + i] = r[i];
  }
  if (clear) {
    const unsigned long long z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(epa::g_lg_sched), &z, sizeof(z)) != hipSuccess) return -1;
    const unsigned long long z4[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(epa::g_lg_reset), z4, sizeof(z4)) != hipSuccess) return -1;
  }
  return (long long)n;
}
namespace epa {
#endif

size_t PlanarLgOrderBytes(int cap) { return sizeof(unsigned) * 3 * LptGenWords(cap); }

int PlanarLgBuildTable(int kl, int model, double* tab) {
  const CheetahModel<double> m = model == mj::kPlanarCheetah  ? kCheetahModelConst
                                 : model == mj::kPlanarWalker ? kWalkerModelConst
                                 : model == mj::kPlanarHopper ? kHopperModelConst
                                                              : kWalkerV5ModelConst;
  static_assert(plg::Tab<1>::kSize <= kPlanarLgTabMax && plg::Tab<2>::kSize <= kPlanarLgTabMax &&
                    plg::Tab<4>::kSize <= kPlanarLgTabMax,
                "table size");
  if (kl == 1) {
    plg::BuildTable<1>(m, tab);
    return plg::Tab<1>::kSize;
  }
  if (kl == 2) {
    plg::BuildTable<2>(m, tab);
    return plg::Tab<2>::kSize;
  }
  plg::BuildTable<4>(m, tab);
  return plg::Tab<4>::kSize;
}

}  // namespace epa
