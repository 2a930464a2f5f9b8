// K3d — gym-MuJoCo Humanoid / HumanoidStandup batched step kernel, ONE ENV PER LANE QUAD
// (mj_hum4.hip.h: trunk replicated, one limb per lane, everything of a forward pass on chip
// except the constraint rows).
//
// Replaces, for the whole batch in one launch, the same reference code as mujoco_humanoid.hip:
//   MujocoEnv::{MujocoReset,MujocoStep}   envpool/mujoco/gym/mujoco_env.h:126-148
//   HumanoidEnvBase::{MujocoResetModel,Reset,Step,IsHealthy,GetMassCenter,WriteState}
//                                         envpool/mujoco/gym/humanoid.h:129-268
//   HumanoidStandupEnvBase::{...}         envpool/mujoco/gym/humanoid_standup.h:119-240
//
// A 64-thread block is one wavefront = 16 envs; lane l of a quad owns limb l (right leg, left
// leg, right arm, left arm).  LDS per wave: the per-limb constant table (3.5 KB), the envs'
// geoms + trunk cdof (20 KB), sharing their block with the PGS stage's per-row scalars, and the
// trunk block of the L'DL factor (7 KB).  HBM per wave: the constraint rows (y = L^-T J', 9 slots per lane and row:
// the lane's four limb entries, its share of the nine trunk entries and of the row's five
// scalars -- every lane reads back only what it wrote itself, the rest of the quad's values
// arrive by DPP broadcast) and the compact contact records.  What persists between steps is
// the same per-env SoA as in mujoco_humanoid.hip (qpos, qvel, warm start, lagged mass centre).
#include "mujoco_humanoid_common.h"
#include "mj_hum4.hip.h"
#include "build/mj_humanoid_consts.inc"  // generated: kHumanoidModelConst, kHumanoidStandupModelConst

namespace epa {
namespace {

namespace H = mj::hum4;
namespace T = mj::tree;

// rows of an env on the register-resident dual matrix (A/B builds: tools/build_alt_hum4.sh)
#ifndef EPA_HUM_REGROWS
#define EPA_HUM_REGROWS 12
#endif
#ifndef EPA_STANDUP_REGROWS
#define EPA_STANDUP_REGROWS 16
#endif
// the constraint stage (rows + solve) behind a call: Hum4::ConstraintStage
#ifndef EPA_HUM_STAGECALL
#define EPA_HUM_STAGECALL 0
#endif
#ifndef EPA_STANDUP_STAGECALL
#define EPA_STANDUP_STAGECALL 1
#endif
#ifndef EPA_HUM_ROWCACHE
#define EPA_HUM_ROWCACHE 0
#endif
#ifndef EPA_STANDUP_ROWCACHE
#define EPA_STANDUP_ROWCACHE 2
#endif
#ifndef EPA_HUM_CACHEROWS
#define EPA_HUM_CACHEROWS 8
#endif
#ifndef EPA_STANDUP_CACHEROWS
#define EPA_STANDUP_CACHEROWS 16
#endif
struct HumanoidMP {
  static constexpr T::TreeModel kM = kHumanoidModelConst;
  static constexpr int kRegRows = EPA_HUM_REGROWS, kCacheRows = EPA_HUM_CACHEROWS;
  static constexpr bool kStageCall = EPA_HUM_STAGECALL != 0;
  static constexpr int kRowCache = EPA_HUM_ROWCACHE;
  static constexpr bool kLazyNact = true;
};
struct StandupMP {
  static constexpr T::TreeModel kM = kHumanoidStandupModelConst;
  static constexpr int kRegRows = EPA_STANDUP_REGROWS, kCacheRows = EPA_STANDUP_CACHEROWS;
  static constexpr bool kStageCall = EPA_STANDUP_STAGECALL != 0;
  static constexpr int kRowCache = EPA_STANDUP_ROWCACHE;
  static constexpr bool kLazyNact = false;
};

constexpr int kBlock = 64, kEnvsPerBlock = 16;
// rows: 17 limits + 4 x 29 floor contacts + 109 pairs; contacts: 29 + 109
constexpr int kMaxRows = 17 + 4 * 29 + 128, kMaxCon = 29 + 128;
constexpr int kRowSlots = 9, kRecSlots = 2;
constexpr int kRkSlots = 10 + 4 * H::kNS;  // RK4 bookkeeping of the running mj_step (trunk part spread over the quad)
constexpr int kWsRk = kMaxRows * kRowSlots + kMaxCon * kRecSlots;
// overflow rows of the hybrid PGS (Hum4::SolvePgsT<true>): their dual-matrix entries against the register rows
constexpr int kMaxOv = 16, kOvSlots = 8;
constexpr int kWsOv = kWsRk + kRkSlots;
constexpr int kWsSlots = kWsOv + kMaxOv * kOvSlots;  // per lane
// LDS of a wave, in doubles.  Env-level slots are [slot][quad].  The geoms + trunk cdof (Position
// .. MakeRows) and the dual problem of the PGS solve (SolvePgsA, after MakeRows) share region U.
constexpr int kGeoSlots = 102, kTcdSlots = 54, kLttSlots = 45 + 9;
constexpr int kUSlots = kGeoSlots + kTcdSlots;  // = Hum4::kFSlots: the PGS stage's shared block
constexpr int kLdsTab = 0, kLdsU = H::kNLC * 4, kLdsGeo = kLdsU, kLdsTcd = kLdsU + kGeoSlots * 16, kLdsSh = kLdsU;
constexpr int kLdsLtt = kLdsU + kUSlots * 16;
constexpr int kLdsStt = kLdsLtt + kLttSlots * 16;  // the trunk part of the env's state (Hum4::kSttSlots)
constexpr int kLdsElems = kLdsStt + 31 * 16;  // 34 KB (+ 3.6 KB of candidate tables)

template <int K>
__device__ __forceinline__ double Bcast(double x) {  // lane K of the quad
#if defined(__HIP_DEVICE_COMPILE__)
  return mj::DppMov<K * 0x55>(x);
#else
  return x;
#endif
}

// (the context crosses a call boundary by value -- Hum4::ConstraintStage -- so its LDS pointer carries the address
// space in its type: a plain double* would turn every LDS access behind the call into a flat load)
using LdsDouble = __attribute__((address_space(3))) double;
using GlobalDouble = __attribute__((address_space(1))) double;

template <class MP>
struct DevCtx {
  using V = double;
  using Tabs = typename H::Hum4<MP, DevCtx<MP>>::Tabs;
  const Tabs* tabs;  // the wave's LDS copy of the candidate tables
  __device__ const Tabs& T() const { return *tabs; }
  LdsDouble* lds;
  GlobalDouble* ws;  // the wave's block, [slot][64]
  int lane, l, quad;

  // the lane coordinates become opaque to the optimiser: what is read through them afterwards cannot
  // be hoisted above this point
  __device__ void Refresh() {
    asm volatile("" : "+v"(l), "+v"(quad), "+v"(lane));
    l &= 3;
    quad &= 15;
    lane &= 63;
  }
  __device__ double LC(int idx) const { return lds[kLdsTab + idx * 4 + l]; }
  // geoms 1..17, 6 slots each, [slot][quad]
  __device__ void GeoPut(int slot, double v) { lds[kLdsGeo + (slot - 6) * 16 + quad] = v; }
  __device__ double GeoGet(int slot) const { return lds[kLdsGeo + (slot - 6) * 16 + quad]; }
  __device__ int LimbGeom(int which) const { return (l < 2 ? 6 + 3 * l : 12 + 3 * (l - 2)) + which; }
  __device__ void GeoPutLimb(int which, H::Vec3<double> pos, H::Vec3<double> axis) {
    const int s = 6 * LimbGeom(which);
    GeoPut(s, pos.x);
    GeoPut(s + 1, pos.y);
    GeoPut(s + 2, pos.z);
    GeoPut(s + 3, axis.x);
    GeoPut(s + 4, axis.y);
    GeoPut(s + 5, axis.z);
  }
  __device__ void GeoGetLimb(int which, H::Vec3<double>* pos, H::Vec3<double>* axis) const {
    const int s = 6 * LimbGeom(which);
    *pos = {GeoGet(s), GeoGet(s + 1), GeoGet(s + 2)};
    *axis = {GeoGet(s + 3), GeoGet(s + 4), GeoGet(s + 5)};
  }
  __device__ void SttPut(int i, double v) { lds[kLdsStt + i * 16 + quad] = v; }
  __device__ double SttGet(int i) const { return lds[kLdsStt + i * 16 + quad]; }
  __device__ void TcdPut(int i, double v) { lds[kLdsTcd + i * 16 + quad] = v; }
  __device__ double TcdGet(int i) const { return lds[kLdsTcd + i * 16 + quad]; }
  __device__ void LttPut(int i, double v) { lds[kLdsLtt + i * 16 + quad] = v; }
  __device__ double LttGet(int i) const { return lds[kLdsLtt + i * 16 + quad]; }
  __device__ void DtPut(int i, double v) { lds[kLdsLtt + (45 + i) * 16 + quad] = v; }
  __device__ double DtGet(int i) const { return lds[kLdsLtt + (45 + i) * 16 + quad]; }
  // the env's shared block [slot][quad] (PGS on the dual matrix)
  __device__ void ShPut(int slot, double v) { lds[kLdsSh + slot * 16 + quad] = v; }
  __device__ double ShGet(int slot) const { return lds[kLdsSh + slot * 16 + quad]; }
  __device__ GlobalDouble& Ws(int slot) const { return ws[(size_t)slot * 64 + lane]; }
  // row r: slots 0..6 the lane's share of y (Hum4::kND); slot 7 the row's scalars 1..4 (A_rr + R_r,
  // R_r, b_r, 1 / (A_rr + R_r)), lane l of the quad holding number 1 + l; slot 8 its force f (every
  // lane a copy).  A lane only ever reads back what it wrote itself; the other lanes' numbers
  // arrive by DPP.
  __device__ void RowPut(int r, const double* yd) {
    mj::static_for<0, 7>([&](auto ic) { Ws(r * kRowSlots + decltype(ic)::value) = yd[decltype(ic)::value]; });
  }
  __device__ void RowGet(int r, double* yd) const {
    mj::static_for<0, 7>([&](auto ic) { yd[decltype(ic)::value] = Ws(r * kRowSlots + decltype(ic)::value); });
  }
  __device__ void RsPut(int r, int k, double v) {
    if (k == 0) Ws(r * kRowSlots + 8) = v;
    else if (k - 1 == l) Ws(r * kRowSlots + 7) = v;
  }
  __device__ void RsPut4(int r, double arr, double R, double b, double ainv) {
    const double lo = (l & 1) ? R : arr, hi = (l & 1) ? ainv : b;
    Ws(r * kRowSlots + 7) = (l & 2) ? hi : lo;
  }
  __device__ double RsGet(int r, int k) const {
    if (k == 0) return Ws(r * kRowSlots + 8);
    return mj::hum4::BcastQ(Ws(r * kRowSlots + 7), k - 1);
  }
  __device__ void RsGet4(int r, double* arr, double* R, double* b, double* ainv) const {
    const double v = Ws(r * kRowSlots + 7);
    *arr = Bcast<0>(v);
    *R = Bcast<1>(v);
    *b = Bcast<2>(v);
    *ainv = Bcast<3>(v);
  }
  // scalar k of row r0 + lane
  __device__ double RsGetLane(int r0, int k) const {
    if (k == 0) return Ws((r0 + l) * kRowSlots + 8);
    double t[4];
    mj::static_for<0, 4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      t[j] = mj::hum4::BcastQ(Ws((r0 + j) * kRowSlots + 7), k - 1);  // scalar k of row r0 + j
    });
    return l == 0 ? t[0] : (l == 1 ? t[1] : (l == 2 ? t[2] : t[3]));
  }
  __device__ double RowIndexLane(int r0) const { return (double)(r0 + l); }
  // overflow row o: entry k = (A)_{kRegRows + o, 4 k + l}
  __device__ void OvPut(int o, int k, double v) { Ws(kWsOv + o * kOvSlots + k) = v; }
  __device__ double OvGet(int o, int k) const { return Ws(kWsOv + o * kOvSlots + k); }
  // RK4 bookkeeping (Hum4::RkAdvance): 37 trunk numbers, number i kept by lane i & 3 in its slot
  // i >> 2 (read back by that lane, broadcast by DPP), + 16 limb numbers
  __device__ void RkPut(int i, double v) {
    if ((i & 3) == l) Ws(kWsRk + (i >> 2)) = v;
  }
  __device__ double RkGet(int i) const { return mj::hum4::BcastQ(Ws(kWsRk + (i >> 2)), i & 3); }
  __device__ void RkPutL(int i, double v) { Ws(kWsRk + 10 + i) = v; }
  __device__ double RkGetL(int i) const { return Ws(kWsRk + 10 + i); }
  __device__ double ShGetTriLane(int r0, int cc, int base) const {  // entry (r0 + lane, cc) of the packed triangle
    const int r = r0 + l;
    return ShGet((r >= cc ? r * (r + 1) / 2 + cc : cc * (cc + 1) / 2 + r) - base);
  }
  // the same without index arithmetic at the point of use: TriRow(r0) once, then entry (R0 + lane, CC) of the
  // packed triangle at base 0 is one LDS read at a per-lane base + a compile-time offset (only the 4 x 4 block
  // on the diagonal needs a per-lane select)
  using TriBase = int;
  __device__ int TriRow(int r0) const { return ((r0 + l) * (r0 + l + 1) / 2) * 16 + quad; }
  template <int R0, int CC, int BASE>
  __device__ double ShGetTriRow(int rb) const {  // (BASE: the triangle entry that sits in slot 0)
    constexpr int kUp = CC * (CC + 1) / 2 + R0 - BASE;  // + l: the entry above the diagonal, by symmetry
    if constexpr (CC < R0) return lds[kLdsSh + rb + (CC - BASE) * 16];
    else if constexpr (CC > R0 + 3) return lds[kLdsSh + (kUp + l) * 16 + quad];
    else return lds[kLdsSh + (R0 + l >= CC ? rb + (CC - BASE) * 16 : (kUp + l) * 16 + quad)];
  }
  __device__ void RecPut(int t, int k, double v) {
    if ((k & 3) == l) Ws(kMaxRows * kRowSlots + t * kRecSlots + (k >> 2)) = v;
  }
  __device__ double RecGet(int t, int k) const {
    const double own = Ws(kMaxRows * kRowSlots + t * kRecSlots + (k >> 2));
    return mj::hum4::BcastQ(own, k & 3);
  }
};

// act u <-> dof: u0 -> 7, u1 -> 6, u2 -> 8, u >= 3 -> u + 6 (humanoid.xml's actuator order)
__device__ __forceinline__ int CtrlOfDof(int d) { return d == 6 ? 1 : (d == 7 ? 0 : d - 6); }

template <class MP, bool kStandup>
__global__ __launch_bounds__(kBlock) void Humanoid4StepKernel(HumDev dev, CommonDev cm, StepArgs a,
                                                              const double* __restrict__ action,
                                                              OutPtrs out, HumTask task) {
  using Ctx = DevCtx<MP>;
  using Eng = H::Hum4<MP, Ctx>;
  constexpr T::TreeModel m = MP::kM;
  static_assert(m.act_dof[0] == 7 && m.act_dof[1] == 6 && m.act_dof[2] == 8 && m.act_dof[3] == 9 &&
                    m.act_dof[16] == 22,
                "CtrlOfDof");
  static constexpr H::LimbTab kTab = H::MakeLimbTab(MP::kM);
  static_assert(MP::kCacheRows <= kMaxOv && MP::kRegRows / 4 <= kOvSlots, "overflow rows fit the workspace");
  __shared__ double lds[kLdsElems];
  __shared__ typename Ctx::Tabs lds_tabs;  // 3.6 KB: pair / geom / body / limit tables, indexed at run time
  const int lane = threadIdx.x, l = lane & 3, quad = lane >> 2;
  {
    static_assert(sizeof(typename Ctx::Tabs) % 4 == 0, "word copy");
    const int* src = reinterpret_cast<const int*>(&Eng::kT);
    int* dst = reinterpret_cast<int*>(&lds_tabs);
    for (int i = lane; i < (int)(sizeof(typename Ctx::Tabs) / 4); i += kBlock) dst[i] = src[i];
  }
  static_assert(kUSlots == Eng::kFSlots, "LDS layout");
  for (int i = lane; i < H::kNLC * 4; i += kBlock) lds[kLdsTab + i] = kTab.c[i >> 2][i & 3];
  // The working region starts from zeros, not from what the previous kernel on this CU left behind:
  // slots of rows / candidates an env does not have are read with ZERO WEIGHTS before anything
  // writes them, and 0 x (a NaN bit pattern left by someone else's integers) is NaN.  Found by
  // tools/hum_poison_check.py (every CU's LDS filled with NaN patterns before each launch) after
  // the run-to-run determinism test failed once in many runs; 68 stores per lane per launch.
  for (int i = kLdsU + lane; i < kLdsElems; i += kBlock) lds[i] = 0.0;
  const int slot = blockIdx.x * kEnvsPerBlock + quad;
  const bool valid = slot < a.k;
  const int row = valid && dev.perm ? dev.perm[slot] : slot;  // cost-sorted scheduling
  const int e = valid ? (a.ids ? a.ids[row] - a.id_offset : row) : 0;
  const int n = cm.n;
  bool done = valid && cm.done[e] != 0;
  int cur = valid ? cm.cur_step[e] : 0;
  const bool reset = valid && (a.force_reset || done);
  // MujocoReset (mujoco_env.h:126-131) + MujocoResetModel (humanoid.h:129-141): one uniform
  // distribution for qpos and qvel.  The env's RNG stream is sequential: the quad's first lane
  // draws and hands the values to the other three through LDS.
  double* xch = lds + kLdsGeo;  // [slot][quad]; the geoms are written later
  if (reset && l == 0) {
    Mt19937 g(cm, e);
    for (int i = 0; i < m.nq; ++i) {
      static constexpr T::TreeModel mm = MP::kM;
      xch[i * 16 + quad] = mm.qpos0[i] + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
    }
    for (int i = 0; i < m.nv; ++i) {
      xch[(m.nq + i) * 16 + quad] = 0.0 + g.UniformReal(-task.reset_noise_scale, task.reset_noise_scale);
    }
    g.Commit();
  }
  __syncthreads();
  if (!valid) return;  // whole quads leave together
  Ctx c{&lds_tabs, (LdsDouble*)lds, (GlobalDouble*)(dev.ws + (size_t)blockIdx.x * kWsSlots * 64), lane, l, quad};
  // persistent state: qpos[24] qvel[23] warm[23] lag[2], SoA [slot][n]
  constexpr int kQ = 0, kV = 24, kW = 47, kLag = 70;
  auto get = [&](int slot) -> double {
    return reset ? (slot < kW ? xch[slot * 16 + quad] : 0.0) : dev.state[(size_t)slot * n + e];
  };
  int dof[H::kNS];  // this lane's global dofs (-1: the arms' dummy slot)
  mj::static_for<0, H::kNS>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int tab[4] = {H::LimbDof(0, s), H::LimbDof(1, s), H::LimbDof(2, s), H::LimbDof(3, s)};
    dof[s] = H::LaneInt(tab);
  });
  typename Eng::State s;
  mj::static_for<0, 10>([&](auto ic) { s.qt[decltype(ic)::value] = get(kQ + decltype(ic)::value); });
  mj::static_for<0, H::kNT>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    s.vt[j] = get(kV + j);
    s.wt[j] = get(kW + j);
  });
  mj::static_for<0, H::kNS>([&](auto sc) {
    constexpr int k = decltype(sc)::value;
    const bool real = dof[k] >= 0;
    const int d = real ? dof[k] : 9;
    s.ql[k] = real ? get(kQ + 1 + d) : 0.0;
    s.vl[k] = real ? get(kV + d) : 0.0;
    s.wl[k] = real ? get(kW + d) : 0.0;
  });
  __syncthreads();  // xch is about to be overwritten by the geoms
  if (reset) {
    cur = 0;
    done = false;
    s.ut[0] = s.ut[1] = s.ut[2] = 0.0;
    mj::static_for<0, H::kNS>([&](auto sc) { s.ul[decltype(sc)::value] = 0.0; });
  } else {
    ++cur;
    const double* act = action + (size_t)row * m.nu;
    s.ut[0] = act[CtrlOfDof(6)];  // clamped in mj_fwdActuation
    s.ut[1] = act[CtrlOfDof(7)];
    s.ut[2] = act[CtrlOfDof(8)];
    mj::static_for<0, H::kNS>([&](auto sc) {
      constexpr int k = decltype(sc)::value;
      s.ul[k] = dof[k] >= 0 ? act[CtrlOfDof(dof[k] >= 0 ? dof[k] : 9)] : 0.0;
    });
  }
  Eng::StoreTrunk(c, s, 15);
  // reset envs: mj_forward once; stepping envs: frame_skip x (4 RK stages).  Every env runs the
  // wave's trip count with its own state updates predicated (no divergent control flow).
  const int nfwd = reset ? 1 : 4 * task.frame_skip;
  const int nmax = mj::WaveAny(!reset) ? 4 * task.frame_skip : 1;
  H::Fwd<double> f;
  typename Eng::RowCount rows{0, 0, 0};
  double at[H::kNT], al[H::kNS];
  // WriteState, humanoid.h:225-268: qpos[skip:] qvel cinert cvel qfrc_actuator cfrc_ext.  The section
  // pointers are (re)computed where they are used: nothing that is only needed by the epilogue
  // should have to live through the step loop.
  struct ObsPtrs {
    double *q, *v, *ci, *cv, *act, *ce;
    int b0, a0;
  };
  auto obs_ptrs = [&]() {
    ObsPtrs o;
    o.b0 = task.exclude_worldbody ? 1 : 0;
    o.a0 = task.exclude_root_actuator ? 6 : 0;
    const int nb = m.nbody - o.b0;
    const int nobs = (m.nq - task.obs_skip) + m.nv + 22 * nb + (m.nv - o.a0);
    double* obs = (double*)out.p[kKeyEnv0] + (size_t)row * nobs;
    o.q = obs - task.obs_skip;
    o.v = obs + (m.nq - task.obs_skip);
    o.ci = o.v + m.nv - 10 * o.b0;                               // + 10 * body
    o.cv = o.v + m.nv + 10 * nb - 6 * o.b0;                      // + 6 * body
    o.act = o.v + m.nv + 16 * nb - o.a0;                         // + dof
    o.ce = o.v + m.nv + 16 * nb + (m.nv - o.a0) - 6 * o.b0;      // + 6 * body
    return o;
  };
  auto put6 = [](double* p, const H::Sp6<double>& x, bool on) {
    p[0] = on ? x.a.x : 0.0;
    p[1] = on ? x.a.y : 0.0;
    p[2] = on ? x.a.z : 0.0;
    p[3] = on ? x.l.x : 0.0;
    p[4] = on ? x.l.y : 0.0;
    p[5] = on ? x.l.z : 0.0;
  };
  const int bodyA = l == 0 ? 4 : (l == 1 ? 7 : (l == 2 ? 10 : 12));
  const int nlb = l < 2 ? 3 : 2;
  double mx = 0.0, my = 0.0;
  int stat[H::kNStat] = {};  // solver statistics of this env-step ("hum_debug" & 16: into info)
  for (int it = 0; it < nmax; ++it) {
    const bool live = it < nfwd;
    const bool last = it == nmax - 1;
    // cinert / cvel / qfrc_actuator of the LAST forward evaluation go straight to the observation
    // (an env that is only kept busy re-evaluates the same state: its values do not change)
    rows = Eng::Forward(c, s, f, live, at, al, task.debug, [&](const H::Fwd<double>& ff) {
      if (!last) return;  // wave uniform
      mx = ff.com.x;  // GetMassCenter, humanoid.h:212-223
      my = ff.com.y;
      const ObsPtrs o = obs_ptrs();
      double *o_ci = o.ci, *o_cv = o.cv, *o_act = o.act;
      const int b0 = o.b0, a0 = o.a0;
      if (l == 0) {
        if (b0 == 0) {
          for (int i = 0; i < 10; ++i) o_ci[i] = 0.0;
          for (int i = 0; i < 6; ++i) o_cv[i] = 0.0;
        }
        mj::static_for<1, H::kNTB + 1>([&](auto bc) {
          constexpr int b = decltype(bc)::value;
          mj::static_for<0, 10>([&](auto ic) { o_ci[10 * b + decltype(ic)::value] = ff.tci[b - 1].v[decltype(ic)::value]; });
          put6(o_cv + 6 * b, ff.tcv[b - 1], true);
        });
        mj::static_for<0, H::kNT>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if (j >= a0) o_act[j] = ff.act_t[j];
        });
      }
      mj::static_for<0, H::kNS>([&](auto sc) {
        constexpr int k = decltype(sc)::value;
        if (dof[k] >= 0) o_act[dof[k]] = ff.act_l[k];
      });
      mj::static_for<0, 3>([&](auto wc) {
        constexpr int w = decltype(wc)::value;
        if (w < nlb) {
          const int b = bodyA + w;
          mj::static_for<0, 10>([&](auto ic) { o_ci[10 * b + decltype(ic)::value] = ff.lci[w].v[decltype(ic)::value]; });
          put6(o_cv + 6 * b, ff.lcv[w > 1 ? 1 : w], true);
        }
      });
    }, stat);
    Eng::RkAdvance(c, s, it & 3, live && !reset, at, al);
  }
  Eng::LoadTrunk(c, s, 7);
  const ObsPtrs o = obs_ptrs();
  double *o_q = o.q, *o_v = o.v, *o_ce = o.ce;
  const int b0 = o.b0;
  const double x_before = reset ? 0.0 : dev.state[(size_t)kLag * n + e];  // the lagged mass centre
  const double y_before = reset ? 0.0 : dev.state[(size_t)(kLag + 1) * n + e];
  double ctrl_cost = 0.0;
  if (!reset) {
    const double* act = action + (size_t)row * m.nu;
    for (int i = 0; i < m.nu; ++i) {
      const double ai = act[i];
      ctrl_cost += task.ctrl_cost_weight * ai * ai;  // humanoid.h:171-174
    }
  }
  // mj_rnePostConstraint after the last mj_step (mujoco_env.h:145-147)
  const bool wrench = task.post_constraint != 0;
  H::Sp6<double> ext_t[H::kNTB + 1], ext_l[3];
  if (wrench) {
    Eng::ContactWrench(c, f, rows, ext_t, ext_l);
  } else {
    mj::static_for<0, H::kNTB + 1>([&](auto bc) { ext_t[decltype(bc)::value] = {{0, 0, 0}, {0, 0, 0}}; });
    mj::static_for<0, 3>([&](auto bc) { ext_l[decltype(bc)::value] = {{0, 0, 0}, {0, 0, 0}}; });
  }
  // what persists
  auto put = [&](int slot, double v) { dev.state[(size_t)slot * n + e] = v; };
  if (l == 0) {
    mj::static_for<0, 10>([&](auto ic) { put(kQ + decltype(ic)::value, s.qt[decltype(ic)::value]); });
    mj::static_for<0, H::kNT>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      put(kV + j, s.vt[j]);
      put(kW + j, s.wt[j]);
    });
    put(kLag, mx);
    put(kLag + 1, my);
  }
  mj::static_for<0, H::kNS>([&](auto sc) {
    constexpr int k = decltype(sc)::value;
    if (dof[k] >= 0) {
      put(kQ + 1 + dof[k], s.ql[k]);
      put(kV + dof[k], s.vl[k]);
      put(kW + dof[k], s.wl[k]);
    }
  });
  const bool have_cfrc = wrench && !reset;  // a reset leaves mj_resetData's zeros
  auto sq6 = [](const H::Sp6<double>& x) {
    return x.a.x * x.a.x + x.a.y * x.a.y + x.a.z * x.a.z + x.l.x * x.l.x + x.l.y * x.l.y + x.l.z * x.l.z;
  };
  float reward = 0.0f;
  double info[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (!reset) {
    double contact_cost = 0.0;
    if ((kStandup || task.use_contact_force) && have_cfrc) {  // humanoid.h:180-187
      double cc = 0.0;
      mj::static_for<0, H::kNTB + 1>([&](auto bc) { cc += sq6(ext_t[decltype(bc)::value]); });
      cc += H::SumQ(sq6(ext_l[0]) + sq6(ext_l[1]) + sq6(ext_l[2]));
      contact_cost = task.contact_cost_weight * cc;
      contact_cost = contact_cost < task.contact_cost_max ? contact_cost : task.contact_cost_max;
    }
    const double z = s.qt[2];
    if constexpr (kStandup) {  // humanoid_standup.h:160-185
      const double xv = z / m.timestep;
      reward = static_cast<float>(xv * task.forward_reward_weight + task.healthy_reward - ctrl_cost -
                                  contact_cost);
      done = cur >= a.max_episode_steps;
      info[0] = xv * task.forward_reward_weight;
      info[1] = -ctrl_cost;
      info[2] = task.healthy_reward;
      info[3] = -contact_cost;
      if (task.debug & 16) {
        info[0] = stat[0];
        info[1] = stat[1];
        info[2] = stat[2];
        info[3] = stat[3];
      }
      if (task.debug & (32 | 64 | 128 | 256)) {  // stage timers (diagnostic build)
        const int o = (task.debug & 32) ? 5 : ((task.debug & 64) ? 9 : ((task.debug & 128) ? 13 : 17));
        info[0] = stat[o];
        info[1] = stat[o + 1];
        info[2] = stat[o + 2];
        info[3] = stat[o + 3];
      }
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
      if (task.debug & 16) {
        info[4] = stat[0];
        info[5] = stat[1];
        info[6] = stat[2];
        info[7] = stat[3];
      }
      if (task.debug & (32 | 64 | 128 | 256)) {  // stage timers (diagnostic build)
        const int o = (task.debug & 32) ? 5 : ((task.debug & 64) ? 9 : ((task.debug & 128) ? 13 : 17));
        info[4] = stat[o];
        info[5] = stat[o + 1];
        info[6] = stat[o + 2];
        info[7] = stat[o + 3];
      }
    }
  } else {
    // the reset WriteState stores `-ctrl_cost` / `-contact_cost` of +0.0: -0.0 (humanoid.h:272-274)
    info[1] = -0.0;
    info[3] = -0.0;
    if (kStandup) info[2] = task.healthy_reward;  // WriteState(0, 0, 0, 0): reward_alive is the constant
  }
  if (l == 0) {
    mj::static_for<0, 10>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (i >= task.obs_skip) o_q[i] = s.qt[i];
    });
    mj::static_for<0, H::kNT>([&](auto jc) { o_v[decltype(jc)::value] = s.vt[decltype(jc)::value]; });
    if (b0 == 0) put6(o_ce, ext_t[0], have_cfrc);
    mj::static_for<1, H::kNTB + 1>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      put6(o_ce + 6 * b, ext_t[b], have_cfrc);
    });
    constexpr int ninfo = kStandup ? 4 : 9;
    for (int i = 0; i < ninfo; ++i) ((double*)out.p[kKeyEnv0 + 1 + i])[row] = info[i];
    cm.done[e] = done ? 1 : 0;
    cm.cur_step[e] = cur;
    dev.cost[e] = reset ? 0 : stat[4];
    WriteCommon(out, row, e + a.id_offset, cur, done, reward, a.max_episode_steps);
  }
  mj::static_for<0, H::kNS>([&](auto sc) {
    constexpr int k = decltype(sc)::value;
    if (dof[k] >= 0) {
      o_q[1 + dof[k]] = s.ql[k];
      o_v[dof[k]] = s.vl[k];
    }
  });
  mj::static_for<0, 3>([&](auto wc) {
    constexpr int w = decltype(wc)::value;
    if (w < nlb) put6(o_ce + 6 * (bodyA + w), ext_l[w], have_cfrc);
  });
}

}  // namespace

void Hum4LaunchStep(hipStream_t st, bool standup, int blocks, HumDev dev, CommonDev cm, StepArgs a,
                    const double* act, OutPtrs out, HumTask task) {
  if (standup) {
    hipLaunchKernelGGL((Humanoid4StepKernel<StandupMP, true>), dim3(blocks), dim3(kBlock), 0, st, dev, cm,
                       a, act, out, task);
  } else {
    hipLaunchKernelGGL((Humanoid4StepKernel<HumanoidMP, false>), dim3(blocks), dim3(kBlock), 0, st, dev, cm,
                       a, act, out, task);
  }
}
namespace {
// Stable counting sort of the launch's rows by their env's cost, one workgroup: thread t owns the
// contiguous chunk of rows [t C, (t + 1) C), counts its rows per bucket, a scan over (bucket major,
// thread minor) gives every thread its start offset per bucket, and it then places its rows in
// order.  Deterministic (no atomics): the same costs always give the same waves.
constexpr int kSortThreads = 512, kSortBuckets = 16;
__device__ __forceinline__ int CostBucket(int cost) {
  // the env's solver cost of its last env-step (row visits + rows built, see Hum4::Forward):
  // geometric buckets, 0 | 1-63 | 64-127 | ... | >= 2^18
  const int b = cost <= 0 ? 0 : 32 - __clz(cost >> 5);
  return b < kSortBuckets - 1 ? b : kSortBuckets - 1;
}
__global__ __launch_bounds__(kSortThreads) void Hum4SortKernel(HumDev dev, StepArgs a) {
  __shared__ int cnt[kSortBuckets][kSortThreads];
  __shared__ int base[kSortBuckets + 1];
  const int t = threadIdx.x;
  const int chunk = (a.k + kSortThreads - 1) / kSortThreads;
  const int lo = t * chunk, hi = lo + chunk < a.k ? lo + chunk : a.k;
  int mine[kSortBuckets];
  for (int b = 0; b < kSortBuckets; ++b) mine[b] = 0;
  for (int r = lo; r < hi; ++r) {
    const int e = a.ids ? a.ids[r] - a.id_offset : r;
    const int b = CostBucket(dev.cost[e]);
    for (int q = 0; q < kSortBuckets; ++q) mine[q] += q == b ? 1 : 0;
  }
  for (int b = 0; b < kSortBuckets; ++b) cnt[b][t] = mine[b];
  __syncthreads();
  // exclusive scan over threads, per bucket (thread b < 16 scans bucket b: 1024 adds)
  if (t < kSortBuckets) {
    int run = 0;
    for (int i = 0; i < kSortThreads; ++i) {
      const int v = cnt[t][i];
      cnt[t][i] = run;
      run += v;
    }
    base[t + 1] = run;
  }
  __syncthreads();
  if (t == 0) {
    base[0] = 0;
    for (int b = 0; b < kSortBuckets; ++b) base[b + 1] += base[b];
  }
  __syncthreads();
  // most expensive bucket first: the long waves start early, the short ones fill the tail
  int at[kSortBuckets];
  for (int b = 0; b < kSortBuckets; ++b) at[b] = (a.k - base[b + 1]) + cnt[b][t];
  for (int r = lo; r < hi; ++r) {
    const int e = a.ids ? a.ids[r] - a.id_offset : r;
    const int b = CostBucket(dev.cost[e]);
    int pos = 0;
    for (int q = 0; q < kSortBuckets; ++q) {
      if (q == b) pos = at[q]++;
    }
    dev.perm[pos] = r;
  }
}
}  // namespace

void Hum4LaunchSort(hipStream_t st, HumDev dev, StepArgs a) {
  hipLaunchKernelGGL(Hum4SortKernel, dim3(1), dim3(kSortThreads), 0, st, dev, a);
}

size_t Hum4WorkspaceBytes(int num_envs) {
  const size_t blocks = ((size_t)num_envs + kEnvsPerBlock - 1) / kEnvsPerBlock;
  return sizeof(double) * blocks * 64 * (size_t)kWsSlots;
}

}  // namespace epa
