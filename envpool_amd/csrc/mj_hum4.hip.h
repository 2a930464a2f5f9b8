// K3d -- `mj_step` for gym Humanoid / HumanoidStandup with ONE ENV SPLIT OVER FOUR LANES.
//
// Same arithmetic as mj_tree.hip.h (MuJoCo 3.6.0's mj_forward / mj_step / mj_rnePostConstraint
// for humanoid.xml / humanoidstandup.xml as called from envpool/mujoco/gym/mujoco_env.h:126-148;
// task code envpool/mujoco/gym/humanoid.h, humanoid_standup.h), re-laid out so that NOTHING of a
// forward pass lives in HBM except the constraint rows:
//
//  * the humanoid is a 3-body trunk (torso, lwaist, pelvis: free joint + 3 hinges = 9 dofs) that
//    carries four limbs (right leg, left leg, right arm, left arm: 4/4/3/3 hinges, 3/3/2/2
//    bodies).  Lane l of a quad owns limb l; the trunk is env-level data, computed identically
//    by the four lanes (so every per-env decision is uniform over the quad by construction).
//    M is an "arrow": a dense 9x9 trunk block, four limb blocks (4x4) and their couplings
//    (4x9).  A lane holds 45 + 36 + 10 numbers of M instead of the 185 of the 23x23, a
//    23-vector is 9 (trunk, replicated) + 4 (limb), and L'DL eliminates the limbs inside their
//    lanes, sums the four Schur complements over the quad (DPP) and factors the trunk block
//    redundantly -- an M^-1 solve touches no memory (mj_tree.hip.h reloads 231 factor entries
//    from HBM per constraint row);
//  * arms are legs with a dummy third hip dof (zero axis, unit armature: exactly decoupled) and a
//    massless third body, so all four lanes run the SAME instructions; what differs between
//    limbs is a table of ~120 per-limb constants (Ctx::LC);
//  * constraint rows are kept as  y_r = L^-T J_r'  (9 + 4 numbers per lane; the row's W = M^-1 J'
//    is never formed):  A_rc = sum_i y_r[i] y_c[i] / D_i,  qacc = qacc_smooth + L^-1 D^-1 sum_r
//    f_r y_r.  With <= 16 rows A + R, b and f sit in LDS and the PGS sweeps touch no memory; with
//    more rows (HumanoidStandup on the floor) the sweep streams y rows and keeps  z = sum_c f_c
//    y_c / D  in registers (res_r = b_r + R_r f_r + y_r . z);
//  * collision candidates are static lists as in mj_tree.hip.h; a lane tests a quarter of the
//    109 geom pairs against the env's 17 geoms in LDS, the active-group masks are OR-reduced over
//    the quad, and rows are built env by env in MuJoCo's order (the unconverged PGS sweep is
//    order dependent), every env taking ITS t-th active group in iteration t.
//
// Two value kinds: V is a lane's value (device: double; host emulation: Q4<double>, four
// lanes in a struct), E an env-level value that is identical on the quad's lanes (device:
// double, recomputed per lane; host: a plain double, computed once).  Quad reductions
// (SumQ, OrQ) turn a V into an E.  The same source runs on the host (tests/cpu_harness).
#ifndef ENVPOOL_AMD_CSRC_MJ_HUM4_HIP_H_
#define ENVPOOL_AMD_CSRC_MJ_HUM4_HIP_H_

#include "mj_ant.hip.h"  // Vec3, Mat3, Sp6, In10 and their algebra (generic over the value type)
#include "mj_quad.hip.h"
#include "mj_tree.hip.h"  // TreeModel

namespace epa {
namespace mj {
namespace hum4 {

using ant::Cross;
using ant::CrossForce;
using ant::CrossMotion;
using ant::Dot;
using ant::In10;
using ant::Mat3;
using ant::Mul;
using ant::MulInert;
using ant::Sp6;
using ant::Vec3;
using tree::TreeModel;
using mj::Sel;
using mj::Sum4;

// stage timers of the diagnostic build ("hum_debug" & 32 / 64: stat[5..12] += cycles / 16 of
// position + detection, smooth dynamics, rows, solver staging, sweeps, solver epilogue, streaming solver; forwards)
#if defined(EPA_HUM_DEBUG) && defined(__HIP_DEVICE_COMPILE__)
#define EPA_HUM_TICK() ((long long)clock64())
#else
#define EPA_HUM_TICK() (0ll)
#endif
// a function that is NOT inlined on the device (its own register allocation; see Hum4::ConstraintStage)
// (the device context is a handful of pointers and goes by value; the host harness's context owns its storage)
#if defined(__HIP_DEVICE_COMPILE__)
#define EPA_HUM_NOINLINE __device__ __noinline__
#define EPA_HUM_CTX(Ctx) Ctx
#else
#define EPA_HUM_NOINLINE inline
#define EPA_HUM_CTX(Ctx) Ctx&
#endif
// a register value the optimiser cannot rematerialise or fold
#if defined(__HIP_DEVICE_COMPILE__)
#define EPA_HUM_PIN(x) asm volatile("" : "+v"(x))
#else
#define EPA_HUM_PIN(x) ((void)0)
#endif
// counters that only the diagnostic build reports (the host harness counts too)
#if defined(EPA_HUM_DEBUG) || !defined(__HIP_DEVICE_COMPILE__)
#define EPA_HUM_COUNT(x) (++(x))
#else
#define EPA_HUM_COUNT(x) ((void)0)
#endif
constexpr int kNStat = 20;
constexpr int kNT = 9;        // trunk dofs 0..8 (free joint 0..5, abdomen z, y, x)
constexpr int kNS = 4;        // limb dof slots of a lane: A0 A1 A2 (first limb body), B (second)
constexpr int kNLimb = 4;
constexpr int kNTB = 3;       // trunk bodies 1, 2, 3

// ---- the model, re-indexed by limb ----------------------------------------------------------
// bodies: limb l has A, B (and C for the legs: the foot, welded to the shin)
constexpr int kLimbA[kNLimb] = {4, 7, 10, 12};
constexpr int kLimbDof0[kNLimb] = {9, 13, 17, 20};
constexpr bool kLimbIsLeg[kNLimb] = {true, true, false, false};
// global dof of limb slot s (-1: the arms' dummy slot)
EPA_HD constexpr int LimbDof(int l, int s) {
  return kLimbIsLeg[l] ? kLimbDof0[l] + s : (s < 2 ? kLimbDof0[l] + s : (s == 3 ? kLimbDof0[l] + 2 : -1));
}
EPA_HD constexpr int LimbOfDof(int d) { return d < 9 ? -1 : (d < 13 ? 0 : (d < 17 ? 1 : (d < 20 ? 2 : 3))); }
EPA_HD constexpr int SlotOfDof(int d) {
  const int l = LimbOfDof(d);
  if (l < 0) return -1;
  const int k = d - kLimbDof0[l];
  return kLimbIsLeg[l] ? k : (k < 2 ? k : 3);
}

// indices into the per-limb constant table
enum LCIdx : int {
  kLcApos = 0, kLcAipos = 3, kLcAin = 6, kLcAmass = 12,
  kLcJpos = 13,   // 4 joints x 3
  kLcJaxis = 25,  // 4 joints x 3
  kLcBpos = 37, kLcBipos = 40, kLcBin = 43, kLcBmass = 49,
  kLcCpos = 50, kLcCipos = 53, kLcCin = 56, kLcCmass = 62,
  kLcLo = 63, kLcHi = 67, kLcStiff = 71, kLcDamp = 75, kLcArm = 79, kLcGear = 83, kLcInvw = 87,
  kLcG0pos = 91, kLcG0ax = 94, kLcG0hl = 97,  // capsule on A (centre, unit axis, half length)
  kLcG1pos = 98, kLcG1ax = 101, kLcG1hl = 104,  // capsule on B
  kLcG2off = 105,                               // sphere: offset in B's frame
  kLcIsLeg = 108,
  kLcG0rad = 109, kLcG1rad = 110, kLcG2rad = 111,
  kNLC = 112
};
struct LimbTab {
  double c[kNLC][kNLimb];
};
constexpr LimbTab MakeLimbTab(const TreeModel& m) {
  LimbTab t{};
  for (int l = 0; l < kNLimb; ++l) {
    const int A = kLimbA[l], B = A + 1, C = A + 2;
    const bool leg = kLimbIsLeg[l];
    for (int k = 0; k < 3; ++k) {
      t.c[kLcApos + k][l] = m.body_pos[A][k];
      t.c[kLcAipos + k][l] = m.body_ipos[A][k];
      t.c[kLcBpos + k][l] = m.body_pos[B][k];
      t.c[kLcBipos + k][l] = m.body_ipos[B][k];
      t.c[kLcCpos + k][l] = leg ? m.body_pos[C][k] : 0.0;
      t.c[kLcCipos + k][l] = leg ? m.body_ipos[C][k] : 0.0;
    }
    for (int k = 0; k < 6; ++k) {
      t.c[kLcAin + k][l] = m.body_inertia[A][k];
      t.c[kLcBin + k][l] = m.body_inertia[B][k];
      t.c[kLcCin + k][l] = leg ? m.body_inertia[C][k] : 0.0;
    }
    t.c[kLcAmass][l] = m.body_mass[A];
    t.c[kLcBmass][l] = m.body_mass[B];
    t.c[kLcCmass][l] = leg ? m.body_mass[C] : 0.0;
    for (int s = 0; s < kNS; ++s) {
      const int d = LimbDof(l, s);
      const int j = d < 0 ? -1 : d - 5;  // joint index: joint 0 is the free joint (6 dofs)
      for (int k = 0; k < 3; ++k) {
        t.c[kLcJpos + 3 * s + k][l] = j < 0 ? 0.0 : m.jnt_pos[j][k];
        t.c[kLcJaxis + 3 * s + k][l] = j < 0 ? 0.0 : m.jnt_axis[j][k];
      }
      // the dummy slot: never limited, no passive force, unit inertia, no actuator
      t.c[kLcLo + s][l] = j < 0 ? -1e30 : m.jnt_lo[j];
      t.c[kLcHi + s][l] = j < 0 ? 1e30 : m.jnt_hi[j];
      t.c[kLcStiff + s][l] = j < 0 ? 0.0 : m.jnt_stiff[j];
      t.c[kLcDamp + s][l] = d < 0 ? 0.0 : m.dof_damp[d];
      t.c[kLcArm + s][l] = d < 0 ? 1.0 : m.dof_arm[d];
      t.c[kLcInvw + s][l] = d < 0 ? 1.0 : m.dof_invw[d];
      double gear = 0.0;
      for (int u = 0; u < m.nu; ++u) {
        if (d >= 0 && m.act_dof[u] == d) gear = m.act_gear[u];
      }
      t.c[kLcGear + s][l] = gear;
    }
    // geoms: legs 6 7 8 / 9 10 11, arms 12 13 14 / 15 16 17
    const int g0 = leg ? 6 + 3 * l : 12 + 3 * (l - 2);
    for (int k = 0; k < 3; ++k) {
      t.c[kLcG0pos + k][l] = m.geom_pos[g0][k];
      t.c[kLcG0ax + k][l] = m.geom_axis[g0][k];
      t.c[kLcG1pos + k][l] = m.geom_pos[g0 + 1][k];
      t.c[kLcG1ax + k][l] = m.geom_axis[g0 + 1][k];
      t.c[kLcG2off + k][l] = (leg ? m.body_pos[C][k] : 0.0) + m.geom_pos[g0 + 2][k];
    }
    t.c[kLcG0hl][l] = m.geom_hl[g0];
    t.c[kLcG1hl][l] = m.geom_hl[g0 + 1];
    t.c[kLcG0rad][l] = m.geom_rad[g0];
    t.c[kLcG1rad][l] = m.geom_rad[g0 + 1];
    t.c[kLcG2rad][l] = m.geom_rad[g0 + 2];
    t.c[kLcIsLeg][l] = leg ? 1.0 : 0.0;
  }
  return t;
}
// what the layout assumes about the model (checked at compile time where it is instantiated)
constexpr bool CheckTopology(const TreeModel& m) {
  if (m.nbody != 14 || m.nv != 23 || m.nq != 24 || m.ngeom != 18 || m.nu != 17) return false;
  if (m.nlimit != 17 || m.nfloor != 29) return false;
  const int parent[14] = {0, 0, 1, 2, 3, 4, 5, 3, 7, 8, 1, 10, 1, 12};
  const int dofnum[14] = {0, 6, 2, 1, 3, 1, 0, 3, 1, 0, 2, 1, 2, 1};
  for (int b = 1; b < 14; ++b) {
    if (m.body_parent[b] != parent[b] || m.body_dofnum[b] != dofnum[b]) return false;
    if (b >= 4 && (m.body_quat[b][0] != 1.0 || m.body_quat[b][1] != 0.0 || m.body_quat[b][2] != 0.0 ||
                   m.body_quat[b][3] != 0.0)) {
      return false;  // limb bodies carry no frame rotation of their own
    }
  }
  const int geom_body[18] = {0, 1, 1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 11, 12, 13, 13};
  const int geom_type[18] = {0, 3, 2, 3, 3, 3, 3, 3, 2, 3, 3, 2, 3, 3, 2, 3, 3, 2};
  for (int g = 0; g < 18; ++g) {
    if (m.geom_body[g] != geom_body[g] || m.geom_type[g] != geom_type[g]) return false;
  }
  for (int g = 0; g < m.nlimit; ++g) {
    if (m.jnt_dadr[m.limit_jnt[g]] != 6 + g) return false;  // limit group g <-> dof 6 + g
  }
  // floor candidates in geom order: capsule +end, -end; sphere
  int c = 0;
  for (int g = 1; g < 18; ++g) {
    if (geom_type[g] == 3) {
      if (m.floor_geom[c] != g || m.floor_sign[c] != 1.0 || m.floor_geom[c + 1] != g ||
          m.floor_sign[c + 1] != -1.0) {
        return false;
      }
      c += 2;
    } else {
      if (m.floor_geom[c] != g) return false;
      c += 1;
    }
  }
  for (int u = 0; u < m.nu; ++u) {
    if (m.act_dof[u] < 6) return false;
  }
  return c == 29;
}

// ---- the two value kinds ---------------------------------------------------------------------
template <typename V>
struct EnvOf {
  using type = V;
};
template <typename T>
struct EnvOf<Q4<T>> {
  using type = T;
};
// host: lane masks of a quad, one 64-bit word per lane
struct M4 {
  unsigned long long v[4];
};
template <typename V>
struct MaskOf {
  using type = unsigned long long;
};
template <typename T>
struct MaskOf<Q4<T>> {
  using type = M4;
};

// quad reductions: V -> E
template <typename T>
inline T SumQ(const Q4<T>& x) {
  return (x.v[0] + x.v[1]) + (x.v[2] + x.v[3]);
}
EPA_HD double SumQ(double x) { return Sum4(x); }
inline unsigned long long OrQ(const M4& m) { return m.v[0] | m.v[1] | m.v[2] | m.v[3]; }
EPA_HD unsigned long long OrQ(unsigned long long m) {
#if defined(__HIP_DEVICE_COMPILE__)
  int lo = (int)(unsigned)m, hi = (int)(unsigned)(m >> 32);
  EPA_QUAD_REDUCE(lo, EPA_QUAD_OR);
  EPA_QUAD_REDUCE(hi, EPA_QUAD_OR);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
#else
  return m;
#endif
}
// lane k of the quad (k env-level)
template <typename V>
struct LaneOps {  // device / scalar host
  using B = bool;
  static EPA_HD bool Is(int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(threadIdx.x & 3u) == k;
#else
    return k == 0;
#endif
  }
  static EPA_HD void SetBit(unsigned long long& m, bool on, int bit) { m |= (on ? 1ull : 0ull) << bit; }
  // f(lane, words): lane-level code whose data differs per lane (the lane's quarter of the pairs)
  template <typename F>
  static EPA_HD void PerLane(unsigned long long (&w)[3], F&& f) {
#if defined(__HIP_DEVICE_COMPILE__)
    f((int)(threadIdx.x & 3u), w);
#else
    f(0, w);
#endif
  }
};
template <typename T>
struct LaneOps<Q4<T>> {
  using B = B4;
  static B4 Is(int k) { return {{k == 0, k == 1, k == 2, k == 3}}; }
  static void SetBit(M4& m, B4 on, int bit) {
    for (int i = 0; i < 4; ++i) m.v[i] |= (on.v[i] ? 1ull : 0ull) << bit;
  }
  template <typename F>
  static void PerLane(M4 (&w)[3], F&& f) {
    for (int l = 0; l < 4; ++l) {
      unsigned long long t[3] = {w[0].v[l], w[1].v[l], w[2].v[l]};
      f(l, t);
      for (int k = 0; k < 3; ++k) w[k].v[l] = t[k];
    }
  }
};
// a per-lane integer constant (tab[l] for the lane of limb l)
EPA_HD int LaneInt(const int (&tab)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (the four values, -128 .. 127 each, packed into one constant and picked with a shift: written as a chain of
  // selects this compiled to lane-divergent BRANCHES -- ~16 instructions per use, 8 uses per constraint row)
  const unsigned packed = ((unsigned)tab[0] & 0xffu) | (((unsigned)tab[1] & 0xffu) << 8) |
                          (((unsigned)tab[2] & 0xffu) << 16) | (((unsigned)tab[3] & 0xffu) << 24);
  return (int)(signed char)((packed >> ((threadIdx.x & 3u) * 8u)) & 0xffu);
#else
  return tab[0];
#endif
}
// set bit tab[lane] of a lane mask
EPA_HD void SetBitLane(unsigned long long& m, bool on, const int (&tab)[4]) {
  m |= (on ? 1ull : 0ull) << LaneInt(tab);
}
inline void SetBitLane(M4& m, B4 on, const int (&tab)[4]) {
  for (int i = 0; i < 4; ++i) m.v[i] |= (on.v[i] ? 1ull : 0ull) << tab[i];
}
// bit tab[lane] of an env-level mask as a lane value 0 / 1
EPA_HD double BitLane(unsigned m, const int (&tab)[4], double) { return (double)((m >> LaneInt(tab)) & 1u); }
template <typename T>
inline Q4<T> BitLane(unsigned m, const int (&tab)[4], Q4<T>) {
  Q4<T> r;
  for (int i = 0; i < 4; ++i) r.v[i] = (T)((m >> tab[i]) & 1u);
  return r;
}
// env-level condition selecting lane values (host: every lane of the Q4 is the same env)
template <typename T>
inline Q4<T> Sel(bool c, const Q4<T>& a, const Q4<T>& b) {
  return c ? a : b;
}
// a lane value of lane k as an env-level value
template <typename T>
inline T BcastQ(const Q4<T>& x, int k) {
  return x.v[k];
}
EPA_HD double BcastQ(double x, int k) { return SumQ(Sel(LaneOps<double>::Is(k), x, 0.0)); }

// entries base + 0..3 of an env-level array, one per lane of the quad
EPA_HD double LanePick4(const double* x, int base) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (a tree of selects on the lane's two bits: the chain form compiled to lane-divergent branches)
  const unsigned l = threadIdx.x & 3u;
  const double lo = (l & 1u) ? x[base + 1] : x[base], hi = (l & 1u) ? x[base + 3] : x[base + 2];
  return (l & 2u) ? hi : lo;
#else
  return x[base];
#endif
}
template <typename T>
inline Q4<T> LanePick4(const T* x, int base, Q4<T> = Q4<T>()) {
  Q4<T> r;
  for (int i = 0; i < 4; ++i) r.v[i] = x[base + i];
  return r;
}
// entry `slot` of limb `limb`'s part of a distributed vector, as an env-level value
template <typename T>
inline T LimbPick(const Q4<T>* xl, int limb, int slot) {
  return xl[slot].v[limb];
}
EPA_HD double LimbPick(const double* xl, int limb, int slot) {
  const double x = slot == 0 ? xl[0] : (slot == 1 ? xl[1] : (slot == 2 ? xl[2] : xl[3]));
  return BcastQ(x, limb);
}
// a lane vector that is `val` in slot `slot` of limb `limb` and 0 elsewhere
template <typename T>
inline void LimbUnit(Q4<T>* xl, int limb, int slot, T val) {
  for (int s = 0; s < kNS; ++s) {
    for (int l = 0; l < 4; ++l) xl[s].v[l] = (s == slot && l == limb) ? val : T(0);
  }
}
EPA_HD void LimbUnit(double* xl, int limb, int slot, double val) {
  const bool mine = LaneOps<double>::Is(limb);
  for (int s = 0; s < kNS; ++s) xl[s] = (mine && s == slot) ? val : 0.0;
}

// max(x, 0) (device: one v_max_f64; the generic form is a compare and two selects per half)
template <typename T>
inline Q4<T> Max0(const Q4<T>& x) {
  Q4<T> r;
  for (int i = 0; i < 4; ++i) r.v[i] = x.v[i] > T(0) ? x.v[i] : T(0);
  return r;
}
EPA_HD double Max0(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fmax(x, 0.0);
#else
  return x > 0.0 ? x : 0.0;
#endif
}
// 1 in lane K of the quad, 0 in the others
template <int K, typename T>
inline Q4<T> LaneUnit(Q4<T>) {
  Q4<T> r(T(0));
  r.v[K] = T(1);
  return r;
}
template <int K>
EPA_HD double LaneUnit(double) {
  return LaneOps<double>::Is(K) ? 1.0 : 0.0;
}
// lane K (compile time) of the quad as an env-level value
template <int K, typename T>
inline T BcastQS(const Q4<T>& x) {
  return x.v[K];
}
template <int K>
EPA_HD double BcastQS(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return DppMov<K * 0x55>(x);  // quad_perm:[K,K,K,K]
#else
  return x;
#endif
}

template <typename X>
EPA_HD X AbsX(X x) {
  return Sel(x < X(0), -x, x);
}
template <typename X>
EPA_HD X MaxX(X a, X b) {
  return Sel(a > b, a, b);
}
template <typename X>
EPA_HD X MinX(X a, X b) {
  return Sel(a < b, a, b);
}
template <typename X>
EPA_HD X ClampX(X x, X lo, X hi) {
  return Sel(x < lo, lo, Sel(x > hi, hi, x));
}
EPA_HD double SqrtX(double x) { return ::sqrt(x); }
template <typename T>
inline Q4<T> SqrtX(const Q4<T>& x) {
  return sqrt(x);
}

template <typename X>
struct Quat {
  X w, x, y, z;
};
template <typename X>
EPA_HD Quat<X> QMul(Quat<X> a, Quat<X> b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
template <typename X>
EPA_HD Quat<X> QNormalize(Quat<X> q) {
  const X inv = X(1) / SqrtX(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
template <typename X>
EPA_HD Mat3<X> QMat(Quat<X> q) {
  const X w = q.w, x = q.x, y = q.y, z = q.z;
  Mat3<X> M;
  M.m[0] = w * w + x * x - y * y - z * z;
  M.m[4] = w * w - x * x + y * y - z * z;
  M.m[8] = w * w - x * x - y * y + z * z;
  M.m[1] = X(2) * (x * y - w * z);
  M.m[2] = X(2) * (x * z + w * y);
  M.m[3] = X(2) * (x * y + w * z);
  M.m[5] = X(2) * (y * z - w * x);
  M.m[6] = X(2) * (x * z - w * y);
  M.m[7] = X(2) * (y * z + w * x);
  return M;
}
template <typename X>
struct Frame {
  Vec3<X> pos;
  Quat<X> q;
  Mat3<X> R;
};
// cinert of a body: inertia (xx yy zz xy xz yz about its COM, body axes) rotated by R, shifted by
// off = xipos - com (mj_comPos); same operation order as Tree::CinertBatch
template <typename X>
EPA_HD In10<X> CinertOf(const X* I, X mass, const Mat3<X>& Rm, Vec3<X> off) {
  const X* Rb = Rm.m;
  X RI[9];
  static_for<0, 3>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    RI[3 * r + 0] = Rb[3 * r] * I[0] + Rb[3 * r + 1] * I[3] + Rb[3 * r + 2] * I[4];
    RI[3 * r + 1] = Rb[3 * r] * I[3] + Rb[3 * r + 1] * I[1] + Rb[3 * r + 2] * I[5];
    RI[3 * r + 2] = Rb[3 * r] * I[4] + Rb[3 * r + 1] * I[5] + Rb[3 * r + 2] * I[2];
  });
  auto iw = [&](int r, int c) {
    return RI[3 * r] * Rb[3 * c] + RI[3 * r + 1] * Rb[3 * c + 1] + RI[3 * r + 2] * Rb[3 * c + 2];
  };
  const X o2 = Dot(off, off);
  In10<X> ci;
  ci.v[0] = iw(0, 0) + mass * (o2 - off.x * off.x);
  ci.v[1] = iw(1, 1) + mass * (o2 - off.y * off.y);
  ci.v[2] = iw(2, 2) + mass * (o2 - off.z * off.z);
  ci.v[3] = iw(0, 1) - mass * off.x * off.y;
  ci.v[4] = iw(0, 2) - mass * off.x * off.z;
  ci.v[5] = iw(1, 2) - mass * off.y * off.z;
  ci.v[6] = mass * off.x;
  ci.v[7] = mass * off.y;
  ci.v[8] = mass * off.z;
  ci.v[9] = mass;
  return ci;
}
template <typename X>
EPA_HD In10<X> AddIn(const In10<X>& a, const In10<X>& b) {
  In10<X> r;
  static_for<0, 10>([&](auto kc) { r.v[decltype(kc)::value] = a.v[decltype(kc)::value] + b.v[decltype(kc)::value]; });
  return r;
}
template <typename X>
EPA_HD Sp6<X> AddSp(const Sp6<X>& a, const Sp6<X>& b) {
  return {a.a + b.a, a.l + b.l};
}
template <typename X, typename S>
EPA_HD void AxpySp(Sp6<X>& y, const Sp6<X>& x, S s) {
  y.a = y.a + x.a * X(s);
  y.l = y.l + x.l * X(s);
}

// packed lower triangle of the trunk block: (i, j), j <= i
EPA_HD constexpr int TT(int i, int j) { return i * (i + 1) / 2 + j; }
constexpr int kNTT = kNT * (kNT + 1) / 2;  // 45
// limb block: (s, t), t <= s
EPA_HD constexpr int LL(int s, int t) { return s * (s + 1) / 2 + t; }
constexpr int kNLL = kNS * (kNS + 1) / 2;  // 10
// trunk body of a trunk dof: 0..5 torso, 6 7 lwaist, 8 pelvis (index into the 3 trunk bodies)
EPA_HD constexpr int TrunkBodyOfDof(int j) { return j < 6 ? 0 : (j < 8 ? 1 : 2); }

// everything one forward pass derives from (qpos, qvel) before the constraint solve
template <typename V>
struct Fwd {
  using E = typename EnvOf<V>::type;
  Sp6<V> lcd[kNS];   // cdof, limb (the trunk's: Ctx::TcdGet)
  In10<E> tci[kNTB];  // cinert of bodies 1 2 3
  In10<V> lci[3];     // cinert of A B C
  Sp6<E> tcv[kNTB];   // cvel of bodies 1 2 3
  Sp6<V> lcv[2];      // cvel of A, B (C moves with B)
  Vec3<E> com;
  // L'DL of M: L strictly lower (unit diagonal implied), D^-1; the trunk block: Ctx::LttGet, DtGet
  V Llt[kNS][kNT], Lll[kNLL], dinv_l[kNS];
  E act_t[kNT];  // qfrc_actuator
  V act_l[kNS];
  E accs_t[kNT];  // qfrc_smooth, then qacc_smooth
  V accs_l[kNS];
};

// The engine.  MP::kM is the constexpr TreeModel; Ctx supplies the storage that differs
// between the device kernel and the host harness:
//   V LC(int idx)                     per-limb constant of this lane's limb
//   E GeoGet(int slot) / GeoPut(int slot, E) / GeoPutLimb(int off, V)   the env's geoms
// (see mujoco_humanoid.hip and tests/cpu_harness/humanoid4_host.cpp).
template <class MP, class Ctx>
struct Hum4 {
  using V = typename Ctx::V;
  using E = typename EnvOf<V>::type;
  using BV = typename LaneOps<V>::B;
  using MV = typename MaskOf<V>::type;
  static constexpr TreeModel kM = MP::kM;
  static_assert(CheckTopology(MP::kM), "mj_hum4.hip.h is laid out for gym's humanoid tree");
  static constexpr int NG = 18;
  // geoms in the env's block: slot 6 g .. 6 g + 2 centre, 6 g + 3 .. 6 g + 5 unit axis
  static EPA_HD constexpr int GeoSlot(int g) { return 6 * g; }
  static constexpr int kGeoSlots = 6 * NG;

  static EPA_HD Vec3<V> LC3(Ctx& c, int idx) { return {c.LC(idx), c.LC(idx + 1), c.LC(idx + 2)}; }
  static EPA_HD BV IsLeg(Ctx& c) { return c.LC(kLcIsLeg) > V(0.5); }
  static EPA_HD Vec3<V> LiftV(Vec3<E> a) { return {V(a.x), V(a.y), V(a.z)}; }
  static EPA_HD Sp6<V> LiftS(const Sp6<E>& a) { return {LiftV(a.a), LiftV(a.l)}; }
  static EPA_HD Vec3<V> SelV(BV b, Vec3<V> x, Vec3<V> y) {
    return {Sel(b, x.x, y.x), Sel(b, x.y, y.y), Sel(b, x.z, y.z)};
  }
  static EPA_HD Sp6<V> SelS(BV b, const Sp6<V>& x, const Sp6<V>& y) { return {SelV(b, x.a, y.a), SelV(b, x.l, y.l)}; }
  static EPA_HD Vec3<E> SumQ3(Vec3<V> a) { return {SumQ(a.x), SumQ(a.y), SumQ(a.z)}; }
  static EPA_HD Sp6<E> SumQ6(const Sp6<V>& a) { return {SumQ3(a.a), SumQ3(a.l)}; }
  template <typename X>
  static EPA_HD void PutGeo(Ctx& c, int g, Vec3<X> pos) {
    c.GeoPut(GeoSlot(g), pos.x);
    c.GeoPut(GeoSlot(g) + 1, pos.y);
    c.GeoPut(GeoSlot(g) + 2, pos.z);
  }

  // trunk cdof and the trunk block of the L'DL factor are env-level data kept by the Ctx (LDS):
  // replicated in every lane's registers they cost 200 VGPRs through the whole constraint stage
  static EPA_HD void TcdPut(Ctx& c, int j, const Sp6<E>& x) {
    c.TcdPut(6 * j, x.a.x);
    c.TcdPut(6 * j + 1, x.a.y);
    c.TcdPut(6 * j + 2, x.a.z);
    c.TcdPut(6 * j + 3, x.l.x);
    c.TcdPut(6 * j + 4, x.l.y);
    c.TcdPut(6 * j + 5, x.l.z);
  }
  static EPA_HD Sp6<E> Tcd(Ctx& c, int j) {
    return {{c.TcdGet(6 * j), c.TcdGet(6 * j + 1), c.TcdGet(6 * j + 2)},
            {c.TcdGet(6 * j + 3), c.TcdGet(6 * j + 4), c.TcdGet(6 * j + 5)}};
  }

  // ---- mj_kinematics + mj_comPos + mj_crb + mj_factorM ------------------------------------------
  // qt: trunk qpos (7 + 3), ql: limb qpos (slot order; the dummy slot holds 0)
  static EPA_HD void Position(Ctx& c, const E* qt, const V* ql, Fwd<V>& f) {
    constexpr TreeModel m = MP::kM;
    // trunk: bodies 1, 2, 3 (env level), like Tree::KinBody
    Frame<E> tf[kNTB];
    Vec3<E> t_anchor[3], t_axis[3];  // joints 1 2 3 (dofs 6 7 8)
    Vec3<E> t_xipos[kNTB];
    {
      Frame<E>& f1 = tf[0];
      f1.pos = {qt[0], qt[1], qt[2]};
      f1.q = QNormalize(Quat<E>{qt[3], qt[4], qt[5], qt[6]});
      f1.R = QMat(f1.q);
      static_for<2, 4>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        const Frame<E>& par = tf[B - 2];
        Frame<E>& fb = tf[B - 1];
        fb.pos = par.pos + Mul(par.R, Vec3<E>{m.body_pos[B][0], m.body_pos[B][1], m.body_pos[B][2]});
        fb.q = QMul(par.q, Quat<E>{m.body_quat[B][0], m.body_quat[B][1], m.body_quat[B][2], m.body_quat[B][3]});
        constexpr int ja = m.body_jntadr[B], jn = m.body_jntnum[B];
        static_for<0, jn>([&](auto jc) {
          constexpr int j = ja + decltype(jc)::value;
          fb.R = QMat(fb.q);
          const Vec3<E> jp = {m.jnt_pos[j][0], m.jnt_pos[j][1], m.jnt_pos[j][2]};
          const Vec3<E> anchor = Mul(fb.R, jp) + fb.pos;
          t_anchor[j - 1] = anchor;
          t_axis[j - 1] = Mul(fb.R, Vec3<E>{m.jnt_axis[j][0], m.jnt_axis[j][1], m.jnt_axis[j][2]});
          E sn, cs;
          SinCos(E(0.5) * (qt[m.jnt_qadr[j]] - E(m.qpos0[m.jnt_qadr[j]])), &sn, &cs);
          fb.q = QMul(fb.q, Quat<E>{cs, E(m.jnt_axis[j][0]) * sn, E(m.jnt_axis[j][1]) * sn, E(m.jnt_axis[j][2]) * sn});
          fb.R = QMat(fb.q);
          fb.pos = anchor - Mul(fb.R, jp);
        });
        fb.q = QNormalize(fb.q);
        fb.R = QMat(fb.q);
      });
      static_for<0, kNTB>([&](auto bc) {
        constexpr int b = decltype(bc)::value + 1;
        t_xipos[b - 1] = tf[b - 1].pos + Mul(tf[b - 1].R, Vec3<E>{m.body_ipos[b][0], m.body_ipos[b][1], m.body_ipos[b][2]});
      });
      // trunk geoms 1..5
      static_for<1, 6>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int b = m.geom_body[g];
        PutGeo(c, g, tf[b - 1].pos + Mul(tf[b - 1].R, Vec3<E>{m.geom_pos[g][0], m.geom_pos[g][1], m.geom_pos[g][2]}));
        if constexpr (m.geom_type[g] == tree::kGeomCapsule) {
          const Vec3<E> ga = Mul(tf[b - 1].R, Vec3<E>{m.geom_axis[g][0], m.geom_axis[g][1], m.geom_axis[g][2]});
          c.GeoPut(GeoSlot(g) + 3, ga.x);
          c.GeoPut(GeoSlot(g) + 4, ga.y);
          c.GeoPut(GeoSlot(g) + 5, ga.z);
        }
      });
    }
    // limb: A hangs off the pelvis (legs) or the torso (arms)
    const BV leg = IsLeg(c);
    Frame<V> fa;
    Vec3<V> l_anchor[kNS], l_axis[kNS];
    Vec3<V> l_xipos[3];
    Mat3<V> RA, RB;
    {
      const Frame<E>&p3 = tf[2], &p1 = tf[0];
      Frame<V> par;
      par.pos = SelV(leg, LiftV(p3.pos), LiftV(p1.pos));
      par.q = {Sel(leg, V(p3.q.w), V(p1.q.w)), Sel(leg, V(p3.q.x), V(p1.q.x)), Sel(leg, V(p3.q.y), V(p1.q.y)),
               Sel(leg, V(p3.q.z), V(p1.q.z))};
      static_for<0, 9>([&](auto kc) { par.R.m[decltype(kc)::value] = Sel(leg, V(p3.R.m[decltype(kc)::value]), V(p1.R.m[decltype(kc)::value])); });
      fa.pos = par.pos + Mul(par.R, LC3(c, kLcApos));
      fa.q = par.q;  // limb bodies carry no frame rotation of their own (CheckTopology)
      static_for<0, 3>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        fa.R = QMat(fa.q);
        const Vec3<V> jp = LC3(c, kLcJpos + 3 * s), ja = LC3(c, kLcJaxis + 3 * s);
        const Vec3<V> anchor = Mul(fa.R, jp) + fa.pos;
        l_anchor[s] = anchor;
        l_axis[s] = Mul(fa.R, ja);
        V sn, cs;
        SinCos(V(0.5) * ql[s], &sn, &cs);  // qpos0 = 0 for every hinge
        fa.q = QMul(fa.q, Quat<V>{cs, ja.x * sn, ja.y * sn, ja.z * sn});
        fa.R = QMat(fa.q);
        fa.pos = anchor - Mul(fa.R, jp);
      });
      fa.q = QNormalize(fa.q);
      fa.R = QMat(fa.q);
      RA = fa.R;
      Frame<V> fb;
      fb.pos = fa.pos + Mul(fa.R, LC3(c, kLcBpos));
      fb.q = fa.q;
      {
        fb.R = QMat(fb.q);
        const Vec3<V> jp = LC3(c, kLcJpos + 9), ja = LC3(c, kLcJaxis + 9);
        const Vec3<V> anchor = Mul(fb.R, jp) + fb.pos;
        l_anchor[3] = anchor;
        l_axis[3] = Mul(fb.R, ja);
        V sn, cs;
        SinCos(V(0.5) * ql[3], &sn, &cs);
        fb.q = QMul(fb.q, Quat<V>{cs, ja.x * sn, ja.y * sn, ja.z * sn});
        fb.R = QMat(fb.q);
        fb.pos = anchor - Mul(fb.R, jp);
      }
      fb.q = QNormalize(fb.q);
      fb.R = QMat(fb.q);
      RB = fb.R;
      // C (the foot) is welded to B: same orientation (a second normalisation of an already
      // normalised quaternion in MuJoCo: identical up to the last bit)
      const Vec3<V> cpos = fb.pos + Mul(fb.R, LC3(c, kLcCpos));
      l_xipos[0] = fa.pos + Mul(fa.R, LC3(c, kLcAipos));
      l_xipos[1] = fb.pos + Mul(fb.R, LC3(c, kLcBipos));
      l_xipos[2] = cpos + Mul(fb.R, LC3(c, kLcCipos));
      // geoms of the limb: capsule on A, capsule on B, sphere
      c.GeoPutLimb(0, fa.pos + Mul(fa.R, LC3(c, kLcG0pos)), Mul(fa.R, LC3(c, kLcG0ax)));
      c.GeoPutLimb(1, fb.pos + Mul(fb.R, LC3(c, kLcG1pos)), Mul(fb.R, LC3(c, kLcG1ax)));
      c.GeoPutLimb(2, fb.pos + Mul(fb.R, LC3(c, kLcG2off)), Vec3<V>{V(0), V(0), V(1)});
    }
    // mj_comPos: system COM = origin of the c-frame
    {
      Vec3<E> ms = {E(0), E(0), E(0)};
      static_for<0, kNTB>([&](auto bc) { ms = ms + t_xipos[decltype(bc)::value] * E(m.body_mass[decltype(bc)::value + 1]); });
      const Vec3<V> ml = l_xipos[0] * c.LC(kLcAmass) + l_xipos[1] * c.LC(kLcBmass) + l_xipos[2] * c.LC(kLcCmass);
      f.com = (ms + SumQ3(ml)) * E(1.0 / m.total_mass);
    }
    const Vec3<V> comv = LiftV(f.com);
    // cdof
    static_for<0, 3>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      Vec3<E> e = {E(k == 0), E(k == 1), E(k == 2)};
      TcdPut(c, k, Sp6<E>{{E(0), E(0), E(0)}, e});
      const Vec3<E> axis = {tf[0].R.m[k], tf[0].R.m[3 + k], tf[0].R.m[6 + k]};
      TcdPut(c, 3 + k, Sp6<E>{axis, Cross(axis, f.com - tf[0].pos)});
      TcdPut(c, 6 + k, Sp6<E>{t_axis[k], Cross(t_axis[k], f.com - t_anchor[k])});
    });
    static_for<0, kNS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      f.lcd[s] = {l_axis[s], Cross(l_axis[s], comv - l_anchor[s])};
    });
    // cinert
    static_for<0, kNTB>([&](auto bc) {
      constexpr int b = decltype(bc)::value + 1;
      const E I[6] = {m.body_inertia[b][0], m.body_inertia[b][1], m.body_inertia[b][2],
                      m.body_inertia[b][3], m.body_inertia[b][4], m.body_inertia[b][5]};
      f.tci[b - 1] = CinertOf<E>(I, E(m.body_mass[b]), tf[b - 1].R, t_xipos[b - 1] - f.com);
    });
    {
      V I[6];
      static_for<0, 6>([&](auto kc) { I[decltype(kc)::value] = c.LC(kLcAin + decltype(kc)::value); });
      f.lci[0] = CinertOf<V>(I, c.LC(kLcAmass), RA, l_xipos[0] - comv);
      static_for<0, 6>([&](auto kc) { I[decltype(kc)::value] = c.LC(kLcBin + decltype(kc)::value); });
      f.lci[1] = CinertOf<V>(I, c.LC(kLcBmass), RB, l_xipos[1] - comv);
      static_for<0, 6>([&](auto kc) { I[decltype(kc)::value] = c.LC(kLcCin + decltype(kc)::value); });
      f.lci[2] = CinertOf<V>(I, c.LC(kLcCmass), RB, l_xipos[2] - comv);
    }
  }

  // ---- mj_crb + mj_factorM (after the velocity pass: cinert dies here, M is born here) --------
  static EPA_HD void MassFactor(Ctx& c, Fwd<V>& f) {
    constexpr TreeModel m = MP::kM;
    const BV leg = IsLeg(c);
    // mj_crb: composite inertias, then M
    const In10<V> cB = AddIn(f.lci[1], f.lci[2]), cA = AddIn(f.lci[0], cB);
    In10<E> legs, arms;
    static_for<0, 10>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      legs.v[k] = SumQ(Sel(leg, cA.v[k], V(0)));
      arms.v[k] = SumQ(Sel(leg, V(0), cA.v[k]));
    });
    In10<E> tcrb[kNTB];
    tcrb[2] = AddIn(f.tci[2], legs);
    tcrb[1] = AddIn(f.tci[1], tcrb[2]);
    tcrb[0] = AddIn(AddIn(f.tci[0], tcrb[1]), arms);
    E Mtt[kNTT];
    V Mlt[kNS][kNT], Mll[kNLL];
    static_for<0, kNT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const Sp6<E> buf = MulInert(tcrb[TrunkBodyOfDof(i)], Tcd(c, i));
      static_for<0, i + 1>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        E v = Dot(Tcd(c, j), buf);
        if constexpr (i == j) v = v + E(m.dof_arm[i]);
        Mtt[TT(i, j)] = v;
      });
    });
    static_for<0, kNS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const Sp6<V> buf = MulInert(s < 3 ? cA : cB, f.lcd[s]);
      static_for<0, s + 1>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        V v = Dot(f.lcd[t], buf);
        if constexpr (s == t) v = v + c.LC(kLcArm + s);
        Mll[LL(s, t)] = v;
      });
      static_for<0, kNT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const V v = Dot(LiftS(Tcd(c, j)), buf);
        // dofs 6 7 8 (abdomen) are ancestors of the legs only
        if constexpr (j >= 6) Mlt[s][j] = Sel(leg, v, V(0));
        else Mlt[s][j] = v;
      });
    });
    // mj_factorM: M = L' D L from the last dof up.  Limb dofs are leaves of the arrow: they are
    // eliminated inside their lane; their updates of the trunk block are summed over the quad.
    V dl[kNS];  // D of the limb dofs
    static_for_down<kNS, 0>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      dl[k] = Mll[LL(k, k)];
      const V inv = V(1) / Mll[LL(k, k)];
      f.dinv_l[k] = inv;
      // ancestors of k: limb slots < k, then the trunk dofs (in decreasing dof order: limb slots
      // first -- they have the larger dof numbers)
      static_for_down<k, 0>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const V tmp = Mll[LL(k, i)] * inv;
        static_for<0, i + 1>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          Mll[LL(i, j)] -= tmp * Mll[LL(k, j)];
        });
        static_for<0, kNT>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          Mlt[i][j] -= tmp * Mlt[k][j];
        });
        f.Lll[LL(k, i)] = tmp;
      });
      static_for<0, kNT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        f.Llt[k][i] = Mlt[k][i] * inv;
      });
    });
    // the limbs' update of the trunk block, sum_k L_ki L_kj D_k, summed over the quad
    static_for<0, kNT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      static_for<0, i + 1>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        V v = V(0);
        static_for<0, kNS>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          v += f.Llt[k][i] * (f.Llt[k][j] * dl[k]);
        });
        Mtt[TT(i, j)] -= SumQ(v);
      });
    });
    static_for_down<kNT, 0>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const E inv = E(1) / Mtt[TT(k, k)];
      c.DtPut(k, inv);
      static_for_down<k, 0>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const E tmp = Mtt[TT(k, i)] * inv;
        static_for<0, i + 1>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          Mtt[TT(i, j)] -= tmp * Mtt[TT(k, j)];
        });
        c.LttPut(TT(k, i), tmp);
      });
    });
  }

  // y = L^-T x (in place): the first half of mj_solveM.  Returns sum_i y_i^2 / D_i.
  // MP::kRowCache: the trunk's cdofs and the trunk block of the factor of M in registers for the duration of the
  // row loop (108 numbers).  Read through from LDS they are ~40 exposed round trips per constraint row; held in
  // registers they only fit where the loop has the register file to itself (HumanoidStandup: behind the call).
  struct TrunkRegs {
    Sp6<E> cd[kNT];
    E ltt[kNTT], dt[kNT];
  };
  static EPA_HD void LoadTrunkRegs(Ctx& c, TrunkRegs& t) {
    static_for<0, kNT>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      t.cd[j] = Tcd(c, j);
      t.dt[j] = c.DtGet(j);
    });
    static_for<0, kNTT>([&](auto ic) { t.ltt[decltype(ic)::value] = c.LttGet(decltype(ic)::value); });
  }
  static EPA_HD E HalfSolve(Ctx& c, const Fwd<V>& f, E* xt, V* xl) {
    TrunkRegs none;
    return HalfSolveT<false>(c, f, xt, xl, none);
  }
  template <bool kPre>
  static EPA_HD E HalfSolveT(Ctx& c, const Fwd<V>& f, E* xt, V* xl, const TrunkRegs& tr) {
    V ct[kNT];
    static_for<0, kNT>([&](auto jc) { ct[decltype(jc)::value] = V(0); });
    static_for_down<kNS, 0>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      static_for<0, k>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        xl[i] -= f.Lll[LL(k, i)] * xl[k];
      });
      static_for<0, kNT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        ct[j] += f.Llt[k][j] * xl[k];
      });
    });
    static_for<0, kNT>([&](auto jc) { xt[decltype(jc)::value] -= SumQ(ct[decltype(jc)::value]); });
    static_for_down<kNT, 0>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      static_for<0, k>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (kPre) xt[i] -= tr.ltt[TT(k, i)] * xt[k];
        else xt[i] -= c.LttGet(TT(k, i)) * xt[k];
      });
    });
    V ql = V(0);
    static_for<0, kNS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      ql += xl[s] * xl[s] * f.dinv_l[s];
    });
    E qt = E(0);
    static_for<0, kNT>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (kPre) qt += xt[j] * xt[j] * tr.dt[j];
      else qt += xt[j] * xt[j] * c.DtGet(j);
    });
    return qt + SumQ(ql);
  }
  // x <- L^-1 D^-1 x: the second half
  static EPA_HD void BackSolve(Ctx& c, const Fwd<V>& f, E* xt, V* xl) {
    static_for<0, kNT>([&](auto jc) { xt[decltype(jc)::value] *= c.DtGet(decltype(jc)::value); });
    static_for<0, kNS>([&](auto sc) { xl[decltype(sc)::value] *= f.dinv_l[decltype(sc)::value]; });
    ForwardSub(c, f, xt, xl);
  }
  // x <- L^-1 x
  static EPA_HD void ForwardSub(Ctx& c, const Fwd<V>& f, E* xt, V* xl) {
    static_for<0, kNT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      static_for<0, i>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        xt[i] -= c.LttGet(TT(i, j)) * xt[j];
      });
    });
    static_for<0, kNS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      static_for<0, kNT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        xl[s] -= f.Llt[s][j] * V(xt[j]);
      });
      static_for<0, s>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        xl[s] -= f.Lll[LL(s, t)] * xl[t];
      });
    });
  }

  // ---- mj_fwdVelocity + mj_fwdActuation + mj_fwdAcceleration -> qacc_smooth -------------------
  // vt / vl: qvel; qt / ql: qpos (joint springs); ut / ul: ctrl by dof (trunk dofs 6 7 8; limb slots)
  static EPA_HD void Velocity(Ctx& c, const E* qt, const V* ql, const E* vt, const V* vl, const E* ut,
                              const V* ul, Fwd<V>& f) {
    constexpr TreeModel m = MP::kM;
    const BV leg = IsLeg(c);
    // down the tree: cvel and the bias acceleration cacc
    Sp6<E> tca[kNTB];
    {
      Sp6<E> cvel = {{E(0), E(0), E(0)}, {E(0), E(0), E(0)}};
      Sp6<E> cacc = {{E(0), E(0), E(0)}, {E(0), E(0), E(m.gravity)}};  // cacc[world] = -gravity
      // free joint: translations first (their cdof_dot is zero), rotations all with the velocity
      // before the rotations (Tree::VelBody)
      cvel.l = cvel.l + Vec3<E>{vt[0], vt[1], vt[2]};
      static_for<0, 3>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        AxpySp(cacc, CrossMotion(cvel, Tcd(c, 3 + k)), vt[3 + k]);
      });
      static_for<0, 3>([&](auto kc) { AxpySp(cvel, Tcd(c, 3 + decltype(kc)::value), vt[3 + decltype(kc)::value]); });
      f.tcv[0] = cvel;
      tca[0] = cacc;
      static_for<6, 8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        AxpySp(cacc, CrossMotion(cvel, Tcd(c, i)), vt[i]);
        AxpySp(cvel, Tcd(c, i), vt[i]);
      });
      f.tcv[1] = cvel;
      tca[1] = cacc;
      AxpySp(cacc, CrossMotion(cvel, Tcd(c, 8)), vt[8]);
      AxpySp(cvel, Tcd(c, 8), vt[8]);
      f.tcv[2] = cvel;
      tca[2] = cacc;
    }
    Sp6<V> lca[2];
    {
      Sp6<V> cvel = SelS(leg, LiftS(f.tcv[2]), LiftS(f.tcv[0]));
      Sp6<V> cacc = SelS(leg, LiftS(tca[2]), LiftS(tca[0]));
      static_for<0, 3>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        AxpySp(cacc, CrossMotion(cvel, f.lcd[s]), vl[s]);
        AxpySp(cvel, f.lcd[s], vl[s]);
      });
      f.lcv[0] = cvel;
      lca[0] = cacc;
      AxpySp(cacc, CrossMotion(cvel, f.lcd[3]), vl[3]);
      AxpySp(cvel, f.lcd[3], vl[3]);
      f.lcv[1] = cvel;
      lca[1] = cacc;
    }
    // cfrc = I cacc + cvel x* (I cvel), summed up the tree
    auto frc = [](const auto& I, const auto& cv, const auto& ca) {
      const auto t1 = MulInert(I, ca);
      const auto t3 = CrossForce(cv, MulInert(I, cv));
      return AddSp(t1, t3);
    };
    const Sp6<V> fC = frc(f.lci[2], f.lcv[1], lca[1]);
    const Sp6<V> fB = AddSp(frc(f.lci[1], f.lcv[1], lca[1]), fC);
    const Sp6<V> fA = AddSp(frc(f.lci[0], f.lcv[0], lca[0]), fB);
    const Sp6<V> zero = {{V(0), V(0), V(0)}, {V(0), V(0), V(0)}};
    const Sp6<E> flegs = SumQ6(SelS(leg, fA, zero)), farms = SumQ6(SelS(leg, zero, fA));
    Sp6<E> tf[kNTB];
    tf[2] = AddSp(frc(f.tci[2], f.tcv[2], tca[2]), flegs);
    tf[1] = AddSp(frc(f.tci[1], f.tcv[1], tca[1]), tf[2]);
    tf[0] = AddSp(AddSp(frc(f.tci[0], f.tcv[0], tca[0]), tf[1]), farms);
    // qfrc_smooth = -bias + passive + actuator
    E xt[kNT];
    V xl[kNS];
    static_for<0, kNT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      E x = -Dot(Tcd(c, i), tf[TrunkBodyOfDof(i)]) - E(m.dof_damp[i]) * vt[i];
      E act = E(0);
      if constexpr (i >= 6) {
        constexpr int j = i - 5;  // joint of the dof
        if constexpr (m.jnt_stiff[j] != 0.0) x -= E(m.jnt_stiff[j]) * (qt[m.jnt_qadr[j]] - E(m.qpos0[m.jnt_qadr[j]]));
        constexpr double gear = GearOfDof(i);
        act = E(gear) * ClampX(ut[i - 6], E(m.ctrl_lo), E(m.ctrl_hi));
      }
      f.act_t[i] = act;
      xt[i] = x + act;
    });
    static_for<0, kNS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      V x = -Dot(f.lcd[s], s < 3 ? fA : fB) - c.LC(kLcDamp + s) * vl[s];
      x -= c.LC(kLcStiff + s) * ql[s];
      const V act = c.LC(kLcGear + s) * ClampX(ul[s], V(m.ctrl_lo), V(m.ctrl_hi));
      f.act_l[s] = act;
      xl[s] = x + act;
    });
    static_for<0, kNT>([&](auto ic) { f.accs_t[decltype(ic)::value] = xt[decltype(ic)::value]; });
    static_for<0, kNS>([&](auto sc) { f.accs_l[decltype(sc)::value] = xl[decltype(sc)::value]; });
  }
  // qacc_smooth = M^-1 qfrc_smooth (in place in f.accs)
  static EPA_HD void SmoothAcc(Ctx& c, Fwd<V>& f) {
    HalfSolve(c, f, f.accs_t, f.accs_l);
    BackSolve(c, f, f.accs_t, f.accs_l);
  }
  // ---- static candidate tables (indexed at run time by an env- or lane-level group number) -----
  // groups: limits 0..16 (dof 6 + g), floor spheres 17..45, geom pairs 46..154 (TreeModel order)
  static constexpr int kNLimit = 17, kNFloor = 29, kNPair = MP::kM.npair;
  static constexpr int kG0Floor = kNLimit, kG0Pair = kNLimit + kNFloor, kNGroup = kG0Pair + kNPair;
  static_assert(kNGroup <= 192, "three 64-bit mask words");
  struct Tabs {
    int pair_g1[128], pair_g2[128];
    double pair_bound[128], pair_diag[128];
    int geom_cap[NG], geom_body[NG];
    double geom_rad[NG], geom_hl[NG];
    unsigned body_mask[16];
    double body_invw[16];
    int floor_geom[32];
    double floor_sign[32];
    double lim_lo[kNLimit], lim_hi[kNLimit], lim_invw[kNLimit];  // limit group g: joint g + 1, dof 6 + g
  };
  static constexpr Tabs MakeTabs() {
    constexpr TreeModel m = MP::kM;
    Tabs t{};
    for (int g = 0; g < NG; ++g) {
      t.geom_cap[g] = m.geom_type[g] == tree::kGeomCapsule ? 1 : 0;
      t.geom_body[g] = m.geom_body[g];
      t.geom_rad[g] = m.geom_rad[g];
      t.geom_hl[g] = m.geom_hl[g];
    }
    for (int b = 0; b < m.nbody; ++b) {
      t.body_mask[b] = m.body_dofmask[b];
      t.body_invw[b] = m.body_invw[b];
    }
    for (int c = 0; c < m.nfloor; ++c) {
      t.floor_geom[c] = m.floor_geom[c];
      t.floor_sign[c] = m.floor_sign[c];
    }
    for (int g = 0; g < kNLimit; ++g) {
      t.lim_lo[g] = m.jnt_lo[g + 1];
      t.lim_hi[g] = m.jnt_hi[g + 1];
      t.lim_invw[g] = m.dof_invw[6 + g];
    }
    for (int p = 0; p < m.npair; ++p) {
      const int g1 = m.pair_g1[p], g2 = m.pair_g2[p];
      t.pair_g1[p] = g1;
      t.pair_g2[p] = g2;
      t.pair_bound[p] = m.geom_rad[g1] + m.geom_hl[g1] + m.geom_rad[g2] + m.geom_hl[g2] + m.margin;
      t.pair_diag[p] = m.body_invw[m.geom_body[g1]] + m.body_invw[m.geom_body[g2]];
    }
    return t;
  }
  static constexpr Tabs kT = MakeTabs();  // Ctx::T(): the device keeps a copy in LDS (run-time indexed)

  struct EMask {
    unsigned long long w[3];
  };
  static EPA_HD Vec3<E> GeoPos(Ctx& c, int g) {
    return {c.GeoGet(GeoSlot(g)), c.GeoGet(GeoSlot(g) + 1), c.GeoGet(GeoSlot(g) + 2)};
  }
  static EPA_HD Vec3<E> GeoAxis(Ctx& c, int g) {
    return {c.GeoGet(GeoSlot(g) + 3), c.GeoGet(GeoSlot(g) + 4), c.GeoGet(GeoSlot(g) + 5)};
  }
  // sphere-sphere: dist; n from 1 to 2; pos midway (mjraw_SphereSphere)
  static EPA_HD E SphereSphere(Vec3<E> p1, E r1, Vec3<E> p2, E r2, Vec3<E>* n, Vec3<E>* pos) {
    const Vec3<E> dif = p2 - p1;
    const E cd = SqrtX(Dot(dif, dif));
    const bool far = cd >= E(tree::kMinVal);
    const E inv = E(1) / Sel(far, cd, E(1));
    *n = {Sel(far, dif.x * inv, E(1)), Sel(far, dif.y * inv, E(0)), Sel(far, dif.z * inv, E(0))};
    const E dist = cd - r1 - r2;
    *pos = p1 + *n * (r1 + E(0.5) * dist);
    return dist;
  }
  // bounding-sphere cull of pair p (p: any lane- or env-level index)
  static EPA_HD bool PairNear(Ctx& c, int p) {
    const Vec3<E> dc = GeoPos(c, c.T().pair_g2[p]) - GeoPos(c, c.T().pair_g1[p]);
    const E bound = c.T().pair_bound[p];
    return Dot(dc, dc) < bound * bound;
  }
  // narrow phase of pair p: sphere / capsule primitives (mjraw_SphereSphere / SphereCapsule /
  // CapsuleCapsule; pair_g1 has the lower geom TYPE: sphere before capsule).  Select-only: the
  // pair differs from lane to lane (Detect) or from env to env (MakeRows).
  static EPA_HD E PairNarrow(Ctx& c, int p, Vec3<E>* n, Vec3<E>* pos) {
    const int g1 = c.T().pair_g1[p], g2 = c.T().pair_g2[p];
    const bool cap1 = c.T().geom_cap[g1] != 0, cap2 = c.T().geom_cap[g2] != 0;
    const E r1 = c.T().geom_rad[g1], r2 = c.T().geom_rad[g2], h1 = c.T().geom_hl[g1], h2 = c.T().geom_hl[g2];
    const Vec3<E> p1 = GeoPos(c, g1), p2 = GeoPos(c, g2);
    const Vec3<E> ax1 = GeoAxis(c, g1), ax2 = GeoAxis(c, g2);
    // both capsules
    const Vec3<E> a1 = ax1 * h1, a2 = ax2 * h2;
    const Vec3<E> dif = p1 - p2;
    const E ma = Dot(a1, a1), mb = -Dot(a1, a2), mc = Dot(a2, a2);
    const E u = -Dot(a1, dif), v = Dot(a2, dif);
    const E det = ma * mc - mb * mb;
    const bool reg = AbsX(det) >= E(tree::kMinVal);
    const E idet = E(1) / Sel(reg, det, E(1));
    E x1 = (mc * u - mb * v) * idet, x2 = (ma * v - mb * u) * idet;
    {
      const bool hi1 = x1 > E(1), lo1 = x1 < E(-1);
      x2 = Sel(hi1, (v - mb) / mc, Sel(lo1, (v + mb) / mc, x2));
      x1 = Sel(hi1, E(1), Sel(lo1, E(-1), x1));
      const bool hi2 = x2 > E(1), lo2 = x2 < E(-1);
      const E y1 = ClampX(Sel(hi2, (u - mb) / ma, (u + mb) / ma), E(-1), E(1));
      x1 = Sel(hi2 || lo2, y1, x1);
      x2 = Sel(hi2, E(1), Sel(lo2, E(-1), x2));
    }
    {  // exactly parallel axes: midpoint of the overlap (see oracle/mjcpu/engine.c)
      const E amb = AbsX(mb);
      const E lo = MaxX(E(-1), (u - amb) / ma), hi = MinX(E(1), (u + amb) / ma);
      const E px1 = Sel(lo <= hi, E(0.5) * (lo + hi), Sel(lo > E(1), E(1), E(-1)));
      const E px2 = ClampX((v - mb * px1) / mc, E(-1), E(1));
      x1 = Sel(reg, x1, px1);
      x2 = Sel(reg, x2, px2);
    }
    // sphere - capsule
    const E ts = ClampX(Dot(ax2, p1 - p2), -h2, h2);
    const Vec3<E> q1 = {Sel(cap1, p1.x + a1.x * x1, p1.x), Sel(cap1, p1.y + a1.y * x1, p1.y),
                        Sel(cap1, p1.z + a1.z * x1, p1.z)};
    const Vec3<E> qc = p2 + a2 * x2, qs = p2 + ax2 * ts;
    const Vec3<E> q2 = {Sel(cap1, qc.x, Sel(cap2, qs.x, p2.x)), Sel(cap1, qc.y, Sel(cap2, qs.y, p2.y)),
                        Sel(cap1, qc.z, Sel(cap2, qs.z, p2.z))};
    return SphereSphere(q1, r1, q2, r2, n, pos);
  }

  // ---- mj_collision + joint-limit detection: the env's mask of active groups -----------------------
  static EPA_HD void Detect(Ctx& c, const E* qt, const V* ql, EMask& act) {
    constexpr TreeModel m = MP::kM;
    MV lane[3];
    lane[0] = lane[1] = lane[2] = MV{};
    unsigned long long env0 = 0ull;
    // joint limits (mj_instantiateLimit): active if q < lo or q > hi (margin 0)
    static_for<0, 3>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int j = g + 1;
      const E q = qt[7 + g];
      const bool on = (q - E(m.jnt_lo[j]) < E(0)) || (E(m.jnt_hi[j]) - q < E(0));
      env0 |= (on ? 1ull : 0ull) << g;
    });
    static_for<0, kNS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int bits[4] = {LimbDof(0, s) - 6, LimbDof(1, s) - 6, LimbDof(2, s) < 0 ? 63 : LimbDof(2, s) - 6,
                               LimbDof(3, s) < 0 ? 63 : LimbDof(3, s) - 6};
      const BV on = (ql[s] - c.LC(kLcLo + s) < V(0)) | (c.LC(kLcHi + s) - ql[s] < V(0));
      SetBitLane(lane[0], on, bits);
    });
    // floor (plane z = 0): mjc_PlaneSphere on spheres and capsule end spheres
    static_for<0, 9>([&](auto cc) {
      constexpr int cand = decltype(cc)::value;
      constexpr int g = m.floor_geom[cand];
      E cz = c.GeoGet(GeoSlot(g) + 2);
      if constexpr (m.geom_type[g] == tree::kGeomCapsule) cz += E(m.floor_sign[cand] * m.geom_hl[g]) * c.GeoGet(GeoSlot(g) + 5);
      const bool on = cz - E(m.geom_rad[g]) < E(m.margin);
      env0 |= (on ? 1ull : 0ull) << (kG0Floor + cand);
    });
    static_for<0, 5>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int which = k / 2;
      constexpr int bits[4] = {kG0Floor + 9 + k, kG0Floor + 14 + k, kG0Floor + 19 + k, kG0Floor + 24 + k};
      Vec3<V> gp, ga;
      c.GeoGetLimb(which, &gp, &ga);
      V cz = gp.z;
      if constexpr (which == 0) cz += V(k == 0 ? 1.0 : -1.0) * c.LC(kLcG0hl) * ga.z;
      if constexpr (which == 1) cz += V(k == 2 ? 1.0 : -1.0) * c.LC(kLcG1hl) * ga.z;
      const V rad = c.LC(which == 0 ? kLcG0rad : (which == 1 ? kLcG1rad : kLcG2rad));
      const BV on = cz - rad < V(m.margin);
      SetBitLane(lane[0], on, bits);
    });
    // geom pairs: lane l tests pairs l, l + 4, ...
    LaneOps<V>::PerLane(lane, [&](int l, unsigned long long* w) {
      for (int i = 0; 4 * i < kNPair + 3; ++i) {
        const int pp = 4 * i + l;
        const bool valid = pp < kNPair;
        const int p = valid ? pp : 0;
        const bool near = valid && PairNear(c, p);
        if (AnyWave(near)) {
          Vec3<E> n, pos;
          const E dist = PairNarrow(c, p, &n, &pos);
          const bool on = near && dist < E(m.margin);
          const int g = kG0Pair + p;
          const unsigned long long bit = (on ? 1ull : 0ull) << (g & 63);
          w[0] |= (g >> 6) == 0 ? bit : 0ull;
          w[1] |= (g >> 6) == 1 ? bit : 0ull;
          w[2] |= (g >> 6) == 2 ? bit : 0ull;
        }
      }
    });
    act.w[0] = OrQ(lane[0]) | env0;
    act.w[1] = OrQ(lane[1]);
    act.w[2] = OrQ(lane[2]);
  }

  static EPA_HD E Impedance(E x_abs) {  // solimp (d0, dmax, width, 0.5, 2)
    constexpr TreeModel m = MP::kM;
    const E x = x_abs * E(1.0 / m.sol_width);
    const E y = Sel(x <= E(0.5), E(2) * x * x, E(1) - E(2) * (E(1) - x) * (E(1) - x));
    return Sel(x >= E(1), E(m.sol_dmax), E(m.sol_d0) + y * E(m.sol_dmax - m.sol_d0));
  }

  // ---- mj_makeConstraint + the y rows ------------------------------------------------------------
  // A 23-vector in DISTRIBUTED form: kND = 7 numbers per lane -- trunk entries l, 4 + l and (lane 0)
  // 8, then the lane's four limb entries -- so that a dot product is 7 multiply-adds and ONE quad
  // reduction, with no trunk entry touched twice.  The rows y_r, z and 1 / D are kept like this.
  static constexpr int kND = 3 + kNS;
  static EPA_HD void Distribute(const E* xt, const V* xl, V* xd) {
    E pad[12];
    static_for<0, kNT>([&](auto jc) { pad[decltype(jc)::value] = xt[decltype(jc)::value]; });
    pad[9] = pad[10] = pad[11] = E(0);
    xd[0] = LanePickV(pad, 0);
    xd[1] = LanePickV(pad, 4);
    xd[2] = LanePickV(pad, 8);
    static_for<0, kNS>([&](auto sc) { xd[3 + decltype(sc)::value] = xl[decltype(sc)::value]; });
  }
  static EPA_HD V LanePickV(const E* x, int base) {
    if constexpr (std::is_same<V, E>::value) return LanePick4(x, base);
    else return LanePick4(x, base, V());
  }
  static EPA_HD void Gather(const V* xd, E* xt, V* xl) {
    static_for<0, kNT>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      xt[j] = BcastQ(xd[j >> 2], j & 3);
    });
    static_for<0, kNS>([&](auto sc) { xl[decltype(sc)::value] = xd[3 + decltype(sc)::value]; });
  }
  static EPA_HD E DotD(const V* a, const V* b) {  // three short chains: this sits on the PGS critical path
    V p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2];
    p0 += a[3] * b[3];
    p1 += a[4] * b[4];
    p2 += a[5] * b[5];
    p0 += a[6] * b[6];
    return SumQ((p0 + p1) + p2);
  }
  // 1 / D in distributed form
  static EPA_HD void DinvD(Ctx& c, const Fwd<V>& f, V* dd) { DinvD(c, f.dinv_l, dd); }
  static EPA_HD void DinvD(Ctx& c, const V* dinv_l, V* dd) {
    E dt[kNT];
    static_for<0, kNT>([&](auto jc) { dt[decltype(jc)::value] = c.DtGet(decltype(jc)::value); });
    Distribute(dt, dinv_l, dd);
  }
  // Row storage (Ctx): RowPut / RowGet(r, yd) the row's y = L^-T J' (distributed); RsPut / RsGet(r, k) its
  // scalars k = 0 f, 1 A_rr + R_r, 2 R_r, 3 b_r = J_r qacc_smooth - aref_r, 4 1 / (A_rr + R_r);
  // RecPut / RecGet(t, k) the contact records (position 3, normal 3, bodies 2) in compact order.
  enum { kRsF = 0, kRsArr = 1, kRsR = 2, kRsB = 3, kRsAinv = 4 };
  struct RowCount {
    int nl, nf, np;  // THIS ENV's limit rows, floor contacts (4 rows each), pair rows (env level)
    EPA_HD int rows() const { return nl + 4 * nf + np; }
  };
  static EPA_HD unsigned long long RangeBits(int wi, int lo, int hi) {
    const int a = lo - 64 * wi, b = hi - 64 * wi;
    if (b <= 0 || a >= 64) return 0ull;
    const unsigned long long upto = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
    const unsigned long long from = a <= 0 ? ~0ull : (~0ull << a);
    return upto & from;
  }
  // vt / vl: qvel, wt / wl: qacc_warmstart.  Leaves zs = sum_r f_r y_r (f: the warm-start forces)
  // and cost = sum_r f_r (R_r f_r / 2 + b_r).
  static EPA_HD RowCount MakeRows(Ctx& c, const Fwd<V>& f, const EMask& act, const E* qt, const V* ql,
                                  const E* vt, const V* vl, const E* wt, const V* wl, V* zsd, E* cost_out) {
    constexpr TreeModel m = MP::kM;
    // Rows and contact records are compact PER ENV: env e's t-th active group goes to its own
    // next slot, so a wave's solve works on max-over-envs rows, not on a per-phase padded union.
    RowCount rc{0, 0, 0};
    E cost = E(0);
    static_for<0, kND>([&](auto ic) { zsd[decltype(ic)::value] = V(0); });
    TrunkRegs tr;
    if constexpr (MP::kRowCache != 0) LoadTrunkRegs(c, tr);  // (1 all of it, 2 the factor only, 3 the cdofs only)
    int row = 0;
    for (int phase = 0; phase < 3; ++phase) {
      const int glo = phase == 0 ? 0 : (phase == 1 ? kG0Floor : kG0Pair);
      const int ghi = phase == 0 ? kNLimit : (phase == 1 ? kG0Pair : kNGroup);
      const int nsub = phase == 1 ? 4 : 1;
      for (int wi = 0; wi < 3; ++wi) {
        const unsigned long long bits = RangeBits(wi, glo, ghi);
        if (bits == 0ull) continue;
        unsigned long long rem = act.w[wi] & bits;
        while (AnyWave(rem != 0ull)) {
          const bool on = rem != 0ull;
          const int g = on ? 64 * wi + __builtin_ctzll(rem) : glo;  // this env's group (glo: inert dummy)
          rem &= rem - 1ull;
          E pos = E(0), lim_s = E(0), diag;
          Vec3<E> cpos = {E(0), E(0), E(0)}, n = {E(0), E(0), E(1)};
          unsigned m1 = 0u, m2 = 0u;
          int b1 = 0, b2 = 0, ld = 0;
          if (phase == 0) {
            ld = 6 + g;
            const E q = ld < kNT ? qt[1 + ld] : LimbPick(ql, LimbOfDof(ld), SlotOfDof(ld));
            const E dlo = q - E(c.T().lim_lo[g]), dhi = E(c.T().lim_hi[g]) - q;
            const bool lo = dlo < E(0);
            pos = Sel(lo, dlo, dhi);
            lim_s = Sel(lo, E(1), E(-1));
            diag = E(c.T().lim_invw[g]);
          } else if (phase == 1) {
            const int cand = g - kG0Floor;
            const int gg = c.T().floor_geom[cand];
            const Vec3<E> ctr = GeoPos(c, gg) + GeoAxis(c, gg) * E(c.T().floor_sign[cand] * c.T().geom_hl[gg]);
            const E dist = ctr.z - E(c.T().geom_rad[gg]);
            pos = dist - E(m.margin);
            cpos = {ctr.x, ctr.y, E(0.5) * dist};
            b2 = c.T().geom_body[gg];
            m2 = c.T().body_mask[b2];
            diag = E(c.T().body_invw[b2] * (1.0 + m.floor_mu * m.floor_mu));
          } else {
            const int p = g - kG0Pair;
            const E dist = PairNarrow(c, p, &n, &cpos);
            pos = dist - E(m.margin);
            b1 = c.T().geom_body[c.T().pair_g1[p]];
            b2 = c.T().geom_body[c.T().pair_g2[p]];
            m1 = c.T().body_mask[b1];
            m2 = c.T().body_mask[b2];
            diag = E(c.T().pair_diag[p]);
          }
          const Vec3<E> off = cpos - f.com;
          const E imp = Impedance(AbsX(pos));
          E R = MaxX(E(tree::kMinVal), (E(1) - imp) * diag / imp);
          if (phase == 1) R = R * E(2.0 * m.floor_mu * m.floor_mu);
          const E kimp = E(m.sol_K) * imp * pos;
          if (phase != 0 && AnyWave(on)) {  // compact contact record for mj_rnePostConstraint
            const int t = rc.nf + rc.np;  // this env's next record
            c.RecPut(t, 0, cpos.x);
            c.RecPut(t, 1, cpos.y);
            c.RecPut(t, 2, cpos.z);
            c.RecPut(t, 3, n.x);
            c.RecPut(t, 4, n.y);
            c.RecPut(t, 5, n.z);
            c.RecPut(t, 6, E(b1));
            c.RecPut(t, 7, E(b2));
          }  // (an env without this group rewrites its next record later, or never reads it)
          for (int k = 0; k < nsub; ++k) {
            // row direction: n, or the pyramid edge n +- mu t (floor frame: n = z, t1 = y, t2 = -x)
            Vec3<E> dir = n;
            if (phase == 1) {
              const E mu = E(m.floor_mu);
              dir = k == 0 ? Vec3<E>{E(0), mu, E(1)} : (k == 1 ? Vec3<E>{E(0), -mu, E(1)}
                           : (k == 2 ? Vec3<E>{-mu, E(0), E(1)} : Vec3<E>{mu, E(0), E(1)}));
            }
            const Vec3<E> mdir = Cross(off, dir);
            const int r = row + k;  // env level
            E Jt[kNT];
            V Jl[kNS];
            if (phase == 0) {
              static_for<0, kNT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                Jt[j] = (on && ld == j) ? lim_s : E(0);
              });
              LimbUnit(Jl, ld < kNT ? -1 : LimbOfDof(ld), ld < kNT ? -1 : SlotOfDof(ld), on ? lim_s : E(0));
            } else {
              static_for<0, kNT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const E coef = E((int)((m2 >> j) & 1u) - (int)((m1 >> j) & 1u));
                Sp6<E> cd;
                if constexpr (MP::kRowCache == 1 || MP::kRowCache == 3) cd = tr.cd[j];
                else cd = Tcd(c, j);
                const E jc0 = coef * (Dot(dir, cd.l) + Dot(mdir, cd.a));
                Jt[j] = on ? jc0 : E(0);
              });
              const Vec3<V> dv = LiftV(dir), mv = LiftV(mdir);
              static_for<0, kNS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int dofs[4] = {LimbDof(0, s), LimbDof(1, s), LimbDof(2, s) < 0 ? 31 : LimbDof(2, s),
                                         LimbDof(3, s) < 0 ? 31 : LimbDof(3, s)};
                const V coef = BitLane(m2, dofs, V()) - BitLane(m1, dofs, V());
                const V jc0 = coef * (Dot(dv, f.lcd[s].l) + Dot(mv, f.lcd[s].a));
                Jl[s] = Sel(on, jc0, V(0));
              });
            }
            // J qvel, J qacc_warmstart, J qacc_smooth
            E dv_t = E(0), dw_t = E(0), da_t = E(0);
            V dv_l = V(0), dw_l = V(0), da_l = V(0);
            static_for<0, kNT>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              dv_t += Jt[j] * vt[j];
              dw_t += Jt[j] * wt[j];
              da_t += Jt[j] * f.accs_t[j];
            });
            static_for<0, kNS>([&](auto sc) {
              constexpr int s = decltype(sc)::value;
              dv_l += Jl[s] * vl[s];
              dw_l += Jl[s] * wl[s];
              da_l += Jl[s] * f.accs_l[s];
            });
            const E vel = dv_t + SumQ(dv_l), jw = dw_t + SumQ(dw_l), ja = da_t + SumQ(da_l);
            const E aref = -E(m.sol_B) * vel - kimp;
            const E jar = jw - aref;
            const E fw = (on && jar < E(0)) ? -jar / R : E(0);
            const E b = on ? ja - aref : E(0);
            const E Rr = on ? R : E(0);
            const E quad = HalfSolveT<MP::kRowCache == 1 || MP::kRowCache == 2>(c, f, Jt, Jl, tr);  // J -> y in place
            V yd[kND];
            Distribute(Jt, Jl, yd);
            const E arr = Rr + quad;
            if (on) {  // (the one lane-divergent branch of this stage: a handful of stores)
              c.RowPut(r, yd);
              c.RsPut(r, kRsF, fw);
              c.RsPut4(r, arr, Rr, b, E(1) / arr);  // (one store per lane: four guarded stores are four divergent branches)
            }
            cost += fw * (E(0.5) * Rr * fw + b);
            static_for<0, kND>([&](auto ic) { zsd[decltype(ic)::value] += V(fw) * yd[decltype(ic)::value]; });
          }
          row += on ? nsub : 0;
          if (phase == 0) rc.nl += on ? 1 : 0;
          if (phase == 1) rc.nf += on ? 1 : 0;
          if (phase == 2) rc.np += on ? 1 : 0;
        }
      }
    }
    *cost_out = cost;
    return rc;
  }

  // the warm start is kept only if its dual cost 1/2 f'(A+R)f + f'b is below the cost of f = 0
  static EPA_HD bool ColdStart(const V* zsd, const V* dd, E cost) {
    V q = V(0);
    static_for<0, kND>([&](auto ic) { q += zsd[decltype(ic)::value] * zsd[decltype(ic)::value] * dd[decltype(ic)::value]; });
    return cost + E(0.5) * SumQ(q) > E(0);
  }
  // qacc = qacc_smooth + L^-1 z,  z = D^-1 sum_r f_r y_r (distributed)
  static EPA_HD void Finish(Ctx& c, const Fwd<V>& f, const V* zd, E* at, V* al) {
    E zt[kNT];
    V zl[kNS];
    Gather(zd, zt, zl);
    ForwardSub(c, f, zt, zl);
    static_for<0, kNT>([&](auto jc) { at[decltype(jc)::value] = f.accs_t[decltype(jc)::value] + zt[decltype(jc)::value]; });
    static_for<0, kNS>([&](auto sc) { al[decltype(sc)::value] = f.accs_l[decltype(sc)::value] + zl[decltype(sc)::value]; });
  }
  // ---- mj_fwdConstraint with mj_solPGS, y-space streaming form ---------------------------------
  //   z = sum_c f_c y_c / D;  res_r = b_r + R_r f_r + y_r . z  (= b_r + sum_c (A + R)_rc f_c)
  // A row visit is a ~200-cycle dependent chain and an HBM round trip is ten times that, so the
  // rows' constant part (y, A_rr + R_r, R_r, b_r, 1 / (A_rr + R_r)) streams through a ring of
  // kRing register buffers, each loaded kRing visits before it is used -- across sweep
  // boundaries too -- while the forces f, the only thing a visit changes, stay in the env's
  // shared block (Ctx::ShGet / ShPut: the LDS that holds A in SolvePgsA).
  static constexpr int kRing = 8;
  // max over the wave of an env-level count < 256
  static EPA_HD int WaveMax(int x) {
    int mx = 0;
    for (int b = 7; b >= 0; --b) {
      const int cand = mx | (1 << b);
      if (AnyWave(x >= cand)) mx = cand;
    }
    return mx;
  }
  struct RowConst {
    V yd[kND];
    E arr, R, b, ainv;
  };
  static EPA_HD void LoadRowConst(Ctx& c, int r, RowConst& t) {
    c.RowGet(r, t.yd);
    c.RsGet4(r, &t.arr, &t.R, &t.b, &t.ainv);  // one load per lane + broadcasts: the sweep is HBM bound
  }
  // nrow_e: this env's rows (0 .. nrow_e - 1, compact); the wave walks max-over-envs rows, an env
  // treats the rows beyond its own as absent (its slots there hold stale data)
  static EPA_HD void SolvePgs(Ctx& c, const Fwd<V>& f, int nrow_e, const V* zsd, E cost, E* at, V* al, int max_iter,
                              int* stat) {
    constexpr TreeModel m = MP::kM;
    const int nrow = WaveMax(nrow_e);
    V dd[kND], zd[kND];
    DinvD(c, f, dd);
    const bool cold = ColdStart(zsd, dd, cost);
    static_for<0, kND>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      zd[i] = Sel(cold, V(0), zsd[i] * dd[i]);
    });
    // (rows beyond the shared block's capacity keep their force in HBM: never seen in practice)
    auto fget = [&](int r) { return r < kFSlots ? c.ShGet(r) : c.RsGet(r, kRsF); };
    auto fput = [&](int r, E v) {
      if (r < kFSlots) c.ShPut(r, v);
      else c.RsPut(r, kRsF, v);
    };
    for (int r = 0; r < nrow; ++r) fput(r, (cold || r >= nrow_e) ? E(0) : c.RsGet(r, kRsF));
    const E scale = E(1.0 / (m.meaninertia * 23.0));
    bool done = nrow_e == 0;
    E improvement = E(0);
    RowConst ring[kRing];
    int pre = 0;  // next row to prefetch
    static_for<0, kRing>([&](auto bc) {
      LoadRowConst(c, pre, ring[decltype(bc)::value]);
      pre = pre + 1 < nrow ? pre + 1 : 0;
    });
    int r = 0, iter = 0;
    bool more = nrow > 0 && max_iter > 0;
    while (more) {
      static_for<0, kRing>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if (more) {  // wave uniform
          RowConst cu = ring[b];
          LoadRowConst(c, pre, ring[b]);
          pre = pre + 1 < nrow ? pre + 1 : 0;
          const bool valid = r < nrow_e;
          static_for<0, kND>([&](auto ic) { cu.yd[decltype(ic)::value] = Sel(valid, cu.yd[decltype(ic)::value], V(0)); });
          const E fr = fget(r);
          const E res = cu.b + cu.R * fr + DotD(cu.yd, zd);
          const E fn = MaxX(E(0), fr - res * cu.ainv);
          E delta = fn - fr;
          const E change = E(0.5) * delta * delta * cu.arr + delta * res;
          const bool keep = valid && !done && !(change > E(1e-10));
          delta = keep ? delta : E(0);
          fput(r, fr + delta);
          static_for<0, kND>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            zd[i] += V(delta) * cu.yd[i] * dd[i];
          });
          improvement -= keep ? change : E(0);
          if (++r == nrow) {  // end of a sweep
            r = 0;
            stat[4] += done ? 0 : nrow_e;
            done = done || improvement * scale < E(1e-8);
            improvement = E(0);
            more = ++iter < max_iter && AnyWave(!done);
          }
        }
      });
    }
    for (int k = 0; k < nrow && k < kFSlots; ++k) {
      if (k < nrow_e) c.RsPut(k, kRsF, c.ShGet(k));  // efc_force, for mj_rnePostConstraint
    }
    Finish(c, f, zd, at, al);
  }

  // ---- mj_solPGS on the dual matrix, in registers ----------------------------------------------------
  // Up to kRegRows rows per env.  A PGS row visit is ONE dependent chain (residual -> clamped force ->
  // cost change -> accept) and a wave runs alone on its SIMD, so the chain length is the cost of a
  // visit.  Working on  z = sum f y / D  puts a 7-term dot product, a quad reduction and an LDS
  // round trip on it (~60 instructions, ~900 cycles measured).  Instead A + R = Y D^-1 Y' + diag(R)
  // is formed once -- the rows stream past four register-resident columns, the entries are staged in
  // the env's shared block (LDS) -- then lane l of the quad takes the rows r with (r & 3) == l into
  // registers (kOwn x kRegRows entries) and the residuals S_r = b_r + sum_c (A + R)_rc f_c are kept UP
  // TO DATE: a visit reads S_r, decides, broadcasts the accepted change (DPP) and every lane adds
  // (A + R)_{its rows, r} * change to its residuals.  ~15 dependent operations per visit.
  static constexpr int kFSlots = 156;  // capacity of the shared block (Ctx::ShGet / ShPut)
  // rows (per env) the register-resident PGS holds: MP::kRegRows (a multiple of 4).  Measured: Humanoid 12
  // (16: -3 %, the benchmark rarely has more); HumanoidStandup 24 (20: -15 %; 28 spills the matrix
  // itself and halves the rate) -- lying on the floor it often has more, those waves stream (SolvePgs)
  static constexpr int kRegRows = MP::kRegRows;
  static constexpr int kOwn = kRegRows / 4;
  static_assert(kRegRows % 4 == 0 && kRegRows <= 32, "register rows");
  static EPA_HD constexpr int Tri(int r, int cc) { return r >= cc ? r * (r + 1) / 2 + cc : cc * (cc + 1) / 2 + r; }
  // kOv (a wave in which some env has more than kRegRows rows, none more than kRegRows + kC): the
  // HYBRID.  Rows 0 .. kRegRows - 1 of every env stay on the register-resident dual matrix with tracked
  // residuals; the next kC rows -- the OVERFLOW rows c = kRegRows + j -- are kept on chip in a cheaper form:
  //   av_j[k] = A_{c, 4 k + l}   the row's entries against the lane's own register rows (kOwn registers),
  //   (A + R)_{c, c'}            the overflow block, packed triangle in the env's shared block (LDS),
  //   So_j = b_c + sum_j' (A + R)_{c c'} f_c'   tracked, and f_c: env-level registers,
  // and a visit forms  res_c = So_j + sum_k av_j[k] f_{4 k + l}  with kOwn multiply-adds and one quad
  // reduction; an accepted change goes through the same av_j into the tracked residuals of the register rows
  // and through one column of the overflow block into So.  Visit order = row order, as in mj_solPGS.
  // An env of such a wave without overflow rows executes the arithmetic of the plain register form plus
  // additions of exact zeros.  Why: HumanoidStandup's benchmark has more than 24 rows in 2 % of its envs
  // (at most ~30) and more than 20 in 10 %, but a wave used to stream ALL rows of ALL its 16 envs in the
  // y-space form (SolvePgs: an HBM round trip per visit) as soon as one env had one row too many --
  // 172 GB of HBM traffic per launch.
  static constexpr int kC = MP::kCacheRows;
  static EPA_HD constexpr int TriO(int r, int cc) { return r >= cc ? r * (r + 1) / 2 + cc : cc * (cc + 1) / 2 + r; }
  static constexpr int kShAinv = kC * (kC + 1) / 2;  // shared block: the packed overflow block, then 1 / (A_cc + R_c)
  // what a solve reads of the forward pass and what it returns
  struct SolveIn {
    V Llt[kNS][kNT], Lll[kNLL], dinv_l[kNS], accs_l[kNS], zsd[kND];
    E accs_t[kNT], cost;
    int nrow_e, max_iter;
  };
  struct SolveOut {
    E at[kNT];
    V al[kNS];
    int stat[kNStat];
    bool rej;  // the cost check of a visit fired: the caller solves again in the exact form
  };
  static EPA_HD void MakeSolveIn(const Fwd<V>& f, int nrow_e, const V* zsd, E cost, int max_iter, SolveIn& in) {
    static_for<0, kNS>([&](auto sc) {
      constexpr int k = decltype(sc)::value;
      static_for<0, kNT>([&](auto jc) { in.Llt[k][decltype(jc)::value] = f.Llt[k][decltype(jc)::value]; });
      in.dinv_l[k] = f.dinv_l[k];
      in.accs_l[k] = f.accs_l[k];
    });
    static_for<0, kNLL>([&](auto ic) { in.Lll[decltype(ic)::value] = f.Lll[decltype(ic)::value]; });
    static_for<0, kND>([&](auto ic) { in.zsd[decltype(ic)::value] = zsd[decltype(ic)::value]; });
    static_for<0, kNT>([&](auto jc) { in.accs_t[decltype(jc)::value] = f.accs_t[decltype(jc)::value]; });
    in.cost = cost;
    in.nrow_e = nrow_e;
    in.max_iter = max_iter;
  }
  static EPA_HD void FactorOf(const SolveIn& in, Fwd<V>& g) {
    static_for<0, kNS>([&](auto sc) {
      constexpr int k = decltype(sc)::value;
      static_for<0, kNT>([&](auto jc) { g.Llt[k][decltype(jc)::value] = in.Llt[k][decltype(jc)::value]; });
      g.dinv_l[k] = in.dinv_l[k];
      g.accs_l[k] = in.accs_l[k];
    });
    static_for<0, kNLL>([&](auto ic) { g.Lll[decltype(ic)::value] = in.Lll[decltype(ic)::value]; });
    static_for<0, kNT>([&](auto jc) { g.accs_t[decltype(jc)::value] = in.accs_t[decltype(jc)::value]; });
  }
  template <bool kOv>
  static EPA_HD void SolvePgsT(Ctx& c, const SolveIn& in, SolveOut& out) {
    const int nrow_e = in.nrow_e, max_iter = in.max_iter;
    const V* zsd = in.zsd;
    const E cost = in.cost;
    int* stat = out.stat;
    for (int i = 0; i < kNStat; ++i) stat[i] = 0;
    out.rej = false;
    constexpr TreeModel m = MP::kM;
    const long long tk_in = EPA_HUM_TICK();
    const int nrow = WaveMax(nrow_e);
    V dd[kND];
    DinvD(c, in.dinv_l, dd);
    const bool cold = ColdStart(zsd, dd, cost);
    // A_rc = y_r . (y_c / D) for c < r, four columns at a time, one row of lookahead; rows an env does
    // not have count as zero.  The shared block holds 16 rows of the packed triangle: more rows are
    // staged in further passes (four rows of the triangle each), the registers being filled pass by pass.
    auto load = [&](int r, V* y) {
      c.RowGet(r, y);
      const bool valid = r < nrow_e;
      static_for<0, kND>([&](auto ic) { y[decltype(ic)::value] = Sel(valid, y[decltype(ic)::value], V(0)); });
    };
    // (hybrid) the overflow rows' entries against the register rows go through the workspace (Ctx::OvPut) FIRST,
    // while the registers are still empty: four register rows resident at a time, the overflow rows stream past.
    // Done after the dual matrix is in registers, the temporaries of this loop push parts of the matrix out to
    // scratch memory for the whole solve.
    if constexpr (kOv) {
      const int nov = nrow - kRegRows;
      for (int c0 = 0; c0 < kRegRows; c0 += 4) {
        V w[4][kND];
        static_for<0, 4>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          load(c0 + k, w[k]);
          static_for<0, kND>([&](auto ic) { w[k][decltype(ic)::value] *= dd[decltype(ic)::value]; });
        });
        V nx[kND];  // (one row of lookahead: a row load is an exposed ~700-cycle round trip otherwise)
        load(kRegRows, nx);
        for (int j = 0; j < nov; ++j) {
          V y[kND];
          static_for<0, kND>([&](auto ic) { y[decltype(ic)::value] = nx[decltype(ic)::value]; });
          load(kRegRows + (j + 1 < nov ? j + 1 : j), nx);
          E a4[4];
          static_for<0, 4>([&](auto kc) { a4[decltype(kc)::value] = DotD(y, w[decltype(kc)::value]); });
          c.OvPut(j, c0 >> 2, LanePickV(a4, 0));
        }
      }
    }
    V fo[kOwn], ainv[kOwn], arr[kOwn], S[kOwn], AR[kOwn][kRegRows];
    static_for<0, kOwn>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const BV have = c.RowIndexLane(4 * k) < V(nrow_e);  // 4 k + lane < nrow_e
      fo[k] = Sel(have, Sel(cold, V(0), c.RsGetLane(4 * k, kRsF)), V(0));
      ainv[k] = Sel(have, c.RsGetLane(4 * k, kRsAinv), V(0));
      arr[k] = Sel(have, c.RsGetLane(4 * k, kRsArr), V(0));
      S[k] = Sel(have, c.RsGetLane(4 * k, kRsB), V(0));
    });
    // staging passes: rows [0, 16), then four rows at a time
    constexpr int kPasses = kRegRows <= 16 ? 1 : 1 + (kRegRows - 16) / 4;
    static_for<0, kPasses>([&](auto pc) {
      constexpr int p = decltype(pc)::value;
      constexpr int r_lo = p == 0 ? 0 : 16 + 4 * (p - 1);
      constexpr int r_hi = p == 0 ? (kRegRows < 16 ? kRegRows : 16) : r_lo + 4;
      constexpr int base = r_lo * (r_lo + 1) / 2;  // Tri(r_lo, 0)
      static_assert(r_hi * (r_hi + 1) / 2 - base <= kFSlots, "a staging pass fits the shared block");
      const int nhi = nrow < r_hi ? nrow : r_hi;  // rows of this pass: [r_lo, nhi)
      if (nhi > r_lo) {  // wave uniform
        for (int c0 = 0; c0 < nhi; c0 += 4) {
          V w[4][kND];
          static_for<0, 4>([&](auto kc) { load(c0 + decltype(kc)::value < nrow ? c0 + decltype(kc)::value : c0, w[decltype(kc)::value]); });
          static_for<0, 4>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            V wk[kND];
            static_for<0, kND>([&](auto ic) { wk[decltype(ic)::value] = w[k][decltype(ic)::value] * dd[decltype(ic)::value]; });
            static_for<k + 1, 4>([&](auto rc4) {
              constexpr int r = decltype(rc4)::value;
              if (c0 >= r_lo && c0 + r < nhi) c.ShPut(Tri(c0 + r, c0 + k) - base, DotD(w[r], wk));
            });
            static_for<0, kND>([&](auto ic) { w[k][decltype(ic)::value] = wk[decltype(ic)::value]; });
          });
          const int r0 = c0 + 4 > r_lo ? c0 + 4 : r_lo;
          if (r0 < nhi) {
            V nx[kND];
            load(r0, nx);
            for (int r = r0; r < nhi; ++r) {
              V y[kND];
              static_for<0, kND>([&](auto ic) { y[decltype(ic)::value] = nx[decltype(ic)::value]; });
              load(r + 1 < nhi ? r + 1 : r, nx);
              static_for<0, 4>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const E a = DotD(y, w[k]);
                if (c0 + k < nrow) c.ShPut(Tri(r, c0 + k) - base, a);
              });
            }
          }
        }
      }
      // the lane's own rows r = 4 k + l take their entries with max(row, column) in this pass's rows
      static_for<0, kOwn>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const BV have = c.RowIndexLane(4 * k) < V(nrow_e);
        static_for<0, kRegRows>([&](auto cc) {
          constexpr int cidx = decltype(cc)::value;
          constexpr int top = 4 * k > cidx ? 4 * k : cidx;  // (4 k .. 4 k + 3 lie in one pass)
          if constexpr (top >= r_lo && top < r_hi) {
            // (entries of rows / columns beyond the wave's row count are stale but finite: they only
            // ever meet a zero force)
            // (the run-time index here; the compile-time form of the overflow visits -- ShGetTriRow -- measured
            // 3 % slower for HumanoidStandup in this place, profiles/archive/r3zj_*)
            const V a = c.ShGetTriLane(4 * k, cidx, base);
            AR[k][cidx] = Sel(c.RowIndexLane(4 * k) == V(cidx), arr[k], Sel(have, a, V(0)));
          }
        });
      });
    });
    // S_r = b_r + sum_c (A + R)_rc f_c at the start forces
    static_for<0, kRegRows>([&](auto cc) {
      constexpr int cidx = decltype(cc)::value;
      const E fc = BcastQS<cidx & 3>(fo[cidx >> 2]);
      static_for<0, kOwn>([&](auto kc) { S[decltype(kc)::value] += AR[decltype(kc)::value][cidx] * V(fc); });
    });
    // overflow rows: A_{c, register rows} (four register rows resident at a time, the overflow rows
    // stream past; through Ctx::OvPut / OvGet into registers indexed at compile time), the overflow block,
    // the start forces and their share of S
    constexpr int kCo = kC / 4;
    static_assert(kC % 4 == 0, "overflow rows: a multiple of 4");
    V av[kC][kOwn], Sov[kCo], fv[kCo];
    typename Ctx::TriBase trb[kCo];  // the lane's overflow rows in the packed triangle (LDS addresses without arithmetic)
    if constexpr (kOv) {
      static_assert(kShAinv + kC <= kFSlots, "the overflow block fits the shared block");
      const int nov = nrow - kRegRows;  // wave level, 1 .. kC
      for (int i = 0; i < kShAinv + kC; ++i) c.ShPut(i, E(0));  // (rows the wave does not have: exact zeros)
      for (int jc = 0; jc < nov; ++jc) {
        V w[kND];
        load(kRegRows + jc, w);
        static_for<0, kND>([&](auto ic) { w[decltype(ic)::value] *= dd[decltype(ic)::value]; });
        const bool hv = kRegRows + jc < nrow_e;
        c.ShPut(TriO(jc, jc), hv ? c.RsGet(kRegRows + jc, kRsArr) : E(0));
        c.ShPut(kShAinv + jc, hv ? c.RsGet(kRegRows + jc, kRsAinv) : E(0));
        if (jc + 1 < nov) {
          V nx[kND];
          load(kRegRows + jc + 1, nx);
          for (int jr = jc + 1; jr < nov; ++jr) {
            V y[kND];
            static_for<0, kND>([&](auto ic) { y[decltype(ic)::value] = nx[decltype(ic)::value]; });
            load(kRegRows + (jr + 1 < nov ? jr + 1 : jr), nx);
            c.ShPut(TriO(jr, jc), DotD(y, w));
          }
        }
      }
      static_for<0, kC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const bool hv = kRegRows + j < nrow_e;
        static_for<0, kOwn>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          av[j][k] = V(0);
          if (j < nov) av[j][k] = Sel(hv, c.OvGet(j, k), V(0));  // (wave uniform branch)
        });
      });
      // lane l keeps the force and the tracked part of the residual of the overflow rows j = 4 m + l
      static_for<0, kCo>([&](auto mc) {
        constexpr int mm = decltype(mc)::value;
        const BV have = c.RowIndexLane(kRegRows + 4 * mm) < V(nrow_e);
        fv[mm] = V(0);
        Sov[mm] = V(0);
        if (4 * mm < nov) {
          fv[mm] = Sel(have, Sel(cold, V(0), c.RsGetLane(kRegRows + 4 * mm, kRsF)), V(0));
          Sov[mm] = Sel(have, c.RsGetLane(kRegRows + 4 * mm, kRsB), V(0));
        }
      });
      static_for<0, kCo>([&](auto mc) { trb[decltype(mc)::value] = c.TriRow(4 * decltype(mc)::value); });
      static_for<0, kC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const E fj = BcastQS<j & 3>(fv[j >> 2]);
        static_for<0, kCo>([&](auto mc) {
          constexpr int mm = decltype(mc)::value;
          Sov[mm] += c.template ShGetTriRow<4 * mm, j, 0>(trb[mm]) * V(fj);
        });
        static_for<0, kOwn>([&](auto kc) { S[decltype(kc)::value] += av[j][decltype(kc)::value] * V(fj); });
      });
    }
    const E scale = E(1.0 / (m.meaninertia * 23.0));
    bool done = nrow_e == 0;
    V unit[4] = {LaneUnit<0>(V()), LaneUnit<1>(V()), LaneUnit<2>(V()), LaneUnit<3>(V())};
    EPA_HUM_PIN(unit[0]);
    EPA_HUM_PIN(unit[1]);
    EPA_HUM_PIN(unit[2]);
    EPA_HUM_PIN(unit[3]);
    const long long tk1 = EPA_HUM_TICK();
    stat[kOv ? 15 : 8] += (int)((tk1 - tk_in) >> 4);
    EPA_HUM_COUNT(stat[kOv ? 17 : 16]);
    // A visit is one dependent chain on a SIMD that runs a single wave, so its LENGTH is the cost.  Kept
    // on the chain: S -> clamped force -> change dv -> broadcast -> S.  Taken off it:
    //  * mj_solPGS's safety check "the cost went up by more than 1e-10: take the update back" -- the update is
    //    applied at once, the cost change follows beside the chain, and a wave that ever sees the check fire
    //    (the exact 1-D minimiser of a convex quadratic does not raise it) starts over in the exact y-space
    //    form (SolvePgs), whose visits decide before they apply;
    //  * the masks: a row the env does not have, or any row of an env that is done, has 1 / (A_rr + R_r) = 0
    //    here, which makes dv an exact zero (forces are >= 0).
    V harr[kOwn];
    static_for<0, kOwn>([&](auto kc) { harr[decltype(kc)::value] = V(0.5) * arr[decltype(kc)::value]; });
    bool rej = false;
    bool done_seen = done;
    int nact = WaveMax(done ? 0 : nrow_e);
    for (int iter = 0; iter < max_iter; ++iter) {
      E improvement = E(0);
      // (device: the overflow block is loop invariant -- without this its LDS reads are hoisted out of the
      // sweep loop into ~90 registers, which then spill)
      if constexpr (kOv) c.Refresh();
      // rows of this env the sweep still visits (none once it is done), and the most of them over the wave:
      // one scalar compare per visit decides whether the wave runs it
      const int nlive = done ? 0 : nrow_e;
      // (MP::kLazyNact: the wave-level maximum changes only when an env finishes -- Humanoid + 2.5 %, HumanoidStandup
      // - 1 %, profiles/archive/r3zj_*; otherwise every sweep)
      if (!MP::kLazyNact || AnyWave(done != done_seen)) {
        nact = WaveMax(nlive);
        done_seen = done;
      }
      V ainv_e[kOwn];
      static_for<0, kOwn>([&](auto kc) { ainv_e[decltype(kc)::value] = Sel(done, V(0), ainv[decltype(kc)::value]); });
      static_for<0, kRegRows>([&](auto rc0) {
        constexpr int r = decltype(rc0)::value;
        constexpr int k = r >> 2, o = r & 3;
        if (r < nact) {
          EPA_HUM_COUNT(stat[kOv ? 19 : 0]);
          // (only lane o's numbers mean row r; the others are never looked at)
          const V dv = Max0(fo[k] - S[k] * ainv_e[k]) - fo[k];
          const V cv = (dv * dv) * harr[k] + dv * S[k];
          const E delta = BcastQS<o>(dv);
          fo[k] += V(delta) * unit[o];
          static_for<0, kOwn>([&](auto kc) { S[decltype(kc)::value] += AR[decltype(kc)::value][r] * V(delta); });
          const E change = BcastQS<o>(cv);
          improvement -= change;
          rej = rej || change > E(1e-10);
        }
      });
      if constexpr (kOv) {
        const E livef = done ? E(0) : E(1);
        static_for<0, kC>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if (kRegRows + j < nact) {
            EPA_HUM_COUNT(stat[0]);
            constexpr int mo = j >> 2, o = j & 3;
            // (all LDS reads of the visit first: their latency passes during the dot product)
            V aoo[kCo];
            static_for<0, kCo>([&](auto mc) {
              constexpr int mm = decltype(mc)::value;
              aoo[mm] = c.template ShGetTriRow<4 * mm, j, 0>(trb[mm]);
            });
            const E harr_j = E(0.5) * c.ShGet(TriO(j, j)), ainv_j = c.ShGet(kShAinv + j) * livef;
            V aj[kOwn];
            static_for<0, kOwn>([&](auto kc) { aj[decltype(kc)::value] = av[j][decltype(kc)::value]; });
            V p0 = aj[0] * fo[0], p1 = V(0), p2 = V(0);
            static_for<1, kOwn>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              if constexpr (k % 3 == 0) p0 += aj[k] * fo[k];
              else if constexpr (k % 3 == 1) p1 += aj[k] * fo[k];
              else p2 += aj[k] * fo[k];
            });
            // (only lane o's numbers mean row j from here on)
            const V res = Sov[mo] + V(SumQ((p0 + p1) + p2));
            const V dv = Max0(fv[mo] - res * V(ainv_j)) - fv[mo];
            const V cv = (dv * dv) * V(harr_j) + dv * res;
            const E delta = BcastQS<o>(dv);
            fv[mo] += V(delta) * unit[o];
            static_for<0, kCo>([&](auto mc) { Sov[decltype(mc)::value] += aoo[decltype(mc)::value] * V(delta); });
            static_for<0, kOwn>([&](auto kc) { S[decltype(kc)::value] += aj[decltype(kc)::value] * V(delta); });
            const E change = BcastQS<o>(cv);
            improvement -= change;
            rej = rej || change > E(1e-10);
          }
        });
      }
      ++stat[1];
      stat[4] += done ? 0 : nrow_e;  // this env's own row visits: the key of the cost-sorted scheduling
      if (AnyWave(rej)) break;
      done = done || improvement * scale < E(1e-8);
      if (!AnyWave(!done)) break;
    }
    if (AnyWave(rej)) {  // (never seen) the wave starts over in the exact form; the rows still hold the start forces
      EPA_HUM_COUNT(stat[13]);
      out.rej = true;
      return;
    }
    // z = D^-1 sum_r f_r y_r (the rows stream past once more); efc_force goes back to the rows
    // (mj_rnePostConstraint)
    const long long tk2 = EPA_HUM_TICK();
    stat[kOv ? 14 : 9] += (int)((tk2 - tk1) >> 4);
    if constexpr (kOv) stat[18] += nrow;
    E fr[kRegRows];
    static_for<0, kRegRows>([&](auto rc0) { fr[decltype(rc0)::value] = BcastQS<decltype(rc0)::value & 3>(fo[decltype(rc0)::value >> 2]); });
    V zd[kND];
    static_for<0, kND>([&](auto ic) { zd[decltype(ic)::value] = V(0); });
    static_for<0, kRegRows>([&](auto rc0) {
      constexpr int r = decltype(rc0)::value;
      if (AnyWave(r < nrow_e)) {
        V y[kND];
        load(r, y);
        if (r < nrow_e) c.RsPut(r, kRsF, fr[r]);
        static_for<0, kND>([&](auto ic) { zd[decltype(ic)::value] += V(fr[r]) * y[decltype(ic)::value]; });
      }
    });
    if constexpr (kOv) {
      static_for<0, kC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (AnyWave(kRegRows + j < nrow_e)) {
          V y[kND];
          load(kRegRows + j, y);
          const E fj = BcastQS<j & 3>(fv[j >> 2]);
          if (kRegRows + j < nrow_e) c.RsPut(kRegRows + j, kRsF, fj);
          static_for<0, kND>([&](auto ic) { zd[decltype(ic)::value] += V(fj) * y[decltype(ic)::value]; });
        }
      });
    }
    static_for<0, kND>([&](auto ic) { zd[decltype(ic)::value] *= dd[decltype(ic)::value]; });
    Fwd<V> g;  // (only what Finish reads)
    FactorOf(in, g);
    Finish(c, g, zd, out.at, out.al);
    stat[10] += (int)((EPA_HUM_TICK() - tk2) >> 4);
  }

  // the env's state, distributed: trunk (env level) + this lane's limb
  struct State {
    E qt[10], vt[kNT], wt[kNT];  // qpos (7 + 3), qvel, qacc_warmstart
    V ql[kNS], vl[kNS], wl[kNS];
    E ut[3];   // ctrl of dofs 6 7 8
    V ul[kNS];
  };
  enum { kSttQ = 0, kSttV = 10, kSttW = 19, kSttU = 28, kSttSlots = 31 };
  // bits: 1 qpos, 2 qvel, 4 warm start, 8 ctrl
  static EPA_HD void LoadTrunk(Ctx& c, State& s, int bits) {
    if (bits & 1) static_for<0, 10>([&](auto ic) { s.qt[decltype(ic)::value] = c.SttGet(kSttQ + decltype(ic)::value); });
    if (bits & 2) static_for<0, kNT>([&](auto ic) { s.vt[decltype(ic)::value] = c.SttGet(kSttV + decltype(ic)::value); });
    if (bits & 4) static_for<0, kNT>([&](auto ic) { s.wt[decltype(ic)::value] = c.SttGet(kSttW + decltype(ic)::value); });
    if (bits & 8) static_for<0, 3>([&](auto ic) { s.ut[decltype(ic)::value] = c.SttGet(kSttU + decltype(ic)::value); });
  }
  static EPA_HD void StoreTrunk(Ctx& c, const State& s, int bits) {
    if (bits & 1) static_for<0, 10>([&](auto ic) { c.SttPut(kSttQ + decltype(ic)::value, s.qt[decltype(ic)::value]); });
    if (bits & 2) static_for<0, kNT>([&](auto ic) { c.SttPut(kSttV + decltype(ic)::value, s.vt[decltype(ic)::value]); });
    if (bits & 4) static_for<0, kNT>([&](auto ic) { c.SttPut(kSttW + decltype(ic)::value, s.wt[decltype(ic)::value]); });
    if (bits & 8) static_for<0, 3>([&](auto ic) { c.SttPut(kSttU + decltype(ic)::value, s.ut[decltype(ic)::value]); });
  }
  // ---- the constraint stage of a forward pass: mj_makeConstraint + mj_solPGS, behind a CALL -----------------
  // Inlined into the step kernel, the solver's sweep loops share one register allocation with the tree stages:
  // ~1000 spilled numbers compete for 256 AGPRs, which of them end up in scratch memory does not depend on how hot
  // they are, and a sweep that reloads part of its matrix from scratch waits ~600 cycles per reload (measured:
  // hybrid sweeps 6x the cost of register-form sweeps per row).  Behind a call the stage is allocated on its own and
  // the caller's live numbers are saved once around it.  The rows are built on this side of the call as well: with
  // only the solver behind it, the row loop -- the hungriest of what was left -- paid for the registers that the
  // values living across the call then occupied (5x slower).  What the stage reads of the pass goes by value.
  struct ConIn {
    Sp6<V> lcd[kNS];
    Vec3<E> com;
    V Llt[kNS][kNT], Lll[kNLL], dinv_l[kNS], accs_l[kNS], ql[kNS], vl[kNS], wl[kNS];
    E accs_t[kNT];
    EMask act;
    int dbg;
  };
  struct ConOut {
    E at[kNT];
    V al[kNS];
    RowCount rc;
    int stat[kNStat];
  };
  // MP::kStageCall: HumanoidStandup (many rows, long solves) gains from the call; Humanoid's solves are short
  // and the call's save / restore of ~500 registers costs more than the cleaner loops give back (-15 %): inlined.
  static EPA_HUM_NOINLINE ConOut ConstraintStage(EPA_HUM_CTX(Ctx) c, ConIn ci) { return ConstraintStageBody(c, ci); }
  static EPA_HD ConOut ConstraintStageBody(Ctx& c, const ConIn& ci) {
    ConOut out;
    int* stat = out.stat;
    for (int i = 0; i < kNStat; ++i) stat[i] = 0;
    long long tk0 = EPA_HUM_TICK();
    auto lap = [&](int slot) {
      const long long t = EPA_HUM_TICK();
      stat[slot] += (int)((t - tk0) >> 4);
      tk0 = t;
    };
    const int dbg = ci.dbg;
    Fwd<V> f;  // (only what the rows and the solver read)
    State s;
    static_for<0, kNS>([&](auto sc) {
      constexpr int k = decltype(sc)::value;
      f.lcd[k] = ci.lcd[k];
      static_for<0, kNT>([&](auto jc) { f.Llt[k][decltype(jc)::value] = ci.Llt[k][decltype(jc)::value]; });
      f.dinv_l[k] = ci.dinv_l[k];
      f.accs_l[k] = ci.accs_l[k];
      s.ql[k] = ci.ql[k];
      s.vl[k] = ci.vl[k];
      s.wl[k] = ci.wl[k];
    });
    static_for<0, kNLL>([&](auto ic) { f.Lll[decltype(ic)::value] = ci.Lll[decltype(ic)::value]; });
    static_for<0, kNT>([&](auto jc) { f.accs_t[decltype(jc)::value] = ci.accs_t[decltype(jc)::value]; });
    f.com = ci.com;
    E* at = out.at;
    V* al = out.al;
    E cost;
    V zsd[kND];
    LoadTrunk(c, s, 1 | 2 | 4);
    RowCount rc = MakeRows(c, f, ci.act, s.qt, s.ql, s.vt, s.vl, s.wt, s.wl, zsd, &cost);
    EPA_LDS_FENCE();
    out.rc = rc;
    stat[4] += 16 * rc.rows();  // building a row costs about as much as 16 visits of it
    lap(7);
    ++stat[12];
    if (dbg & 1) rc = RowCount{0, 0, 0};
    const int max_iter = (dbg & 8) ? 0 : MP::kM.iterations;
    bool exact = AnyWave(rc.rows() > kRegRows + kC);  // more rows than the chip holds: the whole wave streams
    if (!exact) {
      SolveIn in;
      MakeSolveIn(f, rc.rows(), zsd, cost, max_iter, in);
      SolveOut o;
      if (!AnyWave(rc.rows() > kRegRows)) {
        SolvePgsT<false>(c, in, o);
        stat[2] += WaveMax(rc.rows());
      } else {
        SolvePgsT<true>(c, in, o);
        ++stat[3];
      }
      exact = AnyWave(o.rej);
      static_for<0, kNStat>([&](auto ic) { stat[decltype(ic)::value] += o.stat[decltype(ic)::value]; });
      static_for<0, kNT>([&](auto jc) { at[decltype(jc)::value] = o.at[decltype(jc)::value]; });
      static_for<0, kNS>([&](auto sc) { al[decltype(sc)::value] = o.al[decltype(sc)::value]; });
    }
    if (exact) {
      SolvePgs(c, f, rc.rows(), zsd, cost, at, al, max_iter, stat);
      stat[3] += 1000;
      lap(11);
    }
    return out;
  }

  // mj_forward: qacc (at, al); `commit`: store it as the warm start
  // `dbg` (timing runs only, wave uniform): 1 no constraint solve, 2 no rows, 4 no detection, 8 no sweeps;
  // stat[0..4] += row visits, sweeps, the wave's rows (register path), streaming solves, this env's solver cost (own row visits + 16 per row built)
  // `after_velocity(f)`: hook right after the smooth dynamics, while cinert / cvel / qfrc_actuator
  // are at hand (the kernel writes its observation there on the last pass; nothing of them has to
  // stay live through the constraint solve)
  template <typename Hook>
  static EPA_HD RowCount Forward(Ctx& c, State& s, Fwd<V>& f, bool commit, E* at, V* al, int dbg,
                                 Hook&& after_velocity, int* stat) {
    // The trunk part of the state (31 numbers, env level) lives in the env's shared block between
    // the stages (Ctx::SttGet / SttPut) and every stage loads what it reads: replicated in the four
    // lanes' registers through the whole pass it costs 62 VGPRs at the register peaks.
    c.Refresh();  // (device: keeps the loop-invariant LDS reads -- 112 limb constants -- from being hoisted out
                  // of the step loop and spilled)
    long long tk0 = EPA_HUM_TICK();
    auto lap = [&](int slot) {
      const long long t = EPA_HUM_TICK();
      stat[slot] += (int)((t - tk0) >> 4);
      tk0 = t;
    };
    LoadTrunk(c, s, 1);
    Position(c, s.qt, s.ql, f);
    EMask act;
    if (dbg & 4) {
      act.w[0] = act.w[1] = act.w[2] = 0ull;
    } else {
      c.Refresh();
      Detect(c, s.qt, s.ql, act);
    }
    if (dbg & 2) act.w[0] = act.w[1] = act.w[2] = 0ull;
    EPA_LDS_FENCE();
    c.Refresh();
    lap(5);
    LoadTrunk(c, s, 1 | 2 | 8);
    Velocity(c, s.qt, s.ql, s.vt, s.vl, s.ut, s.ul, f);
    after_velocity(f);
    MassFactor(c, f);
    SmoothAcc(c, f);
    EPA_LDS_FENCE();
    c.Refresh();
    lap(6);
    // mj_makeConstraint + mj_solPGS: behind a call (ConstraintStage), with what they read of this pass as values
    ConIn in;
    static_for<0, kNS>([&](auto sc) {
      constexpr int k = decltype(sc)::value;
      in.lcd[k] = f.lcd[k];
      static_for<0, kNT>([&](auto jc) { in.Llt[k][decltype(jc)::value] = f.Llt[k][decltype(jc)::value]; });
      in.dinv_l[k] = f.dinv_l[k];
      in.accs_l[k] = f.accs_l[k];
      in.ql[k] = s.ql[k];
      in.vl[k] = s.vl[k];
      in.wl[k] = s.wl[k];
    });
    static_for<0, kNLL>([&](auto ic) { in.Lll[decltype(ic)::value] = f.Lll[decltype(ic)::value]; });
    static_for<0, kNT>([&](auto jc) { in.accs_t[decltype(jc)::value] = f.accs_t[decltype(jc)::value]; });
    in.com = f.com;
    in.act = act;
    in.dbg = dbg;
    ConOut o;
    if constexpr (MP::kStageCall) o = ConstraintStage(c, in);
    else o = ConstraintStageBody(c, in);
    static_for<0, kNStat>([&](auto ic) { stat[decltype(ic)::value] += o.stat[decltype(ic)::value]; });
    static_for<0, kNT>([&](auto jc) { at[decltype(jc)::value] = o.at[decltype(jc)::value]; });
    static_for<0, kNS>([&](auto sc) { al[decltype(sc)::value] = o.al[decltype(sc)::value]; });
    const RowCount rc = o.rc;
    static_for<0, kNT>([&](auto jc) {
      if (commit) c.SttPut(kSttW + decltype(jc)::value, at[decltype(jc)::value]);  // (four lanes, one value)
    });
    static_for<0, kNS>([&](auto sc) { s.wl[decltype(sc)::value] = Sel(commit, al[decltype(sc)::value], s.wl[decltype(sc)::value]); });
    return rc;
  }

  // mj_integratePos: q <- q0 (+) h * vel
  static EPA_HD void IntegratePos(const E* q0t, const V* q0l, const E* dt, const V* dl, E h, bool live, E* qt, V* ql) {
    static_for<0, 3>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const E v = q0t[k] + h * dt[k];
      qt[k] = live ? v : qt[k];
    });
    {
      const Vec3<E> om = {dt[3], dt[4], dt[5]};
      const E nrm = SqrtX(Dot(om, om));
      const bool rot = nrm * h > E(0);
      const E inv = E(1) / Sel(rot, nrm, E(1));
      E sn, cs;
      SinCos(E(0.5) * nrm * h, &sn, &cs);
      const Quat<E> q0 = {q0t[3], q0t[4], q0t[5], q0t[6]};
      const Quat<E> q1 = QNormalize(QMul(q0, Quat<E>{cs, om.x * inv * sn, om.y * inv * sn, om.z * inv * sn}));
      qt[3] = live ? Sel(rot, q1.w, q0.w) : qt[3];
      qt[4] = live ? Sel(rot, q1.x, q0.x) : qt[4];
      qt[5] = live ? Sel(rot, q1.y, q0.y) : qt[5];
      qt[6] = live ? Sel(rot, q1.z, q0.z) : qt[6];
    }
    static_for<0, 3>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const E v = q0t[7 + k] + h * dt[6 + k];
      qt[7 + k] = live ? v : qt[7 + k];
    });
    static_for<0, kNS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      ql[s] = Sel(live, q0l[s] + V(h) * dl[s], ql[s]);
    });
  }

  // RK4 bookkeeping of one mj_step (mj_RungeKutta, N = 4): X0 and the weighted sums of the stage
  // derivatives, kept by the Ctx between the stages (RkGet / RkPut: trunk numbers, env level;
  // RkGetL / RkPutL: the lane's limb numbers) and touched element by element -- loaded as one
  // block next to the state they cost ~100 registers at the end of every forward pass.
  enum { kRkX0q = 0, kRkX0v = 10, kRkAccq = 19, kRkAccv = 28, kRkTrunk = 37,
         kRkX0ql = 0, kRkX0vl = 4, kRkAccql = 8, kRkAccvl = 12, kRkLimb = 16 };
  // One stage boundary, called after the forward evaluation of stage `stage` (0: at the start
  // state) with its qacc (at, al).  Stages 0..2 move the state to the next stage point, stage 3
  // finishes the step.  `live`: envs that really integrate.
  static EPA_HD void RkAdvance(Ctx& c, State& s, int stage, bool live, const E* at, const V* al) {
    constexpr TreeModel m = MP::kM;
    LoadTrunk(c, s, 1 | 2);
    const E h = E(m.timestep);
    const E B = stage == 0 || stage == 3 ? E(1.0 / 6.0) : E(1.0 / 3.0);
    const E A = stage == 2 ? E(1.0) : E(0.5);
    const bool first = stage == 0, last = stage == 3;
    // velocities and their weighted sums; dq: what mj_integratePos advances the positions by
    E dq[kNT];
    V dql[kNS];
    static_for<0, kNT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (first) c.RkPut(kRkX0v + i, s.vt[i]);
      const E x0v = first ? s.vt[i] : c.RkGet(kRkX0v + i);
      const E accq = (first ? E(0) : c.RkGet(kRkAccq + i)) + B * s.vt[i];
      const E accv = (first ? E(0) : c.RkGet(kRkAccv + i)) + B * at[i];
      if (!last) {
        c.RkPut(kRkAccq + i, accq);
        c.RkPut(kRkAccv + i, accv);
      }
      dq[i] = last ? accq : A * s.vt[i];
      const E vn = last ? x0v + h * accv : x0v + h * A * at[i];
      s.vt[i] = live ? vn : s.vt[i];
    });
    static_for<0, kNS>([&](auto sc) {
      constexpr int i = decltype(sc)::value;
      if (first) c.RkPutL(kRkX0vl + i, s.vl[i]);
      const V x0v = first ? s.vl[i] : c.RkGetL(kRkX0vl + i);
      const V accq = (first ? V(0) : c.RkGetL(kRkAccql + i)) + V(B) * s.vl[i];
      const V accv = (first ? V(0) : c.RkGetL(kRkAccvl + i)) + V(B) * al[i];
      if (!last) {
        c.RkPutL(kRkAccql + i, accq);
        c.RkPutL(kRkAccvl + i, accv);
      }
      dql[i] = last ? accq : V(A) * s.vl[i];
      const V vn = last ? x0v + V(h) * accv : x0v + V(h * A) * al[i];
      s.vl[i] = Sel(live, vn, s.vl[i]);
    });
    // positions: X0 (+) h dq
    E x0q[10];
    V x0ql[kNS];
    static_for<0, 10>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (first) c.RkPut(kRkX0q + i, s.qt[i]);
      x0q[i] = first ? s.qt[i] : c.RkGet(kRkX0q + i);
    });
    static_for<0, kNS>([&](auto sc) {
      constexpr int i = decltype(sc)::value;
      if (first) c.RkPutL(kRkX0ql + i, s.ql[i]);
      x0ql[i] = first ? s.ql[i] : c.RkGetL(kRkX0ql + i);
    });
    IntegratePos(x0q, x0ql, dq, dql, h, live, s.qt, s.ql);
    StoreTrunk(c, s, 1 | 2);
    EPA_LDS_FENCE();
  }

  // mj_rnePostConstraint, cfrc_ext part: contact forces of the LAST forward evaluation as spatial
  // forces [torque; force] about the c-frame origin (mju_decodePyramid for the floor), per body:
  // the world and trunk bodies 0 1 2 3 (env level) and the lane's A B C.
  static EPA_HD void ContactWrench(Ctx& c, const Fwd<V>& f, RowCount rc, Sp6<E>* ext_t, Sp6<V>* ext_l) {
    constexpr TreeModel m = MP::kM;
    static_for<0, kNTB + 1>([&](auto bc) { ext_t[decltype(bc)::value] = {{E(0), E(0), E(0)}, {E(0), E(0), E(0)}}; });
    static_for<0, 3>([&](auto bc) { ext_l[decltype(bc)::value] = {{V(0), V(0), V(0)}, {V(0), V(0), V(0)}}; });
    const int ncon = rc.nf + rc.np;  // this env's contacts: floor first, then pairs
    for (int i = 0; AnyWave(i < ncon); ++i) {
      const bool have = i < ncon;
      const bool is_floor = i < rc.nf;
      const int t = have ? i : 0;
      Vec3<E> F;
      if (is_floor) {
        const int r = rc.nl + 4 * i;
        const E f0 = c.RsGet(r, kRsF), f1 = c.RsGet(r + 1, kRsF), f2 = c.RsGet(r + 2, kRsF), f3 = c.RsGet(r + 3, kRsF);
        // frame rows n = z, t1 = y, t2 = -x
        F = {-(f2 - f3) * E(m.floor_mu), (f0 - f1) * E(m.floor_mu), f0 + f1 + f2 + f3};
      } else {
        const E fr = c.RsGet(have ? rc.nl + 4 * rc.nf + (i - rc.nf) : 0, kRsF);
        F = Vec3<E>{c.RecGet(t, 3), c.RecGet(t, 4), c.RecGet(t, 5)} * fr;
      }
      F = {have ? F.x : E(0), have ? F.y : E(0), have ? F.z : E(0)};
      Vec3<E> off = Vec3<E>{c.RecGet(t, 0), c.RecGet(t, 1), c.RecGet(t, 2)} - f.com;
      off = {have ? off.x : E(0), have ? off.y : E(0), have ? off.z : E(0)};
      const Vec3<E> tq = Cross(off, F);
      const int b1 = have ? (int)c.RecGet(t, 6) : 0, b2 = have ? (int)c.RecGet(t, 7) : 0;
      static_for<0, kNTB + 1>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        const E sg = E((b2 == b ? 1 : 0) - (b1 == b ? 1 : 0));
        ext_t[b].a = ext_t[b].a + tq * sg;
        ext_t[b].l = ext_t[b].l + F * sg;
      });
      static_for<0, 3>([&](auto bc) {
        constexpr int w = decltype(bc)::value;
        constexpr int body[4] = {kLimbA[0] + w, kLimbA[1] + w, w < 2 ? kLimbA[2] + w : 31, w < 2 ? kLimbA[3] + w : 31};
        const unsigned s2 = 1u << b2, s1 = 1u << b1;
        const V sg = BitLane(s2, body, V()) - BitLane(s1, body, V());
        ext_l[w].a = ext_l[w].a + LiftV(tq) * sg;
        ext_l[w].l = ext_l[w].l + LiftV(F) * sg;
      });
    }
  }

  static constexpr double GearOfDof(int d) {
    for (int u = 0; u < MP::kM.nu; ++u) {
      if (MP::kM.act_dof[u] == d) return MP::kM.act_gear[u];
    }
    return 0.0;
  }
};

}  // namespace hum4
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_HUM4_HIP_H_
