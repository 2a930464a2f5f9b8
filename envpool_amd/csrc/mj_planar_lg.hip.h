// K3' — `mj_step` of the planar legged gym robots with ONE ENV PER LANE GROUP: the two-legged HalfCheetah and
// Walker2d split over 2 or 4 adjacent lanes of a wavefront, the one-legged Hopper on a group of ONE lane.
//
// Same arithmetic as mj_cheetah.hip.h (MuJoCo 3.6.0's mj_step for
// third_party/mujoco_gym_xml_patches/{half_cheetah,walker2d,walker2d_v5}_envpool.xml, called
// `frame_skip` times per env-step from envpool/mujoco/gym/mujoco_env.h:137-148; SURVEY.md §8a
// M1-M9), re-laid out for the machine instead of for one thread per env:
//
//  * the robot is a torso (x / z slides + y hinge) with two 3-link legs, so M and the Newton
//    Hessian H = M + J^T D J are "arrow" matrices: a 3x3 torso block, two 3x3 couplings, two 3x3
//    leg blocks and NO leg/leg block.  A lane owns ONE LEG: its three bodies, three hinges, its
//    capsule end spheres, its 3x3 + 3x3 blocks; the torso quantities (pose, velocity, the
//    reduced 3x3) are replicated over the group.  A lane therefore carries a packed 6x6 (21
//    numbers) instead of the 39 structural non-zeros of the 9x9 and 6-vectors instead of
//    9-vectors: the one-env-per-lane kernel needs all 512 registers of a SIMD lane plus scratch;
//    this one needs 256 + ~200 accumulation registers and no scratch.  Both run at ONE wave per SIMD
//    (a second wave would need <= 256 registers in all: measured, it spills 200+ and is slower);
//  * both legs execute the SAME instructions: everything that differs between the back and the
//    front leg (link offsets, masses, joint ranges, gears, end spheres) is a per-lane constant
//    read from a small table (`Cx::C(id)`), everything that is the same for the whole robot is an
//    immediate of the compile-time model;
//  * KL = 2: lane = leg, a lane owns the 6 end spheres of its leg and the 2 of "its" torso geom
//    (back-leg lane: torso capsule, front-leg lane: head capsule), a wave holds 32 envs;
//    KL = 4: the two lanes of a leg duplicate the leg's state and SPLIT its end spheres by
//    parity (4 per lane), a wave holds 16 envs -- more waves for small batches;
//  * leg elimination is local: U U^T from the last dof up eliminates the three hinges inside the
//    lane, the two 3x3 Schur complements are summed over the group (one DPP quad_perm move + add
//    per number) and every lane factors the same 3x3 torso block.  Cross-lane traffic per Newton
//    iteration: 9 numbers for H and the gradient, 9 for the factorisation / solve, 5 for the
//    products with M, 2 per line-search evaluation -- all in registers, no LDS;
//  * as in mj_cheetah.hip.h: per-contact constants live in LDS [slot][lane] and there is no
//    lane-divergent control flow in the solver (branches are on wave-wide ballots, per-lane
//    differences are selects / zero weights); since round 5 a lane visits ITS OWN touching
//    end-sphere slots, one per trip of a loop that runs as often as the busiest lane needs
//    (rounds 3-4: a scalar loop over the wave-uniform union of the touching slots).
// A wave runs as long as its slowest env and as its busiest lane: with 32 or 16 envs per wave
// instead of 64 both maxima shrink.
// The same source runs on the host with V = LV<T, KL> (tests/cpu_harness/planar_lg_host.cpp), so
// it is diffed against oracle/mjcpu on a CPU box before it ever sees a GPU.
#ifndef ENVPOOL_AMD_CSRC_MJ_PLANAR_LG_HIP_H_
#define ENVPOOL_AMD_CSRC_MJ_PLANAR_LG_HIP_H_

#include "mj_cheetah.hip.h"  // EPA_HD, static_for, V3 / In4 algebra, Rsqrt, SinCos, WaveAny, CheetahModel

// Stage timers of the diagnostic build (-DEPA_LG_TIMERS, tools/build_alt_lg.sh; never in the product
// library): EPA_LG_TICK(cx, K) books the cycles since the previous tick to category K of the wave
// (0 load / store / loop overhead (RK4: + the stage updates), 1 kinematics + smooth forces + constraint rows, 2 pass
// over the rows + group sums + stop tests, 3 factor / solve / products with M, 4 line search, 5 the Euler integration
// with implicit damping), EPA_LG_COUNT(cx, K)
// counts wave-level loop trips (0 Newton trips, 1 line-search evaluations, 2 forward passes).
#if defined(EPA_LG_TIMERS) && defined(__HIP_DEVICE_COMPILE__)
#define EPA_LG_TICK(cx, K) (cx).template TickEnd<K>()
#define EPA_LG_COUNT(cx, K) (cx).template Count<K>()
#else
#define EPA_LG_TICK(cx, K) ((void)0)
#define EPA_LG_COUNT(cx, K) ((void)0)
#endif
// host experiments only (tools/lg_desync/rollout_host.cpp, tools/lg_desync_sim.py): per Newton trip of one env, the line-search evaluations it ran
#ifndef EPA_LG_HOST_TRIP
#define EPA_LG_HOST_TRIP(evals) ((void)0)
#endif
#ifndef EPA_LG_HOST_ENDS
#define EPA_LG_HOST_ENDS(ends) ((void)0)
#endif
#ifndef EPA_LG_HOST_OWN
#define EPA_LG_HOST_OWN(own) ((void)0)
#endif
#ifndef EPA_LG_HOST_ROWS
#define EPA_LG_HOST_ROWS() ((void)0)
#endif
#ifndef EPA_LG_LS_RTOL
#define EPA_LG_LS_RTOL 1e-10
#endif
#ifndef EPA_LG_LS_MAX  // evaluations of one line search (1: see plg::Solve; A/B builds and host experiments: 2, 24)
#define EPA_LG_LS_MAX 1
#endif

namespace epa {
namespace mj {
namespace plg {

constexpr int kLV = 6;    // local dofs of a lane: rootx rootz rooty (replicated) + the leg's 3 hinges
constexpr int kLB = 4;    // local bodies: torso (replicated), thigh, shin, foot
constexpr int kLTri = 21; // packed 6x6
EPA_HD constexpr int Tri(int i, int j) { return j * (j + 1) / 2 + i; }  // i <= j
EPA_HD constexpr int LDofBody(int j) { return j < 3 ? 0 : j - 2; }
EPA_HD constexpr bool LInChain(int j, int b) { return j < 3 || (j - 2) <= b; }

template <int KL>
struct Grp {
  static_assert(KL == 1 || KL == 2 || KL == 4, "lane group of 1, 2 or 4");
  static constexpr int kLanes = KL;
  // KL = 1: the single-leg Hopper -- a lane IS the env (torso + its only leg): no cross-lane traffic
  // at all, 64 envs per wave; the torso capsule's two ends + the leg's six are its 8 slots
  static constexpr int kEnds = KL == 1 ? 8 : 16 / KL;  // end-sphere slots of a lane
  // body-body capsule pairs (frictionless rows) exist in the Hopper model only
  static constexpr bool kPairs = KL == 1;
  // line-search cache per end-sphere slot: (Jn.a, Jx.a), left by the pass over the rows (round 5: the (Jn.s, Jx.s)
  // slots of the exact search went with it, see Solve)
  static constexpr int kCachePerEnd = 2;
  // local body of slot s (KL = 1, 2: both ends of a body are consecutive slots; KL = 4: one end per body)
  EPA_HD static constexpr int SlotBody(int s) { return KL == 4 ? s : s / 2; }
  // lane coordinate c (0 .. KL-1) -> leg, parity
  EPA_HD static constexpr int Leg(int c) { return KL == 1 ? 0 : (KL == 2 ? c : c >> 1); }
  EPA_HD static constexpr int Par(int c) { return KL == 4 ? c & 1 : 0; }
  // global end-sphere index (mj_cheetah.hip.h numbering: 2 * geom + end) of slot s of lane c
  EPA_HD static constexpr int GlobalEnd(int c, int s) {
    const int b = SlotBody(s), leg = Leg(c);
    const int geom = b == 0 ? leg : 1 + 3 * leg + b;
    const int end = KL == 4 ? Par(c) : (s & 1);
    return 2 * geom + end;
  }
};
// Body-body collision candidates of the Hopper (mj_cheetah.hip.h: kNPair, PairBody1 / PairBody2): torso-leg,
// torso-foot, thigh-foot; bit 16 + k of the `ends` set, bit 27 + k of the active-row mask
constexpr int kPairEndBit = 16, kPairMaskBit = 27;

// ---- per-lane constant table -----------------------------------------------------------------
// [id][c], c = lane coordinate in the group.  Ids of the leg's bodies / hinges are relative
// (body 1..3 -> +0..2, hinge 0..2).
enum TabId {
  kTLx = 0, kTLz = 3, kTMass = 6, kTIyy = 9, kTCx = 12, kTCz = 15,        // bodies 1..3
  kTStiff = 18, kTDamp = 21, kTArm = 24, kTLo = 27, kTHi = 30, kTGear = 33, kTInvw = 36,  // hinges
  kTMu = 39, kTDiag = 43, kTD2mu = 47,  // per local body 0..3: mu, body_invw (1 + mu^2), 1 / (2 mu^2)
  kTEx = 51,                            // kTEx + s, kTEz + s, kTEr + s for slot s (kEnds each)
};
template <int KL>
struct Tab {
  static constexpr int kEz = kTEx + Grp<KL>::kEnds, kEr = kTEx + 2 * Grp<KL>::kEnds;
  static constexpr int kIds = kTEx + 3 * Grp<KL>::kEnds;
  static constexpr int kSize = kIds * KL;
};
// fills tab[Tab<KL>::kSize] from the 7-body model (host, at pool construction / in the tests)
template <int KL>
inline void BuildTable(const CheetahModel<double>& m, double* tab) {
  using G = Grp<KL>;
  for (int c = 0; c < KL; ++c) {
    const int leg = G::Leg(c);
    auto put = [&](int id, double x) { tab[id * KL + c] = x; };
    for (int k = 0; k < 3; ++k) {
      const int b = 1 + 3 * leg + k, j = 3 * leg + k;
      put(kTLx + k, m.lx[b]);
      put(kTLz + k, m.lz[b]);
      put(kTMass + k, m.mass[b]);
      put(kTIyy + k, m.iyy[b]);
      put(kTCx + k, m.cx[b]);
      put(kTCz + k, m.cz[b]);
      put(kTStiff + k, m.stiff[j]);
      put(kTDamp + k, m.damp[j]);
      put(kTArm + k, m.arm[j]);
      // KL = 4: the limit rows of a leg are counted by its parity-0 lane only (the other lane's
      // range is unbounded, so its rows never switch on)
      put(kTLo + k, G::Par(c) == 0 ? m.lo[j] : -1e30);
      put(kTHi + k, G::Par(c) == 0 ? m.hi[j] : 1e30);
      put(kTGear + k, m.gear[j]);
      put(kTInvw + k, m.dof_invw[j]);
    }
    for (int lb = 0; lb < kLB; ++lb) {
      const int b = lb == 0 ? 0 : 3 * leg + lb;
      const double mu = m.bmu[b];
      put(kTMu + lb, mu);
      put(kTDiag + lb, m.body_invw[b] * (1.0 + mu * mu));  // diagApprox of a pyramidal row
      put(kTD2mu + lb, 1.0 / (2.0 * mu * mu));             // R_py = 2 mu^2 R
    }
    for (int s = 0; s < G::kEnds; ++s) {
      const int e = G::GlobalEnd(c, s);
      put(kTEx + s, m.ex[e]);
      put(Tab<KL>::kEz + s, m.ez[e]);
      put(Tab<KL>::kEr + s, m.er[e]);
    }
  }
}

// LDS slots of a lane, [slot][lane]: 5 per end-sphere slot (cpx cpz aref_n B*mu*vx D), then 2 per
// end-sphere slot that live for one Newton iteration: (Jn.a, Jx.a) left by the pass over the rows for the
// line-search evaluation (which builds J . s itself: one evaluation per iteration, see Solve).  Behind the cache the
// Hopper's three body pairs take kSlotsPerPair each (nx nz cx cz aref D, as in mj_cheetah.hip.h).
constexpr int kSlotsPerEnd = 5;
template <int KL>
constexpr int CacheBase() { return Grp<KL>::kEnds * kSlotsPerEnd; }
template <int KL>
constexpr int PairBase() { return Grp<KL>::kEnds * (kSlotsPerEnd + Grp<KL>::kCachePerEnd); }
template <int KL>
constexpr int LdsSlots() { return PairBase<KL>() + (Grp<KL>::kPairs ? kNPair * kSlotsPerPair : 0); }

// ================================================================================================
// The lane vocabulary.  Device: a value IS a lane's scalar, conditions are bool, the group
// reductions are DPP quad_perm moves.  Host: LV<T, KL> carries the KL lanes of ONE env.
// ================================================================================================
template <typename T, int K>
struct LV;
template <int K>
struct LB {
  bool v[K];
  friend inline LB operator&(LB a, LB b) { LB r; for (int i = 0; i < K; ++i) r.v[i] = a.v[i] && b.v[i]; return r; }
  friend inline LB operator|(LB a, LB b) { LB r; for (int i = 0; i < K; ++i) r.v[i] = a.v[i] || b.v[i]; return r; }
  friend inline LB operator!(LB a) { LB r; for (int i = 0; i < K; ++i) r.v[i] = !a.v[i]; return r; }
};
template <int K>
struct LU {
  unsigned v[K];
};
template <typename T, int K>
struct LV {
  T v[K];
  LV() = default;
  template <typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>
  LV(U x) { for (int i = 0; i < K; ++i) v[i] = (T)x; }  // NOLINT: broadcast
#define EPA_LV_BIN(op)                                                   \
  friend inline LV operator op(const LV& a, const LV& b) {               \
    LV r;                                                                \
    for (int i = 0; i < K; ++i) r.v[i] = a.v[i] op b.v[i];               \
    return r;                                                            \
  }
  EPA_LV_BIN(+)
  EPA_LV_BIN(-)
  EPA_LV_BIN(*)
  EPA_LV_BIN(/)
#undef EPA_LV_BIN
  friend inline LV operator-(const LV& a) { LV r; for (int i = 0; i < K; ++i) r.v[i] = -a.v[i]; return r; }
  LV& operator+=(const LV& b) { return *this = *this + b; }
  LV& operator-=(const LV& b) { return *this = *this - b; }
  LV& operator*=(const LV& b) { return *this = *this * b; }
#define EPA_LV_CMP(op)                                                   \
  friend inline LB<K> operator op(const LV& a, const LV& b) {            \
    LB<K> r;                                                             \
    for (int i = 0; i < K; ++i) r.v[i] = a.v[i] op b.v[i];               \
    return r;                                                            \
  }
  EPA_LV_CMP(<)
  EPA_LV_CMP(<=)
  EPA_LV_CMP(>)
  EPA_LV_CMP(>=)
  EPA_LV_CMP(==)
  EPA_LV_CMP(!=)
#undef EPA_LV_CMP
};
using ::epa::mj::SinCos;  // the scalar one (device: FastSinCos)
template <typename T, int K>
inline void SinCos(LV<T, K> x, LV<T, K>* s, LV<T, K>* c) {
  for (int i = 0; i < K; ++i) {
    s->v[i] = std::sin(x.v[i]);
    c->v[i] = std::cos(x.v[i]);
  }
}
template <typename T, int K>
inline LV<T, K> Sel(LB<K> c, const LV<T, K>& a, const LV<T, K>& b) {
  LV<T, K> r;
  for (int i = 0; i < K; ++i) r.v[i] = c.v[i] ? a.v[i] : b.v[i];
  return r;
}
template <typename T, int K>
inline LV<T, K> SqrtV(const LV<T, K>& x) {
  LV<T, K> r;
  for (int i = 0; i < K; ++i) r.v[i] = std::sqrt(x.v[i]);
  return r;
}
template <typename T, int K>
inline LV<T, K> Rsq(const LV<T, K>& x) {
  LV<T, K> r;
  for (int i = 0; i < K; ++i) r.v[i] = T(1) / std::sqrt(x.v[i]);
  return r;
}
// lane permutations of a group: legs <-> the other leg's lane (same parity), par <-> the other parity
template <int KL>
constexpr int OtherLeg(int c) { return KL == 1 ? c : (KL == 2 ? c ^ 1 : c ^ 2); }
template <int KL, typename T>
inline LV<T, KL> SumLegs(const LV<T, KL>& x) {
  if (KL == 1) return x;  // one leg
  LV<T, KL> r;
  for (int c = 0; c < KL; ++c) r.v[c] = x.v[c] + x.v[OtherLeg<KL>(c)];
  return r;
}
template <int KL, typename T>
inline LV<T, KL> MaxLegs(const LV<T, KL>& x) {
  LV<T, KL> r;
  for (int c = 0; c < KL; ++c) r.v[c] = x.v[c] > x.v[OtherLeg<KL>(c)] ? x.v[c] : x.v[OtherLeg<KL>(c)];
  return r;
}
template <int KL, typename T>
inline LV<T, KL> SumPar(const LV<T, KL>& x) {
  if (KL != 4) return x;
  LV<T, KL> r;
  for (int c = 0; c < KL; ++c) r.v[c] = x.v[c] + x.v[c ^ 1];
  return r;
}
template <int K>
inline LB<K> AllEnv(LB<K> c) {
  bool a = true;
  for (int i = 0; i < K; ++i) a = a && c.v[i];
  LB<K> r;
  for (int i = 0; i < K; ++i) r.v[i] = a;
  return r;
}
template <int K>
inline bool AnyWave(LB<K> c) {
  bool a = false;
  for (int i = 0; i < K; ++i) a = a || c.v[i];
  return a;
}
template <int K>
inline void MaskSet(LU<K>& m, LB<K> on, int bit) {
  for (int i = 0; i < K; ++i) m.v[i] |= (on.v[i] ? 1u : 0u) << bit;
}
template <typename T, int K>
inline void MaskSetNZ(LU<K>& m, const LV<T, K>& w, int bit) {  // bit set where w != 0
  for (int i = 0; i < K; ++i) m.v[i] |= (w.v[i] != T(0) ? 1u : 0u) << bit;
}
template <int K>
inline LB<K> MaskSame(const LU<K>& a, const LU<K>& b) {
  LB<K> r;
  for (int i = 0; i < K; ++i) r.v[i] = a.v[i] == b.v[i];
  return r;
}
// ---- per-lane slot sets (round 5: a lane visits ITS OWN touching slots, not the wave's union) -------------------
// the lane's set of end-sphere slots to visit: `own` where `on`, else empty
template <int K>
inline LU<K> SlotsWhere(const LU<K>& own, LB<K> on) {
  LU<K> r;
  for (int i = 0; i < K; ++i) r.v[i] = on.v[i] ? (own.v[i] & 0xFFFFu) : 0u;
  return r;
}
template <int K>
inline LB<K> AnySlot(const LU<K>& rem) {
  LB<K> r;
  for (int i = 0; i < K; ++i) r.v[i] = rem.v[i] != 0u;
  return r;
}
// takes the lowest slot out of the lane's set (slot 0 where the set is empty: the caller masks it with AnySlot)
template <int K>
inline LU<K> PopSlot(LU<K>& rem) {
  LU<K> r;
  for (int i = 0; i < K; ++i) {
    r.v[i] = rem.v[i] ? (unsigned)__builtin_ctz(rem.v[i]) : 0u;
    rem.v[i] &= rem.v[i] - 1u;
  }
  return r;
}
template <int KL, int K>
inline LU<K> SlotBodyOf(const LU<K>& s) {
  LU<K> r;
  for (int i = 0; i < K; ++i) r.v[i] = KL == 4 ? s.v[i] : s.v[i] >> 1;
  return r;
}
template <int K>
inline LB<K> AtLeast(const LU<K>& a, unsigned b) {
  LB<K> r;
  for (int i = 0; i < K; ++i) r.v[i] = a.v[i] >= b;
  return r;
}
template <typename T, int K>
inline void MaskSetNZAt(LU<K>& m, const LV<T, K>& w, const LU<K>& slot, int base) {  // bit base + 3 slot where w != 0
  for (int i = 0; i < K; ++i) m.v[i] |= (w.v[i] != T(0) ? 1u : 0u) << (base + 3 * (int)slot.v[i]);
}
template <typename V>
struct LaneTypes;
template <typename T, int K>
struct LaneTypes<LV<T, K>> {
  using B = LB<K>;
  using U = LU<K>;
  static B False() { B b; for (int i = 0; i < K; ++i) b.v[i] = false; return b; }
  static B True() { B b; for (int i = 0; i < K; ++i) b.v[i] = true; return b; }
  static U Fill(unsigned x) { U u; for (int i = 0; i < K; ++i) u.v[i] = x; return u; }
};

// ---- device (and one-lane host) realisation ---------------------------------------------------
template <typename T>
EPA_HD T Sel(bool c, T a, T b) {
  return c ? a : b;
}
EPA_HD double Rsq(double x) { return Rsqrt(x); }
EPA_HD double SqrtV(double x) { return Sqrt(x); }
EPA_HD float Rsq(float x) { return Rsqrt(x); }
EPA_HD bool AnyWave(bool c) { return WaveAny(c); }
EPA_HD void MaskSet(unsigned& m, bool on, int bit) { m |= (on ? 1u : 0u) << bit; }
EPA_HD bool MaskSame(unsigned a, unsigned b) { return a == b; }
EPA_HD unsigned SlotsWhere(unsigned own, bool on) { return on ? (own & 0xFFFFu) : 0u; }
EPA_HD bool AnySlot(unsigned rem) { return rem != 0u; }
EPA_HD unsigned PopSlot(unsigned& rem) {
  const unsigned s = rem ? (unsigned)__builtin_ctz(rem) : 0u;
  rem &= rem - 1u;
  return s;
}
template <int KL>
EPA_HD unsigned SlotBodyOf(unsigned s) { return KL == 4 ? s : s >> 1; }
EPA_HD bool AtLeast(unsigned a, unsigned b) { return a >= b; }
EPA_HD void MaskSetNZAt(unsigned& m, double w, unsigned slot, int base) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned hi = (unsigned)__double2hiint(w);  // see MaskSetNZ
  m |= (hi < 1u ? hi : 1u) << (base + 3 * slot);
#else
  m |= (w != 0.0 ? 1u : 0u) << (base + 3 * slot);
#endif
}
// bit set where w != 0, for a weight that is either +0.0 or a positive normal number: its high word is
// then zero / non-zero, and min(1, high word) is the bit -- two integer VALU ops, no compare, no
// lane-mask logic on the scalar unit (the row weights are already selected by `jar < 0` and carry
// D == 0 for lanes without the contact)
EPA_HD void MaskSetNZ(unsigned& m, double w, int bit) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned hi = (unsigned)__double2hiint(w);
  m |= (hi < 1u ? hi : 1u) << bit;
#else
  m |= (w != 0.0 ? 1u : 0u) << bit;
#endif
}
template <>
struct LaneTypes<double> {
  using B = bool;
  using U = unsigned;
  EPA_HD static B False() { return false; }
  EPA_HD static B True() { return true; }
  EPA_HD static U Fill(unsigned x) { return x; }
};
#if defined(__HIP_DEVICE_COMPILE__)
// quad_perm selectors: [1,0,3,2] swaps neighbours, [2,3,0,1] swaps pairs.  bound_ctrl = true: a
// quad_perm never reads out of bounds, and the compiler then knows the `old` operand is dead.
constexpr int kDppNeighbour = 0xB1, kDppPair = 0x4E;
template <int CTRL>
__device__ __forceinline__ int DppMovI(int x) {
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double DppMovD(double x) {
  const int lo = DppMovI<CTRL>(__double2loint(x));
  const int hi = DppMovI<CTRL>(__double2hiint(x));
  return __hiloint2double(hi, lo);
}
template <int KL>
__device__ __forceinline__ double SumLegs(double x) {
  if constexpr (KL == 1) {
    return x;  // one leg: the lane is the env
  } else {
    return x + DppMovD<KL == 2 ? kDppNeighbour : kDppPair>(x);
  }
}
template <int KL>
__device__ __forceinline__ double MaxLegs(double x) {
  if constexpr (KL == 1) {
    return x;
  } else {
    const double y = DppMovD<KL == 2 ? kDppNeighbour : kDppPair>(x);
    return x > y ? x : y;
  }
}
template <int KL>
__device__ __forceinline__ double SumPar(double x) {
  if constexpr (KL != 4) {
    return x;
  } else {
    return x + DppMovD<kDppNeighbour>(x);
  }
}
template <int KL>
__device__ __forceinline__ bool AllEnvI(bool c) {
  if constexpr (KL == 1) {
    return c;
  } else {
    int x = c ? 1 : 0;
    x &= DppMovI<kDppNeighbour>(x);
    if constexpr (KL == 4) x &= DppMovI<kDppPair>(x);
    return x != 0;
  }
}
#else
template <int KL>
EPA_HD double SumLegs(double x) { return x; }  // hipcc's host pass only: never executed
template <int KL>
EPA_HD double SumPar(double x) { return x; }
template <int KL>
EPA_HD double MaxLegs(double x) { return x; }
template <int KL>
EPA_HD bool AllEnvI(bool c) { return c; }
#endif
// a + b over the WHOLE group
template <int KL, typename V>
EPA_HD V SumEnv(const V& x) {
  return SumLegs<KL>(SumPar<KL>(x));
}
template <int KL>
EPA_HD bool AllEnvOf(bool c) { return AllEnvI<KL>(c); }
template <int KL>
inline LB<KL> AllEnvOf(LB<KL> c) { return AllEnv(c); }

template <typename V>
EPA_HD V Abs(V x) {
  return Sel(x < V(0), -x, x);
}
template <typename T, typename V>
EPA_HD V ImpedanceV(T d0, T dmax, T width, V r) {  // mj::Impedance with selects
  V x = Abs(r) * V(T(1) / width);
  V y = Sel(x <= V(0.5), V(2) * x * x, V(1) - V(2) * (V(1) - x) * (V(1) - x));
  return Sel(x >= V(1), V(dmax), V(d0) + y * V(dmax - d0));
}

// ================================================================================================
// forward pass
// ================================================================================================
template <typename V>
struct Pos {
  V sn[kLB], cs[kLB], px[kLB], pz[kLB];  // body frames (anchors = body origins); body 0 replicated
  V comx, comz;                          // COM of the whole robot
  In4<V> cinert[kLB];
  V coz[kLV], cox[kLV];                  // cdof[j] = (1, coz[j], -cox[j]) for j >= 2
  V M[kLTri];                            // rows / columns [torso | my leg] of the joint-space inertia
};
template <typename V>
EPA_HD V3<V> Cdof(const Pos<V>& p, int j) {
  return j == 0 ? V3<V>{V(0), V(1), V(0)} : j == 1 ? V3<V>{V(0), V(0), V(1)} : V3<V>{V(1), p.coz[j], -p.cox[j]};
}

// mj_kinematics, mj_comPos, mj_crb
template <int KL, typename T, typename V, typename Cx>
EPA_HD void Kinematics(const CheetahModel<T>& m, const Cx& cx, const V* q, Pos<V>& p) {
  static_for<0, kLB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    V sj, cj;
    SinCos(q[b + 2], &sj, &cj);
    if constexpr (b == 0) {
      p.px[0] = V(m.lx[0]) + q[0];  // q[0] may be a local (re-centred) x; dynamics are invariant
      p.pz[0] = V(m.lz[0]) + q[1];
      p.sn[0] = sj;
      p.cs[0] = cj;
    } else {
      const V lx = cx.C(kTLx + b - 1), lz = cx.C(kTLz + b - 1);
      p.px[b] = p.px[b - 1] + p.cs[b - 1] * lx + p.sn[b - 1] * lz;
      p.pz[b] = p.pz[b - 1] - p.sn[b - 1] * lx + p.cs[b - 1] * lz;
      p.sn[b] = p.sn[b - 1] * cj + p.cs[b - 1] * sj;
      p.cs[b] = p.cs[b - 1] * cj - p.sn[b - 1] * sj;
    }
  });
  V xi[kLB], zi[kLB], mass[kLB];
  V sx = V(0), sz = V(0);
  static_for<0, kLB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    if constexpr (b == 0) {
      xi[0] = p.px[0] + p.cs[0] * V(m.cx[0]) + p.sn[0] * V(m.cz[0]);
      zi[0] = p.pz[0] - p.sn[0] * V(m.cx[0]) + p.cs[0] * V(m.cz[0]);
      mass[0] = V(m.mass[0]);
    } else {
      const V ccx = cx.C(kTCx + b - 1), ccz = cx.C(kTCz + b - 1);
      xi[b] = p.px[b] + p.cs[b] * ccx + p.sn[b] * ccz;
      zi[b] = p.pz[b] - p.sn[b] * ccx + p.cs[b] * ccz;
      mass[b] = cx.C(kTMass + b - 1);
      sx += mass[b] * xi[b];
      sz += mass[b] * zi[b];
    }
  });
  const T inv_total = T(1) / m.total_mass;
  p.comx = (V(m.mass[0]) * xi[0] + SumLegs<KL>(sx)) * V(inv_total);
  p.comz = (V(m.mass[0]) * zi[0] + SumLegs<KL>(sz)) * V(inv_total);
  static_for<0, kLB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    const V dx = xi[b] - p.comx, dz = zi[b] - p.comz;
    V iyy;
    if constexpr (b == 0) {
      iyy = V(m.iyy[0]);
    } else {
      iyy = cx.C(kTIyy + b - 1);
    }
    p.cinert[b] = {iyy + mass[b] * (dx * dx + dz * dz), mass[b] * dx, mass[b] * dz, mass[b]};
  });
  static_for<2, kLV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int b = LDofBody(j);
    p.coz[j] = p.comz - p.pz[b];  // hinge about +y at the body origin: (1, oz, -ox), o = com - anchor
    p.cox[j] = p.comx - p.px[b];
  });
  // mj_crb: composite inertias up the leg, both legs summed into the torso
  In4<V> crb[kLB];
  crb[3] = p.cinert[3];
  crb[2] = {p.cinert[2].I + crb[3].I, p.cinert[2].mdx + crb[3].mdx, p.cinert[2].mdz + crb[3].mdz,
            p.cinert[2].m + crb[3].m};
  crb[1] = {p.cinert[1].I + crb[2].I, p.cinert[1].mdx + crb[2].mdx, p.cinert[1].mdz + crb[2].mdz,
            p.cinert[1].m + crb[2].m};
  crb[0] = {p.cinert[0].I + SumLegs<KL>(crb[1].I), p.cinert[0].mdx + SumLegs<KL>(crb[1].mdx),
            p.cinert[0].mdz + SumLegs<KL>(crb[1].mdz), V(m.total_mass)};
  static_for<0, kLV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const V3<V> buf = MulInert(crb[LDofBody(i)], Cdof(p, i));
    static_for<0, i + 1>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      p.M[Tri(j, i)] = Dot(Cdof(p, j), buf);
    });
    if constexpr (i >= 3) p.M[Tri(i, i)] += cx.C(kTArm + i - 3);
  });
}

// qfrc_smooth = passive - bias + actuator (mj_fwdVelocity, mj_fwdActuation); ctrl: the leg's 3 motors
template <int KL, typename T, typename V, typename Cx>
EPA_HD void SmoothForces(const CheetahModel<T>& m, const Cx& cx, const Pos<V>& p, const V* q,
                         const V* v, const V* ctrl, V* qfrc_smooth) {
  V3<V> cvel[kLB], cdd[kLV];
  {
    V3<V> cv = {V(0), V(0), V(0)};
    static_for<0, 3>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const V3<V> cd = Cdof(p, j);
      cdd[j] = CrossMotion(cv, cd);
      cv.w += cd.w * v[j];
      cv.x += cd.x * v[j];
      cv.z += cd.z * v[j];
    });
    cvel[0] = cv;
  }
  static_for<1, kLB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr int j = b + 2;
    V3<V> cv = cvel[b - 1];
    const V3<V> cd = Cdof(p, j);
    cdd[j] = CrossMotion(cv, cd);
    cv.w += cd.w * v[j];
    cv.x += cd.x * v[j];
    cv.z += cd.z * v[j];
    cvel[b] = cv;
  });
  // mj_rne (no acceleration term); world cacc = -gravity
  V3<V> cacc[kLB], cfrc[kLB];
  static_for<0, kLB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    V3<V> a;
    if constexpr (b == 0) {
      a = {V(0), V(0), V(m.gravity)};
      static_for<0, 3>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        a.w += cdd[j].w * v[j];
        a.x += cdd[j].x * v[j];
        a.z += cdd[j].z * v[j];
      });
    } else {
      constexpr int j = b + 2;
      a = cacc[b - 1];
      a.w += cdd[j].w * v[j];
      a.x += cdd[j].x * v[j];
      a.z += cdd[j].z * v[j];
    }
    cacc[b] = a;
    const V3<V> f = MulInert(p.cinert[b], a);
    const V3<V> g = CrossForce(cvel[b], MulInert(p.cinert[b], cvel[b]));
    cfrc[b] = {f.w + g.w, f.x + g.x, f.z + g.z};
  });
  cfrc[2] = {cfrc[2].w + cfrc[3].w, cfrc[2].x + cfrc[3].x, cfrc[2].z + cfrc[3].z};
  cfrc[1] = {cfrc[1].w + cfrc[2].w, cfrc[1].x + cfrc[2].x, cfrc[1].z + cfrc[2].z};
  cfrc[0] = {cfrc[0].w + SumLegs<KL>(cfrc[1].w), cfrc[0].x + SumLegs<KL>(cfrc[1].x),
             cfrc[0].z + SumLegs<KL>(cfrc[1].z)};
  static_for<0, kLV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const V bias = Dot(Cdof(p, j), cfrc[LDofBody(j)]);
    if constexpr (j < 3) {
      qfrc_smooth[j] = -bias;
    } else {
      // mj_passive (spring about qpos_spring = 0, damper) + motor
      qfrc_smooth[j] = -cx.C(kTStiff + j - 3) * q[j] - cx.C(kTDamp + j - 3) * v[j] - bias +
                       cx.C(kTGear + j - 3) * ctrl[j - 3];
    }
  });
}

// Jacobian columns of a contact point (cpx, cpz) on local body B: f(j, Jn_j, Jx_j) for every
// chain dof (Jn = d(z velocity)/d qdot_j, Jx = d(x velocity)/d qdot_j)
template <int B, typename V, typename F>
EPA_HD void ForChainCols(const Pos<V>& p, V cpx, V cpz, F&& f) {
  f(IC<0>{}, V(0), V(1));
  f(IC<1>{}, V(1), V(0));
  static_for<2, kLV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (LInChain(j, B)) {
      constexpr int b = LDofBody(j);
      f(jc, -(cpx - p.px[b]), cpz - p.pz[b]);
    }
  });
}
// friction coefficient of the floor pair of local body B
template <int B, typename T, typename V, typename Cx>
EPA_HD V MuOf(const CheetahModel<T>& m, const Cx& cx) {
  if constexpr (B == 0) {
    return V(m.bmu[0]);
  } else {
    return cx.C(kTMu + B);
  }
}

template <typename V>
struct LimitRows {
  V sgn[3], aref[3], D[3];
};

// mj_makeConstraint: limit rows into `lim`, contact constants into the lane's LDS slots.  Returns
// the wave-uniform set of end-sphere SLOTS (bit s) that touch the plane on any lane of the wave.
// `own`: the same bits, per LANE -- the slots that touch on this lane (its D != 0).
template <int KL, typename T, typename V, typename Cx>
EPA_HD unsigned MakeConstraint(const CheetahModel<T>& m, Cx& cx, const Pos<V>& p, const V* q,
                               const V* v, LimitRows<V>& lim, typename LaneTypes<V>::U& own) {
  const T kMinVal = T(1e-15);
  own = LaneTypes<V>::Fill(0u);
  static_for<0, 3>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const V qq = q[j + 3];
    const V dlo = qq - cx.C(kTLo + j), dhi = cx.C(kTHi + j) - qq;
    const auto blo = dlo < V(0), bhi = dhi < V(0);
    const V sgn = Sel(blo, V(1), Sel(bhi, V(-1), V(0)));
    const V dist = Sel(blo, dlo, Sel(bhi, dhi, V(0)));
    const V imp = ImpedanceV(m.lim_d0, m.lim_dmax, m.lim_width, dist);
    // R = max(mjMINVAL, (1 - imp) / imp * diagApprox), D = 1 / R
    const V num = (V(1) - imp) * cx.C(kTInvw + j);
    const V Dj = Sel(num < V(kMinVal) * imp, V(T(1) / kMinVal), imp / num);
    lim.sgn[j] = sgn;
    lim.D[j] = Sel(sgn != V(0), Dj, V(0));
    lim.aref[j] = -V(m.lim_B) * (sgn * v[j + 3]) - V(m.lim_K) * imp * dist;
  });
  unsigned ends = 0;
  static_for<0, Grp<KL>::kEnds>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int b = Grp<KL>::SlotBody(s);
    const V ex = cx.C(kTEx + s), ez = cx.C(Tab<KL>::kEz + s), er = cx.C(Tab<KL>::kEr + s);
    const V wx = p.px[b] + p.cs[b] * ex + p.sn[b] * ez;
    const V wz = p.pz[b] - p.sn[b] * ex + p.cs[b] * ez;
    const V dist = wz - er;  // unused slots of a model carry er = -1e30: never touch
    V D = V(0), an = V(0), ax = V(0);
    const V cpx = wx, cpz = V(0.5) * dist;
    const auto touch = dist < V(m.con_margin);
    if (AnyWave(touch)) {
      ends |= 1u << s;
      MaskSet(own, touch, s);
      V vn = V(0), vx = V(0);
      ForChainCols<b>(p, cpx, cpz, [&](auto jc, V jn, V jx) {
        constexpr int j = decltype(jc)::value;
        vn += jn * v[j];
        vx += jx * v[j];
      });
      const V r = dist - V(m.con_margin);
      const V imp = ImpedanceV(m.con_d0, m.con_dmax, m.con_width, r);
      V diag, d2mu;  // diagApprox (pyramidal) = tran (1 + mu^2); R_py = 2 mu^2 R
      if constexpr (b == 0) {
        diag = V(m.body_invw[0] * (T(1) + m.bmu[0] * m.bmu[0]));
        d2mu = V(T(1) / (T(2) * m.bmu[0] * m.bmu[0]));
      } else {
        diag = cx.C(kTDiag + b);
        d2mu = cx.C(kTD2mu + b);
      }
      const V mu = MuOf<b, T, V>(m, cx);
      const V num = (V(1) - imp) * diag;  // R = max(mjMINVAL, num / imp)
      const V invR = Sel(num < V(kMinVal) * imp, V(T(1) / kMinVal), imp / num);
      D = Sel(touch, invR * d2mu, V(0));
      an = Sel(touch, -V(m.con_B) * vn - V(m.con_K) * imp * r, V(0));
      ax = Sel(touch, V(m.con_B) * mu * vx, V(0));
    }
    // A slot's numbers are only read by lanes that visit it, i.e. whose own bit is set -- then some lane touches and
    // the wave is in here; slot 0 is also what a lane whose set has run out reads (weight 0): always written.
    if (s == 0 || (ends & (1u << s)) != 0) {
      cx.Lds(s * kSlotsPerEnd + 0) = cpx;
      cx.Lds(s * kSlotsPerEnd + 1) = cpz;
      cx.Lds(s * kSlotsPerEnd + 2) = an;
      cx.Lds(s * kSlotsPerEnd + 3) = ax;
      cx.Lds(s * kSlotsPerEnd + 4) = D;
    }
  });
  // body-body capsule pairs of the Hopper (mjc_CapsuleCapsule: closest points of the two axis segments,
  // then sphere-sphere; the arithmetic of mj_cheetah.hip.h::CheetahMakeConstraint with selects for its
  // branches): one frictionless row each (condim 1)
  if constexpr (Grp<KL>::kPairs) {
    static_for<0, kNPair>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int b1 = PairBody1(k), b2 = PairBody2(k);
      constexpr int s1 = 2 * b1, s2 = 2 * b2;  // the end-sphere slots of the two capsules
      auto end_pos = [&](auto bc, int sl, V* x, V* z) {
        constexpr int b = decltype(bc)::value;
        const V ex = cx.C(kTEx + sl), ez = cx.C(Tab<KL>::kEz + sl);
        *x = p.px[b] + p.cs[b] * ex + p.sn[b] * ez;
        *z = p.pz[b] - p.sn[b] * ex + p.cs[b] * ez;
      };
      V ax1, az1, bx1, bz1, ax2, az2, bx2, bz2;
      end_pos(IC<b1>{}, s1, &ax1, &az1);
      end_pos(IC<b1>{}, s1 + 1, &bx1, &bz1);
      end_pos(IC<b2>{}, s2, &ax2, &az2);
      end_pos(IC<b2>{}, s2 + 1, &bx2, &bz2);
      const V c1x = V(0.5) * (ax1 + bx1), c1z = V(0.5) * (az1 + bz1);
      const V c2x = V(0.5) * (ax2 + bx2), c2z = V(0.5) * (az2 + bz2);
      const V h1x = V(0.5) * (ax1 - bx1), h1z = V(0.5) * (az1 - bz1);
      const V h2x = V(0.5) * (ax2 - bx2), h2z = V(0.5) * (az2 - bz2);
      const V dfx = c1x - c2x, dfz = c1z - c2z;
      const V ma = h1x * h1x + h1z * h1z, mb = -(h1x * h2x + h1z * h2z), mc = h2x * h2x + h2z * h2z;
      const V r1 = cx.C(Tab<KL>::kEr + s1), r2 = cx.C(Tab<KL>::kEr + s2);
      // Broad phase (round 5; the three narrow phases were a third of the Hopper's set-up): two points of the
      // segments are at least |c1 - c2| - |h1| - |h2| apart, the pair touches only below rr = r1 + r2 + margin, and
      // (|h1| + |h2| + rr)^2 <= 3 (|h1|^2 + |h2|^2 + rr^2): beyond that no lane can touch, the pair's bit of `ends`
      // stays clear and its LDS slots are never read -- what the narrow phase would have found, without running it.
      const V rr = r1 + r2 + V(m.con_margin);
      const auto near = dfx * dfx + dfz * dfz <= V(3) * (ma + mc + rr * rr);
      if (!AnyWave(near)) return;
      const V u = -(h1x * dfx + h1z * dfz), w = h2x * dfx + h2z * dfz;
      const V det = ma * mc - mb * mb;
      auto clamp1 = [](V x) { return Sel(x > V(1), V(1), Sel(x < V(-1), V(-1), x)); };
      const auto par = Abs(det) < V(kMinVal);
      const V den = Sel(par, V(1), det);
      V x1 = (mc * u - mb * w) / den;
      V x2 = (ma * w - mb * u) / den;
      {
        const auto hi1 = x1 > V(1), lo1 = x1 < V(-1);
        x2 = Sel(hi1, (w - mb) / mc, Sel(lo1, (w + mb) / mc, x2));
        x1 = clamp1(x1);
        const auto hi2 = x2 > V(1), lo2 = x2 < V(-1);
        const V x1b = clamp1(Sel(hi2, (u - mb) / ma, (u + mb) / ma));
        x1 = Sel(hi2 | lo2, x1b, x1);
        x2 = clamp1(x2);
      }
      {
        const V amb = Abs(mb);
        V lo = (u - amb) / ma, hi = (u + amb) / ma;
        lo = Sel(lo < V(-1), V(-1), lo);
        hi = Sel(hi > V(1), V(1), hi);
        const V xp1 = Sel(lo <= hi, V(0.5) * (lo + hi), Sel(lo > V(1), V(1), V(-1)));
        const V xp2 = clamp1((w - mb * xp1) / mc);
        x1 = Sel(par, xp1, x1);
        x2 = Sel(par, xp2, x2);
      }
      const V p1x = c1x + h1x * x1, p1z = c1z + h1z * x1;
      const V p2x = c2x + h2x * x2, p2z = c2z + h2z * x2;
      const V ddx = p2x - p1x, ddz = p2z - p1z;
      const V cd = SqrtV(ddx * ddx + ddz * ddz);
      const V dist = cd - r1 - r2;
      const auto touch = dist < V(m.con_margin);
      V nx = V(1), nz = V(0), ccx = V(0), ccz = V(0), aref = V(0), D = V(0);
      if (AnyWave(touch)) {
        ends |= 1u << (kPairEndBit + k);
        MaskSet(own, touch, kPairEndBit + k);
        const auto tiny = cd < V(kMinVal);
        const V inv = V(1) / Sel(tiny, V(1), cd);
        nx = Sel(tiny, V(1), ddx * inv);
        nz = Sel(tiny, V(0), ddz * inv);
        ccx = p1x + nx * (r1 + V(0.5) * dist);
        ccz = p1z + nz * (r1 + V(0.5) * dist);
        // relative normal velocity: only the hinges between the two bodies contribute
        V vel = V(0);
        static_for<b1 + 3, b2 + 3>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int jb = LDofBody(j);
          vel += (nx * (ccz - p.pz[jb]) - nz * (ccx - p.px[jb])) * v[j];
        });
        const V r = dist - V(m.con_margin);
        const V imp = ImpedanceV(m.con_d0, m.con_dmax, m.con_width, r);
        const V num = (V(1) - imp) * V(m.body_invw[b1] + m.body_invw[b2]);  // condim 1: tran1 + tran2
        const V invR = Sel(num < V(kMinVal) * imp, V(T(1) / kMinVal), imp / num);
        D = Sel(touch, invR, V(0));
        aref = Sel(touch, -V(m.con_B) * vel - V(m.con_K) * imp * r, V(0));
      }
      cx.Lds(PairBase<KL>() + k * kSlotsPerPair + 0) = nx;
      cx.Lds(PairBase<KL>() + k * kSlotsPerPair + 1) = nz;
      cx.Lds(PairBase<KL>() + k * kSlotsPerPair + 2) = ccx;
      cx.Lds(PairBase<KL>() + k * kSlotsPerPair + 3) = ccz;
      cx.Lds(PairBase<KL>() + k * kSlotsPerPair + 4) = aref;
      cx.Lds(PairBase<KL>() + k * kSlotsPerPair + 5) = D;
    });
  }
  return WaveUniform(ends);
}

// Jacobian entries of pair contact K (Hopper): f(j, J_j) for the hinges between its two bodies -- the dofs
// shared by both chains move both bodies alike and drop out of the relative normal velocity
template <int KL, int K, typename V, typename Cx, typename F>
EPA_HD void ForPairCols(const Pos<V>& p, Cx& cx, F&& f) {
  constexpr int b1 = PairBody1(K), b2 = PairBody2(K);
  const V nx = cx.Lds(PairBase<KL>() + K * kSlotsPerPair + 0), nz = cx.Lds(PairBase<KL>() + K * kSlotsPerPair + 1);
  const V ccx = cx.Lds(PairBase<KL>() + K * kSlotsPerPair + 2), ccz = cx.Lds(PairBase<KL>() + K * kSlotsPerPair + 3);
  static_for<b1 + 3, b2 + 3>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int jb = LDofBody(j);
    f(jc, nx * (ccz - p.pz[jb]) - nz * (ccx - p.px[jb]));
  });
}

// One pass over the lane's constraint rows at acceleration `a`: accumulates THIS LANE's part of
// J^T f into gc (6) and, if kHess, of J^T D_active J into Hc (21); the lane's active-row mask in
// `mask`.  The caller sums the torso entries over the group.
// `vis`: the lane's set of end-sphere slots to visit (its own touching slots, or none if its env has finished);
// `ends`: the wave-uniform set MakeConstraint returned (only its body-pair bits are used here).
template <int KL, bool kHess, typename T, typename V, typename Cx, typename U>
EPA_HD void RowsPass(const CheetahModel<T>& m, Cx& cx, const Pos<V>& p, const LimitRows<V>& lim,
                     unsigned ends, const U& vis, const V* a, V* gc, V* Hc, U& mask) {
  static_for<0, 3>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const V jar = lim.sgn[j] * a[j + 3] - lim.aref[j];
    const V w = Sel(jar < V(0), lim.D[j], V(0));  // lim.D == 0 where the limit is not violated
    gc[j + 3] += lim.sgn[j] * w * jar;
    if constexpr (kHess) Hc[Tri(j + 3, j + 3)] += w;
    MaskSetNZ(mask, w, j);
  });
  if constexpr (Grp<KL>::kPairs) {  // frictionless body-body rows (Hopper)
    if ((ends >> kPairEndBit) != 0) {
      static_for<0, kNPair>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (ends & (1u << (kPairEndBit + k))) {
          const V aref = cx.Lds(PairBase<KL>() + k * kSlotsPerPair + 4);
          const V D = cx.Lds(PairBase<KL>() + k * kSlotsPerPair + 5);
          V ja = V(0);
          ForPairCols<KL, k>(p, cx, [&](auto jc, V J) { ja += J * a[decltype(jc)::value]; });
          const V jar = ja - aref;
          const V w = Sel(jar < V(0), D, V(0));  // D == 0 on lanes whose pair does not touch
          MaskSetNZ(mask, w, kPairMaskBit + k);
          ForPairCols<KL, k>(p, cx, [&](auto ic, V Ji) {
            constexpr int i = decltype(ic)::value;
            gc[i] += Ji * (w * jar);
            if constexpr (kHess) {
              const V wi = w * Ji;
              ForPairCols<KL, k>(p, cx, [&](auto jc2, V Jk) {
                constexpr int kk = decltype(jc2)::value;
                if constexpr (kk >= i) Hc[Tri(i, kk)] += wi * Jk;
              });
            }
          });
        }
      });
    }
  }
  // The lane's OWN touching slots, one per trip of the loop (round 5; rounds 3-4 looped over the wave-uniform UNION of
  // the touching slots with a body-specialised visit: 2.5 visits per pass where the busiest lane has 1.6 slots --
  // tools/lg_desync_sim.py).  What differs between lanes -- the slot, its body and so the depth of the chain -- is
  // data: LDS / table reads at per-lane addresses, the columns of hinges beyond the slot's body are selected to zero
  // (exact zeros: the sums they enter are unchanged), a lane whose set has run out visits with weight D = 0.
  // (A software pipeline by one slot measured -0.4 % in round 4, profiles/r4f_prefetch_ab.txt.)
  U rem = vis;
  while (AnyWave(AnySlot(rem))) {
    const auto has = AnySlot(rem);
    const U sl = PopSlot(rem);
    const U body = SlotBodyOf<KL>(sl);
    const V cpx = cx.LdsL(sl, kSlotsPerEnd, 0), cpz = cx.LdsL(sl, kSlotsPerEnd, 1);
    const V an = cx.LdsL(sl, kSlotsPerEnd, 2), ax = cx.LdsL(sl, kSlotsPerEnd, 3);
    const V D = Sel(has, cx.LdsL(sl, kSlotsPerEnd, 4), V(0));
    const V mu = cx.CL(kTMu, body);  // (the table also carries the torso's)
    // Jacobian columns of the contact point: dofs 0, 1 (slides), 2 (torso hinge), then the leg's hinges up to the body
    V jn[kLV], jx[kLV];
    jn[0] = V(0), jx[0] = V(1);
    jn[1] = V(1), jx[1] = V(0);
    jn[2] = -(cpx - p.px[0]), jx[2] = cpz - p.pz[0];
    static_for<3, kLV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const auto in = AtLeast(body, (unsigned)(j - 2));
      jn[j] = Sel(in, -(cpx - p.px[j - 2]), V(0));
      jx[j] = Sel(in, cpz - p.pz[j - 2], V(0));
    });
    // (the two slides' columns are the unit vectors: jn = (0, 1, ...), jx = (1, 0, ...); written out, because
    // 0 x a[j] is an FMA the compiler must keep)
    V jna = a[1], jxa = a[0];
    static_for<2, kLV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      jna += jn[j] * a[j];
      jxa += jx[j] * a[j];
    });
    if constexpr (kHess) {  // the per-iteration pass: kept for the line search
      cx.LdsLStore(sl, Grp<KL>::kCachePerEnd, CacheBase<KL>() + 0, jna, has);
      cx.LdsLStore(sl, Grp<KL>::kCachePerEnd, CacheBase<KL>() + 1, jxa, has);
    }
    // rows: 2 x (Jn), (Jn - mu Jx), (Jn + mu Jx); D == 0 for lanes not in contact, which zeroes every weight below
    const V jar1 = jna - an;
    const V jar2 = jna - mu * jxa - (an + ax);
    const V jar3 = jna + mu * jxa - (an - ax);
    const V w1 = Sel(jar1 < V(0), V(2) * D, V(0));
    const V w2 = Sel(jar2 < V(0), D, V(0));
    const V w3 = Sel(jar3 < V(0), D, V(0));
    MaskSetNZAt(mask, w1, sl, 3);
    MaskSetNZAt(mask, w2, sl, 4);
    MaskSetNZAt(mask, w3, sl, 5);
    const V gn = w1 * jar1 + w2 * jar2 + w3 * jar3;  // coefficient of Jn
    const V gx = mu * (w3 * jar3 - w2 * jar2);      // coefficient of Jx
    const V A = w1 + w2 + w3, Bc = mu * (w3 - w2), C = mu * mu * (w2 + w3);
    // Unconditional (lanes whose rows are inactive add zeros).  Rounds 3-5 skipped the update when no lane of the wave
    // had an active row in the visit; the branch made every accumulator -- 6 gradient and 21 Hessian doubles -- a
    // value with two reaching definitions, and the compiler copied all 27 at the head of the slot loop and back at
    // its end: 54 v_mov_b64 of the loop's 258 instructions.  HalfCheetah 4.13e8 -> 4.33e8, Walker2d +6 %, Hopper +6 %
    // (profiles/r6i_lg_rowpass_ab.txt).
    {
      static_for<0, kLV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        // dofs 0 and 1 (the slides): jn = 0, jx = 1 and jn = 1, jx = 0
        if constexpr (i == 0) {
          gc[0] += gx;
        } else if constexpr (i == 1) {
          gc[1] += gn;
        } else {
          gc[i] += jn[i] * gn + jx[i] * gx;
        }
        if constexpr (kHess) {
          V ui, wi;
          if constexpr (i == 0) {
            ui = Bc, wi = C;
          } else if constexpr (i == 1) {
            ui = A, wi = Bc;
          } else {
            ui = A * jn[i] + Bc * jx[i], wi = Bc * jn[i] + C * jx[i];
          }
          static_for<i, kLV>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k == 0) {
              Hc[Tri(i, k)] += wi;
            } else if constexpr (k == 1) {
              Hc[Tri(i, k)] += ui;
            } else {
              Hc[Tri(i, k)] += ui * jn[k] + wi * jx[k];
            }
          });
        }
      });
    }
  }
}

// this lane's part of phi'(alpha), phi''(alpha) from its rows along `s` from `a`, and the lane's active-row mask
// AT a + alpha s (same bits as RowsPass)
template <int KL, typename T, typename V, typename Cx, typename U>
EPA_HD void LineEval(const CheetahModel<T>& m, Cx& cx, const Pos<V>& p, const LimitRows<V>& lim,
                     unsigned ends, const U& vis, const V* a, const V* s, V alpha, V* d1, V* d2, U& mask) {
  static_for<0, 3>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const V jar = lim.sgn[j] * a[j + 3] - lim.aref[j];
    const V jv = lim.sgn[j] * s[j + 3];
    const V x = jar + alpha * jv;
    const V w = Sel(x < V(0), lim.D[j], V(0));
    *d1 += w * x * jv;
    *d2 += w * jv * jv;
    MaskSetNZ(mask, w, j);
  });
  if constexpr (Grp<KL>::kPairs) {
    if ((ends >> kPairEndBit) != 0) {
      static_for<0, kNPair>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (ends & (1u << (kPairEndBit + k))) {
          const V aref = cx.Lds(PairBase<KL>() + k * kSlotsPerPair + 4);
          const V D = cx.Lds(PairBase<KL>() + k * kSlotsPerPair + 5);
          V ja = V(0), jv = V(0);
          ForPairCols<KL, k>(p, cx, [&](auto jc, V J) {
            ja += J * a[decltype(jc)::value];
            jv += J * s[decltype(jc)::value];
          });
          const V x = (ja - aref) + alpha * jv;
          const V cw = Sel(x < V(0), D, V(0));
          *d1 += cw * x * jv;
          *d2 += cw * jv * jv;
          MaskSetNZ(mask, cw, kPairMaskBit + k);
        }
      });
    }
  }
  // the lane's own touching slots, as in RowsPass
  U rem = vis;
  while (AnyWave(AnySlot(rem))) {
    const auto has = AnySlot(rem);
    const U sl = PopSlot(rem);
    const U body = SlotBodyOf<KL>(sl);
    const V cpx = cx.LdsL(sl, kSlotsPerEnd, 0), cpz = cx.LdsL(sl, kSlotsPerEnd, 1);
    const V an = cx.LdsL(sl, kSlotsPerEnd, 2), ax = cx.LdsL(sl, kSlotsPerEnd, 3);
    const V D = Sel(has, cx.LdsL(sl, kSlotsPerEnd, 4), V(0));
    // (RowsPass<true> at the same `a` left them for the slots it visited: a lane whose set has run out reads
    // slot 0's, which may never have been written -- 0 x NaN would poison the sums)
    const V jna = Sel(has, cx.LdsL(sl, Grp<KL>::kCachePerEnd, CacheBase<KL>() + 0), V(0));
    const V jxa = Sel(has, cx.LdsL(sl, Grp<KL>::kCachePerEnd, CacheBase<KL>() + 1), V(0));
    const V mu = cx.CL(kTMu, body);
    // s . (Jacobian columns of a point on the slot's body): torso dofs, then the hinges up to the body
    V jns = s[1] - (cpx - p.px[0]) * s[2];
    V jxs = s[0] + (cpz - p.pz[0]) * s[2];
    static_for<3, kLV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const auto in = AtLeast(body, (unsigned)(j - 2));
      jns -= Sel(in, cpx - p.px[j - 2], V(0)) * s[j];
      jxs += Sel(in, cpz - p.pz[j - 2], V(0)) * s[j];
    });
    const V jar1 = jna - an, jv1 = jns;
    const V jar2 = jna - mu * jxa - (an + ax), jv2 = jns - mu * jxs;
    const V jar3 = jna + mu * jxa - (an - ax), jv3 = jns + mu * jxs;
    const V x1 = jar1 + alpha * jv1, x2 = jar2 + alpha * jv2, x3 = jar3 + alpha * jv3;
    // D == 0 for lanes not in contact
    const V c1 = Sel(x1 < V(0), V(2) * D, V(0));
    const V c2 = Sel(x2 < V(0), D, V(0));
    const V c3 = Sel(x3 < V(0), D, V(0));
    *d1 += c1 * x1 * jv1 + c2 * x2 * jv2 + c3 * x3 * jv3;
    *d2 += c1 * jv1 * jv1 + c2 * jv2 * jv2 + c3 * jv3 * jv3;
    MaskSetNZAt(mask, c1, sl, 3);
    MaskSetNZAt(mask, c2, sl, 4);
    MaskSetNZAt(mask, c3, sl, 5);
  }
}

// ---- the arrow system ---------------------------------------------------------------------------
// A: packed local 6x6 = rows / columns [torso | my leg] of a 9x9 arrow matrix (torso block
// replicated and COMPLETE, coupling and leg block private).  In place: A = U U^T from the last dof
// up (tree order: no fill between the legs), diagonal stored INVERTED.
template <int KL, typename V>
EPA_HD void FactorArrow(V* A) {
  static_for_down<kLV, 3>([&](auto jc) {  // the leg's hinges, inside the lane
    constexpr int j = decltype(jc)::value;
    V s = A[Tri(j, j)];
    static_for<j + 1, kLV>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      s -= A[Tri(j, k)] * A[Tri(j, k)];
    });
    const V inv = Rsq(s);
    A[Tri(j, j)] = inv;
    static_for<0, j>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      V t = A[Tri(i, j)];
      static_for<j + 1, kLV>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        t -= A[Tri(i, k)] * A[Tri(j, k)];
      });
      A[Tri(i, j)] = t * inv;
    });
  });
  // torso block: minus both legs' Schur complements
  static_for<0, 3>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    static_for<0, j + 1>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      V c = A[Tri(i, 3)] * A[Tri(j, 3)] + A[Tri(i, 4)] * A[Tri(j, 4)] + A[Tri(i, 5)] * A[Tri(j, 5)];
      A[Tri(i, j)] -= SumLegs<KL>(c);
    });
  });
  static_for_down<3, 0>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    V s = A[Tri(j, j)];
    static_for<j + 1, 3>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      s -= A[Tri(j, k)] * A[Tri(j, k)];
    });
    const V inv = Rsq(s);
    A[Tri(j, j)] = inv;
    static_for<0, j>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      V t = A[Tri(i, j)];
      static_for<j + 1, 3>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        t -= A[Tri(i, k)] * A[Tri(j, k)];
      });
      A[Tri(i, j)] = t * inv;
    });
  });
}
// solve U U^T x = b in place (x: torso part replicated, leg part private)
template <int KL, typename V>
EPA_HD void SolveArrow(const V* U, V* x) {
  static_for_down<kLV, 3>([&](auto jc) {  // U y = b, the leg
    constexpr int j = decltype(jc)::value;
    V s = x[j];
    static_for<j + 1, kLV>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      s -= U[Tri(j, k)] * x[k];
    });
    x[j] = s * U[Tri(j, j)];
  });
  static_for_down<3, 0>([&](auto jc) {  // the torso rows see both legs
    constexpr int j = decltype(jc)::value;
    const V r = U[Tri(j, 3)] * x[3] + U[Tri(j, 4)] * x[4] + U[Tri(j, 5)] * x[5];
    V s = x[j] - SumLegs<KL>(r);
    static_for<j + 1, 3>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      s -= U[Tri(j, k)] * x[k];
    });
    x[j] = s * U[Tri(j, j)];
  });
  static_for<0, kLV>([&](auto jc) {  // U^T x = y: no cross-leg terms
    constexpr int j = decltype(jc)::value;
    V s = x[j];
    static_for<0, j>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      s -= U[Tri(i, j)] * x[i];
    });
    x[j] = s * U[Tri(j, j)];
  });
}
// y = A x for the arrow matrix
template <int KL, typename V>
EPA_HD void MulArrow(const V* A, const V* x, V* y) {
  static_for<0, 3>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    V t = V(0), l = V(0);
    static_for<0, 3>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      t += A[i <= j ? Tri(i, j) : Tri(j, i)] * x[j];
    });
    static_for<3, kLV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      l += A[Tri(i, j)] * x[j];
    });
    y[i] = t + SumLegs<KL>(l);
  });
  static_for<3, kLV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    V t = V(0);
    static_for<0, kLV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      t += A[i <= j ? Tri(i, j) : Tri(j, i)] * x[j];
    });
    y[i] = t;
  });
}
// a . b over the 9 dofs of the env (torso part counted once)
template <int KL, typename V>
EPA_HD V DotEnv(const V* a, const V* b) {
  const V t = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  const V l = a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
  return t + SumLegs<KL>(l);
}

// Newton trip of a forward pass from which a wave searches its lines exactly again (plg::Solve); the CPU harness
// builds the source a second time with 1, so that the fallback -- never reached in the benchmark -- is tested too
#ifndef EPA_LG_LS_EXACT_AFTER
#define EPA_LG_LS_EXACT_AFTER 8
#endif
constexpr int kLsExactAfter = EPA_LG_LS_EXACT_AFTER;

template <typename T>
struct SolverCfgLg {
  int max_iter;
  T gtol;  // stop when |grad| <= gtol * (1 + |qfrc_smooth|_inf)
};

// mj_fwdConstraint: exact Newton on the primal objective
//   1/2 (a-a0)^T M (a-a0) + sum_r 1/2 D_r min(0, J_r a - aref_r)^2
// started from qacc (= qacc_warmstart).  The iteration of mj_cheetah.hip.h::CheetahSolve -- finite termination
// (same active set after a full Newton step), wave-uniform control flow -- with a ONE-EVALUATION line search:
// phi'(1) and phi''(1) are evaluated at the full step; if phi'(1) vanishes the full step is taken (and, with the
// active set H was built with, it IS the minimiser), otherwise the step is 1 - phi'(1) / phi''(1), one Newton
// step of the 1-D problem, taken unverified.  Rounds 1-4 ran the 1-D Newton iteration to |phi'| <= 1e-10 |phi'(0)|
// (an exact search, 2.0 evaluations per trip of a wave = 36 % of its time): measured on 1.2 M forward passes of the
// benchmark (tools/lg_desync_sim.py ls, profiles/r5k_*), an env takes exactly as many Newton trips either way (6.74
// per env-step, histogram 1: 70.5 %, 2: 24.4 %, 3: 4.8 %, ...) -- whenever the full step misses, the active set
// changes along the line and the env needs the next trip's Hessian anyway -- and it ends on the same minimiser
// (finite termination decides, not the search).  MuJoCo's own Newton solver searches to a tolerance of 0.01.
// Safety net: from trip kLsExactAfter on (no env of the benchmark ever gets there) a wave falls back to the exact
// search, whose steps cannot increase the objective.  Outputs qacc, Ma = M qacc and the final gradient
// (qfrc_constraint = Ma - qfrc_smooth - grad); returns the env's Newton iterations.
// A lane visits its own touching slots only while its env is still live (with the wave-uniform slot union of round 4
// that was +2.9 % for HalfCheetah and -1.4 % for the RK4 models; with per-lane slot sets it costs nothing: all models).
template <int KL, typename T, typename V, typename Cx>
EPA_HD V Solve(const CheetahModel<T>& m, Cx& cx, const Pos<V>& p, const LimitRows<V>& lim,
               unsigned ends, const typename LaneTypes<V>::U& own, const V* qfrc_smooth,
               const SolverCfgLg<T>& cfg, V* qacc, V* Ma, V* grad) {
  using LT = LaneTypes<V>;
  using B = typename LT::B;
  using U = typename LT::U;
  V fs = V(0);
  static_for<0, kLV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const V x = Abs(qfrc_smooth[i]);
    fs = Sel(x > fs, x, fs);
  });
  fs = MaxLegs<KL>(fs);
  const V gstop = V(cfg.gtol) * (V(1) + fs);
  const V gstop2 = gstop * gstop;
  // rounding floor: once the gradient is this small and has stopped shrinking the iterate is as
  // converged as the arithmetic allows
  const V gfloor = V(T(1e-9)) * (V(1) + fs);
  const V gfloor2 = gfloor * gfloor;
  V prev_gn2 = V(-1);
  MulArrow<KL>(p.M, qacc, Ma);  // kept current incrementally: Ma += alpha * M s
  // (the first trip writes every env's gradient: this is for the Sel in it only)
  static_for<0, kLV>([&](auto ic) { grad[decltype(ic)::value] = V(0); });
  U prev_mask = LT::Fill(~0u);
  B full_step = LT::False();
  B live = LT::True();
  B at_min = LT::False();  // the env stopped at an exact minimiser (finite termination in the line search)
  V iter = V(0);
  for (int it = 0; it < cfg.max_iter; ++it) {
    // A lane visits its own touching slots while its env is still LIVE (after the first trip of a forward pass a
    // third of the envs are left, after the second one in twenty): the envs that have finished keep the gradient of
    // the trip they finished in
    // (their qacc and Ma are frozen, so it is the gradient at their result); H, the masks and the line-search
    // sums are only ever used for live envs.
    const U vis = SlotsWhere(own, it == 0 ? LT::True() : live);
    const B live0 = live;
    V H[kLTri], gc[kLV];
    static_for<0, kLTri>([&](auto kc) { H[decltype(kc)::value] = V(0); });
    static_for<0, kLV>([&](auto ic) { gc[decltype(ic)::value] = V(0); });
    U mask = LT::Fill(0u);
    EPA_LG_TICK(cx, 3);
    EPA_LG_HOST_ROWS();
    RowsPass<KL, true>(m, cx, p, lim, ends, vis, qacc, gc, H, mask);
    // group sums: the torso entries collect every lane of the env, the leg entries the leg's lanes
    static_for<0, kLV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j < 3) {
        gc[j] = SumEnv<KL>(gc[j]);
      } else {
        gc[j] = SumPar<KL>(gc[j]);
      }
      grad[j] = Sel(live0, (Ma[j] - qfrc_smooth[j]) + gc[j], grad[j]);
      static_for<0, j + 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (j < 3) {
          H[Tri(i, j)] = p.M[Tri(i, j)] + SumEnv<KL>(H[Tri(i, j)]);
        } else {
          H[Tri(i, j)] = p.M[Tri(i, j)] + SumPar<KL>(H[Tri(i, j)]);
        }
      });
    });
    const V gn2 = DotEnv<KL>(grad, grad);
    const B same = AllEnvOf<KL>(MaskSame(mask, prev_mask));
    // |grad| <= gstop; or finite termination: same active set after a full Newton step; or at the
    // rounding floor and no longer shrinking (x4)
    const B stop = (gn2 <= gstop2) | (full_step & same) |
                   ((prev_gn2 >= V(0)) & (gn2 <= gfloor2) & (gn2 >= V(0.0625) * prev_gn2));
    live = live & !stop;
    EPA_LG_TICK(cx, 2);
    if (!AnyWave(live)) break;
    EPA_LG_COUNT(cx, 0);
    iter += Sel(live, V(1), V(0));
    prev_gn2 = gn2;
    prev_mask = mask;
    V s[kLV];
    static_for<0, kLV>([&](auto ic) { s[decltype(ic)::value] = -grad[decltype(ic)::value]; });
    FactorArrow<KL>(H);
    SolveArrow<KL>(H, s);
    // line search on the convex piecewise-quadratic phi(alpha): one evaluation at the full step (see above)
    V Ms[kLV], r0[kLV];
    MulArrow<KL>(p.M, s, Ms);
    static_for<0, kLV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      r0[i] = Ma[i] - qfrc_smooth[i];
    });
    const V g1 = DotEnv<KL>(s, r0), g2 = DotEnv<KL>(s, Ms);
    V alpha = V(1), lo = V(0), hi = V(-1);
    full_step = LT::False();
    const V ls_tol = V(T(EPA_LG_LS_RTOL)) * Abs(g1);
    B searching = live;
    B exact = LT::False();
    EPA_LG_TICK(cx, 3);
    // one evaluation (at the full step) + one Newton step of the 1-D problem; the exact search only as the fallback
    const int ls_max = it < kLsExactAfter ? EPA_LG_LS_MAX : 24;
    for (int ls = 0; ls < ls_max; ++ls) {
      EPA_LG_COUNT(cx, 1);
      V d1p = V(0), d2p = V(0);
      U mask1 = LT::Fill(0u);
      LineEval<KL>(m, cx, p, lim, ends, vis, qacc, s, alpha, &d1p, &d2p, mask1);
      const V d1 = (g1 + alpha * g2) + SumEnv<KL>(d1p), d2 = g2 + SumEnv<KL>(d2p);
      const B hit = Abs(d1) <= ls_tol;
      // a full Newton step is exact for the active set H was built with: if the rows active at
      // a + s are the rows H was built with, a + s IS the minimiser (finite termination) and the
      // env is done without another pass over the rows
      if (ls == 0) {
        full_step = searching & hit;
        exact = full_step & AllEnvOf<KL>(MaskSame(mask1, mask));
      }
      searching = searching & !hit;
      const B neg = d1 < V(0);
      lo = Sel(searching & neg, alpha, lo);
      hi = Sel(searching & !neg, alpha, hi);
      V next = alpha - d1 / d2;
      next = Sel((hi >= V(0)) & ((next <= lo) | (next >= hi)), V(0.5) * (lo + hi), next);
      next = Sel(next <= V(0), V(0.5) * alpha, next);
      searching = searching & (next != alpha);
      alpha = Sel(searching, next, alpha);
      if (!AnyWave(searching)) {
        EPA_LG_HOST_TRIP(ls + 1);
        break;
      }
      if (ls + 1 == ls_max) EPA_LG_HOST_TRIP(ls + 1);
    }
    EPA_LG_TICK(cx, 4);
    static_for<0, kLV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      // (a select, not a zero step: s of a finished env is built from partial rows)
      qacc[i] = Sel(live, qacc[i] + alpha * s[i], qacc[i]);
      Ma[i] = Sel(live, Ma[i] + alpha * Ms[i], Ma[i]);
    });
    at_min = at_min | exact;
    live = live & !exact;
    if (!AnyWave(live)) break;
  }
  // At the minimiser the gradient vanishes (to rounding).  Set for every env that stopped there,
  // whether or not its wave went on iterating for other envs and re-evaluated this env's rows:
  // an env's result must not depend on its neighbours.
  static_for<0, kLV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    grad[i] = Sel(at_min, V(0), grad[i]);
  });
  if (AnyWave(live)) {  // iteration cap hit somewhere in the wave: refresh grad
    V gc[kLV];
    static_for<0, kLV>([&](auto ic) { gc[decltype(ic)::value] = V(0); });
    U mask = LT::Fill(0u);
    RowsPass<KL, false>(m, cx, p, lim, ends, SlotsWhere(own, LT::True()), qacc, gc, static_cast<V*>(nullptr), mask);
    static_for<0, kLV>([&](auto jc) {  // only for the envs that did hit the cap
      constexpr int j = decltype(jc)::value;
      V g;
      if constexpr (j < 3) {
        g = SumEnv<KL>(gc[j]);
      } else {
        g = SumPar<KL>(gc[j]);
      }
      grad[j] = Sel(live, (Ma[j] - qfrc_smooth[j]) + g, grad[j]);
    });
  }
  return iter;
}

// mj_forward: qacc at (q, v) under ctrl; `warm` is qacc_warmstart in/out; `p` keeps M for the caller
template <int KL, typename T, typename V, typename Cx>
EPA_HD V Forward(const CheetahModel<T>& m, const SolverCfgLg<T>& cfg, Cx& cx, const V* q, const V* v,
                 V* warm, const V* ctrl, Pos<V>& p, V* qacc, V* Ma, V* grad) {
  cx.Refresh();
  EPA_LG_TICK(cx, 0);
  EPA_LG_COUNT(cx, 2);
  Kinematics<KL>(m, cx, q, p);
  V qfrc_smooth[kLV];
  SmoothForces<KL>(m, cx, p, q, v, ctrl, qfrc_smooth);
  LimitRows<V> lim;
  typename LaneTypes<V>::U own;
  const unsigned ends = MakeConstraint<KL>(m, cx, p, q, v, lim, own);
  EPA_LG_HOST_ENDS(ends);
  EPA_LG_HOST_OWN(own);
  static_for<0, kLV>([&](auto ic) { qacc[decltype(ic)::value] = warm[decltype(ic)::value]; });
  EPA_LG_TICK(cx, 1);
  const V iters = Solve<KL>(m, cx, p, lim, ends, own, qfrc_smooth, cfg, qacc, Ma, grad);
  EPA_LG_TICK(cx, 2);
  static_for<0, kLV>([&](auto ic) { warm[decltype(ic)::value] = qacc[decltype(ic)::value]; });
  return iters;
}

// One mj_step, Euler with implicit joint damping (HalfCheetah).  q[0] is carried as a local offset
// (the caller accumulates the absolute root x in fp64); returns the env's Newton iterations.
template <int KL, typename T, typename V, typename Cx>
EPA_HD V StepEuler(const CheetahModel<T>& m, const SolverCfgLg<T>& cfg, Cx& cx, V* q, V* v, V* warm,
                   const V* ctrl) {
  Pos<V> p;
  V qacc[kLV], Ma[kLV], grad[kLV];
  const V iters = Forward<KL>(m, cfg, cx, q, v, warm, ctrl, p, qacc, Ma, grad);
  // (M + h diag(damping)) qacc_d = qfrc_smooth + qfrc_constraint = Ma - grad
  V rhs[kLV];
  static_for<0, kLV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    rhs[i] = Ma[i] - grad[i];
    if constexpr (i >= 3) p.M[Tri(i, i)] += V(m.timestep) * cx.C(kTDamp + i - 3);
  });
  FactorArrow<KL>(p.M);
  SolveArrow<KL>(p.M, rhs);
  static_for<0, kLV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v[i] += V(m.timestep) * rhs[i];
    q[i] += V(m.timestep) * v[i];
  });
  EPA_LG_TICK(cx, 5);
  return iters;
}

// One mj_step with integrator RK4 (mj_RungeKutta(4)), the Walker2d setting
// (walker2d_envpool.xml:29): four forward evaluations per step, explicit damping (no eulerdamp).
template <int KL, typename T, typename V, typename Cx>
EPA_HD V StepRK4(const CheetahModel<T>& m, const SolverCfgLg<T>& cfg, Cx& cx, V* q, V* v, V* warm,
                 const V* ctrl) {
  const V h = V(m.timestep);
  V q0[kLV], v0[kLV], qs[kLV], vs[kLV], F[kLV], dq[kLV], dv[kLV];
  static_for<0, kLV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    q0[i] = qs[i] = q[i];
    v0[i] = vs[i] = v[i];
    dq[i] = dv[i] = V(0);
  });
  // stages 1..4 through ONE instance of the forward pass (a rolled loop):
  // X_i = X_0 + h a_i (Xv_{i-1}, F_{i-1}), a = 1/2, 1/2, 1; weights 1/6, 1/3, 1/3, 1/6
  V it = V(0);
#if defined(__clang__)
#pragma nounroll
#endif
  for (int stage = 0; stage < 4; ++stage) {
    Pos<V> p;
    V Ma[kLV], grad[kLV];
    it += Forward<KL>(m, cfg, cx, qs, vs, warm, ctrl, p, F, Ma, grad);
    const V bw = V((stage == 0 || stage == 3) ? T(1.0 / 6.0) : T(1.0 / 3.0));
    const V a = V(stage == 2 ? T(1) : T(0.5));
    static_for<0, kLV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dq[i] += bw * vs[i];
      dv[i] += bw * F[i];
      qs[i] = q0[i] + h * (a * vs[i]);  // the state of the next stage (unused after stage 4)
      vs[i] = v0[i] + h * a * F[i];
    });
  }
  static_for<0, kLV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v[i] = v0[i] + h * dv[i];
    q[i] = q0[i] + h * dq[i];
  });
  return it;
}

}  // namespace plg
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_PLANAR_LG_HIP_H_
