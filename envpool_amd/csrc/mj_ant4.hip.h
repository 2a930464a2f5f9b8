// K3b' — Ant `mj_step` with ONE ENV SPLIT OVER FOUR LANES (one lane per leg).
//
// Same arithmetic as mj_ant.hip.h (MuJoCo 3.6.0's mj_step for
// third_party/mujoco_gym_xml_patches/ant_envpool.xml, called from
// envpool/mujoco/gym/mujoco_env.h:137-148), re-laid
// out for the machine instead of for one thread per env:
//
//  * the Ant is a torso with four identical 2-dof legs, so M and the Newton
//    Hessian are "arrow" matrices: a 6x6 torso block, four 6x2 couplings and four
//    2x2 leg blocks.  Lane l of a quad owns leg l: its two bodies, two dofs, six
//    capsule end spheres, its 6x2 + 2x2 blocks and a PARTIAL torso block; the
//    torso quantities (pose, velocity, 6-vectors, the reduced 6x6) are replicated.
//    A lane therefore carries an 8x8 packed matrix (36 numbers) instead of the
//    81 structural non-zeros of the 14x14, an 8-vector instead of a 14-vector,
//    and the kernel needs no scratch memory (the one-env-per-lane version moved
//    3 GB of spills per launch, profiles/archive/r1g_ant_f64_summary.md);
//  * the legs differ only by mirror signs (sx, sy, ankle-axis sign, ankle range),
//    which are per-lane values: all four legs execute the SAME instructions, the
//    9-way body switch of mj_ant.hip.h becomes a 3-way switch on the sphere's link
//    (stub / leg / ankle capsule) and the code is a third of the size;
//  * leg elimination is local: U U^T from the last dof up eliminates ankle and
//    hip inside the lane, the four Schur complements are summed over the quad
//    (DPP butterflies, mj_quad.hip.h) and every lane factors the same 6x6.
//    Reductions per Newton iteration: 7 + 27 + 8 numbers, + 2 per line-search
//    evaluation;
//  * a wave holds 16 envs: N=65536 is 4096 waves instead of 1024, which is what
//    fills the tail of a launch;
//  * the wave-uniform contact mask is over the 7 sphere classes (6 per leg +
//    torso sphere), visited by a scalar loop; lanes whose sphere is outside the
//    margin carry D = 0.  No lane-divergent control flow anywhere.
// The same source runs on the host with V = Q4<T> (tests/cpu_harness).
#ifndef ENVPOOL_AMD_CSRC_MJ_ANT4_HIP_H_
#define ENVPOOL_AMD_CSRC_MJ_ANT4_HIP_H_

// Stage timers of the diagnostic build (-DEPA_ANT_TIMERS, tools/build_ant_timers.sh; never in the product
// library): EPA_ANT_TICK(K) books the wave's cycles since its previous tick to category K (mujoco_ant.hip:
// 0 unit overhead: ticket, state loads / stores, outputs; 1 front end: kinematics, inertias, smooth forces, limit
// rows; 2 contact set-up; 3 pass over the rows + quad sums + stop tests; 4 factor / solve; 5 line search (M s
// and the evaluation); 6 RK4 stage updates + position integration; 7 waiting for the chunk's previous unit),
// EPA_ANT_COUNT(K) counts wave-level events (0 Newton trips, 1 forward passes, 2 units).
#ifndef EPA_ANT_TICK
#define EPA_ANT_TICK(K) ((void)0)
#define EPA_ANT_COUNT(K) ((void)0)
#define EPA_ANT_CLASSES(sph, own) ((void)0)
#endif

#include "mj_ant.hip.h"
#include "mj_quad.hip.h"

namespace epa {
namespace mj {
namespace ant4 {

using ant::AntModel;
using ant::Cross;
using ant::CrossForce;
using ant::CrossMotion;
using ant::Dot;
using ant::In10;
using ant::Mat3;
using ant::MulInert;
using ant::Sp6;
using ant::Vec3;

constexpr int kL = 8;  // local dofs of a lane: torso 0..5 (replicated), hip 6, ankle 7
EPA_HD constexpr int Tri(int i, int j) { return j * (j + 1) / 2 + i; }  // i <= j
constexpr int kLTri = kL * (kL + 1) / 2;  // 36: [torso 21 | hip col 7 | ankle col 8]
constexpr int kTTri = 21;                 // packed 6x6 torso block = entries 0..20

// sphere classes of a lane: w = 0..5 the leg's capsule ends in mj_ant.hip.h's order
// (stub: aux origin, torso origin; leg capsule: foot origin, aux origin; ankle
// capsule: tip, foot origin), w = 6 the torso sphere (first lane of the quad only)
constexpr int kNW = 7;
EPA_HD constexpr int LinkOf(int w) { return w == 6 ? 0 : w / 2; }  // 0 torso-fixed, 1 aux, 2 foot

// LDS block of a wave.  Slots [0, kQuadSlots) hold what is REPLICATED over a quad -- stored
// once per quad, [slot][quad], read as a broadcast: the torso block of M (the first 21
// entries of the packed 8x8), the torso origin, its axes, the robot COM.  The rest is
// lane-private, [slot][lane]: the lane's columns of M (15), its leg geometry (12), and the
// per-pass constants of its contact candidates (4 per sphere class, SetupContacts).
// fp64: 36 x 16 x 8 B + 55 x 64 x 8 B = 32.8 KB per wave.
constexpr int kQuadSlots = 36, kLaneSlots = 55;
constexpr int kSlotPos0 = 21, kSlotRot = 24, kSlotCom = 33;                       // quad
constexpr int kSlotPosA = 51, kSlotPosF = 54, kSlotTip = 57, kSlotAnk = 60;       // lane
constexpr int kSlotCache = 63;                                                    // lane, 4 x kNW
constexpr int kSlots = kQuadSlots + kLaneSlots;
EPA_HD constexpr int MSlot(int packed) { return packed < kTTri ? packed : kQuadSlots + (packed - kTTri); }
EPA_HD constexpr int CenterSlot(int w) {
  return w == 0 ? kSlotPosA
                : w == 1 ? kSlotPos0
                         : w == 2 ? kSlotPosF
                                  : w == 3 ? kSlotPosA : w == 4 ? kSlotTip : w == 5 ? kSlotPosF : kSlotPos0;
}
// element offset of a slot inside the wave's LDS block (device accessor; the slot number is
// a compile-time constant or wave uniform, so this is scalar arithmetic)
EPA_HD constexpr int LdsOffset(int slot, int lane) {
  return slot < kQuadSlots ? slot * 16 + (lane >> 2)
                           : kQuadSlots * 16 + (slot - kQuadSlots) * 64 + lane;
}
constexpr int kLdsElems = kQuadSlots * 16 + kLaneSlots * 64;

// the mirror signs of a lane's leg and the two constants that differ between legs
template <typename V, typename B>
struct Leg {
  V sx, sy, sxy;  // aux/foot/tip offsets are (sx a, sy a, 0); products of inertia carry sx sy
  V axs;          // ankle axis = (axs k, k, 0), k = 1/sqrt 2
  V alo, ahi;     // ankle range
  B first;        // this lane also owns the torso sphere
};

template <typename V>
struct Rows {  // joint-limit rows of the lane's two hinges
  V sgn[2], aref[2], D[2];
};

template <typename V>
EPA_HD V Abs(V x) {
  return Sel(x < V(0), -x, x);
}
template <typename V>
EPA_HD V Max(V a, V b) {
  return Sel(a > b, a, b);
}
template <typename T, typename V>
EPA_HD V ImpedanceV(T d0, T dmax, T width, V r) {  // mj::Impedance with selects
  V x = Abs(r) * V(T(1) / width);
  V y = Sel(x <= V(0.5), V(2) * x * x, V(1) - V(2) * (V(1) - x) * (V(1) - x));
  return Sel(x >= V(1), V(dmax), V(d0) + y * V(dmax - d0));
}
template <typename V>
EPA_HD void NormalizeQuatV(V* q) {
  V n = Sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const auto tiny = n < V(1e-15);
  const V inv = V(1) / Sel(tiny, V(1), n);
  q[0] = Sel(tiny, V(1), q[0] * inv);
  q[1] = Sel(tiny, V(0), q[1] * inv);
  q[2] = Sel(tiny, V(0), q[2] * inv);
  q[3] = Sel(tiny, V(0), q[3] * inv);
}
// body inertia (xx yy zz xy; xz = yz = 0 for every Ant body) rotated into the world
// frame and shifted to offset d = xipos - com (mj_comPos)
template <typename V>
EPA_HD In10<V> CinertOf(V ixx, V iyy, V izz, V ixy, V mass, const Mat3<V>& R, Vec3<V> d) {
  const V* r = R.m;
  // RI = R * I
  const V a0 = r[0] * ixx + r[1] * ixy, a1 = r[0] * ixy + r[1] * iyy, a2 = r[2] * izz;
  const V b0 = r[3] * ixx + r[4] * ixy, b1 = r[3] * ixy + r[4] * iyy, b2 = r[5] * izz;
  const V c0 = r[6] * ixx + r[7] * ixy, c1 = r[6] * ixy + r[7] * iyy, c2 = r[8] * izz;
  const V d2 = Dot(d, d);
  In10<V> c;
  c.v[0] = a0 * r[0] + a1 * r[1] + a2 * r[2] + mass * (d2 - d.x * d.x);
  c.v[1] = b0 * r[3] + b1 * r[4] + b2 * r[5] + mass * (d2 - d.y * d.y);
  c.v[2] = c0 * r[6] + c1 * r[7] + c2 * r[8] + mass * (d2 - d.z * d.z);
  c.v[3] = a0 * r[3] + a1 * r[4] + a2 * r[5] - mass * d.x * d.y;
  c.v[4] = a0 * r[6] + a1 * r[7] + a2 * r[8] - mass * d.x * d.z;
  c.v[5] = b0 * r[6] + b1 * r[7] + b2 * r[8] - mass * d.y * d.z;
  c.v[6] = mass * d.x;
  c.v[7] = mass * d.y;
  c.v[8] = mass * d.z;
  c.v[9] = mass;
  return c;
}
template <typename V>
EPA_HD Vec3<V> Sum4v(Vec3<V> a) {
  return {Sum4(a.x), Sum4(a.y), Sum4(a.z)};
}
// R * (x, y, 0)
template <typename V>
EPA_HD Vec3<V> MulXY(const Mat3<V>& R, V x, V y) {
  return {R.m[0] * x + R.m[1] * y, R.m[3] * x + R.m[4] * y, R.m[6] * x + R.m[7] * y};
}

// ---- dense 6x6 U U^T (torso block after the legs are eliminated) -----------------
template <typename V>
EPA_HD void FactorTorso(V* A) {  // packed upper, in place; diagonal returned inverted
  static_for_down<6, 0>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    V s = A[Tri(j, j)];
    static_for<j + 1, 6>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      s -= A[Tri(j, k)] * A[Tri(j, k)];
    });
    const V inv = Rsq(s);
    A[Tri(j, j)] = inv;
    static_for<0, j>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      V t = A[Tri(i, j)];
      static_for<j + 1, 6>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        t -= A[Tri(i, k)] * A[Tri(j, k)];
      });
      A[Tri(i, j)] = t * inv;
    });
  });
}

// ---- geometry accessor over the lane's LDS block ---------------------------------
template <typename V, typename Lds>
struct Geo {
  Lds& lds;
  EPA_HD Vec3<V> At(int base) const { return {lds(base), lds(base + 1), lds(base + 2)}; }
  EPA_HD Vec3<V> Pos0() const { return At(kSlotPos0); }
  EPA_HD Vec3<V> Rot(int k) const { return At(kSlotRot + 3 * k); }
  EPA_HD Vec3<V> PosA() const { return At(kSlotPosA); }
  EPA_HD Vec3<V> PosF() const { return At(kSlotPosF); }
  EPA_HD Vec3<V> Ank() const { return At(kSlotAnk); }
};

// One contact candidate of the lane: sphere class w (wave uniform) on link K.
// mj_collision (plane-sphere) + mj_instantiateContact + mj_makeImpedance, re-derived
// from the body pose each time it is needed (as in mj_ant.hip.h).  C[k] are the
// non-trivial columns of the 3 x 8 point Jacobian: rot 0..2, then hip, ankle for
// K >= 1, 2 (the translational columns are the unit vectors).
template <typename V>
struct Contact {
  V an, ay, ax, D;
};
// the non-trivial Jacobian columns of contact point cp on link K
template <int K, typename V, typename G>
EPA_HD void ContactCols(const G& g, Vec3<V> cp, Vec3<V>* C) {
  const Vec3<V> r0 = cp - g.Pos0();
  static_for<0, 3>([&](auto kc) { C[decltype(kc)::value] = Cross(g.Rot(decltype(kc)::value), r0); });
  if constexpr (K >= 1) C[3] = Cross(g.Rot(2), cp - g.PosA());  // hip axis = torso z
  if constexpr (K >= 2) C[4] = Cross(g.Ank(), cp - g.PosF());
}
// ContactWrench: the cached constants (SetupContactAt) + the Jacobian columns rebuilt from the pose
template <int K, typename T, typename V, typename G, typename Lds>
EPA_HD void LoadContact(const G& g, int w, T radius, Lds&& lds, Contact<V>& c, Vec3<V>* C) {
  const Vec3<V> ctr = g.At(CenterSlot(w));
  const Vec3<V> cp = {ctr.x, ctr.y, V(0.5) * (ctr.z - V(radius))};
  ContactCols<K>(g, cp, C);
  const int base = kSlotCache + 4 * w;
  c.an = lds(base);
  c.ay = lds(base + 1);
  c.ax = lds(base + 2);
  c.D = lds(base + 3);
}
// The same for a sphere class that differs from lane to lane (w: the lane's own class, `has`: the lane has one): the
// class, its link and so the number of Jacobian columns are DATA -- LDS reads at per-lane addresses, the columns of
// the hinges beyond the class's link are exact zeros (the sums they enter are unchanged), a lane without a class
// carries D = 0 (and reads no cache slot: SetupContact wrote only the classes inside the margin).  All six leg
// spheres have one radius (CheckLegSymmetry), the torso sphere its own.
constexpr unsigned long long kCenterSlotTab =
    (unsigned long long)kSlotPosA | ((unsigned long long)kSlotPos0 << 6) | ((unsigned long long)kSlotPosF << 12) |
    ((unsigned long long)kSlotPosA << 18) | ((unsigned long long)kSlotTip << 24) | ((unsigned long long)kSlotPosF << 30) |
    ((unsigned long long)kSlotPos0 << 36);
template <typename T, typename V, typename B, typename U, typename G>
EPA_HD void LoadContactAt(const AntModel<T>& m, const G& g, const U& w, B has, Contact<V>& c, Vec3<V>* C) {
  const U cs = Tab6(kCenterSlotTab, w);
  const Vec3<V> ctr = {GatherSlot(g.lds, cs), GatherSlot(g.lds, UMad(cs, 1u, 1u)), GatherSlot(g.lds, UMad(cs, 1u, 2u))};
  const B torso = UEq(w, 6u);
  const V radius = Sel(torso, V(m.sph_r[0]), V(m.sph_r[1]));
  const Vec3<V> cp = {ctr.x, ctr.y, V(0.5) * (ctr.z - radius)};
  const Vec3<V> r0 = cp - g.Pos0();
  static_for<0, 3>([&](auto kc) { C[decltype(kc)::value] = Cross(g.Rot(decltype(kc)::value), r0); });
  const B k1 = UGe(w, 2u) & !torso, k2 = UGe(w, 4u) & !torso;  // link >= 1 (aux), == 2 (foot)
  const Vec3<V> c3 = Cross(g.Rot(2), cp - g.PosA()), c4 = Cross(g.Ank(), cp - g.PosF());
  C[3] = {Sel(k1, c3.x, V(0)), Sel(k1, c3.y, V(0)), Sel(k1, c3.z, V(0))};
  C[4] = {Sel(k2, c4.x, V(0)), Sel(k2, c4.y, V(0)), Sel(k2, c4.z, V(0))};
  const U base = UMad(w, 4u, (unsigned)kSlotCache);
  c.an = Sel(has, GatherSlot(g.lds, base), V(0));
  c.ay = Sel(has, GatherSlot(g.lds, UMad(base, 1u, 1u)), V(0));
  c.ax = Sel(has, GatherSlot(g.lds, UMad(base, 1u, 2u)), V(0));
  c.D = Sel(has, GatherSlot(g.lds, UMad(base, 1u, 3u)), V(0));
}
// SetupContact for the lane's own class w (see LoadContactAt): once per forward pass
template <typename T, typename V, typename B, typename U, typename G>
EPA_HD void SetupContactAt(const AntModel<T>& m, const G& g, const U& w, B has, const V* v) {
  Contact<V> unused;
  Vec3<V> C[5];
  LoadContactAt(m, g, w, has, unused, C);
  const U cs = Tab6(kCenterSlotTab, w);
  const B torso = UEq(w, 6u);
  const V dist = GatherSlot(g.lds, UMad(cs, 1u, 2u)) - Sel(torso, V(m.sph_r[0]), V(m.sph_r[1]));
  Vec3<V> vel = {v[0], v[1], v[2]};
  static_for<0, 5>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    vel = vel + C[k] * v[3 + k];
  });
  // body_invweight0 of the class's geom: torso sphere, stub, leg capsule, ankle capsule
  const V invw = Sel(torso, V(m.geom_body_invw[0]),
                     Sel(UGe(w, 4u), V(m.geom_body_invw[3]), Sel(UGe(w, 2u), V(m.geom_body_invw[2]), V(m.geom_body_invw[1]))));
  const V rr = dist - V(m.margin);
  const V imp = ImpedanceV(m.imp_d0, m.imp_dmax, m.imp_width, rr);
  const V num = (V(1) - imp) * (invw * V(T(1) + m.mu * m.mu));  // R = max(mjMINVAL, num / imp)
  const V invR = Sel(num < V(1e-15) * imp, V(1e15), imp / num);
  const U base = UMad(w, 4u, (unsigned)kSlotCache);
  ScatterSlot(g.lds, base, -V(m.con_B) * vel.z - V(m.con_K) * imp * rr, has);               // an
  ScatterSlot(g.lds, UMad(base, 1u, 1u), V(m.con_B * m.mu) * vel.y, has);                    // ay
  ScatterSlot(g.lds, UMad(base, 1u, 2u), V(m.con_B * m.mu) * vel.x, has);                    // ax
  ScatterSlot(g.lds, UMad(base, 1u, 3u), invR * V(T(1) / (T(2) * m.mu * m.mu)), has);        // D_py = 1 / (2 mu^2 R)
}
template <int K, typename V>
EPA_HD Vec3<V> JacMul(const Vec3<V>* C, const V* a) {  // J a
  Vec3<V> r = {a[0], a[1], a[2]};
  static_for<0, 3 + K>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    r = r + C[k] * a[3 + k];
  });
  return r;
}
// the four pyramidal rows in terms of (jx, jy, jz) = J a (mj_ant.hip.h, ContactJar)
template <typename T, typename V>
EPA_HD void ContactJar(const AntModel<T>& m, Vec3<V> ja, const Contact<V>& c, V* jar) {
  jar[0] = ja.z + V(m.mu) * ja.y - (c.an - c.ay);
  jar[1] = ja.z - V(m.mu) * ja.y - (c.an + c.ay);
  jar[2] = ja.z - V(m.mu) * ja.x - (c.an + c.ax);
  jar[3] = ja.z + V(m.mu) * ja.x - (c.an - c.ax);
}

// wave-uniform dispatch of a sphere class to its link; radius / body_invweight0 are
// scalars (torso sphere, stub, leg capsule, ankle capsule are the same on every leg)
template <typename T, typename F>
EPA_HD void DispatchSphere(const AntModel<T>& m, int w, F&& f) {
  if (w >= 4 && w < 6) {
    f(IC<2>{}, m.sph_r[5], m.geom_body_invw[3]);
  } else if (w >= 2 && w < 4) {
    f(IC<1>{}, m.sph_r[3], m.geom_body_invw[2]);
  } else {  // fixed to the torso: the stub capsule's ends, or the torso sphere
    const T radius = w == 6 ? m.sph_r[0] : m.sph_r[1];
    const T invw = w == 6 ? m.geom_body_invw[0] : m.geom_body_invw[1];
    f(IC<0>{}, radius, invw);
  }
}

#if defined(__clang__)
#define EPA_ANT4_NO_UNROLL _Pragma("clang loop unroll(disable)")
#else
#define EPA_ANT4_NO_UNROLL
#endif

// gradient / Hessian contributions of the lane's rows at acceleration a:
// g[0..5] and H[0..20] receive the lane's PARTIAL torso sums, the rest is local.
template <typename T, typename V, typename B, typename U, typename G>
EPA_HD void RowsPass(const AntModel<T>& m, const Leg<V, B>& lg, const G& g, const U& vis,
                     const Rows<V>& r, const V* v, const V* a, V* grad, V* H, U* mask) {
  U m0 = MaskFill(*mask, 0u);
  static_for<0, 2>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const V jar = r.sgn[j] * a[6 + j] - r.aref[j];
    const B on = (r.sgn[j] != V(0)) & (jar < V(0));
    const V w = Sel(on, r.D[j], V(0));
    grad[6 + j] += r.sgn[j] * w * jar;
    H[Tri(6 + j, 6 + j)] += w;
    MaskSet(m0, on, j);
  });
  // The lane's OWN touching classes, one per trip of the loop (round 6; rounds 2-5 looped over the wave-uniform UNION
  // of the touching classes with a link-specialised visit: 3.0 visits per pass in the benchmark's steady state where
  // the busiest lane has 1.6 classes -- profiles/r6c_ant_stage_timers.txt).
  U rem = vis;
  EPA_ANT4_NO_UNROLL
  while (AnyWave(AnySlot(rem))) {
    const B has = AnySlot(rem);
    const U w = PopSlot(rem);
    EPA_LDS_FENCE();
    Contact<V> c;
    Vec3<V> C[5];
    LoadContactAt(m, g, w, has, c, C);
    V jar[4], wt[4];
    ContactJar(m, JacMul<2>(C, a), c, jar);
    const U bit0 = UMad(w, 4u, 2u);
    static_for<0, 4>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      const B on = (c.D > V(0)) & (jar[k] < V(0));
      wt[k] = Sel(on, c.D, V(0));
      MaskSetAt(m0, on, UMad(bit0, 1u, (unsigned)k));
    });
    const V wsum = wt[0] + wt[1] + wt[2] + wt[3];
    // unconditional (lanes without an active row add zeros): a branch around the update makes the 8 gradient and 36
    // Hessian accumulators values with two reaching definitions, copied at the head of the class loop and back at its
    // end -- 92 v_mov_b64 of the loop's 434 instructions (mj_planar_lg.hip.h, RowsPass: the same finding)
    {
      const V mu = V(m.mu);
      const V gz = wt[0] * jar[0] + wt[1] * jar[1] + wt[2] * jar[2] + wt[3] * jar[3];
      const V gy = mu * (wt[0] * jar[0] - wt[1] * jar[1]);
      const V gx = mu * (wt[3] * jar[3] - wt[2] * jar[2]);
      const V hzz = wsum;
      const V hyy = V(m.mu * m.mu) * (wt[0] + wt[1]), hxx = V(m.mu * m.mu) * (wt[2] + wt[3]);
      const V hzy = mu * (wt[0] - wt[1]), hzx = mu * (wt[3] - wt[2]);
      // J = [I3 | C]: J^T g and J^T Hc J with Hc = [hxx 0 hzx; 0 hyy hzy; hzx hzy hzz]
      grad[0] += gx;
      grad[1] += gy;
      grad[2] += gz;
      H[Tri(0, 0)] += hxx;
      H[Tri(1, 1)] += hyy;
      H[Tri(2, 2)] += hzz;
      H[Tri(0, 2)] += hzx;
      H[Tri(1, 2)] += hzy;
      Vec3<V> Uc[5];
      static_for<0, 5>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const Vec3<V> ci = C[i];
        grad[3 + i] += ci.x * gx + ci.y * gy + ci.z * gz;
        Uc[i] = {hxx * ci.x + hzx * ci.z, hyy * ci.y + hzy * ci.z,
                 hzx * ci.x + hzy * ci.y + hzz * ci.z};
        H[Tri(0, 3 + i)] += Uc[i].x;
        H[Tri(1, 3 + i)] += Uc[i].y;
        H[Tri(2, 3 + i)] += Uc[i].z;
        static_for<0, i + 1>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          H[Tri(3 + j, 3 + i)] += Dot(C[j], Uc[i]);
        });
      });
    }
  }
  *mask = m0;
}

// lane-partial first / second derivative of the constraint cost along s at step alpha; kMask: also
// the lane's active-row mask AT a + alpha s (the bits of RowsPass)
template <bool kMask, typename T, typename V, typename B, typename U, typename G>
EPA_HD void LineEval(const AntModel<T>& m, const Leg<V, B>& lg, const G& g, const U& vis,
                     const Rows<V>& r, const V* v, const V* a, const V* s, V alpha, V* d1, V* d2,
                     U* mask) {
  U m1 = MaskFill(*mask, 0u);
  static_for<0, 2>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const V jar = r.sgn[j] * a[6 + j] - r.aref[j];
    const V jv = r.sgn[j] * s[6 + j];
    const V x = jar + alpha * jv;
    const B on = (r.sgn[j] != V(0)) & (x < V(0));
    const V w = Sel(on, r.D[j], V(0));
    *d1 += w * x * jv;
    *d2 += w * jv * jv;
    if constexpr (kMask) MaskSet(m1, on, j);
  });
  U rem = vis;  // the lane's own touching classes, as in RowsPass
  EPA_ANT4_NO_UNROLL
  while (AnyWave(AnySlot(rem))) {
    const B has = AnySlot(rem);
    const U w = PopSlot(rem);
    EPA_LDS_FENCE();
    Contact<V> c;
    Vec3<V> C[5];
    LoadContactAt(m, g, w, has, c, C);
    V jar[4];
    ContactJar(m, JacMul<2>(C, a), c, jar);
    const Vec3<V> js = JacMul<2>(C, s);
    const V mu = V(m.mu);
    const V jv[4] = {js.z + mu * js.y, js.z - mu * js.y, js.z - mu * js.x, js.z + mu * js.x};
    const U bit0 = UMad(w, 4u, 2u);
    static_for<0, 4>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      const V x = jar[k] + alpha * jv[k];
      const V wt = Sel(x < V(0), c.D, V(0));  // D == 0 on lanes without contact
      *d1 += wt * x * jv[k];
      *d2 += wt * jv[k] * jv[k];
      if constexpr (kMask) MaskSetAt(m1, (c.D > V(0)) & (x < V(0)), UMad(bit0, 1u, (unsigned)k));
    });
  }
  if constexpr (kMask) *mask = m1;
}

// y = M x for the arrow-structured M held as the lane-local packed 8x8 in LDS
// (torso block replicated): y[0..5] complete (one quad reduction), y[6..7] local
template <typename V, typename Lds>
EPA_HD void MulM(Lds&& lds, const V* x, V* y) {
  V part[6];
  static_for<0, 6>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    part[i] = lds(MSlot(Tri(i, 6))) * x[6] + lds(MSlot(Tri(i, 7))) * x[7];
  });
  static_for<0, 6>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    V s = Sum4(part[i]);
    static_for<0, 6>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      s += lds(MSlot(i <= j ? Tri(i, j) : Tri(j, i))) * x[j];
    });
    y[i] = s;
  });
  static_for<6, 8>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    V s = V(0);
    static_for<0, 8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      s += lds(MSlot(i <= c ? Tri(i, c) : Tri(c, i))) * x[i];
    });
    y[c] = s;
  });
}

constexpr int kLsExactAfter = 8;  // see Solve

// mj_fwdConstraint: exact Newton on the primal objective (mj_ant.hip.h, AntSolve), with
// the leg blocks eliminated inside each lane.
template <typename U, typename T, typename V, typename B, typename Lds>
EPA_HD void Solve(const AntModel<T>& m, const Leg<V, B>& lg, Lds&& lds, const U& own,
                  const Rows<V>& r, const V* v, const V* qfrc, const SolverCfg<T>& cfg, V* qacc,
                  V* n_env, int* n_wave) {
  const Geo<V, typename std::remove_reference<Lds>::type> g{lds};
  V fs = V(0);
  static_for<0, 6>([&](auto ic) { fs = Max(fs, Abs(qfrc[decltype(ic)::value])); });
  fs = Max(fs, Max4(Max(Abs(qfrc[6]), Abs(qfrc[7]))));
  const V gstop = V(cfg.gtol) * (V(1) + fs);
  const V gfloor = V(sizeof(T) == 4 ? T(1e-4) : T(1e-9)) * (V(1) + fs);
  const V gstop2 = gstop * gstop, gfloor2 = gfloor * gfloor;
  V prev_gn2 = V(-1);
  U pm = MaskFill(U(), ~0u);
  V res[kL];  // M qacc - qfrc_smooth, kept current incrementally
  EPA_LDS_FENCE();
  MulM(lds, qacc, res);
  static_for<0, kL>([&](auto ic) { res[decltype(ic)::value] -= qfrc[decltype(ic)::value]; });
  B full_step = V(0) > V(0);
  B live = !full_step;  // this env is still iterating (finished ones keep a frozen qacc)
  for (int it = 0; it < cfg.max_iter; ++it) {
    V H[kLTri], s[kL];
    EPA_LDS_FENCE();
    static_for<0, kTTri>([&](auto ic) { H[decltype(ic)::value] = V(0); });  // torso: partial
    static_for<kTTri, kLTri>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      H[i] = lds(MSlot(i));
    });
    static_for<0, 6>([&](auto ic) { s[decltype(ic)::value] = V(0); });
    s[6] = res[6];
    s[7] = res[7];
    U m0 = MaskFill(U(), 0u);
    EPA_ANT_TICK(4);
    // a lane visits its own touching classes while its env is still iterating (a finished env's qacc is frozen)
    const U vis = SlotsWhere(own, live);
    RowsPass(m, lg, g, vis, r, v, qacc, s, H, &m0);
    // full gradient: torso = smooth part + sum of the lanes' contact parts
    static_for<0, 6>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      s[i] = res[i] + Sum4(s[i]);
    });
    V gn2 = Sum4(s[6] * s[6] + s[7] * s[7]);
    static_for<0, 6>([&](auto ic) { gn2 += s[decltype(ic)::value] * s[decltype(ic)::value]; });
    const B same = All4(MaskSame(m0, pm));
    const B stop = (gn2 <= gstop2) | (full_step & same) |
                   ((prev_gn2 >= V(0)) & (gn2 <= gfloor2) & (gn2 >= V(0.0625) * prev_gn2));
    live = live & !stop;
    EPA_ANT_TICK(3);
    if (!AnyWave(live)) break;
    EPA_ANT_COUNT(0);
    *n_env += Sel(live, V(1), V(0));  // Newton iterations of this env / executed by the wave
    *n_wave += 1;
    prev_gn2 = gn2;
    pm = m0;
    // ---- H s = -grad.  U U^T from the last dof up: ankle, hip (local), then torso
    const V inv7 = Rsq(H[Tri(7, 7)]);
    static_for<0, 7>([&](auto ic) { H[Tri(decltype(ic)::value, 7)] *= inv7; });
    const V inv6 = Rsq(H[Tri(6, 6)] - H[Tri(6, 7)] * H[Tri(6, 7)]);
    static_for<0, 6>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      H[Tri(i, 6)] = (H[Tri(i, 6)] - H[Tri(i, 7)] * H[Tri(6, 7)]) * inv6;
    });
    // first substitution (rows from the last up) for the leg; b = -grad
    const V y7 = -s[7] * inv7;
    const V y6 = (-s[6] - H[Tri(6, 7)] * y7) * inv6;
    V tt[kTTri], ct[6];
    static_for<0, 6>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      static_for<0, j + 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        tt[Tri(i, j)] = Sum4(H[Tri(i, j)] - H[Tri(i, 6)] * H[Tri(j, 6)] - H[Tri(i, 7)] * H[Tri(j, 7)]) +
                        lds(MSlot(Tri(i, j)));
      });
      ct[j] = -s[j] - Sum4(H[Tri(j, 6)] * y6 + H[Tri(j, 7)] * y7);
    });
    FactorTorso(tt);
    static_for_down<6, 0>([&](auto jc) {  // U y = rhs, rows from the last up
      constexpr int j = decltype(jc)::value;
      V t = ct[j];
      static_for<j + 1, 6>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        t -= tt[Tri(j, k)] * ct[k];
      });
      ct[j] = t * tt[Tri(j, j)];
    });
    static_for<0, 6>([&](auto jc) {  // U^T x = y
      constexpr int j = decltype(jc)::value;
      V t = ct[j];
      static_for<0, j>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        t -= tt[Tri(i, j)] * s[i];
      });
      s[j] = t * tt[Tri(j, j)];
    });
    {
      V t6 = y6, t7 = y7;
      static_for<0, 6>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        t6 -= H[Tri(i, 6)] * s[i];
        t7 -= H[Tri(i, 7)] * s[i];
      });
      s[6] = t6 * inv6;
      s[7] = (t7 - H[Tri(6, 7)] * s[6]) * inv7;
    }
    // ---- line search on the piecewise-quadratic cost (one evaluation at the full step, see below)
    EPA_ANT_TICK(4);
    V Ms[kL];
    EPA_LDS_FENCE();
    MulM(lds, s, Ms);
    V g1 = Sum4(s[6] * res[6] + s[7] * res[7]), g2 = Sum4(s[6] * Ms[6] + s[7] * Ms[7]);
    static_for<0, 6>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      g1 += s[i] * res[i];
      g2 += s[i] * Ms[i];
    });
    V alpha = V(1), lo = V(0), hi = V(-1);
    full_step = V(0) > V(0);
    const V ls_tol = V(sizeof(T) == 4 ? T(1e-4) : T(1e-10)) * Abs(g1);
    B searching = live;
    B exact = V(0) > V(0);
    // ONE evaluation, at the full step: if phi'(1) vanishes the full step is taken (and is the minimiser if the
    // active set is the one H was built with), otherwise one Newton step of the 1-D problem, unverified -- an env
    // takes as many Newton trips as with the exact search of rounds 1-4 and ends on the same minimiser, see
    // mj_planar_lg.hip.h::Solve.  From trip kLsExactAfter on a wave searches exactly again (never seen to happen).
    const int ls_max = it < kLsExactAfter ? 1 : 24;
    for (int ls = 0; ls < ls_max; ++ls) {
      V p1 = V(0), p2 = V(0);
      U m1 = MaskFill(U(), 0u);
      if (ls == 0) {
        LineEval<true>(m, lg, g, vis, r, v, qacc, s, alpha, &p1, &p2, &m1);
      } else {
        LineEval<false>(m, lg, g, vis, r, v, qacc, s, alpha, &p1, &p2, &m1);
      }
      const V d1 = g1 + alpha * g2 + Sum4(p1), d2 = g2 + Sum4(p2);
      const B hit = Abs(d1) <= ls_tol;
      // a full Newton step whose active set at a + s is the one H was built with lands ON the
      // minimiser (finite termination): the env is done without another pass over the rows -- the
      // pass that would only have found `full_step & same` at the top of the next iteration
      if (ls == 0) {
        full_step = searching & hit;
        exact = full_step & All4(MaskSame(m1, m0));
      }
      searching = searching & !hit;
      lo = Sel(searching & (d1 < V(0)), alpha, lo);
      hi = Sel(searching & !(d1 < V(0)), alpha, hi);
      V next = alpha - d1 / d2;
      next = Sel((hi >= V(0)) & ((next <= lo) | (next >= hi)), V(0.5) * (lo + hi), next);
      next = Sel(next <= V(0), V(0.5) * alpha, next);
      searching = searching & (next != alpha);
      alpha = Sel(searching, next, alpha);
      if (!AnyWave(searching)) break;
    }
    const V step = Sel(live, alpha, V(0));
    static_for<0, kL>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      qacc[i] += step * s[i];
      res[i] += step * Ms[i];
    });
    live = live & !exact;
    EPA_ANT_TICK(5);
    if (!AnyWave(live)) break;
  }
  EPA_ANT_TICK(4);
}

// ---- fused front end of one forward pass, one leg per lane ------------------------
// mj_kinematics + mj_comPos + mj_crb + mj_comVel + mj_rne + mj_passive +
// mj_fwdActuation + limit rows (mj_ant.hip.h, AntFrontEnd).  q = torso pose (7,
// replicated) + hip, ankle angle of the lane's leg; v, qfrc likewise (6 + 2).
// Returns the wave-uniform mask of sphere classes inside the contact margin.
template <typename U, typename T, typename V, typename B, typename Lds>
EPA_HD unsigned FrontEnd(const AntModel<T>& m, const Leg<V, B>& lg, V* q, const V* v,
                         const V* ctrl, Lds&& lds, Rows<V>& rows, V* qfrc, U* own) {
  constexpr int A0 = ant::Aux(0), F0 = ant::Foot(0);  // leg 0 is the (+, +) prototype
  auto put = [&](int base, Vec3<V> x) {
    lds(base) = x.x;
    lds(base + 1) = x.y;
    lds(base + 2) = x.z;
  };
  unsigned mask = 0;
  U mine = MaskFill(U(), 0u);  // classes inside the margin on this lane
  auto probe = [&](int w, V z, T radius) {
    const auto in = z - V(radius) < V(m.margin);
    MaskSet(mine, in, w);
    if (AnyWave(in)) mask |= 1u << w;
  };
  NormalizeQuatV(q + 3);  // mj_kinematics
  const Vec3<V> pos0 = {q[0], q[1], q[2]};
  const Mat3<V> R0 = ant::QuatToMat(q[3], q[4], q[5], q[6]);
  // leg frames: aux = torso * Rz(hip), foot = aux * R(ankle axis, ankle)
  const Vec3<V> posA = pos0 + MulXY(R0, lg.sx * V(m.aux_pos[0][0]), lg.sy * V(m.aux_pos[0][1]));
  Mat3<V> RA, RF;
  {
    V s, c;
    SinCos(q[7], &s, &c);
    static_for<0, 3>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      RA.m[3 * r] = R0.m[3 * r] * c + R0.m[3 * r + 1] * s;
      RA.m[3 * r + 1] = R0.m[3 * r + 1] * c - R0.m[3 * r] * s;
      RA.m[3 * r + 2] = R0.m[3 * r + 2];
    });
  }
  const Vec3<V> posF = posA + MulXY(RA, lg.sx * V(m.foot_pos[0][0]), lg.sy * V(m.foot_pos[0][1]));
  const V ka = V(m.ankle_axis[0][1]);  // 1 / sqrt 2; the axis is (axs ka, ka, 0)
  const V axk = lg.axs * ka;
  {
    V s, c;
    SinCos(q[8], &s, &c);
    const V t = V(1) - c;
    // Rodrigues with a = (axk, ka, 0)
    const V r00 = c + axk * axk * t, r11 = c + ka * ka * t, r01 = axk * ka * t;
    const V r02 = ka * s, r12 = -axk * s;
    static_for<0, 3>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const V a = RA.m[3 * r], b = RA.m[3 * r + 1], cc = RA.m[3 * r + 2];
      RF.m[3 * r] = a * r00 + b * r01 - cc * r02;
      RF.m[3 * r + 1] = a * r01 + b * r11 - cc * r12;
      RF.m[3 * r + 2] = a * r02 + b * r12 + cc * c;
    });
  }
  const Vec3<V> tip = posF + MulXY(RF, lg.sx * V(m.sph[5][0]), lg.sy * V(m.sph[5][1]));
  const Vec3<V> hz = ant::Col(R0, 2);  // hip axis: +z of the aux frame = torso z
  const Vec3<V> ha = {RA.m[0] * axk + RA.m[1] * ka, RA.m[3] * axk + RA.m[4] * ka,
                      RA.m[6] * axk + RA.m[7] * ka};
  // mj_comPos: subtree COM of the robot
  const Vec3<V> xi0 = pos0 + ant::Mul(R0, Vec3<V>{V(m.com[0][0]), V(m.com[0][1]), V(m.com[0][2])});
  const Vec3<V> xiA = posA + MulXY(RA, lg.sx * V(m.com[A0][0]), lg.sy * V(m.com[A0][1]));
  const Vec3<V> xiF = posF + MulXY(RF, lg.sx * V(m.com[F0][0]), lg.sy * V(m.com[F0][1]));
  const Vec3<V> com =
      (xi0 * V(m.mass[0]) + Sum4v(xiA * V(m.mass[A0]) + xiF * V(m.mass[F0]))) * V(T(1) / m.total_mass);
  put(kSlotCom, com);
  put(kSlotPos0, pos0);
  put(kSlotPosA, posA);
  put(kSlotPosF, posF);
  put(kSlotTip, tip);
  put(kSlotAnk, ha);
  probe(6, pos0.z, m.sph_r[0]);
  probe(0, posA.z, m.sph_r[1]);
  probe(1, pos0.z, m.sph_r[2]);
  probe(2, posF.z, m.sph_r[3]);
  probe(3, posA.z, m.sph_r[4]);
  probe(4, tip.z, m.sph_r[5]);
  probe(5, posF.z, m.sph_r[6]);
  // torso: cdof of the free joint (translations are (0; e_k)), velocity, acceleration
  Sp6<V> rdof[3];
  static_for<0, 3>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const Vec3<V> ax = ant::Col(R0, k);
    rdof[k] = {ax, Cross(ax, com - pos0)};
    put(kSlotRot + 3 * k, ax);
  });
  Sp6<V> cvel0 = {{V(0), V(0), V(0)}, {v[0], v[1], v[2]}};
  Sp6<V> cacc0 = {{V(0), V(0), V(0)}, {V(0), V(0), V(m.gravity)}};
  {
    const Sp6<V> before = cvel0;  // cdof_dot of the three rotations use the velocity before them
    static_for<0, 3>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      ant::Axpy(cacc0, CrossMotion(before, rdof[k]), v[3 + k]);
      ant::Axpy(cvel0, rdof[k], v[3 + k]);
    });
  }
  // inertias about the robot COM (torso: full tensor, legs: xz = yz = 0)
  In10<V> cinert0;
  {
    const T* I = m.inertia[0];
    // general symmetric tensor for the torso (its products of inertia cancel to rounding)
    Mat3<V> Ib = {{V(I[0]), V(I[3]), V(I[4]), V(I[3]), V(I[1]), V(I[5]), V(I[4]), V(I[5]), V(I[2])}};
    Mat3<V> RI = ant::Mul(R0, Ib);
    const V* r = R0.m;
    const Vec3<V> d = xi0 - com;
    const V mass = V(m.mass[0]), d2 = Dot(d, d);
    V* c = cinert0.v;
    c[0] = RI.m[0] * r[0] + RI.m[1] * r[1] + RI.m[2] * r[2] + mass * (d2 - d.x * d.x);
    c[1] = RI.m[3] * r[3] + RI.m[4] * r[4] + RI.m[5] * r[5] + mass * (d2 - d.y * d.y);
    c[2] = RI.m[6] * r[6] + RI.m[7] * r[7] + RI.m[8] * r[8] + mass * (d2 - d.z * d.z);
    c[3] = RI.m[0] * r[3] + RI.m[1] * r[4] + RI.m[2] * r[5] - mass * d.x * d.y;
    c[4] = RI.m[0] * r[6] + RI.m[1] * r[7] + RI.m[2] * r[8] - mass * d.x * d.z;
    c[5] = RI.m[3] * r[6] + RI.m[4] * r[7] + RI.m[5] * r[8] - mass * d.y * d.z;
    c[6] = mass * d.x;
    c[7] = mass * d.y;
    c[8] = mass * d.z;
    c[9] = mass;
  }
  const In10<V> ciA = CinertOf(V(m.inertia[A0][0]), V(m.inertia[A0][1]), V(m.inertia[A0][2]),
                               lg.sxy * V(m.inertia[A0][3]), V(m.mass[A0]), RA, xiA - com);
  const In10<V> ciF = CinertOf(V(m.inertia[F0][0]), V(m.inertia[F0][1]), V(m.inertia[F0][2]),
                               lg.sxy * V(m.inertia[F0][3]), V(m.mass[F0]), RF, xiF - com);
  const Sp6<V> dh = {hz, Cross(hz, com - posA)};
  const Sp6<V> da = {ha, Cross(ha, com - posF)};
  // dot of root dof k (0..5) with a spatial force
  auto root_dot = [&](auto kc, const Sp6<V>& f) -> V {
    constexpr int k = decltype(kc)::value;
    if constexpr (k == 0) return f.l.x;
    if constexpr (k == 1) return f.l.y;
    if constexpr (k == 2) return f.l.z;
    if constexpr (k >= 3) return Dot(rdof[k - 3], f);
  };
  // mj_crb: composite inertias foot, aux (+ foot), whole robot; the lane's columns of M
  In10<V> crbA = ciA, crb0;
  static_for<0, 10>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    crbA.v[k] += ciF.v[k];
    crb0.v[k] = cinert0.v[k] + Sum4(crbA.v[k]);
  });
  {
    const Sp6<V> buf = MulInert(ciF, da);  // ankle dof
    static_for<0, 6>([&](auto kc) { lds(MSlot(Tri(decltype(kc)::value, 7))) = root_dot(kc, buf); });
    lds(MSlot(Tri(6, 7))) = Dot(dh, buf);
    lds(MSlot(Tri(7, 7))) = Dot(da, buf) + V(m.arm[1]);
  }
  {
    const Sp6<V> buf = MulInert(crbA, dh);  // hip dof
    static_for<0, 6>([&](auto kc) { lds(MSlot(Tri(decltype(kc)::value, 6))) = root_dot(kc, buf); });
    lds(MSlot(Tri(6, 6))) = Dot(dh, buf) + V(m.arm[0]);
  }
  // mj_comVel / mj_rne (flg_acc = 0) down the leg and back
  Sp6<V> cvA = cvel0, caA = cacc0;
  ant::Axpy(caA, CrossMotion(cvel0, dh), v[6]);
  ant::Axpy(cvA, dh, v[6]);
  Sp6<V> cvF = cvA, caF = caA;
  ant::Axpy(caF, CrossMotion(cvA, da), v[7]);
  ant::Axpy(cvF, da, v[7]);
  Sp6<V> frcF, frcA, cfrc0;
  {
    const Sp6<V> x = MulInert(ciF, caF);
    const Sp6<V> gg = CrossForce(cvF, MulInert(ciF, cvF));
    frcF = {x.a + gg.a, x.l + gg.l};
  }
  {
    const Sp6<V> x = MulInert(ciA, caA);
    const Sp6<V> gg = CrossForce(cvA, MulInert(ciA, cvA));
    frcA = {x.a + gg.a + frcF.a, x.l + gg.l + frcF.l};
  }
  {
    const Sp6<V> x = MulInert(cinert0, cacc0);
    const Sp6<V> gg = CrossForce(cvel0, MulInert(cinert0, cvel0));
    cfrc0 = {x.a + gg.a + Sum4v(frcA.a), x.l + gg.l + Sum4v(frcA.l)};
  }
  // hinge damper (stiffness 0) - bias + motor (gear * clamped ctrl)
  qfrc[6] = -V(m.damp[0]) * v[6] - Dot(dh, frcA) + V(m.gear) * ctrl[0];
  qfrc[7] = -V(m.damp[1]) * v[7] - Dot(da, frcF) + V(m.gear) * ctrl[1];
  // root block of M from the composite inertia of the whole robot; root bias
  static_for<0, 6>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    Sp6<V> di;
    if constexpr (i < 3) {
      di = {{V(0), V(0), V(0)}, {V(i == 0), V(i == 1), V(i == 2)}};
    } else {
      di = rdof[i - 3];
    }
    const Sp6<V> buf = MulInert(crb0, di);
    static_for<0, i + 1>([&](auto jc) { lds(MSlot(Tri(decltype(jc)::value, i))) = root_dot(jc, buf); });
    qfrc[i] = -root_dot(ic, cfrc0);
  });
  // mj_instantiateLimit + mj_makeImpedance for the lane's two limited hinges
  const V lo[2] = {V(m.lo[0]), lg.alo}, hi[2] = {V(m.hi[0]), lg.ahi};
  static_for<0, 2>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const V qq = q[7 + j];
    const V dlo = qq - lo[j], dhi = hi[j] - qq;
    const V sgn = Sel(dlo < V(0), V(1), Sel(dhi < V(0), V(-1), V(0)));
    const V dist = Sel(dlo < V(0), dlo, Sel(dhi < V(0), dhi, V(0)));
    const V imp = ImpedanceV(m.imp_d0, m.imp_dmax, m.imp_width, dist);
    const V num = (V(1) - imp) * V(m.dof_invw[j]);  // R = max(mjMINVAL, num / imp)
    const V Dj = Sel(num < V(1e-15) * imp, V(1e15), imp / num);
    rows.sgn[j] = sgn;
    rows.D[j] = Sel(sgn != V(0), Dj, V(0));
    rows.aref[j] = -V(m.con_B) * (sgn * v[6 + j]) - V(m.con_K) * imp * dist;
  });
  *own = mine;
  return WaveUniform(mask);
}

// mj_rnePostConstraint, cfrc_ext part, of the forward pass that just finished
// (mj_ant.hip.h, AntContactWrench): cf[3][6] = [torque about the robot COM; force] on the
// lane's stub / leg / ankle MuJoCo bodies, cf0[6] on the torso body (first lane).
template <typename T, typename V, typename B, typename Lds>
EPA_HD void ContactWrench(const AntModel<T>& m, const Leg<V, B>& lg, Lds&& lds, unsigned sph,
                          const V* v, const V* qacc, V (*cf)[6], V* cf0) {
  const Geo<V, typename std::remove_reference<Lds>::type> g{lds};
  EPA_ANT4_NO_UNROLL
  for (unsigned rem = sph; rem != 0; rem &= rem - 1) {
    const int w = __builtin_ctz(rem);
    EPA_LDS_FENCE();
    DispatchSphere(m, w, [&](auto kc, T radius, T invw) {
      constexpr int K = decltype(kc)::value;
      Contact<V> c;
      Vec3<V> C[5];
      LoadContact<K>(g, w, radius, g.lds, c, C);
      V jar[4], f[4];
      ContactJar(m, JacMul<K>(C, qacc), c, jar);
      static_for<0, 4>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        f[k] = Sel(jar[k] < V(0), -c.D * jar[k], V(0));  // D == 0 on lanes without contact
      });
      const V mu = V(m.mu);
      const Vec3<V> F = {mu * (f[3] - f[2]), mu * (f[0] - f[1]), f[0] + f[1] + f[2] + f[3]};
      const Vec3<V> ctr = g.At(CenterSlot(w));
      const Vec3<V> cp = {ctr.x, ctr.y, V(0.5) * (ctr.z - V(radius))};
      const Vec3<V> tq = Cross(cp - g.At(kSlotCom), F);
      V* dst = w == 6 ? cf0 : cf[K];  // w is wave uniform; K == link of classes 0..5
      dst[0] += tq.x;
      dst[1] += tq.y;
      dst[2] += tq.z;
      dst[3] += F.x;
      dst[4] += F.y;
      dst[5] += F.z;
    });
  }
}

// mj_forward: qacc (in: warm start, out: solution) for state (q, v) under ctrl
template <typename U, bool kWrench, typename T, typename V, typename B, typename Lds>
EPA_HD void Forward(const AntModel<T>& m, const Leg<V, B>& lg, const SolverCfg<T>& cfg, V* q,
                    const V* v, const V* ctrl, V* qacc, Lds&& lds, bool wrench, V (*cf)[6],
                    V* cf0, V* n_env, int* n_wave) {
  V qfrc[kL];
  Rows<V> rows;
  EPA_LDS_FENCE();
  U own;
  EPA_ANT_TICK(6);
  EPA_ANT_COUNT(1);
  const unsigned sph = FrontEnd(m, lg, q, v, ctrl, lds, rows, qfrc, &own);
  EPA_ANT_TICK(1);
  EPA_ANT_CLASSES(sph, own);
  EPA_LDS_FENCE();
  // (the torso sphere is probed on every lane of the quad and owned by the first one)
  const U vis = MaskClear(own, !lg.first, 1u << 6);
  {
    const Geo<V, typename std::remove_reference<Lds>::type> g{lds};
    if constexpr (kWrench) {
      // ContactWrench walks the wave's UNION of classes and reads every lane's cache of them: D = 0 where a
      // lane does not own the class
      if (wrench) {
        EPA_ANT4_NO_UNROLL
        for (unsigned rem = sph; rem != 0; rem &= rem - 1) {
          const int base = kSlotCache + 4 * __builtin_ctz(rem);
          static_for<0, 4>([&](auto kc) { lds(base + decltype(kc)::value) = V(0); });
        }
        EPA_LDS_FENCE();
      }
    }
    U rem = vis;  // the lane's own touching classes, one per trip (see RowsPass)
    EPA_ANT4_NO_UNROLL
    while (AnyWave(AnySlot(rem))) {
      const B has = AnySlot(rem);
      const U w = PopSlot(rem);
      SetupContactAt(m, g, w, has, v);
    }
  }
  EPA_LDS_FENCE();
  EPA_ANT_TICK(2);
  Solve<U>(m, lg, lds, vis, rows, v, qfrc, cfg, qacc, n_env, n_wave);
  // profiling: + 1e3 x sphere classes the wave visits + 1e6 x those of this env
  *n_wave += 1000 * __builtin_popcount(sph) + 1000000 * MaskCount4(own);
  if constexpr (kWrench) {
    if (wrench) ContactWrench(m, lg, lds, sph, v, qacc, cf, cf0);  // wave-uniform flag
  }
}

// mj_integratePos: q <- q (+) h * dq (dq in velocity coordinates), lane layout
template <typename V>
EPA_HD void IntegratePos(V* q, const V* dq, V h) {
  q[0] += h * dq[0];
  q[1] += h * dq[1];
  q[2] += h * dq[2];
  const V wx = dq[3], wy = dq[4], wz = dq[5];
  const V nrm = Sqrt(wx * wx + wy * wy + wz * wz);
  const V ang = nrm * h;
  {
    const auto turn = ang > V(0);  // zero angular velocity leaves the quaternion as is
    V s, c;
    SinCos(V(0.5) * ang, &s, &c);
    const V k = s / Sel(turn, nrm, V(1));
    const V bw = c, bx = wx * k, by = wy * k, bz = wz * k;
    const V aw = q[3], ax = q[4], ay = q[5], az = q[6];
    V nq[4] = {aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
               aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw};
    NormalizeQuatV(nq);
    q[3] = Sel(turn, nq[0], aw);
    q[4] = Sel(turn, nq[1], ax);
    q[5] = Sel(turn, nq[2], ay);
    q[6] = Sel(turn, nq[3], az);
  }
  q[7] += h * dq[6];
  q[8] += h * dq[7];
}

// One mj_step with integrator RK4 (mj_ant.hip.h, AntStep).  q[9], v[8], warm[8] in the
// lane layout (torso replicated); (lagx, lagy) = torso xpos of the LAST forward
// evaluation, which is what data_->xpos holds afterwards (ant.h:169-173).
// The four stages run through ONE instance of the forward pass (code size: these
// kernels sit at the instruction-cache limit): stage i evaluates
//   X_i = X_0 + h a_i (Xv_{i-1}, F_{i-1}),  a = 0, 1/2, 1/2, 1
// where a_0 = 0 makes stage 0 the plain state (mj_integratePos by a zero velocity is
// the identity), and accumulates the B = (1/6, 1/3, 1/3, 1/6) weighted sums.
template <typename U, bool kWrench, typename T, typename V, typename B, typename Lds>
EPA_HD void Step(const AntModel<T>& m, const Leg<V, B>& lg, const SolverCfg<T>& cfg, V* q, V* v,
                 V* warm, const V* ctrl, V* lagx, V* lagy, Lds&& lds, bool wrench, V (*cf)[6],
                 V* cf0, V* n_env, int* n_wave) {
  const V h = V(m.timestep);
  V q0[9], v0[kL], qs[9], vs[kL], dq[kL], dv[kL];
  static_for<0, 9>([&](auto ic) { q0[decltype(ic)::value] = q[decltype(ic)::value]; });
  static_for<0, kL>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v0[i] = v[i];
    vs[i] = v[i];
    dq[i] = V(0);
    dv[i] = V(0);
  });
  EPA_ANT4_NO_UNROLL
  for (int stage = 0; stage < 4; ++stage) {
    const V a = stage == 0 ? V(0) : (stage == 3 ? V(1) : V(0.5));
    const V bw = (stage == 0 || stage == 3) ? V(1.0 / 6.0) : V(1.0 / 3.0);
    V step_dq[kL];
    static_for<0, kL>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      step_dq[i] = a * vs[i];  // vs, warm: previous stage
      vs[i] = v0[i] + h * a * warm[i];
    });
    static_for<0, 9>([&](auto ic) { qs[decltype(ic)::value] = q0[decltype(ic)::value]; });
    IntegratePos(qs, step_dq, h);
    // `warm` is the running qacc: warm start in, solution (F_i) out
    Forward<U, kWrench>(m, lg, cfg, qs, vs, ctrl, warm, lds, wrench && stage == 3, cf, cf0,
                        n_env, n_wave);
    if (stage == 0) {  // mj_kinematics normalised the quaternion of the start state
      static_for<0, 9>([&](auto ic) { q0[decltype(ic)::value] = qs[decltype(ic)::value]; });
    }
    static_for<0, kL>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dq[i] += bw * vs[i];
      dv[i] += bw * warm[i];
    });
  }
  *lagx = qs[0];
  *lagy = qs[1];
  static_for<0, 9>([&](auto ic) { q[decltype(ic)::value] = q0[decltype(ic)::value]; });
  static_for<0, kL>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v[i] = v0[i] + h * dv[i];
  });
  IntegratePos(q, dq, h);
}

// The structural facts of ant_envpool.xml this layout relies on, checked at compile time
// against the generated model constants (static_assert in mujoco_ant.hip) and on the
// host (tests/cpu_harness): legs are mirror images of leg 0 = (+, +) under (sx, sy),
// every leg body has xz = yz = 0 products of inertia and z = 0 offsets, the scalar
// constants are equal on all legs, the ankle axis / range follow LegSx .. LegAnkleRef.
EPA_HD constexpr double LegSx(int l) { return (l == 0 || l == 3) ? 1.0 : -1.0; }
EPA_HD constexpr double LegSy(int l) { return l < 2 ? 1.0 : -1.0; }
EPA_HD constexpr double LegAxs(int l) { return (l & 1) ? 1.0 : -1.0; }
EPA_HD constexpr int LegAnkleRef(int l) { return (l == 0 || l == 3) ? 0 : 1; }  // leg with the same ankle range
constexpr bool CheckLegSymmetry(const AntModel<double>& m) {
  bool ok = true;
  for (int l = 0; l < 4; ++l) {
    const double sx = LegSx(l), sy = LegSy(l);
    const int A = ant::Aux(l), F = ant::Foot(l), A0 = ant::Aux(0), F0 = ant::Foot(0);
    ok = ok && m.aux_pos[l][0] == sx * m.aux_pos[0][0] && m.aux_pos[l][1] == sy * m.aux_pos[0][1] && m.aux_pos[l][2] == 0;
    ok = ok && m.foot_pos[l][0] == sx * m.foot_pos[0][0] && m.foot_pos[l][1] == sy * m.foot_pos[0][1] && m.foot_pos[l][2] == 0;
    ok = ok && m.com[A][0] == sx * m.com[A0][0] && m.com[A][1] == sy * m.com[A0][1] && m.com[A][2] == 0;
    ok = ok && m.com[F][0] == sx * m.com[F0][0] && m.com[F][1] == sy * m.com[F0][1] && m.com[F][2] == 0;
    ok = ok && m.mass[A] == m.mass[A0] && m.mass[F] == m.mass[F0];
    for (int k = 0; k < 3; ++k) ok = ok && m.inertia[A][k] == m.inertia[A0][k] && m.inertia[F][k] == m.inertia[F0][k];
    ok = ok && m.inertia[A][3] == sx * sy * m.inertia[A0][3] && m.inertia[F][3] == sx * sy * m.inertia[F0][3];
    ok = ok && m.inertia[A][4] == 0 && m.inertia[A][5] == 0 && m.inertia[F][4] == 0 && m.inertia[F][5] == 0;
    ok = ok && m.ankle_axis[l][0] == LegAxs(l) * m.ankle_axis[0][1] && m.ankle_axis[l][1] == m.ankle_axis[0][1] && m.ankle_axis[l][2] == 0;
    ok = ok && m.lo[2 * l] == m.lo[0] && m.hi[2 * l] == m.hi[0];
    ok = ok && m.lo[2 * l + 1] == m.lo[2 * LegAnkleRef(l) + 1] && m.hi[2 * l + 1] == m.hi[2 * LegAnkleRef(l) + 1];
    ok = ok && m.dof_invw[2 * l] == m.dof_invw[0] && m.dof_invw[2 * l + 1] == m.dof_invw[1];
    ok = ok && m.damp[2 * l] == m.damp[0] && m.damp[2 * l + 1] == m.damp[1];
    ok = ok && m.arm[2 * l] == m.arm[0] && m.arm[2 * l + 1] == m.arm[1];
    for (int w = 0; w < 3; ++w) ok = ok && m.geom_body_invw[1 + 3 * l + w] == m.geom_body_invw[1 + w];
    const int s0 = 1 + 6 * l;
    const double ends[6] = {m.aux_pos[0][0], 0, m.foot_pos[0][0], 0, m.sph[5][0], 0};
    for (int w = 0; w < 6; ++w) {
      ok = ok && m.sph[s0 + w][0] == sx * ends[w] && m.sph[s0 + w][1] == sy * ends[w] && m.sph[s0 + w][2] == 0;
      ok = ok && m.sph_r[s0 + w] == m.sph_r[1];
    }
  }
  return ok;
}

}  // namespace ant4
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_ANT4_HIP_H_
