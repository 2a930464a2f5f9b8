// K6 — gym Pusher `mj_step`: a 7-hinge serial arm in 3-D pushing a sliding cylinder.
//
// Replaces the arithmetic MuJoCo 3.6.0's mj_step performs for
// third_party/mujoco_gym_xml_patches/pusher_envpool.xml / pusher_v5_envpool.xml each time the
// reference calls it (envpool/mujoco/gym/mujoco_env.h:137-148, task body pusher.h:115-224):
// SURVEY.md §8a M1-M9 with gravity 0, <option iterations="20" integrator="Euler">, condim 1
// (frictionless) contacts and margin 0.002.
//
// Model facts this file relies on (mj_pusher_model.h transcribes them, cited by XML line):
//  * bodies without a joint (r_upper_arm_link, r_forearm_link, tips_arm) sit at offset 0 in
//    their parent, so the arm is SEVEN links with hinge axes z y x y x y x through the link
//    origins; link offsets only along the chain; every body quaternion is the identity;
//  * the object has two slides (y first, then x: obj_slidey / obj_slidex) and no z dof; the
//    goal's two slides carry no force at all (no gravity, no contact, zero velocity), so the
//    goal is a per-episode constant;
//  * colliding geoms (contype / conaffinity != 0): the table plane, the three capsules of the
//    wrist, the object's cylinder.  Plane - wrist capsule = two plane - sphere tests per
//    capsule; wrist capsule - cylinder goes through MuJoCo's convex collider, restated as the
//    closest points between the capsule's axis segment and the solid cylinder (see
//    CapsuleCylinder); plane - cylinder contacts have an identically zero Jacobian (normal z,
//    object slides in x / y) and are not generated.
// One env per thread, fp64, spatial algebra about the world origin in world axes (the choice
// of reference point changes rounding only).  Same primal Newton with exact line search and
// finite termination as the other MuJoCo kernels; Euler with implicit joint damping.
#ifndef ENVPOOL_AMD_CSRC_MJ_PUSHER_HIP_H_
#define ENVPOOL_AMD_CSRC_MJ_PUSHER_HIP_H_

#include "mj_ant.hip.h"  // Vec3, Mat3, Sp6, In10, MulInert, Cross*, static_for, Impedance, WaveAny

namespace epa {
namespace mj {
namespace pusher {

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

constexpr int kNL = 7;         // arm links / hinges
constexpr int kNV = 9;         // arm 7 + object slides (y, x)
constexpr int kNQ = 11;        // + the goal's two slides (constants of an episode)
constexpr int kNSph = 6;       // wrist capsule end spheres vs the table
constexpr int kNCap = 3;       // wrist capsules vs the object's cylinder
constexpr int kNCon = kNSph + kNCap;

// hinge axis of link l (0 x, 1 y, 2 z, through the link origin): z y x y x y x
// (pusher_envpool.xml:37,41,45,52,56,63,66); mj_pusher_model.h checks its table against this
EPA_HD constexpr int LinkAxis(int l) { return l == 0 ? 2 : ((l & 1) ? 1 : 0); }

template <typename T>
struct PusherModel {
  T off[kNL][3];               // link origin in the parent link frame (link 0: in the world)
  T mass[kNL], com[kNL][3], inertia[kNL][6];  // composite of the welded bodies: xx yy zz xy xz yz about com
  T lo[kNL], hi[kNL], damp[kNV], arm[kNV], dof_invw[kNL];
  T cap_p0[kNCap][3], cap_p1[kNCap][3], cap_r;  // wrist capsules in the wrist frame ("from", "to")
  T wrist_invw, obj_invw;      // body_invweight0 (translational) of r_wrist_roll_link / object
  T table_z;                   // the plane z = table_z, normal +z
  T obj_pos[3], obj_mass, cyl_r, cyl_h;         // object body position at qpos = 0; its cylinder
  T goal_pos[3];
  T margin;
  T sol_K, sol_B, imp_d0, imp_dmax, imp_width;  // default solref / solimp everywhere
  T ctrl_lo, ctrl_hi, timestep;
};

// what the task reads after a step: xpos of the LAST forward evaluation (pusher.h:190-224)
template <typename T>
struct PusherLag {
  T tips[3];   // xpos[tips_arm] = origin of the wrist frame
  T obj[2];    // xpos[object].xy
};

template <int axis, typename T>
EPA_HD Mat3<T> RotAxis(const Mat3<T>& R, T ang) {  // R * Rot(axis, ang), axis 0 / 1 / 2
  T s, c;
  SinCos(ang, &s, &c);
  Mat3<T> out;
  constexpr int a = (axis + 1) % 3, b = (axis + 2) % 3;  // columns a, b rotate in their plane
  static_for<0, 3>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    out.m[3 * r + axis] = R.m[3 * r + axis];
    out.m[3 * r + a] = R.m[3 * r + a] * c + R.m[3 * r + b] * s;
    out.m[3 * r + b] = R.m[3 * r + b] * c - R.m[3 * r + a] * s;
  });
  return out;
}

// ---- dense Cholesky on a full N x N (row major), in place; x <- A^-1 x -------------------
template <typename T, int N>
EPA_HD void CholSolveN(T* A, T* x) {
  static_for<0, N>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T d = A[j * N + j];
    static_for<0, j>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      d -= A[j * N + k] * A[j * N + k];
    });
    const T inv = Rsqrt(d);
    A[j * N + j] = inv;  // the diagonal holds 1 / L_jj
    static_for<j + 1, N>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      T t = A[i * N + j];
      static_for<0, j>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        t -= A[i * N + k] * A[j * N + k];
      });
      A[i * N + j] = t * inv;
    });
  });
  static_for<0, N>([&](auto ic) {  // L y = x
    constexpr int i = decltype(ic)::value;
    T t = x[i];
    static_for<0, i>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      t -= A[i * N + k] * x[k];
    });
    x[i] = t * A[i * N + i];
  });
  static_for_down<N, 0>([&](auto ic) {  // L^T x = y
    constexpr int i = decltype(ic)::value;
    T t = x[i];
    static_for<i + 1, N>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      t -= A[k * N + i] * x[k];
    });
    x[i] = t * A[i * N + i];
  });
}

// Closest points between the segment p0 + t dp, t in [0, 1], and the solid cylinder (centre
// c, unit axis +z of the world -- the object never rotates --, radius R, half height H).
// F(t) = dist^2 is convex, g = F' / 2 is monotone: two bisections bracket the minimiser set
// [last t with g < -eps, first t with g > +eps] and the contact takes its midpoint (a unique
// minimum, or the middle of the closest stretch when the segment runs parallel to a face).
// The thresholds make the result insensitive to last-bit differences between builds.
// Stands in for MuJoCo's mjc_Convex on this pair (GJK / EPA, tolerance 1e-6 m): the same
// geometric quantity, defined deterministically.  Returns g; P / Q: the two closest points.
template <typename T>
EPA_HD T CapCylG(Vec3<T> p0, Vec3<T> dp, T t, Vec3<T> c, T R, T H, Vec3<T>* P, Vec3<T>* Q) {
  const Vec3<T> p = p0 + dp * t;
  const Vec3<T> rel = p - c;
  const T z = rel.z, rho = Sqrt(rel.x * rel.x + rel.y * rel.y);
  const T az = z < T(0) ? -z : z;
  T g = T(0);
  g += az > H ? (az - H) * (z > T(0) ? dp.z : -dp.z) : T(0);
  g += rho > R ? (rho - R) * (rel.x * dp.x + rel.y * dp.y) / (rho > R ? rho : T(1)) : T(0);
  if (P != nullptr) {
    const T zc = z > H ? H : (z < -H ? -H : z);
    const T sc = rho > R ? R / rho : T(1);
    *P = p;
    *Q = {c.x + sc * rel.x, c.y + sc * rel.y, c.z + zc};
  }
  return g;
}
template <typename T>
struct CapCyl {
  Vec3<T> pos, n;  // contact point, normal from the capsule to the cylinder
  T dist;
};
template <typename T>
EPA_HD CapCyl<T> CapsuleCylinder(Vec3<T> p0, Vec3<T> p1, T rc, Vec3<T> c, T R, T H) {
  const Vec3<T> dp = p1 - p0;
  const T eps = T(1e-10) * Dot(dp, dp);
  const T g0 = CapCylG<T>(p0, dp, T(0), c, R, H, nullptr, nullptr);
  const T g1 = CapCylG<T>(p0, dp, T(1), c, R, H, nullptr, nullptr);
  // ta: the largest t with g < -eps (0 if none)
  T lo = T(0), hi = T(1);
  const bool a_none = g0 >= -eps, a_all = g1 < -eps;
  // tb: the smallest t with g > +eps (1 if none)
  T lo2 = T(0), hi2 = T(1);
  const bool b_none = g1 <= eps, b_all = g0 > eps;
  for (int it = 0; it < 48; ++it) {
    const T mid = T(0.5) * (lo + hi);
    const T mid2 = T(0.5) * (lo2 + hi2);
    // The two brackets coincide until |g| drops below eps (both predicates then agree on every
    // probe): one evaluation serves both searches while they do on every lane of the wave.
    const T gm = CapCylG<T>(p0, dp, mid, c, R, H, nullptr, nullptr);
    const T gm2 = WaveAny(mid2 != mid) ? CapCylG<T>(p0, dp, mid2, c, R, H, nullptr, nullptr) : gm;
    const bool neg = gm < -eps;
    lo = neg ? mid : lo;
    hi = neg ? hi : mid;
    const bool pos = gm2 > eps;
    hi2 = pos ? mid2 : hi2;
    lo2 = pos ? lo2 : mid2;
  }
  const T ta = a_none ? T(0) : (a_all ? T(1) : T(0.5) * (lo + hi));
  const T tb = b_none ? T(1) : (b_all ? T(0) : T(0.5) * (lo2 + hi2));
  const T t = T(0.5) * (ta + tb);
  Vec3<T> P, Q;
  CapCylG<T>(p0, dp, t, c, R, H, &P, &Q);
  Vec3<T> n = Q - P;
  T cd = Sqrt(Dot(n, n));
  const bool inside = cd < T(1e-12);  // the axis itself is inside the solid: push out radially
  const Vec3<T> rad = {-(P.x - c.x), -(P.y - c.y), T(0)};
  const T rn = Sqrt(Dot(rad, rad));
  const Vec3<T> nin = rn < T(1e-12) ? Vec3<T>{T(1), T(0), T(0)} : rad * (T(1) / (rn < T(1e-12) ? T(1) : rn));
  n = inside ? nin : n * (T(1) / (inside ? T(1) : cd));
  cd = inside ? T(0) : cd;
  CapCyl<T> r;
  r.dist = cd - rc;
  r.n = n;
  r.pos = P + n * (rc + T(0.5) * r.dist);
  return r;
}

// One constraint row of the lane: aref, D (D = 0: inactive) in registers; its Jacobian over the dofs
// in the lane's LDS slots [slot][lane] (the nine rows' J, M, H and the link frames together do not
// fit a lane's 512 registers: 267 spilled VGPRs and 5x the algorithmic HBM traffic before).  Table
// rows (wrist sphere vs plane) only involve the 7 arm dofs, the capsule - cylinder rows all 9.
template <typename T>
struct Row {
  T aref, D;
};
EPA_HD constexpr int RowCols(int r) { return r < kNSph ? kNL : kNV; }
EPA_HD constexpr int RowSlot(int r, int i) { return r < kNSph ? r * kNL + i : kNSph * kNL + (r - kNSph) * kNV + i; }
constexpr int kRowSlots = kNSph * kNL + kNCap * kNV;  // 69 slots = 35 KB per 64-lane wave in fp64

// mj_forward: qacc for (q, v) under ctrl; warm = qacc_warmstart in / out.  q[0..6] arm,
// q[7] obj_slidey, q[8] obj_slidex.  Fills `lag` with the xpos the task reads.
template <typename T, typename Lds>
EPA_HD int PusherForward(const PusherModel<T>& m, const SolverCfg<T>& cfg, const T* q, const T* v,
                         const T* ctrl, T* warm, T* qacc, T* Mout, T* qfrc_out,
                         PusherLag<T>* lag, Lds&& lds) {
  // ---- mj_kinematics + mj_comPos: link frames, spatial inertias about the world origin
  Vec3<T> org[kNL], axw[kNL];
  Mat3<T> R = {{T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}};
  Vec3<T> x = {T(0), T(0), T(0)};
  In10<T> ci[kNL];
  static_for<0, kNL>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    x = x + Mul(R, Vec3<T>{m.off[l][0], m.off[l][1], m.off[l][2]});
    org[l] = x;
    constexpr int ax = LinkAxis(l);
    axw[l] = ant::Col(R, ax);  // a hinge axis along a frame axis is unchanged by its own rotation
    R = RotAxis<ax>(R, q[l]);
    // inertia about the link COM rotated into world axes, shifted to the world origin
    const T* I = m.inertia[l];
    Mat3<T> Ib = {{I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}};
    Mat3<T> RI = Mul(R, Ib);
    const T* r = R.m;
    const Vec3<T> d = x + Mul(R, Vec3<T>{m.com[l][0], m.com[l][1], m.com[l][2]});
    const T mass = m.mass[l], d2 = Dot(d, d);
    T* c = ci[l].v;
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
  });
  const Mat3<T> Rw = R;        // wrist frame (r_wrist_roll_link = tips_arm)
  const Vec3<T> xw = x;
  const Vec3<T> objc = {m.obj_pos[0] + q[8], m.obj_pos[1] + q[7], m.obj_pos[2]};
  lag->tips[0] = xw.x;
  lag->tips[1] = xw.y;
  lag->tips[2] = xw.z;
  lag->obj[0] = objc.x;
  lag->obj[1] = objc.y;
  // cdof about the world origin: [axis; axis x (0 - anchor)]
  Sp6<T> cdof[kNL];
  static_for<0, kNL>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    cdof[l] = {axw[l], Cross(org[l], axw[l])};
  });
  // ---- mj_crb: M (full 9 x 9 row major; the object's two slides are decoupled)
  T M[kNV * kNV];
  static_for<0, kNV * kNV>([&](auto kc) { M[decltype(kc)::value] = T(0); });
  {
    In10<T> crb = ci[kNL - 1];
    static_for_down<kNL, 0>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j < kNL - 1) {
        static_for<0, 10>([&](auto kc) { crb.v[decltype(kc)::value] += ci[j].v[decltype(kc)::value]; });
      }
      const Sp6<T> buf = MulInert(crb, cdof[j]);
      static_for<0, j + 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const T mij = Dot(cdof[i], buf);
        M[i * kNV + j] = mij;
        M[j * kNV + i] = mij;
      });
      M[j * kNV + j] += m.arm[j];
    });
    M[7 * kNV + 7] = m.obj_mass + m.arm[7];
    M[8 * kNV + 8] = m.obj_mass + m.arm[8];
  }
  // ---- mj_comVel + mj_rne (gravity 0) + mj_passive + mj_fwdActuation
  T qfrc[kNV];
  {
    Sp6<T> cv = {{T(0), T(0), T(0)}, {T(0), T(0), T(0)}}, ca = cv, cfrc[kNL];
    static_for<0, kNL>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      const Sp6<T> cdd = CrossMotion(cv, cdof[l]);
      ant::Axpy(ca, cdd, v[l]);
      ant::Axpy(cv, cdof[l], v[l]);
      const Sp6<T> f = MulInert(ci[l], ca);
      const Sp6<T> g = CrossForce(cv, MulInert(ci[l], cv));
      cfrc[l] = {f.a + g.a, f.l + g.l};
    });
    static_for_down<kNL, 1>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      cfrc[l - 1].a = cfrc[l - 1].a + cfrc[l].a;
      cfrc[l - 1].l = cfrc[l - 1].l + cfrc[l].l;
    });
    static_for<0, kNL>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      const T c = ctrl[l] < m.ctrl_lo ? m.ctrl_lo : (ctrl[l] > m.ctrl_hi ? m.ctrl_hi : ctrl[l]);
      qfrc[l] = -m.damp[l] * v[l] - Dot(cdof[l], cfrc[l]) + c;  // motors: gear 1
    });
    qfrc[7] = -m.damp[7] * v[7];
    qfrc[8] = -m.damp[8] * v[8];
  }
  // ---- mj_collision + mj_makeConstraint
  // limits of the seven hinges (diagonal rows; the object's range +-10 m is never reached)
  T lsgn[kNL], lD[kNL], laref[kNL];
  static_for<0, kNL>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const T dlo = q[j] - m.lo[j], dhi = m.hi[j] - q[j];
    const T s = dlo < T(0) ? T(1) : (dhi < T(0) ? T(-1) : T(0));
    const T dist = dlo < T(0) ? dlo : (dhi < T(0) ? dhi : T(0));
    const T imp = Impedance(m.imp_d0, m.imp_dmax, m.imp_width, dist);
    const T num = (T(1) - imp) * m.dof_invw[j];  // R = max(mjMINVAL, num / imp)
    const T Dj = num < T(1e-15) * imp ? T(1e15) : imp / num;
    lsgn[j] = s;
    lD[j] = s != T(0) ? Dj : T(0);
    laref[j] = -m.sol_B * (s * v[j]) - m.sol_K * imp * dist;
  });
  // contact rows: frictionless (condim 1), J = n . (Jac_body2 - Jac_body1) at the contact point
  Row<T> rows[kNCon];
  unsigned rmask = 0;  // wave uniform: rows that touch on ANY lane; only those are built and visited
  auto fill_row = [&](auto rc, bool touch, T dist, Vec3<T> pos, Vec3<T> n, T sign_arm, T jy, T jx,
                      T diag) {
    constexpr int r = decltype(rc)::value;
    // arm columns: point Jacobian of `pos` on the wrist link
    T vel = T(0);
    static_for<0, kNL>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      const Vec3<T> col = Cross(axw[l], pos - org[l]);
      const T J = sign_arm * Dot(n, col);
      lds(RowSlot(r, l)) = J;
      vel += J * v[l];
    });
    if constexpr (r >= kNSph) {
      lds(RowSlot(r, 7)) = jy;
      lds(RowSlot(r, 8)) = jx;
      vel += jy * v[7] + jx * v[8];
    }
    const T rr = dist - m.margin;
    const T imp = Impedance(m.imp_d0, m.imp_dmax, m.imp_width, rr);
    const T num = (T(1) - imp) * diag;
    const T invR = num < T(1e-15) * imp ? T(1e15) : imp / num;
    rows[r].D = touch ? invR : T(0);
    rows[r].aref = touch ? -m.sol_B * vel - m.sol_K * imp * rr : T(0);
  };
  {
    // wrist capsule end spheres vs the table (mjc_PlaneCapsule): "+axis" end first
    const int cap_of[kNSph] = {0, 0, 1, 1, 2, 2};
    const int end_of[kNSph] = {1, 0, 1, 0, 1, 0};
    static_for<0, kNSph>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const T* lp = end_of[s] ? m.cap_p1[cap_of[s]] : m.cap_p0[cap_of[s]];
      const Vec3<T> ctr = xw + Mul(Rw, Vec3<T>{lp[0], lp[1], lp[2]});
      const T dist = ctr.z - m.table_z - m.cap_r;
      const bool touch = dist < m.margin;
      rows[s].D = T(0);
      rows[s].aref = T(0);
      if (WaveAny(touch)) {
        rmask |= 1u << s;
        const Vec3<T> pos = {ctr.x, ctr.y, ctr.z - m.cap_r - T(0.5) * dist};
        // geom1 = plane (world), geom2 = capsule: J = +n . Jac_arm, n = +z
        fill_row(sc, touch, dist, pos, Vec3<T>{T(0), T(0), T(1)}, T(1), T(0), T(0), m.wrist_invw);
      }
    });
    // wrist capsules vs the object's cylinder: geom1 = capsule (arm), geom2 = cylinder (object)
    static_for<0, kNCap>([&](auto cc) {
      constexpr int k = decltype(cc)::value;
      const Vec3<T> p0 = xw + Mul(Rw, Vec3<T>{m.cap_p0[k][0], m.cap_p0[k][1], m.cap_p0[k][2]});
      const Vec3<T> p1 = xw + Mul(Rw, Vec3<T>{m.cap_p1[k][0], m.cap_p1[k][1], m.cap_p1[k][2]});
      rows[kNSph + k].D = T(0);
      rows[kNSph + k].aref = T(0);
      // broad phase (MuJoCo's bounding-sphere test): while the capsule's and the cylinder's bounding
      // spheres are further apart than the margin on EVERY lane of the wave, the narrow phase -- 48
      // bisection steps, a third of this kernel's instructions -- cannot produce a contact
      const Vec3<T> dm = (p0 + p1) * T(0.5) - objc, dh = (p1 - p0) * T(0.5);
      const T reach = Sqrt(Dot(dh, dh)) + m.cap_r + Sqrt(m.cyl_r * m.cyl_r + m.cyl_h * m.cyl_h) + m.margin;
      if (!WaveAny(Dot(dm, dm) <= reach * reach)) return;
      const CapCyl<T> cc2 = CapsuleCylinder(p0, p1, m.cap_r, objc, m.cyl_r, m.cyl_h);
      const bool touch = cc2.dist < m.margin;
      if (WaveAny(touch)) {
        rmask |= 1u << (kNSph + k);
        fill_row(IC<kNSph + k>{}, touch, cc2.dist, cc2.pos, cc2.n, T(-1), cc2.n.y, cc2.n.x,
                 m.wrist_invw + m.obj_invw);
      }
    });
  }
  rmask = WaveUniform(rmask);
  // the row's Jacobian out of LDS (columns the row does not have are structural zeros)
  auto load_row = [&](auto rc, T* J) {
    constexpr int r = decltype(rc)::value;
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < RowCols(r)) {
        J[i] = lds(RowSlot(r, i));
      } else {
        J[i] = T(0);
      }
    });
  };
  const bool wave_contact = rmask != 0;
  // ---- mj_fwdConstraint: Newton on 1/2 (a-a0)^T M (a-a0) + sum 1/2 D min(0, J a - aref)^2
  static_for<0, kNV>([&](auto ic) { qacc[decltype(ic)::value] = warm[decltype(ic)::value]; });
  T fs = T(0);
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const T ax = qfrc[i] < T(0) ? -qfrc[i] : qfrc[i];
    fs = ax > fs ? ax : fs;
  });
  const T gstop = cfg.gtol * (T(1) + fs), gstop2 = gstop * gstop;
  int iter = 0;
  bool live = true, full_step = false;
  unsigned prev_mask = ~0u;
  T grad[kNV];
  for (int it = 0; it < cfg.max_iter; ++it) {
    T H[kNV * kNV];
    static_for<0, kNV * kNV>([&](auto kc) { H[decltype(kc)::value] = M[decltype(kc)::value]; });
    unsigned mask = 0;
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      T g = -qfrc[i];
      static_for<0, kNV>([&](auto jc) { g += M[i * kNV + decltype(jc)::value] * qacc[decltype(jc)::value]; });
      grad[i] = g;
    });
    static_for<0, kNL>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const T jar = lsgn[j] * qacc[j] - laref[j];
      const bool on = lsgn[j] != T(0) && jar < T(0);
      const T w = on ? lD[j] : T(0);
      grad[j] += lsgn[j] * w * jar;
      H[j * kNV + j] += w;
      mask |= (on ? 1u : 0u) << j;
    });
    if (wave_contact) {
      static_for<0, kNCon>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if ((rmask >> r) & 1u) {  // wave uniform
          T J[kNV];
          load_row(rc, J);
          T jar = -rows[r].aref;
          static_for<0, RowCols(r)>([&](auto ic) { jar += J[decltype(ic)::value] * qacc[decltype(ic)::value]; });
          const bool on = rows[r].D > T(0) && jar < T(0);
          const T w = on ? rows[r].D : T(0);
          mask |= (on ? 1u : 0u) << (kNL + r);
          static_for<0, RowCols(r)>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            grad[i] += J[i] * w * jar;
            const T wi = w * J[i];
            static_for<0, RowCols(r)>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              H[i * kNV + j] += wi * J[j];
            });
          });
        }
      });
    }
    T gn2 = T(0);
    static_for<0, kNV>([&](auto ic) { gn2 += grad[decltype(ic)::value] * grad[decltype(ic)::value]; });
    const bool stop = gn2 <= gstop2 || (full_step && mask == prev_mask);
    live = live && !stop;
    if (!WaveAny(live)) break;
    iter += live ? 1 : 0;
    prev_mask = mask;
    T s[kNV];
    static_for<0, kNV>([&](auto ic) { s[decltype(ic)::value] = -grad[decltype(ic)::value]; });
    CholSolveN<T, kNV>(H, s);
    // exact line search along s on the piecewise quadratic
    T g1 = T(0), g2 = T(0);
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      T ms = T(0), ma = -qfrc[i];
      static_for<0, kNV>([&](auto jc) {
        ms += M[i * kNV + decltype(jc)::value] * s[decltype(jc)::value];
        ma += M[i * kNV + decltype(jc)::value] * qacc[decltype(jc)::value];
      });
      g1 += s[i] * ma;
      g2 += s[i] * ms;
    });
    // per-row J a - aref and J s do not change during the search
    T cjar[kNCon], cjv[kNCon];
    if (wave_contact) {
      static_for<0, kNCon>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        T a = -rows[r].aref, b = T(0);
        if ((rmask >> r) & 1u) {  // wave uniform; rows nobody touches keep D = 0 and a = b = 0
          T J[kNV];
          load_row(rc, J);
          static_for<0, RowCols(r)>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            a += J[i] * qacc[i];
            b += J[i] * s[i];
          });
        }
        cjar[r] = a;
        cjv[r] = b;
      });
    }
    T alpha = T(1), lo = T(0), hi = T(-1);
    full_step = false;
    const T ag1 = g1 < T(0) ? -g1 : g1;
    const T ls_tol = T(1e-10) * ag1;
    bool searching = live, exact = false;
    for (int ls = 0; ls < 24; ++ls) {
      T d1 = g1 + alpha * g2, d2 = g2;
      unsigned mask1 = 0;  // the rows active at qacc + alpha s (the bits of `mask`)
      static_for<0, kNL>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const T jar = lsgn[j] * qacc[j] - laref[j], jv = lsgn[j] * s[j];
        const T xx = jar + alpha * jv;
        const bool on = lsgn[j] != T(0) && xx < T(0);
        const T w = on ? lD[j] : T(0);
        d1 += w * xx * jv;
        d2 += w * jv * jv;
        mask1 |= (on ? 1u : 0u) << j;
      });
      if (wave_contact) {
        static_for<0, kNCon>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          const T xx = cjar[r] + alpha * cjv[r];
          const bool on = rows[r].D > T(0) && xx < T(0);
          const T w = on ? rows[r].D : T(0);
          d1 += w * xx * cjv[r];
          d2 += w * cjv[r] * cjv[r];
          mask1 |= (on ? 1u : 0u) << (kNL + r);
        });
      }
      const T ad1 = d1 < T(0) ? -d1 : d1;
      const bool hit = ad1 <= ls_tol;
      full_step = full_step || (searching && hit && ls == 0);
      // finite termination: the full Newton step keeps the active set H was built with, so it lands
      // on the minimiser; no further pass over the rows is needed to find that out
      exact = exact || (searching && hit && ls == 0 && mask1 == mask);
      searching = searching && !hit;
      lo = (searching && d1 < T(0)) ? alpha : lo;
      hi = (searching && !(d1 < T(0))) ? alpha : hi;
      T next = alpha - d1 / d2;
      next = (hi >= T(0) && (next <= lo || next >= hi)) ? T(0.5) * (lo + hi) : next;
      next = next <= T(0) ? T(0.5) * alpha : next;
      searching = searching && next != alpha;
      alpha = searching ? next : alpha;
      if (!WaveAny(searching)) break;
    }
    const T step = live ? alpha : T(0);
    static_for<0, kNV>([&](auto ic) { qacc[decltype(ic)::value] += step * s[decltype(ic)::value]; });
    live = live && !exact;
    if (!WaveAny(live)) break;
  }
  static_for<0, kNV>([&](auto ic) { warm[decltype(ic)::value] = qacc[decltype(ic)::value]; });
  // qfrc_smooth + qfrc_constraint for the integrator: M qacc at the solution
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    T ma = T(0);
    static_for<0, kNV>([&](auto jc) { ma += M[i * kNV + decltype(jc)::value] * qacc[decltype(jc)::value]; });
    qfrc_out[i] = ma;
  });
  static_for<0, kNV * kNV>([&](auto kc) { Mout[decltype(kc)::value] = M[decltype(kc)::value]; });
  return iter;
}

// One mj_step, mj_Euler with implicit joint damping (eulerdamp):
//   (M + h diag(damping)) qacc_d = qfrc_smooth + qfrc_constraint = M qacc
template <typename T, typename Lds>
EPA_HD int PusherStep(const PusherModel<T>& m, const SolverCfg<T>& cfg, T* q, T* v, T* warm,
                      const T* ctrl, PusherLag<T>* lag, Lds&& lds) {
  T qacc[kNV], M[kNV * kNV], rhs[kNV];
  const int it = PusherForward(m, cfg, q, v, ctrl, warm, qacc, M, rhs, lag, lds);
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    M[i * kNV + i] += m.timestep * m.damp[i];
  });
  CholSolveN<T, kNV>(M, rhs);
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v[i] += m.timestep * rhs[i];
    q[i] += m.timestep * v[i];
  });
  return it;
}

}  // namespace pusher
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_PUSHER_HIP_H_
