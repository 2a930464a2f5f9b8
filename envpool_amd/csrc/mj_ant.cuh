// K3b — Ant `mj_step` (free joint + 8 hinges, sphere/capsule-vs-plane contacts,
// RK4) restated as a tree-specialised, statically unrolled per-thread routine.
//
// Replaces the arithmetic MuJoCo 3.6.0's mj_step performs for the gym Ant model
// (third_party/mujoco_gym_xml_patches/ant_envpool.xml) each time the reference
// calls it (envpool/mujoco/gym/mujoco_env.h:137-148): SURVEY.md §8a M1-M9 with
// integrator="RK4" (4 forward evaluations per mj_step).
//
// Design notes (MI355X-first, not MuJoCo's generic engine):
//  * topology is compile time: torso (free) + 4 x (aux: hip hinge, foot: ankle
//    hinge); the four jointless "leg" bodies of the XML are welded into the
//    torso for the dynamics (identical physics) while their own
//    body_invweight0 is kept for the contact regulariser;
//  * spatial vectors use MuJoCo's c-frame (world orientation, about the robot
//    COM), so the mass matrix / RNE recursions need no frame transforms;
//  * M/H have an arrow structure (torso 6x6, four 2x2 leg blocks coupled only
//    through the torso); entries between different legs are never materialised
//    and the U U^T factorisation in tree order has no fill;
//  * contacts are the 25 end spheres (torso sphere + 2 per capsule); the four
//    pyramidal rows of a contact share Jx/Jy/Jz, so gradient and Hessian
//    updates are accumulated in the 3x3 "contact space" first;
//  * same exact-Newton solver with finite termination as mj_cheetah.cuh.
#ifndef ENVPOOL_AMD_CSRC_MJ_ANT_CUH_
#define ENVPOOL_AMD_CSRC_MJ_ANT_CUH_

#include "mj_cheetah.cuh"  // static_for, IC, Sqrt, SinCos, Impedance (generic part)

namespace epa {
namespace mj {
namespace ant {

constexpr int kNQ = 15, kNV = 14, kNU = 8, kNB = 9, kNLeg = 4;
constexpr int kNSph = 25;       // torso sphere + 4 x 6 capsule end spheres
constexpr int kNGeomBody = 13;  // MuJoCo bodies carrying geoms (for invweight)

// bodies: 0 torso, 1+2l aux_l, 2+2l foot_l.  dofs: 0-2 trans, 3-5 rot,
// 6+2l hip_l, 7+2l ankle_l.
EPA_HD constexpr int Aux(int l) { return 1 + 2 * l; }
EPA_HD constexpr int Foot(int l) { return 2 + 2 * l; }
EPA_HD constexpr int Hip(int l) { return 6 + 2 * l; }
EPA_HD constexpr int Ankle(int l) { return 7 + 2 * l; }
EPA_HD constexpr int Parent(int b) { return b == 0 ? -1 : ((b & 1) ? 0 : b - 1); }
EPA_HD constexpr int DofBody(int j) { return j < 6 ? 0 : j - 5; }
EPA_HD constexpr int LegOfDof(int j) { return j < 6 ? -1 : (j - 6) / 2; }
EPA_HD constexpr bool NZ(int i, int j) {
  return i < 6 || j < 6 || LegOfDof(i) == LegOfDof(j);
}
EPA_HD constexpr bool InChain(int j, int b) {  // dof j moves body b
  if (j < 6) return true;
  int jb = DofBody(j);
  for (int x = b; x > 0; x = Parent(x)) {
    if (x == jb) return true;
  }
  return false;
}
// actuator order of the XML (:85-94): hip_4 ankle_4 hip_1 ankle_1 hip_2 ankle_2
// hip_3 ankle_3 -> dof driven by ctrl[u]
EPA_HD constexpr int CtrlDof(int u) { return u < 2 ? 12 + u : 4 + u; }
EPA_HD constexpr int Tri(int i, int j) { return j * (j + 1) / 2 + i; }  // i <= j
constexpr int kTri = kNV * (kNV + 1) / 2;
// sphere s: 0 torso sphere; 1 + 6l + {0,1}: stub capsule (torso frame),
// {2,3}: leg capsule (aux frame), {4,5}: ankle capsule (foot frame)
EPA_HD constexpr int SphBody(int s) {
  if (s == 0) return 0;
  int l = (s - 1) / 6, w = (s - 1) % 6;
  return w < 2 ? 0 : (w < 4 ? Aux(l) : Foot(l));
}
// index of the MuJoCo geom body (for body_invweight0): 0 torso, 1+3l stub,
// 2+3l aux, 3+3l foot
EPA_HD constexpr int SphGeomBody(int s) {
  if (s == 0) return 0;
  int l = (s - 1) / 6, w = (s - 1) % 6;
  return 1 + 3 * l + w / 2;
}

template <typename T>
struct AntModel {
  T mass[kNB], com[kNB][3], inertia[kNB][6];  // xx yy zz xy xz yz, about com
  T aux_pos[kNLeg][3], foot_pos[kNLeg][3];    // body_pos in the parent frame
  T ankle_axis[kNLeg][3];                     // in the foot/aux frame (hip: +z)
  T sph[kNSph][3], sph_r[kNSph];
  T geom_body_invw[kNGeomBody];
  T lo[kNU], hi[kNU], dof_invw[kNU], damp[kNU], arm[kNU];
  T gear;
  T total_mass, mu, margin;
  T con_K, con_B, imp_d0, imp_dmax, imp_width;
  T timestep, gravity;
};

template <typename T>
struct Vec3 {
  T x, y, z;
};
template <typename T>
EPA_HD Vec3<T> operator+(Vec3<T> a, Vec3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T>
EPA_HD Vec3<T> operator-(Vec3<T> a, Vec3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T>
EPA_HD Vec3<T> operator*(Vec3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T>
EPA_HD T Dot(Vec3<T> a, Vec3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T>
EPA_HD Vec3<T> Cross(Vec3<T> a, Vec3<T> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T>
struct Mat3 {  // row major
  T m[9];
};
template <typename T>
EPA_HD Vec3<T> Mul(const Mat3<T>& R, Vec3<T> v) {
  return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z,
          R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
          R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
template <typename T>
EPA_HD Mat3<T> Mul(const Mat3<T>& A, const Mat3<T>& B) {
  Mat3<T> C;
  static_for<0, 3>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    static_for<0, 3>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] +
                       A.m[3 * i + 2] * B.m[6 + j];
    });
  });
  return C;
}
template <typename T>
EPA_HD Vec3<T> Col(const Mat3<T>& R, int k) { return {R.m[k], R.m[3 + k], R.m[6 + k]}; }
template <typename T>
EPA_HD Mat3<T> QuatToMat(T w, T x, T y, T z) {
  Mat3<T> M;
  M.m[0] = w * w + x * x - y * y - z * z;
  M.m[4] = w * w - x * x + y * y - z * z;
  M.m[8] = w * w - x * x - y * y + z * z;
  M.m[1] = T(2) * (x * y - w * z);
  M.m[2] = T(2) * (x * z + w * y);
  M.m[3] = T(2) * (x * y + w * z);
  M.m[5] = T(2) * (y * z - w * x);
  M.m[6] = T(2) * (x * z - w * y);
  M.m[7] = T(2) * (y * z + w * x);
  return M;
}
// rotation by `ang` about unit axis a (Rodrigues)
template <typename T>
EPA_HD Mat3<T> AxisAngle(const T* a, T ang) {
  T s, c;
  SinCos(ang, &s, &c);
  T t = T(1) - c;
  Mat3<T> M;
  M.m[0] = c + a[0] * a[0] * t;
  M.m[4] = c + a[1] * a[1] * t;
  M.m[8] = c + a[2] * a[2] * t;
  M.m[1] = a[0] * a[1] * t - a[2] * s;
  M.m[3] = a[0] * a[1] * t + a[2] * s;
  M.m[2] = a[0] * a[2] * t + a[1] * s;
  M.m[6] = a[0] * a[2] * t - a[1] * s;
  M.m[5] = a[1] * a[2] * t - a[0] * s;
  M.m[7] = a[1] * a[2] * t + a[0] * s;
  return M;
}

template <typename T>
struct Sp6 {  // spatial motion [w; v] or force [tau; f]
  Vec3<T> a, l;
};
template <typename T>
EPA_HD T Dot(const Sp6<T>& p, const Sp6<T>& q) { return Dot(p.a, q.a) + Dot(p.l, q.l); }
template <typename T>
EPA_HD void Axpy(Sp6<T>& y, const Sp6<T>& x, T s) {
  y.a = y.a + x.a * s;
  y.l = y.l + x.l * s;
}
template <typename T>
struct In10 {  // xx yy zz xy xz yz mdx mdy mdz m
  T v[10];
};
template <typename T>
EPA_HD Sp6<T> MulInert(const In10<T>& I, const Sp6<T>& s) {
  const T* i = I.v;
  Sp6<T> r;
  r.a.x = i[0] * s.a.x + i[3] * s.a.y + i[4] * s.a.z - i[8] * s.l.y + i[7] * s.l.z;
  r.a.y = i[3] * s.a.x + i[1] * s.a.y + i[5] * s.a.z + i[8] * s.l.x - i[6] * s.l.z;
  r.a.z = i[4] * s.a.x + i[5] * s.a.y + i[2] * s.a.z - i[7] * s.l.x + i[6] * s.l.y;
  r.l.x = i[8] * s.a.y - i[7] * s.a.z + i[9] * s.l.x;
  r.l.y = i[6] * s.a.z - i[8] * s.a.x + i[9] * s.l.y;
  r.l.z = i[7] * s.a.x - i[6] * s.a.y + i[9] * s.l.z;
  return r;
}
template <typename T>
EPA_HD Sp6<T> CrossMotion(const Sp6<T>& vel, const Sp6<T>& v) {
  return {Cross(vel.a, v.a), Cross(vel.a, v.l) + Cross(vel.l, v.a)};
}
template <typename T>
EPA_HD Sp6<T> CrossForce(const Sp6<T>& vel, const Sp6<T>& f) {
  return {Cross(vel.a, f.a) + Cross(vel.l, f.l), Cross(vel.a, f.l)};
}

// ---- linear algebra on the arrow-structured 14x14 ---------------------------
template <typename T>
EPA_HD void FactorUUt(T* A) {
  static_for_down<kNV, 0>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T s = A[Tri(j, j)];
    static_for<j + 1, kNV>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (NZ(j, k)) s -= A[Tri(j, k)] * A[Tri(j, k)];
    });
    T inv = Rsqrt(s);
    A[Tri(j, j)] = inv;  // the diagonal holds 1 / U_jj
    static_for<0, j>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (NZ(i, j)) {
        T t = A[Tri(i, j)];
        static_for<j + 1, kNV>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (NZ(i, k) && NZ(j, k)) t -= A[Tri(i, k)] * A[Tri(j, k)];
        });
        A[Tri(i, j)] = t * inv;
      }
    });
  });
}
template <typename T>
EPA_HD void SolveUUt(const T* U, T* x) {
  static_for_down<kNV, 0>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T s = x[j];
    static_for<j + 1, kNV>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (NZ(j, k)) s -= U[Tri(j, k)] * x[k];
    });
    x[j] = s * U[Tri(j, j)];
  });
  static_for<0, kNV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T s = x[j];
    static_for<0, j>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (NZ(i, j)) s -= U[Tri(i, j)] * x[i];
    });
    x[j] = s * U[Tri(j, j)];
  });
}
template <typename T>
EPA_HD void SymMul(const T* A, const T* x, T* y) {
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    T s = T(0);
    static_for<0, kNV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (NZ(i, j)) s += A[i <= j ? Tri(i, j) : Tri(j, i)] * x[j];
    });
    y[i] = s;
  });
}

// ---- forward pass -------------------------------------------------------------
template <typename T>
struct AntPos {
  Vec3<T> pos[kNB];
  Mat3<T> R[kNB];
  Vec3<T> com;
  In10<T> cinert[kNB];
  Sp6<T> cdof[kNV];  // entries 0..2 are the constant (0; e_k)
  T M[kTri];
};

template <typename T>
EPA_HD void NormalizeQuat(T* q) {
  T n = Sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const bool tiny = n < T(1e-15);
  const T inv = T(1) / (tiny ? T(1) : n);
  q[0] = tiny ? T(1) : q[0] * inv;
  q[1] = tiny ? T(0) : q[1] * inv;
  q[2] = tiny ? T(0) : q[2] * inv;
  q[3] = tiny ? T(0) : q[3] * inv;
}

template <typename T>
EPA_HD void AntKinematics(const AntModel<T>& m, T* q, AntPos<T>& p) {
  // mj_kinematics (normalises the free-joint quaternion in qpos)
  NormalizeQuat(q + 3);
  p.pos[0] = {q[0], q[1], q[2]};
  p.R[0] = QuatToMat(q[3], q[4], q[5], q[6]);
  const T zaxis[3] = {T(0), T(0), T(1)};
  static_for<0, kNLeg>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    constexpr int A = Aux(l), F = Foot(l);
    p.pos[A] = p.pos[0] + Mul(p.R[0], Vec3<T>{m.aux_pos[l][0], m.aux_pos[l][1], m.aux_pos[l][2]});
    p.R[A] = Mul(p.R[0], AxisAngle(zaxis, q[7 + 2 * l]));
    p.pos[F] = p.pos[A] + Mul(p.R[A], Vec3<T>{m.foot_pos[l][0], m.foot_pos[l][1], m.foot_pos[l][2]});
    p.R[F] = Mul(p.R[A], AxisAngle(m.ankle_axis[l], q[8 + 2 * l]));
  });
  // mj_comPos
  Vec3<T> xi[kNB];
  Vec3<T> s = {T(0), T(0), T(0)};
  static_for<0, kNB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    xi[b] = p.pos[b] + Mul(p.R[b], Vec3<T>{m.com[b][0], m.com[b][1], m.com[b][2]});
    s = s + xi[b] * m.mass[b];
  });
  p.com = s * (T(1) / m.total_mass);
  static_for<0, kNB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    const T* I = m.inertia[b];
    Mat3<T> Ib = {{I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}};
    Mat3<T> RI = Mul(p.R[b], Ib);
    // Iw = RI * R^T (symmetric)
    T w[6];
    const T* r = p.R[b].m;
    w[0] = RI.m[0] * r[0] + RI.m[1] * r[1] + RI.m[2] * r[2];
    w[1] = RI.m[3] * r[3] + RI.m[4] * r[4] + RI.m[5] * r[5];
    w[2] = RI.m[6] * r[6] + RI.m[7] * r[7] + RI.m[8] * r[8];
    w[3] = RI.m[0] * r[3] + RI.m[1] * r[4] + RI.m[2] * r[5];
    w[4] = RI.m[0] * r[6] + RI.m[1] * r[7] + RI.m[2] * r[8];
    w[5] = RI.m[3] * r[6] + RI.m[4] * r[7] + RI.m[5] * r[8];
    Vec3<T> d = xi[b] - p.com;
    T mass = m.mass[b], d2 = Dot(d, d);
    T* c = p.cinert[b].v;
    c[0] = w[0] + mass * (d2 - d.x * d.x);
    c[1] = w[1] + mass * (d2 - d.y * d.y);
    c[2] = w[2] + mass * (d2 - d.z * d.z);
    c[3] = w[3] - mass * d.x * d.y;
    c[4] = w[4] - mass * d.x * d.z;
    c[5] = w[5] - mass * d.y * d.z;
    c[6] = mass * d.x;
    c[7] = mass * d.y;
    c[8] = mass * d.z;
    c[9] = mass;
  });
  // cdof (c-frame: about the robot COM)
  p.cdof[0] = {{T(0), T(0), T(0)}, {T(1), T(0), T(0)}};
  p.cdof[1] = {{T(0), T(0), T(0)}, {T(0), T(1), T(0)}};
  p.cdof[2] = {{T(0), T(0), T(0)}, {T(0), T(0), T(1)}};
  static_for<0, 3>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    Vec3<T> ax = Col(p.R[0], k);
    p.cdof[3 + k] = {ax, Cross(ax, p.com - p.pos[0])};
  });
  static_for<0, kNLeg>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    Vec3<T> hz = Col(p.R[0], 2);  // hip axis: +z of the aux frame = torso z
    p.cdof[Hip(l)] = {hz, Cross(hz, p.com - p.pos[Aux(l)])};
    Vec3<T> ha = Mul(p.R[Aux(l)], Vec3<T>{m.ankle_axis[l][0], m.ankle_axis[l][1], m.ankle_axis[l][2]});
    p.cdof[Ankle(l)] = {ha, Cross(ha, p.com - p.pos[Foot(l)])};
  });
  // mj_crb
  In10<T> crb[kNB];
  static_for<0, kNB>([&](auto bc) { crb[decltype(bc)::value] = p.cinert[decltype(bc)::value]; });
  static_for_down<kNB, 1>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    static_for<0, 10>([&](auto kc) { crb[Parent(b)].v[decltype(kc)::value] += crb[b].v[decltype(kc)::value]; });
  });
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    Sp6<T> buf = MulInert(crb[DofBody(i)], p.cdof[i]);
    static_for<0, i + 1>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (InChain(j, DofBody(i))) p.M[Tri(j, i)] = Dot(p.cdof[j], buf);
    });
    if constexpr (i >= 6) p.M[Tri(i, i)] += m.arm[i - 6];
  });
}

template <typename T>
EPA_HD void AntSmoothForces(const AntModel<T>& m, const AntPos<T>& p, const T* v,
                            const T* ctrl, T* qfrc_smooth) {
  // mj_comVel: free joint = 3 translations (cdof_dot = 0) then 3 rotations
  // whose cdof_dot all use the velocity before the rotations are added
  Sp6<T> cvel[kNB], cdd[kNV];
  {
    Sp6<T> cv = {{T(0), T(0), T(0)}, {v[0], v[1], v[2]}};
    static_for<0, 3>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      cdd[k] = {{T(0), T(0), T(0)}, {T(0), T(0), T(0)}};
      cdd[3 + k] = CrossMotion(cv, p.cdof[3 + k]);
    });
    static_for<3, 6>([&](auto kc) { Axpy(cv, p.cdof[decltype(kc)::value], v[decltype(kc)::value]); });
    cvel[0] = cv;
  }
  static_for<1, kNB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr int j = b + 5;
    Sp6<T> cv = cvel[Parent(b)];
    cdd[j] = CrossMotion(cv, p.cdof[j]);
    Axpy(cv, p.cdof[j], v[j]);
    cvel[b] = cv;
  });
  // mj_rne, flg_acc = 0
  Sp6<T> cacc[kNB], cfrc[kNB];
  static_for<0, kNB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    Sp6<T> a;
    if constexpr (b == 0) {
      a = {{T(0), T(0), T(0)}, {T(0), T(0), m.gravity}};
      static_for<3, 6>([&](auto kc) { Axpy(a, cdd[decltype(kc)::value], v[decltype(kc)::value]); });
    } else {
      a = cacc[Parent(b)];
      Axpy(a, cdd[b + 5], v[b + 5]);
    }
    cacc[b] = a;
    Sp6<T> f = MulInert(p.cinert[b], a);
    Sp6<T> g = CrossForce(cvel[b], MulInert(p.cinert[b], cvel[b]));
    cfrc[b] = {f.a + g.a, f.l + g.l};
  });
  static_for_down<kNB, 1>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    cfrc[Parent(b)].a = cfrc[Parent(b)].a + cfrc[b].a;
    cfrc[Parent(b)].l = cfrc[Parent(b)].l + cfrc[b].l;
  });
  static_for<0, kNV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T bias = Dot(p.cdof[j], cfrc[DofBody(j)]);
    if constexpr (j < 6) {
      qfrc_smooth[j] = -bias;
    } else {
      qfrc_smooth[j] = -m.damp[j - 6] * v[j] - bias;  // hinge damper (stiffness 0)
    }
  });
  static_for<0, kNU>([&](auto uc) {  // motors: gear * clamp(ctrl)
    constexpr int u = decltype(uc)::value;
    qfrc_smooth[CtrlDof(u)] += m.gear * ctrl[u];
  });
}

// Everything the constraint passes need from a forward pass, 60 numbers instead
// of the full AntPos (27 + 81 + 66 + ...): joint anchors are body origins, every
// capsule end sphere sits on a body origin (stub: torso -> aux, leg: aux -> foot)
// except the four foot tips, and a Jacobian column is axis x (point - anchor).
template <typename T>
struct AntGeo {
  Vec3<T> pos[kNB];       // body origins = hinge anchors
  Vec3<T> tip[kNLeg];     // far end of the ankle capsules
  Vec3<T> rot[3];         // torso body axes (free-joint rotational dofs); rot[2] = hip axis
  Vec3<T> ankle[kNLeg];   // ankle axes in the world
  EPA_HD Vec3<T> Pos(int b) const { return pos[b]; }
  EPA_HD Vec3<T> Tip(int l) const { return tip[l]; }
  EPA_HD Vec3<T> Rot(int k) const { return rot[k]; }
  EPA_HD Vec3<T> AnkleAxis(int l) const { return ankle[l]; }
};

// ---- per-lane LDS block of a forward pass ---------------------------------------
// The solver's working set (M 81 + H 81 + geometry 60 + rows + five 14-vectors)
// is far beyond 512 registers in fp64, and a first version that kept it all in
// "registers" moved 24 GB of scratch per launch.  M and the geometry are written
// once per forward pass and only read afterwards, so they live in LDS
// (slot-major [slot][lane]: conflict free, lane-private => no barriers needed):
//   slots [0, 81)    structurally non-zero entries of M (MSlot)
//   slots [81, 141)  AntGeo (pos 27, tip 12, rot 9, ankle 12)
//   slots [141, 144) subtree COM of the robot (reference point of cfrc_ext)
// fp64: 144 * 64 * 8 B = 72 KB per wave (2 waves per CU); fp32: 36 KB (4 per CU).
EPA_HD constexpr int MSlot(int i, int j) {  // i <= j, NZ(i, j)
  int n = 0;
  for (int jj = 0; jj < kNV; ++jj) {
    for (int ii = 0; ii <= jj; ++ii) {
      if (ii == i && jj == j) return n;
      if (NZ(ii, jj)) ++n;
    }
  }
  return n;
}
constexpr int kMSlots = MSlot(kNV - 1, kNV - 1) + 1;  // 81
constexpr int kGeoBase = kMSlots;
constexpr int kGeoPos = kGeoBase, kGeoTip = kGeoPos + 3 * kNB, kGeoRot = kGeoTip + 3 * kNLeg,
              kGeoAnkle = kGeoRot + 9;
constexpr int kGeoCom = kGeoAnkle + 3 * kNLeg;
constexpr int kAntLdsSlots = kGeoCom + 3;  // 144

// Compiler-level fence: LDS contents must not be carried in registers across it
// (otherwise the loads get hoisted out of the solver loops / forwarded from the
// stores and everything lands in VGPRs -> scratch again).
#if defined(__HIP_DEVICE_COMPILE__)
#define EPA_LDS_FENCE() asm volatile("" ::: "memory")
#else
#define EPA_LDS_FENCE() ((void)0)
#endif

template <typename T, typename Lds>
struct AntGeoLds {
  Lds& lds;
  EPA_HD Vec3<T> At(int base) const { return {lds(base), lds(base + 1), lds(base + 2)}; }
  EPA_HD Vec3<T> Pos(int b) const { return At(kGeoPos + 3 * b); }
  EPA_HD Vec3<T> Tip(int l) const { return At(kGeoTip + 3 * l); }
  EPA_HD Vec3<T> Rot(int k) const { return At(kGeoRot + 3 * k); }
  EPA_HD Vec3<T> AnkleAxis(int l) const { return At(kGeoAnkle + 3 * l); }
};

template <typename T>
EPA_HD void AntMakeGeo(const AntModel<T>& m, const AntPos<T>& p, AntGeo<T>& g) {
  static_for<0, kNB>([&](auto bc) { g.pos[decltype(bc)::value] = p.pos[decltype(bc)::value]; });
  static_for<0, 3>([&](auto kc) { g.rot[decltype(kc)::value] = p.cdof[3 + decltype(kc)::value].a; });
  static_for<0, kNLeg>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    constexpr int s = 1 + 6 * l + 4;  // "+axis" end of the ankle capsule
    g.tip[l] = p.pos[Foot(l)] + Mul(p.R[Foot(l)], Vec3<T>{m.sph[s][0], m.sph[s][1], m.sph[s][2]});
    g.ankle[l] = p.cdof[Ankle(l)].a;
  });
}

// world centre of end sphere s (runtime, wave uniform) known to sit on body B
template <int B, typename G>
EPA_HD auto SphCenter(const G& g, int s) -> decltype(g.Pos(0)) {
  const int l = (s - 1) / 6, w = (s - 1) % 6;
  if constexpr (B == 0) {
    // s is wave uniform: scalar branches
    if (s == 0 || w == 1) return g.Pos(0);
    if (l == 1) return g.Pos(Aux(1));
    if (l == 2) return g.Pos(Aux(2));
    if (l == 3) return g.Pos(Aux(3));
    return g.Pos(Aux(0));
  } else if constexpr ((B & 1) == 1) {  // aux_l: leg capsule aux -> foot
    if (w == 2) return g.Pos(B + 1);
    return g.Pos(B);
  } else {  // foot_l: ankle capsule foot -> tip
    if (w == 4) return g.Tip((B - 2) / 2);
    return g.Pos(B);
  }
}

// Publishes M and the geometry of a forward pass into the lane's LDS block and
// returns the set of end spheres (bit s) that are within the contact margin on
// any lane of the wave; the solver passes only visit those.
template <typename T, typename Lds>
EPA_HD unsigned AntPublish(const AntModel<T>& m, const AntPos<T>& p, Lds&& lds) {
  static_for<0, kNV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    static_for<0, j + 1>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (NZ(i, j)) {
        constexpr int slot = MSlot(i, j);  // forced compile-time evaluation
        lds(slot) = p.M[Tri(i, j)];
      }
    });
  });
  AntGeo<T> g;
  AntMakeGeo(m, p, g);
  auto put = [&](int base, Vec3<T> v) {
    lds(base) = v.x;
    lds(base + 1) = v.y;
    lds(base + 2) = v.z;
  };
  static_for<0, kNB>([&](auto bc) { put(kGeoPos + 3 * decltype(bc)::value, g.pos[decltype(bc)::value]); });
  static_for<0, kNLeg>([&](auto lc) {
    put(kGeoTip + 3 * decltype(lc)::value, g.tip[decltype(lc)::value]);
    put(kGeoAnkle + 3 * decltype(lc)::value, g.ankle[decltype(lc)::value]);
  });
  static_for<0, 3>([&](auto kc) { put(kGeoRot + 3 * decltype(kc)::value, g.rot[decltype(kc)::value]); });
  unsigned mask = 0;
  static_for<0, kNSph>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    const T z = SphCenter<SphBody(s)>(g, s).z;
    if (WaveAny(z - m.sph_r[s] < m.margin)) mask |= 1u << s;
  });
  return WaveUniform(mask);
}

// columns of the point Jacobian (3 x nv) of `cp` attached to body B:
// f(j, col) for every chain dof j with col = d(point velocity)/d(qdot_j).
template <int B, typename G, typename T, typename F>
EPA_HD void ForChainCols(const G& g, Vec3<T> cp, F&& f) {
  f(IC<0>{}, Vec3<T>{T(1), T(0), T(0)});
  f(IC<1>{}, Vec3<T>{T(0), T(1), T(0)});
  f(IC<2>{}, Vec3<T>{T(0), T(0), T(1)});
  static_for<0, 3>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    f(IC<3 + k>{}, Cross(g.Rot(k), cp - g.Pos(0)));
  });
  if constexpr (B > 0) {
    constexpr int l = (B - 1) / 2;
    f(IC<Hip(l)>{}, Cross(g.Rot(2), cp - g.Pos(Aux(l))));
    if constexpr (B == Foot(l)) {
      f(IC<Ankle(l)>{}, Cross(g.AnkleAxis(l), cp - g.Pos(Foot(l))));
    }
  }
}

template <typename T>
struct AntRows {  // joint-limit rows (contacts are re-derived per pass, see below)
  T lim_sgn[kNU], lim_aref[kNU], lim_D[kNU];
};

template <typename T>
EPA_HD void AntMakeConstraint(const AntModel<T>& m, const AntPos<T>& p, const T* q,
                              const T* v, AntRows<T>& r) {
  const T kMinVal = T(1e-15);
  static_for<0, kNU>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T qq = q[7 + j];
    T dlo = qq - m.lo[j], dhi = m.hi[j] - qq;
    const T sgn = dlo < T(0) ? T(1) : (dhi < T(0) ? T(-1) : T(0));
    const T dist = dlo < T(0) ? dlo : (dhi < T(0) ? dhi : T(0));
    T imp = Impedance(m.imp_d0, m.imp_dmax, m.imp_width, dist);
    const T num = (T(1) - imp) * m.dof_invw[j];  // R = max(mjMINVAL, num / imp)
    const T Dj = num < kMinVal * imp ? T(1) / kMinVal : imp / num;
    r.lim_sgn[j] = sgn;
    r.lim_D[j] = sgn != T(0) ? Dj : T(0);
    r.lim_aref[j] = -m.con_B * (sgn * v[6 + j]) - m.con_K * imp * dist;
  });
}

#if defined(__clang__)
#define EPA_ANT_NO_UNROLL _Pragma("clang loop unroll(disable)")
#else
#define EPA_ANT_NO_UNROLL
#endif
// wave-uniform switch on the body a sphere is attached to
template <typename F>
EPA_HD void DispatchBody(int b, F&& f) {
  switch (b) {
    case 0: f(IC<0>{}); break;
    case 1: f(IC<1>{}); break;
    case 2: f(IC<2>{}); break;
    case 3: f(IC<3>{}); break;
    case 4: f(IC<4>{}); break;
    case 5: f(IC<5>{}); break;
    case 6: f(IC<6>{}); break;
    case 7: f(IC<7>{}); break;
    default: f(IC<8>{}); break;
  }
}

// One contact candidate (end sphere s on body B), re-derived from the body
// pose each time it is needed instead of being stored: mj_collision
// (plane-sphere) + mj_instantiateContact + mj_makeImpedance for that sphere.
// Only called for spheres inside the margin on some lane of the wave (the mask
// of AntPublish): every lane gets a contact, with D = 0 (=> zero weight in every
// row) on the lanes where the sphere is outside.  Like the planar solver
// (mj_cheetah.cuh, WaveAny) nothing below branches per lane.
template <typename T>
struct AntContact {
  Vec3<T> cp;
  T an, ay, ax, D;
};
template <int B, typename T, typename G>
EPA_HD void AntMakeContact(const AntModel<T>& m, const G& p, const T* v, int s,
                           AntContact<T>& c) {
  Vec3<T> w = SphCenter<B>(p, s);
  T dist = w.z - m.sph_r[s];
  const bool touch = dist < m.margin;
  c.cp = {w.x, w.y, T(0.5) * dist};
  Vec3<T> vel = {T(0), T(0), T(0)};
  ForChainCols<B>(p, c.cp, [&](auto jc, Vec3<T> col) {
    vel = vel + col * v[decltype(jc)::value];
  });
  T rr = dist - m.margin;
  T imp = Impedance(m.imp_d0, m.imp_dmax, m.imp_width, rr);
  T diag = m.geom_body_invw[SphGeomBody(s)] * (T(1) + m.mu * m.mu);
  const T num = (T(1) - imp) * diag;  // R = max(mjMINVAL, num / imp), D_py = 1 / (2 mu^2 R)
  const T invR = num < T(1e-15) * imp ? T(1e15) : imp / num;
  c.D = touch ? invR * (T(1) / (T(2) * m.mu * m.mu)) : T(0);
  c.an = touch ? -m.con_B * vel.z - m.con_K * imp * rr : T(0);
  c.ay = touch ? m.con_B * m.mu * vel.y : T(0);
  c.ax = touch ? m.con_B * m.mu * vel.x : T(0);
}

// the four pyramidal rows of a contact in terms of (jx, jy, jz) = J a:
//   r1 = jz + mu jy, r2 = jz - mu jy, r3 = jz - mu jx, r4 = jz + mu jx
// with aref_1 = an - ay, aref_2 = an + ay, aref_3 = an + ax, aref_4 = an - ax
template <typename T>
EPA_HD void ContactJar(const AntModel<T>& m, Vec3<T> ja, T an, T ay, T ax, T* jar) {
  jar[0] = ja.z + m.mu * ja.y - (an - ay);
  jar[1] = ja.z - m.mu * ja.y - (an + ay);
  jar[2] = ja.z - m.mu * ja.x - (an + ax);
  jar[3] = ja.z + m.mu * ja.x - (an - ax);
}

template <bool kHess, typename T, typename G>
EPA_HD void AntRowsPass(const AntModel<T>& m, const G& p, unsigned sph, const AntRows<T>& r,
                        const T* v, const T* a, T* grad, T* H, unsigned long long* mask0,
                        unsigned long long* mask1) {
  unsigned long long m0 = 0, m1 = 0;
  static_for<0, kNU>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const T jar = r.lim_sgn[j] * a[6 + j] - r.lim_aref[j];
    const bool on = r.lim_sgn[j] != T(0) && jar < T(0);
    const T w = on ? r.lim_D[j] : T(0);
    grad[6 + j] += r.lim_sgn[j] * w * jar;
    if constexpr (kHess) H[Tri(6 + j, 6 + j)] += w;
    m0 |= (on ? 1ull : 0ull) << j;
  });
  EPA_ANT_NO_UNROLL
  for (unsigned rem = sph; rem != 0; rem &= rem - 1) {  // scalar loop over touching spheres
    const int s = __builtin_ctz(rem);
    EPA_LDS_FENCE();
    DispatchBody(SphBody(s), [&](auto bc) {
      constexpr int b = decltype(bc)::value;
      AntContact<T> c;
      AntMakeContact<b>(m, p, v, s, c);
      Vec3<T> ja = {T(0), T(0), T(0)};
      ForChainCols<b>(p, c.cp, [&](auto jc, Vec3<T> col) {
        ja = ja + col * a[decltype(jc)::value];
      });
      T jar[4];
      ContactJar(m, ja, c.an, c.ay, c.ax, jar);
      T w[4];
      static_for<0, 4>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const bool on = c.D > T(0) && jar[k] < T(0);
        w[k] = on ? c.D : T(0);
        const int bit = 8 + 4 * s + k;  // s is wave-uniform
        const unsigned long long one = on ? 1ull : 0ull;
        if (bit < 64) {
          m0 |= one << bit;
        } else {
          m1 |= one << (bit - 64);
        }
      });
      T wsum = w[0] + w[1] + w[2] + w[3];
      if (WaveAny(wsum > T(0))) {
        T gz = w[0] * jar[0] + w[1] * jar[1] + w[2] * jar[2] + w[3] * jar[3];
        T gy = m.mu * (w[0] * jar[0] - w[1] * jar[1]);
        T gx = m.mu * (w[3] * jar[3] - w[2] * jar[2]);
        T hzz = wsum;
        T hyy = m.mu * m.mu * (w[0] + w[1]), hxx = m.mu * m.mu * (w[2] + w[3]);
        T hzy = m.mu * (w[0] - w[1]), hzx = m.mu * (w[3] - w[2]);
        ForChainCols<b>(p, c.cp, [&](auto ic, Vec3<T> ci) {
          constexpr int i = decltype(ic)::value;
          grad[i] += ci.x * gx + ci.y * gy + ci.z * gz;
          if constexpr (kHess) {
            T ux = hxx * ci.x + hzx * ci.z;
            T uy = hyy * ci.y + hzy * ci.z;
            T uz = hzx * ci.x + hzy * ci.y + hzz * ci.z;
            ForChainCols<b>(p, c.cp, [&](auto kc2, Vec3<T> ck) {
              constexpr int k = decltype(kc2)::value;
              if constexpr (k >= i) H[Tri(i, k)] += ux * ck.x + uy * ck.y + uz * ck.z;
            });
          }
        });
      }
    });
  }
  *mask0 = m0;
  *mask1 = m1;
}

template <typename T, typename G>
EPA_HD void AntLineEval(const AntModel<T>& m, const G& p, unsigned sph, const AntRows<T>& r,
                        const T* v, const T* a, const T* s, T alpha, T* d1, T* d2) {
  static_for<0, kNU>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const T jar = r.lim_sgn[j] * a[6 + j] - r.lim_aref[j];
    const T jv = r.lim_sgn[j] * s[6 + j];
    const T x = jar + alpha * jv;
    const T w = (r.lim_sgn[j] != T(0) && x < T(0)) ? r.lim_D[j] : T(0);
    *d1 += w * x * jv;
    *d2 += w * jv * jv;
  });
  EPA_ANT_NO_UNROLL
  for (unsigned rem = sph; rem != 0; rem &= rem - 1) {  // scalar loop over touching spheres
    const int sidx = __builtin_ctz(rem);
    EPA_LDS_FENCE();
    DispatchBody(SphBody(sidx), [&](auto bc) {
      constexpr int b = decltype(bc)::value;
      AntContact<T> c;
      AntMakeContact<b>(m, p, v, sidx, c);
      Vec3<T> ja = {T(0), T(0), T(0)}, js = {T(0), T(0), T(0)};
      ForChainCols<b>(p, c.cp, [&](auto jc, Vec3<T> col) {
        ja = ja + col * a[decltype(jc)::value];
        js = js + col * s[decltype(jc)::value];
      });
      T jar[4];
      ContactJar(m, ja, c.an, c.ay, c.ax, jar);
      T jv[4] = {js.z + m.mu * js.y, js.z - m.mu * js.y, js.z - m.mu * js.x,
                 js.z + m.mu * js.x};
      static_for<0, 4>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const T x = jar[k] + alpha * jv[k];
        const T w = x < T(0) ? c.D : T(0);  // D == 0 on lanes without contact
        *d1 += w * x * jv[k];
        *d2 += w * jv[k] * jv[k];
      });
    });
  }
}

// y = M x with M read from the lane's LDS block
template <typename T, typename Lds>
EPA_HD void SymMulLds(Lds&& lds, const T* x, T* y) {
  static_for<0, kNV>([&](auto ic) { y[decltype(ic)::value] = T(0); });
  static_for<0, kNV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    static_for<0, j + 1>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (NZ(i, j)) {
        constexpr int slot = MSlot(i, j);
        const T mij = lds(slot);
        y[i] += mij * x[j];
        if constexpr (i != j) y[j] += mij * x[i];
      }
    });
  });
}

template <typename T, typename Lds>
EPA_HD int AntSolve(const AntModel<T>& m, Lds&& lds, unsigned sph,
                    const AntRows<T>& r, const T* v, const T* qfrc_smooth,
                    const SolverCfg<T>& cfg, T* qacc) {
  const AntGeoLds<T, typename std::remove_reference<Lds>::type> p{lds};
  T fs = T(0);
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    T x = qfrc_smooth[i] < T(0) ? -qfrc_smooth[i] : qfrc_smooth[i];
    fs = x > fs ? x : fs;
  });
  const T gstop = cfg.gtol * (T(1) + fs);
  const T gfloor = (sizeof(T) == 4 ? T(1e-4) : T(1e-9)) * (T(1) + fs);
  const T gstop2 = gstop * gstop, gfloor2 = gfloor * gfloor;
  T prev_gn2 = T(-1);
  unsigned long long pm0 = ~0ull, pm1 = ~0ull;
  T Ma[kNV];  // M qacc, kept current incrementally (Ma += alpha * M s)
  SymMulLds(lds, qacc, Ma);
  bool full_step = false;
  bool live = true;  // this lane is still iterating (finished lanes keep a frozen qacc)
  int iter = 0;
  for (int it = 0; it < cfg.max_iter; ++it) {
    T H[kTri], grad[kNV];
    EPA_LDS_FENCE();
    static_for<0, kNV>([&](auto jc) {  // H = M (structural non-zeros only)
      constexpr int j = decltype(jc)::value;
      static_for<0, j + 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (NZ(i, j)) {
          constexpr int slot = MSlot(i, j);
          H[Tri(i, j)] = lds(slot);
        }
      });
    });
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      grad[i] = Ma[i] - qfrc_smooth[i];
    });
    unsigned long long m0, m1;
    AntRowsPass<true>(m, p, sph, r, v, qacc, grad, H, &m0, &m1);
    T gn2 = T(0);
    static_for<0, kNV>([&](auto ic) { gn2 += grad[decltype(ic)::value] * grad[decltype(ic)::value]; });
    const bool stop = gn2 <= gstop2 || (full_step && m0 == pm0 && m1 == pm1) ||
                      (prev_gn2 >= T(0) && gn2 <= gfloor2 && gn2 >= T(0.0625) * prev_gn2);
    live = live && !stop;
    if (!WaveAny(live)) break;
    iter += live ? 1 : 0;
    prev_gn2 = gn2;
    pm0 = m0;
    pm1 = m1;
    T s[kNV];
    static_for<0, kNV>([&](auto ic) { s[decltype(ic)::value] = -grad[decltype(ic)::value]; });
    FactorUUt(H);
    SolveUUt(H, s);
    T Ms[kNV];
    EPA_LDS_FENCE();
    SymMulLds(lds, s, Ms);
    T g1 = T(0), g2 = T(0);
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      g1 += s[i] * (Ma[i] - qfrc_smooth[i]);
      g2 += s[i] * Ms[i];
    });
    T alpha = T(1), lo = T(0), hi = T(-1);
    full_step = false;
    const T ag1 = g1 < T(0) ? -g1 : g1;
    const T ls_tol = (sizeof(T) == 4 ? T(1e-4) : T(1e-10)) * ag1;
    bool searching = live;
    for (int ls = 0; ls < 24; ++ls) {
      T d1 = g1 + alpha * g2, d2 = g2;
      AntLineEval(m, p, sph, r, v, qacc, s, alpha, &d1, &d2);
      const T ad1 = d1 < T(0) ? -d1 : d1;
      const bool hit = ad1 <= ls_tol;
      full_step = full_step || (searching && hit && ls == 0);
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
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      qacc[i] += step * s[i];
      Ma[i] += step * Ms[i];
    });
  }
  return iter;
}

// mj_forward: qacc for state (q, v) under ctrl; `warm` is qacc_warmstart in/out.
// Returns Newton iterations.  q's quaternion is normalised in place.
// ---- fused front end ------------------------------------------------------------
// mj_kinematics + mj_comPos + mj_crb + mj_comVel + mj_rne + mj_passive +
// mj_fwdActuation of one forward pass, restructured around the tree: the torso
// first, then ONE LEG AT A TIME (aux + foot body, hip + ankle dof).  The phase-
// by-phase formulation above (AntKinematics / AntSmoothForces, kept for the host
// model compiler) holds every body's frame, inertia, cdof, velocity and force
// at once (~600 numbers) and spills; here a leg's quantities die before the next
// leg starts, its 15 entries of M and its geometry go straight to LDS, and only
// the torso accumulators (composite inertia, force) stay live (~100 numbers).
// Legs are visited last to first, the accumulation order of mj_crb / mj_rne.
template <typename T>
EPA_HD In10<T> AntCinert(const AntModel<T>& m, int b, const Mat3<T>& R, Vec3<T> d) {
  // body inertia rotated into the world frame, shifted to offset d = xipos - com
  const T* I = m.inertia[b];
  Mat3<T> Ib = {{I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}};
  Mat3<T> RI = Mul(R, Ib);
  const T* r = R.m;
  const T w0 = RI.m[0] * r[0] + RI.m[1] * r[1] + RI.m[2] * r[2];
  const T w1 = RI.m[3] * r[3] + RI.m[4] * r[4] + RI.m[5] * r[5];
  const T w2 = RI.m[6] * r[6] + RI.m[7] * r[7] + RI.m[8] * r[8];
  const T w3 = RI.m[0] * r[3] + RI.m[1] * r[4] + RI.m[2] * r[5];
  const T w4 = RI.m[0] * r[6] + RI.m[1] * r[7] + RI.m[2] * r[8];
  const T w5 = RI.m[3] * r[6] + RI.m[4] * r[7] + RI.m[5] * r[8];
  const T mass = m.mass[b], d2 = Dot(d, d);
  In10<T> c;
  c.v[0] = w0 + mass * (d2 - d.x * d.x);
  c.v[1] = w1 + mass * (d2 - d.y * d.y);
  c.v[2] = w2 + mass * (d2 - d.z * d.z);
  c.v[3] = w3 - mass * d.x * d.y;
  c.v[4] = w4 - mass * d.x * d.z;
  c.v[5] = w5 - mass * d.y * d.z;
  c.v[6] = mass * d.x;
  c.v[7] = mass * d.y;
  c.v[8] = mass * d.z;
  c.v[9] = mass;
  return c;
}

template <typename T>
struct AntLegFrames {  // world frames of one leg
  Vec3<T> posA, posF;
  Mat3<T> RA, RF;
};
template <int L, typename T>
EPA_HD AntLegFrames<T> AntLegKinematics(const AntModel<T>& m, const T* q, Vec3<T> pos0,
                                        const Mat3<T>& R0) {
  const T zaxis[3] = {T(0), T(0), T(1)};
  AntLegFrames<T> f;
  f.posA = pos0 + Mul(R0, Vec3<T>{m.aux_pos[L][0], m.aux_pos[L][1], m.aux_pos[L][2]});
  f.RA = Mul(R0, AxisAngle(zaxis, q[7 + 2 * L]));
  f.posF = f.posA + Mul(f.RA, Vec3<T>{m.foot_pos[L][0], m.foot_pos[L][1], m.foot_pos[L][2]});
  f.RF = Mul(f.RA, AxisAngle(m.ankle_axis[L], q[8 + 2 * L]));
  return f;
}

// Returns the wave-uniform mask of end spheres inside the contact margin;
// fills qfrc_smooth, the joint-limit rows and the lane's LDS block (M, geometry).
template <typename T, typename Lds>
EPA_HD unsigned AntFrontEnd(const AntModel<T>& m, T* q, const T* v, const T* ctrl, Lds&& lds,
                            AntRows<T>& rows, T* qfrc_smooth) {
  auto put = [&](int base, Vec3<T> x) {
    lds(base) = x.x;
    lds(base + 1) = x.y;
    lds(base + 2) = x.z;
  };
  auto com_of = [&](int b, Vec3<T> pos, const Mat3<T>& R) {
    return pos + Mul(R, Vec3<T>{m.com[b][0], m.com[b][1], m.com[b][2]});
  };
  unsigned mask = 0;
  auto probe = [&](int s, T z) {  // s compile-time after unrolling
    if (WaveAny(z - m.sph_r[s] < m.margin)) mask |= 1u << s;
  };
  NormalizeQuat(q + 3);  // mj_kinematics normalises the free-joint quaternion in qpos
  const Vec3<T> pos0 = {q[0], q[1], q[2]};
  const Mat3<T> R0 = QuatToMat(q[3], q[4], q[5], q[6]);
  // pass A: subtree COM of the robot (mj_comPos) needs every body once
  Vec3<T> com;
  {
    Vec3<T> s = com_of(0, pos0, R0) * m.mass[0];
    static_for<0, kNLeg>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      const AntLegFrames<T> f = AntLegKinematics<l>(m, q, pos0, R0);
      s = s + com_of(Aux(l), f.posA, f.RA) * m.mass[Aux(l)];
      s = s + com_of(Foot(l), f.posF, f.RF) * m.mass[Foot(l)];
    });
    com = s * (T(1) / m.total_mass);
  }
  put(kGeoCom, com);
  // torso: cdof of the free joint (translations are (0; e_k)), velocity, acceleration
  Sp6<T> rdof[3];  // rotational dofs 3..5
  static_for<0, 3>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    Vec3<T> ax = Col(R0, k);
    rdof[k] = {ax, Cross(ax, com - pos0)};
    put(kGeoRot + 3 * k, ax);
  });
  put(kGeoPos, pos0);
  probe(0, pos0.z);
  Sp6<T> cvel0 = {{T(0), T(0), T(0)}, {v[0], v[1], v[2]}};
  Sp6<T> cacc0 = {{T(0), T(0), T(0)}, {T(0), T(0), m.gravity}};
  {
    // cdof_dot of the three rotations all use the velocity before they are added
    Sp6<T> before = cvel0;
    static_for<0, 3>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      Axpy(cacc0, CrossMotion(before, rdof[k]), v[3 + k]);
      Axpy(cvel0, rdof[k], v[3 + k]);
    });
  }
  const In10<T> cinert0 = AntCinert(m, 0, R0, com_of(0, pos0, R0) - com);
  In10<T> crb0 = cinert0;  // composite inertia of the whole robot (accumulated)
  Sp6<T> cfrc0;            // torso force + every leg's (accumulated)
  {
    Sp6<T> f = MulInert(cinert0, cacc0);
    Sp6<T> g = CrossForce(cvel0, MulInert(cinert0, cvel0));
    cfrc0 = {f.a + g.a, f.l + g.l};
  }
  // dot of root dof k (0..5) with a spatial force
  auto root_dot = [&](auto kc, const Sp6<T>& f) -> T {
    constexpr int k = decltype(kc)::value;
    if constexpr (k == 0) return f.l.x;
    if constexpr (k == 1) return f.l.y;
    if constexpr (k == 2) return f.l.z;
    if constexpr (k >= 3) return Dot(rdof[k - 3], f);
  };
  const T kMinVal = T(1e-15);
  static_for_down<kNLeg, 0>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    constexpr int A = Aux(l), F = Foot(l), jh = Hip(l), ja = Ankle(l);
    const AntLegFrames<T> f = AntLegKinematics<l>(m, q, pos0, R0);
    // geometry block + contact candidates of this leg
    constexpr int s0 = 1 + 6 * l;
    const Vec3<T> tip = f.posF + Mul(f.RF, Vec3<T>{m.sph[s0 + 4][0], m.sph[s0 + 4][1], m.sph[s0 + 4][2]});
    const Vec3<T> hz = Col(R0, 2);  // hip axis: +z of the aux frame = torso z
    const Vec3<T> ha = Mul(f.RA, Vec3<T>{m.ankle_axis[l][0], m.ankle_axis[l][1], m.ankle_axis[l][2]});
    put(kGeoPos + 3 * A, f.posA);
    put(kGeoPos + 3 * F, f.posF);
    put(kGeoTip + 3 * l, tip);
    put(kGeoAnkle + 3 * l, ha);
    probe(s0 + 0, f.posA.z);
    probe(s0 + 1, pos0.z);
    probe(s0 + 2, f.posF.z);
    probe(s0 + 3, f.posA.z);
    probe(s0 + 4, tip.z);
    probe(s0 + 5, f.posF.z);
    // inertias and motion axes about the robot COM
    const In10<T> ciA = AntCinert(m, A, f.RA, com_of(A, f.posA, f.RA) - com);
    const In10<T> ciF = AntCinert(m, F, f.RF, com_of(F, f.posF, f.RF) - com);
    const Sp6<T> dh = {hz, Cross(hz, com - f.posA)};
    const Sp6<T> da = {ha, Cross(ha, com - f.posF)};
    // mj_crb: composite inertias foot, aux(+foot); rows of M owned by this leg
    In10<T> crbA = ciA;
    static_for<0, 10>([&](auto kc) { crbA.v[decltype(kc)::value] += ciF.v[decltype(kc)::value]; });
    static_for<0, 10>([&](auto kc) { crb0.v[decltype(kc)::value] += crbA.v[decltype(kc)::value]; });
    {
      const Sp6<T> buf = MulInert(ciF, da);  // column of the ankle dof
      static_for<0, 6>([&](auto kc) {
        constexpr int slot = MSlot(decltype(kc)::value, ja);
        lds(slot) = root_dot(kc, buf);
      });
      constexpr int s_ha = MSlot(jh, ja), s_aa = MSlot(ja, ja);
      lds(s_ha) = Dot(dh, buf);
      lds(s_aa) = Dot(da, buf) + m.arm[ja - 6];
    }
    {
      const Sp6<T> buf = MulInert(crbA, dh);  // column of the hip dof
      static_for<0, 6>([&](auto kc) {
        constexpr int slot = MSlot(decltype(kc)::value, jh);
        lds(slot) = root_dot(kc, buf);
      });
      constexpr int s_hh = MSlot(jh, jh);
      lds(s_hh) = Dot(dh, buf) + m.arm[jh - 6];
    }
    // mj_comVel / mj_rne (flg_acc = 0) down the leg and back
    Sp6<T> cvA = cvel0, caA = cacc0;
    Axpy(caA, CrossMotion(cvel0, dh), v[jh]);
    Axpy(cvA, dh, v[jh]);
    Sp6<T> cvF = cvA, caF = caA;
    Axpy(caF, CrossMotion(cvA, da), v[ja]);
    Axpy(cvF, da, v[ja]);
    Sp6<T> frcF, frcA;
    {
      Sp6<T> x = MulInert(ciF, caF);
      Sp6<T> g = CrossForce(cvF, MulInert(ciF, cvF));
      frcF = {x.a + g.a, x.l + g.l};
    }
    {
      Sp6<T> x = MulInert(ciA, caA);
      Sp6<T> g = CrossForce(cvA, MulInert(ciA, cvA));
      frcA = {x.a + g.a + frcF.a, x.l + g.l + frcF.l};
    }
    cfrc0.a = cfrc0.a + frcA.a;
    cfrc0.l = cfrc0.l + frcA.l;
    // hinge damper (stiffness 0) - bias; motors are added below
    qfrc_smooth[jh] = -m.damp[jh - 6] * v[jh] - Dot(dh, frcA);
    qfrc_smooth[ja] = -m.damp[ja - 6] * v[ja] - Dot(da, frcF);
  });
  // root block of M from the composite inertia of the whole robot; root bias
  static_for<0, 6>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    Sp6<T> di;
    if constexpr (i < 3) {
      di = {{T(0), T(0), T(0)}, {T(i == 0), T(i == 1), T(i == 2)}};
    } else {
      di = rdof[i - 3];
    }
    const Sp6<T> buf = MulInert(crb0, di);
    static_for<0, i + 1>([&](auto jc) {
      constexpr int slot = MSlot(decltype(jc)::value, i);
      lds(slot) = root_dot(jc, buf);
    });
    qfrc_smooth[i] = -root_dot(ic, cfrc0);
  });
  static_for<0, kNU>([&](auto uc) {  // motors: gear * clamp(ctrl)
    constexpr int u = decltype(uc)::value;
    qfrc_smooth[CtrlDof(u)] += m.gear * ctrl[u];
  });
  // mj_instantiateLimit + mj_makeImpedance for the 8 limited hinges
  static_for<0, kNU>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T qq = q[7 + j];
    T dlo = qq - m.lo[j], dhi = m.hi[j] - qq;
    const T sgn = dlo < T(0) ? T(1) : (dhi < T(0) ? T(-1) : T(0));
    const T dist = dlo < T(0) ? dlo : (dhi < T(0) ? dhi : T(0));
    T imp = Impedance(m.imp_d0, m.imp_dmax, m.imp_width, dist);
    const T num = (T(1) - imp) * m.dof_invw[j];  // R = max(mjMINVAL, num / imp)
    const T Dj = num < kMinVal * imp ? T(1) / kMinVal : imp / num;
    rows.lim_sgn[j] = sgn;
    rows.lim_D[j] = sgn != T(0) ? Dj : T(0);
    rows.lim_aref[j] = -m.con_B * (sgn * v[6 + j]) - m.con_K * imp * dist;
  });
  return WaveUniform(mask);
}

// mj_rnePostConstraint, cfrc_ext part, for the forward pass that just finished:
// sink(g, torque, force) is called once per touching end sphere with the spatial
// force [torque about the robot COM; force] (world frame) the floor applies to
// the MuJoCo body carrying that sphere's geom (g = SphGeomBody(s): 0 torso,
// 1+3l stub, 2+3l leg, 3+3l ankle body); the world body receives the opposite.
// Edge forces f_k = -D min(0, J_k a - aref_k) along (0, mu, 1), (0, -mu, 1),
// (-mu, 0, 1), (mu, 0, 1) (ContactJar), i.e. mju_decodePyramid in world axes.
template <typename T, typename Lds, typename Sink>
EPA_HD void AntContactWrench(const AntModel<T>& m, Lds&& lds, unsigned sph, const T* v,
                             const T* qacc, Sink&& sink) {
  const AntGeoLds<T, typename std::remove_reference<Lds>::type> p{lds};
  EPA_ANT_NO_UNROLL
  for (unsigned rem = sph; rem != 0; rem &= rem - 1) {
    const int s = __builtin_ctz(rem);
    EPA_LDS_FENCE();
    DispatchBody(SphBody(s), [&](auto bc) {
      constexpr int b = decltype(bc)::value;
      AntContact<T> c;
      AntMakeContact<b>(m, p, v, s, c);
      Vec3<T> ja = {T(0), T(0), T(0)};
      ForChainCols<b>(p, c.cp, [&](auto jc, Vec3<T> col) {
        ja = ja + col * qacc[decltype(jc)::value];
      });
      T jar[4], f[4];
      ContactJar(m, ja, c.an, c.ay, c.ax, jar);
      static_for<0, 4>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        f[k] = jar[k] < T(0) ? -c.D * jar[k] : T(0);  // D == 0 on lanes without contact
      });
      const Vec3<T> F = {m.mu * (f[3] - f[2]), m.mu * (f[0] - f[1]), f[0] + f[1] + f[2] + f[3]};
      const Vec3<T> com = {lds(kGeoCom), lds(kGeoCom + 1), lds(kGeoCom + 2)};
      sink(SphGeomBody(s), Cross(c.cp - com, F), F);
    });
  }
}

struct AntNoWrench {
  template <typename V>
  EPA_HD void operator()(int, V, V) const {}
};

// kWrench (compile time: the extra pass must not cost the v4 kernel registers):
// report the contact forces of this evaluation through `sink` when `wrench`.
template <bool kWrench = false, typename T, typename Lds, typename Sink = AntNoWrench>
EPA_HD int AntForward(const AntModel<T>& m, const SolverCfg<T>& cfg, T* q, const T* v,
                      const T* ctrl, T* warm, T* qacc, Lds&& lds, bool wrench = false,
                      Sink&& sink = Sink()) {
  T qfrc_smooth[kNV];
  AntRows<T> rows;
  EPA_LDS_FENCE();
  const unsigned sph = AntFrontEnd(m, q, v, ctrl, lds, rows, qfrc_smooth);
  EPA_LDS_FENCE();
  static_for<0, kNV>([&](auto ic) { qacc[decltype(ic)::value] = warm[decltype(ic)::value]; });
  int it = AntSolve(m, lds, sph, rows, v, qfrc_smooth, cfg, qacc);
  static_for<0, kNV>([&](auto ic) { warm[decltype(ic)::value] = qacc[decltype(ic)::value]; });
  if constexpr (kWrench) {
    if (wrench) AntContactWrench(m, lds, sph, v, qacc, sink);  // wave-uniform flag
  }
  return it;
}

// mj_integratePos for the Ant: q <- q (+) h * dq  (dq in velocity coordinates)
template <typename T>
EPA_HD void AntIntegratePos(T* q, const T* dq, T h) {
  q[0] += h * dq[0];
  q[1] += h * dq[1];
  q[2] += h * dq[2];
  T wx = dq[3], wy = dq[4], wz = dq[5];
  T nrm = Sqrt(wx * wx + wy * wy + wz * wz);
  T ang = nrm * h;
  {
    const bool turn = ang > T(0);  // zero angular velocity leaves the quaternion as is
    T s, c;
    SinCos(T(0.5) * ang, &s, &c);
    T k = s / (turn ? nrm : T(1));
    T bw = c, bx = wx * k, by = wy * k, bz = wz * k;
    T aw = q[3], ax = q[4], ay = q[5], az = q[6];
    T nq[4] = {aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
               aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw};
    NormalizeQuat(nq);
    q[3] = turn ? nq[0] : aw;
    q[4] = turn ? nq[1] : ax;
    q[5] = turn ? nq[2] : ay;
    q[6] = turn ? nq[3] : az;
  }
  static_for<7, kNQ>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    q[i] += h * dq[i - 1];
  });
}

// One mj_step with integrator RK4 (mj_RungeKutta(4)).  On return q, v are the
// new state and (lagx, lagy) the torso xpos of the LAST forward evaluation
// (stage 4), which is what data_->xpos holds afterwards (ant.h:169-173).
// `wrench`: also report the contact forces of the LAST forward evaluation (RK4
// stage 4) through `sink` -- the data mj_rnePostConstraint sees after mj_step.
template <bool kWrench = false, typename T, typename Lds, typename Sink = AntNoWrench>
EPA_HD int AntStep(const AntModel<T>& m, const SolverCfg<T>& cfg, T* q, T* v, T* warm,
                   const T* ctrl, T* lagx, T* lagy, Lds&& lds, bool wrench = false,
                   Sink&& sink = Sink()) {
  const T h = m.timestep;
  // q0, v0: state at the start; qs, vs: state of the current stage; dq, dv: running
  // B-weighted sums.  The previous stage's velocity / acceleration are vs / F
  // themselves, read before they are overwritten.
  T q0[kNQ], v0[kNV], qs[kNQ], vs[kNV];
  T F[kNV], dq[kNV], dv[kNV];
  int it = 0;
  static_for<0, kNQ>([&](auto ic) { q0[decltype(ic)::value] = q[decltype(ic)::value]; });
  static_for<0, kNV>([&](auto ic) { v0[decltype(ic)::value] = v[decltype(ic)::value]; });
  // stage 1 at (q0, v0)
  it += AntForward(m, cfg, q, v, ctrl, warm, F, lds);
  static_for<0, kNQ>([&](auto ic) { q0[decltype(ic)::value] = q[decltype(ic)::value]; });  // normalised quat
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    dq[i] = v0[i] * T(1.0 / 6.0);
    dv[i] = F[i] * T(1.0 / 6.0);
    vs[i] = v0[i];
  });
  // stages 2..4: X_i = X_0 + h * a_i * (Xv_{i-1}, F_{i-1}), a = 1/2, 1/2, 1
  for (int stage = 1; stage < 4; ++stage) {
    const T a = stage == 3 ? T(1) : T(0.5);
    const T bw = stage == 3 ? T(1.0 / 6.0) : T(1.0 / 3.0);
    T step_dq[kNV];
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      step_dq[i] = a * vs[i];  // vs, F: previous stage
      vs[i] = v0[i] + h * a * F[i];
    });
    static_for<0, kNQ>([&](auto ic) { qs[decltype(ic)::value] = q0[decltype(ic)::value]; });
    AntIntegratePos(qs, step_dq, h);
    it += AntForward<kWrench>(m, cfg, qs, vs, ctrl, warm, F, lds, wrench && stage == 3, sink);
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dq[i] += bw * vs[i];
      dv[i] += bw * F[i];
    });
    if (stage == 3) {
      *lagx = qs[0];
      *lagy = qs[1];
    }
  }
  static_for<0, kNQ>([&](auto ic) { q[decltype(ic)::value] = q0[decltype(ic)::value]; });
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v[i] = v0[i] + h * dv[i];
  });
  AntIntegratePos(q, dq, h);
  return it;
}

}  // namespace ant
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_ANT_CUH_
