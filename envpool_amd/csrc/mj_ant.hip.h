// K3b — Ant model algebra shared by the step kernel (mj_ant4.hip.h: one env per lane quad)
// and the host-side model compiler (mj_ant_model.h): topology constants, 3-D spatial
// algebra in MuJoCo's c-frame, the whole-tree kinematics / CRB pass and the
// arrow-structured 14 x 14 factorisation the compiler uses for dof_invweight0 /
// body_invweight0 at qpos0 (mj_setConst).
//
// Replaces the arithmetic MuJoCo 3.6.0's mj_step performs for the gym Ant model
// (third_party/mujoco_gym_xml_patches/ant_envpool.xml) each time the reference
// calls it (envpool/mujoco/gym/mujoco_env.h:137-148): SURVEY.md §8a M1-M9 with
// integrator="RK4" (4 forward evaluations per mj_step).
//
// Facts both users rely on:
//  * topology is compile time: torso (free) + 4 x (aux: hip hinge, foot: ankle
//    hinge); the four jointless "leg" bodies of the XML are welded into the
//    torso for the dynamics (identical physics) while their own
//    body_invweight0 is kept for the contact regulariser;
//  * spatial vectors use MuJoCo's c-frame (world orientation, about the robot
//    COM), so the mass matrix / RNE recursions need no frame transforms;
//  * M/H have an arrow structure (torso 6x6, four 2x2 leg blocks coupled only
//    through the torso); entries between different legs are never materialised
//    and the U U^T factorisation in tree order has no fill;
//  * contacts are the 25 end spheres (torso sphere + 2 per capsule); every
//    capsule end sits on a body origin except the four foot tips.
#ifndef ENVPOOL_AMD_CSRC_MJ_ANT_HIP_H_
#define ENVPOOL_AMD_CSRC_MJ_ANT_HIP_H_

#include "mj_cheetah.hip.h"  // static_for, IC, Sqrt, SinCos, Impedance (generic part)

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

// Compiler-level fence: LDS contents must not be carried in registers across it
// (otherwise the loads get hoisted out of the solver loops / forwarded from the
// stores and everything lands in VGPRs -> scratch again).
#if defined(__HIP_DEVICE_COMPILE__)
#define EPA_LDS_FENCE() asm volatile("" ::: "memory")
#else
#define EPA_LDS_FENCE() ((void)0)
#endif

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

}  // namespace ant
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_ANT_HIP_H_
