// K3 — `mj_step` of the planar gym robots (HalfCheetah, Walker2d, Hopper)
// restated as a tree-specialised, fully unrolled per-thread routine (one env
// per thread).  Tree: torso (x / z slides + y hinge) with two 3-link legs.
//
// What it replaces: the arithmetic MuJoCo 3.6.0's mj_step performs for these
// models each time the reference calls it
// (envpool/mujoco/gym/mujoco_env.h:137-148, `frame_skip x mj_step`), i.e.
// SURVEY.md §8a stages M1-M9: kinematics, comPos, crb(+factor), collision
// (plane-capsule; Hopper also capsule-capsule), makeConstraint (joint limits +
// pyramidal frictional contacts), comVel/passive/rne, actuation, Newton solve of
// the convex constraint objective, Euler with implicit joint damping
// (HalfCheetah) or RK4 (Walker2d, Hopper).
//
// MI355X-first design (not a translation of MuJoCo's generic engine):
//  * the models (third_party/mujoco_gym_xml_patches/{half_cheetah,walker2d,
//    walker2d_v5,hopper}_envpool.xml: 2 slides + y-hinges) move in the x-z
//    plane, so every spatial quantity is a 3-vector (w_y, v_x, v_z) and every
//    inertia 4 numbers; the y-tangent friction rows have identically zero
//    tangential Jacobian and fold into the normal row with weight 2D;
//  * all tree loops are unrolled at compile time (static_for) so the 9x9
//    inertia/Hessian, body poses and Jacobian columns live in VGPRs with static
//    indices; the leg/leg zero blocks of M and H are never materialised;
//  * M/H are factored as U U^T from the last dof upwards (tree order), which
//    has no fill between the two legs; the diagonal is kept inverted;
//  * per-contact constants (contact point, reference accelerations, D) are
//    staged through LDS, laid out [slot][lane] so a wave's accesses are
//    bank-conflict free; Jacobian rows are rebuilt from the contact point
//    instead of being stored; the solver passes run a scalar loop over the
//    wave-uniform set of touching end spheres;
//  * no lane-divergent control flow in the solver (see WaveAny);
//  * the model is a compile-time constant (gen_mj_consts.cpp), so per-model
//    features (contact margin, Hopper's body pairs) fold away elsewhere.
// The same source compiles for the host (EPA_HD) so tests can run it in fp64
// on the CPU against oracle/mjcpu.
#ifndef ENVPOOL_AMD_CSRC_MJ_CHEETAH_HIP_H_
#define ENVPOOL_AMD_CSRC_MJ_CHEETAH_HIP_H_

#include <cmath>
#include <type_traits>
#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EPA_HD __host__ __device__ __forceinline__
#else
#define EPA_HD inline
#endif

namespace epa {
namespace mj {

// ---- compile-time loops -----------------------------------------------------
template <int I>
using IC = std::integral_constant<int, I>;

template <int B, int E, typename F>
EPA_HD void static_for(F&& f) {
  if constexpr (B < E) {
    f(IC<B>{});
    static_for<B + 1, E>(static_cast<F&&>(f));
  }
}
// descending: B-1, B-2, ..., E
template <int B, int E, typename F>
EPA_HD void static_for_down(F&& f) {
  if constexpr (B > E) {
    f(IC<B - 1>{});
    static_for_down<B - 1, E>(static_cast<F&&>(f));
  }
}

template <typename T>
EPA_HD T Sqrt(T x) {
  return sqrt(x);
}
// 1 / sqrt(x), x > 0.  Device fp64: hardware seed (v_rsq_f64) + two Newton
// steps (a couple of ulp) instead of an IEEE sqrt followed by an IEEE divide,
// which together cost ~25 dependent instructions per pivot.
template <typename T>
EPA_HD T Rsqrt(T x) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (std::is_same<T, double>::value) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
  } else {
    return T(1) / sqrt(x);
  }
#else
  return T(1) / std::sqrt(x);
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
// fp64 sin and cos together in ~50 VALU instructions (the device library's sincos is ~200 and
// carries a large-argument branch; seven of them per forward pass are 10 % of the planar
// kernels' code, which sits right at the 64 KB instruction cache): Cody-Waite reduction by
// pi/2 in three pieces -- 1 ulp up to |x| ~ 1e6 rad, degrading smoothly beyond; joint angles
// and the torso pitch stay orders of magnitude below that, NaN / inf give NaN like libm --
// and the classic fdlibm kernel polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void FastSinCos(double x, double* s, double* c) {
  const double n = rint(x * 6.36619772367581382433e-01);  // x * 2/pi
  double r = fma(-n, 1.57079632673412561417e+00, x);    // pi/2, first 33 bits
  r = fma(-n, 6.07710050630396597660e-11, r);           // next 33 bits
  r = fma(-n, 2.02226624879595063154e-21, r);           // tail
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10,
                                                   -2.50507602534068634195e-08),
                                             2.75573137070700676789e-06),
                                       -1.98412698298579493134e-04),
                                 8.33333333332248946124e-03),
                        -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11,
                                                   2.08757232129817482790e-09),
                                             -2.75573143513906633035e-07),
                                       2.48015872894767294178e-05),
                                 -1.38888888888741095749e-03),
                        4.16666666666666019037e-02);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * z * pc);
  const int q = (int)n & 3;
  const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}
#endif
// EPA_SINCOS_MODE (per translation unit, chosen by measurement -- these one-wave-per-SIMD
// kernels react to code size and to how far the scheduler may hoist straight-line code):
//   0 device library, 1 FastSinCos (default), 2 FastSinCos behind a wave-uniform range check
//   with the library as the out-of-range path (the branch also stops hoisting across it).
#ifndef EPA_SINCOS_MODE
#define EPA_SINCOS_MODE 1
#endif
EPA_HD bool WaveAny(bool x);
template <typename T>
EPA_HD void SinCos(T x, T* s, T* c) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (std::is_same<T, float>::value) {
    sincosf(x, s, c);
  } else {
#if EPA_SINCOS_MODE == 0
    sincos(x, s, c);
#elif EPA_SINCOS_MODE == 2
    if (WaveAny(!(fabs(x) < 1.0e5))) {  // also catches NaN / inf
      sincos(x, s, c);
    } else {
      FastSinCos(x, s, c);
    }
#else
    FastSinCos(x, s, c);
#endif
  }
#else
  *s = std::sin(x);
  *c = std::cos(x);
#endif
}

// True if `x` holds for any lane of the wavefront (host: the single "lane").
// The solver below has no lane-divergent control flow at all: every branch is
// taken on WaveAny(...) (a scalar branch on a ballot) and per-lane differences
// are applied as selects / zero weights.  Lanes whose Newton iteration or line
// search has finished keep executing with frozen iterates until the slowest
// lane of the wave is done -- which is what the hardware does with a divergent
// loop anyway -- and the exec mask never changes inside the hot loops.
EPA_HD bool WaveAny(bool x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(x) != 0;
#elif defined(EPA_HOST_WAVE_ANY)
  return EPA_HOST_WAVE_ANY(x);  // tests/cpu_harness: emulate a wave whose other lanes are still busy
#else
  return x;
#endif
}
// Marks an integer that is equal on every lane by construction (it was built
// from WaveAny results) so that it lives in an SGPR and loops over it are scalar.
EPA_HD unsigned WaveUniform(unsigned x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readfirstlane(x);
#else
  return x;
#endif
}

// ---- model --------------------------------------------------------------------
constexpr int kNB = 7;    // torso bthigh bshin bfoot fthigh fshin ffoot
constexpr int kNV = 9;    // rootx rootz rooty + 6 leg hinges
constexpr int kNU = 6;
constexpr int kNEnd = 16; // 8 capsules x 2 end spheres
constexpr int kSlotsPerEnd = 5;
constexpr int kLdsSlots = kNEnd * kSlotsPerEnd;

EPA_HD constexpr int Parent(int b) {
  return b == 0 ? -1 : (b == 4 ? 0 : b - 1);
}
EPA_HD constexpr int EndBody(int e) {
  // geoms in XML order: torso, head (body 0), then one capsule per leg body
  return e < 4 ? 0 : (e - 4) / 2 + 1;
}
// dof j >= 3 is the hinge of body j - 2; dofs 0..2 belong to the torso.
EPA_HD constexpr int DofBody(int j) { return j < 3 ? 0 : j - 2; }
// is dof j on the kinematic chain from the world to body b?
EPA_HD constexpr bool InChain(int j, int b) {
  if (j < 3) return true;
  int jb = j - 2;
  for (int x = b; x > 0; x = Parent(x)) {
    if (x == jb) return true;
  }
  return false;
}
// structurally non-zero entry of M / H (i <= j): the two legs never couple.
EPA_HD constexpr bool NZ(int i, int j) {
  return !((i >= 3 && i <= 5 && j >= 6) || (j >= 3 && j <= 5 && i >= 6));
}
EPA_HD constexpr int TriIdx(int i, int j) {  // i <= j, packed upper triangle
  return j * (j + 1) / 2 + i;
}
constexpr int kTri = kNV * (kNV + 1) / 2;

template <typename T>
struct CheetahModel {
  T lx[kNB], lz[kNB];    // body_pos in the parent frame (torso: world)
  T mass[kNB], iyy[kNB]; // mass, inertia about y through the body COM
  T cx[kNB], cz[kNB];    // body COM in the body frame
  T ex[kNEnd], ez[kNEnd];  // capsule end-sphere centres in the body frame
  T er[kNEnd];           // their radii (unused slots: -1e30, can never touch)
  T stiff[kNU], damp[kNU], arm[kNU], lo[kNU], hi[kNU], gear[kNU];
  T dof_invw[kNU];       // dof_invweight0 of the hinges
  T body_invw[kNB];      // body_invweight0 (translational)
  T total_mass;
  T bmu[kNB];            // sliding friction of the floor/geom pair, per body
  T con_K, con_B;        // contact reference: aref = -B vel - K imp (pos-margin)
  T con_d0, con_dmax, con_width;
  T lim_K, lim_B, lim_d0, lim_dmax, lim_width;
  T timestep, gravity;   // gravity = 9.81 (magnitude along -z)
  T con_margin;          // contact margin of every geom pair (Hopper: 0.001, else 0)
  int n_pairs;           // 3: the Hopper's body-body capsule pairs (kPairBody*), else 0
};

// Body-body collision candidates of the single-leg Hopper model (its geoms have
// contype = conaffinity = 1; parent-child pairs are filtered by MuJoCo): torso-leg,
// torso-foot, thigh-foot.  condim 1 => one frictionless row per contact.
constexpr int kNPair = 3;
EPA_HD constexpr int PairBody1(int k) { return k == 2 ? 1 : 0; }
EPA_HD constexpr int PairBody2(int k) { return k == 0 ? 2 : 3; }
// first end sphere of the (only) capsule of body b in the one-capsule-per-body layout
EPA_HD constexpr int BodyEnd0(int b) { return b == 0 ? 0 : 2 * b + 2; }
constexpr int kPairSlot0 = 50;     // LDS slots of the unused second-leg ends 10..15
constexpr int kSlotsPerPair = 6;   // nx nz px pz aref D

// ---- small planar spatial algebra ---------------------------------------------
template <typename T>
struct V3 {  // motion (w, vx, vz) or force (tau, fx, fz)
  T w, x, z;
};
template <typename T>
struct In4 {  // planar spatial inertia about the reference point
  T I, mdx, mdz, m;
};
template <typename T>
EPA_HD V3<T> MulInert(const In4<T>& i, const V3<T>& v) {
  return {i.I * v.w + i.mdz * v.x - i.mdx * v.z, i.m * v.x + i.mdz * v.w,
          i.m * v.z - i.mdx * v.w};
}
template <typename T>
EPA_HD T Dot(const V3<T>& a, const V3<T>& b) {
  return a.w * b.w + a.x * b.x + a.z * b.z;
}
template <typename T>
EPA_HD V3<T> CrossMotion(const V3<T>& vel, const V3<T>& v) {
  return {T(0), vel.w * v.z - vel.z * v.w, -vel.w * v.x + vel.x * v.w};
}
template <typename T>
EPA_HD V3<T> CrossForce(const V3<T>& vel, const V3<T>& f) {
  return {vel.z * f.x - vel.x * f.z, vel.w * f.z, -vel.w * f.x};
}

// impedance d(r) for power 2, midpoint 0.5 (getimpedance in MuJoCo)
template <typename T>
EPA_HD T Impedance(T d0, T dmax, T width, T r) {
  T x = (r < T(0) ? -r : r) * (T(1) / width);  // width is a model constant
  T y = x <= T(0.5) ? T(2) * x * x : T(1) - T(2) * (T(1) - x) * (T(1) - x);
  return x >= T(1) ? dmax : d0 + y * (dmax - d0);
}

// (Kept on: with fp32 factorisations the Newton loop stalls at its iteration cap
// on a few percent of the envs - 10x slower kernel, failing parity.)
// Wider type for the two ill-conditioned 9x9 solves per Newton iteration when
// the kernel runs in fp32 (cond(H) ~ 1e4: light feet vs 14 kg trunk): the
// factorisation and substitutions are ~5% of the flops, doing them in fp64
// recovers ~2 digits on the leg accelerations.
#ifndef EPA_MJ_SOLVE_WIDE
#define EPA_MJ_SOLVE_WIDE 1
#endif
template <typename T>
struct Wide {
#if EPA_MJ_SOLVE_WIDE
  using type = double;
#else
  using type = T;
#endif
};

// Upper "tree order" Cholesky A = U U^T on a packed upper triangle with the
// leg/leg zero blocks skipped.  In place; the diagonal is returned INVERTED
// (1 / U_jj), which is all SolveUUt needs.
template <typename T>
EPA_HD void FactorUUt(T* A) {
  static_for_down<kNV, 0>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T s = A[TriIdx(j, j)];
    static_for<j + 1, kNV>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (NZ(j, k)) s -= A[TriIdx(j, k)] * A[TriIdx(j, k)];
    });
    T inv = Rsqrt(s);
    A[TriIdx(j, j)] = inv;  // the diagonal holds 1 / U_jj
    static_for<0, j>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (NZ(i, j)) {
        T t = A[TriIdx(i, j)];
        static_for<j + 1, kNV>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (NZ(i, k) && NZ(j, k)) {
            t -= A[TriIdx(i, k)] * A[TriIdx(j, k)];
          }
        });
        A[TriIdx(i, j)] = t * inv;
      }
    });
  });
}
// solve U U^T x = b in place
template <typename T>
EPA_HD void SolveUUt(const T* U, T* x) {
  static_for_down<kNV, 0>([&](auto jc) {  // U y = b
    constexpr int j = decltype(jc)::value;
    T s = x[j];
    static_for<j + 1, kNV>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (NZ(j, k)) s -= U[TriIdx(j, k)] * x[k];
    });
    x[j] = s * U[TriIdx(j, j)];
  });
  static_for<0, kNV>([&](auto jc) {  // U^T x = y
    constexpr int j = decltype(jc)::value;
    T s = x[j];
    static_for<0, j>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (NZ(i, j)) s -= U[TriIdx(i, j)] * x[i];
    });
    x[j] = s * U[TriIdx(j, j)];
  });
}
// y = A x for a packed symmetric matrix with the structural zeros
template <typename T>
EPA_HD void SymMul(const T* A, const T* x, T* y) {
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    T s = T(0);
    static_for<0, kNV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (NZ(i, j)) {
        s += A[i <= j ? TriIdx(i, j) : TriIdx(j, i)] * x[j];
      }
    });
    y[i] = s;
  });
}

// Position-dependent quantities of one forward pass.
template <typename T>
struct CheetahPos {
  T sn[kNB], cs[kNB], px[kNB], pz[kNB];  // body frames (anchors = body origins)
  T comx, comz;                          // subtree COM of the robot
  In4<T> cinert[kNB];
  V3<T> cdof[kNV];
  T M[kTri];
};

template <typename T>
EPA_HD void CheetahKinematics(const CheetahModel<T>& m, const T* q,
                              CheetahPos<T>& p) {
  // mj_kinematics: q[0] may be a local (re-centred) x; dynamics are invariant.
  static_for<0, kNB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr int par = Parent(b);
    T sj, cj;
    SinCos(q[b + 2], &sj, &cj);
    if constexpr (b == 0) {
      p.px[0] = m.lx[0] + q[0];
      p.pz[0] = m.lz[0] + q[1];
      p.sn[0] = sj;
      p.cs[0] = cj;
    } else {
      p.px[b] = p.px[par] + p.cs[par] * m.lx[b] + p.sn[par] * m.lz[b];
      p.pz[b] = p.pz[par] - p.sn[par] * m.lx[b] + p.cs[par] * m.lz[b];
      p.sn[b] = p.sn[par] * cj + p.cs[par] * sj;
      p.cs[b] = p.cs[par] * cj - p.sn[par] * sj;
    }
  });
  // mj_comPos
  T xi[kNB], zi[kNB];
  T sx = T(0), sz = T(0);
  static_for<0, kNB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    xi[b] = p.px[b] + p.cs[b] * m.cx[b] + p.sn[b] * m.cz[b];
    zi[b] = p.pz[b] - p.sn[b] * m.cx[b] + p.cs[b] * m.cz[b];
    sx += m.mass[b] * xi[b];
    sz += m.mass[b] * zi[b];
  });
  p.comx = sx / m.total_mass;
  p.comz = sz / m.total_mass;
  static_for<0, kNB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    T dx = xi[b] - p.comx, dz = zi[b] - p.comz;
    p.cinert[b] = {m.iyy[b] + m.mass[b] * (dx * dx + dz * dz), m.mass[b] * dx,
                   m.mass[b] * dz, m.mass[b]};
  });
  p.cdof[0] = {T(0), T(1), T(0)};
  p.cdof[1] = {T(0), T(0), T(1)};
  static_for<2, kNV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int b = DofBody(j);
    // hinge about +y at the body origin: (1, oz, -ox), o = com - anchor
    p.cdof[j] = {T(1), p.comz - p.pz[b], -(p.comx - p.px[b])};
  });
  // mj_crb
  In4<T> crb[kNB];
  static_for<0, kNB>([&](auto bc) { crb[decltype(bc)::value] = p.cinert[decltype(bc)::value]; });
  static_for_down<kNB, 1>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr int par = Parent(b);
    crb[par].I += crb[b].I;
    crb[par].mdx += crb[b].mdx;
    crb[par].mdz += crb[b].mdz;
    crb[par].m += crb[b].m;
  });
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    V3<T> buf = MulInert(crb[DofBody(i)], p.cdof[i]);
    static_for<0, i + 1>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (InChain(j, DofBody(i))) {
        p.M[TriIdx(j, i)] = Dot(p.cdof[j], buf);
      } else if constexpr (NZ(j, i)) {
        p.M[TriIdx(j, i)] = T(0);
      }
    });
    if constexpr (i >= 3) p.M[TriIdx(i, i)] += m.arm[i - 3];
  });
}

// qfrc_smooth = passive - bias + actuator (mj_fwdVelocity, mj_fwdActuation)
template <typename T>
EPA_HD void CheetahSmoothForces(const CheetahModel<T>& m,
                                const CheetahPos<T>& p, const T* q, const T* v,
                                const T* ctrl, T* qfrc_smooth) {
  // mj_comVel
  V3<T> cvel[kNB], cdd[kNV];
  {
    V3<T> cv = {T(0), T(0), T(0)};
    static_for<0, 3>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      cdd[j] = CrossMotion(cv, p.cdof[j]);
      cv.w += p.cdof[j].w * v[j];
      cv.x += p.cdof[j].x * v[j];
      cv.z += p.cdof[j].z * v[j];
    });
    cvel[0] = cv;
  }
  static_for<1, kNB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr int j = b + 2;
    V3<T> cv = cvel[Parent(b)];
    cdd[j] = CrossMotion(cv, p.cdof[j]);
    cv.w += p.cdof[j].w * v[j];
    cv.x += p.cdof[j].x * v[j];
    cv.z += p.cdof[j].z * v[j];
    cvel[b] = cv;
  });
  // mj_rne (no acceleration term); world cacc = -gravity
  V3<T> cacc[kNB], cfrc[kNB];
  static_for<0, kNB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    V3<T> a;
    if constexpr (b == 0) {
      a = {T(0), T(0), m.gravity};
      static_for<0, 3>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        a.w += cdd[j].w * v[j];
        a.x += cdd[j].x * v[j];
        a.z += cdd[j].z * v[j];
      });
    } else {
      constexpr int j = b + 2;
      a = cacc[Parent(b)];
      a.w += cdd[j].w * v[j];
      a.x += cdd[j].x * v[j];
      a.z += cdd[j].z * v[j];
    }
    cacc[b] = a;
    V3<T> f = MulInert(p.cinert[b], a);
    V3<T> g = CrossForce(cvel[b], MulInert(p.cinert[b], cvel[b]));
    cfrc[b] = {f.w + g.w, f.x + g.x, f.z + g.z};
  });
  static_for_down<kNB, 1>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr int par = Parent(b);
    cfrc[par].w += cfrc[b].w;
    cfrc[par].x += cfrc[b].x;
    cfrc[par].z += cfrc[b].z;
  });
  static_for<0, kNV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T bias = Dot(p.cdof[j], cfrc[DofBody(j)]);
    if constexpr (j < 3) {
      qfrc_smooth[j] = -bias;
    } else {
      // mj_passive (spring about qpos_spring = 0, damper) + motor
      qfrc_smooth[j] = -m.stiff[j - 3] * q[j] - m.damp[j - 3] * v[j] - bias +
                       m.gear[j - 3] * ctrl[j - 3];
    }
  });
}

// Jacobian columns of a contact point (cpx, cpz) on body B:
//   Jn[j] = d(z velocity)/d qdot_j, Jx[j] = d(x velocity)/d qdot_j.
// f(j, jn, jx) is called for every chain dof.
template <int B, typename T, typename F>
EPA_HD void ForChainCols(const CheetahPos<T>& p, T cpx, T cpz, F&& f) {
  f(IC<0>{}, T(0), T(1));
  f(IC<1>{}, T(1), T(0));
  static_for<2, kNV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (InChain(j, B)) {
      constexpr int b = DofBody(j);
      f(jc, -(cpx - p.px[b]), cpz - p.pz[b]);
    }
  });
}

// Runtime loop over the end spheres with a wave-uniform switch on the body:
// keeps the per-contact constants in LDS (dynamic slot index => real ds_read,
// no store-to-load forwarding into long-lived VGPRs) and emits the Jacobian
// code once per body instead of once per end sphere.
// NOTE (ROCm 7.2 / gfx950): the fp64 instantiation (512 VGPR+AGPR plus spills)
// was miscompiled -- run-to-run different results, Newton never terminating --
// whenever this uniform switch sat inside lane-divergent control flow, and again
// (with a different build) with the divergence inside the cases.  Hence the
// rule stated at WaveAny(): no lane-divergent branch anywhere in the solver.
#if defined(__clang__)
#define EPA_NO_UNROLL _Pragma("clang loop unroll(disable)")
#else
#define EPA_NO_UNROLL
#endif
template <typename F>
EPA_HD void DispatchBody(int b, F&& f) {
  switch (b) {
    case 0: f(IC<0>{}); break;
    case 1: f(IC<1>{}); break;
    case 2: f(IC<2>{}); break;
    case 3: f(IC<3>{}); break;
    case 4: f(IC<4>{}); break;
    case 5: f(IC<5>{}); break;
    default: f(IC<6>{}); break;
  }
}

// Limit rows kept in registers; contact rows staged through `lds`
// (lds(slot) -> T&, slot = 5*e + {0:cpx 1:cpz 2:aref_n 3:B*mu*vx 4:D}).
template <typename T>
struct LimitRows {
  T sgn[kNU], aref[kNU], D[kNU];
};

// Returns the set of end spheres (bit e) that touch the plane on any lane of
// the wave; the solver passes only visit those.
template <typename T, typename Lds>
EPA_HD unsigned CheetahMakeConstraint(const CheetahModel<T>& m,
                                  const CheetahPos<T>& p, const T* q,
                                  const T* v, LimitRows<T>& lim, Lds&& lds) {
  const T kMinVal = T(1e-15);
  // mj_instantiateLimit + mj_makeImpedance for the 6 limited hinges
  static_for<0, kNU>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T qq = q[j + 3];
    T dlo = qq - m.lo[j], dhi = m.hi[j] - qq;
    const T sgn = dlo < T(0) ? T(1) : (dhi < T(0) ? T(-1) : T(0));
    const T dist = dlo < T(0) ? dlo : (dhi < T(0) ? dhi : T(0));
    T imp = Impedance(m.lim_d0, m.lim_dmax, m.lim_width, dist);
    // R = max(mjMINVAL, (1 - imp) / imp * diagApprox), D = 1 / R
    const T num = (T(1) - imp) * m.dof_invw[j];
    const T Dj = num < kMinVal * imp ? T(1) / kMinVal : imp / num;
    lim.sgn[j] = sgn;
    lim.D[j] = sgn != T(0) ? Dj : T(0);
    lim.aref[j] = -m.lim_B * (sgn * v[j + 3]) - m.lim_K * imp * dist;
  });
  // mj_collision (plane z=0 vs capsule end spheres) + mj_instantiateContact
  unsigned ends = 0;
  static_for<0, kNEnd>([&](auto ec) {
    constexpr int e = decltype(ec)::value;
    constexpr int b = EndBody(e);
    if (m.er[e] < T(-1e29)) return;  // unused end slot of this model (folds at compile time)
    T wx = p.px[b] + p.cs[b] * m.ex[e] + p.sn[b] * m.ez[e];
    T wz = p.pz[b] - p.sn[b] * m.ex[e] + p.cs[b] * m.ez[e];
    T dist = wz - m.er[e];
    T D = T(0), cpx = wx, cpz = T(0.5) * dist, an = T(0), ax = T(0);
    const bool touch = dist < m.con_margin;
    if (WaveAny(touch)) {
      ends |= 1u << e;
      T vn = T(0), vx = T(0);
      ForChainCols<b>(p, cpx, cpz, [&](auto jc, T jn, T jx) {
        constexpr int j = decltype(jc)::value;
        vn += jn * v[j];
        vx += jx * v[j];
      });
      const T r = dist - m.con_margin;
      T imp = Impedance(m.con_d0, m.con_dmax, m.con_width, r);
      // diagApprox (pyramidal) = tran (1 + mu^2); R_py = 2 mu^2 R
      T diag = m.body_invw[b] * (T(1) + m.bmu[b] * m.bmu[b]);
      const T num = (T(1) - imp) * diag;  // R = max(mjMINVAL, num / imp)
      const T invR = num < kMinVal * imp ? T(1) / kMinVal : imp / num;
      D = touch ? invR * (T(1) / (T(2) * m.bmu[b] * m.bmu[b])) : T(0);
      an = touch ? -m.con_B * vn - m.con_K * imp * r : T(0);
      ax = touch ? m.con_B * m.bmu[b] * vx : T(0);
    }
    lds(e * kSlotsPerEnd + 0) = cpx;
    lds(e * kSlotsPerEnd + 1) = cpz;
    lds(e * kSlotsPerEnd + 2) = an;
    lds(e * kSlotsPerEnd + 3) = ax;
    lds(e * kSlotsPerEnd + 4) = D;
  });
  // body-body capsule pairs (mjc_CapsuleCapsule: closest points of the two axis
  // segments, then sphere-sphere), Hopper model only
  if (m.n_pairs > 0) {
    static_for<0, kNPair>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int b1 = PairBody1(k), b2 = PairBody2(k);
      constexpr int e1 = BodyEnd0(b1), e2 = BodyEnd0(b2);
      auto end_pos = [&](auto bc, int e, T* x, T* z) {
        constexpr int b = decltype(bc)::value;
        *x = p.px[b] + p.cs[b] * m.ex[e] + p.sn[b] * m.ez[e];
        *z = p.pz[b] - p.sn[b] * m.ex[e] + p.cs[b] * m.ez[e];
      };
      T ax1, az1, bx1, bz1, ax2, az2, bx2, bz2;
      end_pos(IC<b1>{}, e1, &ax1, &az1);
      end_pos(IC<b1>{}, e1 + 1, &bx1, &bz1);
      end_pos(IC<b2>{}, e2, &ax2, &az2);
      end_pos(IC<b2>{}, e2 + 1, &bx2, &bz2);
      // centres and half axes (towards the first end = +z of the geom frame)
      const T c1x = T(0.5) * (ax1 + bx1), c1z = T(0.5) * (az1 + bz1);
      const T c2x = T(0.5) * (ax2 + bx2), c2z = T(0.5) * (az2 + bz2);
      const T h1x = T(0.5) * (ax1 - bx1), h1z = T(0.5) * (az1 - bz1);
      const T h2x = T(0.5) * (ax2 - bx2), h2z = T(0.5) * (az2 - bz2);
      const T dfx = c1x - c2x, dfz = c1z - c2z;
      const T ma = h1x * h1x + h1z * h1z, mb = -(h1x * h2x + h1z * h2z), mc = h2x * h2x + h2z * h2z;
      const T u = -(h1x * dfx + h1z * dfz), w = h2x * dfx + h2z * dfz;
      const T det = ma * mc - mb * mb;
      auto clamp1 = [](T x) { return x > T(1) ? T(1) : (x < T(-1) ? T(-1) : x); };
      // general position (|det| >= mjMINVAL); the exactly parallel case falls back to
      // the midpoint of the overlap like oracle/mjcpu (measure zero)
      const bool par = (det < T(0) ? -det : det) < kMinVal;
      T x1 = (mc * u - mb * w) / (par ? T(1) : det);
      T x2 = (ma * w - mb * u) / (par ? T(1) : det);
      {
        const bool hi1 = x1 > T(1), lo1 = x1 < T(-1);
        x2 = hi1 ? (w - mb) / mc : (lo1 ? (w + mb) / mc : x2);
        x1 = clamp1(x1);
        const bool hi2 = x2 > T(1), lo2 = x2 < T(-1);
        const T x1b = clamp1(hi2 ? (u - mb) / ma : (u + mb) / ma);
        x1 = (hi2 || lo2) ? x1b : x1;
        x2 = clamp1(x2);
      }
      {
        const T amb = mb < T(0) ? -mb : mb;
        T lo = (u - amb) / ma, hi = (u + amb) / ma;
        lo = lo < T(-1) ? T(-1) : lo;
        hi = hi > T(1) ? T(1) : hi;
        const T xp1 = lo <= hi ? T(0.5) * (lo + hi) : (lo > T(1) ? T(1) : T(-1));
        const T xp2 = clamp1((w - mb * xp1) / mc);
        x1 = par ? xp1 : x1;
        x2 = par ? xp2 : x2;
      }
      const T p1x = c1x + h1x * x1, p1z = c1z + h1z * x1;
      const T p2x = c2x + h2x * x2, p2z = c2z + h2z * x2;
      const T ddx = p2x - p1x, ddz = p2z - p1z;
      const T cd = Sqrt(ddx * ddx + ddz * ddz);
      const T r1 = m.er[e1], r2 = m.er[e2];
      const T dist = cd - r1 - r2;
      const bool touch = dist < m.con_margin;
      T nx = T(1), nz = T(0), cx = T(0), cz = T(0), aref = T(0), D = T(0);
      if (WaveAny(touch)) {
        ends |= 1u << (16 + k);
        const T inv = T(1) / (cd < kMinVal ? T(1) : cd);
        nx = cd < kMinVal ? T(1) : ddx * inv;
        nz = cd < kMinVal ? T(0) : ddz * inv;
        cx = p1x + nx * (r1 + T(0.5) * dist);
        cz = p1z + nz * (r1 + T(0.5) * dist);
        // relative normal velocity: only the hinges between the two bodies contribute
        // (the dofs shared by both chains move both bodies alike)
        T vel = T(0);
        static_for<3, kNV>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (InChain(j, b2) && !InChain(j, b1)) {
            constexpr int jb = DofBody(j);
            vel += (nx * (cz - p.pz[jb]) - nz * (cx - p.px[jb])) * v[j];
          }
        });
        const T r = dist - m.con_margin;
        const T imp = Impedance(m.con_d0, m.con_dmax, m.con_width, r);
        const T num = (T(1) - imp) * (m.body_invw[b1] + m.body_invw[b2]);  // condim 1: tran1 + tran2
        const T invR = num < kMinVal * imp ? T(1) / kMinVal : imp / num;
        D = touch ? invR : T(0);
        aref = touch ? -m.con_B * vel - m.con_K * imp * r : T(0);
      }
      lds(kPairSlot0 + k * kSlotsPerPair + 0) = nx;
      lds(kPairSlot0 + k * kSlotsPerPair + 1) = nz;
      lds(kPairSlot0 + k * kSlotsPerPair + 2) = cx;
      lds(kPairSlot0 + k * kSlotsPerPair + 3) = cz;
      lds(kPairSlot0 + k * kSlotsPerPair + 4) = aref;
      lds(kPairSlot0 + k * kSlotsPerPair + 5) = D;
    });
  }
  return WaveUniform(ends);
}

// Jacobian entries of pair contact k: f(j, J_j) for the hinges between its two bodies
template <int K, typename T, typename Lds, typename F>
EPA_HD void ForPairCols(const CheetahPos<T>& p, Lds&& lds, F&& f) {
  constexpr int b1 = PairBody1(K), b2 = PairBody2(K);
  const T nx = lds(kPairSlot0 + K * kSlotsPerPair + 0), nz = lds(kPairSlot0 + K * kSlotsPerPair + 1);
  const T cx = lds(kPairSlot0 + K * kSlotsPerPair + 2), cz = lds(kPairSlot0 + K * kSlotsPerPair + 3);
  static_for<3, kNV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (InChain(j, b2) && !InChain(j, b1)) {
      constexpr int jb = DofBody(j);
      f(jc, nx * (cz - p.pz[jb]) - nz * (cx - p.px[jb]));
    }
  });
}

// One pass over all constraint rows at acceleration `a`:
// accumulates grad -= J^T f and (if kHess) H += J^T D_active J, returns a
// 64-bit mask of the active rows.  Inactive rows enter with weight 0.
template <bool kHess, typename T, typename Lds>
EPA_HD unsigned long long CheetahRowsPass(const CheetahModel<T>& m,
                                          const CheetahPos<T>& p,
                                          const LimitRows<T>& lim, Lds&& lds,
                                          unsigned ends, const T* a, T* grad,
                                          T* H) {
  unsigned long long mask = 0;
  static_for<0, kNU>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const T jar = lim.sgn[j] * a[j + 3] - lim.aref[j];
    const bool on = lim.sgn[j] != T(0) && jar < T(0);
    const T w = on ? lim.D[j] : T(0);
    grad[j + 3] += lim.sgn[j] * w * jar;
    if constexpr (kHess) H[TriIdx(j + 3, j + 3)] += w;
    mask |= (on ? 1ull : 0ull) << j;
  });
  EPA_NO_UNROLL
  for (unsigned rem = ends & 0xffffu; rem != 0; rem &= rem - 1) {  // scalar loop over end spheres
    const int e = __builtin_ctz(rem);
    const T D = lds(e * kSlotsPerEnd + 4);
    DispatchBody(EndBody(e), [&](auto bc) {  // wave-uniform switch
      constexpr int b = decltype(bc)::value;
      const T cpx = lds(e * kSlotsPerEnd + 0), cpz = lds(e * kSlotsPerEnd + 1);
      const T an = lds(e * kSlotsPerEnd + 2), ax = lds(e * kSlotsPerEnd + 3);
      T jna = T(0), jxa = T(0);
      ForChainCols<b>(p, cpx, cpz, [&](auto jc, T jn, T jx) {
        constexpr int j = decltype(jc)::value;
        jna += jn * a[j];
        jxa += jx * a[j];
      });
      // rows: 2 x (Jn), (Jn - mu Jx), (Jn + mu Jx); D == 0 for lanes not in
      // contact, which zeroes every weight below
      const T mu = m.bmu[b];
      const T jar1 = jna - an;
      const T jar2 = jna - mu * jxa - (an + ax);
      const T jar3 = jna + mu * jxa - (an - ax);
      const bool on = D > T(0);
      const bool a1 = on && jar1 < T(0), a2 = on && jar2 < T(0), a3 = on && jar3 < T(0);
      const T w1 = a1 ? T(2) * D : T(0);
      const T w2 = a2 ? D : T(0);
      const T w3 = a3 ? D : T(0);
      mask |= (a1 ? 1ull : 0ull) << (6 + 3 * e);
      mask |= (a2 ? 1ull : 0ull) << (7 + 3 * e);
      mask |= (a3 ? 1ull : 0ull) << (8 + 3 * e);
      const T gn = w1 * jar1 + w2 * jar2 + w3 * jar3;   // coefficient of Jn
      const T gx = mu * (w3 * jar3 - w2 * jar2);      // coefficient of Jx
      const T A = w1 + w2 + w3, Bc = mu * (w3 - w2), C = mu * mu * (w2 + w3);
      if (WaveAny(A > T(0))) {
        ForChainCols<b>(p, cpx, cpz, [&](auto ic, T jni, T jxi) {
          constexpr int i = decltype(ic)::value;
          grad[i] += jni * gn + jxi * gx;
          if constexpr (kHess) {
            T ui = A * jni + Bc * jxi, wi = Bc * jni + C * jxi;
            ForChainCols<b>(p, cpx, cpz, [&](auto kc, T jnk, T jxk) {
              constexpr int k = decltype(kc)::value;
              if constexpr (k >= i) H[TriIdx(i, k)] += ui * jnk + wi * jxk;
            });
          }
        });
      }
    });
  }
  if (m.n_pairs > 0) {  // frictionless body-body rows (bits 16.. of `ends`)
    static_for<0, kNPair>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if ((ends >> (16 + k)) & 1u) {  // wave uniform
        const T aref = lds(kPairSlot0 + k * kSlotsPerPair + 4);
        const T D = lds(kPairSlot0 + k * kSlotsPerPair + 5);
        T ja = T(0);
        ForPairCols<k>(p, lds, [&](auto jc, T J) { ja += J * a[decltype(jc)::value]; });
        const T jar = ja - aref;
        const bool on = D > T(0) && jar < T(0);
        const T w = on ? D : T(0);
        mask |= (on ? 1ull : 0ull) << (54 + k);
        ForPairCols<k>(p, lds, [&](auto ic, T Ji) {
          constexpr int i = decltype(ic)::value;
          grad[i] += Ji * w * jar;
          if constexpr (kHess) {
            ForPairCols<k>(p, lds, [&](auto jc2, T Jk) {
              constexpr int kk = decltype(jc2)::value;
              if constexpr (kk >= i) H[TriIdx(i, kk)] += w * Ji * Jk;
            });
          }
        });
      }
    });
  }
  return mask;
}

// phi'(alpha), phi''(alpha) contribution of the rows along `s` from `a`.
template <typename T, typename Lds>
EPA_HD void CheetahLineEval(const CheetahModel<T>& m, const CheetahPos<T>& p,
                            const LimitRows<T>& lim, Lds&& lds, unsigned ends,
                            const T* a, const T* s, T alpha, T* d1, T* d2) {
  static_for<0, kNU>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const T jar = lim.sgn[j] * a[j + 3] - lim.aref[j];
    const T jv = lim.sgn[j] * s[j + 3];
    const T x = jar + alpha * jv;
    const T w = (lim.sgn[j] != T(0) && x < T(0)) ? lim.D[j] : T(0);
    *d1 += w * x * jv;
    *d2 += w * jv * jv;
  });
  EPA_NO_UNROLL
  for (unsigned rem = ends & 0xffffu; rem != 0; rem &= rem - 1) {  // scalar loop over end spheres
    const int e = __builtin_ctz(rem);
    const T D = lds(e * kSlotsPerEnd + 4);
    DispatchBody(EndBody(e), [&](auto bc) {  // wave-uniform switch
      constexpr int b = decltype(bc)::value;
      const T cpx = lds(e * kSlotsPerEnd + 0), cpz = lds(e * kSlotsPerEnd + 1);
      const T an = lds(e * kSlotsPerEnd + 2), ax = lds(e * kSlotsPerEnd + 3);
      T jna = T(0), jxa = T(0), jns = T(0), jxs = T(0);
      ForChainCols<b>(p, cpx, cpz, [&](auto jc, T jn, T jx) {
        constexpr int j = decltype(jc)::value;
        jna += jn * a[j];
        jxa += jx * a[j];
        jns += jn * s[j];
        jxs += jx * s[j];
      });
      const T mu = m.bmu[b];
      const T jar1 = jna - an, jv1 = jns;
      const T jar2 = jna - mu * jxa - (an + ax), jv2 = jns - mu * jxs;
      const T jar3 = jna + mu * jxa - (an - ax), jv3 = jns + mu * jxs;
      const T x1 = jar1 + alpha * jv1, x2 = jar2 + alpha * jv2, x3 = jar3 + alpha * jv3;
      // D == 0 for lanes not in contact
      const T c1 = x1 < T(0) ? T(2) * D : T(0);
      const T c2 = x2 < T(0) ? D : T(0);
      const T c3 = x3 < T(0) ? D : T(0);
      *d1 += c1 * x1 * jv1 + c2 * x2 * jv2 + c3 * x3 * jv3;
      *d2 += c1 * jv1 * jv1 + c2 * jv2 * jv2 + c3 * jv3 * jv3;
    });
  }
  if (m.n_pairs > 0) {
    static_for<0, kNPair>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if ((ends >> (16 + k)) & 1u) {
        const T aref = lds(kPairSlot0 + k * kSlotsPerPair + 4);
        const T D = lds(kPairSlot0 + k * kSlotsPerPair + 5);
        T ja = T(0), js = T(0);
        ForPairCols<k>(p, lds, [&](auto jc, T J) {
          ja += J * a[decltype(jc)::value];
          js += J * s[decltype(jc)::value];
        });
        const T x = ja - aref + alpha * js;
        const T w = x < T(0) ? D : T(0);  // D == 0 on lanes without this contact
        *d1 += w * x * js;
        *d2 += w * js * js;
      }
    });
  }
}

template <typename T>
struct SolverCfg {
  int max_iter;
  T gtol;  // stop when |grad| <= gtol * (1 + |qfrc_smooth|_inf)
};

// mj_fwdConstraint: exact Newton on the primal objective
//   1/2 (a-a0)^T M (a-a0) + sum_r 1/2 D_r min(0, J_r a - aref_r)^2
// started from qacc_warmstart.  Outputs qacc and the final gradient
// (so that qfrc_constraint = M qacc - qfrc_smooth - grad).
template <typename T, typename Lds>
EPA_HD int CheetahSolve(const CheetahModel<T>& m, const CheetahPos<T>& p,
                        const LimitRows<T>& lim, Lds&& lds, unsigned ends,
                        const T* qfrc_smooth, const SolverCfg<T>& cfg, T* qacc,
                        T* Ma, T* grad) {
  T fs = T(0);
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    T x = qfrc_smooth[i] < T(0) ? -qfrc_smooth[i] : qfrc_smooth[i];
    fs = x > fs ? x : fs;
  });
  const T gstop = cfg.gtol * (T(1) + fs);
  const T gstop2 = gstop * gstop;
  // rounding floor: once the gradient is this small and has stopped shrinking
  // the iterate is as converged as the arithmetic allows
  const T gfloor = (sizeof(T) == 4 ? T(1e-4) : T(1e-9)) * (T(1) + fs);
  const T gfloor2 = gfloor * gfloor;
  T prev_gn2 = T(-1);  // squared gradient norm of the previous iterate
  SymMul(p.M, qacc, Ma);  // kept current incrementally: Ma += alpha * M s
  unsigned long long prev_mask = ~0ull;
  bool full_step = false;
  bool live = true;  // this lane is still iterating
  int iter = 0;
  for (int it = 0; it < cfg.max_iter; ++it) {
    // Every lane (also the finished ones, whose qacc and Ma are frozen)
    // rebuilds H and grad at its current qacc, so both are current on exit.
    T H[kTri];
    static_for<0, kTri>([&](auto kc) { H[decltype(kc)::value] = p.M[decltype(kc)::value]; });
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      grad[i] = Ma[i] - qfrc_smooth[i];
    });
    unsigned long long mask =
        CheetahRowsPass<true>(m, p, lim, lds, ends, qacc, grad, H);
    T gn2 = T(0);
    static_for<0, kNV>([&](auto ic) { gn2 += grad[decltype(ic)::value] * grad[decltype(ic)::value]; });
    // |grad| <= gstop; or finite termination: same active set after a full
    // Newton step; or at the rounding floor and no longer shrinking (x4)
    const bool stop = gn2 <= gstop2 || (full_step && mask == prev_mask) ||
                      (prev_gn2 >= T(0) && gn2 <= gfloor2 && gn2 >= T(0.0625) * prev_gn2);
    live = live && !stop;
    if (!WaveAny(live)) break;
    iter += live ? 1 : 0;
    prev_gn2 = gn2;
    prev_mask = mask;
    T s[kNV];
    {
      using S = typename Wide<T>::type;
      S Hs[kTri], ss[kNV];
      static_for<0, kTri>([&](auto kc) { Hs[decltype(kc)::value] = (S)H[decltype(kc)::value]; });
      static_for<0, kNV>([&](auto ic) { ss[decltype(ic)::value] = -(S)grad[decltype(ic)::value]; });
      FactorUUt(Hs);
      SolveUUt(Hs, ss);
      static_for<0, kNV>([&](auto ic) { s[decltype(ic)::value] = (T)ss[decltype(ic)::value]; });
    }
    // exact line search on the convex piecewise-quadratic phi(alpha)
    T Ms[kNV];
    SymMul(p.M, s, Ms);
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
      CheetahLineEval(m, p, lim, lds, ends, qacc, s, alpha, &d1, &d2);
      const T ad1 = d1 < T(0) ? -d1 : d1;
      const bool hit = ad1 <= ls_tol;
      // a full Newton step is exact for the active set H was built with
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
  if (WaveAny(live)) {  // iteration cap hit somewhere in the wave: refresh grad
    T grad2[kNV];
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      grad2[i] = Ma[i] - qfrc_smooth[i];
    });
    CheetahRowsPass<false>(m, p, lim, lds, ends, qacc, grad2, static_cast<T*>(nullptr));
    static_for<0, kNV>([&](auto ic) {  // only for the lanes that did hit the cap
      constexpr int i = decltype(ic)::value;
      grad[i] = live ? grad2[i] : grad[i];
    });
  }
  return iter;
}

// mj_forward: qacc at (q, v) under ctrl; `warm` is qacc_warmstart in/out.
// Returns the number of Newton iterations.
template <typename T, typename Lds>
EPA_HD int PlanarForward(const CheetahModel<T>& m, const SolverCfg<T>& cfg, const T* q,
                         const T* v, T* warm, const T* ctrl, Lds&& lds, T* qacc) {
  CheetahPos<T> p;
  CheetahKinematics(m, q, p);
  T qfrc_smooth[kNV];
  CheetahSmoothForces(m, p, q, v, ctrl, qfrc_smooth);
  LimitRows<T> lim;
  const unsigned ends = CheetahMakeConstraint(m, p, q, v, lim, lds);
  T Ma[kNV], grad[kNV];
  static_for<0, kNV>([&](auto ic) { qacc[decltype(ic)::value] = warm[decltype(ic)::value]; });
  int iters = CheetahSolve(m, p, lim, lds, ends, qfrc_smooth, cfg, qacc, Ma, grad);
  static_for<0, kNV>([&](auto ic) { warm[decltype(ic)::value] = qacc[decltype(ic)::value]; });
  return iters;
}

// One mj_step with integrator RK4 (mj_RungeKutta(4)), the Walker2d setting
// (walker2d_envpool.xml:29): four forward evaluations per step, plain explicit
// damping (no eulerdamp).  Same scheme as ant::AntStep, without quaternions.
template <typename T, typename Lds>
EPA_HD int PlanarStepRK4(const CheetahModel<T>& m, const SolverCfg<T>& cfg, T* q, T* v,
                         T* warm, const T* ctrl, Lds&& lds) {
  const T h = m.timestep;
  // q0, v0: state at the start; qs, vs: state of the current stage; dq, dv: running
  // B-weighted sums.  The previous stage's velocity / acceleration are vs / F
  // themselves (read before they are overwritten), so no extra copies stay live
  // across the forward evaluations.
  T q0[kNV], v0[kNV], qs[kNV], vs[kNV], F[kNV], dq[kNV], dv[kNV];
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    q0[i] = qs[i] = q[i];
    v0[i] = vs[i] = v[i];
    dq[i] = dv[i] = T(0);
  });
  // stages 1..4 through ONE instance of the forward pass (a rolled loop: four inlined copies
  // are 4x the code, far beyond the instruction cache):
  // X_i = X_0 + h * a_i * (Xv_{i-1}, F_{i-1}), a = 1/2, 1/2, 1; weights 1/6, 1/3, 1/3, 1/6
  int it = 0;
#pragma nounroll
  for (int stage = 0; stage < 4; ++stage) {
    it += PlanarForward(m, cfg, qs, vs, warm, ctrl, lds, F);
    const T bw = (stage == 0 || stage == 3) ? T(1.0 / 6.0) : T(1.0 / 3.0);
    const T a = stage == 2 ? T(1) : T(0.5);
    static_for<0, kNV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dq[i] += bw * vs[i];
      dv[i] += bw * F[i];
      qs[i] = q0[i] + h * (a * vs[i]);  // the state of the next stage (unused after stage 4)
      vs[i] = v0[i] + h * a * F[i];
    });
  }
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v[i] = v0[i] + h * dv[i];
    q[i] = q0[i] + h * dq[i];
  });
  return it;
}

// Models sharing this kernel.  kSign[j] = -1 where the MJCF hinge rotates about
// -y: the kernel integrates q' = sign * q (see BuildWalkerModel).
enum PlanarModelId { kPlanarCheetah = 0, kPlanarWalker = 1, kPlanarWalkerV5 = 2, kPlanarHopper = 3 };
EPA_HD constexpr int PlanarDofSign(int model, int j) {
  return (model != kPlanarCheetah && j >= 3) ? -1 : 1;
}

// One mj_step.  q[0] is carried as a local offset (caller accumulates the
// absolute root x in fp64); returns the number of Newton iterations.
template <typename T, typename Lds>
EPA_HD int CheetahStep(const CheetahModel<T>& m, const SolverCfg<T>& cfg, T* q,
                       T* v, T* warm, const T* ctrl, Lds&& lds) {
  CheetahPos<T> p;
  CheetahKinematics(m, q, p);
  T qfrc_smooth[kNV];
  CheetahSmoothForces(m, p, q, v, ctrl, qfrc_smooth);
  LimitRows<T> lim;
  const unsigned ends = CheetahMakeConstraint(m, p, q, v, lim, lds);
  T qacc[kNV], Ma[kNV], grad[kNV];
  static_for<0, kNV>([&](auto ic) { qacc[decltype(ic)::value] = warm[decltype(ic)::value]; });
  int iters = CheetahSolve(m, p, lim, lds, ends, qfrc_smooth, cfg, qacc, Ma, grad);
  static_for<0, kNV>([&](auto ic) { warm[decltype(ic)::value] = qacc[decltype(ic)::value]; });
  // mj_Euler with implicit joint damping:
  //   (M + h diag(damping)) qacc_d = qfrc_smooth + qfrc_constraint = Ma - grad
  using S = typename Wide<T>::type;
  S A[kTri], rhs[kNV];
  static_for<0, kTri>([&](auto kc) { A[decltype(kc)::value] = (S)p.M[decltype(kc)::value]; });
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    rhs[i] = (S)Ma[i] - (S)grad[i];
    if constexpr (i >= 3) A[TriIdx(i, i)] += (S)m.timestep * (S)m.damp[i - 3];
  });
  FactorUUt(A);
  SolveUUt(A, rhs);
  static_for<0, kNV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v[i] += m.timestep * (T)rhs[i];
    q[i] += m.timestep * v[i];
  });
  return iters;
}

}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_CHEETAH_HIP_H_
