// Planar N-link chain, optionally on a sliding cart (gym InvertedPendulum N=1 and
// InvertedDoublePendulum N=2 with cart; Reacher N=2 without: its arm moves in the
// xy plane about +z, which is this file's (x, z) plane about +y after z := -y):
// the MuJoCo 3.6.0 forward pipeline restated for this chain, same scheme
// as mj_cheetah.cuh (planar spatial algebra about the system COM, CRB, RNE,
// primal Newton on the constraint objective, RK4) -- without contacts: every
// geom of inverted_pendulum_envpool.xml:21 / inverted_double_pendulum_envpool.xml:39
// (reacher_envpool.xml:21) has contype=0, so joint limits are the only constraint rows.
// Call sites in the reference: envpool/mujoco/gym/mujoco_env.h:126-148
// (mj_resetData, mj_forward, frame_skip x mj_step); the arithmetic itself lives
// in un-vendored MuJoCo, see oracle/mjcpu/mjcpu.h (PARITY UNPINNED).
// One env per thread; with a cart dof 0 is its slide (x) and hinge j is dof j,
// without one the hinges are dofs 0..N-1 and the first anchor is the origin.
#ifndef ENVPOOL_AMD_CSRC_MJ_PENDULUM_CUH_
#define ENVPOOL_AMD_CSRC_MJ_PENDULUM_CUH_

#include "mj_cheetah.cuh"  // static_for, V3, In4, MulInert, Dot, Cross*, Impedance, WaveAny, SolverCfg

namespace epa {
namespace mj {
namespace pend {

template <typename T, int NL, bool kCart = true>
struct PendModel {
  static constexpr int kC = kCart ? 1 : 0;
  static constexpr int kNV = NL + kC;
  T cart_mass;          // 0 without a cart
  T mass[NL], iyy[NL];  // link mass, inertia about the plane normal through its COM
  T cx[NL], cz[NL];     // link COM in the link frame (origin = its hinge)
  T lx[NL], lz[NL];     // next hinge (last link: the "tip" point) in the link frame
  T damp[NL + 1], arm[NL + 1];  // per dof (entry NL unused without a cart)
  T grav_x, grav_z;     // in-plane gravity: (1e-5, -9.81) for the double pendulum, 0 for Reacher
  T gear[NL + 1];       // motor gear per dof (0: not actuated); ctrl is indexed by dof
  T ctrl_lo, ctrl_hi;
  int limited[NL + 1];
  T lo[NL + 1], hi[NL + 1], margin[NL + 1], dof_invw[NL + 1];
  T lim_K, lim_B, lim_d0, lim_dmax, lim_width;
  T timestep, total_mass;
};

// What a forward pass leaves behind besides qacc (MuJoCo keeps these in mjData;
// after an RK4 step they belong to the LAST stage's evaluation).
template <typename T, int NL>
struct PendAux {
  T tip_x, tip_z;
  T qfrc_constraint[NL + 1];
};
template <typename T, int NL, bool kCart>
using PendM = PendModel<T, NL, kCart>;

template <typename T, int N>
EPA_HD void CholSolve(T* A, T* x) {  // A: full N x N SPD (row major), in place; x <- A^-1 x
  static_for<0, N>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T s = A[j * N + j];
    static_for<0, j>([&](auto kc) { s -= A[j * N + decltype(kc)::value] * A[j * N + decltype(kc)::value]; });
    const T inv = Rsqrt(s);
    A[j * N + j] = inv;
    static_for<j + 1, N>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      T t = A[i * N + j];
      static_for<0, j>([&](auto kc) { t -= A[i * N + decltype(kc)::value] * A[j * N + decltype(kc)::value]; });
      A[i * N + j] = t * inv;
    });
  });
  static_for<0, N>([&](auto ic) {  // L y = b
    constexpr int i = decltype(ic)::value;
    T s = x[i];
    static_for<0, i>([&](auto kc) { s -= A[i * N + decltype(kc)::value] * x[decltype(kc)::value]; });
    x[i] = s * A[i * N + i];
  });
  static_for_down<N, 0>([&](auto ic) {  // L^T x = y
    constexpr int i = decltype(ic)::value;
    T s = x[i];
    static_for<i + 1, N>([&](auto kc) { s -= A[decltype(kc)::value * N + i] * x[decltype(kc)::value]; });
    x[i] = s * A[i * N + i];
  });
}

// Position-dependent part of a forward pass.
template <typename T, int NL>
struct PendPos {
  In4<T> cinert[NL + 1];
  V3<T> cdof[NL + 1];
  T M[(NL + 1) * (NL + 1)];
  T tip_x, tip_z;
};

// mj_kinematics + mj_comPos + mj_crb.  Bodies/dofs: [cart,] link 0 .. link NL-1.
template <typename T, int NL, bool kCart>
EPA_HD void PendKinematics(const PendModel<T, NL, kCart>& m, const T* q, PendPos<T, NL>& p) {
  constexpr int C = kCart ? 1 : 0, NV = NL + C, NB = NL + C;
  T ax[NL], az[NL], px[NB], pz[NB];  // hinge anchors, body COMs
  const T x0 = kCart ? q[0] : T(0);
  if constexpr (kCart) {
    px[0] = x0;
    pz[0] = T(0);
  }
  {
    T phi = T(0);
    T nx = x0, nz = T(0);  // anchor of the next link
    static_for<0, NL>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      T sn, cs;
      phi += q[l + C];
      SinCos(phi, &sn, &cs);
      ax[l] = nx;
      az[l] = nz;
      px[l + C] = nx + cs * m.cx[l] + sn * m.cz[l];
      pz[l + C] = nz - sn * m.cx[l] + cs * m.cz[l];
      const T tx = nx + cs * m.lx[l] + sn * m.lz[l];
      const T tz = nz - sn * m.lx[l] + cs * m.lz[l];
      nx = tx;
      nz = tz;
    });
    p.tip_x = nx;
    p.tip_z = nz;
  }
  // mj_comPos
  T comx = kCart ? m.cart_mass * px[0] : T(0), comz = T(0);
  static_for<0, NL>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    comx += m.mass[l] * px[l + C];
    comz += m.mass[l] * pz[l + C];
  });
  comx /= m.total_mass;
  comz /= m.total_mass;
  static_for<0, NB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr bool cart = kCart && b == 0;
    constexpr int l = cart ? 0 : b - C;
    const T mass = cart ? m.cart_mass : m.mass[l];
    const T iyy = cart ? T(0) : m.iyy[l];  // the cart never rotates
    const T dx = px[b] - comx, dz = pz[b] - comz;
    p.cinert[b] = {iyy + mass * (dx * dx + dz * dz), mass * dx, mass * dz, mass};
    if constexpr (cart) {
      p.cdof[0] = {T(0), T(1), T(0)};
    } else {
      p.cdof[b] = {T(1), comz - az[l], -(comx - ax[l])};
    }
  });
  // mj_crb
  In4<T> crb[NB];
  static_for<0, NB>([&](auto bc) { crb[decltype(bc)::value] = p.cinert[decltype(bc)::value]; });
  static_for_down<NB, 1>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    crb[b - 1].I += crb[b].I;
    crb[b - 1].mdx += crb[b].mdx;
    crb[b - 1].mdz += crb[b].mdz;
    crb[b - 1].m += crb[b].m;
  });
  static_for<0, NV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    V3<T> buf = MulInert(crb[i], p.cdof[i]);
    static_for<0, i + 1>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const T x = Dot(p.cdof[j], buf) + (i == j ? m.arm[i] : T(0));
      p.M[i * NV + j] = x;
      p.M[j * NV + i] = x;
    });
  });
}

// mj_forward.  q, v: state; ctrl[dof]: raw action of the motor on that dof
// (clamped to ctrlrange here, the motors are ctrllimited); warm: qacc_warmstart in/out.
template <typename T, int NL, bool kCart>
EPA_HD int PendForward(const PendModel<T, NL, kCart>& m, const SolverCfg<T>& cfg, const T* q,
                       const T* v, const T* ctrl, T* warm, T* qacc, PendAux<T, NL>& aux) {
  constexpr int NV = NL + (kCart ? 1 : 0), NB = NV;
  PendPos<T, NL> pp;
  PendKinematics(m, q, pp);
  aux.tip_x = pp.tip_x;
  aux.tip_z = pp.tip_z;
  const In4<T>* cinert = pp.cinert;
  const V3<T>* cdof = pp.cdof;
  const T* M = pp.M;
  // mj_comVel + mj_rne (flg_acc = 0) + mj_passive + mj_fwdActuation
  T qfrc_smooth[NV];
  {
    V3<T> cvel[NB], cacc[NB], cfrc[NB];
    V3<T> cv = {T(0), T(0), T(0)};
    V3<T> ca = {T(0), -m.grav_x, -m.grav_z};  // world cacc = -gravity
    static_for<0, NB>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      V3<T> cdd = CrossMotion(cv, cdof[b]);
      ca.w += cdd.w * v[b];
      ca.x += cdd.x * v[b];
      ca.z += cdd.z * v[b];
      cv.w += cdof[b].w * v[b];
      cv.x += cdof[b].x * v[b];
      cv.z += cdof[b].z * v[b];
      cvel[b] = cv;
      cacc[b] = ca;
      V3<T> f = MulInert(cinert[b], ca);
      V3<T> g = CrossForce(cv, MulInert(cinert[b], cv));
      cfrc[b] = {f.w + g.w, f.x + g.x, f.z + g.z};
    });
    static_for_down<NB, 1>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      cfrc[b - 1].w += cfrc[b].w;
      cfrc[b - 1].x += cfrc[b].x;
      cfrc[b - 1].z += cfrc[b].z;
    });
    static_for<0, NV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      qfrc_smooth[j] = -m.damp[j] * v[j] - Dot(cdof[j], cfrc[j]);
    });
    static_for<0, NV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const T c = ctrl[j] < m.ctrl_lo ? m.ctrl_lo : (ctrl[j] > m.ctrl_hi ? m.ctrl_hi : ctrl[j]);
      qfrc_smooth[j] += m.gear[j] * c;
    });
  }
  // mj_instantiateLimit + mj_makeImpedance: row j is J = sgn e_j when joint j
  // is within `margin` of a bound
  T sgn[NV], D[NV], aref[NV];
  static_for<0, NV>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const T dlo = q[j] - m.lo[j], dhi = m.hi[j] - q[j];
    const bool lim = m.limited[j] != 0;
    const T s = (lim && dlo < m.margin[j]) ? T(1) : ((lim && dhi < m.margin[j]) ? T(-1) : T(0));
    const T dist = s > T(0) ? dlo : (s < T(0) ? dhi : T(0));
    const T imp = Impedance(m.lim_d0, m.lim_dmax, m.lim_width, dist - m.margin[j]);
    const T num = (T(1) - imp) * m.dof_invw[j];  // R = max(mjMINVAL, num / imp)
    const T Dj = num < T(1e-15) * imp ? T(1e15) : imp / num;
    sgn[j] = s;
    D[j] = s != T(0) ? Dj : T(0);
    aref[j] = -m.lim_B * (s * v[j]) - m.lim_K * imp * (dist - m.margin[j]);
  });
  // mj_fwdConstraint: Newton on 1/2 (a-a0)^T M (a-a0) + sum 1/2 D min(0, sgn a_j - aref)^2
  static_for<0, NV>([&](auto ic) { qacc[decltype(ic)::value] = warm[decltype(ic)::value]; });
  T fs = T(0);
  static_for<0, NV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const T x = qfrc_smooth[i] < T(0) ? -qfrc_smooth[i] : qfrc_smooth[i];
    fs = x > fs ? x : fs;
  });
  const T gstop = cfg.gtol * (T(1) + fs), gstop2 = gstop * gstop;
  int iter = 0;
  bool live = true;
  unsigned prev_mask = ~0u;
  bool full_step = false;
  T grad[NV];
  for (int it = 0; it < cfg.max_iter; ++it) {
    T H[NV * NV];
    static_for<0, NV * NV>([&](auto kc) { H[decltype(kc)::value] = M[decltype(kc)::value]; });
    unsigned mask = 0;
    T gn2 = T(0);
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      T g = -qfrc_smooth[i];
      static_for<0, NV>([&](auto jc) { g += M[i * NV + decltype(jc)::value] * qacc[decltype(jc)::value]; });
      const T jar = sgn[i] * qacc[i] - aref[i];
      const bool on = sgn[i] != T(0) && jar < T(0);
      const T w = on ? D[i] : T(0);
      g += sgn[i] * w * jar;
      H[i * NV + i] += w;
      mask |= (on ? 1u : 0u) << i;
      grad[i] = g;
      gn2 += g * g;
    });
    const bool stop = gn2 <= gstop2 || (full_step && mask == prev_mask);
    live = live && !stop;
    if (!WaveAny(live)) break;
    iter += live ? 1 : 0;
    prev_mask = mask;
    T s[NV];
    static_for<0, NV>([&](auto ic) { s[decltype(ic)::value] = -grad[decltype(ic)::value]; });
    CholSolve<T, NV>(H, s);
    // exact line search along s on the piecewise quadratic
    T g1 = T(0), g2 = T(0);
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      T ms = T(0), ma = -qfrc_smooth[i];
      static_for<0, NV>([&](auto jc) {
        ms += M[i * NV + decltype(jc)::value] * s[decltype(jc)::value];
        ma += M[i * NV + decltype(jc)::value] * qacc[decltype(jc)::value];
      });
      g1 += s[i] * ma;
      g2 += s[i] * ms;
    });
    T alpha = T(1), lo = T(0), hi = T(-1);
    full_step = false;
    const T ag1 = g1 < T(0) ? -g1 : g1;
    const T ls_tol = (sizeof(T) == 4 ? T(1e-4) : T(1e-10)) * ag1;
    bool searching = live;
    for (int ls = 0; ls < 24; ++ls) {
      T d1 = g1 + alpha * g2, d2 = g2;
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const T jar = sgn[i] * qacc[i] - aref[i], jv = sgn[i] * s[i];
        const T x = jar + alpha * jv;
        const T w = (sgn[i] != T(0) && x < T(0)) ? D[i] : T(0);
        d1 += w * x * jv;
        d2 += w * jv * jv;
      });
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
    static_for<0, NV>([&](auto ic) { qacc[decltype(ic)::value] += step * s[decltype(ic)::value]; });
  }
  static_for<0, NV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    warm[i] = qacc[i];
    const T jar = sgn[i] * qacc[i] - aref[i];
    const T f = (sgn[i] != T(0) && jar < T(0)) ? -D[i] * jar : T(0);  // efc_force
    aux.qfrc_constraint[i] = sgn[i] * f;
  });
  return iter;
}

// mj_step with integrator RK4 (both models: <option integrator="RK4">).
template <typename T, int NL, bool kCart>
EPA_HD int PendStepRK4(const PendModel<T, NL, kCart>& m, const SolverCfg<T>& cfg, T* q, T* v,
                       T* warm, const T* ctrl, PendAux<T, NL>& aux) {
  constexpr int NV = NL + (kCart ? 1 : 0);
  const T h = m.timestep;
  T q0[NV], v0[NV], qs[NV], vs[NV], F[NV], dq[NV], dv[NV], Xv[NV], Fp[NV];
  int it = PendForward(m, cfg, q, v, ctrl, warm, F, aux);
  static_for<0, NV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    q0[i] = q[i];
    v0[i] = v[i];
    dq[i] = v0[i] * T(1.0 / 6.0);
    dv[i] = F[i] * T(1.0 / 6.0);
    Xv[i] = v0[i];
    Fp[i] = F[i];
  });
  for (int stage = 1; stage < 4; ++stage) {
    const T a = stage == 3 ? T(1) : T(0.5);
    const T bw = stage == 3 ? T(1.0 / 6.0) : T(1.0 / 3.0);
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      qs[i] = q0[i] + h * (a * Xv[i]);
      vs[i] = v0[i] + h * a * Fp[i];
    });
    it += PendForward(m, cfg, qs, vs, ctrl, warm, F, aux);
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dq[i] += bw * vs[i];
      dv[i] += bw * F[i];
      Xv[i] = vs[i];
      Fp[i] = F[i];
    });
  }
  static_for<0, NV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v[i] = v0[i] + h * dv[i];
    q[i] = q0[i] + h * dq[i];
  });
  return it;
}

}  // namespace pend
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_PENDULUM_CUH_
