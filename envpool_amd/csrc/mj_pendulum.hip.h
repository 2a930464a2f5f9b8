// Planar N-link serial chain on one of three bases (gym InvertedPendulum N=1 and
// InvertedDoublePendulum N=2: sliding cart; Reacher N=2: fixed base; Swimmer N=3:
// planar floating base = two slides + a hinge on the first link).  Reacher and
// Swimmer move in the xy plane about +z, which is this file's (x, z) plane about
// +y after z := -y.  The MuJoCo 3.6.0 forward pipeline restated for this chain, same scheme
// as mj_cheetah.hip.h (planar spatial algebra about the system COM, CRB, RNE,
// primal Newton on the constraint objective, RK4) -- without contacts: every
// geom of inverted_pendulum_envpool.xml:21 / inverted_double_pendulum_envpool.xml:39 /
// reacher_envpool.xml:21 / swimmer_envpool.xml:21 has contype=0, so joint limits
// are the only constraint rows.  Swimmer adds MuJoCo's inertia-box fluid forces
// (mj_passive with <option density viscosity>, swimmer_envpool.xml:19).
// Call sites in the reference: envpool/mujoco/gym/mujoco_env.h:126-148
// (mj_resetData, mj_forward, frame_skip x mj_step); the arithmetic itself lives
// in un-vendored MuJoCo, see oracle/mjcpu/mjcpu.h (PARITY UNPINNED).
// One env per thread.  Dofs: [base dofs,] then one hinge per link:
//   kBaseFixed: hinge l = dof l, first anchor at the origin
//   kBaseCart : dof 0 = cart slide x, hinge l = dof l + 1
//   kBaseFree : dofs 0, 1 = slides x, z and dof 2 = hinge of link 0, hinge l = dof l + 2
#ifndef ENVPOOL_AMD_CSRC_MJ_PENDULUM_HIP_H_
#define ENVPOOL_AMD_CSRC_MJ_PENDULUM_HIP_H_

#include "mj_cheetah.hip.h"  // static_for, V3, In4, MulInert, Dot, Cross*, Impedance, WaveAny, SolverCfg

namespace epa {
namespace mj {
namespace pend {

enum { kBaseFixed = 0, kBaseCart = 1, kBaseFree = 2 };

template <typename T, int NL, int kBase = kBaseCart>
struct PendModel {
  static constexpr int kNV = NL + kBase;        // 0 / 1 / 2 extra dofs
  static constexpr int kNB = NL + (kBase == kBaseCart ? 1 : 0);
  T cart_mass;          // kBaseCart only
  T mass[NL], iyy[NL];  // link mass, inertia about the plane normal through its COM
  T cx[NL], cz[NL];     // link COM in the link frame (origin = its hinge)
  T lx[NL], lz[NL];     // next hinge (last link: the "tip" point) in the link frame
  T damp[NL + 2], arm[NL + 2];  // per dof
  T grav_x, grav_z;     // in-plane gravity: (1e-5, -9.81) for the double pendulum, else 0 / -9.81
  T gear[NL + 2];       // motor gear per dof (0: not actuated); ctrl is indexed by dof
  T ctrl_lo, ctrl_hi;
  int limited[NL + 2];
  T lo[NL + 2], hi[NL + 2], margin[NL + 2], dof_invw[NL + 2];
  T lim_K, lim_B, lim_d0, lim_dmax, lim_width;
  // inertia-box fluid model (mj_inertiaBoxFluidModel): medium density / viscosity
  // and the equivalent box of every link: [0] along the link x axis, [1] the
  // in-plane lateral axis, [2] the plane normal (full sizes)
  T fluid_density, fluid_viscosity;
  T box[NL][3];
  T timestep, total_mass;
};

// What a forward pass leaves behind besides qacc (MuJoCo keeps these in mjData;
// after an RK4 step they belong to the LAST stage's evaluation).
template <typename T, int NL>
struct PendAux {
  T tip_x, tip_z;
  T qfrc_constraint[NL + 2];
};

// dof j -> index of the body (in [cart,] link order) it moves
template <int kBase>
EPA_HD constexpr int PendDofBody(int j) {
  return kBase == kBaseFree ? (j < 3 ? 0 : j - 2) : j;
}

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
  V3<T> cdof[NL + 2];
  T M[(NL + 2) * (NL + 2)];
  T px[NL + 1], pz[NL + 1];  // body COMs
  T sn[NL], cs[NL];          // link frames
  T comx, comz;              // system COM (reference point of the spatial vectors)
  T tip_x, tip_z;
};

// mj_kinematics + mj_comPos + mj_crb.  Bodies: [cart,] link 0 .. link NL-1.
template <typename T, int NL, int kBase>
EPA_HD void PendKinematics(const PendModel<T, NL, kBase>& m, const T* q, PendPos<T, NL>& p) {
  constexpr int C = kBase == kBaseCart ? 1 : 0;   // body index of link 0
  constexpr int H = kBase == kBaseFree ? 2 : C;   // dof index of the hinge of link 0
  constexpr int NV = NL + kBase, NB = NL + C;
  T ax[NL], az[NL];  // hinge anchors
  const T x0 = kBase == kBaseFixed ? T(0) : q[0];
  const T z0 = kBase == kBaseFree ? q[1] : T(0);
  if constexpr (kBase == kBaseCart) {
    p.px[0] = x0;
    p.pz[0] = T(0);
  }
  {
    T phi = T(0);
    T nx = x0, nz = z0;  // anchor of the next link
    static_for<0, NL>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      phi += q[l + H];
      SinCos(phi, &p.sn[l], &p.cs[l]);
      const T sn = p.sn[l], cs = p.cs[l];
      ax[l] = nx;
      az[l] = nz;
      p.px[l + C] = nx + cs * m.cx[l] + sn * m.cz[l];
      p.pz[l + C] = nz - sn * m.cx[l] + cs * m.cz[l];
      const T tx = nx + cs * m.lx[l] + sn * m.lz[l];
      const T tz = nz - sn * m.lx[l] + cs * m.lz[l];
      nx = tx;
      nz = tz;
    });
    p.tip_x = nx;
    p.tip_z = nz;
  }
  // mj_comPos
  T comx = kBase == kBaseCart ? m.cart_mass * p.px[0] : T(0), comz = T(0);
  static_for<0, NL>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    comx += m.mass[l] * p.px[l + C];
    comz += m.mass[l] * p.pz[l + C];
  });
  comx /= m.total_mass;
  comz /= m.total_mass;
  p.comx = comx;
  p.comz = comz;
  static_for<0, NB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr bool cart = kBase == kBaseCart && b == 0;
    constexpr int l = cart ? 0 : b - C;
    const T mass = cart ? m.cart_mass : m.mass[l];
    const T iyy = cart ? T(0) : m.iyy[l];  // the cart never rotates
    const T dx = p.px[b] - comx, dz = p.pz[b] - comz;
    p.cinert[b] = {iyy + mass * (dx * dx + dz * dz), mass * dx, mass * dz, mass};
  });
  if constexpr (kBase != kBaseFixed) p.cdof[0] = {T(0), T(1), T(0)};
  if constexpr (kBase == kBaseFree) p.cdof[1] = {T(0), T(0), T(1)};
  static_for<0, NL>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    p.cdof[l + H] = {T(1), comz - az[l], -(comx - ax[l])};
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
    V3<T> buf = MulInert(crb[PendDofBody<kBase>(i)], p.cdof[i]);
    static_for<0, i + 1>([&](auto jc) {  // a serial chain: every lower dof is an ancestor
      constexpr int j = decltype(jc)::value;
      const T x = Dot(p.cdof[j], buf) + (i == j ? m.arm[i] : T(0));
      p.M[i * NV + j] = x;
      p.M[j * NV + i] = x;
    });
  });
}

// mj_forward.  q, v: state; ctrl[dof]: raw action of the motor on that dof
// (clamped to ctrlrange here, the motors are ctrllimited); warm: qacc_warmstart in/out.
template <typename T, int NL, int kBase>
EPA_HD int PendForward(const PendModel<T, NL, kBase>& m, const SolverCfg<T>& cfg, const T* q,
                       const T* v, const T* ctrl, T* warm, T* qacc, PendAux<T, NL>& aux) {
  constexpr int C = kBase == kBaseCart ? 1 : 0;
  constexpr int NV = NL + kBase, NB = NL + C;
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
    // cfrc[b]: RNE bias force of body b; with a medium, minus the fluid force on it
    V3<T> cfrc[NB];
    V3<T> cv = {T(0), T(0), T(0)};
    V3<T> ca = {T(0), -m.grav_x, -m.grav_z};  // world cacc = -gravity
    static_for<0, NV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int b = PendDofBody<kBase>(j);
      // dofs of a body in joint order: cdof_dot uses the velocity accumulated so far
      V3<T> cdd = CrossMotion(cv, cdof[j]);
      ca.w += cdd.w * v[j];
      ca.x += cdd.x * v[j];
      ca.z += cdd.z * v[j];
      cv.w += cdof[j].w * v[j];
      cv.x += cdof[j].x * v[j];
      cv.z += cdof[j].z * v[j];
      if constexpr (j + 1 == NV || PendDofBody<kBase>(j + 1 < NV ? j + 1 : j) != b) {  // last dof of b
        V3<T> f = MulInert(cinert[b], ca);
        V3<T> g = CrossForce(cv, MulInert(cinert[b], cv));
        cfrc[b] = {f.w + g.w, f.x + g.x, f.z + g.z};
        if constexpr (kBase == kBaseFree) {  // fluid forces exist only in the Swimmer model
          constexpr int l = b - C;
          // velocity of the link COM (spatial velocity is about the system COM), in link axes
          const T dx = pp.px[b] - pp.comx, dz = pp.pz[b] - pp.comz;
          const T vx = cv.x + cv.w * dz, vz = cv.z - cv.w * dx, om = cv.w;
          const T sn = pp.sn[l], cs = pp.cs[l];
          const T vl0 = vx * cs - vz * sn, vl1 = vx * sn + vz * cs;  // along / across the link
          const T b0 = m.box[l][0], b1 = m.box[l][1], b2 = m.box[l][2];
          const T diam = (b0 + b1 + b2) / T(3);
          const T kPi = T(3.14159265358979323846);
          const T a0 = vl0 < T(0) ? -vl0 : vl0, a1 = vl1 < T(0) ? -vl1 : vl1;
          const T ao = om < T(0) ? -om : om;
          // mj_inertiaBoxFluidModel: viscous (Stokes, equivalent sphere) + quadratic drag
          const T f0 = -T(3) * kPi * diam * m.fluid_viscosity * vl0 -
                       T(0.5) * m.fluid_density * b1 * b2 * a0 * vl0;
          const T f1 = -T(3) * kPi * diam * m.fluid_viscosity * vl1 -
                       T(0.5) * m.fluid_density * b0 * b2 * a1 * vl1;
          const T tq = -kPi * diam * diam * diam * m.fluid_viscosity * om -
                       m.fluid_density * b2 * (b0 * b0 * b0 * b0 + b1 * b1 * b1 * b1) * ao * om / T(64);
          const T fx = f0 * cs + f1 * sn, fz = -f0 * sn + f1 * cs;  // back to plane axes
          // spatial force about the system COM; passive forces enter with the opposite
          // sign of the bias
          cfrc[b].w -= tq + dz * fx - dx * fz;
          cfrc[b].x -= fx;
          cfrc[b].z -= fz;
        }
      }
    });
    static_for_down<NB, 1>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      cfrc[b - 1].w += cfrc[b].w;
      cfrc[b - 1].x += cfrc[b].x;
      cfrc[b - 1].z += cfrc[b].z;
    });
    static_for<0, NV>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      qfrc_smooth[j] = -m.damp[j] * v[j] - Dot(cdof[j], cfrc[PendDofBody<kBase>(j)]);
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
  constexpr int kChainLsExactAfter = 8;
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
    T r0[NV];  // M qacc - qfrc_smooth: the smooth part of the gradient, again the line search's phi'(0) term
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      T g = -qfrc_smooth[i];
      static_for<0, NV>([&](auto jc) { g += M[i * NV + decltype(jc)::value] * qacc[decltype(jc)::value]; });
      r0[i] = g;
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
      T ms = T(0);
      static_for<0, NV>([&](auto jc) { ms += M[i * NV + decltype(jc)::value] * s[decltype(jc)::value]; });
      g1 += s[i] * r0[i];
      g2 += s[i] * ms;
    });
    T alpha = T(1), lo = T(0), hi = T(-1);
    full_step = false;
    const T ag1 = g1 < T(0) ? -g1 : g1;
    const T ls_tol = (sizeof(T) == 4 ? T(1e-4) : T(1e-10)) * ag1;
    bool searching = live;
    // Round 6 (the planar kernels' solver, mj_planar_lg.hip.h::Solve): ONE evaluation, at the full step -- phi'(1) = 0
    // with the active set H was built with means a + s IS the minimiser (the env is done, without the pass over the
    // rows that would only have confirmed it); otherwise one Newton step of the 1-D problem, unverified.  From trip
    // kChainLsExactAfter on the search is exact again (its steps cannot increase the objective).
    bool exact = false;
    const int ls_max = it < kChainLsExactAfter ? 1 : 24;
    for (int ls = 0; ls < ls_max; ++ls) {
      T d1 = g1 + alpha * g2, d2 = g2;
      unsigned mask1 = 0;
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const T jar = sgn[i] * qacc[i] - aref[i], jv = sgn[i] * s[i];
        const T x = jar + alpha * jv;
        const bool on1 = sgn[i] != T(0) && x < T(0);
        const T w = on1 ? D[i] : T(0);
        d1 += w * x * jv;
        d2 += w * jv * jv;
        mask1 |= (on1 ? 1u : 0u) << i;
      });
      const T ad1 = d1 < T(0) ? -d1 : d1;
      const bool hit = ad1 <= ls_tol;
      if (ls == 0) {
        full_step = searching && hit;
        exact = full_step && mask1 == mask;
      }
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
    live = live && !exact;
    if (!WaveAny(live)) break;
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

// mj_step with integrator RK4 (all four models: <option integrator="RK4">).
template <typename T, int NL, int kBase>
EPA_HD int PendStepRK4(const PendModel<T, NL, kBase>& m, const SolverCfg<T>& cfg, T* q, T* v,
                       T* warm, const T* ctrl, PendAux<T, NL>& aux) {
  constexpr int NV = NL + kBase;
  const T h = m.timestep;
  T q0[NV], v0[NV], qs[NV], vs[NV], F[NV], dq[NV], dv[NV];  // vs / F double as the previous stage's
  int it = PendForward(m, cfg, q, v, ctrl, warm, F, aux);
  static_for<0, NV>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    q0[i] = q[i];
    v0[i] = v[i];
    dq[i] = v0[i] * T(1.0 / 6.0);
    dv[i] = F[i] * T(1.0 / 6.0);
    vs[i] = v0[i];
  });
  for (int stage = 1; stage < 4; ++stage) {
    const T a = stage == 3 ? T(1) : T(0.5);
    const T bw = stage == 3 ? T(1.0 / 6.0) : T(1.0 / 3.0);
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      qs[i] = q0[i] + h * (a * vs[i]);
      vs[i] = v0[i] + h * a * F[i];
    });
    it += PendForward(m, cfg, qs, vs, ctrl, warm, F, aux);
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dq[i] += bw * vs[i];
      dv[i] += bw * F[i];
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

#endif  // ENVPOOL_AMD_CSRC_MJ_PENDULUM_HIP_H_
