// TEST HARNESS (not product): compiles the host instantiation of the lane-group planar step
// (envpool_amd/csrc/mj_planar_lg.hip.h) with g++, the KL lanes of one env emulated by LV<double, KL>,
// so the exact kernel source can be diffed against oracle/mjcpu and against the one-env-per-lane
// formulation (mj_cheetah.hip.h) on a CPU box.  Nothing in envpool_amd/ links or loads this.
#include <cstdlib>
#include <cstring>

#include "../../envpool_amd/csrc/mj_cheetah_model.h"
#include "../../envpool_amd/csrc/mj_planar_lg.hip.h"

using namespace epa::mj;

template <int KL>
struct HostCx {
  using V = plg::LV<double, KL>;
  const double* tab;
  V lds[plg::LdsSlots<KL>()];
  V C(int id) const {
    V r;
    for (int c = 0; c < KL; ++c) r.v[c] = tab[id * KL + c];
    return r;
  }
  V& Lds(int slot) { return lds[slot]; }
  // per-lane addressing (a lane visits its own touching slots)
  V LdsL(const plg::LU<KL>& s, int mul, int add) const {
    V r;
    for (int c = 0; c < KL; ++c) r.v[c] = lds[s.v[c] * mul + add].v[c];
    return r;
  }
  void LdsLStore(const plg::LU<KL>& s, int mul, int add, const V& x, plg::LB<KL> on) {
    for (int c = 0; c < KL; ++c) {
      if (on.v[c]) lds[s.v[c] * mul + add].v[c] = x.v[c];
    }
  }
  V CL(int base, const plg::LU<KL>& idx) const {
    V r;
    for (int c = 0; c < KL; ++c) r.v[c] = tab[(base + idx.v[c]) * KL + c];
    return r;
  }
  void Refresh() {}
};

// returns 0, or a negative code if lanes that must agree (replicated torso state, the parity
// lanes of a leg) ended up with different bits
template <int KL>
static int Run(int model, const double* q, const double* v, const double* warm, const double* ctrl,
               int nsub, double* qo, double* vo, double* wo, int* iters) {
  using V = plg::LV<double, KL>;
  using G = plg::Grp<KL>;
  // model: 0 HalfCheetah, 1 Walker2d, 2 Walker2d-v5, 3 Hopper (KL = 1: the lane is the env)
  const CheetahModel<double> m = model == 0 ? BuildCheetahModel()
                                 : model == 3 ? BuildHopperModel() : BuildWalkerModel(model == 2);
  const int pm = model == 0 ? kPlanarCheetah : (model == 3 ? kPlanarHopper : kPlanarWalker);
  double tab[plg::Tab<KL>::kSize];
  plg::BuildTable<KL>(m, tab);
  HostCx<KL> cx;
  cx.tab = tab;
  plg::SolverCfgLg<double> cfg{50, 1e-13};
  if (const char* g = getenv("EPA_LG_GTOL")) cfg.gtol = atof(g);  // experiments with the stopping rule
  V lq[plg::kLV], lv[plg::kLV], lw[plg::kLV], lc[3];
  const double x0 = q[0];
  for (int c = 0; c < KL; ++c) {
    const int leg = G::Leg(c);
    for (int i = 0; i < plg::kLV; ++i) {
      const int g = i < 3 ? i : 3 + 3 * leg + (i - 3);
      const int sg = PlanarDofSign(pm, g);
      lq[i].v[c] = sg * q[g];
      lv[i].v[c] = sg * v[g];
      lw[i].v[c] = sg * warm[g];
    }
    lq[0].v[c] = 0;  // local x
    for (int k = 0; k < 3; ++k) {
      const double a = ctrl[3 * leg + k];
      lc[k].v[c] = a < -1 ? -1 : (a > 1 ? 1 : a);
    }
  }
  V it = V(0);
  for (int s = 0; s < nsub; ++s) {
    it += model == 0 ? plg::StepEuler<KL>(m, cfg, cx, lq, lv, lw, lc)
                     : plg::StepRK4<KL>(m, cfg, cx, lq, lv, lw, lc);
  }
  int rc = 0;
  for (int c = 0; c < KL; ++c) {
    const int leg = G::Leg(c);
    for (int i = 0; i < plg::kLV; ++i) {
      const int g = i < 3 ? i : 3 + 3 * leg + (i - 3);
      const int sg = PlanarDofSign(pm, g);
      const double a = sg * lq[i].v[c], b = sg * lv[i].v[c], w = sg * lw[i].v[c];
      if (c == 0 || (i >= 3 && G::Par(c) == 0)) {
        qo[g] = a;
        vo[g] = b;
        wo[g] = w;
      } else if (std::memcmp(&qo[g], &a, 8) || std::memcmp(&vo[g], &b, 8) || std::memcmp(&wo[g], &w, 8)) {
        rc = -1 - g;  // replicated values must be bit-identical
      }
    }
    if (it.v[c] != it.v[0]) rc = -100;
  }
  qo[0] += x0;
  *iters = (int)it.v[0];
  return rc;
}

extern "C" int planar_lg_step(int model, int kl, const double* q, const double* v, const double* warm,
                              const double* ctrl, int nsub, double* qo, double* vo, double* wo,
                              int* iters) {
  if (kl == 1) return Run<1>(model, q, v, warm, ctrl, nsub, qo, vo, wo, iters);
  return kl == 2 ? Run<2>(model, q, v, warm, ctrl, nsub, qo, vo, wo, iters)
                 : Run<4>(model, q, v, warm, ctrl, nsub, qo, vo, wo, iters);
}
