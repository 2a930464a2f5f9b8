// TEST HARNESS (not product): host instantiation of mj_ant4.hip.h -- the product's
// four-lanes-per-env Ant kernel source with a lane quad emulated by Q4<T> -- for
// diffing against oracle/mjcpu on a CPU box.  Not linked by envpool_amd/.
#include "../../envpool_amd/csrc/mj_ant4.hip.h"
#include "../../envpool_amd/csrc/mj_ant_model.h"

using epa::mj::B4;
using epa::mj::Q4;
using epa::mj::SolverCfg;
using epa::mj::U4;
using namespace epa::mj::ant;
namespace A4 = epa::mj::ant4;

template <typename T>
static void Run(const double* q, const double* v, const double* warm, const double* ctrl,
                int nsub, int wrench, double* qo, double* vo, double* wo, double* lag,
                double* cfrc, int* iters) {
  using V = Q4<T>;
  const AntModel<double> md = BuildAntModel();
  AntModel<T> m = CastAntModel<T>(md);
  SolverCfg<T> cfg{sizeof(T) == 4 ? 12 : 50, sizeof(T) == 4 ? T(1e-6) : T(1e-13)};
  A4::Leg<V, B4> lg;
  V tq[9], tv[8], tw[8], tc[2], lx = V(0), ly = V(0);
  for (int l = 0; l < 4; ++l) {
    lg.sx.v[l] = (T)A4::LegSx(l);
    lg.sy.v[l] = (T)A4::LegSy(l);
    lg.sxy.v[l] = (T)(A4::LegSx(l) * A4::LegSy(l));
    lg.axs.v[l] = (T)A4::LegAxs(l);
    lg.alo.v[l] = (T)md.lo[2 * A4::LegAnkleRef(l) + 1];
    lg.ahi.v[l] = (T)md.hi[2 * A4::LegAnkleRef(l) + 1];
    lg.first.v[l] = l == 0;
    for (int i = 0; i < 7; ++i) tq[i].v[l] = (T)q[i];
    for (int i = 0; i < 6; ++i) {
      tv[i].v[l] = (T)v[i];
      tw[i].v[l] = (T)warm[i];
    }
    for (int c = 0; c < 2; ++c) {
      tq[7 + c].v[l] = (T)q[7 + 2 * l + c];
      tv[6 + c].v[l] = (T)v[6 + 2 * l + c];
      tw[6 + c].v[l] = (T)warm[6 + 2 * l + c];
      // ctrl[u] drives dof CtrlDof(u): hip_4 ankle_4 hip_1 ankle_1 ... (ant_envpool.xml:85-94)
      const double a = ctrl[(2 + 2 * l + c) % 8];
      tc[c].v[l] = (T)(a < -1 ? -1 : (a > 1 ? 1 : a));
    }
  }
  V lds_block[A4::kSlots];  // host: one "lane" holds the quad, every slot is private
  auto lds = [&](int slot) -> V& { return lds_block[slot]; };
  V cf[3][6], cf0[6], n_env = V(0);
  int n_wave = 0;
  for (int s = 0; s < nsub; ++s) {
    const bool last = wrench && s == nsub - 1;
    if (last) {
      for (int k = 0; k < 6; ++k) {
        cf0[k] = V(0);
        for (int b = 0; b < 3; ++b) cf[b][k] = V(0);
      }
    }
    A4::Step<U4, true>(m, lg, cfg, tq, tv, tw, tc, &lx, &ly, lds, last, cf, cf0, &n_env, &n_wave);
  }
  for (int i = 0; i < 7; ++i) qo[i] = tq[i].v[0];
  for (int i = 0; i < 6; ++i) {
    vo[i] = tv[i].v[0];
    wo[i] = tw[i].v[0];
  }
  for (int l = 0; l < 4; ++l) {
    for (int c = 0; c < 2; ++c) {
      qo[7 + 2 * l + c] = tq[7 + c].v[l];
      vo[6 + 2 * l + c] = tv[6 + c].v[l];
      wo[6 + 2 * l + c] = tw[6 + c].v[l];
    }
  }
  *iters = (int)n_env.v[0];
  lag[0] = lx.v[0];
  lag[1] = ly.v[0];
  if (cfrc) {  // [14][6]: world, torso, 4 x (stub, leg, ankle) MuJoCo bodies
    for (int k = 0; k < 14 * 6; ++k) cfrc[k] = 0;
    for (int k = 0; k < 6; ++k) {
      cfrc[6 + k] = cf0[k].v[0];
      cfrc[k] -= cf0[k].v[0];
      for (int l = 0; l < 4; ++l) {
        for (int b = 0; b < 3; ++b) {
          cfrc[(2 + 3 * l + b) * 6 + k] = cf[b][k].v[l];
          cfrc[k] -= cf[b][k].v[l];
        }
      }
    }
  }
}

extern "C" {
void ant_host_step(const double* q, const double* v, const double* warm, const double* ctrl,
                   int nsub, int use_float, double* qo, double* vo, double* wo, double* lag,
                   int* iters) {
  if (use_float) Run<float>(q, v, warm, ctrl, nsub, 0, qo, vo, wo, lag, nullptr, iters);
  else Run<double>(q, v, warm, ctrl, nsub, 0, qo, vo, wo, lag, nullptr, iters);
}
void ant_host_step_wrench(const double* q, const double* v, const double* warm,
                          const double* ctrl, int nsub, double* qo, double* vo, double* wo,
                          double* lag, double* cfrc) {
  int iters = 0;
  Run<double>(q, v, warm, ctrl, nsub, 1, qo, vo, wo, lag, cfrc, &iters);
}
int ant_host_symmetric() { return A4::CheckLegSymmetry(BuildAntModel()) ? 1 : 0; }
}
