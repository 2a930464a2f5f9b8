// TEST HARNESS (not product): host fp64/fp32 instantiation of mj_ant.cuh for
// diffing against oracle/mjcpu on a CPU box.  Not linked by envpool_amd/.
#include "../../envpool_amd/csrc/mj_ant_model.h"

using epa::mj::SolverCfg;
using namespace epa::mj::ant;

template <typename T>
static void Run(const double* q, const double* v, const double* warm, const double* ctrl,
                int nsub, double* qo, double* vo, double* wo, double* lag, int* iters) {
  AntModel<T> m = CastAntModel<T>(BuildAntModel());
  SolverCfg<T> cfg{sizeof(T) == 4 ? 12 : 50, sizeof(T) == 4 ? T(1e-6) : T(1e-13)};
  T tq[kNQ], tv[kNV], tw[kNV], tc[kNU], lx = 0, ly = 0;
  for (int i = 0; i < kNQ; ++i) tq[i] = (T)q[i];
  for (int i = 0; i < kNV; ++i) {
    tv[i] = (T)v[i];
    tw[i] = (T)warm[i];
  }
  for (int i = 0; i < kNU; ++i) tc[i] = (T)(ctrl[i] < -1 ? -1 : (ctrl[i] > 1 ? 1 : ctrl[i]));
  int it = 0;
  T lds_block[kAntLdsSlots];
  auto lds = [&](int slot) -> T& { return lds_block[slot]; };
  for (int s = 0; s < nsub; ++s) it += AntStep(m, cfg, tq, tv, tw, tc, &lx, &ly, lds);
  for (int i = 0; i < kNQ; ++i) qo[i] = tq[i];
  for (int i = 0; i < kNV; ++i) {
    vo[i] = tv[i];
    wo[i] = tw[i];
  }
  lag[0] = lx;
  lag[1] = ly;
  *iters = it;
}

extern "C" {
void ant_host_step(const double* q, const double* v, const double* warm, const double* ctrl,
                   int nsub, int use_float, double* qo, double* vo, double* wo, double* lag,
                   int* iters) {
  if (use_float) Run<float>(q, v, warm, ctrl, nsub, qo, vo, wo, lag, iters);
  else Run<double>(q, v, warm, ctrl, nsub, qo, vo, wo, lag, iters);
}
// [mass(9) dof_invw(8) geom_body_invw(13) total_mass]
void ant_host_model(double* out) {
  AntModel<double> m = BuildAntModel();
  int k = 0;
  for (int b = 0; b < kNB; ++b) out[k++] = m.mass[b];
  for (int j = 0; j < kNU; ++j) out[k++] = m.dof_invw[j];
  for (int g = 0; g < kNGeomBody; ++g) out[k++] = m.geom_body_invw[g];
  out[k++] = m.total_mass;
}
}
