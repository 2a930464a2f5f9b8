// TEST HARNESS (not product): host instantiation of mj_pendulum.hip.h for diffing
// against oracle/mjcpu on a CPU box.  Not linked by envpool_amd/.
#include "../../envpool_amd/csrc/mj_pendulum_model.h"

using epa::mj::SolverCfg;
using namespace epa::mj::pend;

template <int NL, int kBase>
static void Run(const PendModel<double, NL, kBase>& m, const double* q, const double* v,
                const double* warm, const double* ctrl, int nsub, double* qo, double* vo,
                double* wo, double* aux_out, int* iters) {
  constexpr int NV = NL + kBase;
  SolverCfg<double> cfg{50, 1e-13};
  double tq[NV], tv[NV], tw[NV];
  for (int i = 0; i < NV; ++i) {
    tq[i] = q[i];
    tv[i] = v[i];
    tw[i] = warm[i];
  }
  PendAux<double, NL> aux{};
  int it = 0;
  for (int s = 0; s < nsub; ++s) it += PendStepRK4(m, cfg, tq, tv, tw, ctrl, aux);
  for (int i = 0; i < NV; ++i) {
    qo[i] = tq[i];
    vo[i] = tv[i];
    wo[i] = tw[i];
    aux_out[2 + i] = aux.qfrc_constraint[i];
  }
  aux_out[0] = aux.tip_x;
  aux_out[1] = aux.tip_z;
  *iters = it;
}

extern "C" {
// nl = 1: InvertedPendulum, nl = 2: InvertedDoublePendulum (ctrl acts on dof 0)
void pendulum_host_step(int nl, const double* q, const double* v, const double* warm,
                        double ctrl, int nsub, double* qo, double* vo, double* wo,
                        double* aux_out, int* iters) {
  const double c[3] = {ctrl, 0, 0};
  if (nl == 1) {
    Run<1, kBaseCart>(BuildInvertedPendulum(), q, v, warm, c, nsub, qo, vo, wo, aux_out, iters);
  } else {
    Run<2, kBaseCart>(BuildInvertedDoublePendulum(), q, v, warm, c, nsub, qo, vo, wo, aux_out, iters);
  }
}
// Reacher arm: q, v, warm, ctrl have 2 entries; aux_out[0..1] = fingertip (x, z = -y)
void reacher_host_step(const double* q, const double* v, const double* warm, const double* ctrl,
                       int nsub, double* qo, double* vo, double* wo, double* aux_out,
                       int* iters) {
  Run<2, kBaseFixed>(BuildReacher(), q, v, warm, ctrl, nsub, qo, vo, wo, aux_out, iters);
}
// Swimmer: 5 dofs in the kernel's coordinates (slide y already mirrored by the caller)
void swimmer_host_step(const double* q, const double* v, const double* warm, const double* ctrl,
                       int nsub, double* qo, double* vo, double* wo, double* aux_out,
                       int* iters) {
  Run<3, kBaseFree>(BuildSwimmer(), q, v, warm, ctrl, nsub, qo, vo, wo, aux_out, iters);
}
void swimmer_host_model(double* out) {  // [total_mass, dof_invw x5]
  auto m = BuildSwimmer();
  out[0] = m.total_mass;
  for (int j = 0; j < 5; ++j) out[1 + j] = m.dof_invw[j];
}
void reacher_host_model(double* out) {  // [total_mass, dof_invw0, dof_invw1]
  auto m = BuildReacher();
  out[0] = m.total_mass;
  out[1] = m.dof_invw[0];
  out[2] = m.dof_invw[1];
}
// [total_mass, dof_invw...]
void pendulum_host_model(int nl, double* out) {
  if (nl == 1) {
    auto m = BuildInvertedPendulum();
    out[0] = m.total_mass;
    for (int j = 0; j < 2; ++j) out[1 + j] = m.dof_invw[j];
  } else {
    auto m = BuildInvertedDoublePendulum();
    out[0] = m.total_mass;
    for (int j = 0; j < 3; ++j) out[1 + j] = m.dof_invw[j];
  }
}
}
