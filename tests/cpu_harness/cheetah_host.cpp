// TEST HARNESS (not product): compiles the host instantiation of the planar
// HalfCheetah step template (envpool_amd/csrc/mj_cheetah.hip.h, EPA_HD) with g++
// so the exact kernel source can be diffed against oracle/mjcpu on a CPU box.
// Nothing in envpool_amd/ links or loads this.
#include <cstring>
#include <vector>

#include "../../envpool_amd/csrc/mj_cheetah_model.h"

using namespace epa::mj;

template <typename T>
static void Run(const double* q, const double* v, const double* warm,
                const double* ctrl, int nsub, double* qo, double* vo,
                double* wo, int* iters) {
  CheetahModel<double> md = BuildCheetahModel();
  CheetahModel<T> m = CastCheetahModel<T>(md);
  SolverCfg<T> cfg{sizeof(T) == 4 ? 12 : 50, sizeof(T) == 4 ? T(1e-6) : T(1e-13)};
  T tq[kNV], tv[kNV], tw[kNV], tc[kNU];
  double x0 = q[0];
  for (int i = 0; i < kNV; ++i) {
    tq[i] = (T)q[i];
    tv[i] = (T)v[i];
    tw[i] = (T)warm[i];
  }
  tq[0] = 0;  // local x
  for (int i = 0; i < kNU; ++i) {
    double c = ctrl[i] < -1 ? -1 : (ctrl[i] > 1 ? 1 : ctrl[i]);
    tc[i] = (T)c;
  }
  T lds[kLdsSlots];
  int it = 0;
  for (int s = 0; s < nsub; ++s) {
    it += CheetahStep(m, cfg, tq, tv, tw, tc, [&](int slot) -> T& { return lds[slot]; });
  }
  for (int i = 0; i < kNV; ++i) {
    qo[i] = tq[i];
    vo[i] = tv[i];
    wo[i] = tw[i];
  }
  qo[0] += x0;
  *iters = it;
}

// Walker2d on the same planar code: sign-mirrored leg hinges, RK4.
template <typename T>
static void RunWalker(const double* q, const double* v, const double* warm,
                      const double* ctrl, int nsub, int v5, double* qo, double* vo,
                      double* wo, int* iters) {
  CheetahModel<T> m = CastCheetahModel<T>(BuildWalkerModel(v5 != 0));
  SolverCfg<T> cfg{sizeof(T) == 4 ? 12 : 50, sizeof(T) == 4 ? T(1e-6) : T(1e-13)};
  T tq[kNV], tv[kNV], tw[kNV], tc[kNU];
  double x0 = q[0];
  for (int i = 0; i < kNV; ++i) {
    const int sg = PlanarDofSign(kPlanarWalker, i);
    tq[i] = (T)(sg * q[i]);
    tv[i] = (T)(sg * v[i]);
    tw[i] = (T)(sg * warm[i]);
  }
  tq[0] = 0;
  for (int i = 0; i < kNU; ++i) {
    double c = ctrl[i] < -1 ? -1 : (ctrl[i] > 1 ? 1 : ctrl[i]);
    tc[i] = (T)c;
  }
  T lds[kLdsSlots];
  int it = 0;
  for (int s = 0; s < nsub; ++s) {
    it += PlanarStepRK4(m, cfg, tq, tv, tw, tc, [&](int slot) -> T& { return lds[slot]; });
  }
  for (int i = 0; i < kNV; ++i) {
    const int sg = PlanarDofSign(kPlanarWalker, i);
    qo[i] = sg * (double)tq[i];
    vo[i] = sg * (double)tv[i];
    wo[i] = sg * (double)tw[i];
  }
  qo[0] += x0;
  *iters = it;
}

// Hopper on the same planar code: dofs 0..5 of the model are dofs 0..5 of the tree,
// the ghost leg (dofs 6..8) stays at rest.
template <typename T>
static void RunHopper(const double* q, const double* v, const double* warm, const double* ctrl,
                      int nsub, double* qo, double* vo, double* wo, int* iters) {
  CheetahModel<T> m = CastCheetahModel<T>(BuildHopperModel());
  SolverCfg<T> cfg{sizeof(T) == 4 ? 12 : 50, sizeof(T) == 4 ? T(1e-6) : T(1e-13)};
  T tq[kNV] = {0}, tv[kNV] = {0}, tw[kNV] = {0}, tc[kNU] = {0};
  double x0 = q[0];
  for (int i = 0; i < 6; ++i) {
    const int sg = PlanarDofSign(kPlanarHopper, i);
    tq[i] = (T)(sg * q[i]);
    tv[i] = (T)(sg * v[i]);
    tw[i] = (T)(sg * warm[i]);
  }
  tq[0] = 0;
  for (int i = 0; i < 3; ++i) {
    double c = ctrl[i] < -1 ? -1 : (ctrl[i] > 1 ? 1 : ctrl[i]);
    tc[i] = (T)c;
  }
  T lds[kLdsSlots];
  int it = 0;
  for (int s = 0; s < nsub; ++s) {
    it += PlanarStepRK4(m, cfg, tq, tv, tw, tc, [&](int slot) -> T& { return lds[slot]; });
  }
  for (int i = 0; i < 6; ++i) {
    const int sg = PlanarDofSign(kPlanarHopper, i);
    qo[i] = sg * (double)tq[i];
    vo[i] = sg * (double)tv[i];
    wo[i] = sg * (double)tw[i];
  }
  qo[0] += x0;
  // the ghost leg must not have moved
  for (int i = 6; i < kNV; ++i) {
    if (tq[i] != T(0) || tv[i] != T(0)) it = -1000000;
  }
  *iters = it;
}

extern "C" {
void hopper_host_step(const double* q, const double* v, const double* warm, const double* ctrl,
                      int nsub, int use_float, double* qo, double* vo, double* wo, int* iters) {
  if (use_float) {
    RunHopper<float>(q, v, warm, ctrl, nsub, qo, vo, wo, iters);
  } else {
    RunHopper<double>(q, v, warm, ctrl, nsub, qo, vo, wo, iters);
  }
}
// [mass(4) dof_invw(3) body_invw(4)]
void hopper_host_model(double* out) {
  CheetahModel<double> m = BuildHopperModel();
  int k = 0;
  for (int b = 0; b < 4; ++b) out[k++] = m.mass[b];
  for (int j = 0; j < 3; ++j) out[k++] = m.dof_invw[j];
  for (int b = 0; b < 4; ++b) out[k++] = m.body_invw[b];
}
void walker_host_step(const double* q, const double* v, const double* warm,
                      const double* ctrl, int nsub, int v5, int use_float, double* qo,
                      double* vo, double* wo, int* iters) {
  if (use_float) {
    RunWalker<float>(q, v, warm, ctrl, nsub, v5, qo, vo, wo, iters);
  } else {
    RunWalker<double>(q, v, warm, ctrl, nsub, v5, qo, vo, wo, iters);
  }
}
// [mass(7) iyy(7) dof_invw(6) body_invw(7)]
void walker_host_model(int v5, double* out) {
  CheetahModel<double> m = BuildWalkerModel(v5 != 0);
  int k = 0;
  for (int b = 0; b < kNB; ++b) out[k++] = m.mass[b];
  for (int b = 0; b < kNB; ++b) out[k++] = m.iyy[b];
  for (int j = 0; j < kNU; ++j) out[k++] = m.dof_invw[j];
  for (int b = 0; b < kNB; ++b) out[k++] = m.body_invw[b];
}
void cheetah_host_step(const double* q, const double* v, const double* warm,
                       const double* ctrl, int nsub, int use_float, double* qo,
                       double* vo, double* wo, int* iters) {
  if (use_float) {
    Run<float>(q, v, warm, ctrl, nsub, qo, vo, wo, iters);
  } else {
    Run<double>(q, v, warm, ctrl, nsub, qo, vo, wo, iters);
  }
}
// [mass(7) iyy(7) dof_invw(6) body_invw(7)]
void cheetah_host_model(double* out) {
  CheetahModel<double> m = BuildCheetahModel();
  int k = 0;
  for (int b = 0; b < kNB; ++b) out[k++] = m.mass[b];
  for (int b = 0; b < kNB; ++b) out[k++] = m.iyy[b];
  for (int j = 0; j < kNU; ++j) out[k++] = m.dof_invw[j];
  for (int b = 0; b < kNB; ++b) out[k++] = m.body_invw[b];
}
}
