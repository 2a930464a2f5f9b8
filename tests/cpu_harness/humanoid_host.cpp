// TEST HARNESS (not product): host instantiation of mj_tree.hip.h (lane stride 1) for diffing
// against oracle/mjcpu on a CPU box.  Not linked by envpool_amd/.
#include <vector>

#include "../../envpool_amd/csrc/mj_tree_model.h"
#include "../../envpool_amd/csrc/build/mj_humanoid_consts.inc"

namespace T = epa::mj::tree;
struct Walk { static constexpr T::TreeModel kM = kHumanoidModelConst; };
struct Stand { static constexpr T::TreeModel kM = kHumanoidStandupModelConst; };

// out: qpos[24] qvel[23] warm[23] | cinert[14*10] cvel[14*6] qfrc_actuator[23] cfrc_ext[14*6]
//      | mass centre x y | nrows active, max |f|
template <class MP>
static void Run(const double* q, const double* v, const double* warm, const double* ctrl, int nsub,
                int post_constraint, double* out) {
  using E = T::Tree<MP>;
  std::vector<double> buf(E::kL.total, 0.0);
  T::Ws w{buf.data(), 0u};
  for (int i = 0; i < E::NQ; ++i) w(E::kL.qpos + i) = q[i];
  for (int i = 0; i < E::NV; ++i) {
    w(E::kL.qvel + i) = v[i];
    w(E::kL.warm + i) = warm[i];
  }
  for (int i = 0; i < E::NU; ++i) w(E::kL.ctrl + i) = ctrl[i];
  typename E::RowCount rc{0, 0, 0};
  std::vector<double> lds(E::kArLds, 0.0);
  if (nsub == 0) rc = E::Forward(w, true, lds.data());  // mj_forward only (reset)
  for (int s = 0; s < nsub; ++s) {
    for (int stage = 0; stage < 4; ++stage) {
      rc = E::Forward(w, true, lds.data());
      E::RkAdvance(w, stage, true);
    }
  }
  if (post_constraint) E::ContactWrench(w, rc);
  int k = 0;
  for (int i = 0; i < E::NQ; ++i) out[k++] = w(E::kL.qpos + i);
  for (int i = 0; i < E::NV; ++i) out[k++] = w(E::kL.qvel + i);
  for (int i = 0; i < E::NV; ++i) out[k++] = w(E::kL.warm + i);
  for (int i = 0; i < 10; ++i) out[k++] = 0.0;
  for (int i = 10; i < 10 * E::NB; ++i) out[k++] = w(E::kL.cinert + i);
  for (int i = 0; i < 6 * E::NB; ++i) out[k++] = w(E::kL.cvel + i);
  for (int i = 0; i < E::NV; ++i) out[k++] = w(E::kL.act + i);
  for (int i = 0; i < 6 * E::NB; ++i) out[k++] = post_constraint ? w(E::kL.cext + i) : 0.0;
  double mx = 0, my = 0;
  for (int b = 1; b < E::NB; ++b) {
    mx += MP::kM.body_mass[b] * w(E::kL.xipos + 3 * b);
    my += MP::kM.body_mass[b] * w(E::kL.xipos + 3 * b + 1);
  }
  out[k++] = mx / MP::kM.total_mass;
  out[k++] = my / MP::kM.total_mass;
  out[k++] = rc.nl + rc.nf + rc.np;  // active groups of the last forward pass
}

extern "C" {
void humanoid_host_step(const double* q, const double* v, const double* warm, const double* ctrl,
                        int nsub, int standup, int post_constraint, double* out) {
  if (standup) Run<Stand>(q, v, warm, ctrl, nsub, post_constraint, out);
  else Run<Walk>(q, v, warm, ctrl, nsub, post_constraint, out);
}
// [meaninertia total_mass | body_mass[14] | dof_invw[23] | body_invw[14]]
void humanoid_host_model(int standup, double* out) {
  const T::TreeModel m = T::BuildHumanoidModel(standup != 0);
  int k = 0;
  out[k++] = m.meaninertia;
  out[k++] = m.total_mass;
  for (int b = 0; b < m.nbody; ++b) out[k++] = m.body_mass[b];
  for (int i = 0; i < m.nv; ++i) out[k++] = m.dof_invw[i];
  for (int b = 0; b < m.nbody; ++b) out[k++] = m.body_invw[b];
}
}
