// TEST HARNESS (not product): host instantiation of mj_pusher.hip.h for diffing against
// oracle/mjcpu on a CPU box.  Not linked by envpool_amd/.
// g_busy emulates the device situation "another lane of my wave still iterates / has a contact":
// every WaveAny() is true, so this env runs all solver iterations with frozen iterates and takes
// the contact-row code with all-inactive rows -- results must not change.
static bool g_busy = false;
#define EPA_HOST_WAVE_ANY(x) ((x) || g_busy)
#include "../../envpool_amd/csrc/mj_pusher_model.h"

using epa::mj::SolverCfg;
using namespace epa::mj::pusher;

extern "C" {
void pusher_host_set_busy(int on) { g_busy = on != 0; }
// q, v, warm: 9 (arm 7, obj_slidey, obj_slidex); ctrl 7; lag out: tips xyz, object xy
void pusher_host_step(const double* q, const double* v, const double* warm, const double* ctrl,
                      int nsub, int v5, double* qo, double* vo, double* wo, double* lag,
                      int* iters) {
  const PusherModel<double> m = BuildPusherModel(v5 != 0);
  SolverCfg<double> cfg{50, 1e-13};
  double tq[kNV], tv[kNV], tw[kNV];
  for (int i = 0; i < kNV; ++i) {
    tq[i] = q[i];
    tv[i] = v[i];
    tw[i] = warm[i];
  }
  PusherLag<double> lg{};
  int it = 0;
  double row_lds[kRowSlots];
  for (int s = 0; s < nsub; ++s) {
    it += PusherStep(m, cfg, tq, tv, tw, ctrl, &lg, [&](int slot) -> double& { return row_lds[slot]; });
  }
  for (int i = 0; i < kNV; ++i) {
    qo[i] = tq[i];
    vo[i] = tv[i];
    wo[i] = tw[i];
  }
  for (int k = 0; k < 3; ++k) lag[k] = lg.tips[k];
  lag[3] = lg.obj[0];
  lag[4] = lg.obj[1];
  *iters = it;
}
// the product's capsule - cylinder rule on raw geometry: out = dist, pos(3), normal(3)
void pusher_host_capcyl(const double* p0, const double* p1, double rc, const double* c, double R, double H,
                        double* out7) {
  using V = epa::mj::ant::Vec3<double>;
  const CapCyl<double> r = CapsuleCylinder<double>(V{p0[0], p0[1], p0[2]}, V{p1[0], p1[1], p1[2]}, rc,
                                                   V{c[0], c[1], c[2]}, R, H);
  out7[0] = r.dist;
  out7[1] = r.pos.x; out7[2] = r.pos.y; out7[3] = r.pos.z;
  out7[4] = r.n.x; out7[5] = r.n.y; out7[6] = r.n.z;
}
// [mass(7) dof_invw(7) wrist_invw obj_invw obj_mass]
void pusher_host_model(int v5, double* out) {
  const PusherModel<double> m = BuildPusherModel(v5 != 0);
  int k = 0;
  for (int l = 0; l < kNL; ++l) out[k++] = m.mass[l];
  for (int l = 0; l < kNL; ++l) out[k++] = m.dof_invw[l];
  out[k++] = m.wrist_invw;
  out[k++] = m.obj_invw;
  out[k++] = m.obj_mass;
}
}
