// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Driver of the operation-counting build of oracle/mjcpu (count_real.h): runs frame_skip mj_steps
// (+ mj_rnePostConstraint on request) from a given (qpos, qvel, qacc_warmstart, ctrl) and returns
// the counted operations per pipeline stage together with the resulting state, so that
// tools/count_flops.py can (a) feed it the states of a real rollout of the plain port and (b) check
// that the counted build computes the same bits.
#include "count_real.h"

#include "../mjcpu/mjcpu.h"

thread_local unsigned long long mjc_count[MJC_NSTAGE][MJC_NKIND];
thread_local int mjc_stage = 0;

#undef double

namespace {
struct Handle {
  mjc_model m;
  mjc_data d;
};
}  // namespace

extern "C" {

void* mjc_count_create(const char* model) {
  auto* h = new Handle();
  const char* s = model;
  if (!strcmp(s, "HalfCheetah")) mjc_build_half_cheetah(&h->m);
  else if (!strcmp(s, "Ant")) mjc_build_ant(&h->m);
  else if (!strcmp(s, "Walker2d")) mjc_build_walker2d(&h->m, 0);
  else if (!strcmp(s, "Hopper")) mjc_build_hopper(&h->m);
  else if (!strcmp(s, "Swimmer")) mjc_build_swimmer(&h->m);
  else if (!strcmp(s, "Reacher")) mjc_build_reacher(&h->m);
  else if (!strcmp(s, "Pusher")) mjc_build_pusher(&h->m, 0);
  else if (!strcmp(s, "InvertedPendulum")) mjc_build_inverted_pendulum(&h->m);
  else if (!strcmp(s, "InvertedDoublePendulum")) mjc_build_inverted_double_pendulum(&h->m);
  else if (!strcmp(s, "Humanoid")) mjc_build_humanoid(&h->m, 0);
  else if (!strcmp(s, "HumanoidStandup")) mjc_build_humanoid(&h->m, 1);
  else { delete h; return nullptr; }
  mjc_reset_data(&h->m, &h->d);
  return h;
}

void mjc_count_dims(void* hv, int* out) {
  auto* h = static_cast<Handle*>(hv);
  out[0] = h->m.nq; out[1] = h->m.nv; out[2] = h->m.nu; out[3] = h->m.nbody;
}

// state: qpos[nq] qvel[nv] qacc_warmstart[nv] (read and written back); ctrl[nu].
// counts: [MJC_NSTAGE][MJC_NKIND] accumulated over the call (caller zeroes); stats[0..2] += nefc,
// ncon, solver iterations summed over the forward evaluations, stats[3] += forward evaluations.
void mjc_count_env_step(void* hv, double* qpos, double* qvel, double* warm, const double* ctrl,
                        int frame_skip, int post_constraint, unsigned long long* counts,
                        double* stats) {
  auto* h = static_cast<Handle*>(hv);
  const mjc_model* m = &h->m;
  mjc_data* d = &h->d;
  for (int i = 0; i < m->nq; ++i) d->qpos[i] = qpos[i];
  for (int i = 0; i < m->nv; ++i) d->qvel[i] = qvel[i];
  for (int i = 0; i < m->nv; ++i) d->qacc_warmstart[i] = warm[i];
  for (int i = 0; i < m->nu; ++i) d->ctrl[i] = ctrl[i];
  memset(mjc_count, 0, sizeof(mjc_count));
  mjc_stage = 0;
  for (int s = 0; s < frame_skip; ++s) {
    mjc_step(m, d);
    stats[0] += d->nefc;  // of the last forward evaluation of the sub-step
    stats[1] += d->ncon;
    stats[2] += d->solver_iter;
    stats[3] += 1;
  }
  if (post_constraint) mjc_rne_post_constraint(m, d);
  for (int a = 0; a < MJC_NSTAGE; ++a) {
    for (int b = 0; b < MJC_NKIND; ++b) counts[a * MJC_NKIND + b] += mjc_count[a][b];
  }
  for (int i = 0; i < m->nq; ++i) qpos[i] = static_cast<double>(d->qpos[i]);
  for (int i = 0; i < m->nv; ++i) qvel[i] = static_cast<double>(d->qvel[i]);
  for (int i = 0; i < m->nv; ++i) warm[i] = static_cast<double>(d->qacc_warmstart[i]);
}

void mjc_count_destroy(void* hv) { delete static_cast<Handle*>(hv); }

}  // extern "C"
