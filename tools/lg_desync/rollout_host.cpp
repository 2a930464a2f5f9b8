// HOST EXPERIMENT (not product, not a test): rolls HalfCheetah envs forward with the host instantiation of the
// lane-group step (envpool_amd/csrc/mj_planar_lg.hip.h, one env at a time) under random actions and records, per
// env-step, env and mj_step: the passes over the rows, the Newton trips and the line-search evaluations of every trip.
// tools/lg_desync_sim.py replays those counts through wave-scheduling policies.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static int g_rows, g_trips;
static int g_evals[64];
static unsigned g_own[2];
#define EPA_LG_HOST_OWN(own) (g_own[0] = (own).v[0], g_own[1] = (own).v[1])
#define EPA_LG_HOST_ROWS() (++g_rows)
#define EPA_LG_HOST_TRIP(evals) (g_evals[g_trips < 64 ? g_trips++ : 63] = (evals))
#include "../../envpool_amd/csrc/mj_cheetah_model.h"
#include "../../envpool_amd/csrc/mj_planar_lg.hip.h"

using namespace epa::mj;

struct HostCx {
  using V = plg::LV<double, 2>;
  const double* tab;
  V lds[plg::LdsSlots<2>()];
  V C(int id) const {
    V r;
    for (int c = 0; c < 2; ++c) r.v[c] = tab[id * 2 + c];
    return r;
  }
  V& Lds(int slot) { return lds[slot]; }
  V LdsL(const plg::LU<2>& s, int mul, int add) const {
    V r;
    for (int c = 0; c < 2; ++c) r.v[c] = lds[s.v[c] * mul + add].v[c];
    return r;
  }
  void LdsLStore(const plg::LU<2>& s, int mul, int add, const V& x, plg::LB<2> on) {
    for (int c = 0; c < 2; ++c) {
      if (on.v[c]) lds[s.v[c] * mul + add].v[c] = x.v[c];
    }
  }
  V CL(int base, const plg::LU<2>& idx) const {
    V r;
    for (int c = 0; c < 2; ++c) r.v[c] = tab[(base + idx.v[c]) * 2 + c];
    return r;
  }
  void Refresh() {}
};

// out: [steps][envs][5][2 + 12] int16: rows, trips, evals of trips 0..11
// out: [steps][envs][5][2 + 12] int16: rows, trips, evals of trips 0..11; own (may be null): [steps][envs][5][2] the
// touching end-sphere slots of the two lanes (bit s)
extern "C" int lg_rollout2(int envs, int warm_steps, int steps, unsigned seed, int16_t* out, uint16_t* own);
extern "C" int lg_rollout(int envs, int warm_steps, int steps, unsigned seed, int16_t* out) {
  return lg_rollout2(envs, warm_steps, steps, seed, out, nullptr);
}
extern "C" int lg_rollout2(int envs, int warm_steps, int steps, unsigned seed, int16_t* out, uint16_t* own) {
  using V = plg::LV<double, 2>;
  const CheetahModel<double> m = BuildCheetahModel();
  double tab[plg::Tab<2>::kSize];
  plg::BuildTable<2>(m, tab);
  plg::SolverCfgLg<double> cfg{50, 1e-13};
  std::mt19937_64 gen(seed);
  std::uniform_real_distribution<double> un(-1, 1), noise(-0.1, 0.1);
  std::normal_distribution<double> nrm(0, 0.1);
  for (int e = 0; e < envs; ++e) {
    HostCx cx;
    cx.tab = tab;
    V q[plg::kLV], v[plg::kLV], w[plg::kLV], c[3];
    double tq[3], tv[3];
    for (int i = 0; i < 3; ++i) tq[i] = noise(gen), tv[i] = nrm(gen);
    for (int i = 0; i < plg::kLV; ++i) {
      for (int l = 0; l < 2; ++l) {
        q[i].v[l] = i < 3 ? tq[i] : noise(gen);
        v[i].v[l] = i < 3 ? tv[i] : nrm(gen);
        w[i].v[l] = 0;
      }
    }
    for (int t = 0; t < warm_steps + steps; ++t) {
      for (int k = 0; k < 3; ++k) c[k].v[0] = un(gen), c[k].v[1] = un(gen);
      for (int l = 0; l < 2; ++l) q[0].v[l] = 0;
      for (int s = 0; s < 5; ++s) {
        g_rows = g_trips = 0;
        plg::StepEuler<2>(m, cfg, cx, q, v, w, c);
        if (t >= warm_steps) {
          int16_t* o = out + (((size_t)(t - warm_steps) * envs + e) * 5 + s) * 14;
          o[0] = (int16_t)g_rows;
          o[1] = (int16_t)g_trips;
          for (int k = 0; k < 12; ++k) o[2 + k] = k < g_trips ? (int16_t)g_evals[k] : 0;
          if (own) {
            uint16_t* w = own + (((size_t)(t - warm_steps) * envs + e) * 5 + s) * 2;
            w[0] = (uint16_t)g_own[0];
            w[1] = (uint16_t)g_own[1];
          }
        }
      }
    }
  }
  return 0;
}
