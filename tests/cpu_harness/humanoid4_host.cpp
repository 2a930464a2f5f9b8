// TEST HARNESS (not product): host instantiation of mj_hum4.hip.h -- the four-lanes-per-env
// Humanoid kernel source with a lane quad emulated by Q4<double> -- next to the one-env-per-lane
// mj_tree.hip.h, for diffing both against oracle/mjcpu on a CPU box.  Not linked by envpool_amd/.
#include <cstring>
#include <vector>

#include "../../envpool_amd/csrc/mj_hum4.hip.h"
#include "../../envpool_amd/csrc/mj_tree_model.h"
#include "../../envpool_amd/csrc/build/mj_humanoid_consts.inc"

namespace T = epa::mj::tree;
namespace H = epa::mj::hum4;
using epa::mj::Q4;
struct Walk { static constexpr T::TreeModel kM = kHumanoidModelConst; static constexpr int kRegRows = 12, kCacheRows = 8; static constexpr bool kStageCall = false; static constexpr int kRowCache = 0; static constexpr bool kLazyNact = true; };
struct Stand { static constexpr T::TreeModel kM = kHumanoidStandupModelConst; static constexpr int kRegRows = 16, kCacheRows = 16; static constexpr bool kStageCall = true; static constexpr int kRowCache = 2; static constexpr bool kLazyNact = false; };
// few register rows: nearly every solve takes the hybrid (overflow) form of the PGS
struct StandOv { static constexpr T::TreeModel kM = kHumanoidStandupModelConst; static constexpr int kRegRows = 8, kCacheRows = 12; static constexpr bool kStageCall = true; static constexpr int kRowCache = 1; static constexpr bool kLazyNact = true; };
struct WalkOv { static constexpr T::TreeModel kM = kHumanoidModelConst; static constexpr int kRegRows = 4, kCacheRows = 8; static constexpr bool kStageCall = false; static constexpr int kRowCache = 0; static constexpr bool kLazyNact = true; };

template <class MP>
struct HostCtx {
  using V = Q4<double>;
  using E = double;
  static constexpr H::LimbTab kTab = H::MakeLimbTab(MP::kM);
  double geo[6 * 18];
  Q4<double> rows[320][7];
  double rs[320][5];
  double rec[160][8];
  void RowPut(int r, const V* yd) {
    for (int i = 0; i < 7; ++i) rows[r][i] = yd[i];
  }
  void RowGet(int r, V* yd) const {
    for (int i = 0; i < 7; ++i) yd[i] = rows[r][i];
  }
  const typename H::Hum4<MP, HostCtx<MP>>::Tabs& T() const { return H::Hum4<MP, HostCtx<MP>>::kT; }
  void Refresh() {}
  double rk_t[37];
  Q4<double> rk_l[16];
  void RkPut(int i, double v) { rk_t[i] = v; }
  double RkGet(int i) const { return rk_t[i]; }
  void RkPutL(int i, V v) { rk_l[i] = v; }
  V RkGetL(int i) const { return rk_l[i]; }
  double stt[31];
  void SttPut(int i, double v) { stt[i] = v; }
  double SttGet(int i) const { return stt[i]; }
  double tcd[54], ltt[45], dt[9];
  void TcdPut(int i, double v) { tcd[i] = v; }
  double TcdGet(int i) const { return tcd[i]; }
  void LttPut(int i, double v) { ltt[i] = v; }
  double LttGet(int i) const { return ltt[i]; }
  void DtPut(int i, double v) { dt[i] = v; }
  double DtGet(int i) const { return dt[i]; }
  double sh[512];
  void ShPut(int slot, double v) { sh[slot] = v; }
  double ShGet(int slot) const { return sh[slot]; }
  void RsPut(int r, int k, double v) { rs[r][k] = v; }
  void RsPut4(int r, double arr, double R, double b, double ainv) {
    rs[r][1] = arr;
    rs[r][2] = R;
    rs[r][3] = b;
    rs[r][4] = ainv;
  }
  void RsGet4(int r, double* arr, double* R, double* b, double* ainv) const {
    *arr = rs[r][1];
    *R = rs[r][2];
    *b = rs[r][3];
    *ainv = rs[r][4];
  }
  V RsGetLane(int r0, int k) const {  // scalar k of row r0 + lane
    V x;
    for (int l = 0; l < 4; ++l) x.v[l] = rs[r0 + l][k];
    return x;
  }
  V ShGetTriLane(int r0, int cc, int base) const {  // entry (r0 + lane, cc) of the packed symmetric matrix
    V x;
    for (int l = 0; l < 4; ++l) {
      const int r = r0 + l;
      x.v[l] = sh[(r >= cc ? r * (r + 1) / 2 + cc : cc * (cc + 1) / 2 + r) - base];
    }
    return x;
  }
  Q4<double> ov[64][8];
  void OvPut(int o, int k, V v) { ov[o][k] = v; }
  V OvGet(int o, int k) const { return ov[o][k]; }
  struct TriBase { int r0; };
  TriBase TriRow(int r0) const { return {r0}; }
  template <int R0, int CC, int BASE>
  V ShGetTriRow(TriBase) const { return ShGetTriLane(R0, CC, BASE); }
  V RowIndexLane(int r0) const {
    V x;
    for (int l = 0; l < 4; ++l) x.v[l] = r0 + l;
    return x;
  }
  double RsGet(int r, int k) const { return rs[r][k]; }
  void RecPut(int t, int k, double v) { rec[t][k] = v; }
  double RecGet(int t, int k) const { return rec[t][k]; }
  void GeoGetLimb(int which, H::Vec3<V>* pos, H::Vec3<V>* axis) const {
    for (int l = 0; l < 4; ++l) {
      const int g = (l < 2 ? 6 + 3 * l : 12 + 3 * (l - 2)) + which;
      pos->x.v[l] = geo[6 * g];
      pos->y.v[l] = geo[6 * g + 1];
      pos->z.v[l] = geo[6 * g + 2];
      axis->x.v[l] = geo[6 * g + 3];
      axis->y.v[l] = geo[6 * g + 4];
      axis->z.v[l] = geo[6 * g + 5];
    }
  }
  V LC(int idx) const {
    V r;
    for (int l = 0; l < 4; ++l) r.v[l] = kTab.c[idx][l];
    return r;
  }
  void GeoPut(int slot, double v) { geo[slot] = v; }
  double GeoGet(int slot) const { return geo[slot]; }
  void GeoPutLimb(int which, H::Vec3<V> pos, H::Vec3<V> axis) {
    for (int l = 0; l < 4; ++l) {
      const int g = (l < 2 ? 6 + 3 * l : 12 + 3 * (l - 2)) + which;
      geo[6 * g] = pos.x.v[l];
      geo[6 * g + 1] = pos.y.v[l];
      geo[6 * g + 2] = pos.z.v[l];
      geo[6 * g + 3] = axis.x.v[l];
      geo[6 * g + 4] = axis.y.v[l];
      geo[6 * g + 5] = axis.z.v[l];
    }
  }
};

// distribute a 23-vector (dof order) / the 24 qpos over trunk + limb slots
static void SplitV(const double* x, double* xt, Q4<double>* xl) {
  for (int j = 0; j < 9; ++j) xt[j] = x[j];
  for (int s = 0; s < 4; ++s) {
    for (int l = 0; l < 4; ++l) {
      const int d = H::LimbDof(l, s);
      xl[s].v[l] = d < 0 ? 0.0 : x[d];
    }
  }
}
static void JoinV(const double* xt, const Q4<double>* xl, double* x) {
  for (int j = 0; j < 9; ++j) x[j] = xt[j];
  for (int s = 0; s < 4; ++s) {
    for (int l = 0; l < 4; ++l) {
      const int d = H::LimbDof(l, s);
      if (d >= 0) x[d] = xl[s].v[l];
    }
  }
}

template <class MP>
static void Smooth(const double* q, const double* v, const double* ctrl, double* out) {
  using C = HostCtx<MP>;
  using Eng = H::Hum4<MP, C>;
  static C c;
  H::Fwd<Q4<double>> f;
  double qt[10], vt[9], ut[3];
  Q4<double> ql[4], vl[4], ul[4];
  for (int i = 0; i < 10; ++i) qt[i] = q[i];
  double qd[23], ud[23] = {0};
  for (int d = 0; d < 23; ++d) qd[d] = d < 6 ? 0.0 : q[d + 1];
  for (int u = 0; u < 17; ++u) ud[MP::kM.act_dof[u]] = ctrl[u];
  double tmp[9];
  SplitV(qd, tmp, ql);
  SplitV(v, vt, vl);
  SplitV(ud, tmp, ul);
  for (int i = 0; i < 3; ++i) ut[i] = ud[6 + i];
  Eng::Position(c, qt, ql, f);
  Eng::Velocity(c, qt, ql, vt, vl, ut, ul, f);
  Eng::MassFactor(c, f);
  Eng::SmoothAcc(c, f);
  int k = 0;
  double a[23];
  JoinV(f.accs_t, f.accs_l, a);
  for (int i = 0; i < 23; ++i) out[k++] = a[i];
  out[k++] = f.com.x;
  out[k++] = f.com.y;
  out[k++] = f.com.z;
  // cinert [14][10], cvel [14][6] in body order
  for (int b = 0; b < 14; ++b) {
    for (int i = 0; i < 10; ++i) {
      double x = 0;
      if (b >= 1 && b <= 3) x = f.tci[b - 1].v[i];
      if (b >= 4) {
        const int l = b < 7 ? 0 : (b < 10 ? 1 : (b < 12 ? 2 : 3));
        x = f.lci[b - H::kLimbA[l]].v[i].v[l];
      }
      out[k++] = x;
    }
  }
  for (int b = 0; b < 14; ++b) {
    for (int i = 0; i < 6; ++i) {
      double x = 0;
      auto pick = [&](const auto& s6, int idx) { return idx < 3 ? (idx == 0 ? s6.a.x : idx == 1 ? s6.a.y : s6.a.z)
                                                                : (idx == 3 ? s6.l.x : idx == 4 ? s6.l.y : s6.l.z); };
      if (b >= 1 && b <= 3) x = pick(f.tcv[b - 1], i);
      if (b >= 4) {
        const int l = b < 7 ? 0 : (b < 10 ? 1 : (b < 12 ? 2 : 3));
        const int w = b - H::kLimbA[l];
        x = pick(f.lcv[w > 1 ? 1 : w], i).v[l];
      }
      out[k++] = x;
    }
  }
  JoinV(f.act_t, f.act_l, a);
  for (int i = 0; i < 23; ++i) out[k++] = a[i];
  for (int i = 0; i < 6 * 18; ++i) out[k++] = c.geo[i];
}

// the same quantities from mj_tree.hip.h
template <class MP>
static void SmoothTree(const double* q, const double* v, const double* ctrl, double* out) {
  using E = T::Tree<MP>;
  std::vector<double> buf(E::kL.total, 0.0);
  T::Ws w{buf.data(), 0u};
  for (int i = 0; i < E::NQ; ++i) w(E::kL.qpos + i) = q[i];
  for (int i = 0; i < E::NV; ++i) w(E::kL.qvel + i) = v[i];
  for (int i = 0; i < E::NU; ++i) w(E::kL.ctrl + i) = ctrl[i];
  E::Kinematics(w);
  E::ComPos(w);
  E::CrbFactor(w);
  E::Velocity(w);
  int k = 0;
  for (int i = 0; i < 23; ++i) out[k++] = w(E::kL.accs + i);
  for (int i = 0; i < 3; ++i) out[k++] = w(E::kL.com + i);
  for (int i = 0; i < 140; ++i) out[k++] = i < 10 ? 0.0 : w(E::kL.cinert + i);
  for (int i = 0; i < 84; ++i) out[k++] = w(E::kL.cvel + i);
  for (int i = 0; i < 23; ++i) out[k++] = w(E::kL.act + i);
  for (int g = 0; g < 18; ++g) {
    for (int i = 0; i < 3; ++i) out[k++] = g == 0 ? 0.0 : w(E::kL.gpos + 3 * g + i);
    for (int i = 0; i < 3; ++i) out[k++] = (g == 0 || MP::kM.geom_type[g] != T::kGeomCapsule) ? 0.0 : w(E::kL.gaxis + 3 * g + i);
  }
}

// the layout of humanoid_host.cpp's Run(): qpos[24] qvel[23] warm[23] | cinert[14*10] cvel[14*6]
// qfrc_actuator[23] cfrc_ext[14*6] | mass centre x y | active groups of the last forward pass
template <class MP>
static void Step4(const double* q, const double* v, const double* warm, const double* ctrl, int nsub,
                  int post_constraint, double* out) {
  using C = HostCtx<MP>;
  using Eng = H::Hum4<MP, C>;
  static C c;  // (large)
  H::Fwd<Q4<double>> f;
  typename Eng::State s;
  for (int i = 0; i < 10; ++i) s.qt[i] = q[i];
  double qd[23], ud[23] = {0}, tmp[9];
  for (int d = 0; d < 23; ++d) qd[d] = d < 6 ? 0.0 : q[d + 1];
  for (int u = 0; u < 17; ++u) ud[MP::kM.act_dof[u]] = ctrl[u];
  SplitV(qd, tmp, s.ql);
  SplitV(v, s.vt, s.vl);
  SplitV(warm, s.wt, s.wl);
  SplitV(ud, tmp, s.ul);
  for (int i = 0; i < 3; ++i) s.ut[i] = ud[6 + i];
  Eng::StoreTrunk(c, s, 15);
  double at[9];
  Q4<double> al[4];
  typename Eng::RowCount rc{0, 0, 0};
  int stat[H::kNStat] = {};
  if (nsub == 0) rc = Eng::Forward(c, s, f, true, at, al, 0, [](const H::Fwd<Q4<double>>&) {}, stat);
  for (int k = 0; k < nsub; ++k) {
    for (int stage = 0; stage < 4; ++stage) {
      rc = Eng::Forward(c, s, f, true, at, al, 0, [](const H::Fwd<Q4<double>>&) {}, stat);
      Eng::RkAdvance(c, s, stage, true, at, al);
    }
  }
  Eng::LoadTrunk(c, s, 7);
  H::Sp6<double> ext_t[4];
  H::Sp6<Q4<double>> ext_l[3];
  if (post_constraint) Eng::ContactWrench(c, f, rc, ext_t, ext_l);
  int k = 0;
  for (int i = 0; i < 7; ++i) out[k++] = s.qt[i];
  double x[23];
  for (int d = 0; d < 6; ++d) x[d] = 0;
  {
    double xt[9];
    for (int j = 0; j < 9; ++j) xt[j] = j < 6 ? 0.0 : s.qt[1 + j];
    JoinV(xt, s.ql, x);
  }
  for (int d = 6; d < 23; ++d) out[k++] = x[d];
  JoinV(s.vt, s.vl, x);
  for (int d = 0; d < 23; ++d) out[k++] = x[d];
  JoinV(s.wt, s.wl, x);
  for (int d = 0; d < 23; ++d) out[k++] = x[d];
  auto limb_of = [](int b) { return b < 7 ? 0 : (b < 10 ? 1 : (b < 12 ? 2 : 3)); };
  auto pick = [&](const auto& s6, int idx) { return idx < 3 ? (idx == 0 ? s6.a.x : idx == 1 ? s6.a.y : s6.a.z)
                                                            : (idx == 3 ? s6.l.x : idx == 4 ? s6.l.y : s6.l.z); };
  for (int b = 0; b < 14; ++b) {
    for (int i = 0; i < 10; ++i) {
      double xx = 0;
      if (b >= 1 && b <= 3) xx = f.tci[b - 1].v[i];
      if (b >= 4) xx = f.lci[b - H::kLimbA[limb_of(b)]].v[i].v[limb_of(b)];
      out[k++] = xx;
    }
  }
  for (int b = 0; b < 14; ++b) {
    for (int i = 0; i < 6; ++i) {
      double xx = 0;
      if (b >= 1 && b <= 3) xx = pick(f.tcv[b - 1], i);
      if (b >= 4) {
        const int w = b - H::kLimbA[limb_of(b)];
        xx = pick(f.lcv[w > 1 ? 1 : w], i).v[limb_of(b)];
      }
      out[k++] = xx;
    }
  }
  JoinV(f.act_t, f.act_l, x);
  for (int d = 0; d < 23; ++d) out[k++] = x[d];
  for (int b = 0; b < 14; ++b) {
    for (int i = 0; i < 6; ++i) {
      double xx = 0;
      if (post_constraint && b <= 3) xx = pick(ext_t[b], i);
      if (post_constraint && b >= 4) xx = pick(ext_l[b - H::kLimbA[limb_of(b)]], i).v[limb_of(b)];
      out[k++] = xx;
    }
  }
  out[k++] = f.com.x;
  out[k++] = f.com.y;
  out[k++] = rc.nl + rc.nf + rc.np;
  out[k++] = stat[3];  // solves that took the hybrid / streaming form
  out[k++] = rc.rows();
  out[k++] = stat[13];  // solves that started over in the exact form (the cost check of a visit fired)
}

extern "C" {
void humanoid4_host_step(const double* q, const double* v, const double* warm, const double* ctrl,
                         int nsub, int standup, int post_constraint, double* out) {
  if (standup == 1) Step4<Stand>(q, v, warm, ctrl, nsub, post_constraint, out);
  else if (standup == 3) Step4<StandOv>(q, v, warm, ctrl, nsub, post_constraint, out);
  else if (standup == 2) Step4<WalkOv>(q, v, warm, ctrl, nsub, post_constraint, out);
  else Step4<Walk>(q, v, warm, ctrl, nsub, post_constraint, out);
}
// out: accs[23] com[3] cinert[140] cvel[84] act[23] geoms[108] = 381
void hum4_smooth(const double* q, const double* v, const double* ctrl, int standup, int use_tree, double* out) {
  if (use_tree) {
    if (standup) SmoothTree<Stand>(q, v, ctrl, out);
    else SmoothTree<Walk>(q, v, ctrl, out);
  } else {
    if (standup) Smooth<Stand>(q, v, ctrl, out);
    else Smooth<Walk>(q, v, ctrl, out);
  }
}
}
