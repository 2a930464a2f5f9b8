// Host-side "model compiler" for the gym Ant: MJCF numbers of
// third_party/mujoco_gym_xml_patches/ant_envpool.xml (hand transcribed, cited by
// XML line) -> the constants mj_ant.hip.h needs, i.e. what MuJoCo's compiler +
// mj_setConst produce (inertiafromgeom with density 5, body_invweight0 /
// dof_invweight0 at qpos0).  fp64; cast afterwards.
#ifndef ENVPOOL_AMD_CSRC_MJ_ANT_MODEL_H_
#define ENVPOOL_AMD_CSRC_MJ_ANT_MODEL_H_

#include <cmath>

#include "mj_ant.hip.h"

namespace epa {
namespace mj {
namespace ant {

namespace detail {
struct Cap {  // capsule fromto (0,0,0) -> to, radius r
  double to[3];
};
inline void CapsuleMassInertia(double r, double half_len, double density,
                               double* mass, double* iperp, double* iax) {
  const double kPi = 3.14159265358979323846;
  double h = 2 * half_len;
  double vol = kPi * (r * r * h + 4.0 * r * r * r / 3.0);
  *mass = density * vol;
  double sphere_mass = *mass * 4 * r / (4 * r + 3 * h);
  double cyl_mass = *mass - sphere_mass;
  *iperp = cyl_mass * (3 * r * r + h * h) / 12 + 2 * sphere_mass * r * r / 5 +
           sphere_mass * h * (3 * r + 2 * h) / 8;
  *iax = cyl_mass * r * r / 2 + 2 * sphere_mass * r * r / 5;
}
// inertia tensor (xx yy zz xy xz yz) of a capsule with unit axis u about its centre
inline void CapsuleTensor(double iperp, double iax, const double* u, double* I) {
  I[0] = iperp * (1 - u[0] * u[0]) + iax * u[0] * u[0];
  I[1] = iperp * (1 - u[1] * u[1]) + iax * u[1] * u[1];
  I[2] = iperp * (1 - u[2] * u[2]) + iax * u[2] * u[2];
  I[3] = (iax - iperp) * u[0] * u[1];
  I[4] = (iax - iperp) * u[0] * u[2];
  I[5] = (iax - iperp) * u[1] * u[2];
}
inline void AddShifted(double* I, double mass, const double* d) {  // parallel axis
  double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  I[0] += mass * (d2 - d[0] * d[0]);
  I[1] += mass * (d2 - d[1] * d[1]);
  I[2] += mass * (d2 - d[2] * d[2]);
  I[3] -= mass * d[0] * d[1];
  I[4] -= mass * d[0] * d[2];
  I[5] -= mass * d[1] * d[2];
}
}  // namespace detail

inline AntModel<double> BuildAntModel() {
  using namespace detail;
  const double kPi = 3.14159265358979323846;
  const double deg = kPi / 180.0;  // <compiler angle="degree"> :18
  AntModel<double> m{};
  const double density = 5.0, r_caps = 0.08, r_torso = 0.25;  // :25, :37
  const double sx[4] = {1, -1, -1, 1}, sy[4] = {1, 1, -1, -1};  // legs :39-82
  // ankle joints: axis / range (degrees) :45, :56, :67, :78
  const double ax_sign[4] = {-1, 1, -1, 1};  // axis = (ax_sign, 1, 0)/sqrt2
  const double ank_lo[4] = {30, -70, -70, 30}, ank_hi[4] = {70, -30, -30, 70};
  // torso sphere
  double torso_mass = density * 4.0 / 3.0 * kPi * r_torso * r_torso * r_torso;
  double torso_I = 2.0 * torso_mass * r_torso * r_torso / 5.0;
  // stub capsule (on the welded leg body, torso frame): (0,0,0)->(.2sx,.2sy,0)
  double stub_hl = std::sqrt(0.08) / 2, ank_hl = std::sqrt(0.32) / 2;
  double cm, ciperp, ciax, am, aiperp, aiax;
  CapsuleMassInertia(r_caps, stub_hl, density, &cm, &ciperp, &ciax);
  CapsuleMassInertia(r_caps, ank_hl, density, &am, &aiperp, &aiax);
  // body 0: torso sphere + 4 stubs (welded): com stays at the origin by symmetry
  m.mass[0] = torso_mass + 4 * cm;
  m.com[0][0] = m.com[0][1] = m.com[0][2] = 0;
  m.inertia[0][0] = m.inertia[0][1] = m.inertia[0][2] = torso_I;
  for (int l = 0; l < 4; ++l) {
    double u[3] = {sx[l] / std::sqrt(2.0), sy[l] / std::sqrt(2.0), 0};
    double I[6], c[3] = {0.1 * sx[l], 0.1 * sy[l], 0};
    CapsuleTensor(ciperp, ciax, u, I);
    AddShifted(I, cm, c);
    for (int k = 0; k < 6; ++k) m.inertia[0][k] += I[k];
    // aux_l: leg capsule (0,0,0)->(.2sx,.2sy,0) in the aux frame
    int A = Aux(l), F = Foot(l);
    m.mass[A] = cm;
    m.com[A][0] = c[0];
    m.com[A][1] = c[1];
    m.com[A][2] = 0;
    CapsuleTensor(ciperp, ciax, u, m.inertia[A]);
    // foot_l: ankle capsule (0,0,0)->(.4sx,.4sy,0)
    m.mass[F] = am;
    m.com[F][0] = 0.2 * sx[l];
    m.com[F][1] = 0.2 * sy[l];
    m.com[F][2] = 0;
    CapsuleTensor(aiperp, aiax, u, m.inertia[F]);
    m.aux_pos[l][0] = m.foot_pos[l][0] = 0.2 * sx[l];
    m.aux_pos[l][1] = m.foot_pos[l][1] = 0.2 * sy[l];
    m.aux_pos[l][2] = m.foot_pos[l][2] = 0;
    m.ankle_axis[l][0] = ax_sign[l] / std::sqrt(2.0);
    m.ankle_axis[l][1] = 1 / std::sqrt(2.0);
    m.ankle_axis[l][2] = 0;
    // joints: hip range -30..30 (:42), ankle ranges; armature 1 damping 1 (:24)
    m.lo[2 * l] = -30 * deg;
    m.hi[2 * l] = 30 * deg;
    m.lo[2 * l + 1] = ank_lo[l] * deg;
    m.hi[2 * l + 1] = ank_hi[l] * deg;
    // end spheres: "+axis" end (= `to`) first, then the `from` end
    int s0 = 1 + 6 * l;
    const double ends[6][2] = {{0.2, 0.2}, {0, 0}, {0.2, 0.2}, {0, 0}, {0.4, 0.4}, {0, 0}};
    for (int w = 0; w < 6; ++w) {
      m.sph[s0 + w][0] = ends[w][0] * sx[l];
      m.sph[s0 + w][1] = ends[w][1] * sy[l];
      m.sph[s0 + w][2] = 0;
      m.sph_r[s0 + w] = r_caps;
    }
  }
  m.sph[0][0] = m.sph[0][1] = m.sph[0][2] = 0;
  m.sph_r[0] = r_torso;
  m.total_mass = 0;
  for (int b = 0; b < kNB; ++b) m.total_mass += m.mass[b];
  for (int j = 0; j < kNU; ++j) {
    m.damp[j] = 1.0;
    m.arm[j] = 1.0;
  }
  m.gear = 150.0;  // :85-94
  m.mu = 1.0;      // friction="1 0.5 0.5" :25
  m.margin = 0.01; // :25 (max of the pair)
  m.timestep = 0.01;  // :19
  m.gravity = 9.81;
  // MuJoCo defaults solref .02 1, solimp .9 .95 .001 .5 2; refsafe
  const double tc = std::fmax(0.02, 2 * m.timestep), dr = 1.0;
  m.imp_d0 = 0.9;
  m.imp_dmax = 0.95;
  m.imp_width = 0.001;
  m.con_K = 1.0 / (m.imp_dmax * m.imp_dmax * tc * tc * dr * dr);
  m.con_B = 2.0 / (m.imp_dmax * tc);
  // mj_setConst at qpos0 = (0 0 .75 | 1 0 0 0 | 0...)
  double q0[kNQ] = {0, 0, 0.75, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  AntPos<double> p;
  AntKinematics(m, q0, p);
  double U[kTri];
  for (int k = 0; k < kTri; ++k) U[k] = p.M[k];
  FactorUUt(U);
  double Minv[kNV][kNV];
  for (int c = 0; c < kNV; ++c) {
    double e[kNV] = {0};
    e[c] = 1;
    SolveUUt(U, e);
    for (int rr = 0; rr < kNV; ++rr) Minv[rr][c] = e[rr];
  }
  for (int j = 0; j < kNU; ++j) m.dof_invw[j] = Minv[6 + j][6 + j];
  AntGeo<double> geo;
  AntMakeGeo(m, p, geo);
  // body_invweight0 (translational) of the 13 geom-carrying MuJoCo bodies
  for (int g = 0; g < kNGeomBody; ++g) {
    int body;           // dynamic body the MuJoCo body is attached to
    double local[3];    // its COM in that body's frame
    if (g == 0) {
      body = 0;
      local[0] = local[1] = local[2] = 0;
    } else {
      int l = (g - 1) / 3, w = (g - 1) % 3;
      body = w == 0 ? 0 : (w == 1 ? Aux(l) : Foot(l));
      double f = w == 2 ? 0.2 : 0.1;
      local[0] = f * sx[l];
      local[1] = f * sy[l];
      local[2] = 0;
    }
    Vec3<double> P = p.pos[body] + Mul(p.R[body], Vec3<double>{local[0], local[1], local[2]});
    double J[3][kNV] = {{0}};
    auto fill = [&](auto jc, Vec3<double> col) {
      constexpr int j = decltype(jc)::value;
      J[0][j] = col.x;
      J[1][j] = col.y;
      J[2][j] = col.z;
    };
    switch (body) {
      case 0: ForChainCols<0>(geo, P, fill); break;
      case 1: ForChainCols<1>(geo, P, fill); break;
      case 2: ForChainCols<2>(geo, P, fill); break;
      case 3: ForChainCols<3>(geo, P, fill); break;
      case 4: ForChainCols<4>(geo, P, fill); break;
      case 5: ForChainCols<5>(geo, P, fill); break;
      case 6: ForChainCols<6>(geo, P, fill); break;
      case 7: ForChainCols<7>(geo, P, fill); break;
      default: ForChainCols<8>(geo, P, fill); break;
    }
    double tr = 0;
    for (int rr = 0; rr < 3; ++rr) {
      for (int i = 0; i < kNV; ++i) {
        for (int j = 0; j < kNV; ++j) tr += J[rr][i] * Minv[i][j] * J[rr][j];
      }
    }
    m.geom_body_invw[g] = tr / 3.0;
  }
  return m;
}

template <typename T>
constexpr AntModel<T> CastAntModel(const AntModel<double>& d) {
  AntModel<T> m{};
  for (int b = 0; b < kNB; ++b) {
    m.mass[b] = (T)d.mass[b];
    for (int k = 0; k < 3; ++k) m.com[b][k] = (T)d.com[b][k];
    for (int k = 0; k < 6; ++k) m.inertia[b][k] = (T)d.inertia[b][k];
  }
  for (int l = 0; l < kNLeg; ++l) {
    for (int k = 0; k < 3; ++k) {
      m.aux_pos[l][k] = (T)d.aux_pos[l][k];
      m.foot_pos[l][k] = (T)d.foot_pos[l][k];
      m.ankle_axis[l][k] = (T)d.ankle_axis[l][k];
    }
  }
  for (int s = 0; s < kNSph; ++s) {
    for (int k = 0; k < 3; ++k) m.sph[s][k] = (T)d.sph[s][k];
    m.sph_r[s] = (T)d.sph_r[s];
  }
  for (int g = 0; g < kNGeomBody; ++g) m.geom_body_invw[g] = (T)d.geom_body_invw[g];
  for (int j = 0; j < kNU; ++j) {
    m.lo[j] = (T)d.lo[j];
    m.hi[j] = (T)d.hi[j];
    m.dof_invw[j] = (T)d.dof_invw[j];
    m.damp[j] = (T)d.damp[j];
    m.arm[j] = (T)d.arm[j];
  }
  m.gear = (T)d.gear;
  m.total_mass = (T)d.total_mass;
  m.mu = (T)d.mu;
  m.margin = (T)d.margin;
  m.con_K = (T)d.con_K;
  m.con_B = (T)d.con_B;
  m.imp_d0 = (T)d.imp_d0;
  m.imp_dmax = (T)d.imp_dmax;
  m.imp_width = (T)d.imp_width;
  m.timestep = (T)d.timestep;
  m.gravity = (T)d.gravity;
  return m;
}

}  // namespace ant
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_ANT_MODEL_H_
