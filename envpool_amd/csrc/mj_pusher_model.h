// Host-side "model compiler" for the gym Pusher: MJCF numbers of
// third_party/mujoco_gym_xml_patches/pusher_envpool.xml (v2 / v4) and pusher_v5_envpool.xml
// (v5: no sphere on the object, cylinder density 0.01), hand transcribed and cited by the line
// of the former -> the constants mj_pusher.hip.h needs, i.e. what MuJoCo's compiler +
// mj_setConst produce (inertiafromgeom with density 300, dof_invweight0 / body_invweight0 at
// qpos0 = 0).  fp64.
#ifndef ENVPOOL_AMD_CSRC_MJ_PUSHER_MODEL_H_
#define ENVPOOL_AMD_CSRC_MJ_PUSHER_MODEL_H_

#include <cmath>
#include <stdexcept>

#include "mj_ant_model.h"  // detail::CapsuleMassInertia / CapsuleTensor / AddShifted
#include "mj_pusher.hip.h"

namespace epa {
namespace mj {
namespace pusher {

namespace detail {
struct Accum {  // mass, first moment and inertia about the link origin of a set of geoms
  double mass{0}, mom[3]{0, 0, 0}, I[6]{0, 0, 0, 0, 0, 0};
  void Add(double m, const double* c, const double* Ic) {  // Ic about the geom's own centre
    mass += m;
    for (int k = 0; k < 3; ++k) mom[k] += m * c[k];
    double J[6];
    for (int k = 0; k < 6; ++k) J[k] = Ic[k];
    ant::detail::AddShifted(J, m, c);
    for (int k = 0; k < 6; ++k) I[k] += J[k];
  }
  void Sphere(double x, double y, double z, double r, double density) {
    const double kPi = 3.14159265358979323846;
    const double m = density * 4.0 / 3.0 * kPi * r * r * r, i = 2.0 * m * r * r / 5.0;
    const double c[3] = {x, y, z}, Ic[6] = {i, i, i, 0, 0, 0};
    Add(m, c, Ic);
  }
  void Capsule(double x0, double y0, double z0, double x1, double y1, double z1, double r,
               double density) {
    const double d[3] = {x1 - x0, y1 - y0, z1 - z0};
    const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double u[3] = {d[0] / len, d[1] / len, d[2] / len};
    double m, iperp, iax, Ic[6];
    ant::detail::CapsuleMassInertia(r, len / 2, density, &m, &iperp, &iax);
    ant::detail::CapsuleTensor(iperp, iax, u, Ic);
    const double c[3] = {0.5 * (x0 + x1), 0.5 * (y0 + y1), 0.5 * (z0 + z1)};
    Add(m, c, Ic);
  }
  // mass, com, inertia about the com (xx yy zz xy xz yz)
  void Finish(double* m, double* com, double* Icom) const {
    *m = mass;
    for (int k = 0; k < 3; ++k) com[k] = mom[k] / mass;
    const double neg[3] = {com[0], com[1], com[2]};
    for (int k = 0; k < 6; ++k) Icom[k] = I[k];
    // I_origin = I_com + m (|c|^2 1 - c c^T)  =>  subtract the shift
    const double c2 = neg[0] * neg[0] + neg[1] * neg[1] + neg[2] * neg[2];
    Icom[0] -= mass * (c2 - neg[0] * neg[0]);
    Icom[1] -= mass * (c2 - neg[1] * neg[1]);
    Icom[2] -= mass * (c2 - neg[2] * neg[2]);
    Icom[3] += mass * neg[0] * neg[1];
    Icom[4] += mass * neg[0] * neg[2];
    Icom[5] += mass * neg[1] * neg[2];
  }
};
}  // namespace detail

inline PusherModel<double> BuildPusherModel(bool v5) {
  using detail::Accum;
  const double kPi = 3.14159265358979323846;
  const double dens = 300;  // <default><geom density="300"> :24
  PusherModel<double> m{};
  // links: r_shoulder_pan :31, r_shoulder_lift :39, r_upper_arm_roll (+ r_upper_arm) :43,
  // r_elbow_flex :50, r_forearm_roll (+ r_forearm) :54, r_wrist_flex :61,
  // r_wrist_roll (+ tips_arm) :65
  const double off[kNL][3] = {{0, -0.6, 0}, {0.1, 0, 0}, {0, 0, 0}, {0.4, 0, 0},
                              {0, 0, 0},    {0.321, 0, 0}, {0, 0, 0}};
  const int axis[kNL] = {2, 1, 0, 1, 0, 1, 0};  // :37 z, :41 y, :45 x, :52 y, :56 x, :63 y, :66 x
  const double lo[kNL] = {-2.2854, -0.5236, -1.5, -2.3213, -1.5, -1.094, -1.5};
  const double hi[kNL] = {1.714602, 1.3963, 1.7, 0, 1.5, 0, 1.5};
  const double damp[kNL] = {1.0, 1.0, 0.1, 0.1, 0.1, 0.1, 0.1};
  for (int l = 0; l < kNL; ++l) {
    if (axis[l] != LinkAxis(l)) throw std::logic_error("pusher: hinge axis pattern changed");
    for (int k = 0; k < 3; ++k) m.off[l][k] = off[l][k];
    m.lo[l] = lo[l];
    m.hi[l] = hi[l];
    m.damp[l] = damp[l];
    m.arm[l] = 0.04;  // <default><joint armature='0.04'> :23
  }
  m.damp[7] = m.damp[8] = 0.5;  // obj_slidey / obj_slidex :88-89
  m.arm[7] = m.arm[8] = 0.04;
  Accum L[kNL];
  L[0].Sphere(-0.06, 0.05, 0.2, 0.05, dens);  // e1 :32
  L[0].Sphere(0.06, 0.05, 0.2, 0.05, dens);   // e2
  L[0].Sphere(-0.06, 0.09, 0.2, 0.03, dens);  // e1p
  L[0].Sphere(0.06, 0.09, 0.2, 0.03, dens);   // e2p
  L[0].Capsule(0, 0, -0.4, 0, 0, 0.2, 0.1, dens);      // sp :36
  L[1].Capsule(0, -0.1, 0, 0, 0.1, 0, 0.1, dens);      // sl :40
  L[2].Capsule(-0.1, 0, 0, 0.1, 0, 0, 0.02, dens);     // uar :44
  L[2].Capsule(0, 0, 0, 0.4, 0, 0, 0.06, dens);        // ua :48 (r_upper_arm_link, welded)
  L[3].Capsule(0, -0.02, 0, 0, 0.02, 0, 0.06, dens);   // ef :51
  L[4].Capsule(-0.1, 0, 0, 0.1, 0, 0, 0.02, dens);     // fr :55
  L[4].Capsule(0, 0, 0, 0.291, 0, 0, 0.05, dens);      // fa :59 (r_forearm_link, welded)
  L[5].Capsule(0, -0.02, 0, 0, 0.02, 0, 0.01, dens);   // wf :62
  // r_wrist_roll_link: the three colliding capsules :71-73; tips_arm (welded): two spheres :68-69
  const double cap[kNCap][2][3] = {{{0, -0.1, 0}, {0, 0.1, 0}},
                                   {{0, -0.1, 0}, {0.1, -0.1, 0}},
                                   {{0, 0.1, 0}, {0.1, 0.1, 0}}};
  Accum wrist_body;  // the MuJoCo body r_wrist_roll_link alone (for its body_invweight0)
  for (int k = 0; k < kNCap; ++k) {
    L[6].Capsule(cap[k][0][0], cap[k][0][1], cap[k][0][2], cap[k][1][0], cap[k][1][1], cap[k][1][2], 0.02, dens);
    wrist_body.Capsule(cap[k][0][0], cap[k][0][1], cap[k][0][2], cap[k][1][0], cap[k][1][1], cap[k][1][2], 0.02, dens);
    for (int c = 0; c < 3; ++c) {
      m.cap_p0[k][c] = cap[k][0][c];
      m.cap_p1[k][c] = cap[k][1][c];
    }
  }
  m.cap_r = 0.02;
  L[6].Sphere(0.1, -0.1, 0, 0.01, dens);
  L[6].Sphere(0.1, 0.1, 0, 0.01, dens);
  for (int l = 0; l < kNL; ++l) L[l].Finish(&m.mass[l], m.com[l], m.inertia[l]);
  // object :85-89: sphere r 0.05 (v2 / v4 only, density 1e-5) + cylinder r 0.05 half height 0.05
  m.obj_pos[0] = 0.45;
  m.obj_pos[1] = -0.05;
  m.obj_pos[2] = -0.275;
  m.cyl_r = 0.05;
  m.cyl_h = 0.05;
  const double cyl_density = v5 ? 0.01 : 0.00001;
  m.obj_mass = cyl_density * kPi * m.cyl_r * m.cyl_r * 2 * m.cyl_h;
  if (!v5) m.obj_mass += 0.00001 * 4.0 / 3.0 * kPi * 0.05 * 0.05 * 0.05;
  m.goal_pos[0] = 0.45;  // :92
  m.goal_pos[1] = -0.05;
  m.goal_pos[2] = -0.3230;
  m.table_z = -0.325;  // :29
  m.margin = 0.002;    // :24
  // MuJoCo defaults solref .02 1, solimp .9 .95 .001 .5 2; refsafe: timeconst >= 2 h
  m.timestep = 0.01;  // :20
  const double tc = std::fmax(0.02, 2 * m.timestep), dr = 1.0;
  m.imp_d0 = 0.9;
  m.imp_dmax = 0.95;
  m.imp_width = 0.001;
  m.sol_K = 1.0 / (m.imp_dmax * m.imp_dmax * tc * tc * dr * dr);
  m.sol_B = 2.0 / (m.imp_dmax * tc);
  m.ctrl_lo = -2.0;  // :99-107
  m.ctrl_hi = 2.0;
  // mj_setConst at qpos0 = 0: M from the product's own forward pass (no row is active there)
  double q[kNV] = {0}, v[kNV] = {0}, ctrl[kNL] = {0}, warm[kNV] = {0}, qacc[kNV], M[kNV * kNV], f[kNV];
  PusherLag<double> lag;
  SolverCfg<double> cfg{50, 1e-13};
  m.wrist_invw = m.obj_invw = 1.0;
  for (int l = 0; l < kNL; ++l) m.dof_invw[l] = 1.0;
  double row_lds[kRowSlots];
  PusherForward(m, cfg, q, v, ctrl, warm, qacc, M, f, &lag, [&](int slot) -> double& { return row_lds[slot]; });
  double Minv[kNV][kNV];
  for (int c = 0; c < kNV; ++c) {
    double A[kNV * kNV], e[kNV] = {0};
    for (int k = 0; k < kNV * kNV; ++k) A[k] = M[k];
    e[c] = 1;
    CholSolveN<double, kNV>(A, e);
    for (int r = 0; r < kNV; ++r) Minv[r][c] = e[r];
  }
  for (int l = 0; l < kNL; ++l) m.dof_invw[l] = Minv[l][l];
  // body_invweight0 (translational) of r_wrist_roll_link at its own COM; at qpos0 every
  // frame is the identity: link origins are the cumulated offsets, hinge axes the unit axes
  double wm, wcom[3], wI[6], org[kNL][3], acc[3] = {0, 0, 0};
  wrist_body.Finish(&wm, wcom, wI);
  for (int l = 0; l < kNL; ++l) {
    for (int k = 0; k < 3; ++k) {
      acc[k] += off[l][k];
      org[l][k] = acc[k];
    }
  }
  const double P[3] = {acc[0] + wcom[0], acc[1] + wcom[1], acc[2] + wcom[2]};
  double J[3][kNL];
  for (int l = 0; l < kNL; ++l) {
    double a[3] = {0, 0, 0}, r[3] = {P[0] - org[l][0], P[1] - org[l][1], P[2] - org[l][2]};
    a[axis[l]] = 1;
    J[0][l] = a[1] * r[2] - a[2] * r[1];
    J[1][l] = a[2] * r[0] - a[0] * r[2];
    J[2][l] = a[0] * r[1] - a[1] * r[0];
  }
  double tr = 0;
  for (int rr = 0; rr < 3; ++rr) {
    for (int i = 0; i < kNL; ++i) {
      for (int j = 0; j < kNL; ++j) tr += J[rr][i] * Minv[i][j] * J[rr][j];
    }
  }
  m.wrist_invw = tr / 3.0;
  m.obj_invw = (Minv[7][7] + Minv[8][8]) / 3.0;  // slides along y and x; no z dof
  return m;
}

}  // namespace pusher
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_PUSHER_MODEL_H_
