// Host-side model compiler for the two gym inverted pendulums and Reacher: the
// numbers of third_party/mujoco_gym_xml_patches/inverted_pendulum_envpool.xml,
// inverted_double_pendulum_envpool.xml and reacher_envpool.xml (hand transcribed, cited by XML line; the
// reference loads them at envpool/mujoco/gym/mujoco_env.h:50-58,87) turned into
// what MuJoCo's compiler + mj_setConst produce: capsule mass / inertia
// (inertiafromgeom, default density 1000), dof_invweight0 at qpos0 = 0.
#ifndef ENVPOOL_AMD_CSRC_MJ_PENDULUM_MODEL_H_
#define ENVPOOL_AMD_CSRC_MJ_PENDULUM_MODEL_H_

#include <cmath>

#include "mj_pendulum.hip.h"

namespace epa {
namespace mj {
namespace pend {

inline void CapsuleMassInertia(double r, double half, double* mass, double* iperp) {
  const double kPi = 3.14159265358979323846, density = 1000.0;
  const double h = 2 * half;
  const double gm = density * kPi * (r * r * h + 4.0 * r * r * r / 3.0);
  const double sphere_mass = gm * 4 * r / (4 * r + 3 * h), cyl_mass = gm - sphere_mass;
  *mass = gm;
  // about an axis perpendicular to the capsule axis, through its centre
  *iperp = cyl_mass * (3 * r * r + h * h) / 12 + 2 * sphere_mass * r * r / 5 +
           sphere_mass * h * (3 * r + 2 * h) / 8;
}

// total mass + dof_invweight0 = diag(M^-1) at qpos0 = 0 (mj_setConst)
template <int NL, int kBase>
inline void PendSetConst(PendModel<double, NL, kBase>& m) {
  constexpr int NV = NL + kBase;
  m.total_mass = kBase == kBaseCart ? m.cart_mass : 0.0;
  for (int i = 0; i < NL; ++i) m.total_mass += m.mass[i];
  const double q0[NV] = {0};
  PendPos<double, NL> p;
  PendKinematics(m, q0, p);
  for (int c = 0; c < NV; ++c) {
    double A[NV * NV], e[NV] = {0};
    for (int k = 0; k < NV * NV; ++k) A[k] = p.M[k];
    e[c] = 1;
    CholSolve<double, NV>(A, e);
    m.dof_invw[c] = e[c];
  }
}

inline void PendDefaults(double timestep, double* K, double* B, double* d0, double* dmax,
                         double* width) {
  // MuJoCo defaults: solref .02 1 (timeconst clamped to 2 * timestep), solimp .9 .95 .001
  const double tc = std::fmax(0.02, 2 * timestep), dr = 1.0;
  *d0 = 0.9;
  *dmax = 0.95;
  *width = 0.001;
  *K = 1.0 / (0.95 * 0.95 * tc * tc * dr * dr);
  *B = 2.0 / (0.95 * tc);
}

// inverted_pendulum_envpool.xml
inline PendModel<double, 1, kBaseCart> BuildInvertedPendulum() {
  const double deg = 3.14159265358979323846 / 180.0;  // <compiler> default angle = degree
  PendModel<double, 1, kBaseCart> m{};
  double iperp;
  CapsuleMassInertia(0.1, 0.1, &m.cart_mass, &iperp);  // cart capsule size=".1 .1" :32
  // pole: fromto="0 0 0 0.001 0 0.6" size="0.049 0.3" :35 (fromto fixes the length)
  const double tx = 0.001, tz = 0.6, half = 0.5 * std::sqrt(tx * tx + tz * tz);
  CapsuleMassInertia(0.049, half, &m.mass[0], &m.iyy[0]);
  m.cx[0] = 0.5 * tx;
  m.cz[0] = 0.5 * tz;
  m.lx[0] = tx;  // no site in this model: the far end of the pole
  m.lz[0] = tz;
  // <joint armature="0" damping="1" limited="true"/> :20; slider range -1 1 :31,
  // hinge range -90 90 (degrees) :34
  m.damp[0] = m.damp[1] = 1.0;
  m.limited[0] = m.limited[1] = 1;
  m.lo[0] = -1;
  m.hi[0] = 1;
  m.lo[1] = -90 * deg;
  m.hi[1] = 90 * deg;
  m.margin[0] = m.margin[1] = 0;
  m.grav_x = 0;
  m.grav_z = -9.81;  // :25
  m.timestep = 0.02;  // :25, integrator RK4
  m.gear[0] = 100;    // :41, ctrlrange -3 3
  m.ctrl_lo = -3;
  m.ctrl_hi = 3;
  PendDefaults(m.timestep, &m.lim_K, &m.lim_B, &m.lim_d0, &m.lim_dmax, &m.lim_width);
  PendSetConst(m);
  return m;
}

// inverted_double_pendulum_envpool.xml
inline PendModel<double, 2, kBaseCart> BuildInvertedDoublePendulum() {
  PendModel<double, 2, kBaseCart> m{};
  double iperp;
  CapsuleMassInertia(0.1, 0.1, &m.cart_mass, &iperp);  // cart :48
  for (int i = 0; i < 2; ++i) {  // poles: fromto="0 0 0 0 0 0.6" size="0.045 0.3" :51,:54
    CapsuleMassInertia(0.045, 0.3, &m.mass[i], &m.iyy[i]);
    m.cx[i] = 0;
    m.cz[i] = 0.3;
    m.lx[i] = 0;  // pole2 at pos="0 0 0.6" :52; "tip" site pos="0 0 .6" :55
    m.lz[i] = 0.6;
  }
  // <joint damping="0.05"/> :38; only the slider is limited: range -1 1, margin 0.01 :47
  for (int j = 0; j < 3; ++j) {
    m.damp[j] = 0.05;
    m.limited[j] = 0;
    m.lo[j] = m.hi[j] = m.margin[j] = 0;
  }
  m.limited[0] = 1;
  m.lo[0] = -1;
  m.hi[0] = 1;
  m.margin[0] = 0.01;
  m.grav_x = 1e-5;  // gravity="1e-5 0 -9.81" :41
  m.grav_z = -9.81;
  m.timestep = 0.01;  // :41, integrator RK4
  m.gear[0] = 500;    // :60, ctrlrange -1 1
  m.ctrl_lo = -1;
  m.ctrl_hi = 1;
  PendDefaults(m.timestep, &m.lim_K, &m.lim_B, &m.lim_d0, &m.lim_dmax, &m.lim_width);
  PendSetConst(m);
  return m;
}

// reacher_envpool.xml: the two-link arm (hinges about +z in the xy plane; see the
// header of mj_pendulum.hip.h for the z := -y mapping).  The target body only has
// two undamped slides along x / y with zero velocity and no in-plane force (gravity
// is along -z), so it never moves: its qpos are constants of an episode and are
// handled by the step kernel, not by the dynamics.
inline PendModel<double, 2, kBaseFixed> BuildReacher() {
  const double kPi = 3.14159265358979323846, density = 1000.0;
  PendModel<double, 2, kBaseFixed> m{};
  m.cart_mass = 0;
  double cm, ci;
  CapsuleMassInertia(0.01, 0.05, &cm, &ci);  // link0 / link1: fromto 0 0 0 .1 0 0, size .01 :35,:39
  // link 0 (body0 :34): one capsule, next hinge at pos=".1 0 0" :37
  m.mass[0] = cm;
  m.iyy[0] = ci;
  m.cx[0] = 0.05;
  m.cz[0] = 0;
  m.lx[0] = 0.1;
  m.lz[0] = 0;
  // link 1 (body1 :37) + the jointless fingertip body (:40-42: sphere r=.01 at .11 0 0)
  const double sm = density * 4.0 / 3.0 * kPi * 1e-6, si = 0.4 * sm * 1e-4;
  const double mt = cm + sm, c1 = (cm * 0.05 + sm * 0.11) / mt;
  m.mass[1] = mt;
  m.cx[1] = c1;
  m.cz[1] = 0;
  m.iyy[1] = ci + cm * (0.05 - c1) * (0.05 - c1) + si + sm * (0.11 - c1) * (0.11 - c1);
  m.lx[1] = 0.11;  // the fingertip body origin
  m.lz[1] = 0;
  // <joint armature="1" damping="1" limited="true"/> :20; joint0 limited="false" :36,
  // joint1 range -3 3 (radian) :38
  for (int j = 0; j < 2; ++j) {
    m.damp[j] = 1;
    m.arm[j] = 1;
    m.gear[j] = 200;  // :53-54, ctrlrange -1 1
    m.limited[j] = j == 1;
    m.lo[j] = j == 1 ? -3.0 : 0;
    m.hi[j] = j == 1 ? 3.0 : 0;
    m.margin[j] = 0;
  }
  m.ctrl_lo = -1;
  m.ctrl_hi = 1;
  m.grav_x = m.grav_z = 0;  // gravity "0 0 -9.81" :23 is normal to the plane of motion
  m.timestep = 0.01;        // :23, integrator RK4
  PendDefaults(m.timestep, &m.lim_K, &m.lim_B, &m.lim_d0, &m.lim_dmax, &m.lim_width);
  PendSetConst(m);
  return m;
}

// swimmer_envpool.xml: three capsules in the xy plane on a planar floating base
// (slider1 x, slider2 y, free_body_rot z :36-38), two motorised hinges, a medium
// with density 4000 / viscosity 0.1 (:19).  y maps to -z of the kernel's plane.
inline PendModel<double, 3, kBaseFree> BuildSwimmer() {
  const double kPi = 3.14159265358979323846, deg = kPi / 180.0;  // angle="degree" :18
  PendModel<double, 3, kBaseFree> m{};
  // every link: capsule of length 1, size=".1", density 1000 (:35,:40,:43) running along -x
  // from its hinge (torso: fromto 1.5 0 0 .5 0 0 with the hinge at the origin)
  const double r = 0.1, half = 0.5, h = 2 * half, density = 1000.0;
  const double gm = density * kPi * (r * r * h + 4.0 * r * r * r / 3.0);
  const double sphere_mass = gm * 4 * r / (4 * r + 3 * h), cyl_mass = gm - sphere_mass;
  const double iperp = cyl_mass * (3 * r * r + h * h) / 12 + 2 * sphere_mass * r * r / 5 +
                       sphere_mass * h * (3 * r + 2 * h) / 8;
  const double iax = cyl_mass * r * r / 2 + 2 * sphere_mass * r * r / 5;
  const double com_x[3] = {1.0, -0.5, -0.5};   // capsule centre in the link frame
  const double next_x[3] = {0.5, -1.0, -1.0};  // mid at pos=".5 0 0" :39, back at "-1 0 0" :42
  for (int l = 0; l < 3; ++l) {
    m.mass[l] = gm;
    m.iyy[l] = iperp;
    m.cx[l] = com_x[l];
    m.cz[l] = 0;
    m.lx[l] = next_x[l];
    m.lz[l] = 0;
    // equivalent inertia box (mj_inertiaBoxFluidModel): full sizes from the principal inertias
    m.box[l][0] = std::sqrt(6.0 * (iperp + iperp - iax) / gm);
    m.box[l][1] = std::sqrt(6.0 * (iax + iperp - iperp) / gm);
    m.box[l][2] = m.box[l][1];
  }
  m.fluid_density = 4000;
  m.fluid_viscosity = 0.1;
  // <joint armature='0.1'/> :22 on all five joints, no damping; motor1_rot / motor2_rot
  // limited to +-100 deg (:41,:44), gear 150 (:50-51)
  for (int j = 0; j < 5; ++j) {
    m.damp[j] = 0;
    m.arm[j] = 0.1;
    m.gear[j] = j >= 3 ? 150.0 : 0.0;
    m.limited[j] = j >= 3;
    m.lo[j] = j >= 3 ? -100 * deg : 0;
    m.hi[j] = j >= 3 ? 100 * deg : 0;
    m.margin[j] = 0;
  }
  m.ctrl_lo = -1;
  m.ctrl_hi = 1;
  m.grav_x = m.grav_z = 0;  // default gravity (0 0 -9.81) is normal to the plane of motion
  m.timestep = 0.01;        // :19, integrator RK4
  PendDefaults(m.timestep, &m.lim_K, &m.lim_B, &m.lim_d0, &m.lim_dmax, &m.lim_width);
  PendSetConst(m);
  return m;
}

template <typename T, int NL, int kBase>
inline PendModel<T, NL, kBase> CastPendModel(const PendModel<double, NL, kBase>& d) {
  PendModel<T, NL, kBase> m{};
  m.cart_mass = (T)d.cart_mass;
  for (int i = 0; i < NL; ++i) {
    m.mass[i] = (T)d.mass[i];
    m.iyy[i] = (T)d.iyy[i];
    m.cx[i] = (T)d.cx[i];
    m.cz[i] = (T)d.cz[i];
    m.lx[i] = (T)d.lx[i];
    m.lz[i] = (T)d.lz[i];
  }
  for (int l = 0; l < NL; ++l) {
    for (int k = 0; k < 3; ++k) m.box[l][k] = (T)d.box[l][k];
  }
  m.fluid_density = (T)d.fluid_density;
  m.fluid_viscosity = (T)d.fluid_viscosity;
  for (int j = 0; j < NL + 2; ++j) {
    m.damp[j] = (T)d.damp[j];
    m.arm[j] = (T)d.arm[j];
    m.gear[j] = (T)d.gear[j];
    m.limited[j] = d.limited[j];
    m.lo[j] = (T)d.lo[j];
    m.hi[j] = (T)d.hi[j];
    m.margin[j] = (T)d.margin[j];
    m.dof_invw[j] = (T)d.dof_invw[j];
  }
  m.grav_x = (T)d.grav_x;
  m.grav_z = (T)d.grav_z;
  m.ctrl_lo = (T)d.ctrl_lo;
  m.ctrl_hi = (T)d.ctrl_hi;
  m.lim_K = (T)d.lim_K;
  m.lim_B = (T)d.lim_B;
  m.lim_d0 = (T)d.lim_d0;
  m.lim_dmax = (T)d.lim_dmax;
  m.lim_width = (T)d.lim_width;
  m.timestep = (T)d.timestep;
  m.total_mass = (T)d.total_mass;
  return m;
}

}  // namespace pend
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_PENDULUM_MODEL_H_
