// Host-side "model compiler" for the gym HalfCheetah: turns the MJCF numbers of
// third_party/mujoco_gym_xml_patches/half_cheetah_envpool.xml (hand
// transcribed below, cited by XML line; the reference loads that file at
// envpool/mujoco/gym/mujoco_env.h:50-58,87) into the constants the planar
// kernel needs — what MuJoCo's compiler + mj_setConst would produce:
// capsule mass/inertia (inertiafromgeom), settotalmass=14, capsule end-sphere
// centres, dof_invweight0 / body_invweight0 at qpos0.  Everything in fp64;
// cast to the kernel's arithmetic type afterwards.
#ifndef ENVPOOL_AMD_CSRC_MJ_CHEETAH_MODEL_H_
#define ENVPOOL_AMD_CSRC_MJ_CHEETAH_MODEL_H_

#include <cmath>

#include "mj_cheetah.hip.h"

namespace epa {
namespace mj {

struct CheetahCapsule {  // <geom type="capsule" .../> attached to `body`
  int body;
  double px, pz;    // geom centre in the body frame
  double angle;     // rotation about +y of the capsule's local z axis
  double half_len;  // size[1]
};

// mj_setConst: M at qpos0 -> dof_invweight0, body_invweight0
inline void PlanarSetConst(CheetahModel<double>& m, const double* q0) {
  CheetahPos<double> p;
  CheetahKinematics(m, q0, p);
  double U[kTri];
  for (int k = 0; k < kTri; ++k) U[k] = p.M[k];
  // structural zeros are never written by the kernel code: clear them here
  for (int i = 3; i <= 5; ++i) {
    for (int j = 6; j <= 8; ++j) U[TriIdx(i, j)] = 0;
  }
  FactorUUt(U);
  double Minv[kNV][kNV];
  for (int c = 0; c < kNV; ++c) {
    double e[kNV] = {0};
    e[c] = 1;
    SolveUUt(U, e);
    for (int rr = 0; rr < kNV; ++rr) Minv[rr][c] = e[rr];
  }
  for (int j = 0; j < kNU; ++j) m.dof_invw[j] = Minv[j + 3][j + 3];
  for (int b = 0; b < kNB; ++b) {
    // translational Jacobian of the body COM (x and z rows; y row is zero)
    double xi = p.px[b] + p.cs[b] * m.cx[b] + p.sn[b] * m.cz[b];
    double zi = p.pz[b] - p.sn[b] * m.cx[b] + p.cs[b] * m.cz[b];
    double Jx[kNV] = {0}, Jz[kNV] = {0};
    Jx[0] = 1;
    Jz[1] = 1;
    for (int j = 2; j < kNV; ++j) {
      if (!InChain(j, b)) continue;
      int jb = DofBody(j);
      Jx[j] = zi - p.pz[jb];
      Jz[j] = -(xi - p.px[jb]);
    }
    double a = 0;
    for (int i = 0; i < kNV; ++i) {
      for (int j = 0; j < kNV; ++j) {
        a += Jx[i] * Minv[i][j] * Jx[j] + Jz[i] * Minv[i][j] * Jz[j];
      }
    }
    m.body_invw[b] = a / 3.0;
  }
}

inline CheetahModel<double> BuildCheetahModel() {
  const double kPi = 3.14159265358979323846;
  CheetahModel<double> m{};
  // body_pos: torso :70, bthigh :78, bshin :81, bfoot :84, fthigh :92,
  // fshin :95, ffoot :98
  const double lx[kNB] = {0.0, -0.5, 0.16, -0.28, 0.5, -0.14, 0.13};
  const double lz[kNB] = {0.7, 0.0, -0.25, -0.14, 0.0, -0.24, -0.18};
  // capsules (radius 0.046): torso fromto -.5 0 0 .5 0 0 (:75) = centre 0,
  // axis +x (angle pi/2), half length .5; the others use axisangle about y.
  const CheetahCapsule caps[8] = {
      {0, 0.0, 0.0, kPi / 2, 0.5},       // torso :75
      {0, 0.6, 0.1, 0.87, 0.15},         // head :76
      {1, 0.1, -0.13, -3.8, 0.145},      // bthigh :80
      {2, -0.14, -0.07, -2.03, 0.15},    // bshin :83
      {3, 0.03, -0.097, -0.27, 0.094},   // bfoot :86
      {4, -0.07, -0.12, 0.52, 0.133},    // fthigh :94
      {5, 0.065, -0.09, -0.6, 0.106},    // fshin :97
      {6, 0.045, -0.07, -0.6, 0.07},     // ffoot :100
  };
  const double r = 0.046, density = 1000.0;  // MuJoCo default density
  for (int e = 0; e < kNEnd; ++e) m.er[e] = r;
  double mass[kNB] = {0}, mcx[kNB] = {0}, mcz[kNB] = {0};
  double gm[8], gi[8];
  for (int g = 0; g < 8; ++g) {
    double h = 2 * caps[g].half_len;
    double vol = kPi * (r * r * h + 4.0 * r * r * r / 3.0);
    gm[g] = density * vol;
    double sphere_mass = gm[g] * 4 * r / (4 * r + 3 * h);
    double cyl_mass = gm[g] - sphere_mass;
    // inertia about an axis perpendicular to the capsule axis (= world y)
    gi[g] = cyl_mass * (3 * r * r + h * h) / 12 + 2 * sphere_mass * r * r / 5 +
            sphere_mass * h * (3 * r + 2 * h) / 8;
    int b = caps[g].body;
    mass[b] += gm[g];
    mcx[b] += gm[g] * caps[g].px;
    mcz[b] += gm[g] * caps[g].pz;
    // end-sphere centres: +axis end first, like mjc_PlaneCapsule
    double ux = std::sin(caps[g].angle), uz = std::cos(caps[g].angle);
    m.ex[2 * g] = caps[g].px + caps[g].half_len * ux;
    m.ez[2 * g] = caps[g].pz + caps[g].half_len * uz;
    m.ex[2 * g + 1] = caps[g].px - caps[g].half_len * ux;
    m.ez[2 * g + 1] = caps[g].pz - caps[g].half_len * uz;
  }
  double total = 0;
  for (int b = 0; b < kNB; ++b) {
    m.lx[b] = lx[b];
    m.lz[b] = lz[b];
    m.cx[b] = mcx[b] / mass[b];
    m.cz[b] = mcz[b] / mass[b];
    m.mass[b] = mass[b];
    total += mass[b];
  }
  for (int g = 0; g < 8; ++g) {
    int b = caps[g].body;
    double dx = caps[g].px - m.cx[b], dz = caps[g].pz - m.cz[b];
    m.iyy[b] += gi[g] + gm[g] * (dx * dx + dz * dz);
  }
  const double scale = 14.0 / total;  // settotalmass="14" :52
  for (int b = 0; b < kNB; ++b) {
    m.mass[b] *= scale;
    m.iyy[b] *= scale;
  }
  m.total_mass = 14.0;
  // joints :79-97 (stiffness, damping, range), defaults :54 (armature .1),
  // actuators :105-110 (gear)
  const double stiff[kNU] = {240, 180, 120, 180, 120, 60};
  const double damp[kNU] = {6, 4.5, 3, 4.5, 3, 1.5};
  const double lo[kNU] = {-0.52, -0.785, -0.4, -1.0, -1.2, -0.5};
  const double hi[kNU] = {1.05, 0.785, 0.785, 0.7, 0.87, 0.5};
  const double gear[kNU] = {120, 90, 60, 120, 60, 30};
  for (int j = 0; j < kNU; ++j) {
    m.stiff[j] = stiff[j];
    m.damp[j] = damp[j];
    m.arm[j] = 0.1;
    m.lo[j] = lo[j];
    m.hi[j] = hi[j];
    m.gear[j] = gear[j];
  }
  m.timestep = 0.01;  // :59
  m.gravity = 9.81;   // :59
  for (int b = 0; b < kNB; ++b) m.bmu[b] = 0.4;  // friction=".4 .1 .1" :55 (max of the pair, identical)
  // solref=".02 1" :55 / solreflimit=".02 1" :54, refsafe: tc >= 2*timestep
  const double tc = std::fmax(0.02, 2 * m.timestep), dr = 1.0;
  // solimp="0 .8 .01" :55, solimplimit="0 .8 .03" :54, d0 clamped to 1e-4
  m.con_d0 = 0.0001;
  m.con_dmax = 0.8;
  m.con_width = 0.01;
  m.lim_d0 = 0.0001;
  m.lim_dmax = 0.8;
  m.lim_width = 0.03;
  m.con_K = 1.0 / (m.con_dmax * m.con_dmax * tc * tc * dr * dr);
  m.con_B = 2.0 / (m.con_dmax * tc);
  m.lim_K = 1.0 / (m.lim_dmax * m.lim_dmax * tc * tc * dr * dr);
  m.lim_B = 2.0 / (m.lim_dmax * tc);
  const double q0[kNV] = {0};
  PlanarSetConst(m, q0);
  return m;
}


// ---- Walker2d -------------------------------------------------------------------
// third_party/mujoco_gym_xml_patches/walker2d_envpool.xml (v3/v4) and
// walker2d_v5_envpool.xml (v5: right foot friction 1.9, :47), the files the
// reference loads for Walker2d (envpool/mujoco/gym/walker2d.h:41,
// gym/registration.py:79-83, mujoco_env.h:50-58).  Same kinematic tree as the
// HalfCheetah (torso + two 3-link legs), so it runs on the same planar kernel
// after two exact changes of variables made here, not in the kernel:
//  * joint anchors: leg_joint / foot_joint sit at `pos` != 0 in their body
//    frame (:43,:46); every body frame is re-centred on its hinge anchor
//    (body_pos' = body_pos - parent_jpos + jpos, local points -= jpos), which
//    leaves the motion of every material point unchanged;
//  * the six leg hinges rotate about -y (:40-59); the kernel's hinges rotate
//    about +y, so it integrates q' = -q (ranges mirrored, gear negated) and the
//    step kernel flips the sign of those qpos / qvel / warm-start entries on
//    load and store (PlanarTaskSigns).
inline CheetahModel<double> BuildWalkerModel(bool v5) {
  const double kPi = 3.14159265358979323846, deg = kPi / 180.0;  // angle="degree" :24
  CheetahModel<double> m{};
  // original body_pos / joint pos (x, z): torso :33 (rootz ref=1.25 => world z
  // of the torso origin is qpos[1] itself), thigh :39, leg :42-43, foot :45-46
  const double bpx[kNB] = {0, 0, 0, 0.20000000000000001, 0, 0, 0.20000000000000001};
  const double bpz[kNB] = {0, -0.19999999999999996, -0.70000000000000007, -0.34999999999999998,
                           -0.19999999999999996, -0.70000000000000007, -0.34999999999999998};
  const double jpx[kNB] = {0, 0, 0, -0.20000000000000001, 0, 0, -0.20000000000000001};
  const double jpz[kNB] = {0, 0, 0.25, 0.10000000000000001, 0, 0.25, 0.10000000000000001};
  for (int b = 0; b < kNB; ++b) {
    const int par = Parent(b);
    m.lx[b] = bpx[b] + jpx[b] - (par >= 0 ? jpx[par] : 0.0);
    m.lz[b] = bpz[b] + jpz[b] - (par >= 0 ? jpz[par] : 0.0);
  }
  // capsules (:38,:41,:44,:47 and the left copies): centre (original frame),
  // rotation about +y of the local z axis (foot quat = -90 deg about y), radius, half length
  const double gpx[kNB] = {0, 0, 0, -0.10000000000000001, 0, 0, -0.10000000000000001};
  const double gpz[kNB] = {0, -0.22500000000000009, 0, 0.10000000000000001,
                           -0.22500000000000009, 0, 0.10000000000000001};
  const double gang[kNB] = {0, 0, 0, -kPi / 2, 0, 0, -kPi / 2};
  const double grad[kNB] = {0.050000000000000003, 0.050000000000000003, 0.040000000000000001,
                            0.059999999999999998, 0.050000000000000003, 0.040000000000000001,
                            0.059999999999999998};
  const double ghalf[kNB] = {0.19999999999999996, 0.22500000000000003, 0.25, 0.10000000000000001,
                             0.22500000000000003, 0.25, 0.10000000000000001};
  const double density = 1000.0;  // :27
  for (int e = 0; e < kNEnd; ++e) {
    m.ex[e] = m.ez[e] = 0;
    m.er[e] = -1e30;  // ends 2, 3 of the torso stay unused (one torso capsule only)
  }
  for (int b = 0; b < kNB; ++b) {  // one capsule per body
    const double r = grad[b], h = 2 * ghalf[b];
    const double vol = kPi * (r * r * h + 4.0 * r * r * r / 3.0);
    const double gm = density * vol;
    const double sphere_mass = gm * 4 * r / (4 * r + 3 * h), cyl_mass = gm - sphere_mass;
    const double gi = cyl_mass * (3 * r * r + h * h) / 12 + 2 * sphere_mass * r * r / 5 +
                      sphere_mass * h * (3 * r + 2 * h) / 8;
    m.mass[b] = gm;  // inertiafromgeom, no settotalmass
    m.iyy[b] = gi;
    m.cx[b] = gpx[b] - jpx[b];
    m.cz[b] = gpz[b] - jpz[b];
    const double ux = std::sin(gang[b]), uz = std::cos(gang[b]);
    const int e0 = b == 0 ? 0 : 2 * b + 2;  // EndBody(e) = e < 4 ? 0 : (e - 4) / 2 + 1
    m.ex[e0] = m.cx[b] + ghalf[b] * ux;
    m.ez[e0] = m.cz[b] + ghalf[b] * uz;
    m.ex[e0 + 1] = m.cx[b] - ghalf[b] * ux;
    m.ez[e0 + 1] = m.cz[b] - ghalf[b] * uz;
    m.er[e0] = m.er[e0 + 1] = r;
    m.total_mass += gm;
  }
  // friction: max(floor .7 (:27,:32), geom): .9 everywhere, left foot 1.9 (:60),
  // right foot 1.9 in the v5 file only
  for (int b = 0; b < kNB; ++b) m.bmu[b] = 0.9;
  m.bmu[6] = 1.9;
  if (v5) m.bmu[3] = 1.9;
  // <joint armature="0.01" damping=".1" limited="true"/> :26; ranges :40-59
  // mirrored for q' = -q; motors gear 100 (:68-73) negated
  const double lo_deg[3] = {-150, -150, -45}, hi_deg[3] = {0, 0, 45};
  for (int j = 0; j < kNU; ++j) {
    m.stiff[j] = 0;
    m.damp[j] = 0.1;
    m.arm[j] = 0.01;
    m.lo[j] = -hi_deg[j % 3] * deg;
    m.hi[j] = -lo_deg[j % 3] * deg;
    m.gear[j] = -100;
  }
  m.timestep = 0.002;  // :29
  m.gravity = 9.81;
  // MuJoCo defaults: solref .02 1, solimp .9 .95 .001 (contacts and limits)
  const double tc = std::fmax(0.02, 2 * m.timestep), dr = 1.0;
  m.con_d0 = m.lim_d0 = 0.9;
  m.con_dmax = m.lim_dmax = 0.95;
  m.con_width = m.lim_width = 0.001;
  m.con_K = m.lim_K = 1.0 / (0.95 * 0.95 * tc * tc * dr * dr);
  m.con_B = m.lim_B = 2.0 / (0.95 * tc);
  const double q0[kNV] = {0, 1.25, 0, 0, 0, 0, 0, 0, 0};  // qpos0 (rootz ref)
  PlanarSetConst(m, q0);
  return m;
}

// ---- Hopper -----------------------------------------------------------------------
// third_party/mujoco_gym_xml_patches/hopper_envpool.xml (hopper.h:38).  One leg:
// the torso + thigh/leg/foot chain occupies bodies 0..3 of the planar tree; the
// second leg (bodies 4..6) is a massless ghost with armature 1 on its hinges, so
// its three dofs stay at rest and never couple (M rows = diag(1)).  Frames are
// re-centred on the hinge anchors and the -y hinges mirrored like BuildWalkerModel.
// Unlike the other planar models the geoms collide with each other (contype =
// conaffinity = 1, condim 1, margin 0.001 :26): see kPairBody* in mj_cheetah.hip.h.
inline CheetahModel<double> BuildHopperModel() {
  const double kPi = 3.14159265358979323846, deg = kPi / 180.0;  // angle="degree" :23
  CheetahModel<double> m{};
  // original body_pos / joint pos (x, z): torso :36 (rootz ref=1.25), thigh :42,
  // leg :45-46, foot :48-49; ghost bodies at their parents' origins
  const double bpx[kNB] = {0, 0, 0, 0.13, 0, 0, 0};
  const double bpz[kNB] = {0, -0.19999999999999996, -0.70000000000000007, -0.35, 0, 0, 0};
  const double jpx[kNB] = {0, 0, 0, -0.13, 0, 0, 0};
  const double jpz[kNB] = {0, 0, 0.25, 0.1, 0, 0, 0};
  for (int b = 0; b < kNB; ++b) {
    const int par = Parent(b);
    m.lx[b] = bpx[b] + jpx[b] - (par >= 0 ? jpx[par] : 0.0);
    m.lz[b] = bpz[b] + jpz[b] - (par >= 0 ? jpz[par] : 0.0);
  }
  // capsules :41,:44,:47,:50: centre (original frame), rotation of the local z axis
  // about +y (foot quat = -90 deg), radius, half length
  const double gpx[4] = {0, 0, 0, -0.065}, gpz[4] = {0, -0.22500000000000009, 0, 0.1};
  const double gang[4] = {0, 0, 0, -kPi / 2};
  const double grad[4] = {0.05, 0.05, 0.04, 0.06};
  const double ghalf[4] = {0.19999999999999996, 0.22500000000000003, 0.25, 0.195};
  const double density = 1000.0;  // MuJoCo default
  for (int e = 0; e < kNEnd; ++e) {
    m.ex[e] = m.ez[e] = 0;
    m.er[e] = -1e30;  // unused: torso ends 2, 3 and the whole ghost leg
  }
  for (int b = 0; b < 4; ++b) {
    const double r = grad[b], h = 2 * ghalf[b];
    const double vol = kPi * (r * r * h + 4.0 * r * r * r / 3.0);
    const double gm = density * vol;
    const double sphere_mass = gm * 4 * r / (4 * r + 3 * h), cyl_mass = gm - sphere_mass;
    m.mass[b] = gm;
    m.iyy[b] = cyl_mass * (3 * r * r + h * h) / 12 + 2 * sphere_mass * r * r / 5 +
               sphere_mass * h * (3 * r + 2 * h) / 8;
    m.cx[b] = gpx[b] - jpx[b];
    m.cz[b] = gpz[b] - jpz[b];
    const double ux = std::sin(gang[b]), uz = std::cos(gang[b]);
    const int e0 = BodyEnd0(b);
    m.ex[e0] = m.cx[b] + ghalf[b] * ux;
    m.ez[e0] = m.cz[b] + ghalf[b] * uz;
    m.ex[e0 + 1] = m.cx[b] - ghalf[b] * ux;
    m.ez[e0 + 1] = m.cz[b] - ghalf[b] * uz;
    m.er[e0] = m.er[e0 + 1] = r;
    m.total_mass += gm;
  }
  // friction of the floor pair: max(floor 1 (MuJoCo default, :35), geom): .9 -> 1, foot 2.0 (:50)
  for (int b = 0; b < kNB; ++b) m.bmu[b] = 1.0;
  m.bmu[3] = 2.0;
  // <joint armature="1" damping="1" limited="true"/> :25; ranges :43,:46,:49 mirrored for
  // q' = -q; motors gear 200 (:57-59) negated.  Ghost hinges: armature only.
  const double lo_deg[3] = {-150, -150, -45}, hi_deg[3] = {0, 0, 45};
  for (int j = 0; j < kNU; ++j) {
    const bool real = j < 3;
    m.stiff[j] = 0;
    m.damp[j] = real ? 1.0 : 0.0;
    m.arm[j] = 1.0;
    m.lo[j] = real ? -hi_deg[j] * deg : -1e30;
    m.hi[j] = real ? -lo_deg[j] * deg : 1e30;
    m.gear[j] = real ? -200.0 : 0.0;
  }
  m.timestep = 0.002;  // :29, integrator RK4
  m.gravity = 9.81;
  // contacts: solref ".02 1", solimp ".8 .8 .01" (:26, both geoms of every pair);
  // limits: MuJoCo defaults solref .02 1, solimp .9 .95 .001
  const double tc = std::fmax(0.02, 2 * m.timestep), dr = 1.0;
  m.con_d0 = 0.8;
  m.con_dmax = 0.8;
  m.con_width = 0.01;
  m.con_K = 1.0 / (0.8 * 0.8 * tc * tc * dr * dr);
  m.con_B = 2.0 / (0.8 * tc);
  m.lim_d0 = 0.9;
  m.lim_dmax = 0.95;
  m.lim_width = 0.001;
  m.lim_K = 1.0 / (0.95 * 0.95 * tc * tc * dr * dr);
  m.lim_B = 2.0 / (0.95 * tc);
  m.con_margin = 0.001;  // margin="0.001" :26
  m.n_pairs = kNPair;
  const double q0[kNV] = {0, 1.25, 0, 0, 0, 0, 0, 0, 0};
  PlanarSetConst(m, q0);
  return m;
}

template <typename T>
constexpr CheetahModel<T> CastCheetahModel(const CheetahModel<double>& d) {
  CheetahModel<T> m{};
  for (int b = 0; b < kNB; ++b) {
    m.lx[b] = (T)d.lx[b];
    m.lz[b] = (T)d.lz[b];
    m.mass[b] = (T)d.mass[b];
    m.iyy[b] = (T)d.iyy[b];
    m.cx[b] = (T)d.cx[b];
    m.cz[b] = (T)d.cz[b];
    m.body_invw[b] = (T)d.body_invw[b];
    m.bmu[b] = (T)d.bmu[b];
  }
  for (int e = 0; e < kNEnd; ++e) {
    m.ex[e] = (T)d.ex[e];
    m.ez[e] = (T)d.ez[e];
    m.er[e] = (T)d.er[e];
  }
  for (int j = 0; j < kNU; ++j) {
    m.stiff[j] = (T)d.stiff[j];
    m.damp[j] = (T)d.damp[j];
    m.arm[j] = (T)d.arm[j];
    m.lo[j] = (T)d.lo[j];
    m.hi[j] = (T)d.hi[j];
    m.gear[j] = (T)d.gear[j];
    m.dof_invw[j] = (T)d.dof_invw[j];
  }
  m.total_mass = (T)d.total_mass;
  m.con_K = (T)d.con_K;
  m.con_B = (T)d.con_B;
  m.con_d0 = (T)d.con_d0;
  m.con_dmax = (T)d.con_dmax;
  m.con_width = (T)d.con_width;
  m.lim_K = (T)d.lim_K;
  m.lim_B = (T)d.lim_B;
  m.lim_d0 = (T)d.lim_d0;
  m.lim_dmax = (T)d.lim_dmax;
  m.lim_width = (T)d.lim_width;
  m.timestep = (T)d.timestep;
  m.gravity = (T)d.gravity;
  m.con_margin = (T)d.con_margin;
  m.n_pairs = d.n_pairs;
  return m;
}

}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_CHEETAH_MODEL_H_
