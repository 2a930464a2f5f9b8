/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See mjcpu.h (PARITY UNPINNED).
 *
 * Hand transcription of the gym MJCF models the reference loads
 * (envpool/mujoco/gym/mujoco_env.h:50-58 prefers the `_envpool.xml` variants):
 *   third_party/mujoco_gym_xml_patches/half_cheetah_envpool.xml
 *   third_party/mujoco_gym_xml_patches/ant_envpool.xml
 *   third_party/mujoco_gym_xml_patches/walker2d_envpool.xml (v3/v4),
 *   walker2d_v5_envpool.xml (v5: right foot friction 1.9 instead of 0.9, :47)
 *   third_party/mujoco_gym_xml_patches/inverted_pendulum_envpool.xml,
 *   inverted_double_pendulum_envpool.xml, reacher_envpool.xml, swimmer_envpool.xml,
 *   hopper_envpool.xml
 * Numbers are cited by XML line (":NN").
 */
#include <math.h>
#include <string.h>

#include "mjcpu.h"

static const double kZero3[3] = {0, 0, 0};

/* ---- HalfCheetah ------------------------------------------------------------ */
static int cheetah_geom_defaults(mjc_model* m, int g) {
  /* <geom conaffinity="0" condim="3" contype="1" friction=".4 .1 .1"
   *  solimp="0.0 0.8 0.01" solref="0.02 1"/>  :55 */
  m->geom_conaffinity[g] = 0;
  m->geom_contype[g] = 1;
  m->geom_condim[g] = 3;
  m->geom_friction[g][0] = 0.4;
  m->geom_friction[g][1] = 0.1;
  m->geom_friction[g][2] = 0.1;
  m->geom_solimp[g][0] = 0.0;
  m->geom_solimp[g][1] = 0.8;
  m->geom_solimp[g][2] = 0.01;
  m->geom_solref[g][0] = 0.02;
  m->geom_solref[g][1] = 1;
  return g;
}

static int cheetah_hinge(mjc_model* m, int body, double lo, double hi,
                         double stiffness, double damping) {
  /* <joint armature=".1" damping=".01" limited="true" solimplimit="0 .8 .03"
   *  solreflimit=".02 1" stiffness="8"/>  :54, per-joint overrides :79-97 */
  const double axis[3] = {0, 1, 0};
  int j = mjc_add_joint(m, body, MJC_JNT_HINGE, kZero3, axis, 1, lo, hi,
                        stiffness, damping, 0.1);
  m->jnt_solimp[j][0] = 0;
  m->jnt_solimp[j][1] = 0.8;
  m->jnt_solimp[j][2] = 0.03;
  m->jnt_solref[j][0] = 0.02;
  m->jnt_solref[j][1] = 1;
  return j;
}

void mjc_build_half_cheetah(mjc_model* m) {
  const double yaxis[3] = {0, 1, 0}, xaxis[3] = {1, 0, 0}, zaxis[3] = {0, 0, 1};
  mjc_model_init(m);
  m->timestep = 0.01; /* :59 */
  m->gravity[2] = -9.81;
  m->integrator = MJC_INT_EULER; /* default */
  m->settotalmass = 14;          /* :52 */
  /* floor :69 */
  {
    const double size[3] = {40, 40, 40}, quat[4] = {1, 0, 0, 0};
    int g = cheetah_geom_defaults(
        m, mjc_add_geom(m, 0, MJC_GEOM_PLANE, size, kZero3, quat));
    m->geom_conaffinity[g] = 1;
  }
  /* torso :70-77 */
  const double torso_pos[3] = {0, 0, 0.7};
  int torso = mjc_add_body(m, 0, torso_pos);
  mjc_add_joint(m, torso, MJC_JNT_SLIDE, kZero3, xaxis, 0, 0, 0, 0, 0, 0);
  mjc_add_joint(m, torso, MJC_JNT_SLIDE, kZero3, zaxis, 0, 0, 0, 0, 0, 0);
  mjc_add_joint(m, torso, MJC_JNT_HINGE, kZero3, yaxis, 0, 0, 0, 0, 0, 0);
  {
    const double from[3] = {-0.5, 0, 0}, to[3] = {0.5, 0, 0};
    cheetah_geom_defaults(m, mjc_add_capsule_fromto(m, torso, from, to, 0.046));
    const double hp[3] = {0.6, 0, 0.1};
    cheetah_geom_defaults(
        m, mjc_add_capsule_axisangle(m, torso, hp, yaxis, 0.87, 0.046, 0.15));
  }
  /* back leg :78-91 */
  const double bthigh_pos[3] = {-0.5, 0, 0};
  int bthigh = mjc_add_body(m, torso, bthigh_pos);
  int j_bthigh = cheetah_hinge(m, bthigh, -0.52, 1.05, 240, 6);
  {
    const double p[3] = {0.1, 0, -0.13};
    cheetah_geom_defaults(
        m, mjc_add_capsule_axisangle(m, bthigh, p, yaxis, -3.8, 0.046, 0.145));
  }
  const double bshin_pos[3] = {0.16, 0, -0.25};
  int bshin = mjc_add_body(m, bthigh, bshin_pos);
  int j_bshin = cheetah_hinge(m, bshin, -0.785, 0.785, 180, 4.5);
  {
    const double p[3] = {-0.14, 0, -0.07};
    cheetah_geom_defaults(
        m, mjc_add_capsule_axisangle(m, bshin, p, yaxis, -2.03, 0.046, 0.15));
  }
  const double bfoot_pos[3] = {-0.28, 0, -0.14};
  int bfoot = mjc_add_body(m, bshin, bfoot_pos);
  int j_bfoot = cheetah_hinge(m, bfoot, -0.4, 0.785, 120, 3);
  {
    const double p[3] = {0.03, 0, -0.097};
    cheetah_geom_defaults(
        m, mjc_add_capsule_axisangle(m, bfoot, p, yaxis, -0.27, 0.046, 0.094));
  }
  /* front leg :92-105 */
  const double fthigh_pos[3] = {0.5, 0, 0};
  int fthigh = mjc_add_body(m, torso, fthigh_pos);
  int j_fthigh = cheetah_hinge(m, fthigh, -1, 0.7, 180, 4.5);
  {
    const double p[3] = {-0.07, 0, -0.12};
    cheetah_geom_defaults(
        m, mjc_add_capsule_axisangle(m, fthigh, p, yaxis, 0.52, 0.046, 0.133));
  }
  const double fshin_pos[3] = {-0.14, 0, -0.24};
  int fshin = mjc_add_body(m, fthigh, fshin_pos);
  int j_fshin = cheetah_hinge(m, fshin, -1.2, 0.87, 120, 3);
  {
    const double p[3] = {0.065, 0, -0.09};
    cheetah_geom_defaults(
        m, mjc_add_capsule_axisangle(m, fshin, p, yaxis, -0.6, 0.046, 0.106));
  }
  const double ffoot_pos[3] = {0.13, 0, -0.18};
  int ffoot = mjc_add_body(m, fshin, ffoot_pos);
  int j_ffoot = cheetah_hinge(m, ffoot, -0.5, 0.5, 60, 1.5);
  {
    const double p[3] = {0.045, 0, -0.07};
    cheetah_geom_defaults(
        m, mjc_add_capsule_axisangle(m, ffoot, p, yaxis, -0.6, 0.046, 0.07));
  }
  /* actuators :108-115 */
  mjc_add_motor(m, j_bthigh, 120);
  mjc_add_motor(m, j_bshin, 90);
  mjc_add_motor(m, j_bfoot, 60);
  mjc_add_motor(m, j_fthigh, 120);
  mjc_add_motor(m, j_fshin, 60);
  mjc_add_motor(m, j_ffoot, 30);
  mjc_compile(m);
}

/* ---- Ant ----------------------------------------------------------------------- */
static int ant_geom_defaults(mjc_model* m, int g) {
  /* <geom conaffinity="0" condim="3" density="5.0" friction="1 0.5 0.5"
   *  margin="0.01"/>  :25 */
  m->geom_conaffinity[g] = 0;
  m->geom_contype[g] = 1;
  m->geom_condim[g] = 3;
  m->geom_density[g] = 5.0;
  m->geom_friction[g][0] = 1;
  m->geom_friction[g][1] = 0.5;
  m->geom_friction[g][2] = 0.5;
  m->geom_margin[g] = 0.01;
  return g;
}

static int ant_leg(mjc_model* m, int torso, double sx, double sy,
                   const double ankle_axis[3], double ankle_lo,
                   double ankle_hi, int* hip, int* ankle) {
  const double deg = 3.14159265358979323846 / 180.0; /* angle="degree" :18 */
  const double zaxis[3] = {0, 0, 1};
  /* <joint armature="1" damping="1" limited="true"/> :24 */
  int leg = mjc_add_body(m, torso, kZero3);
  const double a[3] = {0.2 * sx, 0.2 * sy, 0}, b[3] = {0.4 * sx, 0.4 * sy, 0};
  ant_geom_defaults(m, mjc_add_capsule_fromto(m, leg, kZero3, a, 0.08));
  int aux = mjc_add_body(m, leg, a);
  *hip = mjc_add_joint(m, aux, MJC_JNT_HINGE, kZero3, zaxis, 1, -30 * deg,
                       30 * deg, 0, 1, 1);
  ant_geom_defaults(m, mjc_add_capsule_fromto(m, aux, kZero3, a, 0.08));
  int foot = mjc_add_body(m, aux, a);
  *ankle = mjc_add_joint(m, foot, MJC_JNT_HINGE, kZero3, ankle_axis, 1,
                         ankle_lo * deg, ankle_hi * deg, 0, 1, 1);
  ant_geom_defaults(m, mjc_add_capsule_fromto(m, foot, kZero3, b, 0.08));
  return leg;
}

void mjc_build_ant(mjc_model* m) {
  mjc_model_init(m);
  m->timestep = 0.01;          /* :19 */
  m->integrator = MJC_INT_RK4; /* :19 */
  m->gravity[2] = -9.81;       /* MuJoCo default */
  {                            /* floor :34 */
    const double size[3] = {40, 40, 40}, quat[4] = {1, 0, 0, 0};
    int g = ant_geom_defaults(
        m, mjc_add_geom(m, 0, MJC_GEOM_PLANE, size, kZero3, quat));
    m->geom_conaffinity[g] = 1;
  }
  const double torso_pos[3] = {0, 0, 0.75}; /* :35 */
  int torso = mjc_add_body(m, 0, torso_pos);
  {
    const double size[3] = {0.25, 0, 0}, quat[4] = {1, 0, 0, 0}; /* :37 */
    ant_geom_defaults(m,
                      mjc_add_geom(m, torso, MJC_GEOM_SPHERE, size, kZero3, quat));
  }
  /* <joint armature="0" damping="0" limited="false" type="free"/> :38 */
  mjc_add_joint(m, torso, MJC_JNT_FREE, kZero3, kZero3, 0, 0, 0, 0, 0, 0);
  int hip[5], ankle[5];
  const double ax_m11[3] = {-1, 1, 0}, ax_11[3] = {1, 1, 0};
  ant_leg(m, torso, 1, 1, ax_m11, 30, 70, &hip[1], &ankle[1]);    /* :39-49 */
  ant_leg(m, torso, -1, 1, ax_11, -70, -30, &hip[2], &ankle[2]);  /* :50-60 */
  ant_leg(m, torso, -1, -1, ax_m11, -70, -30, &hip[3], &ankle[3]); /* :61-71 */
  ant_leg(m, torso, 1, -1, ax_11, 30, 70, &hip[4], &ankle[4]);    /* :72-82 */
  /* actuators :85-94, gear 150 */
  mjc_add_motor(m, hip[4], 150);
  mjc_add_motor(m, ankle[4], 150);
  mjc_add_motor(m, hip[1], 150);
  mjc_add_motor(m, ankle[1], 150);
  mjc_add_motor(m, hip[2], 150);
  mjc_add_motor(m, ankle[2], 150);
  mjc_add_motor(m, hip[3], 150);
  mjc_add_motor(m, ankle[3], 150);
  mjc_compile(m);
}

/* ---- Walker2d -------------------------------------------------------------------- */
static int walker_geom(mjc_model* m, int g, double friction) {
  /* <geom conaffinity="0" condim="3" contype="1" density="1000"
   *  friction=".7 .1 .1"/>  :27; every body geom overrides friction[0] */
  m->geom_conaffinity[g] = 0;
  m->geom_contype[g] = 1;
  m->geom_condim[g] = 3;
  m->geom_density[g] = 1000;
  m->geom_friction[g][0] = friction;
  m->geom_friction[g][1] = 0.1;
  m->geom_friction[g][2] = 0.1;
  return g;
}

/* <joint armature="0.01" damping=".1" limited="true"/> :26, axis "0 -1 0" */
static int walker_hinge(mjc_model* m, int body, const double pos[3], double lo_deg,
                        double hi_deg) {
  const double deg = 3.14159265358979323846 / 180.0; /* angle="degree" :24 */
  const double axis[3] = {0, -1, 0};
  return mjc_add_joint(m, body, MJC_JNT_HINGE, pos, axis, 1, lo_deg * deg,
                       hi_deg * deg, 0, 0.1, 0.01);
}

static void walker_leg(mjc_model* m, int torso, double foot_friction, int* j) {
  const double quat_id[4] = {1, 0, 0, 0};
  /* thigh :39-41 / :52-54 */
  const double thigh_pos[3] = {0, 0, -0.19999999999999996};
  int thigh = mjc_add_body(m, torso, thigh_pos);
  j[0] = walker_hinge(m, thigh, kZero3, -150, 0);
  {
    const double size[3] = {0.050000000000000003, 0.22500000000000003, 0};
    const double pos[3] = {0, 0, -0.22500000000000009};
    walker_geom(m, mjc_add_geom(m, thigh, MJC_GEOM_CAPSULE, size, pos, quat_id), 0.9);
  }
  /* leg :42-44 / :55-57 */
  const double leg_pos[3] = {0, 0, -0.70000000000000007};
  int leg = mjc_add_body(m, thigh, leg_pos);
  {
    const double jpos[3] = {0, 0, 0.25};
    j[1] = walker_hinge(m, leg, jpos, -150, 0);
    const double size[3] = {0.040000000000000001, 0.25, 0};
    walker_geom(m, mjc_add_geom(m, leg, MJC_GEOM_CAPSULE, size, kZero3, quat_id), 0.9);
  }
  /* foot :45-47 / :58-60 */
  const double foot_pos[3] = {0.20000000000000001, 0, -0.34999999999999998};
  int foot = mjc_add_body(m, leg, foot_pos);
  {
    const double jpos[3] = {-0.20000000000000001, 0, 0.10000000000000001};
    j[2] = walker_hinge(m, foot, jpos, -45, 45);
    const double size[3] = {0.059999999999999998, 0.10000000000000001, 0};
    const double pos[3] = {-0.10000000000000001, 0, 0.10000000000000001};
    const double quat[4] = {0.70710678118654757, 0, -0.70710678118654746, 0};
    walker_geom(m, mjc_add_geom(m, foot, MJC_GEOM_CAPSULE, size, pos, quat),
                foot_friction);
  }
}

void mjc_build_walker2d(mjc_model* m, int v5) {
  const double yaxis[3] = {0, 1, 0}, xaxis[3] = {1, 0, 0}, zaxis[3] = {0, 0, 1};
  const double quat_id[4] = {1, 0, 0, 0};
  mjc_model_init(m);
  m->timestep = 0.002;         /* :29 */
  m->integrator = MJC_INT_RK4; /* :29 */
  m->gravity[2] = -9.81;       /* MuJoCo default */
  { /* floor :32: default friction .7 .1 .1, conaffinity 1 */
    const double size[3] = {40, 40, 40};
    int g = walker_geom(m, mjc_add_geom(m, 0, MJC_GEOM_PLANE, size, kZero3, quat_id), 0.7);
    m->geom_conaffinity[g] = 1;
  }
  /* torso :33-38 */
  const double torso_pos[3] = {0, 0, 1.25};
  int torso = mjc_add_body(m, 0, torso_pos);
  const double root_pos[3] = {0, 0, -1.25};
  mjc_add_joint(m, torso, MJC_JNT_SLIDE, root_pos, xaxis, 0, 0, 0, 0, 0, 0);
  int rootz = mjc_add_joint(m, torso, MJC_JNT_SLIDE, root_pos, zaxis, 0, 0, 0, 0, 0, 0);
  m->jnt_ref[rootz] = 1.25; /* ref="1.25" :36 */
  mjc_add_joint(m, torso, MJC_JNT_HINGE, kZero3, yaxis, 0, 0, 0, 0, 0, 0);
  {
    const double size[3] = {0.050000000000000003, 0.19999999999999996, 0};
    walker_geom(m, mjc_add_geom(m, torso, MJC_GEOM_CAPSULE, size, kZero3, quat_id), 0.9);
  }
  int jr[3], jl[3];
  walker_leg(m, torso, v5 ? 1.9 : 0.9, jr); /* right: :39-50 */
  walker_leg(m, torso, 1.9, jl);            /* left: :52-63 */
  /* actuators :68-73, gear 100 */
  for (int i = 0; i < 3; ++i) mjc_add_motor(m, jr[i], 100);
  for (int i = 0; i < 3; ++i) mjc_add_motor(m, jl[i], 100);
  mjc_compile(m);
}

/* ---- InvertedPendulum / InvertedDoublePendulum -------------------------------------- */
/* every geom: <geom contype="0" friction="1 0.1 0.1"/> (:21 / :39): with the
 * default conaffinity=1 no pair passes the contype/conaffinity filter => no
 * contacts at all; the geoms only provide mass and inertia. */
static int pend_geom(mjc_model* m, int g) {
  m->geom_contype[g] = 0;
  m->geom_friction[g][0] = 1;
  m->geom_friction[g][1] = 0.1;
  m->geom_friction[g][2] = 0.1;
  return g;
}

static void pend_quat_y90(double quat[4]) { /* quat="0.707 0 0.707 0", normalised */
  const double n = sqrt(0.707 * 0.707 * 2);
  quat[0] = 0.707 / n;
  quat[1] = 0;
  quat[2] = 0.707 / n;
  quat[3] = 0;
}

void mjc_build_inverted_pendulum(mjc_model* m) {
  const double deg = 3.14159265358979323846 / 180.0; /* <compiler> default: degree */
  const double xaxis[3] = {1, 0, 0}, yaxis[3] = {0, 1, 0};
  double q90[4];
  pend_quat_y90(q90);
  mjc_model_init(m);
  m->timestep = 0.02;          /* :25 */
  m->integrator = MJC_INT_RK4; /* :25 */
  m->gravity[2] = -9.81;       /* :25 */
  { /* rail :29 (worldbody geom: mass irrelevant) */
    const double size[3] = {0.02, 1, 0};
    pend_geom(m, mjc_add_geom(m, 0, MJC_GEOM_CAPSULE, size, kZero3, q90));
  }
  /* <joint armature="0" damping="1" limited="true"/> :20 */
  int cart = mjc_add_body(m, 0, kZero3); /* :30 */
  int slider = mjc_add_joint(m, cart, MJC_JNT_SLIDE, kZero3, xaxis, 1, -1, 1, 0, 1, 0); /* :31 */
  {
    const double size[3] = {0.1, 0.1, 0}; /* :32 */
    pend_geom(m, mjc_add_geom(m, cart, MJC_GEOM_CAPSULE, size, kZero3, q90));
  }
  int pole = mjc_add_body(m, cart, kZero3); /* :33 */
  mjc_add_joint(m, pole, MJC_JNT_HINGE, kZero3, yaxis, 1, -90 * deg, 90 * deg, 0, 1, 0); /* :34 */
  {
    /* fromto="0 0 0 0.001 0 0.6" size="0.049 0.3" :35 (fromto overrides the length) */
    const double from[3] = {0, 0, 0}, to[3] = {0.001, 0, 0.6};
    pend_geom(m, mjc_add_capsule_fromto(m, pole, from, to, 0.049));
  }
  mjc_add_motor(m, slider, 100); /* :41 gear 100, ctrlrange -3 3 */
  m->act_ctrlrange[0][0] = -3;
  m->act_ctrlrange[0][1] = 3;
  mjc_compile(m);
}

void mjc_build_inverted_double_pendulum(mjc_model* m) {
  const double xaxis[3] = {1, 0, 0}, yaxis[3] = {0, 1, 0};
  const double quat_id[4] = {1, 0, 0, 0};
  double q90[4];
  pend_quat_y90(q90);
  mjc_model_init(m);
  m->timestep = 0.01;          /* :41 */
  m->integrator = MJC_INT_RK4; /* :41 */
  m->gravity[0] = 1e-5;        /* gravity="1e-5 0 -9.81" :41 */
  m->gravity[2] = -9.81;
  { /* floor :44 and rail :45 (world geoms, contype 0) */
    const double fsize[3] = {40, 40, 40}, fpos[3] = {0, 0, -3.0};
    pend_geom(m, mjc_add_geom(m, 0, MJC_GEOM_PLANE, fsize, fpos, quat_id));
    const double size[3] = {0.02, 1, 0};
    pend_geom(m, mjc_add_geom(m, 0, MJC_GEOM_CAPSULE, size, kZero3, q90));
  }
  /* <joint damping="0.05"/> :38 (armature 0, not limited unless stated) */
  int cart = mjc_add_body(m, 0, kZero3); /* :46 */
  int slider = mjc_add_joint(m, cart, MJC_JNT_SLIDE, kZero3, xaxis, 1, -1, 1, 0, 0.05, 0);
  m->jnt_margin[slider] = 0.01; /* margin="0.01" :47 */
  {
    const double size[3] = {0.1, 0.1, 0}; /* :48 */
    pend_geom(m, mjc_add_geom(m, cart, MJC_GEOM_CAPSULE, size, kZero3, q90));
  }
  int pole = mjc_add_body(m, cart, kZero3); /* :49 */
  mjc_add_joint(m, pole, MJC_JNT_HINGE, kZero3, yaxis, 0, 0, 0, 0, 0.05, 0); /* :50 */
  const double from[3] = {0, 0, 0}, to[3] = {0, 0, 0.6};
  pend_geom(m, mjc_add_capsule_fromto(m, pole, from, to, 0.045)); /* :51 */
  const double p2[3] = {0, 0, 0.6};
  int pole2 = mjc_add_body(m, pole, p2); /* :52 */
  mjc_add_joint(m, pole2, MJC_JNT_HINGE, kZero3, yaxis, 0, 0, 0, 0, 0.05, 0); /* :53 */
  pend_geom(m, mjc_add_capsule_fromto(m, pole2, from, to, 0.045)); /* :54 */
  mjc_add_motor(m, slider, 500); /* :60 gear 500, ctrlrange -1 1 */
  mjc_compile(m);
}

/* ---- Reacher ------------------------------------------------------------------------- */
void mjc_build_reacher(mjc_model* m) {
  const double xaxis[3] = {1, 0, 0}, yaxis[3] = {0, 1, 0}, zaxis[3] = {0, 0, 1};
  const double quat_id[4] = {1, 0, 0, 0};
  mjc_model_init(m);
  m->timestep = 0.01;          /* :23 */
  m->integrator = MJC_INT_RK4; /* :23 */
  m->gravity[2] = -9.81;       /* :23 */
  /* world geoms (ground, side walls, root cylinder :26-32) carry contype 0 /
   * conaffinity 0 and belong to the world body: neither mass nor contacts.
   * <joint armature="1" damping="1" limited="true"/> :20, <geom contype="0"/> :21,
   * angle="radian" :18 */
  const double p0[3] = {0, 0, 0.01};
  int body0 = mjc_add_body(m, 0, p0); /* :33 */
  const double from[3] = {0, 0, 0}, to[3] = {0.1, 0, 0};
  pend_geom(m, mjc_add_capsule_fromto(m, body0, from, to, 0.01)); /* link0 :34 */
  int j0 = mjc_add_joint(m, body0, MJC_JNT_HINGE, kZero3, zaxis, 0, 0, 0, 0, 1, 1); /* :35 */
  const double p1[3] = {0.1, 0, 0};
  int body1 = mjc_add_body(m, body0, p1); /* :36 */
  int j1 = mjc_add_joint(m, body1, MJC_JNT_HINGE, kZero3, zaxis, 1, -3.0, 3.0, 0, 1, 1); /* :37 */
  pend_geom(m, mjc_add_capsule_fromto(m, body1, from, to, 0.01)); /* link1 :38 */
  const double pf[3] = {0.11, 0, 0};
  int tip = mjc_add_body(m, body1, pf); /* fingertip :39, no joint */
  {
    const double size[3] = {0.01, 0, 0};
    pend_geom(m, mjc_add_geom(m, tip, MJC_GEOM_SPHERE, size, kZero3, quat_id)); /* :40 */
  }
  const double pt[3] = {0.1, -0.1, 0.01};
  int target = mjc_add_body(m, 0, pt); /* :44 */
  int tx = mjc_add_joint(m, target, MJC_JNT_SLIDE, kZero3, xaxis, 1, -0.27, 0.27, 0, 0, 0); /* :45 */
  int ty = mjc_add_joint(m, target, MJC_JNT_SLIDE, kZero3, yaxis, 1, -0.27, 0.27, 0, 0, 0); /* :46 */
  m->jnt_ref[tx] = 0.1;
  m->jnt_ref[ty] = -0.1;
  {
    const double size[3] = {0.009, 0, 0};
    int g = pend_geom(m, mjc_add_geom(m, target, MJC_GEOM_SPHERE, size, kZero3, quat_id)); /* :47 */
    m->geom_conaffinity[g] = 0;
  }
  mjc_add_motor(m, j0, 200); /* :51-52 */
  mjc_add_motor(m, j1, 200);
  mjc_compile(m);
}

/* ---- Swimmer -------------------------------------------------------------------------- */
static int swimmer_capsule(mjc_model* m, int body, double x0, double x1) {
  /* <geom conaffinity="0" condim="1" contype="0"/> :21, density="1000" size="0.1" */
  const double from[3] = {x0, 0, 0}, to[3] = {x1, 0, 0};
  int g = mjc_add_capsule_fromto(m, body, from, to, 0.1);
  m->geom_contype[g] = 0;
  m->geom_conaffinity[g] = 0;
  m->geom_condim[g] = 1;
  return g;
}

void mjc_build_swimmer(mjc_model* m) {
  const double deg = 3.14159265358979323846 / 180.0; /* angle="degree" :18 */
  const double xaxis[3] = {1, 0, 0}, yaxis[3] = {0, 1, 0}, zaxis[3] = {0, 0, 1};
  mjc_model_init(m);
  m->timestep = 0.01;          /* :19 */
  m->integrator = MJC_INT_RK4; /* :19 */
  m->gravity[2] = -9.81;       /* MuJoCo default */
  m->opt_density = 4000;       /* :19 */
  m->opt_viscosity = 0.1;      /* :19 */
  { /* floor :32: contype 0 (default :21) => never collides */
    const double size[3] = {40, 40, 0.1}, pos[3] = {0, 0, -0.1}, quat[4] = {1, 0, 0, 0};
    int g = mjc_add_geom(m, 0, MJC_GEOM_PLANE, size, pos, quat);
    m->geom_contype[g] = 0;
    m->geom_conaffinity[g] = 0;
  }
  /* <joint armature='0.1'/> :22 */
  int torso = mjc_add_body(m, 0, kZero3); /* :34 */
  swimmer_capsule(m, torso, 1.5, 0.5);    /* :36 */
  mjc_add_joint(m, torso, MJC_JNT_SLIDE, kZero3, xaxis, 0, 0, 0, 0, 0, 0.1); /* slider1 :37 */
  mjc_add_joint(m, torso, MJC_JNT_SLIDE, kZero3, yaxis, 0, 0, 0, 0, 0, 0.1); /* slider2 :38 */
  mjc_add_joint(m, torso, MJC_JNT_HINGE, kZero3, zaxis, 0, 0, 0, 0, 0, 0.1); /* free_body_rot :39 */
  const double pm[3] = {0.5, 0, 0};
  int mid = mjc_add_body(m, torso, pm); /* :40 */
  swimmer_capsule(m, mid, 0, -1);       /* :41 */
  int j1 = mjc_add_joint(m, mid, MJC_JNT_HINGE, kZero3, zaxis, 1, -100 * deg, 100 * deg, 0, 0, 0.1);
  const double pb[3] = {-1, 0, 0};
  int back = mjc_add_body(m, mid, pb); /* :43 */
  swimmer_capsule(m, back, 0, -1);     /* :44 */
  int j2 = mjc_add_joint(m, back, MJC_JNT_HINGE, kZero3, zaxis, 1, -100 * deg, 100 * deg, 0, 0, 0.1);
  mjc_add_motor(m, j1, 150); /* :50-51 */
  mjc_add_motor(m, j2, 150);
  mjc_compile(m);
}

/* ---- Hopper --------------------------------------------------------------------------- */
static int hopper_geom(mjc_model* m, int g, double friction) {
  /* <geom conaffinity="1" condim="1" contype="1" margin="0.001" solimp=".8 .8 .01"
   *  solref=".02 1"/> :26; body geoms set friction[0] only (the rest stays at
   *  MuJoCo's 0.005 0.0001) */
  m->geom_conaffinity[g] = 1;
  m->geom_contype[g] = 1;
  m->geom_condim[g] = 1;
  m->geom_margin[g] = 0.001;
  m->geom_solimp[g][0] = 0.8;
  m->geom_solimp[g][1] = 0.8;
  m->geom_solimp[g][2] = 0.01;
  m->geom_solref[g][0] = 0.02;
  m->geom_solref[g][1] = 1;
  m->geom_friction[g][0] = friction;
  return g;
}

void mjc_build_hopper(mjc_model* m) {
  const double deg = 3.14159265358979323846 / 180.0; /* angle="degree" :23 */
  const double yaxis[3] = {0, 1, 0}, xaxis[3] = {1, 0, 0}, zaxis[3] = {0, 0, 1};
  const double naxis[3] = {0, -1, 0};
  const double quat_id[4] = {1, 0, 0, 0};
  mjc_model_init(m);
  m->timestep = 0.002;         /* :29 */
  m->integrator = MJC_INT_RK4; /* :29 */
  m->gravity[2] = -9.81;
  { /* floor :35: class defaults + condim 3, friction MuJoCo default 1 */
    const double size[3] = {20, 20, 0.125};
    int g = hopper_geom(m, mjc_add_geom(m, 0, MJC_GEOM_PLANE, size, kZero3, quat_id), 1.0);
    m->geom_condim[g] = 3;
  }
  /* <joint armature="1" damping="1" limited="true"/> :25 */
  const double torso_pos[3] = {0, 0, 1.25};
  int torso = mjc_add_body(m, 0, torso_pos); /* :36 */
  const double root_pos[3] = {0, 0, -1.25};
  mjc_add_joint(m, torso, MJC_JNT_SLIDE, root_pos, xaxis, 0, 0, 0, 0, 0, 0); /* :38 */
  int rootz = mjc_add_joint(m, torso, MJC_JNT_SLIDE, root_pos, zaxis, 0, 0, 0, 0, 0, 0);
  m->jnt_ref[rootz] = 1.25; /* :39 */
  mjc_add_joint(m, torso, MJC_JNT_HINGE, kZero3, yaxis, 0, 0, 0, 0, 0, 0); /* :40 */
  {
    const double size[3] = {0.05, 0.19999999999999996, 0}; /* :41 */
    hopper_geom(m, mjc_add_geom(m, torso, MJC_GEOM_CAPSULE, size, kZero3, quat_id), 0.9);
  }
  const double thigh_pos[3] = {0, 0, -0.19999999999999996};
  int thigh = mjc_add_body(m, torso, thigh_pos); /* :42 */
  int j_thigh = mjc_add_joint(m, thigh, MJC_JNT_HINGE, kZero3, naxis, 1, -150 * deg, 0, 0, 1, 1);
  {
    const double size[3] = {0.05, 0.22500000000000003, 0}, pos[3] = {0, 0, -0.22500000000000009};
    hopper_geom(m, mjc_add_geom(m, thigh, MJC_GEOM_CAPSULE, size, pos, quat_id), 0.9); /* :44 */
  }
  const double leg_pos[3] = {0, 0, -0.70000000000000007};
  int leg = mjc_add_body(m, thigh, leg_pos); /* :45 */
  const double leg_jpos[3] = {0, 0, 0.25};
  int j_leg = mjc_add_joint(m, leg, MJC_JNT_HINGE, leg_jpos, naxis, 1, -150 * deg, 0, 0, 1, 1);
  {
    const double size[3] = {0.04, 0.25, 0};
    hopper_geom(m, mjc_add_geom(m, leg, MJC_GEOM_CAPSULE, size, kZero3, quat_id), 0.9); /* :47 */
  }
  const double foot_pos[3] = {0.13, 0, -0.35};
  int foot = mjc_add_body(m, leg, foot_pos); /* :48 */
  const double foot_jpos[3] = {-0.13, 0, 0.1};
  int j_foot = mjc_add_joint(m, foot, MJC_JNT_HINGE, foot_jpos, naxis, 1, -45 * deg, 45 * deg, 0, 1, 1);
  {
    const double size[3] = {0.06, 0.195, 0}, pos[3] = {-0.065, 0, 0.1};
    const double quat[4] = {0.70710678118654757, 0, -0.70710678118654746, 0};
    hopper_geom(m, mjc_add_geom(m, foot, MJC_GEOM_CAPSULE, size, pos, quat), 2.0); /* :50 */
  }
  mjc_add_motor(m, j_thigh, 200); /* :56-58 */
  mjc_add_motor(m, j_leg, 200);
  mjc_add_motor(m, j_foot, 200);
  mjc_compile(m);
}

/* ---- Humanoid / HumanoidStandup ------------------------------------------------
 * humanoid_envpool.xml (line numbers below) and humanoidstandup_envpool.xml, which is
 * the same kinematic tree laid on its back: only body / geom placements along the spine
 * and the legs differ (and left_hip_y's lower range, :72), the diff is spelled out at
 * each use of `su`.
 *   <compiler angle="degree" inertiafromgeom="true"/>                             :18
 *   <joint armature="1" damping="1" limited="true"/>                              :20
 *   <geom conaffinity="1" condim="1" contype="1" margin="0.001" .../>             :21
 *   <motor ctrllimited="true" ctrlrange="-.4 .4"/>                                :22
 *   <option integrator="RK4" iterations="50" solver="PGS" timestep="0.003">       :24
 * The two fixed tendons (:107-116) have no limit, spring or actuator: no effect. */
static int hum_geom(mjc_model* m, int g) {
  m->geom_conaffinity[g] = 1;
  m->geom_contype[g] = 1;
  m->geom_condim[g] = 1;
  m->geom_margin[g] = 0.001;
  return g;
}
static int hum_capsule(mjc_model* m, int body, double x0, double y0, double z0, double x1,
                       double y1, double z1, double radius) {
  const double from[3] = {x0, y0, z0}, to[3] = {x1, y1, z1};
  return hum_geom(m, mjc_add_capsule_fromto(m, body, from, to, radius));
}
static int hum_sphere(mjc_model* m, int body, double x, double y, double z, double radius) {
  const double size[3] = {radius, 0, 0}, pos[3] = {x, y, z}, quat[4] = {1, 0, 0, 0};
  return hum_geom(m, mjc_add_geom(m, body, MJC_GEOM_SPHERE, size, pos, quat));
}
static int hum_hinge(mjc_model* m, int body, double px, double py, double pz, double ax,
                     double ay, double az, double lo_deg, double hi_deg, double stiffness,
                     double damping, double armature) {
  const double pos[3] = {px, py, pz}, axis[3] = {ax, ay, az};
  const double d2r = 3.14159265358979323846 / 180.0;
  return mjc_add_joint(m, body, MJC_JNT_HINGE, pos, axis, 1, lo_deg * d2r, hi_deg * d2r,
                       stiffness, damping, armature);
}
static int hum_body(mjc_model* m, int parent, double x, double y, double z) {
  const double pos[3] = {x, y, z};
  return mjc_add_body(m, parent, pos);
}
static void hum_quat(mjc_model* m, int b) { /* quat="1.000 0 -0.002 0", normalised :50,:54 */
  double n = sqrt(1.0 + 0.002 * 0.002);
  m->body_quat[b][0] = 1.0 / n;
  m->body_quat[b][1] = 0;
  m->body_quat[b][2] = -0.002 / n;
  m->body_quat[b][3] = 0;
}

void mjc_build_humanoid(mjc_model* m, int su) {
  const double size0[3] = {20, 20, 0.125}, quat0[4] = {1, 0, 0, 0};
  mjc_model_init(m);
  m->timestep = 0.003;          /* :24 */
  m->integrator = MJC_INT_RK4;  /* :24 */
  m->solver = MJC_SOL_PGS;      /* :24 */
  m->iterations = 50;           /* :24 */
  /* floor :41 (condim 3, friction 1 .1 .1; the class default gives it margin 0.001) */
  int floor = hum_geom(m, mjc_add_geom(m, 0, MJC_GEOM_PLANE, size0, kZero3, quat0));
  m->geom_condim[floor] = 3;
  m->geom_friction[floor][0] = 1;
  m->geom_friction[floor][1] = 0.1;
  m->geom_friction[floor][2] = 0.1;
  /* torso :43-48 (standup: pos 0 0 .105, head at -.15 0 0, uwaist at x=.11) */
  int torso = hum_body(m, 0, 0, 0, su ? 0.105 : 1.4);
  {
    const double axis[3] = {0, 0, 1};
    mjc_add_joint(m, torso, MJC_JNT_FREE, kZero3, axis, 0, 0, 0, 0, 0, 0); /* :45 */
  }
  hum_capsule(m, torso, 0, -.07, 0, 0, .07, 0, 0.07); /* torso1 :46 */
  if (su) {
    hum_sphere(m, torso, -.15, 0, 0, .09);                   /* head */
    hum_capsule(m, torso, .11, -.06, 0, .11, .06, 0, 0.06);  /* uwaist */
  } else {
    hum_sphere(m, torso, 0, 0, .19, .09);                       /* head :47 */
    hum_capsule(m, torso, -.01, -.06, -.12, -.01, .06, -.12, 0.06); /* uwaist :48 */
  }
  /* lwaist :49-52 */
  int lwaist = su ? hum_body(m, torso, .21, 0, 0) : hum_body(m, torso, -.01, 0, -0.260);
  hum_quat(m, lwaist);
  hum_capsule(m, lwaist, 0, -.06, 0, 0, .06, 0, 0.06);
  hum_hinge(m, lwaist, 0, 0, 0.065, 0, 0, 1, -45, 45, 20, 5, 0.02); /* abdomen_z :51 */
  hum_hinge(m, lwaist, 0, 0, 0.065, 0, 1, 0, -75, 30, 10, 5, 0.02); /* abdomen_y :52 */
  /* pelvis :53-55 */
  int pelvis = su ? hum_body(m, lwaist, 0.165, 0, 0) : hum_body(m, lwaist, 0, 0, -0.165);
  hum_quat(m, pelvis);
  hum_hinge(m, pelvis, 0, 0, 0.1, 1, 0, 0, -35, 35, 10, 5, 0.02); /* abdomen_x :54 */
  hum_capsule(m, pelvis, -.02, -.07, 0, -.02, .07, 0, 0.09);      /* butt :55 */
  /* legs: right :56-68, left :69-81 */
  for (int side = 0; side < 2; ++side) {
    double s = side == 0 ? -1.0 : 1.0; /* right leg is at y = -0.1 */
    int thigh = hum_body(m, pelvis, 0, s * 0.1, su ? 0 : -0.04);
    if (side == 0) {
      hum_hinge(m, thigh, 0, 0, 0, 1, 0, 0, -25, 5, 10, 5, 0.01);    /* right_hip_x :57 */
      hum_hinge(m, thigh, 0, 0, 0, 0, 0, 1, -60, 35, 10, 5, 0.01);   /* right_hip_z :58 */
      hum_hinge(m, thigh, 0, 0, 0, 0, 1, 0, -110, 20, 20, 5, 0.008); /* right_hip_y :59 */
    } else {
      hum_hinge(m, thigh, 0, 0, 0, -1, 0, 0, -25, 5, 10, 5, 0.01);  /* left_hip_x :70 */
      hum_hinge(m, thigh, 0, 0, 0, 0, 0, -1, -60, 35, 10, 5, 0.01); /* left_hip_z :71 */
      hum_hinge(m, thigh, 0, 0, 0, 0, 1, 0, su ? -120 : -110, 20, 20, 5, 0.01); /* left_hip_y :72 */
    }
    int shin;
    if (su) {
      hum_capsule(m, thigh, 0, 0, 0, 0.34, -s * 0.01, 0, 0.06);
      shin = hum_body(m, thigh, 0.403, -s * 0.01, 0);
    } else {
      hum_capsule(m, thigh, 0, 0, 0, 0, -s * 0.01, -.34, 0.06); /* thigh1 :60 / :73 */
      shin = hum_body(m, thigh, 0, -s * 0.01, -0.403);          /* :61 / :74 */
    }
    /* knee :62 (right: no stiffness) / :75 (left: stiffness 1); damping is the class default 1 */
    hum_hinge(m, shin, 0, 0, .02, 0, -1, 0, -160, -2, side == 0 ? 0 : 1, 1, 0.006);
    int foot;
    if (su) {
      hum_capsule(m, shin, 0, 0, 0, 0.3, 0, 0, 0.049);
      foot = hum_body(m, shin, 0.35, 0, -.10);
    } else {
      hum_capsule(m, shin, 0, 0, 0, 0, 0, -.3, 0.049); /* shin1 :63 */
      foot = hum_body(m, shin, 0, 0, -0.45);           /* :64 */
    }
    hum_sphere(m, foot, 0, 0, 0.1, 0.075); /* :65 */
  }
  /* arms: right :84-94, left :95-104 */
  for (int side = 0; side < 2; ++side) {
    double s = side == 0 ? -1.0 : 1.0;
    int uarm = hum_body(m, torso, 0, s * 0.17, 0.06);
    if (side == 0) {
      hum_hinge(m, uarm, 0, 0, 0, 2, 1, 1, -85, 60, 1, 1, 0.0068);  /* right_shoulder1 :85 */
      hum_hinge(m, uarm, 0, 0, 0, 0, -1, 1, -85, 60, 1, 1, 0.0051); /* right_shoulder2 :86 */
    } else {
      hum_hinge(m, uarm, 0, 0, 0, 2, -1, 1, -60, 85, 1, 1, 0.0068); /* left_shoulder1 :96 */
      hum_hinge(m, uarm, 0, 0, 0, 0, 1, 1, -60, 85, 1, 1, 0.0051);  /* left_shoulder2 :97 */
    }
    hum_capsule(m, uarm, 0, 0, 0, .16, s * .16, -.16, 0.04); /* uarm1 :87 / :98 */
    int larm = hum_body(m, uarm, .18, s * .18, -.18);        /* :88 / :99 */
    if (side == 0) {
      hum_hinge(m, larm, 0, 0, 0, 0, -1, 1, -90, 50, 0, 1, 0.0028); /* right_elbow :89 */
    } else {
      hum_hinge(m, larm, 0, 0, 0, 0, -1, -1, -90, 50, 0, 1, 0.0028); /* left_elbow :100 */
    }
    hum_capsule(m, larm, 0.01, -s * 0.01, 0.01, .17, -s * .17, .17, 0.031); /* larm :90 / :101 */
    hum_sphere(m, larm, .18, -s * .18, .18, 0.04);                          /* hand :91 / :102 */
  }
  /* actuators :118-136: joint ids are 0 root, 1 abdomen_z, 2 abdomen_y, 3 abdomen_x,
   * 4-6 right hip x z y, 7 right knee, 8-10 left hip x z y, 11 left knee,
   * 12-14 right shoulder1 shoulder2 elbow, 15-17 left */
  static const int jnt[17] = {2, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17};
  static const double gear[17] = {100, 100, 100, 100, 100, 300, 200, 100, 100,
                                  300, 200, 25,  25,  25,  25,  25,  25};
  for (int u = 0; u < 17; ++u) {
    int a = mjc_add_motor(m, jnt[u], gear[u]);
    m->act_ctrlrange[a][0] = -0.4;
    m->act_ctrlrange[a][1] = 0.4;
  }
  mjc_compile(m);
}

/* ---- Pusher ----------------------------------------------------------------------------
 * third_party/mujoco_gym_xml_patches/pusher_envpool.xml (v2 / v4) and pusher_v5_envpool.xml
 * (v5: the object's sphere geom is gone and its cylinder has density 0.01 instead of 1e-5),
 * line numbers of the former.  <default>: joint armature 0.04 damping 1 limited; geom friction
 * .8 .1 .1 density 300 margin 0.002 condim 1 contype 0 conaffinity 0 (:22-25).
 * <option timestep 0.01 gravity 0 iterations 20 integrator Euler> (:20). */
static int pusher_geom(mjc_model* m, int g, int con) {
  m->geom_friction[g][0] = 0.8;
  m->geom_friction[g][1] = 0.1;
  m->geom_friction[g][2] = 0.1;
  m->geom_density[g] = 300;
  m->geom_margin[g] = 0.002;
  m->geom_condim[g] = 1;
  m->geom_contype[g] = con;
  m->geom_conaffinity[g] = con;
  return g;
}
static void pusher_sphere(mjc_model* m, int body, double x, double y, double z, double r) {
  const double size[3] = {r, 0, 0}, pos[3] = {x, y, z}, quat[4] = {1, 0, 0, 0};
  pusher_geom(m, mjc_add_geom(m, body, MJC_GEOM_SPHERE, size, pos, quat), 0);
}
static int pusher_capsule(mjc_model* m, int body, double x0, double y0, double z0, double x1,
                          double y1, double z1, double r, int con) {
  const double from[3] = {x0, y0, z0}, to[3] = {x1, y1, z1};
  return pusher_geom(m, mjc_add_capsule_fromto(m, body, from, to, r), con);
}
void mjc_build_pusher(mjc_model* m, int v5) {
  const double xaxis[3] = {1, 0, 0}, yaxis[3] = {0, 1, 0}, zaxis[3] = {0, 0, 1};
  const double quat_id[4] = {1, 0, 0, 0};
  const double arm = 0.04;
  mjc_model_init(m);
  m->timestep = 0.01; /* :20 */
  m->integrator = MJC_INT_EULER;
  m->gravity[0] = m->gravity[1] = m->gravity[2] = 0;
  m->iterations = 20;
  { /* table :29 */
    const double size[3] = {1, 1, 0.1}, pos[3] = {0, 0.5, -0.325};
    pusher_geom(m, mjc_add_geom(m, 0, MJC_GEOM_PLANE, size, pos, quat_id), 1);
  }
  const double p_pan[3] = {0, -0.6, 0};
  int pan = mjc_add_body(m, 0, p_pan); /* r_shoulder_pan_link :31 */
  pusher_sphere(m, pan, -0.06, 0.05, 0.2, 0.05); /* e1 :32 */
  pusher_sphere(m, pan, 0.06, 0.05, 0.2, 0.05);  /* e2 */
  pusher_sphere(m, pan, -0.06, 0.09, 0.2, 0.03); /* e1p */
  pusher_sphere(m, pan, 0.06, 0.09, 0.2, 0.03);  /* e2p */
  pusher_capsule(m, pan, 0, 0, -0.4, 0, 0, 0.2, 0.1, 0); /* sp :36 */
  int j1 = mjc_add_joint(m, pan, MJC_JNT_HINGE, kZero3, zaxis, 1, -2.2854, 1.714602, 0, 1.0, arm); /* :37 */
  const double p_lift[3] = {0.1, 0, 0};
  int lift = mjc_add_body(m, pan, p_lift); /* :39 */
  pusher_capsule(m, lift, 0, -0.1, 0, 0, 0.1, 0, 0.1, 0); /* sl */
  int j2 = mjc_add_joint(m, lift, MJC_JNT_HINGE, kZero3, yaxis, 1, -0.5236, 1.3963, 0, 1.0, arm);
  int uroll = mjc_add_body(m, lift, kZero3); /* r_upper_arm_roll_link :43 */
  pusher_capsule(m, uroll, -0.1, 0, 0, 0.1, 0, 0, 0.02, 0); /* uar */
  int j3 = mjc_add_joint(m, uroll, MJC_JNT_HINGE, kZero3, xaxis, 1, -1.5, 1.7, 0, 0.1, arm);
  int ua = mjc_add_body(m, uroll, kZero3); /* r_upper_arm_link :47, no joint */
  pusher_capsule(m, ua, 0, 0, 0, 0.4, 0, 0, 0.06, 0);
  const double p_elbow[3] = {0.4, 0, 0};
  int elbow = mjc_add_body(m, ua, p_elbow); /* :50 */
  pusher_capsule(m, elbow, 0, -0.02, 0, 0, 0.02, 0, 0.06, 0); /* ef */
  int j4 = mjc_add_joint(m, elbow, MJC_JNT_HINGE, kZero3, yaxis, 1, -2.3213, 0, 0, 0.1, arm);
  int froll = mjc_add_body(m, elbow, kZero3); /* r_forearm_roll_link :54 */
  pusher_capsule(m, froll, -0.1, 0, 0, 0.1, 0, 0, 0.02, 0); /* fr */
  int j5 = mjc_add_joint(m, froll, MJC_JNT_HINGE, kZero3, xaxis, 1, -1.5, 1.5, 0, 0.1, arm);
  int fa = mjc_add_body(m, froll, kZero3); /* r_forearm_link :58, no joint */
  pusher_capsule(m, fa, 0, 0, 0, 0.291, 0, 0, 0.05, 0);
  const double p_wrist[3] = {0.321, 0, 0};
  int wflex = mjc_add_body(m, fa, p_wrist); /* :61 */
  pusher_capsule(m, wflex, 0, -0.02, 0, 0, 0.02, 0, 0.01, 0); /* wf */
  int j6 = mjc_add_joint(m, wflex, MJC_JNT_HINGE, kZero3, yaxis, 1, -1.094, 0, 0, 0.1, arm);
  int wroll = mjc_add_body(m, wflex, kZero3); /* r_wrist_roll_link :65 */
  int j7 = mjc_add_joint(m, wroll, MJC_JNT_HINGE, kZero3, xaxis, 1, -1.5, 1.5, 0, 0.1, arm);
  /* geoms are numbered body by body: the three colliding capsules of the wrist (:71-73) come
   * before the spheres of its child tips_arm (:67-70) */
  pusher_capsule(m, wroll, 0, -0.1, 0, 0, 0.1, 0, 0.02, 1);
  pusher_capsule(m, wroll, 0, -0.1, 0, 0.1, -0.1, 0, 0.02, 1);
  pusher_capsule(m, wroll, 0, 0.1, 0, 0.1, 0.1, 0, 0.02, 1);
  int tips = mjc_add_body(m, wroll, kZero3); /* tips_arm :67, no joint */
  pusher_sphere(m, tips, 0.1, -0.1, 0, 0.01);
  pusher_sphere(m, tips, 0.1, 0.1, 0, 0.01);
  const double p_obj[3] = {0.45, -0.05, -0.275};
  int obj = mjc_add_body(m, 0, p_obj); /* object :85 */
  if (!v5) { /* :86 sphere, density 1e-5, conaffinity 0 (contype default 0) */
    const double size[3] = {0.05, 0, 0};
    int g = pusher_geom(m, mjc_add_geom(m, obj, MJC_GEOM_SPHERE, size, kZero3, quat_id), 0);
    m->geom_density[g] = 0.00001;
  }
  { /* :87 cylinder size 0.05 0.05, contype 1 conaffinity 0 */
    const double size[3] = {0.05, 0.05, 0};
    int g = pusher_geom(m, mjc_add_geom(m, obj, MJC_GEOM_CYLINDER, size, kZero3, quat_id), 0);
    m->geom_density[g] = v5 ? 0.01 : 0.00001;
    m->geom_contype[g] = 1;
    m->geom_conaffinity[g] = 0;
  }
  mjc_add_joint(m, obj, MJC_JNT_SLIDE, kZero3, yaxis, 1, -10.3213, 10.3, 0, 0.5, arm); /* obj_slidey :88 */
  mjc_add_joint(m, obj, MJC_JNT_SLIDE, kZero3, xaxis, 1, -10.3213, 10.3, 0, 0.5, arm); /* obj_slidex :89 */
  const double p_goal[3] = {0.45, -0.05, -0.3230};
  int goal = mjc_add_body(m, 0, p_goal); /* :92 */
  {
    const double size[3] = {0.08, 0.001, 0};
    int g = pusher_geom(m, mjc_add_geom(m, goal, MJC_GEOM_CYLINDER, size, kZero3, quat_id), 0);
    m->geom_density[g] = 0.00001;
  }
  mjc_add_joint(m, goal, MJC_JNT_SLIDE, kZero3, yaxis, 1, -10.3213, 10.3, 0, 0.5, arm); /* :94 */
  mjc_add_joint(m, goal, MJC_JNT_SLIDE, kZero3, xaxis, 1, -10.3213, 10.3, 0, 0.5, arm); /* :95 */
  const int jj[7] = {j1, j2, j3, j4, j5, j6, j7}; /* motors :99-107: gear 1, ctrlrange -2 2 */
  for (int i = 0; i < 7; ++i) {
    int u = mjc_add_motor(m, jj[i], 1.0);
    m->act_ctrlrange[u][0] = -2;
    m->act_ctrlrange[u][1] = 2;
  }
  mjc_compile(m);
}
