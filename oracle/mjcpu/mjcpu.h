/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * oracle/mjcpu: float64 CPU restatement of the part of the MuJoCo 3.6.0
 * pipeline that `mj_step` / `mj_forward` execute for the gym HalfCheetah, Ant,
 * Walker2d, Hopper, Swimmer, Reacher, InvertedPendulum, InvertedDoublePendulum,
 * Humanoid and HumanoidStandup models, plus the task logic around it.
 *
 * PARITY UNPINNED for the engine part: the arithmetic lives in the third-party
 * dependency google-deepmind/mujoco tag 3.6.0 (pinned in the reference at
 * envpool/workspace0.bzl:559-575), which is neither vendored under
 * /root/reference nor installed in this image, and the reference holds no
 * golden vectors for these envs.  The restatement therefore follows MuJoCo's
 * published algorithm ("Computation" chapter + engine layout; SURVEY.md
 * Appendix A) and is anchored on what IS in the reference tree:
 *   - call sites: envpool/mujoco/gym/mujoco_env.h:126-148 (mj_resetData,
 *     mj_forward, ctrl <- action, frame_skip x mj_step)
 *   - task logic: envpool/mujoco/gym/half_cheetah.h:105-185, ant.h:135-278,
 *     walker2d.h:119-203, inverted_pendulum.h:100-185,
 *     inverted_double_pendulum.h:108-186
 *   - model constants: third_party/mujoco_gym_xml_patches/half_cheetah_envpool.xml,
 *     ant_envpool.xml, walker2d_envpool.xml, walker2d_v5_envpool.xml,
 *     inverted_pendulum_envpool.xml, inverted_double_pendulum_envpool.xml
 *     (hand-transcribed in models.c, line-cited)
 * and is checked by physics invariants (tests/test_mjcpu_invariants.py).
 * tools/pin_with_mujoco.py dumps golden vectors wherever `mujoco==3.6.0` is
 * importable; tests/test_mjcpu_golden.py activates when they exist.
 */
#ifndef ORACLE_MJCPU_H_
#define ORACLE_MJCPU_H_

#define MJC_MAXBODY 16
#define MJC_MAXJNT 20
#define MJC_MAXQ 24
#define MJC_MAXV 24
#define MJC_MAXGEOM 24
#define MJC_MAXU 20
#define MJC_MAXCON 64
#define MJC_MAXEFC 256

enum { MJC_JNT_FREE = 0, MJC_JNT_SLIDE = 2, MJC_JNT_HINGE = 3 };
enum { MJC_GEOM_PLANE = 0, MJC_GEOM_SPHERE = 2, MJC_GEOM_CAPSULE = 3, MJC_GEOM_CYLINDER = 5 };
enum { MJC_INT_EULER = 0, MJC_INT_RK4 = 1 };
enum { MJC_SOL_NEWTON = 0, MJC_SOL_PGS = 1 };

typedef struct {
  int nq, nv, nu, nbody, njnt, ngeom;
  /* options */
  double timestep, gravity[3];
  double opt_density, opt_viscosity; /* <option density viscosity>: medium for the fluid forces */
  int integrator;
  int solver, iterations; /* <option solver iterations>: Newton / 100 unless the XML says otherwise */
  int disable_contact, disable_limit, disable_actuation; /* invariant tests */
  int disable_selfcollide; /* tests: drop body-body (capsule-capsule) pairs */
  /* Where qacc_warmstart is saved (SURVEY Appendix A tags the place "[M]", i.e. from memory):
   *   0 (default): at the end of every mj_fwdConstraint, so RK4 stages 2-4 warm-start from the
   *                previous STAGE ("save result for next step warmstart" in mj_fwdConstraint);
   *   1          : once per mj_step, after the integrator (all four RK4 stages warm-start from
   *                the last forward evaluation of the previous step).
   * Converged Newton does not depend on it; the 50-sweep PGS of the Humanoid models does.
   * tools/warmstart_rule_probe.py measures the difference. */
  int warmstart_rule;
  /* bodies */
  int body_parent[MJC_MAXBODY], body_rootid[MJC_MAXBODY], body_weldid[MJC_MAXBODY];
  int body_jntadr[MJC_MAXBODY], body_jntnum[MJC_MAXBODY];
  int body_dofadr[MJC_MAXBODY], body_dofnum[MJC_MAXBODY];
  double body_pos[MJC_MAXBODY][3], body_quat[MJC_MAXBODY][4];
  double body_ipos[MJC_MAXBODY][3];
  double body_mass[MJC_MAXBODY];
  double body_inertia[MJC_MAXBODY][9]; /* full 3x3 about ipos, body frame */
  double body_invweight0[MJC_MAXBODY][2];
  /* joints */
  int jnt_type[MJC_MAXJNT], jnt_body[MJC_MAXJNT];
  int jnt_qposadr[MJC_MAXJNT], jnt_dofadr[MJC_MAXJNT], jnt_limited[MJC_MAXJNT];
  double jnt_pos[MJC_MAXJNT][3], jnt_axis[MJC_MAXJNT][3];
  double jnt_range[MJC_MAXJNT][2], jnt_stiffness[MJC_MAXJNT];
  double jnt_margin[MJC_MAXJNT];
  double jnt_ref[MJC_MAXJNT]; /* <joint ref=...>: qpos0 of a slide / hinge */
  double jnt_solref[MJC_MAXJNT][2], jnt_solimp[MJC_MAXJNT][5];
  /* dofs */
  int dof_body[MJC_MAXV], dof_jnt[MJC_MAXV], dof_parent[MJC_MAXV];
  double dof_armature[MJC_MAXV], dof_damping[MJC_MAXV];
  double dof_invweight0[MJC_MAXV];
  /* geoms */
  int geom_type[MJC_MAXGEOM], geom_body[MJC_MAXGEOM];
  int geom_contype[MJC_MAXGEOM], geom_conaffinity[MJC_MAXGEOM];
  int geom_condim[MJC_MAXGEOM];
  double geom_size[MJC_MAXGEOM][3], geom_pos[MJC_MAXGEOM][3];
  double geom_quat[MJC_MAXGEOM][4], geom_friction[MJC_MAXGEOM][3];
  double geom_margin[MJC_MAXGEOM], geom_density[MJC_MAXGEOM];
  double geom_solref[MJC_MAXGEOM][2], geom_solimp[MJC_MAXGEOM][5];
  /* actuators (motors on joints) */
  int act_jnt[MJC_MAXU];
  double act_gear[MJC_MAXU], act_ctrlrange[MJC_MAXU][2];
  /* derived */
  double qpos0[MJC_MAXQ];
  double meaninertia;
  double settotalmass; /* <=0: off */
} mjc_model;

typedef struct {
  double dist, pos[3], frame[9]; /* frame rows: normal, t1, t2 */
  int geom1, geom2;
  double friction, includemargin;
  double solref[2], solimp[5];
  int dim;         /* condim: 1 frictionless (1 row), 3 pyramidal (4 rows) */
  int efc_address; /* first of its rows, -1 if outside the margin */
} mjc_contact;

typedef struct {
  /* state */
  double qpos[MJC_MAXQ], qvel[MJC_MAXV], ctrl[MJC_MAXU];
  double qacc_warmstart[MJC_MAXV], time;
  /* position-dependent */
  double xpos[MJC_MAXBODY][3], xquat[MJC_MAXBODY][4], xmat[MJC_MAXBODY][9];
  double xipos[MJC_MAXBODY][3];
  double xanchor[MJC_MAXJNT][3], xaxis[MJC_MAXJNT][3];
  double geom_xpos[MJC_MAXGEOM][3], geom_xmat[MJC_MAXGEOM][9];
  double subtree_com[MJC_MAXBODY][3];
  double cinert[MJC_MAXBODY][10];
  double cdof[MJC_MAXV][6];
  double M[MJC_MAXV][MJC_MAXV];
  int ncon;
  mjc_contact contact[MJC_MAXCON];
  int nefc;
  double efc_J[MJC_MAXEFC][MJC_MAXV];
  double efc_pos[MJC_MAXEFC], efc_margin[MJC_MAXEFC];
  double efc_D[MJC_MAXEFC], efc_R[MJC_MAXEFC], efc_aref[MJC_MAXEFC];
  double efc_vel[MJC_MAXEFC], efc_force[MJC_MAXEFC];
  double efc_diagApprox[MJC_MAXEFC];
  double efc_KBI[MJC_MAXEFC][3];
  /* velocity-dependent */
  double cvel[MJC_MAXBODY][6], cdof_dot[MJC_MAXV][6];
  double qfrc_passive[MJC_MAXV], qfrc_bias[MJC_MAXV];
  double qfrc_actuator[MJC_MAXV], qfrc_smooth[MJC_MAXV];
  double qacc_smooth[MJC_MAXV], qacc[MJC_MAXV], qfrc_constraint[MJC_MAXV];
  int solver_iter;
  /* mj_rnePostConstraint (only on request, like MuJoCo) */
  double cfrc_ext[MJC_MAXBODY][6]; /* [torque; force] about subtree_com[rootid] */
} mjc_data;

/* model.c */
void mjc_model_init(mjc_model* m);
int mjc_add_body(mjc_model* m, int parent, const double pos[3]);
int mjc_add_joint(mjc_model* m, int body, int type, const double pos[3],
                  const double axis[3], int limited, double lo, double hi,
                  double stiffness, double damping, double armature);
int mjc_add_geom(mjc_model* m, int body, int type, const double size[3],
                 const double pos[3], const double quat[4]);
int mjc_add_capsule_fromto(mjc_model* m, int body, const double from[3],
                           const double to[3], double radius);
int mjc_add_capsule_axisangle(mjc_model* m, int body, const double pos[3],
                              const double axis[3], double angle, double radius,
                              double halflen);
int mjc_add_motor(mjc_model* m, int jnt, double gear);
void mjc_compile(mjc_model* m);
/* models.c: hand-transcribed gym models */
void mjc_build_half_cheetah(mjc_model* m);
void mjc_build_ant(mjc_model* m);
void mjc_build_walker2d(mjc_model* m, int v5);
void mjc_build_inverted_pendulum(mjc_model* m);
void mjc_build_inverted_double_pendulum(mjc_model* m);
void mjc_build_reacher(mjc_model* m);
void mjc_build_swimmer(mjc_model* m);
void mjc_build_hopper(mjc_model* m);
void mjc_build_humanoid(mjc_model* m, int standup);
void mjc_build_pusher(mjc_model* m, int v5);

/* engine.c */
void mjc_reset_data(const mjc_model* m, mjc_data* d);
void mjc_forward(const mjc_model* m, mjc_data* d);
void mjc_step(const mjc_model* m, mjc_data* d);
void mjc_rne_post_constraint(const mjc_model* m, mjc_data* d); /* cfrc_ext only */
void mjc_fwd_position(const mjc_model* m, mjc_data* d);
double mjc_energy_kinetic(const mjc_model* m, mjc_data* d);
double mjc_energy_potential(const mjc_model* m, mjc_data* d);

#endif /* ORACLE_MJCPU_H_ */
