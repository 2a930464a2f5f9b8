/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Stand-in for <mujoco.h> (google-deepmind/mujoco 3.6.0 is neither vendored under
 * /root/reference nor installed in this image).  It carries exactly the part of the MuJoCo API
 * that the reference's gym task wrappers touch:
 *   functions  mj_loadXML mj_makeData mj_resetData mj_forward mj_step mj_rnePostConstraint
 *              mj_deleteModel mj_deleteData mj_name2id mjv_defaultCamera
 *              (envpool/mujoco/gym/mujoco_env.h:82-148, 218-240; ant.h:115; pusher.h:97-99;
 *              reacher.h:97-98)
 *   mjModel    nq nv nu nbody ngeom opt.timestep stat.extent body_mass
 *   mjData     qpos qvel ctrl xpos xipos site_xpos geom_xpos cvel cinert cfrc_ext
 *              qfrc_actuator qfrc_constraint
 * Two users:
 *   - oracle/ref_mujoco_driver.cc + ref_mujoco_shim.cc give the functions BODIES that forward
 *     to oracle/mjcpu, so the reference's own HalfCheetahEnv ... HumanoidStandupEnv run inside
 *     its own AsyncEnvPool: reset draw order through libstdc++'s real distributions, reward /
 *     healthy / termination / observation / info assembly, post_constraint and frame_stack are
 *     then THE REFERENCE'S CODE; only the engine arithmetic underneath is the restatement
 *     (oracle/mjcpu/engine.c, parity unpinned -- see mjcpu.h).
 *   - integration/refbind only needs the declarations (its env classes are never instantiated).
 */
#ifndef ORACLE_REF_SHIMS_MUJOCO_MUJOCO_H_
#define ORACLE_REF_SHIMS_MUJOCO_MUJOCO_H_

typedef double mjtNum;
typedef struct mjOption_ { mjtNum timestep; } mjOption;
typedef struct mjStatistic_ { mjtNum extent; mjtNum center[3]; } mjStatistic;
typedef struct mjModel_ {
  int nq, nv, nu, na, nbody, ngeom, ncam;
  mjOption opt;
  mjStatistic stat;
  mjtNum* qpos0;
  mjtNum* body_mass;
  void* impl; /* shim: the compiled oracle/mjcpu model + body-name table */
} mjModel;
typedef struct mjData_ {
  mjtNum time;
  mjtNum *qpos, *qvel, *qacc, *ctrl, *xpos, *xipos, *cfrc_ext, *cinert, *cvel;
  mjtNum *qfrc_actuator, *qfrc_constraint, *geom_xpos, *site_xpos, *subtree_com;
  void* impl; /* shim: the oracle/mjcpu data block the pointers above point into */
} mjData;
typedef enum { mjCAMERA_FREE = 0, mjCAMERA_TRACKING, mjCAMERA_FIXED, mjCAMERA_USER } mjtCamera;
typedef enum { mjOBJ_UNKNOWN = 0, mjOBJ_BODY, mjOBJ_XBODY, mjOBJ_JOINT, mjOBJ_DOF, mjOBJ_GEOM,
               mjOBJ_SITE, mjOBJ_CAMERA } mjtObj;
typedef struct mjvCamera_ {
  int type, fixedcamid, trackbodyid;
  mjtNum lookat[3], distance, azimuth, elevation;
  int orthographic;
} mjvCamera;
typedef struct mjvOption_ { int flags[32]; } mjvOption;
typedef struct mjvPerturb_ { int select; } mjvPerturb;
typedef struct mjvScene_ { int ngeom; } mjvScene;
typedef struct mjrContext_ { int offWidth; } mjrContext;

#ifdef __cplusplus
extern "C" {
#endif
mjModel* mj_loadXML(const char* filename, const void* vfs, char* error, int error_sz);
mjData* mj_makeData(const mjModel* m);
void mj_deleteModel(mjModel* m);
void mj_deleteData(mjData* d);
void mj_resetData(const mjModel* m, mjData* d);
void mj_forward(const mjModel* m, mjData* d);
void mj_step(const mjModel* m, mjData* d);
void mj_rnePostConstraint(const mjModel* m, mjData* d);
int mj_name2id(const mjModel* m, int type, const char* name);
void mjv_defaultCamera(mjvCamera* cam);
#ifdef __cplusplus
}
#endif
#endif /* ORACLE_REF_SHIMS_MUJOCO_MUJOCO_H_ */
