/* Declaration-only stand-in for <mujoco.h> (MuJoCo 3.6.0 is not vendored in the
 * reference and not installed here).  It exists so that the reference's task
 * headers (envpool/mujoco/gym/half_cheetah.h, ant.h ...) PARSE: the binding only
 * uses their `XxxEnvFns` / `XxxEnvSpec` (config + state/action specs).  The env
 * classes themselves (HalfCheetahEnv: mj_step on the CPU) are never instantiated
 * -- the device pool replaces them -- so no mj_* symbol is ever linked. */
#ifndef INTEGRATION_REFBIND_SHIMS_MUJOCO_H_
#define INTEGRATION_REFBIND_SHIMS_MUJOCO_H_

typedef double mjtNum;
typedef struct mjOption_ { mjtNum timestep; } mjOption;
typedef struct mjStatistic_ { mjtNum extent; mjtNum center[3]; } mjStatistic;
typedef struct mjModel_ {
  int nq, nv, nu, na, nbody, ngeom, ncam;
  mjOption opt;
  mjStatistic stat;
  mjtNum* qpos0;
  mjtNum* body_mass;
} mjModel;
typedef struct mjData_ {
  mjtNum time;
  mjtNum *qpos, *qvel, *qacc, *ctrl, *xpos, *xipos, *cfrc_ext, *cinert, *cvel;
  mjtNum *qfrc_actuator, *qfrc_constraint, *geom_xpos, *site_xpos, *subtree_com;
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
#endif /* INTEGRATION_REFBIND_SHIMS_MUJOCO_H_ */
