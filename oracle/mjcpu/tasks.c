/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See mjcpu.h (PARITY UNPINNED for
 * the engine; the task logic below IS pinned to in-tree reference code).
 *
 * Restatement of the gym task wrappers around mj_step:
 *   MujocoEnv::{MujocoReset,MujocoStep}   envpool/mujoco/gym/mujoco_env.h:126-148
 *   HalfCheetahEnvBase::{MujocoResetModel,Reset,Step,WriteState}
 *                                         envpool/mujoco/gym/half_cheetah.h:105-185
 *   AntEnvBase::{MujocoResetModel,Reset,Step,IsHealthy,WriteState}
 *                                         envpool/mujoco/gym/ant.h:135-278
 * with the runtime bookkeeping of envpool/core/env.h:184-256 and
 * async_envpool.h:127, exposed through the `orc_*` oracle API (orc_api.c).
 * v4 registration defaults: envpool/mujoco/gym/registration.py:84-93
 * (post_constraint=False, max_episode_steps=1000, frame_skip=5).
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif
#include "../restate/rng.h"
#include "mjcpu.h"

enum { DT_I32 = 0, DT_F32 = 1, DT_F64 = 2, DT_BOOL = 3 };

typedef struct {
  mjc_data d;
  orc_mt19937 gen;
  orc_normal_state nstate; /* dist_qvel_ member: persists across resets */
  int current_step, done, elapsed_step;
  int lag_set;       /* set_state gave the lagged mass centre (humanoid tasks) */
  double lag_mc[2];
  double lag_push[5]; /* Pusher: set_state gave xpos of tips_arm (3) and object (x, y) */
} mj_env;

enum { TASK_CHEETAH = 0, TASK_ANT = 1, TASK_WALKER = 2, TASK_IPEND = 3, TASK_IDPEND = 4, TASK_REACHER = 5, TASK_SWIMMER = 6, TASK_HOPPER = 7, TASK_HUMANOID = 8, TASK_STANDUP = 9, TASK_PUSHER = 10 };

typedef struct {
  int is_ant;
  int task;
  mjc_model m;
  int num_envs, max_episode_steps, frame_skip;
  double ctrl_cost_weight, forward_reward_weight, reset_noise_scale;
  double healthy_reward, healthy_z_min, healthy_z_max;
  double healthy_angle_min, healthy_angle_max, velocity_min, velocity_max;
  int terminate_when_unhealthy, legacy_healthy_reward;
  int reward_if_not_terminated, constraint_obs_dim; /* inverted pendulums */
  /* Ant-v3 / v5 (gym/registration.py:39-46) */
  int use_contact_force, post_constraint, exclude_worldbody;
  int exclude_root_actuator; /* Humanoid-v5: humanoid.h:37-38 */
  double contact_cost_max;
  /* Reacher (reacher.h:32-43) */
  int reward_after_step, obs_include_z, target;
  double dist_cost_weight, reset_qpos_scale, reset_qvel_scale, reset_goal_scale;
  double contact_cost_weight, contact_force_min, contact_force_max;
  double observation_min, observation_max;
  int torso;
  /* Pusher (pusher.h:33-46): bodies and the cylinder placement of MujocoResetModel */
  int tips, object, goal, weighted_reward_info;
  double near_cost_weight, cyl_x_min, cyl_x_max, cyl_y_min, cyl_y_max, cyl_dist_min;
  mj_env* envs;
  int nkeys;
  const char* key_names[24];
  int key_dtype[24], key_elems[24];
} mj_pool;

static const char* kCommonNames[8] = {
    "info:env_id", "info:players.env_id", "elapsed_step", "done",
    "reward",      "discount",            "step_type",    "trunc"};
static const int kCommonDtype[8] = {DT_I32, DT_I32, DT_I32, DT_BOOL,
                                    DT_F32, DT_F32, DT_I32, DT_BOOL};

static double extra_or(const double* e, int n, int i, double d) {
  return (e && i < n) ? e[i] : d;
}

/* extra (optional): [frame_skip, ctrl_cost_weight, forward_reward_weight,
 * reset_noise_scale, disable_contact, disable_limit, disable_actuation] */
void* mjcpu_create(const char* task, int num_envs, int seed,
                   int max_episode_steps, const double* extra, int n_extra) {
  int is_ant = 0, kind;
  if (strcmp(task, "HalfCheetah") == 0) {
    kind = TASK_CHEETAH;
  } else if (strcmp(task, "Ant") == 0) {
    kind = TASK_ANT;
    is_ant = 1;
  } else if (strcmp(task, "Walker2d") == 0 || strcmp(task, "Walker2dV5") == 0) {
    kind = TASK_WALKER;
  } else if (strcmp(task, "InvertedPendulum") == 0) {
    kind = TASK_IPEND;
  } else if (strcmp(task, "InvertedDoublePendulum") == 0) {
    kind = TASK_IDPEND;
  } else if (strcmp(task, "Reacher") == 0) {
    kind = TASK_REACHER;
  } else if (strcmp(task, "Swimmer") == 0) {
    kind = TASK_SWIMMER;
  } else if (strcmp(task, "Hopper") == 0) {
    kind = TASK_HOPPER;
  } else if (strcmp(task, "Humanoid") == 0) {
    kind = TASK_HUMANOID;
  } else if (strcmp(task, "HumanoidStandup") == 0) {
    kind = TASK_STANDUP;
  } else if (strcmp(task, "Pusher") == 0 || strcmp(task, "PusherV5") == 0) {
    kind = TASK_PUSHER;
  } else {
    return NULL;
  }
  mj_pool* p = (mj_pool*)calloc(1, sizeof(mj_pool));
  p->is_ant = is_ant;
  p->task = kind;
  const int walker = kind == TASK_WALKER;
  const int v5 = strcmp(task, "Walker2dV5") == 0;
  if (is_ant) {
    mjc_build_ant(&p->m);
  } else if (walker) {
    mjc_build_walker2d(&p->m, v5);
  } else if (kind == TASK_IPEND) {
    mjc_build_inverted_pendulum(&p->m);
  } else if (kind == TASK_IDPEND) {
    mjc_build_inverted_double_pendulum(&p->m);
  } else if (kind == TASK_REACHER) {
    mjc_build_reacher(&p->m);
  } else if (kind == TASK_SWIMMER) {
    mjc_build_swimmer(&p->m);
  } else if (kind == TASK_HOPPER) {
    mjc_build_hopper(&p->m);
  } else if (kind == TASK_HUMANOID || kind == TASK_STANDUP) {
    mjc_build_humanoid(&p->m, kind == TASK_STANDUP);
  } else if (kind == TASK_PUSHER) {
    mjc_build_pusher(&p->m, strcmp(task, "PusherV5") == 0);
  } else {
    mjc_build_half_cheetah(&p->m);
  }
  p->num_envs = num_envs;
  p->max_episode_steps = max_episode_steps > 0 ? max_episode_steps : INT_MAX;
  const int pend = kind == TASK_IPEND || kind == TASK_IDPEND;
  const int reacher = kind == TASK_REACHER, swimmer = kind == TASK_SWIMMER;
  const int hopper = kind == TASK_HOPPER;
  const int humanoid = kind == TASK_HUMANOID || kind == TASK_STANDUP;
  p->frame_skip = (int)extra_or(
      extra, n_extra, 0,
      (walker || swimmer || hopper) ? 4 : ((kind == TASK_IPEND || reacher) ? 2 : 5));
  /* half_cheetah.h:33-43 / ant.h:33-50 / walker2d.h:32-47 defaults */
  p->ctrl_cost_weight =
      extra_or(extra, n_extra, 1,
               is_ant ? 0.5
               : (walker || hopper) ? 0.001
                                    : (reacher ? 1.0 : (swimmer ? 1e-4 : 0.1)));
  p->reward_after_step = extra_or(extra, n_extra, 16, 0) != 0; /* Reacher-v5 */
  p->obs_include_z = extra_or(extra, n_extra, 17, 1) != 0;
  p->dist_cost_weight = 1.0;
  p->reset_qpos_scale = 0.1;
  p->reset_qvel_scale = 0.005;
  p->reset_goal_scale = 0.2;
  /* humanoid.h:39 (1.25), humanoid_standup.h:35 (1.0) */
  p->forward_reward_weight = extra_or(extra, n_extra, 2, kind == TASK_HUMANOID ? 1.25 : 1.0);
  /* inverted_pendulum.h:32-41 (noise 0.01), inverted_double_pendulum.h:32-44 (0.1) */
  p->reset_noise_scale =
      extra_or(extra, n_extra, 3, (walker || hopper) ? 0.005 : ((kind == TASK_IPEND || humanoid) ? 0.01 : 0.1));
  p->reward_if_not_terminated = extra_or(extra, n_extra, 10, 0) != 0;
  p->constraint_obs_dim = (int)extra_or(extra, n_extra, 11, 3);
  p->observation_min = -10.0;
  p->observation_max = 10.0;
  p->use_contact_force = extra_or(extra, n_extra, 12, 0) != 0;
  p->post_constraint = extra_or(extra, n_extra, 13, 0) != 0;
  p->exclude_worldbody = extra_or(extra, n_extra, 14, 0) != 0;
  p->exclude_root_actuator = extra_or(extra, n_extra, 19, 0) != 0;
  p->contact_cost_weight = humanoid ? 5e-7 : 5e-4; /* ant.h:44-47, humanoid.h:45-46 */
  p->contact_cost_max = 10.0;
  p->contact_force_min = -1.0;
  p->contact_force_max = 1.0;
  p->m.disable_contact = extra_or(extra, n_extra, 4, 0) != 0;
  p->m.disable_limit = extra_or(extra, n_extra, 5, 0) != 0;
  p->m.disable_actuation = extra_or(extra, n_extra, 6, 0) != 0;
  p->m.disable_selfcollide = extra_or(extra, n_extra, 18, 0) != 0;
  p->m.warmstart_rule = (int)extra_or(extra, n_extra, 24, 0); /* mjcpu.h: 0 per forward, 1 per step */
  if (extra_or(extra, n_extra, 7, 0) != 0) { /* invariant tests: no passive */
    for (int i = 0; i < p->m.nv; ++i) p->m.dof_damping[i] = 0;
    for (int j = 0; j < p->m.njnt; ++j) p->m.jnt_stiffness[j] = 0;
  }
  if (extra_or(extra, n_extra, 8, -1) >= 0) p->m.integrator = (int)extra[8];
  if (extra_or(extra, n_extra, 9, 0) > 0) p->m.timestep = extra[9];
  p->healthy_reward = kind == TASK_IDPEND ? 10.0 : (kind == TASK_HUMANOID ? 5.0 : 1.0);
  p->healthy_z_min = walker ? 0.8 : (hopper ? 0.7 : (kind == TASK_IPEND ? -0.2 : (humanoid ? 1.0 : 0.2)));
  p->healthy_z_max = (walker || humanoid) ? 2.0 : (kind == TASK_IPEND ? 0.2 : 1.0);
  p->healthy_angle_min = hopper ? -0.2 : -1.0; /* hopper.h:44-46 */
  p->healthy_angle_max = hopper ? 0.2 : 1.0;
  p->velocity_min = -10.0;
  p->velocity_max = 10.0;
  p->terminate_when_unhealthy = 1;
  p->legacy_healthy_reward = v5 ? 0 : 1; /* gym/registration.py:79-83 */
  if (extra_or(extra, n_extra, 15, -1) >= 0) p->legacy_healthy_reward = extra[15] != 0;
  p->torso = 1; /* mj_name2id(model, mjOBJ_XBODY, "torso"), ant.h:119 */
  if (reacher) { /* the lagged body of this task is the fingertip (reacher.h:175-181) */
    p->torso = 3;
    p->target = 4;
  }
  const int pusher = kind == TASK_PUSHER;
  if (pusher) { /* pusher.h:33-46; extra: 0 frame_skip 1 ctrl_cost_weight 16 reward_after_step
                 * 20 dist_cost_weight 21 near_cost_weight 22 weighted_reward_info */
    p->dist_cost_weight = extra_or(extra, n_extra, 20, 1.0);
    p->near_cost_weight = extra_or(extra, n_extra, 21, 0.5);
    p->weighted_reward_info = extra_or(extra, n_extra, 22, 0) != 0;
    p->cyl_x_min = -0.3;
    p->cyl_x_max = 0.0;
    p->cyl_y_min = -0.2;
    p->cyl_y_max = 0.2;
    p->cyl_dist_min = 0.17;
    p->tips = 10;   /* mj_name2id(model, mjOBJ_XBODY, "tips_arm") */
    p->object = 11;
    p->goal = 12;
    p->torso = p->tips;
  }
  for (int i = 0; i < 8; ++i) {
    p->key_names[i] = kCommonNames[i];
    p->key_dtype[i] = kCommonDtype[i];
    p->key_elems[i] = 1;
  }
  int k = 8;
  p->key_names[k] = "obs";
  p->key_dtype[k] = DT_F64;
  p->key_elems[k++] = is_ant ? 27 + (p->use_contact_force
                                         ? 6 * (p->m.nbody - (p->exclude_worldbody ? 1 : 0))
                                         : 0)
                      : kind == TASK_IPEND ? 4
                      : kind == TASK_IDPEND ? 8 + p->constraint_obs_dim
                      : reacher ? (p->obs_include_z ? 11 : 10)
                      : pusher ? 23
                      : swimmer ? 8
                      : hopper ? 11
                      : humanoid ? 376 - (p->exclude_worldbody ? 22 : 0) -
                                       (p->exclude_root_actuator ? 6 : 0)
                                            : 17;
  static const char* cheetah_info[4] = {"info:reward_run", "info:reward_ctrl",
                                        "info:x_position", "info:x_velocity"};
  static const char* ant_info[9] = {
      "info:reward_forward", "info:reward_ctrl",  "info:reward_contact",
      "info:reward_survive", "info:x_position",   "info:y_position",
      "info:distance_from_origin", "info:x_velocity", "info:y_velocity"};
  /* walker2d.h:60-61: info:x_position, info:x_velocity */
  static const char* reacher_info[2] = {"info:reward_dist", "info:reward_ctrl"};
  /* swimmer.h:51-60 */
  static const char* swimmer_info[7] = {"info:reward_fwd", "info:reward_ctrl", "info:x_position",
                                        "info:y_position", "info:distance_from_origin",
                                        "info:x_velocity", "info:y_velocity"};
  /* humanoid.h:66-74, humanoid_standup.h:62-65 */
  static const char* humanoid_info[9] = {
      "info:reward_linvel", "info:reward_quadctrl", "info:reward_alive",
      "info:reward_impact", "info:x_position",      "info:y_position",
      "info:distance_from_origin", "info:x_velocity", "info:y_velocity"};
  static const char* standup_info[4] = {"info:reward_linup", "info:reward_quadctrl",
                                        "info:reward_alive", "info:reward_impact"};
  static const char* pusher_info[3] = {"info:reward_dist", "info:reward_ctrl", "info:reward_near"};
  int ninfo = is_ant ? 9 : (swimmer ? 7 : (walker || reacher || hopper ? 2 : (pend ? 0 : 4)));
  if (humanoid) ninfo = kind == TASK_HUMANOID ? 9 : 4;
  if (pusher) ninfo = 3;
  for (int i = 0; i < ninfo; ++i) {
    p->key_names[k] = kind == TASK_HUMANOID ? humanoid_info[i]
                      : kind == TASK_STANDUP ? standup_info[i]
                      : is_ant ? ant_info[i]
                      : swimmer ? swimmer_info[i]
                      : reacher ? reacher_info[i]
                      : pusher ? pusher_info[i]
                                : cheetah_info[((walker || hopper) ? 2 : 0) + i];
    p->key_dtype[k] = DT_F64;
    p->key_elems[k++] = 1;
  }
  p->nkeys = k;
  p->envs = (mj_env*)calloc((size_t)num_envs, sizeof(mj_env));
  for (int i = 0; i < num_envs; ++i) {
    mj_env* e = &p->envs[i];
    mjc_reset_data(&p->m, &e->d);
    orc_mt_seed(&e->gen, (uint32_t)(seed + i));
    e->current_step = -1;
    e->done = 1;
    e->elapsed_step = p->max_episode_steps + 1;
  }
  return p;
}

int mjcpu_num_state_keys(void* h) { return ((mj_pool*)h)->nkeys; }
int mjcpu_state_key(void* h, int i, char* name, int* dtype, int* elems) {
  mj_pool* p = (mj_pool*)h;
  if (i < 0 || i >= p->nkeys) return -1;
  strncpy(name, p->key_names[i], 63);
  name[63] = 0;
  *dtype = p->key_dtype[i];
  *elems = p->key_elems[i];
  return 0;
}
int mjcpu_action_info(void* h, int* dtype, int* elems) {
  mj_pool* p = (mj_pool*)h;
  *dtype = DT_F64;
  *elems = p->m.nu;
  return 0;
}

static void write_common(mj_pool* p, mj_env* e, int eid, void** out, int row,
                         float reward) {
  int done = e->done;
  ((int*)out[0])[row] = eid;
  ((int*)out[1])[row] = eid;
  ((int*)out[2])[row] = e->current_step;
  ((unsigned char*)out[3])[row] = (unsigned char)done;
  ((float*)out[4])[row] = reward;
  ((float*)out[5])[row] = (float)(!done);
  int st = 1;
  if (e->current_step == 0) {
    st = 0;
  } else if (done) {
    st = 2;
  }
  ((int*)out[6])[row] = st;
  ((unsigned char*)out[7])[row] =
      (unsigned char)(done && e->current_step >= p->max_episode_steps);
}

static double clip_obs(const mj_pool* p, double x) { /* std::min(max_, x) then std::max(min_, x) */
  x = p->observation_max < x ? p->observation_max : x;
  x = p->observation_min > x ? p->observation_min : x;
  return x;
}

/* world position of the "tip" site (inverted_double_pendulum_envpool.xml:55):
 * pos="0 0 .6" in the frame of the last pole */
static void idp_tip(const mj_pool* p, const mj_env* e, double* x, double* z) {
  int b = p->m.nbody - 1;
  const double* R = e->d.xmat[b];
  *x = e->d.xpos[b][0] + R[2] * 0.6;
  *z = e->d.xpos[b][2] + R[8] * 0.6;
}

static void reacher_dist(const mj_pool* p, const mj_env* e, double* dist) {
  for (int k = 0; k < 3; ++k) dist[k] = e->d.xpos[p->torso][k] - e->d.xpos[p->target][k];
}

static double body_dist(const mj_env* e, int b0, int b1) { /* PusherEnvBase::GetDist, pusher.h:190-195 */
  double x = e->d.xpos[b0][0] - e->d.xpos[b1][0];
  double y = e->d.xpos[b0][1] - e->d.xpos[b1][1];
  double z = e->d.xpos[b0][2] - e->d.xpos[b1][2];
  return sqrt(x * x + y * y + z * z);
}

static void write_obs(mj_pool* p, mj_env* e, void** out, int row) {
  if (p->task == TASK_PUSHER) { /* pusher.h:207-224: xpos of the LAST forward evaluation */
    double* obs = (double*)out[8] + (size_t)row * 23;
    for (int i = 0; i < 7; ++i) *(obs++) = e->d.qpos[i];
    for (int i = 0; i < 7; ++i) *(obs++) = e->d.qvel[i];
    for (int i = 0; i < 3; ++i) *(obs++) = e->d.xpos[p->tips][i];
    for (int i = 0; i < 3; ++i) *(obs++) = e->d.xpos[p->object][i];
    for (int i = 0; i < 3; ++i) *(obs++) = e->d.xpos[p->goal][i];
    return;
  }
  if (p->task == TASK_REACHER) { /* reacher.h:196-216 */
    int n = p->obs_include_z ? 11 : 10;
    double* obs = (double*)out[8] + (size_t)row * n, dist[3];
    *(obs++) = cos(e->d.qpos[0]);
    *(obs++) = cos(e->d.qpos[1]);
    *(obs++) = sin(e->d.qpos[0]);
    *(obs++) = sin(e->d.qpos[1]);
    for (int i = 2; i < 4; ++i) *(obs++) = e->d.qpos[i];
    for (int i = 0; i < 2; ++i) *(obs++) = e->d.qvel[i];
    reacher_dist(p, e, dist);
    *(obs++) = dist[0];
    *(obs++) = dist[1];
    if (p->obs_include_z) *(obs++) = dist[2];
    return;
  }
  if (p->task == TASK_HUMANOID || p->task == TASK_STANDUP) { /* humanoid.h:229-257 */
    double* obs = (double*)out[8] + (size_t)row * p->key_elems[8];
    int b0 = p->exclude_worldbody ? 1 : 0;
    for (int i = 2; i < p->m.nq; ++i) *(obs++) = e->d.qpos[i];
    for (int i = 0; i < p->m.nv; ++i) *(obs++) = e->d.qvel[i];
    for (int b = b0; b < p->m.nbody; ++b) {
      for (int j = 0; j < 10; ++j) *(obs++) = e->d.cinert[b][j];
    }
    for (int b = b0; b < p->m.nbody; ++b) {
      for (int j = 0; j < 6; ++j) *(obs++) = e->d.cvel[b][j];
    }
    for (int i = p->exclude_root_actuator ? 6 : 0; i < p->m.nv; ++i) *(obs++) = e->d.qfrc_actuator[i];
    for (int b = b0; b < p->m.nbody; ++b) {
      for (int j = 0; j < 6; ++j) *(obs++) = e->d.cfrc_ext[b][j];
    }
    return;
  }
  if (p->task == TASK_IPEND) { /* inverted_pendulum.h:172-178 */
    double* obs = (double*)out[8] + (size_t)row * 4;
    for (int i = 0; i < 2; ++i) obs[i] = e->d.qpos[i];
    for (int i = 0; i < 2; ++i) obs[2 + i] = e->d.qvel[i];
    return;
  }
  if (p->task == TASK_IDPEND) { /* inverted_double_pendulum.h:160-181 */
    double* obs = (double*)out[8] + (size_t)row * (8 + p->constraint_obs_dim);
    *(obs++) = e->d.qpos[0];
    *(obs++) = sin(e->d.qpos[1]);
    *(obs++) = sin(e->d.qpos[2]);
    *(obs++) = cos(e->d.qpos[1]);
    *(obs++) = cos(e->d.qpos[2]);
    for (int i = 0; i < 3; ++i) *(obs++) = clip_obs(p, e->d.qvel[i]);
    for (int i = 0; i < p->constraint_obs_dim; ++i) *(obs++) = clip_obs(p, e->d.qfrc_constraint[i]);
    return;
  }
  int skip = (p->is_ant || p->task == TASK_SWIMMER) ? 2 : 1; /* exclude_current_positions... */
  int n = (p->is_ant || p->task == TASK_SWIMMER || p->task == TASK_HOPPER) ? p->key_elems[8] : 17;
  double* obs = (double*)out[8] + (size_t)row * n;
  for (int i = skip; i < p->m.nq; ++i) *(obs++) = e->d.qpos[i];
  for (int i = 0; i < p->m.nv; ++i) {
    double x = e->d.qvel[i];
    if (p->task == TASK_WALKER || p->task == TASK_HOPPER) { /* walker2d.h:196-201, hopper.h */
      x = x < p->velocity_min ? p->velocity_min : x;
      x = x > p->velocity_max ? p->velocity_max : x;
    }
    *(obs++) = x;
  }
  if (p->is_ant && p->use_contact_force) { /* ant.h:248-258 */
    for (int b = p->exclude_worldbody ? 1 : 0; b < p->m.nbody; ++b) {
      for (int j = 0; j < 6; ++j) {
        double x = e->d.cfrc_ext[b][j];
        x = fmin(fmax(x, p->contact_force_min), p->contact_force_max);
        *(obs++) = x;
      }
    }
  }
}

/* MujocoReset + MujocoResetModel: mujoco_env.h:126-131, half_cheetah.h:105-117 */
static void mujoco_reset(mj_pool* p, mj_env* e) {
  double warm[MJC_MAXV];
  (void)warm;
  mjc_reset_data(&p->m, &e->d); /* mj_resetData */
  if (p->task == TASK_PUSHER) { /* pusher.h:115-136 */
    int nq = p->m.nq, nv = p->m.nv;
    for (int i = 0; i < nq - 4; ++i) e->d.qpos[i] = p->m.qpos0[i];
    for (;;) {
      double x = orc_uniform_real(&e->gen, p->cyl_x_min, p->cyl_x_max);
      double y = orc_uniform_real(&e->gen, p->cyl_y_min, p->cyl_y_max);
      if (sqrt(x * x + y * y) > p->cyl_dist_min) {
        e->d.qpos[nq - 4] = x;
        e->d.qpos[nq - 3] = y;
        e->d.qpos[nq - 2] = 0.0;
        e->d.qpos[nq - 1] = 0.0;
        break;
      }
    }
    for (int i = 0; i < nv; ++i) {
      e->d.qvel[i] = i < nv - 4 ? 0.0 + orc_uniform_real(&e->gen, -p->reset_qvel_scale,
                                                         p->reset_qvel_scale)
                                : 0.0;
    }
    mjc_forward(&p->m, &e->d);
    return;
  }
  if (p->task == TASK_REACHER) { /* reacher.h:112-132 */
    int nq = p->m.nq, nv = p->m.nv;
    for (int i = 0; i < nq - 2; ++i) {
      e->d.qpos[i] = p->m.qpos0[i] +
                     orc_uniform_real(&e->gen, -p->reset_qpos_scale, p->reset_qpos_scale);
    }
    for (;;) {
      double x = orc_uniform_real(&e->gen, -p->reset_goal_scale, p->reset_goal_scale);
      double y = orc_uniform_real(&e->gen, -p->reset_goal_scale, p->reset_goal_scale);
      if (sqrt(x * x + y * y) < p->reset_goal_scale) {
        e->d.qpos[nq - 2] = x;
        e->d.qpos[nq - 1] = y;
        break;
      }
    }
    for (int i = 0; i < nv; ++i) {
      e->d.qvel[i] = i < nv - 2 ? 0.0 + orc_uniform_real(&e->gen, -p->reset_qvel_scale,
                                                         p->reset_qvel_scale)
                                : 0.0;
    }
    mjc_forward(&p->m, &e->d);
    return;
  }
  for (int i = 0; i < p->m.nq; ++i) {
    e->d.qpos[i] = p->m.qpos0[i] +
                   orc_uniform_real(&e->gen, -p->reset_noise_scale,
                                    p->reset_noise_scale);
  }
  for (int i = 0; i < p->m.nv; ++i) {
    if (p->task == TASK_WALKER || p->task == TASK_IPEND || p->task == TASK_SWIMMER ||
        p->task == TASK_HOPPER || p->task == TASK_HUMANOID || p->task == TASK_STANDUP) {
      /* walker2d.h:119-126, inverted_pendulum.h:100-107: uniform for qvel too */
      e->d.qvel[i] = 0.0 + orc_uniform_real(&e->gen, -p->reset_noise_scale,
                                            p->reset_noise_scale);
    } else {
      e->d.qvel[i] = 0.0 + orc_normal(&e->gen, &e->nstate, 0, p->reset_noise_scale);
    }
  }
  mjc_forward(&p->m, &e->d); /* mj_forward */
}

/* HumanoidEnvBase::GetMassCenter, humanoid.h:212-223: from the xipos of the LAST forward
 * evaluation */
static void mass_center(const mj_pool* p, const mj_env* e, double* mc) {
  double sum = 0, x = 0, y = 0;
  for (int b = 0; b < p->m.nbody; ++b) {
    double mass = p->m.body_mass[b];
    sum += mass;
    x += mass * e->d.xipos[b][0];
    y += mass * e->d.xipos[b][1];
  }
  mc[0] = x / sum;
  mc[1] = y / sum;
}

static int ant_is_healthy(mj_pool* p, mj_env* e) { /* ant.h:214-229 */
  if (e->d.qpos[2] < p->healthy_z_min || e->d.qpos[2] > p->healthy_z_max) return 0;
  for (int i = 0; i < p->m.nq; ++i) {
    if (!isfinite(e->d.qpos[i])) return 0;
  }
  for (int i = 0; i < p->m.nv; ++i) {
    if (!isfinite(e->d.qvel[i])) return 0;
  }
  return 1;
}

static void env_step(mj_pool* p, int eid, int force_reset, const double* act,
                     void** out, int row) {
  mj_env* e = &p->envs[eid];
  int reset = force_reset || e->done; /* async_envpool.h:127 */
  float reward = 0.0f;
  int ninfo = (p->is_ant || p->task == TASK_HUMANOID) ? 9
              : p->task == TASK_STANDUP ? 4
              : p->task == TASK_SWIMMER ? 7
              : p->task == TASK_PUSHER ? 3
              : (p->task == TASK_WALKER || p->task == TASK_REACHER || p->task == TASK_HOPPER) ? 2
              : (p->task >= TASK_IPEND ? 0 : 4);
  double info[9] = {0};
  if (reset) {
    e->current_step = 0;
    e->done = 0;
    e->elapsed_step = 0;
    mujoco_reset(p, e);
    /* WriteState(0.0, 0, ...) on reset: ant.h:160-164 passes zeros */
    if (p->is_ant) info[6] = sqrt(0.0);
    /* The reset WriteState receives the costs as +0.0 and stores them NEGATED: the info key
     * holds -0.0 (found by running the reference's own wrappers, oracle/_ref/libref_mujoco.so).
     * half_cheetah.h:178, ant.h:262-263, swimmer.h:172, reacher.h:219-220, pusher.h:225-232,
     * humanoid.h:272-274, humanoid_standup.h:227-228 */
    switch (p->task) {
      case TASK_CHEETAH: info[1] = -0.0; break;
      case TASK_ANT: info[1] = info[2] = -0.0; break;
      case TASK_SWIMMER: info[1] = -0.0; break;
      case TASK_REACHER: info[0] = info[1] = -0.0; break;
      case TASK_PUSHER: info[0] = info[1] = info[2] = -0.0; break;
      case TASK_HUMANOID: info[1] = info[3] = -0.0; break;
      case TASK_STANDUP: info[1] = info[3] = -0.0; break;
      default: break;
    }
    /* humanoid_standup.h:226: WriteState stores the member healthy_reward_ on resets too */
    if (p->task == TASK_STANDUP) info[2] = p->healthy_reward;
    e->lag_set = 0;
  } else {
    ++e->current_step;
    double dt = p->frame_skip * p->m.timestep;
    double ctrl_cost = 0;
    for (int i = 0; i < p->m.nu; ++i) ctrl_cost += p->ctrl_cost_weight * act[i] * act[i];
    if (p->task == TASK_HUMANOID || p->task == TASK_STANDUP) {
      /* humanoid.h:164-205, humanoid_standup.h:160-185 */
      double before[2], after[2];
      mass_center(p, e, before);
      if (e->lag_set) {
        before[0] = e->lag_mc[0];
        before[1] = e->lag_mc[1];
        e->lag_set = 0;
      }
      for (int i = 0; i < p->m.nu; ++i) e->d.ctrl[i] = act[i];
      for (int i = 0; i < p->frame_skip; ++i) mjc_step(&p->m, &e->d);
      if (p->post_constraint) mjc_rne_post_constraint(&p->m, &e->d);
      mass_center(p, e, after);
      double contact_cost = 0.0;
      if (p->use_contact_force || p->task == TASK_STANDUP) {
        for (int b = 0; b < p->m.nbody; ++b) {
          for (int j = 0; j < 6; ++j) {
            double x = e->d.cfrc_ext[b][j];
            contact_cost += p->contact_cost_weight * x * x;
          }
        }
        contact_cost = fmin(contact_cost, p->contact_cost_max);
      }
      if (p->task == TASK_STANDUP) {
        double xv = e->d.qpos[2] / p->m.timestep;
        reward = (float)(xv * p->forward_reward_weight + p->healthy_reward - ctrl_cost - contact_cost);
        e->done = (++e->elapsed_step >= p->max_episode_steps);
        info[0] = xv * p->forward_reward_weight;
        info[1] = -ctrl_cost;
        info[2] = p->healthy_reward;
        info[3] = -contact_cost;
      } else {
        double xv = (after[0] - before[0]) / dt, yv = (after[1] - before[1]) / dt;
        int healthy = p->healthy_z_min < e->d.qpos[2] && e->d.qpos[2] < p->healthy_z_max;
        int give = healthy;
        if (p->legacy_healthy_reward) give = p->terminate_when_unhealthy || healthy;
        double healthy_reward = give ? p->healthy_reward : 0.0;
        reward = (float)(xv * p->forward_reward_weight + healthy_reward - ctrl_cost - contact_cost);
        ++e->elapsed_step;
        e->done = (p->terminate_when_unhealthy ? !healthy : 0) ||
                  (e->elapsed_step >= p->max_episode_steps);
        info[0] = xv * p->forward_reward_weight;
        info[1] = -ctrl_cost;
        info[2] = healthy_reward;
        info[3] = -contact_cost;
        info[4] = after[0];
        info[5] = after[1];
        info[6] = sqrt(after[0] * after[0] + after[1] * after[1]);
        info[7] = xv;
        info[8] = yv;
      }
    } else if (p->task == TASK_HOPPER) { /* hopper.h:158-203 */
      double x_before = e->d.qpos[0];
      for (int i = 0; i < p->m.nu; ++i) e->d.ctrl[i] = act[i];
      for (int i = 0; i < p->frame_skip; ++i) mjc_step(&p->m, &e->d);
      double x_after = e->d.qpos[0];
      double xv = (x_after - x_before) / dt;
      int healthy = !(e->d.qpos[2] <= p->healthy_angle_min || e->d.qpos[2] >= p->healthy_angle_max ||
                      e->d.qpos[1] <= p->healthy_z_min);
      for (int i = 2; i < p->m.nq; ++i) {
        if (e->d.qpos[i] <= -100.0 || e->d.qpos[i] >= 100.0) healthy = 0;
      }
      for (int i = 0; i < p->m.nv; ++i) {
        if (e->d.qvel[i] <= -100.0 || e->d.qvel[i] >= 100.0) healthy = 0;
      }
      int give = healthy;
      if (p->legacy_healthy_reward) give = p->terminate_when_unhealthy || healthy;
      double healthy_reward = give ? p->healthy_reward : 0.0;
      reward = (float)(xv * p->forward_reward_weight + healthy_reward - ctrl_cost);
      ++e->elapsed_step;
      e->done = (p->terminate_when_unhealthy ? !healthy : 0) ||
                (e->elapsed_step >= p->max_episode_steps);
      info[0] = x_after;
      info[1] = xv;
    } else if (p->task == TASK_SWIMMER) { /* swimmer.h:131-152 */
      double x_before = e->d.qpos[0], y_before = e->d.qpos[1];
      for (int i = 0; i < p->m.nu; ++i) e->d.ctrl[i] = act[i];
      for (int i = 0; i < p->frame_skip; ++i) mjc_step(&p->m, &e->d);
      double x_after = e->d.qpos[0], y_after = e->d.qpos[1];
      double xv = (x_after - x_before) / dt, yv = (y_after - y_before) / dt;
      reward = (float)(xv * p->forward_reward_weight - ctrl_cost);
      e->done = (++e->elapsed_step >= p->max_episode_steps);
      info[0] = xv * p->forward_reward_weight;
      info[1] = -ctrl_cost;
      info[2] = x_after;
      info[3] = y_after;
      info[4] = sqrt(x_after * x_after + y_after * y_after);
      info[5] = xv;
      info[6] = yv;
    } else if (p->task == TASK_REACHER) { /* reacher.h:152-177 */
      double dist[3] = {0, 0, 0};
      if (!p->reward_after_step) reacher_dist(p, e, dist);
      for (int i = 0; i < p->m.nu; ++i) e->d.ctrl[i] = act[i];
      for (int i = 0; i < p->frame_skip; ++i) mjc_step(&p->m, &e->d);
      if (p->reward_after_step) reacher_dist(p, e, dist);
      double dist_cost =
          p->dist_cost_weight * sqrt(dist[0] * dist[0] + dist[1] * dist[1] + dist[2] * dist[2]);
      reward = (float)(-dist_cost - ctrl_cost);
      e->done = (++e->elapsed_step >= p->max_episode_steps);
      info[0] = -dist_cost;
      info[1] = -ctrl_cost;
    } else if (p->task == TASK_PUSHER) { /* pusher.h:156-186 */
      double near_cost = 0, dist_cost = 0;
      if (!p->reward_after_step) {
        near_cost = body_dist(e, p->object, p->tips);
        dist_cost = body_dist(e, p->object, p->goal);
      }
      for (int i = 0; i < p->m.nu; ++i) e->d.ctrl[i] = act[i];
      for (int i = 0; i < p->frame_skip; ++i) mjc_step(&p->m, &e->d);
      if (p->reward_after_step) {
        near_cost = body_dist(e, p->object, p->tips);
        dist_cost = body_dist(e, p->object, p->goal);
      }
      double cc = 0;
      for (int i = 0; i < p->m.nu; ++i) cc += act[i] * act[i];
      reward = (float)(-cc * p->ctrl_cost_weight - dist_cost * p->dist_cost_weight -
                       near_cost * p->near_cost_weight);
      e->done = (++e->elapsed_step >= p->max_episode_steps);
      info[0] = -dist_cost * (p->weighted_reward_info ? p->dist_cost_weight : 1.0);
      info[1] = -cc * (p->weighted_reward_info ? p->ctrl_cost_weight : 1.0);
      info[2] = -near_cost * p->near_cost_weight;
    } else if (p->task == TASK_IPEND) { /* inverted_pendulum.h:137-148 */
      for (int i = 0; i < p->m.nu; ++i) e->d.ctrl[i] = act[i];
      for (int i = 0; i < p->frame_skip; ++i) mjc_step(&p->m, &e->d);
      int healthy = !(e->d.qpos[1] < p->healthy_z_min || e->d.qpos[1] > p->healthy_z_max);
      for (int i = 0; i < p->m.nq; ++i) healthy = healthy && isfinite(e->d.qpos[i]);
      for (int i = 0; i < p->m.nv; ++i) healthy = healthy && isfinite(e->d.qvel[i]);
      int terminated = !healthy;
      ++e->elapsed_step;
      e->done = terminated || (e->elapsed_step >= p->max_episode_steps);
      reward = p->reward_if_not_terminated ? (float)(!terminated) : 1.0f;
    } else if (p->task == TASK_IDPEND) { /* inverted_double_pendulum.h:126-148 */
      for (int i = 0; i < p->m.nu; ++i) e->d.ctrl[i] = act[i];
      for (int i = 0; i < p->frame_skip; ++i) mjc_step(&p->m, &e->d);
      double x, y; /* site_xpos of the last forward evaluation (RK4 stage 4) */
      idp_tip(p, e, &x, &y);
      double dist_penalty = 0.01 * x * x + (y - 2) * (y - 2);
      double v1 = e->d.qvel[1], v2 = e->d.qvel[2];
      double vel_penalty = 1e-3 * v1 * v1 + 5e-3 * v2 * v2;
      int terminated = !(y > p->healthy_z_max);
      double alive_bonus = p->reward_if_not_terminated ? p->healthy_reward * (int)(!terminated)
                                                       : p->healthy_reward;
      reward = (float)(alive_bonus - dist_penalty - vel_penalty);
      ++e->elapsed_step;
      e->done = terminated || (e->elapsed_step >= p->max_episode_steps);
    } else if (p->task == TASK_WALKER) { /* walker2d.h:150-178 */
      double x_before = e->d.qpos[0];
      for (int i = 0; i < p->m.nu; ++i) e->d.ctrl[i] = act[i];
      for (int i = 0; i < p->frame_skip; ++i) mjc_step(&p->m, &e->d);
      double x_after = e->d.qpos[0];
      double xv = (x_after - x_before) / dt;
      int healthy = !(e->d.qpos[1] < p->healthy_z_min || e->d.qpos[1] > p->healthy_z_max ||
                      e->d.qpos[2] < p->healthy_angle_min ||
                      e->d.qpos[2] > p->healthy_angle_max); /* :181-190 */
      int give = healthy;
      if (p->legacy_healthy_reward) give = p->terminate_when_unhealthy || healthy;
      double healthy_reward = give ? p->healthy_reward : 0.0;
      reward = (float)(xv * p->forward_reward_weight + healthy_reward - ctrl_cost);
      ++e->elapsed_step;
      e->done = (p->terminate_when_unhealthy ? !healthy : 0) ||
                (e->elapsed_step >= p->max_episode_steps);
      info[0] = x_after;
      info[1] = xv;
    } else if (!p->is_ant) { /* half_cheetah.h:136-155 */
      double x_before = e->d.qpos[0];
      for (int i = 0; i < p->m.nu; ++i) e->d.ctrl[i] = act[i];
      for (int i = 0; i < p->frame_skip; ++i) mjc_step(&p->m, &e->d);
      double x_after = e->d.qpos[0];
      double xv = (x_after - x_before) / dt;
      reward = (float)(xv * p->forward_reward_weight - ctrl_cost);
      e->done = (++e->elapsed_step >= p->max_episode_steps);
      info[0] = xv * p->forward_reward_weight;
      info[1] = -ctrl_cost;
      info[2] = x_after;
      info[3] = xv;
    } else { /* ant.h:166-212 */
      double x_before = e->d.xpos[p->torso][0], y_before = e->d.xpos[p->torso][1];
      for (int i = 0; i < p->m.nu; ++i) e->d.ctrl[i] = act[i];
      for (int i = 0; i < p->frame_skip; ++i) mjc_step(&p->m, &e->d);
      /* mujoco_env.h:145-147; without it cfrc_ext keeps the zeros of mj_resetData
       * (MuJoCo 3 only fills it in mj_rnePostConstraint) */
      if (p->post_constraint) mjc_rne_post_constraint(&p->m, &e->d);
      double x_after = e->d.xpos[p->torso][0], y_after = e->d.xpos[p->torso][1];
      double xv = (x_after - x_before) / dt, yv = (y_after - y_before) / dt;
      double contact_cost = 0.0;
      if (p->use_contact_force) { /* ant.h:183-194 */
        for (int b = p->exclude_worldbody ? 1 : 0; b < p->m.nbody; ++b) {
          for (int j = 0; j < 6; ++j) {
            double x = e->d.cfrc_ext[b][j];
            x = fmin(p->contact_force_max, x);
            x = fmax(p->contact_force_min, x);
            contact_cost += p->contact_cost_weight * x * x;
          }
        }
      }
      int healthy = ant_is_healthy(p, e);
      int give = healthy;
      if (p->legacy_healthy_reward) give = p->terminate_when_unhealthy || healthy;
      double healthy_reward = give ? p->healthy_reward : 0.0;
      reward = (float)(xv * p->forward_reward_weight + healthy_reward -
                       ctrl_cost - contact_cost);
      ++e->elapsed_step;
      e->done = (p->terminate_when_unhealthy ? !healthy : 0) ||
                (e->elapsed_step >= p->max_episode_steps);
      info[0] = xv * p->forward_reward_weight;
      info[1] = -ctrl_cost;
      info[2] = -contact_cost;
      info[3] = healthy_reward;
      info[4] = x_after;
      info[5] = y_after;
      info[6] = sqrt(x_after * x_after + y_after * y_after);
      info[7] = xv;
      info[8] = yv;
    }
  }
  write_obs(p, e, out, row);
  for (int i = 0; i < ninfo; ++i) ((double*)out[9 + i])[row] = info[i];
  write_common(p, e, eid, out, row, reward);
}

void mjcpu_reset(void* h, const int* ids, int k, void** out) {
  mj_pool* p = (mj_pool*)h;
  for (int i = 0; i < k; ++i) env_step(p, ids[i], 1, NULL, out, i);
}

void mjcpu_step(void* h, const int* ids, int k, const void* action, void** out) {
  mj_pool* p = (mj_pool*)h;
  const double* a = (const double*)action;
  /* envs are independent (one mjData each): spread them over the host cores, like
   * the reference's worker threads (async_envpool.h:118-132).  Row i is written
   * by exactly one thread, so the result does not depend on the thread count. */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < k; ++i) env_step(p, ids[i], 0, a + (size_t)i * p->m.nu, out, i);
}
int mjcpu_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void mjcpu_destroy(void* h) {
  mj_pool* p = (mj_pool*)h;
  free(p->envs);
  free(p);
}

/* flat state: qpos[nq] qvel[nv] qacc_warmstart[nv] time xlag ylag done
 * cur_step normal_saved normal_avail */
int mjcpu_state_dim(void* h) {
  mj_pool* p = (mj_pool*)h;
  /* Pusher: + xpos of tips_arm (3) and of the object (x, y) of the last forward evaluation */
  return p->m.nq + 2 * p->m.nv + 7 + (p->task == TASK_PUSHER ? 5 : 0);
}
void mjcpu_get_state(void* h, const int* ids, int k, double* out) {
  mj_pool* p = (mj_pool*)h;
  int dim = mjcpu_state_dim(h), nq = p->m.nq, nv = p->m.nv;
  for (int i = 0; i < k; ++i) {
    mj_env* e = &p->envs[ids[i]];
    double* o = out + (size_t)i * dim;
    memcpy(o, e->d.qpos, sizeof(double) * nq);
    memcpy(o + nq, e->d.qvel, sizeof(double) * nv);
    memcpy(o + nq + nv, e->d.qacc_warmstart, sizeof(double) * nv);
    double* t = o + nq + 2 * nv;
    t[0] = e->d.time;
    t[1] = e->d.xpos[p->torso][0];
    t[2] = e->d.xpos[p->torso][1];
    if (p->task == TASK_HUMANOID || p->task == TASK_STANDUP) mass_center(p, e, t + 1);
    t[3] = e->done;
    t[4] = e->current_step;
    t[5] = e->nstate.saved;
    t[6] = e->nstate.saved_available;
    if (p->task == TASK_PUSHER) {
      for (int c = 0; c < 3; ++c) t[7 + c] = e->d.xpos[p->tips][c];
      t[10] = e->d.xpos[p->object][0];
      t[11] = e->d.xpos[p->object][1];
    }
  }
}
void mjcpu_set_state(void* h, const int* ids, int k, const double* in) {
  mj_pool* p = (mj_pool*)h;
  int dim = mjcpu_state_dim(h), nq = p->m.nq, nv = p->m.nv;
  for (int i = 0; i < k; ++i) {
    mj_env* e = &p->envs[ids[i]];
    const double* o = in + (size_t)i * dim;
    memcpy(e->d.qpos, o, sizeof(double) * nq);
    memcpy(e->d.qvel, o + nq, sizeof(double) * nv);
    memcpy(e->d.qacc_warmstart, o + nq + nv, sizeof(double) * nv);
    const double* t = o + nq + 2 * nv;
    e->d.time = t[0];
    e->d.xpos[p->torso][0] = t[1];
    e->d.xpos[p->torso][1] = t[2];
    e->lag_set = 1;
    e->lag_mc[0] = t[1];
    e->lag_mc[1] = t[2];
    e->done = t[3] != 0;
    e->current_step = (int)t[4];
    e->elapsed_step = e->current_step;
    e->nstate.saved = t[5];
    e->nstate.saved_available = t[6] != 0;
    if (p->task == TASK_PUSHER) { /* the lagged positions the next step's costs / obs read */
      for (int c = 0; c < 3; ++c) e->d.xpos[p->tips][c] = t[7 + c];
      e->d.xpos[p->object][0] = t[10];
      e->d.xpos[p->object][1] = t[11];
      e->d.xpos[p->object][2] = p->m.body_pos[p->object][2];
      for (int c = 0; c < 3; ++c) {
        e->d.xpos[p->goal][c] = p->m.body_pos[p->goal][c];
      }
      e->d.xpos[p->goal][1] += e->d.qpos[p->m.nq - 2]; /* goal_slidey, goal_slidex */
      e->d.xpos[p->goal][0] += e->d.qpos[p->m.nq - 1];
    }
  }
}

/* debug / invariant hooks (tests only) */
const mjc_model* mjcpu_model(void* h) { return &((mj_pool*)h)->m; }
mjc_data* mjcpu_data(void* h, int env) { return &((mj_pool*)h)->envs[env].d; }
void mjcpu_model_scalars(void* h, double* out) {
  /* [nq nv nu nbody ngeom meaninertia total_mass] + body_mass[nbody]
   * + dof_invweight0[nv] + body_invweight0[nbody*2] */
  mj_pool* p = (mj_pool*)h;
  const mjc_model* m = &p->m;
  int k = 0;
  out[k++] = m->nq; out[k++] = m->nv; out[k++] = m->nu; out[k++] = m->nbody;
  out[k++] = m->ngeom; out[k++] = m->meaninertia;
  double tot = 0;
  for (int b = 0; b < m->nbody; ++b) tot += m->body_mass[b];
  out[k++] = tot;
  for (int b = 0; b < m->nbody; ++b) out[k++] = m->body_mass[b];
  for (int i = 0; i < m->nv; ++i) out[k++] = m->dof_invweight0[i];
  for (int b = 0; b < m->nbody; ++b) {
    out[k++] = m->body_invweight0[b][0];
    out[k++] = m->body_invweight0[b][1];
  }
}
/* raw physics access for invariant tests: set qpos/qvel/ctrl, run n mj_steps,
 * read back qpos, qvel, energies, contact/efc stats */
void mjcpu_raw_set(void* h, int env, const double* qpos, const double* qvel,
                   const double* ctrl) {
  mj_pool* p = (mj_pool*)h;
  mjc_data* d = &p->envs[env].d;
  mjc_reset_data(&p->m, d);
  memcpy(d->qpos, qpos, sizeof(double) * p->m.nq);
  memcpy(d->qvel, qvel, sizeof(double) * p->m.nv);
  memcpy(d->ctrl, ctrl, sizeof(double) * p->m.nu);
  mjc_forward(&p->m, d);
}
/* as mjcpu_raw_set, but with the warm start of a running simulation and WITHOUT a forward
 * pass: the state a golden vector of tools/pin_with_mujoco.py was recorded from (the PGS
 * result depends on qacc_warmstart) */
void mjcpu_raw_set_warm(void* h, int env, const double* qpos, const double* qvel,
                        const double* ctrl, const double* warm) {
  mj_pool* p = (mj_pool*)h;
  mjc_data* d = &p->envs[env].d;
  mjc_reset_data(&p->m, d);
  memcpy(d->qpos, qpos, sizeof(double) * p->m.nq);
  memcpy(d->qvel, qvel, sizeof(double) * p->m.nv);
  memcpy(d->ctrl, ctrl, sizeof(double) * p->m.nu);
  memcpy(d->qacc_warmstart, warm, sizeof(double) * p->m.nv);
}
/* fields of the LAST forward evaluation that the Humanoid tasks observe:
 * out = cinert[nbody*10] cvel[nbody*6] qfrc_actuator[nv] cfrc_ext[nbody*6] (after
 * mj_rnePostConstraint) */
void mjcpu_raw_observed(void* h, int env, double* out) {
  mj_pool* p = (mj_pool*)h;
  mjc_data* d = &p->envs[env].d;
  int k = 0;
  for (int b = 0; b < p->m.nbody; ++b) for (int j = 0; j < 10; ++j) out[k++] = d->cinert[b][j];
  for (int b = 0; b < p->m.nbody; ++b) for (int j = 0; j < 6; ++j) out[k++] = d->cvel[b][j];
  for (int i = 0; i < p->m.nv; ++i) out[k++] = d->qfrc_actuator[i];
  mjc_rne_post_constraint(&p->m, d);
  for (int b = 0; b < p->m.nbody; ++b) for (int j = 0; j < 6; ++j) out[k++] = d->cfrc_ext[b][j];
}
void mjcpu_raw_step(void* h, int env, int n) {
  mj_pool* p = (mj_pool*)h;
  for (int i = 0; i < n; ++i) mjc_step(&p->m, &p->envs[env].d);
}
void mjcpu_raw_get(void* h, int env, double* qpos, double* qvel, double* misc) {
  mj_pool* p = (mj_pool*)h;
  mjc_data* d = &p->envs[env].d;
  memcpy(qpos, d->qpos, sizeof(double) * p->m.nq);
  memcpy(qvel, d->qvel, sizeof(double) * p->m.nv);
  /* refresh position-dependent quantities for energy */
  mjc_data tmp = *d;
  mjc_forward(&p->m, &tmp);
  misc[0] = mjc_energy_kinetic(&p->m, &tmp);
  misc[1] = mjc_energy_potential(&p->m, &tmp);
  misc[2] = tmp.ncon;
  misc[3] = tmp.nefc;
  misc[4] = tmp.solver_iter;
  misc[5] = d->time;
  double fmin_ = 0, gn = 0;
  for (int r = 0; r < tmp.nefc; ++r) fmin_ = fmin(fmin_, tmp.efc_force[r]);
  /* optimality residual: M qacc - qfrc_smooth - qfrc_constraint */
  for (int i = 0; i < p->m.nv; ++i) {
    double s = -tmp.qfrc_smooth[i] - tmp.qfrc_constraint[i];
    for (int j = 0; j < p->m.nv; ++j) s += tmp.M[i][j] * tmp.qacc[j];
    gn += s * s;
  }
  misc[6] = fmin_;
  misc[7] = sqrt(gn);
  double asym = 0;
  for (int i = 0; i < p->m.nv; ++i) {
    for (int j = 0; j < p->m.nv; ++j) asym = fmax(asym, fabs(tmp.M[i][j] - tmp.M[j][i]));
  }
  misc[8] = asym;
  misc[9] = tmp.xpos[1][2];
  double fsum = 0;
  for (int r = 0; r < tmp.nefc; ++r) fsum += tmp.efc_force[r];
  misc[10] = fsum;
}

/* debug hook (tests): contacts of the last forward evaluation of env `env`:
 * out[10 * c + ...] = geom1 geom2 dist pos[3] normal[3] efc_address; returns ncon */
int mjcpu_raw_contacts(void* h, int env, double* out) {
  mj_pool* p = (mj_pool*)h;
  const mjc_data* d = &p->envs[env].d;
  for (int c = 0; c < d->ncon; ++c) {
    const mjc_contact* k = &d->contact[c];
    double* o = out + 10 * c;
    o[0] = k->geom1;
    o[1] = k->geom2;
    o[2] = k->dist;
    for (int i = 0; i < 3; ++i) o[3 + i] = k->pos[i];
    for (int i = 0; i < 3; ++i) o[6 + i] = k->frame[i];
    o[9] = k->efc_address;
  }
  return d->ncon;
}

/* Pinning hook (tests): one mj_forward at the state set by mjcpu_raw_set_warm, then the
 * named per-stage field of mjData, flattened row-major into `out` (capacity `cap` doubles).
 * Returns the number of doubles written, -1 for an unknown name.  The names are MuJoCo's
 * (mjData / mjContact members), so tools/pin_with_mujoco.py can dump the same fields from
 * real MuJoCo and a mismatch localises to one pipeline stage. */
void mjcpu_raw_forward(void* h, int env) {
  mj_pool* p = (mj_pool*)h;
  mjc_forward(&p->m, &p->envs[env].d);
}
int mjcpu_raw_stage(void* h, int env, const char* name, double* out, int cap) {
  mj_pool* p = (mj_pool*)h;
  const mjc_model* m = &p->m;
  const mjc_data* d = &p->envs[env].d;
  int nv = m->nv, nb = m->nbody, k = 0;
#define PUT(x) do { if (k < cap) out[k] = (x); ++k; } while (0)
  if (!strcmp(name, "qM")) { /* dense nv x nv (mj_fullM) */
    for (int i = 0; i < nv; ++i) for (int j = 0; j < nv; ++j) PUT(d->M[i][j]);
  } else if (!strcmp(name, "xpos")) {
    for (int b = 0; b < nb; ++b) for (int j = 0; j < 3; ++j) PUT(d->xpos[b][j]);
  } else if (!strcmp(name, "xquat")) {
    for (int b = 0; b < nb; ++b) for (int j = 0; j < 4; ++j) PUT(d->xquat[b][j]);
  } else if (!strcmp(name, "xipos")) {
    for (int b = 0; b < nb; ++b) for (int j = 0; j < 3; ++j) PUT(d->xipos[b][j]);
  } else if (!strcmp(name, "subtree_com")) {
    for (int b = 0; b < nb; ++b) for (int j = 0; j < 3; ++j) PUT(d->subtree_com[b][j]);
  } else if (!strcmp(name, "cinert")) {
    for (int b = 0; b < nb; ++b) for (int j = 0; j < 10; ++j) PUT(d->cinert[b][j]);
  } else if (!strcmp(name, "cdof")) {
    for (int i = 0; i < nv; ++i) for (int j = 0; j < 6; ++j) PUT(d->cdof[i][j]);
  } else if (!strcmp(name, "cvel")) {
    for (int b = 0; b < nb; ++b) for (int j = 0; j < 6; ++j) PUT(d->cvel[b][j]);
  } else if (!strcmp(name, "qfrc_passive")) {
    for (int i = 0; i < nv; ++i) PUT(d->qfrc_passive[i]);
  } else if (!strcmp(name, "qfrc_bias")) {
    for (int i = 0; i < nv; ++i) PUT(d->qfrc_bias[i]);
  } else if (!strcmp(name, "qfrc_actuator")) {
    for (int i = 0; i < nv; ++i) PUT(d->qfrc_actuator[i]);
  } else if (!strcmp(name, "qacc_smooth")) {
    for (int i = 0; i < nv; ++i) PUT(d->qacc_smooth[i]);
  } else if (!strcmp(name, "qacc")) {
    for (int i = 0; i < nv; ++i) PUT(d->qacc[i]);
  } else if (!strcmp(name, "qfrc_constraint")) {
    for (int i = 0; i < nv; ++i) PUT(d->qfrc_constraint[i]);
  } else if (!strcmp(name, "efc_J")) { /* dense nefc x nv */
    for (int r = 0; r < d->nefc; ++r) for (int j = 0; j < nv; ++j) PUT(d->efc_J[r][j]);
  } else if (!strcmp(name, "efc_pos")) {
    for (int r = 0; r < d->nefc; ++r) PUT(d->efc_pos[r]);
  } else if (!strcmp(name, "efc_margin")) {
    for (int r = 0; r < d->nefc; ++r) PUT(d->efc_margin[r]);
  } else if (!strcmp(name, "efc_vel")) {
    for (int r = 0; r < d->nefc; ++r) PUT(d->efc_vel[r]);
  } else if (!strcmp(name, "efc_aref")) {
    for (int r = 0; r < d->nefc; ++r) PUT(d->efc_aref[r]);
  } else if (!strcmp(name, "efc_R")) {
    for (int r = 0; r < d->nefc; ++r) PUT(d->efc_R[r]);
  } else if (!strcmp(name, "efc_D")) {
    for (int r = 0; r < d->nefc; ++r) PUT(d->efc_D[r]);
  } else if (!strcmp(name, "efc_diagApprox")) {
    for (int r = 0; r < d->nefc; ++r) PUT(d->efc_diagApprox[r]);
  } else if (!strcmp(name, "efc_KBIP")) { /* K, B, I per row (MuJoCo stores K B I P) */
    for (int r = 0; r < d->nefc; ++r) for (int j = 0; j < 3; ++j) PUT(d->efc_KBI[r][j]);
  } else if (!strcmp(name, "efc_force")) {
    for (int r = 0; r < d->nefc; ++r) PUT(d->efc_force[r]);
  } else if (!strcmp(name, "contact")) { /* per contact: geom1 geom2 dist includemargin pos[3] frame[9] dim efc_address */
    for (int c = 0; c < d->ncon; ++c) {
      const mjc_contact* q = &d->contact[c];
      PUT(q->geom1); PUT(q->geom2); PUT(q->dist); PUT(q->includemargin);
      for (int j = 0; j < 3; ++j) PUT(q->pos[j]);
      for (int j = 0; j < 9; ++j) PUT(q->frame[j]);
      PUT(q->dim); PUT(q->efc_address);
    }
  } else if (!strcmp(name, "counts")) { /* ncon nefc solver_niter */
    PUT(d->ncon); PUT(d->nefc); PUT(d->solver_iter);
  } else {
    return -1;
  }
#undef PUT
  return k;
}
