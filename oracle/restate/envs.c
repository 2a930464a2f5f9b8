/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded restatement of the reference's batched-step path
 * for classic_control and toy_text, behind the same `orc_*` C API as
 * oracle/_ref (oracle/ref_driver.cc) so tests can diff restatement, compiled
 * reference and the HIP engine with one harness.
 *
 * PINNED: tests/test_oracle_pinned.py checks this file bit-for-bit against
 * (a) oracle/_ref (the reference compiled in place) whenever it is present and
 * (b) the golden rollouts under tests/golden/ that were generated from
 * oracle/_ref by tests/golden/make_golden.py.
 *
 * Runtime semantics restated (per env, sync mode):
 *   envpool/core/async_envpool.h:118-132  reset = force_reset || IsDone()
 *   envpool/core/env.h:184-217            EnvStep / PreProcess (current_step_)
 *   envpool/core/env.h:224-256            Allocate(): done, discount, step_type,
 *                                         trunc, info:env_id, elapsed_step
 *   envpool/core/state_buffer.h:94-97     row = position in the send batch
 * Env bodies: see the comment above each function.
 * Compile with -ffp-contract=off.
 */
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rng.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

enum { DT_I32 = 0, DT_F32 = 1, DT_F64 = 2, DT_BOOL = 3 };

typedef struct {
  const char* name;
  int dtype;
  int elems;
} orc_key;

typedef struct {
  orc_mt19937 gen;
  int current_step; /* Env::current_step_, env.h:86 (starts at -1) */
  int done;         /* XxxEnv::done_{true} */
  int elapsed_step; /* XxxEnv::elapsed_step_ */
  double s[6];      /* continuous state */
  int i[12];        /* discrete state */
} orc_env;

struct orc_pool;
typedef void (*reset_fn)(struct orc_pool*, orc_env*, void** out, int row);
typedef void (*step_fn)(struct orc_pool*, orc_env*, const void* act, void** out,
                        int row);

typedef struct orc_pool {
  int num_envs;
  int max_episode_steps;
  int nkeys;
  orc_key keys[12];
  int action_dtype, action_elems;
  reset_fn reset;
  step_fn step;
  orc_env* envs;
  /* family config */
  int version;     /* Pendulum */
  int size;        /* FrozenLake */
  int is_slippery; /* CliffWalking */
  int natural, sab; /* Blackjack */
  int height, width; /* Catch */
} orc_pool;

/* common state keys, envpool/core/env_spec.h:37-43 */
enum {
  K_ENV_ID = 0,
  K_PLAYERS_ENV_ID,
  K_ELAPSED,
  K_DONE,
  K_REWARD,
  K_DISCOUNT,
  K_STEP_TYPE,
  K_TRUNC,
  K_FIRST_ENV_KEY
};

#define OUT_I32(out, key, row, n) (((int*)(out)[key]) + (size_t)(row) * (n))
#define OUT_F32(out, key, row, n) (((float*)(out)[key]) + (size_t)(row) * (n))
#define OUT_U8(out, key, row) (((unsigned char*)(out)[key]) + (size_t)(row))

/* Env::Allocate, envpool/core/env.h:224-256 */
static void write_common(orc_pool* p, orc_env* e, int eid, void** out, int row,
                         float reward) {
  int done = e->done;
  *OUT_I32(out, K_ENV_ID, row, 1) = eid;
  *OUT_I32(out, K_PLAYERS_ENV_ID, row, 1) = eid;
  *OUT_I32(out, K_ELAPSED, row, 1) = e->current_step;
  *OUT_U8(out, K_DONE, row) = (unsigned char)done;
  *OUT_F32(out, K_REWARD, row, 1) = reward;
  *OUT_F32(out, K_DISCOUNT, row, 1) = (float)(!done);
  int step_type = 1;
  if (e->current_step == 0) {
    step_type = 0;
  } else if (done) {
    step_type = 2;
  }
  *OUT_I32(out, K_STEP_TYPE, row, 1) = step_type;
  *OUT_U8(out, K_TRUNC, row) =
      (unsigned char)(done && (e->current_step >= p->max_episode_steps));
}

/* ------------------------------------------------------------------ */
/* CartPole: envpool/classic_control/cartpole.h:53-131                 */
static void cartpole_write(orc_env* e, void** out, int row) {
  float* obs = OUT_F32(out, K_FIRST_ENV_KEY, row, 4);
  obs[0] = (float)e->s[0];
  obs[1] = (float)e->s[1];
  obs[2] = (float)e->s[2];
  obs[3] = (float)e->s[3];
}
static void cartpole_reset(orc_pool* p, orc_env* e, void** out, int row) {
  (void)p;
  e->s[0] = orc_uniform_real(&e->gen, -0.05, 0.05); /* x */
  e->s[1] = orc_uniform_real(&e->gen, -0.05, 0.05); /* x_dot */
  e->s[2] = orc_uniform_real(&e->gen, -0.05, 0.05); /* theta */
  e->s[3] = orc_uniform_real(&e->gen, -0.05, 0.05); /* theta_dot */
  e->done = 0;
  e->elapsed_step = 0;
  cartpole_write(e, out, row);
}
static float cartpole_step(orc_pool* p, orc_env* e, const void* act, void** out,
                           int row) {
  const double kGravity = 9.8, kMassCart = 1.0, kMassPole = 0.1;
  const double kMassTotal = kMassCart + kMassPole, kLength = 0.5;
  const double kMassPoleLength = kMassPole * kLength, kForceMag = 10.0;
  const double kTau = 0.02, kThetaThresholdRadians = 12 * 2 * M_PI / 360;
  const double kXThreshold = 2.4;
  double x = e->s[0], x_dot = e->s[1], theta = e->s[2], theta_dot = e->s[3];
  e->done = (++e->elapsed_step >= p->max_episode_steps);
  int a = *(const int*)act;
  double force = a == 1 ? kForceMag : -kForceMag;
  double costheta = cos(theta);
  double sintheta = sin(theta);
  double temp =
      (force + kMassPoleLength * theta_dot * theta_dot * sintheta) / kMassTotal;
  double theta_acc =
      (kGravity * sintheta - costheta * temp) /
      (kLength * (4.0 / 3.0 - kMassPole * costheta * costheta / kMassTotal));
  double x_acc = temp - kMassPoleLength * theta_acc * costheta / kMassTotal;
  x += kTau * x_dot;
  x_dot += kTau * x_acc;
  theta += kTau * theta_dot;
  theta_dot += kTau * theta_acc;
  if (x < -kXThreshold || x > kXThreshold || theta < -kThetaThresholdRadians ||
      theta > kThetaThresholdRadians) {
    e->done = 1;
  }
  e->s[0] = x;
  e->s[1] = x_dot;
  e->s[2] = theta;
  e->s[3] = theta_dot;
  cartpole_write(e, out, row);
  return 1.0f;
}

/* ------------------------------------------------------------------ */
/* Pendulum: envpool/classic_control/pendulum.h:51-135                  */
static void pendulum_write(orc_env* e, void** out, int row) {
  float* obs = OUT_F32(out, K_FIRST_ENV_KEY, row, 3);
  obs[0] = (float)cos(e->s[0]);
  obs[1] = (float)sin(e->s[0]);
  obs[2] = (float)e->s[1];
}
static void pendulum_reset(orc_pool* p, orc_env* e, void** out, int row) {
  (void)p;
  e->s[0] = orc_uniform_real(&e->gen, -M_PI, M_PI);
  e->s[1] = orc_uniform_real(&e->gen, -1, 1);
  e->done = 0;
  e->elapsed_step = 0;
  pendulum_write(e, out, row);
}
static float pendulum_step(orc_pool* p, orc_env* e, const void* actp,
                           void** out, int row) {
  const double kMaxSpeed = 8, kMaxTorque = 2, kDt = 0.05, kGravity = 10;
  double theta = e->s[0], theta_dot = e->s[1];
  e->done = (++e->elapsed_step >= p->max_episode_steps);
  float act = *(const float*)actp;
  double u = act;
  if (act < -kMaxTorque) {
    u = -kMaxTorque;
  } else if (act > kMaxTorque) {
    u = kMaxTorque;
  }
  double cost = theta * theta + 0.1 * theta_dot * theta_dot + 0.001 * u * u;
  double new_theta_dot = theta_dot + 3 * (kGravity / 2 * sin(theta) + u) * kDt;
  if (p->version == 0) {
    theta += new_theta_dot * kDt;
  }
  theta_dot = new_theta_dot;
  if (new_theta_dot < -kMaxSpeed) {
    theta_dot = -kMaxSpeed;
  } else if (new_theta_dot > kMaxSpeed) {
    theta_dot = kMaxSpeed;
  }
  if (p->version == 1) {
    theta += new_theta_dot * kDt;
  }
  while (theta < -M_PI) theta += M_PI * 2;
  while (theta >= M_PI) theta -= M_PI * 2;
  e->s[0] = theta;
  e->s[1] = theta_dot;
  pendulum_write(e, out, row);
  return (float)(-cost);
}

/* ------------------------------------------------------------------ */
/* MountainCar: envpool/classic_control/mountain_car.h:51-130           */
static void mc_write(orc_env* e, void** out, int row) {
  float* obs = OUT_F32(out, K_FIRST_ENV_KEY, row, 2);
  obs[0] = (float)e->s[0];
  obs[1] = (float)e->s[1];
}
static void mc_reset(orc_pool* p, orc_env* e, void** out, int row) {
  (void)p;
  e->s[0] = orc_uniform_real(&e->gen, -0.6, -0.4);
  e->s[1] = 0.0;
  e->done = 0;
  e->elapsed_step = 0;
  mc_write(e, out, row);
}
static float mc_step(orc_pool* p, orc_env* e, const void* actp, void** out,
                     int row) {
  const double kMinPos = -1.2, kMaxPos = 0.6, kMaxSpeed = 0.07, kForce = 0.001;
  const double kGoalPos = 0.5, kGoalVel = 0, kGravity = 0.0025;
  double pos = e->s[0], vel = e->s[1];
  e->done = (++e->elapsed_step >= p->max_episode_steps);
  double act = *(const int*)actp - 1;
  vel += act * kForce - cos(3 * pos) * kGravity;
  if (vel < -kMaxSpeed) {
    vel = -kMaxSpeed;
  } else if (vel > kMaxSpeed) {
    vel = kMaxSpeed;
  }
  pos += vel;
  if (pos < kMinPos) {
    pos = kMinPos;
  } else if (pos > kMaxPos) {
    pos = kMaxPos;
  }
  if (pos == kMinPos && vel < 0) vel = 0;
  if (pos >= kGoalPos && vel >= kGoalVel) e->done = 1;
  e->s[0] = pos;
  e->s[1] = vel;
  mc_write(e, out, row);
  return -1.0f;
}
/* MountainCarContinuous: mountain_car_continuous.h:52-138              */
static float mcc_step(orc_pool* p, orc_env* e, const void* actp, void** out,
                      int row) {
  const double kMinPos = -1.2, kMaxPos = 0.6, kMaxSpeed = 0.07, kPower = 0.0015;
  const double kGoalPos = 0.45, kGoalVel = 0, kGravity = 0.0025;
  double pos = e->s[0], vel = e->s[1];
  e->done = (++e->elapsed_step >= p->max_episode_steps);
  double act = *(const float*)actp;
  double reward = -0.1 * act * act;
  if (act < -1) {
    act = -1;
  } else if (act > 1) {
    act = 1;
  }
  vel += act * kPower - cos(3 * pos) * kGravity;
  if (vel < -kMaxSpeed) {
    vel = -kMaxSpeed;
  } else if (vel > kMaxSpeed) {
    vel = kMaxSpeed;
  }
  pos += vel;
  if (pos < kMinPos) {
    pos = kMinPos;
  } else if (pos > kMaxPos) {
    pos = kMaxPos;
  }
  if (pos == kMinPos && vel < 0) vel = 0;
  if (pos >= kGoalPos && vel >= kGoalVel) {
    e->done = 1;
    reward += 100;
  }
  e->s[0] = pos;
  e->s[1] = vel;
  mc_write(e, out, row);
  return (float)reward;
}

/* ------------------------------------------------------------------ */
/* Acrobot: envpool/classic_control/acrobot.h:50-197                    */
typedef struct {
  double s0, s1, s2, s3, s4;
} v5;
static v5 v5_add(v5 a, v5 b) {
  v5 r = {a.s0 + b.s0, a.s1 + b.s1, a.s2 + b.s2, a.s3 + b.s3, a.s4 + b.s4};
  return r;
}
static v5 v5_mul(v5 a, double v) {
  v5 r = {a.s0 * v, a.s1 * v, a.s2 * v, a.s3 * v, a.s4 * v};
  return r;
}
static v5 acrobot_derivs(v5 s) { /* acrobot.h:156-178 */
  const double kG = 9.8, kL = 1.0, kM = 1.0, kLC = 0.5, kI = 1.0;
  double theta1 = s.s0, theta2 = s.s1, dtheta1 = s.s2, dtheta2 = s.s3;
  double a = s.s4;
  double d1 = kM * kLC * kLC +
              kM * (kL * kL + kLC * kLC + 2 * kL * kLC * cos(theta2)) + kI * 2;
  double d2 = kM * (kLC * kLC + kL * kLC * cos(theta2)) + kI;
  double phi2 = kM * kLC * kG * cos(theta1 + theta2 - M_PI / 2);
  double phi1 =
      -(dtheta2 + 2 * dtheta1) * kM * kL * kLC * dtheta2 * sin(theta2) +
      kM * (kLC + kL) * kG * cos(theta1 - M_PI / 2) + phi2;
  double ddtheta2 = (a + d2 / d1 * phi1 -
                     kM * kL * kLC * dtheta1 * dtheta1 * sin(theta2) - phi2) /
                    (kM * kLC * kLC + kI - d2 * d2 / d1);
  double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
  v5 r = {dtheta1, dtheta2, ddtheta1, ddtheta2, 0};
  return r;
}
static v5 acrobot_rk4(v5 y0) { /* acrobot.h:148-154 */
  const double kDt = 0.2;
  v5 k1 = acrobot_derivs(y0);
  v5 k2 = acrobot_derivs(v5_add(y0, v5_mul(k1, kDt / 2)));
  v5 k3 = acrobot_derivs(v5_add(y0, v5_mul(k2, kDt / 2)));
  v5 k4 = acrobot_derivs(v5_add(y0, v5_mul(k3, kDt)));
  return v5_add(
      y0, v5_mul(v5_add(v5_add(v5_add(k1, v5_mul(k2, 2)), v5_mul(k3, 2)), k4),
                 kDt / 6.0));
}
static void acrobot_write(orc_env* e, void** out, int row) {
  float* obs = OUT_F32(out, K_FIRST_ENV_KEY, row, 6);
  float* st = OUT_F32(out, K_FIRST_ENV_KEY + 1, row, 2);
  obs[0] = (float)cos(e->s[0]);
  obs[1] = (float)sin(e->s[0]);
  obs[2] = (float)cos(e->s[1]);
  obs[3] = (float)sin(e->s[1]);
  obs[4] = (float)e->s[2];
  obs[5] = (float)e->s[3];
  st[0] = (float)e->s[0];
  st[1] = (float)e->s[1];
}
static void acrobot_reset(orc_pool* p, orc_env* e, void** out, int row) {
  (void)p;
  e->s[0] = orc_uniform_real(&e->gen, -0.1, 0.1);
  e->s[1] = orc_uniform_real(&e->gen, -0.1, 0.1);
  e->s[2] = orc_uniform_real(&e->gen, -0.1, 0.1);
  e->s[3] = orc_uniform_real(&e->gen, -0.1, 0.1);
  e->s[4] = 0;
  e->done = 0;
  e->elapsed_step = 0;
  acrobot_write(e, out, row);
}
static float acrobot_step(orc_pool* p, orc_env* e, const void* actp, void** out,
                          int row) {
  const double kMaxVel1 = 4 * M_PI, kMaxVel2 = 9 * M_PI;
  e->done = (++e->elapsed_step >= p->max_episode_steps);
  int act = *(const int*)actp;
  float reward = -1.0f;
  v5 s = {e->s[0], e->s[1], e->s[2], e->s[3], e->s[4]};
  s.s4 = act - 1;
  s = acrobot_rk4(s);
  while (s.s0 < -M_PI) s.s0 += M_PI * 2;
  while (s.s1 < -M_PI) s.s1 += M_PI * 2;
  while (s.s0 >= M_PI) s.s0 -= M_PI * 2;
  while (s.s1 >= M_PI) s.s1 -= M_PI * 2;
  if (s.s2 < -kMaxVel1) s.s2 = -kMaxVel1;
  if (s.s3 < -kMaxVel2) s.s3 = -kMaxVel2;
  if (s.s2 > kMaxVel1) s.s2 = kMaxVel1;
  if (s.s3 > kMaxVel2) s.s3 = kMaxVel2;
  if (-cos(s.s0) - cos(s.s0 + s.s1) > 1) {
    e->done = 1;
    reward = 0.0f;
  }
  e->s[0] = s.s0;
  e->s[1] = s.s1;
  e->s[2] = s.s2;
  e->s[3] = s.s3;
  e->s[4] = s.s4;
  acrobot_write(e, out, row);
  return reward;
}

/* ------------------------------------------------------------------ */
/* Catch: envpool/toy_text/catch.h:49-93. i[0]=x i[1]=y i[2]=paddle     */
static void catch_write(orc_pool* p, orc_env* e, void** out, int row) {
  int n = p->height * p->width;
  float* obs = OUT_F32(out, K_FIRST_ENV_KEY, row, n);
  /* the reference writes into a freshly zero-initialised StateBuffer
   * (state_buffer_queue.h:72-85, array.h:77-81) */
  memset(obs, 0, sizeof(float) * n);
  obs[e->i[0] * p->width + e->i[1]] = 1.0f;
  obs[(p->height - 1) * p->width + e->i[2]] = 1.0f;
}
static void catch_reset(orc_pool* p, orc_env* e, void** out, int row) {
  e->i[0] = 0;
  e->i[1] = orc_uniform_int(&e->gen, 0, p->width - 1);
  e->i[2] = p->width / 2;
  e->done = 0;
  catch_write(p, e, out, row);
}
static float catch_step(orc_pool* p, orc_env* e, const void* actp, void** out,
                        int row) {
  int act = *(const int*)actp;
  float reward = 0.0f;
  e->i[2] += act - 1;
  if (e->i[2] < 0) e->i[2] = 0;
  if (e->i[2] >= p->width) e->i[2] = p->width - 1;
  if (++e->i[0] == p->height - 1) {
    e->done = 1;
    reward = e->i[1] == e->i[2] ? 1.0f : -1.0f;
  }
  catch_write(p, e, out, row);
  return reward;
}

/* ------------------------------------------------------------------ */
/* FrozenLake: envpool/toy_text/frozen_lake.h:50-111. i[0]=x i[1]=y     */
static const char* kLake4[4] = {"SFFF", "FHFH", "FFFH", "HFFG"};
static const char* kLake8[8] = {"SFFFFFFF", "FFFFFFFF", "FFFHFFFF", "FFFFFHFF",
                                "FFFHFFFF", "FHHFFFHF", "FHFFHFHF", "FFFHFFFG"};
static void lake_write(orc_pool* p, orc_env* e, void** out, int row) {
  *OUT_I32(out, K_FIRST_ENV_KEY, row, 1) = e->i[0] * p->size + e->i[1];
}
static void lake_reset(orc_pool* p, orc_env* e, void** out, int row) {
  e->i[0] = e->i[1] = 0;
  e->done = 0;
  e->elapsed_step = 0;
  lake_write(p, e, out, row);
}
static float lake_step(orc_pool* p, orc_env* e, const void* actp, void** out,
                       int row) {
  e->done = (++e->elapsed_step >= p->max_episode_steps);
  int act = *(const int*)actp;
  act = (act + orc_uniform_int(&e->gen, -1, 1) + 4) % 4;
  int x = e->i[0], y = e->i[1];
  if (act == 0) {
    --y;
  } else if (act == 1) {
    ++x;
  } else if (act == 2) {
    ++y;
  } else {
    --x;
  }
  int hi = p->size - 1;
  x = x < 0 ? 0 : (x > hi ? hi : x);
  y = y < 0 ? 0 : (y > hi ? hi : y);
  char c = p->size != 8 ? kLake4[x][y] : kLake8[x][y];
  float reward = 0.0f;
  if (c == 'H' || c == 'G') {
    e->done = 1;
    reward = c == 'G' ? 1.0f : 0.0f;
  }
  e->i[0] = x;
  e->i[1] = y;
  lake_write(p, e, out, row);
  return reward;
}

/* ------------------------------------------------------------------ */
/* Taxi: envpool/toy_text/taxi.h:48-130. i[0]=x i[1]=y i[2]=s i[3]=t    */
static const int kTaxiLoc[4][2] = {{0, 0}, {0, 4}, {4, 0}, {4, 3}};
static const char* kTaxiMap[5] = {"|:|::|", "|:|::|", "|::::|", "||:|:|",
                                  "||:|:|"};
static const char* kTaxiLocMap[5] = {"0   1", "     ", "     ", "     ",
                                     "2  3 "};
static void taxi_write(orc_env* e, void** out, int row) {
  *OUT_I32(out, K_FIRST_ENV_KEY, row, 1) =
      ((e->i[0] * 5 + e->i[1]) * 5 + e->i[2]) * 4 + e->i[3];
}
static void taxi_reset(orc_pool* p, orc_env* e, void** out, int row) {
  (void)p;
  e->i[0] = orc_uniform_int(&e->gen, 0, 4);
  e->i[1] = orc_uniform_int(&e->gen, 0, 4);
  e->i[2] = orc_uniform_int(&e->gen, 0, 3);
  e->i[3] = orc_uniform_int(&e->gen, 0, 3);
  e->done = 0;
  e->elapsed_step = 0;
  taxi_write(e, out, row);
}
static float taxi_step(orc_pool* p, orc_env* e, const void* actp, void** out,
                       int row) {
  e->done = (++e->elapsed_step >= p->max_episode_steps);
  int act = *(const int*)actp;
  int x = e->i[0], y = e->i[1], s = e->i[2], t = e->i[3];
  float reward = -1.0f;
  if (act == 0) {
    if (x < 4) ++x;
  } else if (act == 1) {
    if (x > 0) --x;
  } else if (act == 2) {
    if (kTaxiMap[x][y + 1] == ':') ++y;
  } else if (act == 3) {
    if (kTaxiMap[x][y] == ':') --y;
  } else if (act == 4) {
    if (s < 4 && x == kTaxiLoc[s][0] && y == kTaxiLoc[s][1]) {
      s = 4;
    } else {
      reward = -10.0f;
    }
  } else {
    if (s == 4 && x == kTaxiLoc[t][0] && y == kTaxiLoc[t][1]) {
      s = t;
      e->done = 1;
      reward = 20.0f;
    } else if (s == 4 && kTaxiLocMap[x][y] != ' ') {
      s = kTaxiLocMap[x][y] - '0';
    } else {
      reward = -10.0f;
    }
  }
  e->i[0] = x;
  e->i[1] = y;
  e->i[2] = s;
  e->i[3] = t;
  taxi_write(e, out, row);
  return reward;
}

/* ------------------------------------------------------------------ */
/* NChain: envpool/toy_text/nchain.h:47-97. i[0]=s                      */
static void nchain_reset(orc_pool* p, orc_env* e, void** out, int row) {
  (void)p;
  e->i[0] = 0;
  e->done = 0;
  e->elapsed_step = 0;
  *OUT_I32(out, K_FIRST_ENV_KEY, row, 1) = 0;
}
static float nchain_step(orc_pool* p, orc_env* e, const void* actp, void** out,
                         int row) {
  e->done = (++e->elapsed_step >= p->max_episode_steps);
  int act = *(const int*)actp;
  if (orc_uniform_real(&e->gen, 0, 1) < 0.2) act = 1 - act;
  float reward = 0.0f;
  if (act != 0) {
    reward = 2.0f;
    e->i[0] = 0;
  } else if (e->i[0] < 4) {
    ++e->i[0];
  } else {
    reward = 10.0f;
  }
  *OUT_I32(out, K_FIRST_ENV_KEY, row, 1) = e->i[0];
  return reward;
}

/* ------------------------------------------------------------------ */
/* CliffWalking: envpool/toy_text/cliffwalking.h:50-113. i[0]=x i[1]=y  */
static void cliff_write(orc_env* e, void** out, int row, float prob) {
  *OUT_I32(out, K_FIRST_ENV_KEY, row, 1) = e->i[0] * 12 + e->i[1];
  *OUT_F32(out, K_FIRST_ENV_KEY + 1, row, 1) = prob;
}
static void cliff_reset(orc_pool* p, orc_env* e, void** out, int row) {
  (void)p;
  e->i[0] = 3;
  e->i[1] = 0;
  e->done = 0;
  cliff_write(e, out, row, 1.0f);
}
static float cliff_step(orc_pool* p, orc_env* e, const void* actp, void** out,
                        int row) {
  int act = *(const int*)actp;
  if (p->is_slippery) { /* SampleAction, cliffwalking.h:97-104 */
    static const int k_offsets[3] = {-1, 0, 1};
    act = (act + k_offsets[orc_uniform_int(&e->gen, 0, 2)] + 4) % 4;
  }
  int x = e->i[0], y = e->i[1];
  float reward = -1.0f;
  if (act == 0) {
    --x;
  } else if (act == 1) {
    ++y;
  } else if (act == 2) {
    ++x;
  } else {
    --y;
  }
  x = x > 3 ? 3 : (x < 0 ? 0 : x);
  y = y > 11 ? 11 : (y < 0 ? 0 : y);
  if (x == 3 && y > 0 && y < 11) {
    reward = -100.0f;
    x = 3;
    y = 0;
  }
  if (x == 3 && y == 11) e->done = 1;
  e->i[0] = x;
  e->i[1] = y;
  cliff_write(e, out, row, p->is_slippery ? 1.0f / 3.0f : 1.0f);
  return reward;
}

/* ------------------------------------------------------------------ */
/* Blackjack: envpool/toy_text/blackjack.h:49-152.
 * The reference keeps std::vector hands; only sum / any-ace / size / first
 * two cards / dealer_[0] are ever read, so hands are folded to:
 *   player: i[0]=sum i[1]=has_ace i[2]=count i[3],i[4]=first two cards
 *   dealer: i[5]=sum i[6]=has_ace i[7]=count i[8],i[9]=first two cards     */
static int bj_draw(orc_env* e) { /* DrawCard, blackjack.h:112 */
  int c = orc_uniform_int(&e->gen, 1, 13);
  return c < 10 ? c : 10;
}
static void bj_push(int* h, int card) {
  if (h[2] < 2) h[3 + h[2]] = card;
  h[0] += card;
  if (card == 1) h[1] = 1;
  h[2] += 1;
}
static int bj_sum_hand(const int* h) { /* SumHand :123-132 */
  if (h[1] != 0 && h[0] + 10 <= 21) return h[0] + 10;
  return h[0];
}
static int bj_score(const int* h) { /* Score :138-141 */
  int r = bj_sum_hand(h);
  return r > 21 ? 0 : r;
}
static int bj_is_natural(const int* h) { /* IsNatural :143-146 */
  return h[2] == 2 && ((h[3] == 1 && h[4] == 10) || (h[3] == 10 && h[4] == 1));
}
static void bj_write(orc_env* e, void** out, int row) {
  int* obs = OUT_I32(out, K_FIRST_ENV_KEY, row, 3);
  obs[0] = bj_sum_hand(e->i);
  obs[1] = e->i[8];
  obs[2] = e->i[1];
}
static void bj_reset(orc_pool* p, orc_env* e, void** out, int row) {
  (void)p;
  memset(e->i, 0, sizeof(int) * 10);
  bj_push(e->i, bj_draw(e));
  bj_push(e->i, bj_draw(e));
  bj_push(e->i + 5, bj_draw(e));
  bj_push(e->i + 5, bj_draw(e));
  e->done = 0;
  bj_write(e, out, row);
}
static float bj_step(orc_pool* p, orc_env* e, const void* actp, void** out,
                     int row) {
  int act = *(const int*)actp;
  float reward = 0.0f;
  if (act != 0) {
    bj_push(e->i, bj_draw(e));
    if (bj_sum_hand(e->i) > 21) {
      e->done = 1;
      reward = -1.0f;
    }
  } else {
    e->done = 1;
    while (bj_sum_hand(e->i + 5) < 17) bj_push(e->i + 5, bj_draw(e));
    int ps = bj_score(e->i), ds = bj_score(e->i + 5);
    reward = (ps > ds ? 1.0f : 0.0f) - (ps < ds ? 1.0f : 0.0f);
    if (p->sab && bj_is_natural(e->i) && !bj_is_natural(e->i + 5)) {
      reward = 1.0f;
    } else if (!p->sab && p->natural && bj_is_natural(e->i) && reward == 1.0f) {
      reward = 1.5f;
    }
  }
  bj_write(e, out, row);
  return reward;
}

/* ------------------------------------------------------------------ */
typedef float (*step_reward_fn)(orc_pool*, orc_env*, const void*, void**, int);

typedef struct {
  const char* task;
  reset_fn reset;
  step_reward_fn step;
  int action_dtype;
  int n_env_keys;
  orc_key env_keys[2];
} family;

static const family kFamilies[] = {
    {"CartPole", cartpole_reset, cartpole_step, DT_I32, 1,
     {{"obs", DT_F32, 4}}},
    {"Pendulum", pendulum_reset, pendulum_step, DT_F32, 1,
     {{"obs", DT_F32, 3}}},
    {"MountainCar", mc_reset, mc_step, DT_I32, 1, {{"obs", DT_F32, 2}}},
    {"MountainCarContinuous", mc_reset, mcc_step, DT_F32, 1,
     {{"obs", DT_F32, 2}}},
    {"Acrobot", acrobot_reset, acrobot_step, DT_I32, 2,
     {{"obs", DT_F32, 6}, {"info:state", DT_F32, 2}}},
    {"Catch", catch_reset, catch_step, DT_I32, 1, {{"obs", DT_F32, 50}}},
    {"FrozenLake", lake_reset, lake_step, DT_I32, 1, {{"obs", DT_I32, 1}}},
    {"Taxi", taxi_reset, taxi_step, DT_I32, 1, {{"obs", DT_I32, 1}}},
    {"NChain", nchain_reset, nchain_step, DT_I32, 1, {{"obs", DT_I32, 1}}},
    {"CliffWalking", cliff_reset, cliff_step, DT_I32, 2,
     {{"obs", DT_I32, 1}, {"info:prob", DT_F32, 1}}},
    {"Blackjack", bj_reset, bj_step, DT_I32, 1, {{"obs", DT_I32, 3}}},
};

typedef struct {
  orc_pool p;
  step_reward_fn step_reward;
} pool_impl;

static double extra_or(const double* extra, int n, int i, double d) {
  return (extra && i < n) ? extra[i] : d;
}

void* restate_create(const char* task, int num_envs, int seed,
                     int max_episode_steps, const double* extra, int n_extra) {
  const family* f = NULL;
  for (size_t i = 0; i < sizeof(kFamilies) / sizeof(kFamilies[0]); ++i) {
    if (strcmp(kFamilies[i].task, task) == 0) f = &kFamilies[i];
  }
  if (!f) return NULL;
  pool_impl* pi = (pool_impl*)calloc(1, sizeof(pool_impl));
  orc_pool* p = &pi->p;
  p->num_envs = num_envs;
  /* common_config default: numeric_limits<int>::max(), env_spec.h:31 */
  p->max_episode_steps = max_episode_steps > 0 ? max_episode_steps : INT_MAX;
  static const orc_key common[8] = {
      {"info:env_id", DT_I32, 1}, {"info:players.env_id", DT_I32, 1},
      {"elapsed_step", DT_I32, 1}, {"done", DT_BOOL, 1},
      {"reward", DT_F32, 1},       {"discount", DT_F32, 1},
      {"step_type", DT_I32, 1},    {"trunc", DT_BOOL, 1}};
  memcpy(p->keys, common, sizeof(common));
  p->nkeys = 8 + f->n_env_keys;
  for (int i = 0; i < f->n_env_keys; ++i) p->keys[8 + i] = f->env_keys[i];
  p->action_dtype = f->action_dtype;
  p->action_elems = 1;
  p->reset = f->reset;
  pi->step_reward = f->step;
  p->version = (int)extra_or(extra, n_extra, 0, 0);
  p->size = (int)extra_or(extra, n_extra, 0, 4);
  p->is_slippery = extra_or(extra, n_extra, 0, 0) != 0;
  p->natural = extra_or(extra, n_extra, 0, 0) != 0;
  p->sab = extra_or(extra, n_extra, 1, 1) != 0;
  p->height = (int)extra_or(extra, n_extra, 0, 10);
  p->width = (int)extra_or(extra, n_extra, 1, 5);
  if (strcmp(task, "Catch") == 0) p->keys[8].elems = p->height * p->width;
  p->envs = (orc_env*)calloc((size_t)num_envs, sizeof(orc_env));
  for (int i = 0; i < num_envs; ++i) {
    orc_env* e = &p->envs[i];
    orc_mt_seed(&e->gen, (uint32_t)(seed + i)); /* env.h:109,117 */
    e->current_step = -1;
    e->done = 1;
    e->elapsed_step = p->max_episode_steps + 1; /* cartpole.h:66 (unused) */
  }
  return pi;
}

int restate_num_state_keys(void* h) { return ((pool_impl*)h)->p.nkeys; }

int restate_state_key(void* h, int i, char* name, int* dtype, int* elems) {
  orc_pool* p = &((pool_impl*)h)->p;
  if (i < 0 || i >= p->nkeys) return -1;
  strncpy(name, p->keys[i].name, 63);
  name[63] = 0;
  *dtype = p->keys[i].dtype;
  *elems = p->keys[i].elems;
  return 0;
}

int restate_action_info(void* h, int* dtype, int* elems) {
  orc_pool* p = &((pool_impl*)h)->p;
  *dtype = p->action_dtype;
  *elems = p->action_elems;
  return 0;
}

static void env_step(pool_impl* pi, int eid, int force_reset, const void* act,
                     void** out, int row) {
  orc_pool* p = &pi->p;
  orc_env* e = &p->envs[eid];
  int reset = force_reset || e->done; /* async_envpool.h:127 */
  float reward = 0.0f;
  if (reset) { /* env.h:207-217 */
    e->current_step = 0;
    p->reset(p, e, out, row);
  } else {
    ++e->current_step;
    reward = pi->step_reward(p, e, act, out, row);
  }
  write_common(p, e, eid, out, row, reward);
}

void restate_reset(void* h, const int* ids, int k, void** out) {
  pool_impl* pi = (pool_impl*)h;
  for (int i = 0; i < k; ++i) env_step(pi, ids[i], 1, NULL, out, i);
}

void restate_step(void* h, const int* ids, int k, const void* action,
                  void** out) {
  pool_impl* pi = (pool_impl*)h;
  const char* a = (const char*)action;
  for (int i = 0; i < k; ++i) env_step(pi, ids[i], 0, a + 4 * (size_t)i, out, i);
}

void restate_destroy(void* h) {
  pool_impl* pi = (pool_impl*)h;
  free(pi->p.envs);
  free(pi);
}

/* Test hooks: flat per-env state [s0..s4, done, current_step] (classic) so a
 * test can teacher-force the HIP engine from the oracle's exact fp64 state. */
void restate_get_state(void* h, const int* ids, int k, double* out) {
  orc_pool* p = &((pool_impl*)h)->p;
  for (int i = 0; i < k; ++i) {
    orc_env* e = &p->envs[ids[i]];
    for (int j = 0; j < 5; ++j) out[i * 7 + j] = e->s[j];
    out[i * 7 + 5] = e->done;
    out[i * 7 + 6] = e->current_step;
  }
}
void restate_set_state(void* h, const int* ids, int k, const double* in) {
  orc_pool* p = &((pool_impl*)h)->p;
  for (int i = 0; i < k; ++i) {
    orc_env* e = &p->envs[ids[i]];
    for (int j = 0; j < 5; ++j) e->s[j] = in[i * 7 + j];
    e->done = in[i * 7 + 5] != 0.0;
    e->current_step = (int)in[i * 7 + 6];
    e->elapsed_step = e->current_step;
  }
}
