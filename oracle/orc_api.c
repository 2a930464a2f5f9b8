/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 * `orc_*` C API of the plain-C oracle ("port"): dispatches a task name to the
 * restated family (oracle/restate/envs.c for classic_control / toy_text,
 * oracle/mjcpu for the MuJoCo tasks).  Same signatures as oracle/ref_driver.cc.
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

void* restate_create(const char*, int, int, int, const double*, int);
int restate_num_state_keys(void*);
int restate_state_key(void*, int, char*, int*, int*);
int restate_action_info(void*, int*, int*);
void restate_reset(void*, const int*, int, void**);
void restate_step(void*, const int*, int, const void*, void**);
void restate_destroy(void*);
void restate_get_state(void*, const int*, int, double*);
void restate_set_state(void*, const int*, int, const double*);

void* mjcpu_create(const char*, int, int, int, const double*, int);
int mjcpu_num_state_keys(void*);
int mjcpu_state_key(void*, int, char*, int*, int*);
int mjcpu_action_info(void*, int*, int*);
void mjcpu_reset(void*, const int*, int, void**);
void mjcpu_step(void*, const int*, int, const void*, void**);
void mjcpu_destroy(void*);
int mjcpu_state_dim(void*);
void mjcpu_get_state(void*, const int*, int, double*);
void mjcpu_set_state(void*, const int*, int, const double*);

typedef struct {
  int kind; /* 0 restate, 1 mjcpu */
  void* h;
  int num_envs;
} handle;

void* orc_create(const char* task, int num_envs, int seed, int max_episode_steps,
                 const double* extra, int n_extra, int num_threads) {
  (void)num_threads;
  handle* hd = (handle*)calloc(1, sizeof(handle));
  hd->num_envs = num_envs;
  hd->h = restate_create(task, num_envs, seed, max_episode_steps, extra, n_extra);
  hd->kind = 0;
  if (!hd->h) {
    hd->h = mjcpu_create(task, num_envs, seed, max_episode_steps, extra, n_extra);
    hd->kind = 1;
  }
  if (!hd->h) {
    free(hd);
    return NULL;
  }
  return hd;
}

int orc_num_state_keys(void* h) {
  handle* hd = (handle*)h;
  return hd->kind == 0 ? restate_num_state_keys(hd->h) : mjcpu_num_state_keys(hd->h);
}
int orc_state_key(void* h, int i, char* name, int* dtype, int* elems) {
  handle* hd = (handle*)h;
  return hd->kind == 0 ? restate_state_key(hd->h, i, name, dtype, elems)
                       : mjcpu_state_key(hd->h, i, name, dtype, elems);
}
int orc_action_info(void* h, int* dtype, int* elems) {
  handle* hd = (handle*)h;
  return hd->kind == 0 ? restate_action_info(hd->h, dtype, elems)
                       : mjcpu_action_info(hd->h, dtype, elems);
}
void orc_reset(void* h, const int* ids, int k, void** out) {
  handle* hd = (handle*)h;
  if (hd->kind == 0) restate_reset(hd->h, ids, k, out);
  else mjcpu_reset(hd->h, ids, k, out);
}
void orc_step(void* h, const int* ids, int k, const void* action, void** out) {
  handle* hd = (handle*)h;
  if (hd->kind == 0) restate_step(hd->h, ids, k, action, out);
  else mjcpu_step(hd->h, ids, k, action, out);
}

/* Timed loop of `steps` full-batch steps (single thread; outputs written to a
 * scratch block that is reused, which favours the port over the reference). */
double orc_time_steps(void* h, int steps, const void* action) {
  handle* hd = (handle*)h;
  int nk = orc_num_state_keys(h), n = hd->num_envs;
  void** out = (void**)calloc((size_t)nk, sizeof(void*));
  char name[64];
  for (int i = 0; i < nk; ++i) {
    int dt, el;
    orc_state_key(h, i, name, &dt, &el);
    out[i] = calloc((size_t)n * el, 8);
  }
  int* ids = (int*)malloc(sizeof(int) * n);
  for (int i = 0; i < n; ++i) ids[i] = i;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int s = 0; s < steps; ++s) orc_step(h, ids, n, action, out);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (int i = 0; i < nk; ++i) free(out[i]);
  free(out);
  free(ids);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

void orc_destroy(void* h) {
  handle* hd = (handle*)h;
  if (hd->kind == 0) restate_destroy(hd->h);
  else mjcpu_destroy(hd->h);
  free(hd);
}

/* flat per-env state for teacher-forced parity tests */
int orc_state_dim(void* h) {
  handle* hd = (handle*)h;
  return hd->kind == 0 ? 7 : mjcpu_state_dim(hd->h);
}
void orc_get_state(void* h, const int* ids, int k, double* out) {
  handle* hd = (handle*)h;
  if (hd->kind == 0) restate_get_state(hd->h, ids, k, out);
  else mjcpu_get_state(hd->h, ids, k, out);
}
void orc_set_state(void* h, const int* ids, int k, const double* in) {
  handle* hd = (handle*)h;
  if (hd->kind == 0) restate_set_state(hd->h, ids, k, in);
  else mjcpu_set_state(hd->h, ids, k, in);
}

const char* orc_kind(void) { return "port"; }
