/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Small fp64 vector / quaternion /
 * Cholesky helpers for oracle/mjcpu. */
#ifndef ORACLE_MJMATH_H_
#define ORACLE_MJMATH_H_
#include <math.h>

#include "mjcpu.h"

static inline void v3_copy(double* r, const double* a) {
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
}
static inline void v3_add(double* r, const double* a, const double* b) {
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
}
static inline void v3_sub(double* r, const double* a, const double* b) {
  r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2];
}
static inline void v3_scale(double* r, const double* a, double s) {
  r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s;
}
static inline void v3_addscl(double* r, const double* a, double s) {
  r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s;
}
static inline double v3_dot(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static inline void v3_cross(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline double v3_norm(const double* a) { return sqrt(v3_dot(a, a)); }
static inline double v3_normalize(double* a) {
  double n = v3_norm(a);
  if (n < 1e-15) { a[0] = 1; a[1] = a[2] = 0; return n; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
/* r = M v, M row-major 3x3 */
static inline void m3_mulvec(double* r, const double* M, const double* v) {
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void m3_transpose(double* r, const double* M) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r[3 * i + j] = M[3 * j + i];
}
static inline void m3_mul(double* r, const double* A, const double* B) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
    r[3 * i + j] = s;
  }
}
static inline void quat_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static inline void quat_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static inline void quat2mat(double* M, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = w * w + x * x - y * y - z * z;
  M[4] = w * w - x * x + y * y - z * z;
  M[8] = w * w - x * x - y * y + z * z;
  M[1] = 2 * (x * y - w * z); M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z); M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y); M[7] = 2 * (y * z + w * x);
}
static inline void quat_rotvec(double* r, const double* q, const double* v) {
  double M[9];
  quat2mat(M, q);
  m3_mulvec(r, M, v);
}
static inline void quat_axisangle(double* q, const double* axis, double ang) {
  double s = sin(ang / 2);
  q[0] = cos(ang / 2); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* in-place lower Cholesky of a dense n x n SPD matrix (row-major, stride n) */
static inline int chol_factor(double* A, int n) {
  for (int j = 0; j < n; ++j) {
    double s = A[j * n + j];
    for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
    if (s <= 0) return -1;
    double d = sqrt(s);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double t = A[i * n + j];
      for (int k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / d;
    }
  }
  return 0;
}
static inline void chol_solve(const double* L, int n, double* x) {
  for (int i = 0; i < n; ++i) {
    double s = x[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

/* engine.c: translational / rotational Jacobian of a world point attached to
 * `body` (MuJoCo mj_jac) */
void mjc_jac(const mjc_model* m, const mjc_data* d, double jacp[3][MJC_MAXV],
             double jacr[3][MJC_MAXV], const double point[3], int body);

#endif /* ORACLE_MJMATH_H_ */
