/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See mjcpu.h (PARITY UNPINNED).
 *
 * fp64 restatement of the MuJoCo 3.6.0 forward pipeline + integrators for
 * tree-structured models with free/slide/hinge joints, plane/sphere/capsule
 * geoms and joint motors.  Stage names follow MuJoCo's engine functions
 * (SURVEY.md §8a M1-M9, Appendix A.2-A.8); each function says which.
 * Spatial vectors are [rot(3); lin(3)], expressed in world orientation about
 * the subtree COM of the kinematic tree's root body, as MuJoCo's c-frame.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mjcpu.h"
#include "mjmath.h"

/* pipeline-stage markers of the operation-counting build (oracle/flopcount/count_real.h,
 * -DMJC_COUNT_FLOPS: rows M1-M9 of SURVEY.md section 8a); nothing otherwise */
#ifndef MJC_STAGE
#define MJC_STAGE(k) ((void)0)
#endif

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999

void mjc_reset_data(const mjc_model* m, mjc_data* d) { /* mj_resetData */
  memset(d, 0, sizeof(*d));
  for (int i = 0; i < m->nq; ++i) d->qpos[i] = m->qpos0[i];
}

/* ---- M1: mj_kinematics --------------------------------------------------- */
static void kinematics(const mjc_model* m, mjc_data* d) {
  /* MuJoCo >= 3.1.4 (changelog: "quaternions in mjData.qpos are no longer normalised
   * in place by mj_kinematics; they are normalised when they are used") -- so qpos keeps
   * whatever the task wrote (the reset observation of Ant / Humanoid shows the raw
   * init_qpos + noise quaternion) and only the local copy below is normalised.  After the
   * first mj_step qpos is unit again (mj_integratePos normalises its result). */
  memset(d->xpos[0], 0, sizeof(d->xpos[0]));
  d->xquat[0][0] = 1;
  d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  quat2mat(d->xmat[0], d->xquat[0]);
  memset(d->xipos[0], 0, sizeof(d->xipos[0]));
  for (int b = 1; b < m->nbody; ++b) {
    int p = m->body_parent[b];
    double xpos[3], xquat[4], tmp[3];
    int ja = m->body_jntadr[b], jn = m->body_jntnum[b];
    if (jn == 1 && m->jnt_type[ja] == MJC_JNT_FREE) {
      int qa = m->jnt_qposadr[ja];
      v3_copy(xpos, d->qpos + qa);
      for (int i = 0; i < 4; ++i) xquat[i] = d->qpos[qa + 3 + i];
      quat_normalize(xquat);
      v3_copy(d->xanchor[ja], xpos);
      v3_copy(d->xaxis[ja], m->jnt_axis[ja]);
    } else {
      m3_mulvec(tmp, d->xmat[p], m->body_pos[b]);
      v3_add(xpos, d->xpos[p], tmp);
      quat_mul(xquat, d->xquat[p], m->body_quat[b]);
      for (int j = ja; j < ja + jn; ++j) {
        quat_rotvec(tmp, xquat, m->jnt_pos[j]);
        v3_add(d->xanchor[j], tmp, xpos);
        quat_rotvec(d->xaxis[j], xquat, m->jnt_axis[j]);
        double q = d->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]];
        if (m->jnt_type[j] == MJC_JNT_SLIDE) {
          v3_addscl(xpos, d->xaxis[j], q);
        } else { /* hinge: rotate about local axis, correct off-centre */
          double qloc[4], nq[4];
          quat_axisangle(qloc, m->jnt_axis[j], q);
          quat_mul(nq, xquat, qloc);
          for (int i = 0; i < 4; ++i) xquat[i] = nq[i];
          quat_rotvec(tmp, xquat, m->jnt_pos[j]);
          v3_sub(xpos, d->xanchor[j], tmp);
        }
      }
    }
    quat_normalize(xquat);
    v3_copy(d->xpos[b], xpos);
    for (int i = 0; i < 4; ++i) d->xquat[b][i] = xquat[i];
    quat2mat(d->xmat[b], xquat);
    m3_mulvec(tmp, d->xmat[b], m->body_ipos[b]);
    v3_add(d->xipos[b], xpos, tmp);
  }
  for (int g = 0; g < m->ngeom; ++g) {
    int b = m->geom_body[g];
    double tmp[3], q[4];
    m3_mulvec(tmp, d->xmat[b], m->geom_pos[g]);
    v3_add(d->geom_xpos[g], d->xpos[b], tmp);
    quat_mul(q, d->xquat[b], m->geom_quat[g]);
    quat_normalize(q);
    quat2mat(d->geom_xmat[g], q);
  }
}

/* ---- M1: mj_comPos --------------------------------------------------------- */
static void com_pos(const mjc_model* m, mjc_data* d) {
  double smass[MJC_MAXBODY];
  for (int b = 0; b < m->nbody; ++b) {
    smass[b] = m->body_mass[b];
    v3_scale(d->subtree_com[b], d->xipos[b], m->body_mass[b]);
  }
  for (int b = m->nbody - 1; b > 0; --b) {
    int p = m->body_parent[b];
    smass[p] += smass[b];
    v3_add(d->subtree_com[p], d->subtree_com[p], d->subtree_com[b]);
  }
  for (int b = 0; b < m->nbody; ++b) {
    if (smass[b] < MINVAL) {
      v3_copy(d->subtree_com[b], d->xipos[b]);
    } else {
      v3_scale(d->subtree_com[b], d->subtree_com[b], 1.0 / smass[b]);
    }
  }
  /* cinert: body inertia about subtree_com[root], world orientation */
  for (int b = 1; b < m->nbody; ++b) {
    const double* c = d->subtree_com[m->body_rootid[b]];
    double Rt[9], RI[9], Iw[9], off[3];
    m3_mul(RI, d->xmat[b], m->body_inertia[b]);
    m3_transpose(Rt, d->xmat[b]);
    m3_mul(Iw, RI, Rt);
    v3_sub(off, d->xipos[b], c);
    double mass = m->body_mass[b], o2 = v3_dot(off, off);
    double* ci = d->cinert[b];
    ci[0] = Iw[0] + mass * (o2 - off[0] * off[0]);
    ci[1] = Iw[4] + mass * (o2 - off[1] * off[1]);
    ci[2] = Iw[8] + mass * (o2 - off[2] * off[2]);
    ci[3] = Iw[1] - mass * off[0] * off[1];
    ci[4] = Iw[2] - mass * off[0] * off[2];
    ci[5] = Iw[5] - mass * off[1] * off[2];
    ci[6] = mass * off[0];
    ci[7] = mass * off[1];
    ci[8] = mass * off[2];
    ci[9] = mass;
  }
  memset(d->cinert[0], 0, sizeof(d->cinert[0]));
  /* cdof */
  for (int j = 0; j < m->njnt; ++j) {
    int b = m->jnt_body[j], a = m->jnt_dofadr[j];
    const double* c = d->subtree_com[m->body_rootid[b]];
    double off[3];
    v3_sub(off, c, d->xanchor[j]);
    if (m->jnt_type[j] == MJC_JNT_SLIDE) {
      memset(d->cdof[a], 0, sizeof(double) * 3);
      v3_copy(d->cdof[a] + 3, d->xaxis[j]);
    } else if (m->jnt_type[j] == MJC_JNT_HINGE) {
      v3_copy(d->cdof[a], d->xaxis[j]);
      v3_cross(d->cdof[a] + 3, d->xaxis[j], off);
    } else { /* free: 3 world translations, 3 body-frame rotations */
      for (int k = 0; k < 3; ++k) {
        memset(d->cdof[a + k], 0, sizeof(double) * 6);
        d->cdof[a + k][3 + k] = 1;
        double ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]};
        v3_copy(d->cdof[a + 3 + k], ax);
        v3_cross(d->cdof[a + 3 + k] + 3, ax, off);
      }
    }
  }
}

/* spatial inertia (10-vector) times motion vector */
static void mul_inert_vec(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] +
         a[5] * b[5];
}
static void cross_motion(double* r, const double* vel, const double* v) {
  double a[3], b[3];
  v3_cross(r, vel, v);
  v3_cross(a, vel, v + 3);
  v3_cross(b, vel + 3, v);
  v3_add(r + 3, a, b);
}
static void cross_force(double* r, const double* vel, const double* f) {
  double a[3], b[3];
  v3_cross(a, vel, f);
  v3_cross(b, vel + 3, f + 3);
  v3_add(r, a, b);
  v3_cross(r + 3, vel, f + 3);
}

/* ---- M2: mj_crb (+ armature); factorisation happens at the solves --------- */
static void crb(const mjc_model* m, mjc_data* d) {
  double crbI[MJC_MAXBODY][10];
  memcpy(crbI, d->cinert, sizeof(crbI));
  for (int b = m->nbody - 1; b > 0; --b) {
    int p = m->body_parent[b];
    if (p > 0) {
      for (int k = 0; k < 10; ++k) crbI[p][k] += crbI[b][k];
    }
  }
  memset(d->M, 0, sizeof(d->M));
  for (int i = 0; i < m->nv; ++i) {
    double buf[6];
    mul_inert_vec(buf, crbI[m->dof_body[i]], d->cdof[i]);
    for (int j = i; j >= 0; j = m->dof_parent[j]) {
      double v = dot6(d->cdof[j], buf);
      d->M[i][j] = v;
      d->M[j][i] = v;
    }
    d->M[i][i] += m->dof_armature[i];
  }
}

/* ---- mj_jac ------------------------------------------------------------------ */
void mjc_jac(const mjc_model* m, const mjc_data* d, double jacp[3][MJC_MAXV],
             double jacr[3][MJC_MAXV], const double point[3], int body) {
  for (int r = 0; r < 3; ++r) {
    for (int i = 0; i < m->nv; ++i) jacp[r][i] = jacr[r][i] = 0;
  }
  int b = body;
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parent[b];
  if (b == 0) return;
  double off[3];
  v3_sub(off, point, d->subtree_com[m->body_rootid[body]]);
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0;
       i = m->dof_parent[i]) {
    double tmp[3];
    v3_cross(tmp, d->cdof[i], off);
    for (int r = 0; r < 3; ++r) {
      jacr[r][i] = d->cdof[i][r];
      jacp[r][i] = d->cdof[i][3 + r] + tmp[r];
    }
  }
}

/* ---- M3: mj_collision (plane-sphere, plane-capsule, capsule-capsule) -------------- */
static void make_frame(double* frame) { /* mju_makeFrame */
  double* x = frame;
  double* y = frame + 3;
  v3_normalize(x);
  y[0] = y[1] = y[2] = 0;
  if (x[1] < 0.5 && x[1] > -0.5) {
    y[1] = 1;
  } else {
    y[2] = 1;
  }
  double t = v3_dot(x, y);
  v3_addscl(y, x, -t);
  v3_normalize(y);
  v3_cross(frame + 6, x, y);
}

/* mj_contactParam for a geom pair: condim max, friction max, solref / solimp
 * mixed with solmix 1:1 */
static void contact_params(const mjc_model* m, mjc_contact* c, int g1, int g2, double margin) {
  c->geom1 = g1;
  c->geom2 = g2;
  c->includemargin = margin;
  c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
  c->friction = fmax(m->geom_friction[g1][0], m->geom_friction[g2][0]);
  for (int i = 0; i < 2; ++i) {
    c->solref[i] = 0.5 * (m->geom_solref[g1][i] + m->geom_solref[g2][i]);
  }
  for (int i = 0; i < 5; ++i) {
    c->solimp[i] = 0.5 * (m->geom_solimp[g1][i] + m->geom_solimp[g2][i]);
  }
}

/* mjraw_SphereSphere on two points of the capsule axes */
static void add_sphere_sphere(const mjc_model* m, mjc_data* d, int g1, int g2, const double* p1,
                              double r1, const double* p2, double r2, double margin) {
  double dif[3];
  v3_sub(dif, p2, p1);
  double cdist = v3_norm(dif);
  if (cdist > margin + r1 + r2) return;
  if (d->ncon >= MJC_MAXCON) return;
  mjc_contact* c = &d->contact[d->ncon++];
  c->dist = cdist - r1 - r2;
  if (cdist < 1e-15) { /* coincident centres: MuJoCo picks +x */
    c->frame[0] = 1;
    c->frame[1] = c->frame[2] = 0;
  } else {
    for (int k = 0; k < 3; ++k) c->frame[k] = dif[k] / cdist;
  }
  for (int k = 0; k < 3; ++k) c->pos[k] = p1[k] + c->frame[k] * (r1 + 0.5 * c->dist);
  make_frame(c->frame);
  contact_params(m, c, g1, g2, margin);
}

/* mjraw_CapsuleCapsule: closest points of the two axis segments, then sphere-sphere.
 * Exactly parallel axes (|det| < mjMINVAL) make MuJoCo emit up to two contacts from the
 * end-point projections; here the overlap midpoint is used (one contact) -- a
 * measure-zero configuration for the hopper, flagged in DESIGN.md. */
static void add_capsule_capsule(const mjc_model* m, mjc_data* d, int g1, int g2, double margin) {
  const double *m1 = d->geom_xmat[g1], *m2 = d->geom_xmat[g2];
  double h1 = m->geom_size[g1][1], h2 = m->geom_size[g2][1];
  double a1[3] = {m1[2] * h1, m1[5] * h1, m1[8] * h1}; /* axes scaled by the half lengths */
  double a2[3] = {m2[2] * h2, m2[5] * h2, m2[8] * h2};
  double dif[3];
  v3_sub(dif, d->geom_xpos[g1], d->geom_xpos[g2]);
  double ma = v3_dot(a1, a1), mb = -v3_dot(a1, a2), mc = v3_dot(a2, a2);
  double u = -v3_dot(a1, dif), v = v3_dot(a2, dif);
  double det = ma * mc - mb * mb;
  double x1, x2;
  if (fabs(det) >= MINVAL) {
    x1 = (mc * u - mb * v) / det;
    x2 = (ma * v - mb * u) / det;
    if (x1 > 1) {
      x1 = 1;
      x2 = (v - mb) / mc;
    } else if (x1 < -1) {
      x1 = -1;
      x2 = (v + mb) / mc;
    }
    if (x2 > 1) {
      x2 = 1;
      x1 = (u - mb) / ma;
      x1 = x1 > 1 ? 1 : (x1 < -1 ? -1 : x1);
    } else if (x2 < -1) {
      x2 = -1;
      x1 = (u + mb) / ma;
      x1 = x1 > 1 ? 1 : (x1 < -1 ? -1 : x1);
    }
  } else { /* parallel: midpoint of the overlap of segment 2 projected on axis 1 */
    double lo = fmax(-1.0, (-v3_dot(a1, dif) - fabs(mb)) / ma);
    double hi = fmin(1.0, (-v3_dot(a1, dif) + fabs(mb)) / ma);
    x1 = lo <= hi ? 0.5 * (lo + hi) : (lo > 1 ? 1 : -1);
    x2 = (v - mb * x1) / mc;
    x2 = x2 > 1 ? 1 : (x2 < -1 ? -1 : x2);
  }
  double p1[3], p2[3];
  for (int k = 0; k < 3; ++k) {
    p1[k] = d->geom_xpos[g1][k] + a1[k] * x1;
    p2[k] = d->geom_xpos[g2][k] + a2[k] * x2;
  }
  add_sphere_sphere(m, d, g1, g2, p1, m->geom_size[g1][0], p2, m->geom_size[g2][0], margin);
}

/* Capsule (g1) vs cylinder (g2).  MuJoCo sends this pair through its general convex
 * collider (mjc_Convex: GJK / EPA, tolerance 1e-6), which is not restated; this is the
 * geometric quantity that collider converges to -- the closest points between the capsule's
 * axis segment and the solid cylinder -- by a deterministic rule:
 *   F(t) = dist^2(P(t), cylinder) is convex in the segment parameter t; g(t) = F'(t) / 2 is
 *   monotone.  Two bisections bracket the minimiser set [ta, tb] = [last t with g < -eps,
 *   first t with g > +eps] and the contact uses its midpoint (unique minimum: ta ~ tb; a
 *   segment parallel to a face or the side: the middle of the closest stretch).
 * One contact (multiccd is off by default), frame normal from the capsule to the cylinder.
 * Agreement with MuJoCo itself is bounded by its collider's tolerance: 1e-6 m in dist. */
static double capcyl_g(const double* p0, const double* dp, double t, const double* c,
                       const double* u, double R, double H, double* Pout, double* Qout) {
  double P[3], rel[3], rad[3];
  for (int k = 0; k < 3; ++k) P[k] = p0[k] + t * dp[k];
  v3_sub(rel, P, c);
  double z = v3_dot(rel, u);
  for (int k = 0; k < 3; ++k) rad[k] = rel[k] - z * u[k];
  double rho = v3_norm(rad);
  double zd = v3_dot(dp, u), rd[3];
  for (int k = 0; k < 3; ++k) rd[k] = dp[k] - zd * u[k];
  double g = 0;
  double az = fabs(z);
  if (az > H) g += (az - H) * (z > 0 ? zd : -zd);
  if (rho > R) g += (rho - R) * v3_dot(rad, rd) / rho;
  if (Pout) {
    double zc = z > H ? H : (z < -H ? -H : z);
    double sc = rho > R ? R / rho : 1.0;
    for (int k = 0; k < 3; ++k) {
      Pout[k] = P[k];
      Qout[k] = c[k] + zc * u[k] + sc * rad[k];
    }
  }
  return g;
}
/* the geometric part: capsule axis segment p0 + t dp (t in [0, 1]), radius rc, against the solid
 * cylinder (centre c, unit axis u, radius R, half height H).  Returns the distance between the
 * surfaces; pos = contact point, n = unit normal from the capsule to the cylinder. */
static double capcyl_contact(const double* p0, const double* dp, double rc, const double* cc,
                             const double* u, double R, double H, double* pos, double* n) {
  const double eps = 1e-10 * v3_dot(dp, dp);
  double lo = 0, hi = 1; /* ta: largest t with g < -eps */
  if (capcyl_g(p0, dp, 0, cc, u, R, H, NULL, NULL) >= -eps) {
    hi = 0;
  } else if (capcyl_g(p0, dp, 1, cc, u, R, H, NULL, NULL) < -eps) {
    lo = 1;
  } else {
    for (int it = 0; it < 48; ++it) {
      double mid = 0.5 * (lo + hi);
      if (capcyl_g(p0, dp, mid, cc, u, R, H, NULL, NULL) < -eps) lo = mid; else hi = mid;
    }
  }
  const double ta = lo <= 0 && hi <= 0 ? 0 : (lo >= 1 ? 1 : 0.5 * (lo + hi));
  lo = 0;
  hi = 1; /* tb: smallest t with g > +eps */
  if (capcyl_g(p0, dp, 1, cc, u, R, H, NULL, NULL) <= eps) {
    lo = 1;
  } else if (capcyl_g(p0, dp, 0, cc, u, R, H, NULL, NULL) > eps) {
    hi = 0;
  } else {
    for (int it = 0; it < 48; ++it) {
      double mid = 0.5 * (lo + hi);
      if (capcyl_g(p0, dp, mid, cc, u, R, H, NULL, NULL) > eps) hi = mid; else lo = mid;
    }
  }
  const double tb = hi <= 0 ? 0 : (lo >= 1 && hi >= 1 ? 1 : 0.5 * (lo + hi));
  double t = 0.5 * (ta + tb), P[3], Q[3];
  capcyl_g(p0, dp, t, cc, u, R, H, P, Q);
  v3_sub(n, Q, P);
  double cd = v3_norm(n);
  if (cd < 1e-12) { /* the axis itself is inside the solid: push out radially */
    double rel[3];
    v3_sub(rel, P, cc);
    double z = v3_dot(rel, u);
    for (int k = 0; k < 3; ++k) n[k] = -(rel[k] - z * u[k]);
    if (v3_norm(n) < 1e-12) {
      n[0] = 1;
      n[1] = n[2] = 0;
    }
    v3_normalize(n);
    cd = 0;
  } else {
    for (int k = 0; k < 3; ++k) n[k] /= cd;
  }
  const double dist = cd - rc;
  for (int k = 0; k < 3; ++k) pos[k] = P[k] + n[k] * (rc + 0.5 * dist);
  return dist;
}
/* test hook (tests/test_mjcpu_invariants.py): the rule above on raw geometry, cylinder axis +z */
void mjcpu_capsule_cylinder(const double* p0, const double* p1, double rc, const double* cc, double R,
                            double H, double* out7) {
  const double u[3] = {0, 0, 1};
  double dp[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  out7[0] = capcyl_contact(p0, dp, rc, cc, u, R, H, out7 + 1, out7 + 4);
}
static void add_capsule_cylinder(const mjc_model* m, mjc_data* d, int g1, int g2, double margin) {
  const double* cm1 = d->geom_xmat[g1];
  const double* cm2 = d->geom_xmat[g2];
  double a1[3] = {cm1[2], cm1[5], cm1[8]}, u[3] = {cm2[2], cm2[5], cm2[8]};
  double hl = m->geom_size[g1][1], rc = m->geom_size[g1][0];
  double R = m->geom_size[g2][0], H = m->geom_size[g2][1];
  double p0[3], dp[3];
  for (int k = 0; k < 3; ++k) {
    p0[k] = d->geom_xpos[g1][k] - hl * a1[k];
    dp[k] = 2 * hl * a1[k];
  }
  double pos[3], n[3];
  const double dist = capcyl_contact(p0, dp, rc, d->geom_xpos[g2], u, R, H, pos, n);
  if (dist > margin) return;
  if (d->ncon >= MJC_MAXCON) return;
  mjc_contact* c = &d->contact[d->ncon++];
  c->dist = dist;
  for (int k = 0; k < 3; ++k) {
    c->frame[k] = n[k];
    c->pos[k] = pos[k];
  }
  make_frame(c->frame);
  contact_params(m, c, g1, g2, margin);
}

static int geoms_can_collide(const mjc_model* m, int g1, int g2) {
  int b1 = m->geom_body[g1], b2 = m->geom_body[g2];
  if (b1 == b2) return 0;
  if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) ||
        (m->geom_contype[g2] & m->geom_conaffinity[g1]))) {
    return 0;
  }
  /* bodies without joints are welded to their parent and filtered as one body
   * (body_weldid); filterparent: no collisions between a (weld) body and its parent,
   * unless the parent is the (static) world body */
  int w1 = m->body_weldid[b1], w2 = m->body_weldid[b2];
  if (w1 == w2) return 0;
  int p1 = m->body_weldid[m->body_parent[w1]], p2 = m->body_weldid[m->body_parent[w2]];
  if (w1 != 0 && w2 != 0 && (w1 == p2 || w2 == p1)) return 0;
  return 1;
}

static void add_plane_sphere(const mjc_model* m, mjc_data* d, int g1, int g2,
                             const double* center, double radius,
                             double margin) {
  const double* pm = d->geom_xmat[g1];
  double normal[3] = {pm[2], pm[5], pm[8]}, tmp[3];
  v3_sub(tmp, center, d->geom_xpos[g1]);
  double cdist = v3_dot(tmp, normal);
  if (cdist > margin + radius) return;
  if (d->ncon >= MJC_MAXCON) return;
  mjc_contact* c = &d->contact[d->ncon++];
  c->dist = cdist - radius;
  v3_copy(c->pos, center);
  v3_addscl(c->pos, normal, -c->dist / 2 - radius);
  v3_copy(c->frame, normal);
  make_frame(c->frame);
  contact_params(m, c, g1, g2, margin);
}

static void collision(const mjc_model* m, mjc_data* d) {
  d->ncon = 0;
  if (m->disable_contact) return;
  for (int g1 = 0; g1 < m->ngeom; ++g1) {
    if (m->geom_type[g1] != MJC_GEOM_PLANE) continue;
    for (int g2 = 0; g2 < m->ngeom; ++g2) {
      if (g2 == g1 || m->geom_type[g2] == MJC_GEOM_PLANE) continue;
      if (!geoms_can_collide(m, g1, g2)) continue;
      double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
      if (m->geom_type[g2] == MJC_GEOM_SPHERE) {
        add_plane_sphere(m, d, g1, g2, d->geom_xpos[g2], m->geom_size[g2][0],
                         margin);
      } else if (m->geom_type[g2] == MJC_GEOM_CAPSULE) {
        const double* cm = d->geom_xmat[g2];
        double axis[3] = {cm[2], cm[5], cm[8]}, p[3];
        double hl = m->geom_size[g2][1];
        v3_copy(p, d->geom_xpos[g2]);
        v3_addscl(p, axis, hl);
        add_plane_sphere(m, d, g1, g2, p, m->geom_size[g2][0], margin);
        v3_copy(p, d->geom_xpos[g2]);
        v3_addscl(p, axis, -hl);
        add_plane_sphere(m, d, g1, g2, p, m->geom_size[g2][0], margin);
      }
      /* MJC_GEOM_CYLINDER vs plane (the Pusher's object on the table): the object only slides
       * in x and y, so every such contact has an identically zero Jacobian (normal = z) and no
       * effect on qacc; the rows are not generated */
    }
  }
  /* body-body pairs (hopper, humanoid self collisions): sphere / capsule primitives.
   * Pairs are visited in (geom1 < geom2) order, which is body-pair order because geoms are
   * numbered body by body; mj_collideGeoms puts the lower geom TYPE first, so a capsule
   * vs a later sphere is reported as (sphere, capsule) with the normal from the sphere. */
  for (int ga = 0; ga < m->ngeom && !m->disable_selfcollide; ++ga) {
    if (m->geom_type[ga] == MJC_GEOM_PLANE) continue;
    for (int gb = ga + 1; gb < m->ngeom; ++gb) {
      if (m->geom_type[gb] == MJC_GEOM_PLANE || !geoms_can_collide(m, ga, gb)) continue;
      int g1 = ga, g2 = gb;
      if (m->geom_type[g1] > m->geom_type[g2]) {
        g1 = gb;
        g2 = ga;
      }
      double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
      if (m->geom_type[g2] == MJC_GEOM_CYLINDER) { /* Pusher: wrist capsules vs the object */
        if (m->geom_type[g1] == MJC_GEOM_CAPSULE) add_capsule_cylinder(m, d, g1, g2, margin);
        continue; /* sphere / cylinder - cylinder: not needed by any model here */
      }
      if (m->geom_type[g1] == MJC_GEOM_CAPSULE) { /* both capsules */
        add_capsule_capsule(m, d, g1, g2, margin);
      } else if (m->geom_type[g2] == MJC_GEOM_SPHERE) { /* both spheres */
        add_sphere_sphere(m, d, g1, g2, d->geom_xpos[g1], m->geom_size[g1][0], d->geom_xpos[g2],
                          m->geom_size[g2][0], margin);
      } else { /* mjraw_SphereCapsule: closest point of the capsule axis to the centre */
        const double* cm = d->geom_xmat[g2];
        double axis[3] = {cm[2], cm[5], cm[8]}, dif[3], p[3];
        v3_sub(dif, d->geom_xpos[g1], d->geom_xpos[g2]);
        double x = v3_dot(axis, dif), hl = m->geom_size[g2][1];
        x = x > hl ? hl : (x < -hl ? -hl : x);
        v3_copy(p, d->geom_xpos[g2]);
        v3_addscl(p, axis, x);
        add_sphere_sphere(m, d, g1, g2, d->geom_xpos[g1], m->geom_size[g1][0], p,
                          m->geom_size[g2][0], margin);
      }
    }
  }
}

/* ---- M4: mj_makeConstraint + mj_makeImpedance ------------------------------------ */
static void sol_params(const mjc_model* m, const double* solref_in,
                       const double* solimp_in, double* solref, double* solimp) {
  solref[0] = solref_in[0];
  solref[1] = solref_in[1];
  /* refsafe: timeconst >= 2*timestep */
  if (solref[0] > 0) solref[0] = fmax(solref[0], 2 * m->timestep);
  solimp[0] = fmin(MAXIMP, fmax(MINIMP, solimp_in[0]));
  solimp[1] = fmin(MAXIMP, fmax(MINIMP, solimp_in[1]));
  solimp[2] = fmax(0, solimp_in[2]);
  solimp[3] = fmin(MAXIMP, fmax(MINIMP, solimp_in[3]));
  solimp[4] = fmax(1, solimp_in[4]);
}

static double impedance(const double* solimp, double pos, double margin) {
  if (solimp[0] == solimp[1] || solimp[2] <= MINVAL) {
    return 0.5 * (solimp[0] + solimp[1]);
  }
  double x = (pos - margin) / solimp[2];
  if (x < 0) x = -x;
  if (x >= 1) return solimp[1];
  if (x <= 0) return solimp[0];
  double y;
  if (solimp[4] == 1) {
    y = x;
  } else if (x <= solimp[3]) {
    double a = 1 / pow(solimp[3], solimp[4] - 1);
    y = a * pow(x, solimp[4]);
  } else {
    double b = 1 / pow(1 - solimp[3], solimp[4] - 1);
    y = 1 - b * pow(1 - x, solimp[4]);
  }
  return solimp[0] + y * (solimp[1] - solimp[0]);
}

static void add_row(const mjc_model* m, mjc_data* d, const double* J,
                    double pos, double margin, double diag,
                    const double* solref_in, const double* solimp_in) {
  if (d->nefc >= MJC_MAXEFC) return;
  int r = d->nefc++;
  double solref[2], solimp[5];
  sol_params(m, solref_in, solimp_in, solref, solimp);
  double vel = 0;
  for (int i = 0; i < m->nv; ++i) {
    d->efc_J[r][i] = J[i];
    vel += J[i] * d->qvel[i];
  }
  d->efc_pos[r] = pos;
  d->efc_margin[r] = margin;
  d->efc_vel[r] = vel;
  d->efc_diagApprox[r] = diag;
  double imp = impedance(solimp, pos, margin);
  double K, B;
  if (solref[0] > 0) { /* standard: timeconst, dampratio */
    double dmax = solimp[1];
    K = 1 / fmax(MINVAL, dmax * dmax * solref[0] * solref[0] * solref[1] * solref[1]);
    B = 2 / fmax(MINVAL, dmax * solref[0]);
  } else { /* direct */
    K = -solref[0] / fmax(MINVAL, solimp[1] * solimp[1]);
    B = -solref[1] / fmax(MINVAL, solimp[1]);
  }
  d->efc_KBI[r][0] = K;
  d->efc_KBI[r][1] = B;
  d->efc_KBI[r][2] = imp;
  d->efc_R[r] = fmax(MINVAL, (1 - imp) * diag / imp);
}

static void make_constraint(const mjc_model* m, mjc_data* d) {
  int nv = m->nv;
  d->nefc = 0;
  /* joint limits (mj_instantiateLimit) */
  if (!m->disable_limit) {
    for (int j = 0; j < m->njnt; ++j) {
      if (!m->jnt_limited[j] || m->jnt_type[j] == MJC_JNT_FREE) continue;
      double value = d->qpos[m->jnt_qposadr[j]];
      for (int side = -1; side <= 1; side += 2) {
        double dist = side * (m->jnt_range[j][(side + 1) / 2] - value);
        if (dist < m->jnt_margin[j]) {
          double J[MJC_MAXV] = {0};
          J[m->jnt_dofadr[j]] = -side;
          add_row(m, d, J, dist, m->jnt_margin[j],
                  m->dof_invweight0[m->jnt_dofadr[j]], m->jnt_solref[j],
                  m->jnt_solimp[j]);
        }
      }
    }
  }
  /* contacts, pyramidal cones (mj_instantiateContact) */
  for (int c = 0; c < d->ncon; ++c) {
    mjc_contact* con = &d->contact[c];
    con->efc_address = -1;
    if (con->dist >= con->includemargin) continue;
    con->efc_address = d->nefc;
    int b1 = m->geom_body[con->geom1], b2 = m->geom_body[con->geom2];
    double jp1[3][MJC_MAXV], jr1[3][MJC_MAXV], jp2[3][MJC_MAXV], jr2[3][MJC_MAXV];
    mjc_jac(m, d, jp1, jr1, con->pos, b1);
    mjc_jac(m, d, jp2, jr2, con->pos, b2);
    double Jc[3][MJC_MAXV]; /* rows: normal, t1, t2 */
    for (int r = 0; r < 3; ++r) {
      for (int i = 0; i < nv; ++i) {
        double s = 0;
        for (int k = 0; k < 3; ++k) {
          s += con->frame[3 * r + k] * (jp2[k][i] - jp1[k][i]);
        }
        Jc[r][i] = s;
      }
    }
    double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
    if (con->dim == 1) { /* frictionless: the normal row alone, diagApprox = tran */
      add_row(m, d, Jc[0], con->dist, con->includemargin, tran, con->solref, con->solimp);
      continue;
    }
    double mu = con->friction;
    double diag = tran + mu * mu * tran; /* mj_diagApprox, pyramidal */
    int first = d->nefc;
    for (int k = 0; k < 2; ++k) {
      for (int sgn = 1; sgn >= -1; sgn -= 2) {
        double J[MJC_MAXV];
        for (int i = 0; i < nv; ++i) J[i] = Jc[0][i] + sgn * mu * Jc[1 + k][i];
        add_row(m, d, J, con->dist, con->includemargin, diag, con->solref,
                con->solimp);
      }
    }
    /* pyramidal regulariser: all rows share Rpy = 2 mu^2 R(first) [L] */
    double Rpy = 2 * mu * mu * d->efc_R[first];
    for (int r = first; r < d->nefc; ++r) d->efc_R[r] = Rpy;
  }
  /* D and aref (mj_referenceConstraint) */
  for (int r = 0; r < d->nefc; ++r) {
    d->efc_D[r] = 1 / d->efc_R[r];
    d->efc_aref[r] = -d->efc_KBI[r][1] * d->efc_vel[r] -
                     d->efc_KBI[r][0] * d->efc_KBI[r][2] *
                         (d->efc_pos[r] - d->efc_margin[r]);
  }
}

/* ---- fwdPosition ------------------------------------------------------------------ */
void mjc_fwd_position(const mjc_model* m, mjc_data* d) {
  MJC_STAGE(1);
  kinematics(m, d);
  com_pos(m, d);
  MJC_STAGE(2);
  crb(m, d);
  MJC_STAGE(3);
  collision(m, d);
  MJC_STAGE(4);
  make_constraint(m, d);
  MJC_STAGE(0);
}

/* ---- M5: mj_fwdVelocity = comVel + passive + rne ------------------------------------- */
static void fwd_velocity(const mjc_model* m, mjc_data* d) {
  int nv = m->nv;
  /* efc_vel depends on qvel: refresh (make_constraint used current qvel) */
  memset(d->cvel[0], 0, sizeof(d->cvel[0]));
  for (int b = 1; b < m->nbody; ++b) {
    double cvel[6];
    memcpy(cvel, d->cvel[m->body_parent[b]], sizeof(cvel));
    int a = m->body_dofadr[b], n = m->body_dofnum[b];
    int j = 0;
    while (j < n) {
      int jt = m->jnt_type[m->dof_jnt[a + j]];
      if (jt == MJC_JNT_FREE) {
        for (int k = 0; k < 3; ++k) {
          memset(d->cdof_dot[a + k], 0, sizeof(double) * 6);
          for (int r = 0; r < 6; ++r) cvel[r] += d->cdof[a + k][r] * d->qvel[a + k];
        }
        for (int k = 3; k < 6; ++k) cross_motion(d->cdof_dot[a + k], cvel, d->cdof[a + k]);
        for (int k = 3; k < 6; ++k) {
          for (int r = 0; r < 6; ++r) cvel[r] += d->cdof[a + k][r] * d->qvel[a + k];
        }
        j += 6;
      } else {
        cross_motion(d->cdof_dot[a + j], cvel, d->cdof[a + j]);
        for (int r = 0; r < 6; ++r) cvel[r] += d->cdof[a + j][r] * d->qvel[a + j];
        j += 1;
      }
    }
    memcpy(d->cvel[b], cvel, sizeof(cvel));
  }
  /* mj_passive: joint springs and dampers */
  for (int i = 0; i < nv; ++i) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
  for (int j = 0; j < m->njnt; ++j) {
    if (m->jnt_type[j] == MJC_JNT_FREE) continue;
    int qa = m->jnt_qposadr[j];
    d->qfrc_passive[m->jnt_dofadr[j]] -=
        m->jnt_stiffness[j] * (d->qpos[qa] - m->qpos0[qa]);
  }
  /* mj_passive, fluid part (mj_inertiaBoxFluidModel): viscous + quadratic drag of
   * the body's equivalent inertia box in a medium of density rho / viscosity beta,
   * applied at the body COM (mj_applyFT).  Principal inertia axes are taken to be
   * the body axes (true for the swimmer's x-aligned capsules; checked). */
  if (m->opt_density > 0 || m->opt_viscosity > 0) {
    const double pi = 3.14159265358979323846;
    for (int b = 1; b < m->nbody; ++b) {
      if (m->body_mass[b] < 1e-15) continue;
      const double* I = m->body_inertia[b];
      if (fabs(I[1]) + fabs(I[2]) + fabs(I[5]) > 1e-12 * (I[0] + I[4] + I[8])) abort();
      double mass = m->body_mass[b];
      double box[3] = {sqrt(fmax(1e-15, I[4] + I[8] - I[0]) / mass * 6.0),
                       sqrt(fmax(1e-15, I[0] + I[8] - I[4]) / mass * 6.0),
                       sqrt(fmax(1e-15, I[0] + I[4] - I[8]) / mass * 6.0)};
      /* 6D velocity at the body COM in body axes: cvel is about subtree_com[root] */
      double off[3], w[3], v[3], lw[3], lv[3], tmp[3];
      v3_sub(off, d->xipos[b], d->subtree_com[m->body_rootid[b]]);
      for (int k = 0; k < 3; ++k) w[k] = d->cvel[b][k];
      v3_cross(tmp, w, off);
      for (int k = 0; k < 3; ++k) v[k] = d->cvel[b][3 + k] + tmp[k];
      const double* R = d->xmat[b]; /* columns = body axes in the world */
      for (int k = 0; k < 3; ++k) {
        lw[k] = R[k] * w[0] + R[3 + k] * w[1] + R[6 + k] * w[2];
        lv[k] = R[k] * v[0] + R[3 + k] * v[1] + R[6 + k] * v[2];
      }
      double lf[3], lt[3];
      double diam = (box[0] + box[1] + box[2]) / 3.0;
      for (int k = 0; k < 3; ++k) {
        int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        lt[k] = -pi * diam * diam * diam * m->opt_viscosity * lw[k] -
                m->opt_density * box[k] * (pow(box[k1], 4) + pow(box[k2], 4)) * fabs(lw[k]) *
                    lw[k] / 64.0;
        lf[k] = -3.0 * pi * diam * m->opt_viscosity * lv[k] -
                0.5 * m->opt_density * box[k1] * box[k2] * fabs(lv[k]) * lv[k];
      }
      double F[3], Tq[3];
      for (int k = 0; k < 3; ++k) {
        F[k] = R[3 * k] * lf[0] + R[3 * k + 1] * lf[1] + R[3 * k + 2] * lf[2];
        Tq[k] = R[3 * k] * lt[0] + R[3 * k + 1] * lt[1] + R[3 * k + 2] * lt[2];
      }
      double jp[3][MJC_MAXV], jr[3][MJC_MAXV];
      mjc_jac(m, d, jp, jr, d->xipos[b], b);
      for (int i = 0; i < nv; ++i) {
        for (int k = 0; k < 3; ++k) d->qfrc_passive[i] += jp[k][i] * F[k] + jr[k][i] * Tq[k];
      }
    }
  }
  /* mj_rne(flg_acc = 0) */
  double cacc[MJC_MAXBODY][6], cfrc[MJC_MAXBODY][6];
  memset(cacc[0], 0, sizeof(cacc[0]));
  for (int k = 0; k < 3; ++k) cacc[0][3 + k] = -m->gravity[k];
  memset(cfrc[0], 0, sizeof(cfrc[0]));
  for (int b = 1; b < m->nbody; ++b) {
    int a = m->body_dofadr[b], n = m->body_dofnum[b];
    memcpy(cacc[b], cacc[m->body_parent[b]], sizeof(cacc[b]));
    for (int j = 0; j < n; ++j) {
      for (int r = 0; r < 6; ++r) cacc[b][r] += d->cdof_dot[a + j][r] * d->qvel[a + j];
    }
    double t1[6], t2[6], t3[6];
    mul_inert_vec(t1, d->cinert[b], cacc[b]);
    mul_inert_vec(t2, d->cinert[b], d->cvel[b]);
    cross_force(t3, d->cvel[b], t2);
    for (int r = 0; r < 6; ++r) cfrc[b][r] = t1[r] + t3[r];
  }
  for (int b = m->nbody - 1; b > 0; --b) {
    int p = m->body_parent[b];
    for (int r = 0; r < 6; ++r) cfrc[p][r] += cfrc[b][r];
  }
  for (int i = 0; i < nv; ++i) d->qfrc_bias[i] = dot6(d->cdof[i], cfrc[m->dof_body[i]]);
}

/* ---- M6: mj_fwdActuation ----------------------------------------------------------- */
static void fwd_actuation(const mjc_model* m, mjc_data* d) {
  for (int i = 0; i < m->nv; ++i) d->qfrc_actuator[i] = 0;
  if (m->disable_actuation) return;
  for (int u = 0; u < m->nu; ++u) {
    double c = d->ctrl[u];
    c = fmax(m->act_ctrlrange[u][0], fmin(m->act_ctrlrange[u][1], c));
    d->qfrc_actuator[m->jnt_dofadr[m->act_jnt[u]]] += m->act_gear[u] * c;
  }
}

/* ---- M7: mj_fwdAcceleration ------------------------------------------------------- */
static void fwd_acceleration(const mjc_model* m, mjc_data* d) {
  int nv = m->nv;
  double L[MJC_MAXV * MJC_MAXV];
  for (int i = 0; i < nv; ++i) {
    d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
    d->qacc_smooth[i] = d->qfrc_smooth[i];
    for (int j = 0; j < nv; ++j) L[i * nv + j] = d->M[i][j];
  }
  chol_factor(L, nv);
  chol_solve(L, nv, d->qacc_smooth);
}

/* ---- M8: mj_fwdConstraint: Newton on the primal (pyramidal) objective ------------- */
static double constraint_cost(const mjc_data* d, int nefc, const double* jar) {
  double c = 0;
  for (int r = 0; r < nefc; ++r) {
    if (jar[r] < 0) c += 0.5 * d->efc_D[r] * jar[r] * jar[r];
  }
  return c;
}

/* mj_solPGS on the dual problem  min_f 1/2 f'(A+R)f + f'b,  f >= 0  with
 * A = J M^-1 J' (mj_projectConstraint) and b = J qacc_smooth - aref.  Unlike Newton
 * this is NOT run to convergence (humanoid.xml: iterations=50), so sweep order, the
 * warm start and the stopping rule are part of the result:
 *  - warm start: forces of qacc_warmstart through the primal law f = -D min(0, J a - aref);
 *    kept only if their dual cost is below the cost of f = 0 (which is 0);
 *  - one sweep: rows in order, f_i <- max(0, f_i - res_i / AR_ii), res_i = AR_i. f + b_i;
 *    a row update that raises the cost by more than 1e-10 is reverted;
 *  - stop when scale * (cost decrease of the sweep) < tolerance (1e-8). */
static _Thread_local double pgs_AR[MJC_MAXEFC * MJC_MAXEFC];
static _Thread_local double pgs_W[MJC_MAXEFC][MJC_MAXV];

static void fwd_constraint_pgs(const mjc_model* m, mjc_data* d) {
  int nv = m->nv, nefc = d->nefc;
  double L[MJC_MAXV * MJC_MAXV], b[MJC_MAXEFC];
  double* f = d->efc_force;
  for (int i = 0; i < nv; ++i) {
    for (int j = 0; j < nv; ++j) L[i * nv + j] = d->M[i][j];
  }
  chol_factor(L, nv);
  for (int r = 0; r < nefc; ++r) {
    memcpy(pgs_W[r], d->efc_J[r], sizeof(double) * nv);
    chol_solve(L, nv, pgs_W[r]);
  }
  for (int r = 0; r < nefc; ++r) {
    for (int c = 0; c < nefc; ++c) {
      double s = 0;
      for (int i = 0; i < nv; ++i) s += d->efc_J[r][i] * pgs_W[c][i];
      pgs_AR[r * nefc + c] = s;
    }
    pgs_AR[r * nefc + r] += d->efc_R[r];
    double s = 0, w = 0;
    for (int i = 0; i < nv; ++i) {
      s += d->efc_J[r][i] * d->qacc_smooth[i];
      w += d->efc_J[r][i] * d->qacc_warmstart[i];
    }
    b[r] = s - d->efc_aref[r];
    double jar = w - d->efc_aref[r];
    f[r] = jar < 0 ? -d->efc_D[r] * jar : 0;
  }
  double cost = 0;
  for (int r = 0; r < nefc; ++r) {
    double s = 0;
    for (int c = 0; c < nefc; ++c) s += pgs_AR[r * nefc + c] * f[c];
    cost += 0.5 * f[r] * s + f[r] * b[r];
  }
  if (cost > 0) memset(f, 0, sizeof(double) * nefc);
  double scale = 1 / (m->meaninertia * (nv > 1 ? nv : 1));
  for (int iter = 0; iter < m->iterations; ++iter) {
    double improvement = 0;
    for (int r = 0; r < nefc; ++r) {
      double res = b[r];
      for (int c = 0; c < nefc; ++c) res += pgs_AR[r * nefc + c] * f[c];
      double old = f[r], Arr = pgs_AR[r * nefc + r];
      f[r] -= res / Arr;
      if (f[r] < 0) f[r] = 0;
      double delta = f[r] - old;
      double change = 0.5 * delta * delta * Arr + delta * res;
      if (change > 1e-10) {
        f[r] = old;
        change = 0;
      }
      improvement -= change;
    }
    d->solver_iter = iter + 1;
    if (improvement * scale < 1e-8) break;
  }
  for (int r = 0; r < nefc; ++r) {
    for (int i = 0; i < nv; ++i) d->qfrc_constraint[i] += d->efc_J[r][i] * f[r];
  }
  memcpy(d->qacc, d->qfrc_constraint, sizeof(double) * nv);
  chol_solve(L, nv, d->qacc);
  for (int i = 0; i < nv; ++i) d->qacc[i] += d->qacc_smooth[i];
  if (m->warmstart_rule == 0) memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
}

static void fwd_constraint(const mjc_model* m, mjc_data* d) {
  int nv = m->nv, nefc = d->nefc;
  d->solver_iter = 0;
  memset(d->qfrc_constraint, 0, sizeof(d->qfrc_constraint));
  if (nefc == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    if (m->warmstart_rule == 0) memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof(double) * nv);
    return;
  }
  if (m->solver == MJC_SOL_PGS) {
    fwd_constraint_pgs(m, d);
    return;
  }
  /* efc_vel / aref were built in make_constraint with the current qvel */
  double qacc[MJC_MAXV], Ma[MJC_MAXV], jar[MJC_MAXEFC];
  /* warmstart: take qacc_warmstart unless qacc_smooth has lower cost */
  {
    double jar_w[MJC_MAXEFC], jar_s[MJC_MAXEFC], cw, cs, gauss = 0;
    for (int r = 0; r < nefc; ++r) {
      double a = 0, b = 0;
      for (int i = 0; i < nv; ++i) {
        a += d->efc_J[r][i] * d->qacc_warmstart[i];
        b += d->efc_J[r][i] * d->qacc_smooth[i];
      }
      jar_w[r] = a - d->efc_aref[r];
      jar_s[r] = b - d->efc_aref[r];
    }
    for (int i = 0; i < nv; ++i) {
      double s = 0;
      for (int j = 0; j < nv; ++j) {
        s += d->M[i][j] * (d->qacc_warmstart[j] - d->qacc_smooth[j]);
      }
      gauss += 0.5 * s * (d->qacc_warmstart[i] - d->qacc_smooth[i]);
    }
    cw = constraint_cost(d, nefc, jar_w) + gauss;
    cs = constraint_cost(d, nefc, jar_s);
    memcpy(qacc, cw < cs ? d->qacc_warmstart : d->qacc_smooth, sizeof(double) * nv);
  }
  double scale = 1 / (m->meaninertia * (nv > 1 ? nv : 1));
  const double tolerance = 1e-8;
  double cost = 0, grad[MJC_MAXV], search[MJC_MAXV];
  for (int iter = 0; iter < 100; ++iter) {
    /* PrimalUpdateConstraint */
    for (int i = 0; i < nv; ++i) {
      double s = 0;
      for (int j = 0; j < nv; ++j) s += d->M[i][j] * qacc[j];
      Ma[i] = s;
    }
    for (int r = 0; r < nefc; ++r) {
      double s = 0;
      for (int i = 0; i < nv; ++i) s += d->efc_J[r][i] * qacc[i];
      jar[r] = s - d->efc_aref[r];
    }
    double gauss = 0;
    for (int i = 0; i < nv; ++i) {
      gauss += 0.5 * (Ma[i] - d->qfrc_smooth[i]) * (qacc[i] - d->qacc_smooth[i]);
    }
    cost = gauss + constraint_cost(d, nefc, jar);
    (void)cost;
    /* gradient and Hessian */
    double H[MJC_MAXV * MJC_MAXV];
    for (int i = 0; i < nv; ++i) {
      grad[i] = Ma[i] - d->qfrc_smooth[i];
      for (int j = 0; j < nv; ++j) H[i * nv + j] = d->M[i][j];
    }
    for (int r = 0; r < nefc; ++r) {
      if (jar[r] >= 0) continue;
      double f = -d->efc_D[r] * jar[r];
      for (int i = 0; i < nv; ++i) {
        grad[i] -= d->efc_J[r][i] * f;
        double dj = d->efc_D[r] * d->efc_J[r][i];
        if (dj == 0) continue;
        for (int j = 0; j < nv; ++j) H[i * nv + j] += dj * d->efc_J[r][j];
      }
    }
    double gnorm = 0;
    for (int i = 0; i < nv; ++i) gnorm += grad[i] * grad[i];
    gnorm = sqrt(gnorm);
    /* tighter than MuJoCo's 1e-8 so the restatement sits at the unique
     * minimiser; MuJoCo's own iterate is within its tolerance of it */
    if (scale * gnorm < tolerance * 1e-4) break;
    chol_factor(H, nv);
    for (int i = 0; i < nv; ++i) search[i] = -grad[i];
    chol_solve(H, nv, search);
    /* exact line search on the piecewise-quadratic phi(alpha) */
    double Mv[MJC_MAXV], jv[MJC_MAXEFC];
    for (int i = 0; i < nv; ++i) {
      double s = 0;
      for (int j = 0; j < nv; ++j) s += d->M[i][j] * search[j];
      Mv[i] = s;
    }
    for (int r = 0; r < nefc; ++r) {
      double s = 0;
      for (int i = 0; i < nv; ++i) s += d->efc_J[r][i] * search[i];
      jv[r] = s;
    }
    double g1 = 0, g2 = 0; /* gauss part: phi' = g1 + alpha*g2 */
    for (int i = 0; i < nv; ++i) {
      g1 += search[i] * (Ma[i] - d->qfrc_smooth[i]);
      g2 += search[i] * Mv[i];
    }
    double alpha = 0, lo = 0, hi = -1;
    for (int ls = 0; ls < 100; ++ls) {
      double d1 = g1 + alpha * g2, d2 = g2;
      for (int r = 0; r < nefc; ++r) {
        double x = jar[r] + alpha * jv[r];
        if (x < 0) {
          d1 += d->efc_D[r] * x * jv[r];
          d2 += d->efc_D[r] * jv[r] * jv[r];
        }
      }
      if (d1 < 0) lo = alpha; else hi = alpha;
      if (fabs(d1) < 1e-14 * (fabs(g1) + 1e-300)) break;
      double next = alpha - d1 / d2;
      if (hi >= 0 && (next <= lo || next >= hi)) next = 0.5 * (lo + hi);
      if (next < 0) next = 0;
      if (next == alpha) break;
      alpha = next;
    }
    if (alpha == 0) break;
    for (int i = 0; i < nv; ++i) qacc[i] += alpha * search[i];
    d->solver_iter = iter + 1;
  }
  memcpy(d->qacc, qacc, sizeof(double) * nv);
  for (int r = 0; r < nefc; ++r) {
    double s = 0;
    for (int i = 0; i < nv; ++i) s += d->efc_J[r][i] * qacc[i];
    double x = s - d->efc_aref[r];
    d->efc_force[r] = x < 0 ? -d->efc_D[r] * x : 0;
    for (int i = 0; i < nv; ++i) d->qfrc_constraint[i] += d->efc_J[r][i] * d->efc_force[r];
  }
  if (m->warmstart_rule == 0) memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
}

void mjc_forward(const mjc_model* m, mjc_data* d) { /* mj_forward */
  mjc_fwd_position(m, d);
  MJC_STAGE(5);
  fwd_velocity(m, d);
  MJC_STAGE(6);
  fwd_actuation(m, d);
  MJC_STAGE(7);
  fwd_acceleration(m, d);
  MJC_STAGE(8);
  fwd_constraint(m, d);
  MJC_STAGE(0);
}

/* mj_integratePos */
static void integrate_pos(const mjc_model* m, double* qpos, const double* qvel,
                          double h) {
  for (int j = 0; j < m->njnt; ++j) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == MJC_JNT_FREE) {
      for (int k = 0; k < 3; ++k) qpos[qa + k] += h * qvel[da + k];
      double w[3] = {qvel[da + 3], qvel[da + 4], qvel[da + 5]};
      double ang = v3_norm(w) * h;
      if (ang > 0) {
        double ax[3], dq[4], nq[4];
        v3_copy(ax, w);
        v3_normalize(ax);
        quat_axisangle(dq, ax, ang);
        quat_mul(nq, qpos + qa + 3, dq);
        quat_normalize(nq);
        for (int k = 0; k < 4; ++k) qpos[qa + 3 + k] = nq[k];
      }
    } else {
      qpos[qa] += h * qvel[da];
    }
  }
}

/* ---- M9: mj_Euler (implicit joint damping) / mj_RungeKutta(4) ---------------------- */
static void euler(const mjc_model* m, mjc_data* d) {
  int nv = m->nv;
  double h = m->timestep, qacc[MJC_MAXV];
  int damped = 0;
  for (int i = 0; i < nv; ++i) damped |= m->dof_damping[i] > 0;
  if (damped) { /* eulerdamp: (M + h*diag(damping)) qacc = qfrc_smooth + qfrc_constraint */
    double L[MJC_MAXV * MJC_MAXV];
    for (int i = 0; i < nv; ++i) {
      for (int j = 0; j < nv; ++j) L[i * nv + j] = d->M[i][j];
      L[i * nv + i] += h * m->dof_damping[i];
      qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    }
    chol_factor(L, nv);
    chol_solve(L, nv, qacc);
  } else {
    memcpy(qacc, d->qacc, sizeof(double) * nv);
  }
  for (int i = 0; i < nv; ++i) d->qvel[i] += h * qacc[i];
  integrate_pos(m, d->qpos, d->qvel, h);
  d->time += h;
}

static void rk4(const mjc_model* m, mjc_data* d) {
  static const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1}};
  static const double B[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
  static const double T[3] = {0.5, 0.5, 1.0};
  int nv = m->nv, nq = m->nq;
  double h = m->timestep, time = d->time;
  double X0q[MJC_MAXQ], Xv[4][MJC_MAXV], F[4][MJC_MAXV];
  memcpy(X0q, d->qpos, sizeof(double) * nq);
  memcpy(Xv[0], d->qvel, sizeof(double) * nv);
  memcpy(F[0], d->qacc, sizeof(double) * nv);
  for (int i = 1; i < 4; ++i) {
    double dq[MJC_MAXV] = {0}, dv[MJC_MAXV] = {0};
    for (int j = 0; j < i; ++j) {
      for (int k = 0; k < nv; ++k) {
        dq[k] += A[i - 1][j] * Xv[j][k];
        dv[k] += A[i - 1][j] * F[j][k];
      }
    }
    memcpy(d->qpos, X0q, sizeof(double) * nq);
    integrate_pos(m, d->qpos, dq, h);
    for (int k = 0; k < nv; ++k) d->qvel[k] = Xv[0][k] + h * dv[k];
    memcpy(Xv[i], d->qvel, sizeof(double) * nv);
    d->time = time + T[i - 1] * h;
    mjc_forward(m, d);
    MJC_STAGE(9);
    memcpy(F[i], d->qacc, sizeof(double) * nv);
  }
  double dq[MJC_MAXV] = {0}, dv[MJC_MAXV] = {0};
  for (int j = 0; j < 4; ++j) {
    for (int k = 0; k < nv; ++k) {
      dq[k] += B[j] * Xv[j][k];
      dv[k] += B[j] * F[j][k];
    }
  }
  memcpy(d->qpos, X0q, sizeof(double) * nq);
  for (int k = 0; k < nv; ++k) d->qvel[k] = Xv[0][k] + h * dv[k];
  integrate_pos(m, d->qpos, dq, h);
  d->time = time + h;
}

/* mj_rnePostConstraint, the cfrc_ext part: external (contact) force on every
 * body as a spatial force [torque; force] about the subtree COM of its kinematic
 * tree's root, from the contacts and efc_force of the LAST forward evaluation.
 * Pyramidal decode as mju_decodePyramid: normal = sum of the 2(dim-1) edge
 * forces, tangent_k = (f[2k] - f[2k+1]) * mu_k. */
void mjc_rne_post_constraint(const mjc_model* m, mjc_data* d) {
  MJC_STAGE(10);
  for (int b = 0; b < m->nbody; ++b) {
    for (int k = 0; k < 6; ++k) d->cfrc_ext[b][k] = 0;
  }
  for (int c = 0; c < d->ncon; ++c) {
    const mjc_contact* con = &d->contact[c];
    if (con->efc_address < 0) continue;
    const double* f = d->efc_force + con->efc_address;
    double lf[3] = {f[0], 0, 0};
    if (con->dim != 1) {
      lf[0] = f[0] + f[1] + f[2] + f[3];
      lf[1] = (f[0] - f[1]) * con->friction;
      lf[2] = (f[2] - f[3]) * con->friction;
    }
    double F[3]; /* world frame: frame^T lf (frame rows: normal, t1, t2) */
    for (int k = 0; k < 3; ++k) {
      F[k] = con->frame[k] * lf[0] + con->frame[3 + k] * lf[1] + con->frame[6 + k] * lf[2];
    }
    for (int side = 0; side < 2; ++side) {
      int b = m->geom_body[side == 0 ? con->geom1 : con->geom2];
      double sgn = side == 0 ? -1.0 : 1.0; /* the force acts on body 2; body 1 gets -F */
      double off[3], tq[3];
      v3_sub(off, con->pos, d->subtree_com[m->body_rootid[b]]);
      v3_cross(tq, off, F);
      for (int k = 0; k < 3; ++k) {
        d->cfrc_ext[b][k] += sgn * tq[k];
        d->cfrc_ext[b][3 + k] += sgn * F[k];
      }
    }
  }
  MJC_STAGE(0);
}

void mjc_step(const mjc_model* m, mjc_data* d) { /* mj_step */
  mjc_forward(m, d);
  MJC_STAGE(9);
  if (m->integrator == MJC_INT_RK4) {
    rk4(m, d); /* its three further forward evaluations count under their own stages */
  } else {
    euler(m, d);
  }
  MJC_STAGE(0);
  /* warmstart_rule 1: d->qacc is still that of the last forward evaluation */
  if (m->warmstart_rule == 1) memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * m->nv);
}

double mjc_energy_kinetic(const mjc_model* m, mjc_data* d) {
  double e = 0;
  for (int i = 0; i < m->nv; ++i) {
    for (int j = 0; j < m->nv; ++j) e += 0.5 * d->qvel[i] * d->M[i][j] * d->qvel[j];
  }
  return e;
}

double mjc_energy_potential(const mjc_model* m, mjc_data* d) {
  double e = 0;
  for (int b = 1; b < m->nbody; ++b) {
    e -= m->body_mass[b] * v3_dot(m->gravity, d->xipos[b]);
  }
  for (int j = 0; j < m->njnt; ++j) {
    if (m->jnt_type[j] == MJC_JNT_FREE) continue;
    int qa = m->jnt_qposadr[j];
    double dq = d->qpos[qa] - m->qpos0[qa];
    e += 0.5 * m->jnt_stiffness[j] * dq * dq;
  }
  return e;
}
