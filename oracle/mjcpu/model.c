/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See mjcpu.h (PARITY UNPINNED).
 *
 * Mini "model compiler": turns a hand-transcribed MJCF subset (bodies, joints,
 * capsule/sphere/plane geoms, motors) into the constants MuJoCo's compiler
 * would produce: geom frames from fromto/axisangle,
 * inertiafromgeom (analytic sphere / capsule), settotalmass, qpos0, dof tree,
 * dof_invweight0 / body_invweight0 / meaninertia at qpos0.
 */
#include <math.h>
#include <string.h>

#include "mjcpu.h"
#include "mjmath.h"

void mjc_model_init(mjc_model* m) {
  memset(m, 0, sizeof(*m));
  m->timestep = 0.002;
  m->gravity[2] = -9.81;
  m->integrator = MJC_INT_EULER;
  m->solver = MJC_SOL_NEWTON;
  m->iterations = 100;
  /* world body */
  m->nbody = 1;
  m->body_parent[0] = 0;
  m->body_quat[0][0] = 1;
  m->body_jntadr[0] = -1;
  m->body_dofadr[0] = -1;
}

int mjc_add_body(mjc_model* m, int parent, const double pos[3]) {
  int b = m->nbody++;
  m->body_parent[b] = parent;
  v3_copy(m->body_pos[b], pos);
  m->body_quat[b][0] = 1;
  m->body_jntadr[b] = -1;
  m->body_dofadr[b] = -1;
  return b;
}

int mjc_add_joint(mjc_model* m, int body, int type, const double pos[3],
                  const double axis[3], int limited, double lo, double hi,
                  double stiffness, double damping, double armature) {
  int j = m->njnt++;
  m->jnt_type[j] = type;
  m->jnt_body[j] = body;
  v3_copy(m->jnt_pos[j], pos);
  v3_copy(m->jnt_axis[j], axis);
  if (type != MJC_JNT_FREE) v3_normalize(m->jnt_axis[j]);
  m->jnt_limited[j] = limited;
  m->jnt_range[j][0] = lo;
  m->jnt_range[j][1] = hi;
  m->jnt_stiffness[j] = stiffness;
  m->jnt_margin[j] = 0;
  m->jnt_ref[j] = 0;
  /* MuJoCo defaults: solref 0.02 1, solimp 0.9 0.95 0.001 0.5 2 */
  m->jnt_solref[j][0] = 0.02;
  m->jnt_solref[j][1] = 1;
  m->jnt_solimp[j][0] = 0.9;
  m->jnt_solimp[j][1] = 0.95;
  m->jnt_solimp[j][2] = 0.001;
  m->jnt_solimp[j][3] = 0.5;
  m->jnt_solimp[j][4] = 2;
  m->jnt_qposadr[j] = m->nq;
  m->jnt_dofadr[j] = m->nv;
  if (m->body_jntadr[body] < 0) {
    m->body_jntadr[body] = j;
    m->body_dofadr[body] = m->nv;
  }
  m->body_jntnum[body]++;
  int ndof = type == MJC_JNT_FREE ? 6 : 1;
  int nqj = type == MJC_JNT_FREE ? 7 : 1;
  for (int k = 0; k < ndof; ++k) {
    int d = m->nv + k;
    m->dof_body[d] = body;
    m->dof_jnt[d] = j;
    m->dof_armature[d] = armature;
    m->dof_damping[d] = damping;
  }
  m->nv += ndof;
  m->nq += nqj;
  m->body_dofnum[body] += ndof;
  return j;
}

int mjc_add_geom(mjc_model* m, int body, int type, const double size[3],
                 const double pos[3], const double quat[4]) {
  int g = m->ngeom++;
  m->geom_type[g] = type;
  m->geom_body[g] = body;
  v3_copy(m->geom_size[g], size);
  v3_copy(m->geom_pos[g], pos);
  for (int i = 0; i < 4; ++i) m->geom_quat[g][i] = quat[i];
  /* MuJoCo defaults */
  m->geom_contype[g] = 1;
  m->geom_conaffinity[g] = 1;
  m->geom_condim[g] = 3;
  m->geom_friction[g][0] = 1;
  m->geom_friction[g][1] = 0.005;
  m->geom_friction[g][2] = 0.0001;
  m->geom_margin[g] = 0;
  m->geom_density[g] = 1000;
  m->geom_solref[g][0] = 0.02;
  m->geom_solref[g][1] = 1;
  m->geom_solimp[g][0] = 0.9;
  m->geom_solimp[g][1] = 0.95;
  m->geom_solimp[g][2] = 0.001;
  m->geom_solimp[g][3] = 0.5;
  m->geom_solimp[g][4] = 2;
  return g;
}

/* quaternion rotating +z onto `vec` (MuJoCo's mjuu_z2quat) */
static void z2quat(double quat[4], const double vec[3]) {
  double z[3] = {0, 0, 1}, axis[3], v[3];
  v3_copy(v, vec);
  v3_normalize(v);
  v3_cross(axis, z, v);
  double s = v3_norm(axis);
  if (s < 1e-10) {
    axis[0] = 1;
    axis[1] = axis[2] = 0;
  } else {
    v3_scale(axis, axis, 1.0 / s);
  }
  double ang = atan2(s, v[2]);
  quat[0] = cos(ang / 2);
  quat[1] = axis[0] * sin(ang / 2);
  quat[2] = axis[1] * sin(ang / 2);
  quat[3] = axis[2] * sin(ang / 2);
}

int mjc_add_capsule_fromto(mjc_model* m, int body, const double from[3],
                           const double to[3], double radius) {
  double d[3], pos[3], quat[4], size[3];
  v3_sub(d, to, from);
  double len = v3_norm(d);
  for (int i = 0; i < 3; ++i) pos[i] = 0.5 * (from[i] + to[i]);
  z2quat(quat, d);
  size[0] = radius;
  size[1] = len / 2;
  size[2] = 0;
  return mjc_add_geom(m, body, MJC_GEOM_CAPSULE, size, pos, quat);
}

int mjc_add_capsule_axisangle(mjc_model* m, int body, const double pos[3],
                              const double axis[3], double angle, double radius,
                              double halflen) {
  double a[3], quat[4], size[3] = {radius, halflen, 0};
  v3_copy(a, axis);
  v3_normalize(a);
  quat[0] = cos(angle / 2);
  quat[1] = a[0] * sin(angle / 2);
  quat[2] = a[1] * sin(angle / 2);
  quat[3] = a[2] * sin(angle / 2);
  return mjc_add_geom(m, body, MJC_GEOM_CAPSULE, size, pos, quat);
}

int mjc_add_motor(mjc_model* m, int jnt, double gear) {
  int u = m->nu++;
  m->act_jnt[u] = jnt;
  m->act_gear[u] = gear;
  m->act_ctrlrange[u][0] = -1;
  m->act_ctrlrange[u][1] = 1;
  return u;
}

/* mass and diagonal inertia of a geom in its own frame (A.3) */
static void geom_inertia(const mjc_model* m, int g, double* mass,
                         double inertia[3]) {
  const double pi = 3.14159265358979323846;
  double r = m->geom_size[g][0];
  *mass = 0;
  inertia[0] = inertia[1] = inertia[2] = 0;
  if (m->geom_type[g] == MJC_GEOM_SPHERE) {
    *mass = m->geom_density[g] * 4.0 / 3.0 * pi * r * r * r;
    inertia[0] = inertia[1] = inertia[2] = 2.0 * (*mass) * r * r / 5.0;
  } else if (m->geom_type[g] == MJC_GEOM_CAPSULE) {
    double height = 2 * m->geom_size[g][1];
    double vol = pi * (r * r * height + 4.0 * r * r * r / 3.0);
    *mass = m->geom_density[g] * vol;
    double sphere_mass = (*mass) * 4 * r / (4 * r + 3 * height);
    double cyl_mass = (*mass) - sphere_mass;
    inertia[0] = inertia[1] = cyl_mass * (3 * r * r + height * height) / 12;
    inertia[2] = cyl_mass * r * r / 2;
    double sph = 2 * sphere_mass * r * r / 5;
    double shift = sphere_mass * height * (3 * r + 2 * height) / 8;
    inertia[0] += sph + shift;
    inertia[1] += sph + shift;
    inertia[2] += sph;
  } else if (m->geom_type[g] == MJC_GEOM_CYLINDER) { /* size = radius, half height */
    double height = 2 * m->geom_size[g][1];
    *mass = m->geom_density[g] * pi * r * r * height;
    inertia[0] = inertia[1] = (*mass) * (3 * r * r + height * height) / 12;
    inertia[2] = (*mass) * r * r / 2;
  }
}

void mjc_compile(mjc_model* m) {
  /* inertiafromgeom */
  double total = 0;
  for (int b = 0; b < m->nbody; ++b) {
    double mass = 0, com[3] = {0, 0, 0};
    for (int g = 0; g < m->ngeom; ++g) {
      if (m->geom_body[g] != b) continue;
      double gm, gi[3];
      geom_inertia(m, g, &gm, gi);
      mass += gm;
      for (int i = 0; i < 3; ++i) com[i] += gm * m->geom_pos[g][i];
    }
    m->body_mass[b] = mass;
    if (mass > 0) v3_scale(com, com, 1.0 / mass);
    v3_copy(m->body_ipos[b], com);
    double I[9] = {0};
    for (int g = 0; g < m->ngeom; ++g) {
      if (m->geom_body[g] != b) continue;
      double gm, gi[3], R[9], RI[9], Rt[9], tmp[9], d[3];
      geom_inertia(m, g, &gm, gi);
      if (gm <= 0) continue;
      quat2mat(R, m->geom_quat[g]);
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) RI[3 * r + c] = R[3 * r + c] * gi[c];
      }
      m3_transpose(Rt, R);
      m3_mul(tmp, RI, Rt);
      v3_sub(d, m->geom_pos[g], com);
      double d2 = v3_dot(d, d);
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
          I[3 * r + c] += tmp[3 * r + c] +
                          gm * ((r == c ? d2 : 0.0) - d[r] * d[c]);
        }
      }
    }
    memcpy(m->body_inertia[b], I, sizeof(I));
    total += mass;
  }
  if (m->settotalmass > 0 && total > 0) {
    double s = m->settotalmass / total;
    for (int b = 0; b < m->nbody; ++b) {
      m->body_mass[b] *= s;
      for (int i = 0; i < 9; ++i) m->body_inertia[b][i] *= s;
    }
  }
  /* tree bookkeeping */
  for (int b = 0; b < m->nbody; ++b) {
    int r = b;
    while (r > 0 && m->body_parent[r] > 0) r = m->body_parent[r];
    m->body_rootid[b] = r;
    /* body_weldid: the nearest ancestor-or-self that moves relative to its parent */
    int w = b;
    while (w > 0 && m->body_dofnum[w] == 0) w = m->body_parent[w];
    m->body_weldid[b] = w;
  }
  for (int d = 0; d < m->nv; ++d) {
    int b = m->dof_body[d];
    if (d > m->body_dofadr[b]) {
      m->dof_parent[d] = d - 1;
    } else {
      int p = m->body_parent[b];
      while (p > 0 && m->body_dofnum[p] == 0) p = m->body_parent[p];
      m->dof_parent[d] = p > 0 ? m->body_dofadr[p] + m->body_dofnum[p] - 1 : -1;
    }
  }
  /* qpos0 */
  for (int j = 0; j < m->njnt; ++j) {
    int a = m->jnt_qposadr[j];
    if (m->jnt_type[j] == MJC_JNT_FREE) {
      int b = m->jnt_body[j];
      for (int i = 0; i < 3; ++i) m->qpos0[a + i] = m->body_pos[b][i];
      for (int i = 0; i < 4; ++i) m->qpos0[a + 3 + i] = m->body_quat[b][i];
    } else {
      m->qpos0[a] = m->jnt_ref[j];
    }
  }
  /* constants at qpos0 (MuJoCo mj_setConst / set0) */
  static mjc_data d;
  mjc_reset_data(m, &d);
  mjc_fwd_position(m, &d);
  int nv = m->nv;
  double Minv[MJC_MAXV][MJC_MAXV];
  {
    double L[MJC_MAXV * MJC_MAXV];
    for (int i = 0; i < nv; ++i) {
      for (int j = 0; j < nv; ++j) L[i * nv + j] = d.M[i][j];
    }
    chol_factor(L, nv);
    for (int c = 0; c < nv; ++c) {
      double e[MJC_MAXV] = {0};
      e[c] = 1;
      chol_solve(L, nv, e);
      for (int r = 0; r < nv; ++r) Minv[r][c] = e[r];
    }
  }
  double tr = 0;
  for (int i = 0; i < nv; ++i) tr += d.M[i][i];
  m->meaninertia = nv > 0 ? tr / nv : 1;
  for (int j = 0; j < m->njnt; ++j) {
    int a = m->jnt_dofadr[j];
    if (m->jnt_type[j] == MJC_JNT_FREE) {
      double t = (Minv[a][a] + Minv[a + 1][a + 1] + Minv[a + 2][a + 2]) / 3;
      double r = (Minv[a + 3][a + 3] + Minv[a + 4][a + 4] + Minv[a + 5][a + 5]) / 3;
      for (int k = 0; k < 3; ++k) {
        m->dof_invweight0[a + k] = t;
        m->dof_invweight0[a + 3 + k] = r;
      }
    } else {
      m->dof_invweight0[a] = Minv[a][a];
    }
  }
  for (int b = 1; b < m->nbody; ++b) {
    double jacp[3][MJC_MAXV], jacr[3][MJC_MAXV];
    mjc_jac(m, &d, jacp, jacr, d.xipos[b], b);
    double tp = 0, trr = 0;
    for (int r = 0; r < 3; ++r) {
      for (int i = 0; i < nv; ++i) {
        for (int j = 0; j < nv; ++j) {
          tp += jacp[r][i] * Minv[i][j] * jacp[r][j];
          trr += jacr[r][i] * Minv[i][j] * jacr[r][j];
        }
      }
    }
    m->body_invweight0[b][0] = tp / 3;
    m->body_invweight0[b][1] = trr / 3;
  }
}
