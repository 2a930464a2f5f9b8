/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See mjcpu.h.
 *
 * mjcpu_model_dump: the hand-transcribed gym models of models.c (after mjc_build_* + mjc_compile) as one JSON
 * object, field by field, so that tests/test_models_vs_xml.py can hold EVERY raw attribute against the MJCF the
 * reference loads (third_party/mujoco_gym_xml_patches/<stem>_envpool.xml, envpool/mujoco/gym/mujoco_env.h:50-58)
 * and the compiled masses / inertias against a third, XML-driven model compiler (tools/mjcf_subset.py). */
#include <stdio.h>
#include <string.h>

#include "mjcpu.h"

typedef struct {
  char* p;
  int cap, len;
} sbuf;

static void put(sbuf* s, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static void put(sbuf* s, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  int room = s->cap - s->len;
  int n = vsnprintf(room > 0 ? s->p + s->len : NULL, room > 0 ? (size_t)room : 0, fmt, ap);
  va_end(ap);
  s->len += n; /* keeps counting past the end: the caller sees the size it needs */
}
static void arr_d(sbuf* s, const char* name, const double* v, int n, int m) {
  put(s, "\"%s\": [", name);
  for (int i = 0; i < n; ++i) {
    if (m > 1) put(s, "%s[", i ? ", " : "");
    for (int j = 0; j < m; ++j) put(s, "%s%.17g", (j || (m == 1 && i)) ? ", " : "", v[i * m + j]);
    if (m > 1) put(s, "]");
  }
  put(s, "],\n");
}
static void arr_i(sbuf* s, const char* name, const int* v, int n) {
  put(s, "\"%s\": [", name);
  for (int i = 0; i < n; ++i) put(s, "%s%d", i ? ", " : "", v[i]);
  put(s, "],\n");
}

int mjcpu_build_by_xml_stem(const char* stem, mjc_model* m) {
  if (!strcmp(stem, "half_cheetah")) mjc_build_half_cheetah(m);
  else if (!strcmp(stem, "ant")) mjc_build_ant(m);
  else if (!strcmp(stem, "walker2d")) mjc_build_walker2d(m, 0);
  else if (!strcmp(stem, "walker2d_v5")) mjc_build_walker2d(m, 1);
  else if (!strcmp(stem, "hopper")) mjc_build_hopper(m);
  else if (!strcmp(stem, "swimmer")) mjc_build_swimmer(m);
  else if (!strcmp(stem, "reacher")) mjc_build_reacher(m);
  else if (!strcmp(stem, "pusher")) mjc_build_pusher(m, 0);
  else if (!strcmp(stem, "pusher_v5")) mjc_build_pusher(m, 1);
  else if (!strcmp(stem, "inverted_pendulum")) mjc_build_inverted_pendulum(m);
  else if (!strcmp(stem, "inverted_double_pendulum")) mjc_build_inverted_double_pendulum(m);
  else if (!strcmp(stem, "humanoid")) mjc_build_humanoid(m, 0);
  else if (!strcmp(stem, "humanoidstandup")) mjc_build_humanoid(m, 1);
  else return -1;
  return 0;
}

/* returns the number of bytes the JSON needs (excluding the terminator); writes at most cap */
int mjcpu_model_dump(const char* stem, char* buf, int cap) {
  static mjc_model m; /* tests are single-threaded */
  if (mjcpu_build_by_xml_stem(stem, &m)) return -1;
  sbuf s = {buf, cap, 0};
  put(&s, "{\n\"nq\": %d, \"nv\": %d, \"nu\": %d, \"nbody\": %d, \"njnt\": %d, \"ngeom\": %d,\n", m.nq, m.nv, m.nu,
      m.nbody, m.njnt, m.ngeom);
  put(&s, "\"timestep\": %.17g, \"integrator\": %d, \"solver\": %d, \"iterations\": %d,\n", m.timestep, m.integrator,
      m.solver, m.iterations);
  put(&s, "\"opt_density\": %.17g, \"opt_viscosity\": %.17g, \"settotalmass\": %.17g, \"meaninertia\": %.17g,\n",
      m.opt_density, m.opt_viscosity, m.settotalmass, m.meaninertia);
  arr_d(&s, "gravity", m.gravity, 3, 1);
  arr_i(&s, "body_parent", m.body_parent, m.nbody);
  arr_i(&s, "body_jntadr", m.body_jntadr, m.nbody);
  arr_i(&s, "body_jntnum", m.body_jntnum, m.nbody);
  arr_i(&s, "body_dofadr", m.body_dofadr, m.nbody);
  arr_i(&s, "body_dofnum", m.body_dofnum, m.nbody);
  arr_d(&s, "body_pos", &m.body_pos[0][0], m.nbody, 3);
  arr_d(&s, "body_quat", &m.body_quat[0][0], m.nbody, 4);
  arr_d(&s, "body_ipos", &m.body_ipos[0][0], m.nbody, 3);
  arr_d(&s, "body_mass", m.body_mass, m.nbody, 1);
  arr_d(&s, "body_inertia", &m.body_inertia[0][0], m.nbody, 9);
  arr_d(&s, "body_invweight0", &m.body_invweight0[0][0], m.nbody, 2);
  arr_i(&s, "jnt_type", m.jnt_type, m.njnt);
  arr_i(&s, "jnt_body", m.jnt_body, m.njnt);
  arr_i(&s, "jnt_qposadr", m.jnt_qposadr, m.njnt);
  arr_i(&s, "jnt_dofadr", m.jnt_dofadr, m.njnt);
  arr_i(&s, "jnt_limited", m.jnt_limited, m.njnt);
  arr_d(&s, "jnt_pos", &m.jnt_pos[0][0], m.njnt, 3);
  arr_d(&s, "jnt_axis", &m.jnt_axis[0][0], m.njnt, 3);
  arr_d(&s, "jnt_range", &m.jnt_range[0][0], m.njnt, 2);
  arr_d(&s, "jnt_stiffness", m.jnt_stiffness, m.njnt, 1);
  arr_d(&s, "jnt_margin", m.jnt_margin, m.njnt, 1);
  arr_d(&s, "jnt_ref", m.jnt_ref, m.njnt, 1);
  arr_d(&s, "jnt_solref", &m.jnt_solref[0][0], m.njnt, 2);
  arr_d(&s, "jnt_solimp", &m.jnt_solimp[0][0], m.njnt, 5);
  arr_i(&s, "dof_jnt", m.dof_jnt, m.nv);
  arr_d(&s, "dof_armature", m.dof_armature, m.nv, 1);
  arr_d(&s, "dof_damping", m.dof_damping, m.nv, 1);
  arr_d(&s, "dof_invweight0", m.dof_invweight0, m.nv, 1);
  arr_i(&s, "geom_type", m.geom_type, m.ngeom);
  arr_i(&s, "geom_body", m.geom_body, m.ngeom);
  arr_i(&s, "geom_contype", m.geom_contype, m.ngeom);
  arr_i(&s, "geom_conaffinity", m.geom_conaffinity, m.ngeom);
  arr_i(&s, "geom_condim", m.geom_condim, m.ngeom);
  arr_d(&s, "geom_size", &m.geom_size[0][0], m.ngeom, 3);
  arr_d(&s, "geom_pos", &m.geom_pos[0][0], m.ngeom, 3);
  arr_d(&s, "geom_quat", &m.geom_quat[0][0], m.ngeom, 4);
  arr_d(&s, "geom_friction", &m.geom_friction[0][0], m.ngeom, 3);
  arr_d(&s, "geom_margin", m.geom_margin, m.ngeom, 1);
  arr_d(&s, "geom_density", m.geom_density, m.ngeom, 1);
  arr_d(&s, "geom_solref", &m.geom_solref[0][0], m.ngeom, 2);
  arr_d(&s, "geom_solimp", &m.geom_solimp[0][0], m.ngeom, 5);
  arr_i(&s, "act_jnt", m.act_jnt, m.nu);
  arr_d(&s, "act_gear", m.act_gear, m.nu, 1);
  arr_d(&s, "act_ctrlrange", &m.act_ctrlrange[0][0], m.nu, 2);
  put(&s, "\"qpos0\": [");
  for (int i = 0; i < m.nq; ++i) put(&s, "%s%.17g", i ? ", " : "", m.qpos0[i]);
  put(&s, "]\n}\n");
  if (cap > 0) buf[s.len < cap ? s.len : cap - 1] = 0;
  return s.len;
}
