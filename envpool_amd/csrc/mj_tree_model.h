// Host-side model compiler for mj_tree.hip.h: turns a hand-transcribed MJCF subset (free
// root body, hinge joints, sphere / capsule geoms, joint motors, a floor plane) into the
// constants MuJoCo's compiler would produce -- inertiafromgeom, body frames, the dof tree,
// qpos0, and dof_invweight0 / body_invweight0 / meaninertia evaluated at qpos0 -- plus the
// static constraint-group lists the kernel uses (limited joints, floor spheres, geom pairs
// that pass MuJoCo's contype/conaffinity, same-weld-body and parent-child filters).
// Runs at build time inside gen_mj_consts (-> build/mj_humanoid_consts.inc) and in the CPU
// test harness.  Models: third_party/mujoco_gym_xml_patches/humanoid_envpool.xml and
// humanoidstandup_envpool.xml of the reference tree (line numbers cited at each use).
#ifndef ENVPOOL_AMD_CSRC_MJ_TREE_MODEL_H_
#define ENVPOOL_AMD_CSRC_MJ_TREE_MODEL_H_

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "mj_tree.hip.h"

namespace epa {
namespace mj {
namespace tree {

class TreeBuilder {
 public:
  TreeBuilder() {
    std::memset(&m_, 0, sizeof(m_));
    m_.nbody = 1;  // world
    m_.body_quat[0][0] = 1;
    m_.body_jntadr[0] = -1;
    m_.body_dofadr[0] = -1;
    m_.ngeom = 1;  // geom 0: the floor plane z = 0
    m_.geom_type[0] = kGeomPlane;
  }
  int AddBody(int parent, double x, double y, double z, double qw = 1, double qx = 0, double qy = 0,
              double qz = 0) {
    const int b = m_.nbody++;
    m_.body_parent[b] = parent;
    m_.body_pos[b][0] = x;
    m_.body_pos[b][1] = y;
    m_.body_pos[b][2] = z;
    const double n = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    m_.body_quat[b][0] = qw / n;
    m_.body_quat[b][1] = qx / n;
    m_.body_quat[b][2] = qy / n;
    m_.body_quat[b][3] = qz / n;
    m_.body_jntadr[b] = -1;
    m_.body_dofadr[b] = -1;
    return b;
  }
  int AddFree(int body) { return AddJoint(body, kJntFree, 0, 0, 0, 0, 0, 1, false, 0, 0, 0, 0, 0); }
  int AddHinge(int body, double px, double py, double pz, double ax, double ay, double az,
               double lo_deg, double hi_deg, double stiffness, double damping, double armature) {
    const double d2r = 3.14159265358979323846 / 180.0;
    return AddJoint(body, kJntHinge, px, py, pz, ax, ay, az, true, lo_deg * d2r, hi_deg * d2r,
                    stiffness, damping, armature);
  }
  int AddSphere(int body, double x, double y, double z, double r) {
    const int g = m_.ngeom++;
    m_.geom_type[g] = kGeomSphere;
    m_.geom_body[g] = body;
    m_.geom_pos[g][0] = x;
    m_.geom_pos[g][1] = y;
    m_.geom_pos[g][2] = z;
    m_.geom_axis[g][2] = 1;
    m_.geom_rad[g] = r;
    m_.geom_hl[g] = 0;
    return g;
  }
  int AddCapsule(int body, double x0, double y0, double z0, double x1, double y1, double z1, double r) {
    const int g = m_.ngeom++;
    const double d[3] = {x1 - x0, y1 - y0, z1 - z0};
    const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    m_.geom_type[g] = kGeomCapsule;
    m_.geom_body[g] = body;
    m_.geom_pos[g][0] = 0.5 * (x0 + x1);
    m_.geom_pos[g][1] = 0.5 * (y0 + y1);
    m_.geom_pos[g][2] = 0.5 * (z0 + z1);
    for (int k = 0; k < 3; ++k) m_.geom_axis[g][k] = d[k] / len;
    m_.geom_rad[g] = r;
    m_.geom_hl[g] = 0.5 * len;
    return g;
  }
  int AddMotor(int jnt, double gear) {
    const int u = m_.nu++;
    m_.act_dof[u] = m_.jnt_dadr[jnt];
    m_.act_gear[u] = gear;
    return u;
  }
  TreeModel& model() { return m_; }

  // density: kg / m^3 of every geom (MuJoCo default 1000)
  TreeModel Compile(double density = 1000.0) {
    TreeModel& m = m_;
    const double pi = 3.14159265358979323846;
    // inertiafromgeom: analytic sphere / capsule, composed about the body COM
    m.total_mass = 0;
    for (int b = 1; b < m.nbody; ++b) {
      double mass = 0, com[3] = {0, 0, 0};
      std::vector<double> gm(m.ngeom, 0.0);
      std::vector<double> gia(m.ngeom, 0.0), git(m.ngeom, 0.0);  // axial / transverse inertia
      for (int g = 1; g < m.ngeom; ++g) {
        if (m.geom_body[g] != b) continue;
        const double r = m.geom_rad[g];
        if (m.geom_type[g] == kGeomSphere) {
          gm[g] = density * 4.0 / 3.0 * pi * r * r * r;
          gia[g] = git[g] = 0.4 * gm[g] * r * r;
        } else {
          const double h = 2 * m.geom_hl[g];
          gm[g] = density * pi * (r * r * h + 4.0 * r * r * r / 3.0);
          const double sm = gm[g] * 4 * r / (4 * r + 3 * h), cm = gm[g] - sm;
          git[g] = cm * (3 * r * r + h * h) / 12 + 0.4 * sm * r * r + sm * h * (3 * r + 2 * h) / 8;
          gia[g] = cm * r * r / 2 + 0.4 * sm * r * r;
        }
        mass += gm[g];
        for (int k = 0; k < 3; ++k) com[k] += gm[g] * m.geom_pos[g][k];
      }
      for (int k = 0; k < 3; ++k) com[k] /= mass;
      double I[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int g = 1; g < m.ngeom; ++g) {
        if (m.geom_body[g] != b) continue;
        const double* a = m.geom_axis[g];
        double d[3];
        for (int k = 0; k < 3; ++k) d[k] = m.geom_pos[g][k] - com[k];
        const double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c) {
            // it (1 - a a') + ia a a'  +  m (d2 1 - d d')
            I[r][c] += git[g] * ((r == c ? 1.0 : 0.0) - a[r] * a[c]) + gia[g] * a[r] * a[c] +
                       gm[g] * ((r == c ? d2 : 0.0) - d[r] * d[c]);
          }
        }
      }
      m.body_mass[b] = mass;
      for (int k = 0; k < 3; ++k) m.body_ipos[b][k] = com[k];
      m.body_inertia[b][0] = I[0][0];
      m.body_inertia[b][1] = I[1][1];
      m.body_inertia[b][2] = I[2][2];
      m.body_inertia[b][3] = I[0][1];
      m.body_inertia[b][4] = I[0][2];
      m.body_inertia[b][5] = I[1][2];
      m.total_mass += mass;
    }
    // dof tree
    for (int d = 0; d < m.nv; ++d) {
      const int b = m.dof_body[d];
      if (d > m.body_dofadr[b]) {
        m.dof_parent[d] = d - 1;
      } else {
        int p = m.body_parent[b];
        while (p > 0 && m.body_dofnum[p] == 0) p = m.body_parent[p];
        m.dof_parent[d] = p > 0 ? m.body_dofadr[p] + m.body_dofnum[p] - 1 : -1;
      }
    }
    for (int b = 1; b < m.nbody; ++b) {
      int w = b;
      while (w > 0 && m.body_dofnum[w] == 0) w = m.body_parent[w];
      unsigned mask = 0;
      for (int d = w > 0 ? m.body_dofadr[w] + m.body_dofnum[w] - 1 : -1; d >= 0; d = m.dof_parent[d]) {
        mask |= 1u << d;
      }
      m.body_dofmask[b] = mask;
    }
    // qpos0
    for (int j = 0; j < m.njnt; ++j) {
      const int a = m.jnt_qadr[j], b = m.jnt_body[j];
      if (m.jnt_type[j] == kJntFree) {
        for (int k = 0; k < 3; ++k) m.qpos0[a + k] = m.body_pos[b][k];
        for (int k = 0; k < 4; ++k) m.qpos0[a + 3 + k] = m.body_quat[b][k];
      } else {
        m.qpos0[a] = 0;
      }
    }
    // constraint groups
    m.nlimit = 0;
    for (int j = 0; j < m.njnt; ++j) {
      if (m.jnt_limited[j]) m.limit_jnt[m.nlimit++] = j;
    }
    m.nfloor = 0;
    for (int g = 1; g < m.ngeom; ++g) {
      if (m.geom_type[g] == kGeomCapsule) {  // mjc_PlaneCapsule: +axis end first
        m.floor_geom[m.nfloor] = g;
        m.floor_sign[m.nfloor++] = 1;
        m.floor_geom[m.nfloor] = g;
        m.floor_sign[m.nfloor++] = -1;
      } else {
        m.floor_geom[m.nfloor] = g;
        m.floor_sign[m.nfloor++] = 0;
      }
    }
    m.npair = 0;
    std::vector<int> weld(m.nbody, 0);
    for (int b = 1; b < m.nbody; ++b) {
      int w = b;
      while (w > 0 && m.body_dofnum[w] == 0) w = m.body_parent[w];
      weld[b] = w;
    }
    for (int ga = 1; ga < m.ngeom; ++ga) {
      for (int gb = ga + 1; gb < m.ngeom; ++gb) {
        const int w1 = weld[m.geom_body[ga]], w2 = weld[m.geom_body[gb]];
        if (w1 == w2) continue;  // same (welded) body
        const int p1 = weld[m.body_parent[w1]], p2 = weld[m.body_parent[w2]];
        if (w1 != 0 && w2 != 0 && (w1 == p2 || w2 == p1)) continue;  // filterparent
        if (m.npair >= kMaxPair) throw std::runtime_error("mj_tree: too many geom pairs");
        const bool swap = m.geom_type[ga] > m.geom_type[gb];  // mj_collideGeoms: lower type first
        m.pair_g1[m.npair] = swap ? gb : ga;
        m.pair_g2[m.npair++] = swap ? ga : gb;
      }
    }
    if (m.nlimit + m.nfloor + m.npair > kMaxGroup) throw std::runtime_error("mj_tree: too many groups");
    SetConst();
    return m;
  }

 private:
  int AddJoint(int body, int type, double px, double py, double pz, double ax, double ay, double az,
               bool limited, double lo, double hi, double stiffness, double damping, double armature) {
    TreeModel& m = m_;
    const int j = m.njnt++;
    m.jnt_type[j] = type;
    m.jnt_body[j] = body;
    m.jnt_pos[j][0] = px;
    m.jnt_pos[j][1] = py;
    m.jnt_pos[j][2] = pz;
    const double n = std::sqrt(ax * ax + ay * ay + az * az);
    m.jnt_axis[j][0] = ax / n;
    m.jnt_axis[j][1] = ay / n;
    m.jnt_axis[j][2] = az / n;
    m.jnt_limited[j] = limited;
    m.jnt_lo[j] = lo;
    m.jnt_hi[j] = hi;
    m.jnt_stiff[j] = stiffness;
    m.jnt_qadr[j] = m.nq;
    m.jnt_dadr[j] = m.nv;
    if (m.body_jntadr[body] < 0) {
      m.body_jntadr[body] = j;
      m.body_dofadr[body] = m.nv;
    }
    m.body_jntnum[body]++;
    const int ndof = type == kJntFree ? 6 : 1;
    for (int k = 0; k < ndof; ++k) {
      m.dof_body[m.nv + k] = body;
      m.dof_arm[m.nv + k] = armature;
      m.dof_damp[m.nv + k] = damping;
    }
    m.nv += ndof;
    m.nq += type == kJntFree ? 7 : 1;
    m.body_dofnum[body] += ndof;
    return j;
  }

  // mj_setConst: M at qpos0 -> dof_invweight0, body_invweight0, meaninertia.  Plain dense
  // algebra with world-frame Jacobians (independent of the kernel's c-frame recursion).
  void SetConst() {
    TreeModel& m = m_;
    const int nv = m.nv, nb = m.nbody;
    std::vector<double> xpos(3 * nb, 0.0), xmat(9 * nb, 0.0), xipos(3 * nb, 0.0);
    std::vector<double> anchor(3 * m.njnt, 0.0), axis(3 * m.njnt, 0.0);
    xmat[0] = xmat[4] = xmat[8] = 1;
    auto qmat = [](const double* q, double* M) {
      Quat qq = {q[0], q[1], q[2], q[3]};
      QMat(qq, M);
    };
    std::vector<double> xquat(4 * nb, 0.0);
    xquat[0] = 1;
    for (int b = 1; b < nb; ++b) {
      const int p = m.body_parent[b];
      for (int r = 0; r < 3; ++r) {
        xpos[3 * b + r] = xpos[3 * p + r];
        for (int c = 0; c < 3; ++c) xpos[3 * b + r] += xmat[9 * p + 3 * r + c] * m.body_pos[b][c];
      }
      Quat q = QMul(Quat{xquat[4 * p], xquat[4 * p + 1], xquat[4 * p + 2], xquat[4 * p + 3]},
                    Quat{m.body_quat[b][0], m.body_quat[b][1], m.body_quat[b][2], m.body_quat[b][3]});
      if (p == 0) q = Quat{m.body_quat[b][0], m.body_quat[b][1], m.body_quat[b][2], m.body_quat[b][3]};
      xquat[4 * b] = q.w;
      xquat[4 * b + 1] = q.x;
      xquat[4 * b + 2] = q.y;
      xquat[4 * b + 3] = q.z;
      qmat(&xquat[4 * b], &xmat[9 * b]);  // all hinges at 0
      for (int r = 0; r < 3; ++r) {
        xipos[3 * b + r] = xpos[3 * b + r];
        for (int c = 0; c < 3; ++c) xipos[3 * b + r] += xmat[9 * b + 3 * r + c] * m.body_ipos[b][c];
      }
      for (int j = m.body_jntadr[b]; j >= 0 && j < m.body_jntadr[b] + m.body_jntnum[b]; ++j) {
        for (int r = 0; r < 3; ++r) {
          anchor[3 * j + r] = xpos[3 * b + r];
          axis[3 * j + r] = 0;
          for (int c = 0; c < 3; ++c) {
            anchor[3 * j + r] += xmat[9 * b + 3 * r + c] * m.jnt_pos[j][c];
            axis[3 * j + r] += xmat[9 * b + 3 * r + c] * m.jnt_axis[j][c];
          }
        }
      }
    }
    // body Jacobians at the body COM: Jp (3 x nv), Jr (3 x nv)
    auto jac = [&](int b, const double* point, std::vector<double>& Jp, std::vector<double>& Jr) {
      Jp.assign(3 * nv, 0.0);
      Jr.assign(3 * nv, 0.0);
      for (int d = 0; d < nv; ++d) {
        if (!((m.body_dofmask[b] >> d) & 1u)) continue;
        const int jb = m.dof_body[d];
        int j = m.body_jntadr[jb];
        while (!(m.jnt_dadr[j] <= d && d < m.jnt_dadr[j] + (m.jnt_type[j] == kJntFree ? 6 : 1))) ++j;
        if (m.jnt_type[j] == kJntFree) {
          const int k = d - m.jnt_dadr[j];
          if (k < 3) {
            Jp[k * nv + d] = 1;
          } else {  // rotation about body axis k-3
            double ax[3], off[3];
            for (int r = 0; r < 3; ++r) {
              ax[r] = xmat[9 * jb + 3 * r + (k - 3)];
              off[r] = point[r] - xpos[3 * jb + r];
            }
            Jr[0 * nv + d] = ax[0];
            Jr[1 * nv + d] = ax[1];
            Jr[2 * nv + d] = ax[2];
            Jp[0 * nv + d] = ax[1] * off[2] - ax[2] * off[1];
            Jp[1 * nv + d] = ax[2] * off[0] - ax[0] * off[2];
            Jp[2 * nv + d] = ax[0] * off[1] - ax[1] * off[0];
          }
        } else {
          double off[3];
          const double* ax = &axis[3 * j];
          for (int r = 0; r < 3; ++r) off[r] = point[r] - anchor[3 * j + r];
          Jr[0 * nv + d] = ax[0];
          Jr[1 * nv + d] = ax[1];
          Jr[2 * nv + d] = ax[2];
          Jp[0 * nv + d] = ax[1] * off[2] - ax[2] * off[1];
          Jp[1 * nv + d] = ax[2] * off[0] - ax[0] * off[2];
          Jp[2 * nv + d] = ax[0] * off[1] - ax[1] * off[0];
        }
      }
    };
    // M = sum_b m Jp'Jp + Jr' Iw Jr  (+ armature)
    std::vector<double> M(nv * nv, 0.0), Jp, Jr;
    for (int b = 1; b < nb; ++b) {
      jac(b, &xipos[3 * b], Jp, Jr);
      const double* R = &xmat[9 * b];
      const double* in = m.body_inertia[b];
      const double Ib[3][3] = {{in[0], in[3], in[4]}, {in[3], in[1], in[5]}, {in[4], in[5], in[2]}};
      double Iw[3][3];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
          double s = 0;
          for (int k = 0; k < 3; ++k) {
            for (int l = 0; l < 3; ++l) s += R[3 * r + k] * Ib[k][l] * R[3 * c + l];
          }
          Iw[r][c] = s;
        }
      }
      for (int i = 0; i < nv; ++i) {
        for (int j = 0; j < nv; ++j) {
          double s = 0;
          for (int r = 0; r < 3; ++r) {
            s += m.body_mass[b] * Jp[r * nv + i] * Jp[r * nv + j];
            for (int c = 0; c < 3; ++c) s += Jr[r * nv + i] * Iw[r][c] * Jr[c * nv + j];
          }
          M[i * nv + j] += s;
        }
      }
    }
    double tr = 0;
    for (int i = 0; i < nv; ++i) {
      M[i * nv + i] += m.dof_arm[i];
      tr += M[i * nv + i];
    }
    m.meaninertia = tr / nv;
    // Minv by Cholesky
    std::vector<double> Lc(M), Minv(nv * nv, 0.0);
    for (int j = 0; j < nv; ++j) {
      double s = Lc[j * nv + j];
      for (int k = 0; k < j; ++k) s -= Lc[j * nv + k] * Lc[j * nv + k];
      const double d = std::sqrt(s);
      Lc[j * nv + j] = d;
      for (int i = j + 1; i < nv; ++i) {
        double t = Lc[i * nv + j];
        for (int k = 0; k < j; ++k) t -= Lc[i * nv + k] * Lc[j * nv + k];
        Lc[i * nv + j] = t / d;
      }
    }
    for (int c = 0; c < nv; ++c) {
      std::vector<double> x(nv, 0.0);
      x[c] = 1;
      for (int i = 0; i < nv; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= Lc[i * nv + k] * x[k];
        x[i] = s / Lc[i * nv + i];
      }
      for (int i = nv - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < nv; ++k) s -= Lc[k * nv + i] * x[k];
        x[i] = s / Lc[i * nv + i];
      }
      for (int r = 0; r < nv; ++r) Minv[r * nv + c] = x[r];
    }
    for (int j = 0; j < m.njnt; ++j) {
      const int a = m.jnt_dadr[j];
      if (m.jnt_type[j] == kJntFree) {
        const double t = (Minv[a * nv + a] + Minv[(a + 1) * nv + a + 1] + Minv[(a + 2) * nv + a + 2]) / 3;
        const double r = (Minv[(a + 3) * nv + a + 3] + Minv[(a + 4) * nv + a + 4] + Minv[(a + 5) * nv + a + 5]) / 3;
        for (int k = 0; k < 3; ++k) {
          m.dof_invw[a + k] = t;
          m.dof_invw[a + 3 + k] = r;
        }
      } else {
        m.dof_invw[a] = Minv[a * nv + a];
      }
    }
    for (int b = 1; b < nb; ++b) {
      jac(b, &xipos[3 * b], Jp, Jr);
      double tp = 0;
      for (int r = 0; r < 3; ++r) {
        for (int i = 0; i < nv; ++i) {
          for (int j = 0; j < nv; ++j) tp += Jp[r * nv + i] * Minv[i * nv + j] * Jp[r * nv + j];
        }
      }
      m.body_invw[b] = tp / 3;
    }
  }

  TreeModel m_;
};

// humanoid_envpool.xml / humanoidstandup_envpool.xml (`su`): same tree, the standup model
// lies on its back so only placements along the spine and the legs differ (and the lower
// range of left_hip_y).
//   <compiler angle="degree" inertiafromgeom="true"/>                                  :18
//   <joint armature="1" damping="1" limited="true"/>                                   :20
//   <geom conaffinity="1" condim="1" contype="1" margin="0.001" .../>                  :21
//   <motor ctrllimited="true" ctrlrange="-.4 .4"/>                                     :22
//   <option integrator="RK4" iterations="50" solver="PGS" timestep="0.003">            :24
//   floor: condim 3, friction 1 .1 .1                                                  :41
// The two fixed tendons (:107-116) carry no limit, spring or actuator: no dynamics.
inline TreeModel BuildHumanoidModel(bool su) {
  TreeBuilder tb;
  TreeModel& m = tb.model();
  m.timestep = 0.003;
  m.gravity = 9.81;
  m.iterations = 50;
  m.margin = 0.001;
  m.floor_mu = 1.0;
  m.ctrl_lo = -0.4;
  m.ctrl_hi = 0.4;
  {  // default solref 0.02 1 (timeconst >= 2 h holds), solimp 0.9 0.95 0.001 0.5 2
    const double tc = 0.02, dr = 1.0, dmax = 0.95;
    m.sol_K = 1.0 / (dmax * dmax * tc * tc * dr * dr);
    m.sol_B = 2.0 / (dmax * tc);
    m.sol_d0 = 0.9;
    m.sol_dmax = dmax;
    m.sol_width = 0.001;
  }
  const int torso = tb.AddBody(0, 0, 0, su ? 0.105 : 1.4);  // :43
  tb.AddFree(torso);                                         // :45
  tb.AddCapsule(torso, 0, -.07, 0, 0, .07, 0, 0.07);         // torso1 :46
  if (su) {
    tb.AddSphere(torso, -.15, 0, 0, .09);
    tb.AddCapsule(torso, .11, -.06, 0, .11, .06, 0, 0.06);
  } else {
    tb.AddSphere(torso, 0, 0, .19, .09);                          // head :47
    tb.AddCapsule(torso, -.01, -.06, -.12, -.01, .06, -.12, 0.06);  // uwaist :48
  }
  const int lwaist = su ? tb.AddBody(torso, .21, 0, 0, 1.0, 0, -0.002, 0)
                        : tb.AddBody(torso, -.01, 0, -0.260, 1.0, 0, -0.002, 0);  // :49
  tb.AddCapsule(lwaist, 0, -.06, 0, 0, .06, 0, 0.06);                            // :50
  const int abdomen_z = tb.AddHinge(lwaist, 0, 0, 0.065, 0, 0, 1, -45, 45, 20, 5, 0.02);  // :51
  const int abdomen_y = tb.AddHinge(lwaist, 0, 0, 0.065, 0, 1, 0, -75, 30, 10, 5, 0.02);  // :52
  const int pelvis = su ? tb.AddBody(lwaist, 0.165, 0, 0, 1.0, 0, -0.002, 0)
                        : tb.AddBody(lwaist, 0, 0, -0.165, 1.0, 0, -0.002, 0);  // :53
  const int abdomen_x = tb.AddHinge(pelvis, 0, 0, 0.1, 1, 0, 0, -35, 35, 10, 5, 0.02);  // :54
  tb.AddCapsule(pelvis, -.02, -.07, 0, -.02, .07, 0, 0.09);                            // butt :55
  int hip[2][3], knee[2];
  for (int side = 0; side < 2; ++side) {  // right :56-68, left :69-81
    const double s = side == 0 ? -1.0 : 1.0;
    const int thigh = tb.AddBody(pelvis, 0, s * 0.1, su ? 0 : -0.04);
    if (side == 0) {
      hip[0][0] = tb.AddHinge(thigh, 0, 0, 0, 1, 0, 0, -25, 5, 10, 5, 0.01);     // right_hip_x :57
      hip[0][1] = tb.AddHinge(thigh, 0, 0, 0, 0, 0, 1, -60, 35, 10, 5, 0.01);    // right_hip_z :58
      hip[0][2] = tb.AddHinge(thigh, 0, 0, 0, 0, 1, 0, -110, 20, 20, 5, 0.008);  // right_hip_y :59
    } else {
      hip[1][0] = tb.AddHinge(thigh, 0, 0, 0, -1, 0, 0, -25, 5, 10, 5, 0.01);   // left_hip_x :70
      hip[1][1] = tb.AddHinge(thigh, 0, 0, 0, 0, 0, -1, -60, 35, 10, 5, 0.01);  // left_hip_z :71
      hip[1][2] = tb.AddHinge(thigh, 0, 0, 0, 0, 1, 0, su ? -120 : -110, 20, 20, 5, 0.01);  // :72
    }
    int shin;
    if (su) {
      tb.AddCapsule(thigh, 0, 0, 0, 0.34, -s * 0.01, 0, 0.06);
      shin = tb.AddBody(thigh, 0.403, -s * 0.01, 0);
    } else {
      tb.AddCapsule(thigh, 0, 0, 0, 0, -s * 0.01, -.34, 0.06);  // :60 / :73
      shin = tb.AddBody(thigh, 0, -s * 0.01, -0.403);           // :61 / :74
    }
    // knees :62 / :75 (only the left one has stiffness="1"); damping: class default
    knee[side] = tb.AddHinge(shin, 0, 0, .02, 0, -1, 0, -160, -2, side == 0 ? 0 : 1, 1, 0.006);
    int foot;
    if (su) {
      tb.AddCapsule(shin, 0, 0, 0, 0.3, 0, 0, 0.049);
      foot = tb.AddBody(shin, 0.35, 0, -.10);
    } else {
      tb.AddCapsule(shin, 0, 0, 0, 0, 0, -.3, 0.049);  // :63
      foot = tb.AddBody(shin, 0, 0, -0.45);            // :64
    }
    tb.AddSphere(foot, 0, 0, 0.1, 0.075);  // :65
  }
  int shoulder[2][2], elbow[2];
  for (int side = 0; side < 2; ++side) {  // right :84-94, left :95-104
    const double s = side == 0 ? -1.0 : 1.0;
    const int uarm = tb.AddBody(torso, 0, s * 0.17, 0.06);
    if (side == 0) {
      shoulder[0][0] = tb.AddHinge(uarm, 0, 0, 0, 2, 1, 1, -85, 60, 1, 1, 0.0068);   // :85
      shoulder[0][1] = tb.AddHinge(uarm, 0, 0, 0, 0, -1, 1, -85, 60, 1, 1, 0.0051);  // :86
    } else {
      shoulder[1][0] = tb.AddHinge(uarm, 0, 0, 0, 2, -1, 1, -60, 85, 1, 1, 0.0068);  // :96
      shoulder[1][1] = tb.AddHinge(uarm, 0, 0, 0, 0, 1, 1, -60, 85, 1, 1, 0.0051);   // :97
    }
    tb.AddCapsule(uarm, 0, 0, 0, .16, s * .16, -.16, 0.04);  // :87 / :98
    const int larm = tb.AddBody(uarm, .18, s * .18, -.18);   // :88 / :99
    elbow[side] = side == 0 ? tb.AddHinge(larm, 0, 0, 0, 0, -1, 1, -90, 50, 0, 1, 0.0028)     // :89
                            : tb.AddHinge(larm, 0, 0, 0, 0, -1, -1, -90, 50, 0, 1, 0.0028);  // :100
    tb.AddCapsule(larm, 0.01, -s * 0.01, 0.01, .17, -s * .17, .17, 0.031);  // :90 / :101
    tb.AddSphere(larm, .18, -s * .18, .18, 0.04);                           // :91 / :102
  }
  // actuators :118-136
  tb.AddMotor(abdomen_y, 100);
  tb.AddMotor(abdomen_z, 100);
  tb.AddMotor(abdomen_x, 100);
  for (int side = 0; side < 2; ++side) {
    tb.AddMotor(hip[side][0], 100);
    tb.AddMotor(hip[side][1], 100);
    tb.AddMotor(hip[side][2], 300);
    tb.AddMotor(knee[side], 200);
  }
  for (int side = 0; side < 2; ++side) {
    tb.AddMotor(shoulder[side][0], 25);
    tb.AddMotor(shoulder[side][1], 25);
    tb.AddMotor(elbow[side], 25);
  }
  return tb.Compile();
}

}  // namespace tree
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_TREE_MODEL_H_
