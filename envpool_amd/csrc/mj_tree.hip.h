// K3c -- `mj_step` for a general 3-D kinematic tree with a floating base (gym Humanoid /
// HumanoidStandup: 13 bodies, free joint + 17 hinges, nv = 23, sphere / capsule geoms that
// also collide with each other, PGS solver, RK4).
//
// What it replaces: the arithmetic MuJoCo 3.6.0 performs each time the reference calls
// mj_forward / mj_step / mj_rnePostConstraint for these models
// (envpool/mujoco/gym/mujoco_env.h:126-148; task code envpool/mujoco/gym/humanoid.h,
// humanoid_standup.h).  The engine itself is un-vendored third-party code: see
// oracle/mjcpu/mjcpu.h (PARITY UNPINNED) for what the restatement is anchored on.
//
// MI355X-first design.  One env per lane, one wave per workgroup.  A 23-dof tree does not
// fit a lane's registers (M alone is 185 numbers, a constraint row 23, up to ~150 rows), so
// unlike the planar / Ant kernels the per-env data lives in an HBM workspace:
//  * layout [wave][slot][lane]: a wave's 64 columns of one slot are one 512 B line (fully
//    coalesced), a wave's whole workspace is one contiguous block (TLB / DRAM page
//    locality), and every slot index is wave-uniform, so an address is an SGPR base plus
//    the lane offset.  The block belongs to the launch's wave, not to an env: the few
//    values that persist between steps (qpos, qvel, warm start, lagged mass centre) are
//    copied in from / out to a per-env SoA by the step kernel;
//  * the tree is a compile-time constant (gen_mj_consts.cpp) and every stage has the shape
//    [issue all loads] -> fence -> [arithmetic in VGPRs] -> [stores]: kinematics and the
//    velocity pass are depth-first template recursions with the parent's frame in
//    registers, M is built, loaded once and factored (tree-sparse L'DL) in registers, an
//    M^-1 solve loads the factor in one batch.  A wave runs alone on its SIMD, so a
//    dependent HBM round trip has nothing to overlap it: batching is what counts;
//  * constraint candidates are STATIC lists (limited joints, floor spheres, geom pairs);
//    detection leaves a per-lane bitmask of active groups, and rows are built into COMPACT
//    slots, every lane taking ITS t-th active group in iteration t -- the wave pays
//    max-over-lanes rows, each lane keeps MuJoCo's row order (which the unconverged,
//    order-dependent PGS sweep needs);
//  * PGS: with <= 16 compact rows A + R, b and f live in VGPRs and the sweeps touch no
//    memory; otherwise the sweep works on  a = qacc_smooth + M^-1 J' f  (res_r = J_r . a -
//    aref_r + R_r f_r,  a += delta W_r,  W_r = M^-1 J_r') and streams (J_r, W_r) pairs
//    through a register ring prefetched two visits ahead;
//  * no lane-divergent control flow except one branch: per-lane differences are selects,
//    finished lanes run with frozen iterates (WaveAny, see mj_cheetah.hip.h); only in the
//    streaming sweep do converged lanes skip their row traffic.
// The same source compiles for the host (EPA_HD, lane stride 1) so tests run it on the CPU
// against oracle/mjcpu.
#ifndef ENVPOOL_AMD_CSRC_MJ_TREE_HIP_H_
#define ENVPOOL_AMD_CSRC_MJ_TREE_HIP_H_

#include "mj_cheetah.hip.h"  // EPA_HD, static_for, WaveAny, WaveUniform, SinCos

namespace epa {
namespace mj {
namespace tree {

constexpr int kMaxBody = 16, kMaxJnt = 20, kMaxV = 24, kMaxQ = 24, kMaxGeom = 20, kMaxU = 20;
constexpr int kMaxFloor = 32, kMaxPair = 128, kMaxGroup = 192;
enum { kJntFree = 0, kJntHinge = 3 };
enum { kGeomPlane = 0, kGeomSphere = 2, kGeomCapsule = 3 };

#if defined(__HIP_DEVICE_COMPILE__)
constexpr int kLaneStride = 64;
#else
constexpr int kLaneStride = 1;
#endif

// The compiled model (mj_tree_model.h builds it on the host; the kernels see it as a
// compile-time constant).  Single tree: body 1 carries the free joint, every other joint
// is a hinge; geom 0 is the floor plane z = 0.
struct TreeModel {
  int nbody, njnt, nq, nv, ngeom, nu;
  int nlimit, nfloor, npair;  // constraint groups: limited joints, floor spheres, geom pairs
  int iterations;             // <option iterations>
  double timestep, gravity;   // gravity: magnitude along -z
  double margin, floor_mu;    // contact margin of every pair; sliding friction on the floor
  double sol_K, sol_B, sol_d0, sol_dmax, sol_width;  // default solref / solimp everywhere
  double meaninertia, total_mass;
  int body_parent[kMaxBody], body_jntadr[kMaxBody], body_jntnum[kMaxBody];
  int body_dofadr[kMaxBody], body_dofnum[kMaxBody];
  unsigned body_dofmask[kMaxBody];  // dofs on the path world -> body
  double body_pos[kMaxBody][3], body_quat[kMaxBody][4], body_ipos[kMaxBody][3];
  double body_mass[kMaxBody], body_inertia[kMaxBody][6];  // xx yy zz xy xz yz about ipos
  double body_invw[kMaxBody];                             // body_invweight0, translational
  int jnt_type[kMaxJnt], jnt_body[kMaxJnt], jnt_qadr[kMaxJnt], jnt_dadr[kMaxJnt];
  int jnt_limited[kMaxJnt];
  double jnt_pos[kMaxJnt][3], jnt_axis[kMaxJnt][3], jnt_lo[kMaxJnt], jnt_hi[kMaxJnt];
  double jnt_stiff[kMaxJnt];
  int dof_parent[kMaxV], dof_body[kMaxV];
  double dof_arm[kMaxV], dof_damp[kMaxV], dof_invw[kMaxV];
  int geom_type[kMaxGeom], geom_body[kMaxGeom];
  double geom_pos[kMaxGeom][3], geom_axis[kMaxGeom][3], geom_rad[kMaxGeom], geom_hl[kMaxGeom];
  int act_dof[kMaxU];
  double act_gear[kMaxU], ctrl_lo, ctrl_hi;
  int limit_jnt[kMaxJnt];                              // limit group -> joint
  int floor_geom[kMaxFloor]; double floor_sign[kMaxFloor];  // sphere = gpos + sign * hl * axis
  int pair_g1[kMaxPair], pair_g2[kMaxPair];            // lower geom TYPE first (mj_collideGeoms)
  double qpos0[kMaxQ];
};

// ---- workspace ------------------------------------------------------------------------
struct Layout {
  int qpos, qvel, warm, lag, npersist;               // slots [0, npersist) persist between steps
  int ctrl;
  int xpos, xmat, xipos, anchor, axis, gpos, gaxis, com;
  int cinert, cdof, cvel, crb, cacc;
  int M, dinv;                                       // M (lower, tree-sparse) -> L'DL in place
  int act, accs, qacc;                               // qfrc_actuator, qacc_smooth, qacc
  int limd, lims, condist, conpos, connrm;           // detection results per group
  // constraint rows as 16-byte pairs per lane: rowJW pair r * nv + i = (J_i, W_i); rowS pairs
  // 3 r .. 3 r + 2 = (f, 1 / A_rr), (A_rr, R), (aref, -).  Slot numbers are in doubles.
  int rowJW, rowS;
  int ccpos, ccnrm, ccbody;                          // compact contact records (floor, then pairs)
  int x0q, x0v, accq, accv;                          // RK4
  int cext;                                          // cfrc_ext [nbody][6]
  int total;
};
constexpr Layout MakeLayout(const TreeModel& m) {
  Layout L{};
  int s = 0;
  auto take = [&s](int n) { int at = s; s += n; return at; };
  const int ncon = m.nfloor + m.npair, nrow = m.nlimit + 4 * m.nfloor + m.npair;
  L.qpos = take(m.nq); L.qvel = take(m.nv); L.warm = take(m.nv); L.lag = take(2);
  L.npersist = s;
  L.ctrl = take(m.nu);
  L.xpos = take(3 * m.nbody); L.xmat = take(9 * m.nbody);
  L.xipos = take(3 * m.nbody); L.anchor = take(3 * m.njnt); L.axis = take(3 * m.njnt);
  L.gpos = take(3 * m.ngeom); L.gaxis = take(3 * m.ngeom); L.com = take(3);
  L.cinert = take(10 * m.nbody); L.cdof = take(6 * m.nv); L.cvel = take(6 * m.nbody);
  L.crb = take(10 * m.nbody); L.cacc = take(6 * m.nbody);
  L.M = take(m.nv * m.nv); L.dinv = take(m.nv);
  L.act = take(m.nv); L.accs = take(m.nv); L.qacc = take(m.nv);
  L.limd = take(m.nlimit); L.lims = take(m.nlimit);
  L.condist = take(ncon); L.conpos = take(3 * ncon); L.connrm = take(3 * ncon);
  s += s & 1;  // 16-byte alignment of the pair regions
  L.rowJW = take(2 * nrow * m.nv); L.rowS = take(2 * 3 * nrow);
  L.ccpos = take(3 * ncon); L.ccnrm = take(3 * ncon); L.ccbody = take(2 * ncon);
  L.x0q = take(m.nq); L.x0v = take(m.nv); L.accq = take(m.nv); L.accv = take(m.nv);
  L.cext = take(6 * m.nbody);
  L.total = s;
  return L;
}

struct alignas(16) D2 {
  double x, y;
};

// `base` is the wave's block (wave-uniform: it lives in SGPRs and slot offsets are scalar
// arithmetic), `lane` the column inside it.
struct Ws {
  double* base;
  unsigned lane;
  // byte offset in 32 bits: the access becomes `global_load v, v_off, s[base]` (SGPR base +
  // 32-bit lane offset) instead of a 64-bit per-lane address per slot
  // (signed arithmetic: no-wrap lets alias analysis tell two slots apart, so loads are not
  // pinned behind unrelated stores)
  EPA_HD double& operator()(int slot) const {
    const int off = (slot * kLaneStride + (int)lane) * 8;
    return *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off);
  }
  // 16-byte element `pair` of a pair region starting at (even) slot `region`
  EPA_HD D2& Pair(int region, int pair) const {
    const int off = (region * kLaneStride + (pair * kLaneStride + (int)lane) * 2) * 8;
    return *reinterpret_cast<D2*>(reinterpret_cast<char*>(base) + off);
  }
  // A fresh copy whose lane offset is opaque to the optimiser: used at the top of loop bodies
  // so that the (hundreds of) slot addresses are recomputed where needed instead of being
  // hoisted out of the loop and spilled.
  EPA_HD Ws Fresh() const {
    Ws r = *this;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(r.lane));
    r.lane &= (unsigned)(kLaneStride - 1);  // keeps the known range [0, 64)
#endif
    return r;
  }
};

// per-group tables indexed at run time by a wave-uniform group number
struct GroupTab {
  int b1[kMaxGroup], b2[kMaxGroup];           // bodies of a contact group (b1 = 0: floor)
  unsigned mask1[kMaxGroup], mask2[kMaxGroup];
  int dof[kMaxGroup];                         // limit group: its dof
  double diag[kMaxGroup];                     // mj_diagApprox of the (first) row
};
constexpr GroupTab MakeGroupTab(const TreeModel& m) {
  GroupTab t{};
  for (int g = 0; g < m.nlimit; ++g) {
    t.dof[g] = m.jnt_dadr[m.limit_jnt[g]];
    t.diag[g] = m.dof_invw[t.dof[g]];
  }
  for (int c = 0; c < m.nfloor; ++c) {
    const int g = m.nlimit + c, b = m.geom_body[m.floor_geom[c]];
    t.b1[g] = 0;
    t.b2[g] = b;
    t.mask1[g] = 0;
    t.mask2[g] = m.body_dofmask[b];
    t.diag[g] = m.body_invw[b] * (1.0 + m.floor_mu * m.floor_mu);
  }
  for (int p = 0; p < m.npair; ++p) {
    const int g = m.nlimit + m.nfloor + p;
    const int b1 = m.geom_body[m.pair_g1[p]], b2 = m.geom_body[m.pair_g2[p]];
    t.b1[g] = b1;
    t.b2[g] = b2;
    t.mask1[g] = m.body_dofmask[b1];
    t.mask2[g] = m.body_dofmask[b2];
    t.diag[g] = m.body_invw[b1] + m.body_invw[b2];
  }
  return t;
}

constexpr int kGW = kMaxGroup / 64;
struct GMask {
  unsigned long long w[kGW];
};

// ---- small algebra ------------------------------------------------------------------------
struct Vec3 {
  double x, y, z;
};
struct Quat {
  double w, x, y, z;
};
EPA_HD Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
EPA_HD Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
EPA_HD Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
EPA_HD double Dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
EPA_HD Vec3 Cross(Vec3 a, Vec3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
EPA_HD Quat QMul(Quat a, Quat b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
          a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
EPA_HD Quat QNormalize(Quat q) {
  const double inv = 1.0 / ::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
EPA_HD void QMat(Quat q, double* M) {  // row-major 3x3
  const double w = q.w, x = q.x, y = q.y, z = q.z;
  M[0] = w * w + x * x - y * y - z * z;
  M[4] = w * w - x * x + y * y - z * z;
  M[8] = w * w - x * x - y * y + z * z;
  M[1] = 2 * (x * y - w * z);
  M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z);
  M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y);
  M[7] = 2 * (y * z + w * x);
}
EPA_HD Vec3 MulV(const double* M, Vec3 v) {
  return {M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z,
          M[6] * v.x + M[7] * v.y + M[8] * v.z};
}
// spatial inertia (10-vector about the c-frame origin) times a motion vector [ang; lin]
EPA_HD void MulInertVec(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
EPA_HD double Dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
EPA_HD void CrossMotion(double* r, const double* vel, const double* v) {  // mju_crossMotion
  r[0] = vel[1] * v[2] - vel[2] * v[1];
  r[1] = vel[2] * v[0] - vel[0] * v[2];
  r[2] = vel[0] * v[1] - vel[1] * v[0];
  r[3] = vel[1] * v[5] - vel[2] * v[4] + vel[4] * v[2] - vel[5] * v[1];
  r[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  r[5] = vel[0] * v[4] - vel[1] * v[3] + vel[3] * v[1] - vel[4] * v[0];
}
EPA_HD void CrossForce(double* r, const double* vel, const double* f) {  // mju_crossForce
  r[0] = vel[1] * f[2] - vel[2] * f[1] + vel[4] * f[5] - vel[5] * f[4];
  r[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  r[2] = vel[0] * f[1] - vel[1] * f[0] + vel[3] * f[4] - vel[4] * f[3];
  r[3] = vel[1] * f[5] - vel[2] * f[4];
  r[4] = vel[2] * f[3] - vel[0] * f[5];
  r[5] = vel[0] * f[4] - vel[1] * f[3];
}
EPA_HD double Sel(bool c, double a, double b) { return c ? a : b; }
EPA_HD double Clamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
EPA_HD int Ctz64(unsigned long long x) {
  return __builtin_ctzll(x);
}

constexpr double kMinVal = 1e-15;

// Keeps the instruction scheduler from hoisting a whole unrolled stage's loads in front of its
// arithmetic (hundreds of live VGPRs -> spills): nothing moves across this point.
#if defined(__HIP_DEVICE_COMPILE__)
#define EPA_TREE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define EPA_TREE_FENCE() ((void)0)
#endif

// MP::kM is the constexpr TreeModel
template <class MP>
struct Tree {
  static constexpr TreeModel kM = MP::kM;
  static constexpr Layout kL = MakeLayout(MP::kM);
  static constexpr int NB = kM.nbody, NV = kM.nv, NQ = kM.nq, NJ = kM.njnt, NG = kM.ngeom,
                       NU = kM.nu;
  static constexpr int kNLimit = kM.nlimit, kNFloor = kM.nfloor, kNPair = kM.npair;
  static constexpr int kNGroup = kNLimit + kNFloor + kNPair;
  static constexpr int kNRow = kNLimit + 4 * kNFloor + kNPair;
  static_assert(kNGroup <= kMaxGroup, "raise kMaxGroup");
  static constexpr int MIdx(int i, int j) { return kL.M + i * NV + j; }  // i >= j

  // f(IC<j>) for j = I, parent(I), ... (the dofs on the path from dof I to the root)
  template <int I, typename F>
  static EPA_HD void Chain(F&& f) {
    if constexpr (I >= 0) {
      f(IC<I>{});
      Chain<MP::kM.dof_parent[I]>(static_cast<F&&>(f));
    }
  }

  // Every stage below has the shape  [issue all loads] -> FENCE -> [arithmetic in VGPRs] ->
  // [stores]:  a wave runs alone on its SIMD (512 VGPRs), so a dependent HBM round trip costs
  // ~1 us with nothing to overlap it; batching turns the ~5000 dependent round trips of a naive
  // body-by-body pipeline into ~100 per forward pass.

  // ---- mj_kinematics: depth-first over the tree, the parent's frame stays in registers -------
  struct Frame {
    Vec3 pos;
    Quat q;
    double R[9];
  };
  template <int B>
  static EPA_HD void KinBody(Ws w, const Frame& par, const double* qp, Vec3& msum) {
    constexpr TreeModel m = MP::kM;
    constexpr int ja = m.body_jntadr[B], jn = m.body_jntnum[B];
    Frame f;
    if constexpr (jn == 1 && m.jnt_type[ja < 0 ? 0 : ja] == kJntFree) {
      constexpr int qa = m.jnt_qadr[ja];
      f.pos = {qp[qa], qp[qa + 1], qp[qa + 2]};
      // mj_kinematics (MuJoCo >= 3.1.4) normalises a local copy; qpos itself is left alone
      f.q = QNormalize({qp[qa + 3], qp[qa + 4], qp[qa + 5], qp[qa + 6]});
    } else {
      f.pos = par.pos + MulV(par.R, Vec3{m.body_pos[B][0], m.body_pos[B][1], m.body_pos[B][2]});
      f.q = QMul(par.q, Quat{m.body_quat[B][0], m.body_quat[B][1], m.body_quat[B][2],
                             m.body_quat[B][3]});
      static_for<0, jn>([&](auto jc) {
        constexpr int j = ja + decltype(jc)::value;
        static_assert(m.jnt_type[j] == kJntHinge, "free root + hinges only");
        QMat(f.q, f.R);
        const Vec3 jp = {m.jnt_pos[j][0], m.jnt_pos[j][1], m.jnt_pos[j][2]};
        const Vec3 anchor = MulV(f.R, jp) + f.pos;
        const Vec3 axis = MulV(f.R, Vec3{m.jnt_axis[j][0], m.jnt_axis[j][1], m.jnt_axis[j][2]});
        w(kL.anchor + 3 * j) = anchor.x;
        w(kL.anchor + 3 * j + 1) = anchor.y;
        w(kL.anchor + 3 * j + 2) = anchor.z;
        w(kL.axis + 3 * j) = axis.x;
        w(kL.axis + 3 * j + 1) = axis.y;
        w(kL.axis + 3 * j + 2) = axis.z;
        double sn, cs;
        SinCos(0.5 * (qp[m.jnt_qadr[j]] - m.qpos0[m.jnt_qadr[j]]), &sn, &cs);
        f.q = QMul(f.q, Quat{cs, m.jnt_axis[j][0] * sn, m.jnt_axis[j][1] * sn, m.jnt_axis[j][2] * sn});
        QMat(f.q, f.R);
        f.pos = anchor - MulV(f.R, jp);
      });
      f.q = QNormalize(f.q);
    }
    QMat(f.q, f.R);
    w(kL.xpos + 3 * B) = f.pos.x;
    w(kL.xpos + 3 * B + 1) = f.pos.y;
    w(kL.xpos + 3 * B + 2) = f.pos.z;
    static_for<0, 9>([&](auto kc) { w(kL.xmat + 9 * B + decltype(kc)::value) = f.R[decltype(kc)::value]; });
    const Vec3 ip = f.pos + MulV(f.R, Vec3{m.body_ipos[B][0], m.body_ipos[B][1], m.body_ipos[B][2]});
    w(kL.xipos + 3 * B) = ip.x;
    w(kL.xipos + 3 * B + 1) = ip.y;
    w(kL.xipos + 3 * B + 2) = ip.z;
    msum = msum + ip * m.body_mass[B];
    // geoms of this body: centre and (capsules) axis
    static_for<1, NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (m.geom_body[g] == B) {
        const Vec3 gp = f.pos + MulV(f.R, Vec3{m.geom_pos[g][0], m.geom_pos[g][1], m.geom_pos[g][2]});
        w(kL.gpos + 3 * g) = gp.x;
        w(kL.gpos + 3 * g + 1) = gp.y;
        w(kL.gpos + 3 * g + 2) = gp.z;
        if constexpr (m.geom_type[g] == kGeomCapsule) {
          const Vec3 ga = MulV(f.R, Vec3{m.geom_axis[g][0], m.geom_axis[g][1], m.geom_axis[g][2]});
          w(kL.gaxis + 3 * g) = ga.x;
          w(kL.gaxis + 3 * g + 1) = ga.y;
          w(kL.gaxis + 3 * g + 2) = ga.z;
        }
      }
    });
    static_for<B + 1, NB>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if constexpr (m.body_parent[c] == B) KinBody<c>(w, f, qp, msum);
    });
  }
  static EPA_HD void Kinematics(Ws w) {
    constexpr TreeModel m = MP::kM;
    double qp[NQ];
    static_for<0, NQ>([&](auto ic) { qp[decltype(ic)::value] = w(kL.qpos + decltype(ic)::value); });
    EPA_TREE_FENCE();
    Frame world;
    world.pos = {0, 0, 0};
    world.q = {1, 0, 0, 0};
    QMat(world.q, world.R);
    Vec3 msum = {0, 0, 0};
    KinBody<1>(w, world, qp, msum);
    // mj_comPos, first half: system COM = origin of the c-frame (single tree)
    msum = msum * (1.0 / m.total_mass);
    w(kL.com) = msum.x;
    w(kL.com + 1) = msum.y;
    w(kL.com + 2) = msum.z;
  }

  // ---- mj_comPos (cinert, cdof) + the composite inertias of mj_crb ----------------------------
  template <int B0, int B1>
  static EPA_HD void CinertBatch(Ws w, Vec3 c, double (*ci)[10]) {
    constexpr TreeModel m = MP::kM;
    double R[B1 - B0][9], xi[B1 - B0][3];
    static_for<B0, B1>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      static_for<0, 9>([&](auto kc) { R[b - B0][decltype(kc)::value] = w(kL.xmat + 9 * b + decltype(kc)::value); });
      static_for<0, 3>([&](auto kc) { xi[b - B0][decltype(kc)::value] = w(kL.xipos + 3 * b + decltype(kc)::value); });
    });
    EPA_TREE_FENCE();
    static_for<B0, B1>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      const double* Rb = R[b - B0];
      // Iw = R I R^T, I symmetric (xx yy zz xy xz yz)
      constexpr double Ixx = m.body_inertia[b][0], Iyy = m.body_inertia[b][1],
                       Izz = m.body_inertia[b][2], Ixy = m.body_inertia[b][3],
                       Ixz = m.body_inertia[b][4], Iyz = m.body_inertia[b][5];
      double RI[9];
      static_for<0, 3>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        RI[3 * r + 0] = Rb[3 * r] * Ixx + Rb[3 * r + 1] * Ixy + Rb[3 * r + 2] * Ixz;
        RI[3 * r + 1] = Rb[3 * r] * Ixy + Rb[3 * r + 1] * Iyy + Rb[3 * r + 2] * Iyz;
        RI[3 * r + 2] = Rb[3 * r] * Ixz + Rb[3 * r + 1] * Iyz + Rb[3 * r + 2] * Izz;
      });
      auto iw = [&](int r, int cidx) {
        return RI[3 * r] * Rb[3 * cidx] + RI[3 * r + 1] * Rb[3 * cidx + 1] + RI[3 * r + 2] * Rb[3 * cidx + 2];
      };
      const Vec3 off = Vec3{xi[b - B0][0], xi[b - B0][1], xi[b - B0][2]} - c;
      constexpr double mass = m.body_mass[b];
      const double o2 = Dot(off, off);
      ci[b][0] = iw(0, 0) + mass * (o2 - off.x * off.x);
      ci[b][1] = iw(1, 1) + mass * (o2 - off.y * off.y);
      ci[b][2] = iw(2, 2) + mass * (o2 - off.z * off.z);
      ci[b][3] = iw(0, 1) - mass * off.x * off.y;
      ci[b][4] = iw(0, 2) - mass * off.x * off.z;
      ci[b][5] = iw(1, 2) - mass * off.y * off.z;
      ci[b][6] = mass * off.x;
      ci[b][7] = mass * off.y;
      ci[b][8] = mass * off.z;
      ci[b][9] = mass;
      static_for<0, 10>([&](auto kc) { w(kL.cinert + 10 * b + decltype(kc)::value) = ci[b][decltype(kc)::value]; });
    });
  }
  static EPA_HD void ComPos(Ws w) {
    constexpr TreeModel m = MP::kM;
    const Vec3 c = {w(kL.com), w(kL.com + 1), w(kL.com + 2)};
    {  // cdof from the joint anchors / axes
      double an[NJ][3], ax[NJ][3], rp[3], rR[9];
      static_for<0, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (m.jnt_type[j] == kJntHinge) {
          static_for<0, 3>([&](auto kc) {
            an[j][decltype(kc)::value] = w(kL.anchor + 3 * j + decltype(kc)::value);
            ax[j][decltype(kc)::value] = w(kL.axis + 3 * j + decltype(kc)::value);
          });
        } else {
          constexpr int b = m.jnt_body[j];
          static_for<0, 3>([&](auto kc) { rp[decltype(kc)::value] = w(kL.xpos + 3 * b + decltype(kc)::value); });
          static_for<0, 9>([&](auto kc) { rR[decltype(kc)::value] = w(kL.xmat + 9 * b + decltype(kc)::value); });
        }
      });
      EPA_TREE_FENCE();
      static_for<0, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int a = m.jnt_dadr[j];
        if constexpr (m.jnt_type[j] == kJntHinge) {
          const Vec3 axis = {ax[j][0], ax[j][1], ax[j][2]};
          const Vec3 lin = Cross(axis, c - Vec3{an[j][0], an[j][1], an[j][2]});
          w(kL.cdof + 6 * a) = axis.x;
          w(kL.cdof + 6 * a + 1) = axis.y;
          w(kL.cdof + 6 * a + 2) = axis.z;
          w(kL.cdof + 6 * a + 3) = lin.x;
          w(kL.cdof + 6 * a + 4) = lin.y;
          w(kL.cdof + 6 * a + 5) = lin.z;
        } else {  // free: 3 world translations, then rotations about the body axes
          const Vec3 off = c - Vec3{rp[0], rp[1], rp[2]};
          static_for<0, 3>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            static_for<0, 6>([&](auto rc) {
              w(kL.cdof + 6 * (a + k) + decltype(rc)::value) = decltype(rc)::value == 3 + k ? 1.0 : 0.0;
            });
            const Vec3 axis = {rR[k], rR[3 + k], rR[6 + k]};
            const Vec3 lin = Cross(axis, off);
            w(kL.cdof + 6 * (a + 3 + k)) = axis.x;
            w(kL.cdof + 6 * (a + 3 + k) + 1) = axis.y;
            w(kL.cdof + 6 * (a + 3 + k) + 2) = axis.z;
            w(kL.cdof + 6 * (a + 3 + k) + 3) = lin.x;
            w(kL.cdof + 6 * (a + 3 + k) + 4) = lin.y;
            w(kL.cdof + 6 * (a + 3 + k) + 5) = lin.z;
          });
        }
      });
    }
    EPA_TREE_FENCE();
    // cinert in two register batches, then the composite inertias (mj_crb, backward pass)
    double ci[NB][10];
    constexpr int kHalf = (NB + 1) / 2;
    CinertBatch<1, kHalf>(w, c, ci);
    EPA_TREE_FENCE();
    CinertBatch<kHalf, NB>(w, c, ci);
    static_for_down<NB, 2>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      constexpr int p = m.body_parent[b];
      if constexpr (p > 0) {
        static_for<0, 10>([&](auto kc) { ci[p][decltype(kc)::value] += ci[b][decltype(kc)::value]; });
      }
    });
    static_for<1, NB>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      static_for<0, 10>([&](auto kc) { w(kL.crb + 10 * b + decltype(kc)::value) = ci[b][decltype(kc)::value]; });
    });
  }

  // ---- mj_crb (M, + armature), then mj_factorM, all of M in registers ----------------------------
  template <int B0, int B1>
  static EPA_HD void MassBatch(Ws w, const double (*cd)[6]) {
    constexpr TreeModel m = MP::kM;
    double in[B1 - B0][10];
    static_for<B0, B1>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      static_for<0, 10>([&](auto kc) { in[b - B0][decltype(kc)::value] = w(kL.crb + 10 * b + decltype(kc)::value); });
    });
    EPA_TREE_FENCE();
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int b = m.dof_body[i];
      if constexpr (b >= B0 && b < B1) {
        double buf[6];
        MulInertVec(buf, in[b - B0], cd[i]);
        Chain<i>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          double v = Dot6(cd[j], buf);
          if constexpr (i == j) v += m.dof_arm[i];
          w(MIdx(i, j)) = v;
        });
      }
    });
  }
  static EPA_HD void CrbFactor(Ws w) {
    constexpr TreeModel m = MP::kM;
    {
      double cd[NV][6];
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<0, 6>([&](auto rc) { cd[i][decltype(rc)::value] = w(kL.cdof + 6 * i + decltype(rc)::value); });
      });
      constexpr int kHalf = (NB + 1) / 2;
      MassBatch<1, kHalf>(w, cd);
      EPA_TREE_FENCE();
      MassBatch<kHalf, NB>(w, cd);
    }
    EPA_TREE_FENCE();
    // L'DL: M = L' D L, L unit lower with the tree's sparsity (MuJoCo mj_factorM)
    double L[NV * NV];  // only the tree-sparse entries are ever touched (static indices)
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      Chain<i>([&](auto jc) { L[i * NV + decltype(jc)::value] = w(MIdx(i, decltype(jc)::value)); });
    });
    EPA_TREE_FENCE();
    static_for_down<NV, 0>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const double inv = 1.0 / L[k * NV + k];
      w(kL.dinv + k) = inv;
      Chain<m.dof_parent[k]>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const double tmp = L[k * NV + i] * inv;
        Chain<i>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          L[i * NV + j] -= tmp * L[k * NV + j];
        });
        L[k * NV + i] = tmp;
        w(MIdx(k, i)) = tmp;
      });
    });
  }

  // x <- M^-1 x (mj_solveM), x in registers; the factor is loaded in one batch.  Returns
  // x0' M^-1 x0 = sum_i y_i^2 / D_i with y = L^-T x0 (the diagonal of A for a constraint row).
  static EPA_HD double SolveM(Ws w, double* x) {
    constexpr TreeModel m = MP::kM;
    double L[NV * NV], dinv[NV];
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      dinv[i] = w(kL.dinv + i);
      Chain<m.dof_parent[i]>([&](auto jc) { L[i * NV + decltype(jc)::value] = w(MIdx(i, decltype(jc)::value)); });
    });
    EPA_TREE_FENCE();
    static_for_down<NV, 0>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      Chain<m.dof_parent[i]>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        x[j] -= L[i * NV + j] * x[i];
      });
    });
    double quad = 0.0;
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const double y = x[i];
      x[i] = y * dinv[i];
      quad += y * x[i];
    });
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      Chain<m.dof_parent[i]>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        x[i] -= L[i * NV + j] * x[j];
      });
    });
    return quad;
  }

  // ---- mj_fwdVelocity + mj_fwdActuation + mj_fwdAcceleration ---------------------------------
  // down the tree: cvel and the bias acceleration cacc (parent values in registers)
  template <int B>
  static EPA_HD void VelBody(Ws w, const double (*cd)[6], const double* qv, const double* pvel,
                             const double* pacc) {
    constexpr TreeModel m = MP::kM;
    constexpr int a = m.body_dofadr[B], n = m.body_dofnum[B];
    double cvel[6], cacc[6];
    static_for<0, 6>([&](auto rc) {
      cvel[decltype(rc)::value] = pvel[decltype(rc)::value];
      cacc[decltype(rc)::value] = pacc[decltype(rc)::value];
    });
    if constexpr (n == 6) {  // free joint: translations first, their cdof_dot is zero
      static_for<0, 3>([&](auto kc) { cvel[3 + decltype(kc)::value] += qv[a + decltype(kc)::value]; });
      double dot[6];
      static_for<0, 3>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        CrossMotion(dot, cvel, cd[a + 3 + k]);  // all three with the velocity before the rotations
        static_for<0, 6>([&](auto rc) { cacc[decltype(rc)::value] += dot[decltype(rc)::value] * qv[a + 3 + k]; });
      });
      static_for<0, 3>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        static_for<0, 6>([&](auto rc) { cvel[decltype(rc)::value] += cd[a + 3 + k][decltype(rc)::value] * qv[a + 3 + k]; });
      });
    } else {
      static_for<0, n>([&](auto jc) {
        constexpr int i = a + decltype(jc)::value;
        double dot[6];
        CrossMotion(dot, cvel, cd[i]);
        static_for<0, 6>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          cacc[r] += dot[r] * qv[i];
          cvel[r] += cd[i][r] * qv[i];
        });
      });
    }
    static_for<0, 6>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      w(kL.cvel + 6 * B + r) = cvel[r];
      w(kL.cacc + 6 * B + r) = cacc[r];
    });
    static_for<B + 1, NB>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if constexpr (m.body_parent[c] == B) VelBody<c>(w, cd, qv, cvel, cacc);
    });
  }
  // cfrc = I cacc + cvel x* (I cvel) for a batch of bodies
  template <int B0, int B1>
  static EPA_HD void RneBatch(Ws w, double (*cf)[6]) {
    double in[B1 - B0][10], cv[B1 - B0][6], ca[B1 - B0][6];
    static_for<B0, B1>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      static_for<0, 10>([&](auto kc) { in[b - B0][decltype(kc)::value] = w(kL.cinert + 10 * b + decltype(kc)::value); });
      static_for<0, 6>([&](auto kc) {
        cv[b - B0][decltype(kc)::value] = w(kL.cvel + 6 * b + decltype(kc)::value);
        ca[b - B0][decltype(kc)::value] = w(kL.cacc + 6 * b + decltype(kc)::value);
      });
    });
    EPA_TREE_FENCE();
    static_for<B0, B1>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      double t1[6], t2[6], t3[6];
      MulInertVec(t1, in[b - B0], ca[b - B0]);
      MulInertVec(t2, in[b - B0], cv[b - B0]);
      CrossForce(t3, cv[b - B0], t2);
      static_for<0, 6>([&](auto rc) { cf[b][decltype(rc)::value] = t1[decltype(rc)::value] + t3[decltype(rc)::value]; });
    });
  }
  static EPA_HD void Velocity(Ws w) {
    constexpr TreeModel m = MP::kM;
    {
      double cd[NV][6], qv[NV];
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        qv[i] = w(kL.qvel + i);
        static_for<0, 6>([&](auto rc) { cd[i][decltype(rc)::value] = w(kL.cdof + 6 * i + decltype(rc)::value); });
      });
      EPA_TREE_FENCE();
      const double zero[6] = {0, 0, 0, 0, 0, 0};
      const double grav[6] = {0, 0, 0, 0, 0, m.gravity};  // cacc[world] = -gravity
      static_for<0, 6>([&](auto rc) { w(kL.cvel + decltype(rc)::value) = 0.0; });
      VelBody<1>(w, cd, qv, zero, grav);
    }
    EPA_TREE_FENCE();
    double cf[NB][6];
    constexpr int kHalf = (NB + 1) / 2;
    RneBatch<1, kHalf>(w, cf);
    EPA_TREE_FENCE();
    RneBatch<kHalf, NB>(w, cf);
    static_for_down<NB, 2>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      constexpr int p = m.body_parent[b];
      if constexpr (p > 0) {
        static_for<0, 6>([&](auto rc) { cf[p][decltype(rc)::value] += cf[b][decltype(rc)::value]; });
      }
    });
    EPA_TREE_FENCE();
    double x[NV];
    {
      double cd[NV][6], qv[NV], qs[NJ], ct[NU];
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        qv[i] = w(kL.qvel + i);
        static_for<0, 6>([&](auto rc) { cd[i][decltype(rc)::value] = w(kL.cdof + 6 * i + decltype(rc)::value); });
      });
      static_for<1, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (m.jnt_stiff[j] != 0.0) qs[j] = w(kL.qpos + m.jnt_qadr[j]);
      });
      static_for<0, NU>([&](auto uc) { ct[decltype(uc)::value] = w(kL.ctrl + decltype(uc)::value); });
      EPA_TREE_FENCE();
      double act[NV];
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        x[i] = -Dot6(cd[i], cf[m.dof_body[i]]) - m.dof_damp[i] * qv[i];  // -bias + damper
        act[i] = 0.0;
      });
      static_for<1, NJ>([&](auto jc) {  // joint springs (hinges; joint 0 is the free root)
        constexpr int j = decltype(jc)::value;
        if constexpr (m.jnt_stiff[j] != 0.0) {
          x[m.jnt_dadr[j]] -= m.jnt_stiff[j] * (qs[j] - m.qpos0[m.jnt_qadr[j]]);
        }
      });
      static_for<0, NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        act[m.act_dof[u]] += m.act_gear[u] * Clamp(ct[u], m.ctrl_lo, m.ctrl_hi);
      });
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        x[i] += act[i];
        w(kL.act + i) = act[i];
      });
    }
    EPA_TREE_FENCE();
    SolveM(w, x);
    static_for<0, NV>([&](auto ic) { w(kL.accs + decltype(ic)::value) = x[decltype(ic)::value]; });
  }

  // ---- mj_collision + joint-limit detection (phase A) ---------------------------------------
  // All candidates are static: unrolled tests on geoms held in registers (one batch of loads);
  // the narrow phase of a pair only runs if some lane passes the bounding-sphere cull.
  // Results per group in the workspace; `act` = this lane's active groups, `uni` = the
  // wave-uniform union.
  static EPA_HD void SetBit(GMask& act, GMask& uni, int g, bool on) {
    const unsigned long long bit = 1ull << (g & 63);
    const bool wany = WaveAny(on);
    static_for<0, kGW>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const bool here = (g >> 6) == k;
      act.w[k] |= (here && on) ? bit : 0ull;
      uni.w[k] |= (here && wany) ? bit : 0ull;
    });
  }
  // sphere-sphere: returns dist; n from 1 to 2; pos midway
  static EPA_HD double SphereSphere(Vec3 p1, double r1, Vec3 p2, double r2, Vec3* n, Vec3* pos) {
    const Vec3 dif = p2 - p1;
    const double cd = ::sqrt(Dot(dif, dif));
    const bool far = cd >= kMinVal;
    const double inv = 1.0 / Sel(far, cd, 1.0);
    *n = {Sel(far, dif.x * inv, 1.0), Sel(far, dif.y * inv, 0.0), Sel(far, dif.z * inv, 0.0)};
    const double dist = cd - r1 - r2;
    *pos = p1 + *n * (r1 + 0.5 * dist);
    return dist;
  }
  static EPA_HD void Detect(Ws w, GMask& act, GMask& uni) {
    constexpr TreeModel m = MP::kM;
    static_for<0, kGW>([&](auto kc) { act.w[decltype(kc)::value] = uni.w[decltype(kc)::value] = 0ull; });
    double ql[kNLimit > 0 ? kNLimit : 1], gp[NG][3], ga[NG][3];
    static_for<0, kNLimit>([&](auto gc) {
      ql[decltype(gc)::value] = w(kL.qpos + m.jnt_qadr[m.limit_jnt[decltype(gc)::value]]);
    });
    static_for<1, NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      static_for<0, 3>([&](auto rc) {
        gp[g][decltype(rc)::value] = w(kL.gpos + 3 * g + decltype(rc)::value);
        if constexpr (m.geom_type[g] == kGeomCapsule) ga[g][decltype(rc)::value] = w(kL.gaxis + 3 * g + decltype(rc)::value);
      });
    });
    EPA_TREE_FENCE();
    // joint limits (mj_instantiateLimit): dist = q - lo (J = +1) or hi - q (J = -1); margin 0
    static_for<0, kNLimit>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int j = m.limit_jnt[g];
      const double dlo = ql[g] - m.jnt_lo[j], dhi = m.jnt_hi[j] - ql[g];
      const bool lo = dlo < 0.0, on = lo || dhi < 0.0;
      SetBit(act, uni, g, on);
      if (WaveAny(on)) {
        w(kL.limd + g) = Sel(lo, dlo, dhi);
        w(kL.lims + g) = Sel(lo, 1.0, -1.0);
      }
    });
    // floor (plane z = 0, normal +z): mjc_PlaneSphere on spheres and capsule end spheres
    static_for<0, kNFloor>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int g = m.floor_geom[c];
      constexpr double s = m.floor_sign[c] * m.geom_hl[g];
      double cx = gp[g][0], cy = gp[g][1], cz = gp[g][2];
      if constexpr (m.geom_type[g] == kGeomCapsule) {
        cx += s * ga[g][0];
        cy += s * ga[g][1];
        cz += s * ga[g][2];
      }
      const double dist = cz - m.geom_rad[g];
      const bool on = dist < m.margin;
      SetBit(act, uni, kNLimit + c, on);
      if (WaveAny(on)) {
        w(kL.condist + c) = dist;
        w(kL.conpos + 3 * c) = cx;
        w(kL.conpos + 3 * c + 1) = cy;
        w(kL.conpos + 3 * c + 2) = 0.5 * dist;  // centre - n (r + dist / 2)
      }
    });
    // geom pairs: sphere / capsule primitives (mjraw_SphereSphere / SphereCapsule / CapsuleCapsule)
    static_for<0, kNPair>([&](auto pc) {
      constexpr int p = decltype(pc)::value;
      constexpr int g1 = m.pair_g1[p], g2 = m.pair_g2[p];
      constexpr int c = kNFloor + p;
      constexpr double r1 = m.geom_rad[g1], r2 = m.geom_rad[g2];
      constexpr double h1 = m.geom_hl[g1], h2 = m.geom_hl[g2];
      const Vec3 p1 = {gp[g1][0], gp[g1][1], gp[g1][2]}, p2 = {gp[g2][0], gp[g2][1], gp[g2][2]};
      // wave-level cull on bounding spheres
      constexpr double bound = r1 + h1 + r2 + h2 + m.margin;
      const Vec3 dc = p2 - p1;
      const bool near = Dot(dc, dc) < bound * bound;
      if (WaveAny(near)) {
        Vec3 q1 = p1, q2 = p2;
        if constexpr (m.geom_type[g1] == kGeomCapsule) {  // both capsules
          const Vec3 a1 = Vec3{ga[g1][0], ga[g1][1], ga[g1][2]} * h1;
          const Vec3 a2 = Vec3{ga[g2][0], ga[g2][1], ga[g2][2]} * h2;
          const Vec3 dif = p1 - p2;
          const double ma = Dot(a1, a1), mb = -Dot(a1, a2), mc = Dot(a2, a2);
          const double u = -Dot(a1, dif), v = Dot(a2, dif);
          const double det = ma * mc - mb * mb;
          const bool reg = fabs(det) >= kMinVal;
          const double idet = 1.0 / Sel(reg, det, 1.0);
          double x1 = (mc * u - mb * v) * idet, x2 = (ma * v - mb * u) * idet;
          {  // clamp x1, recompute x2; then clamp x2, recompute x1
            const bool hi1 = x1 > 1.0, lo1 = x1 < -1.0;
            x2 = Sel(hi1, (v - mb) / mc, Sel(lo1, (v + mb) / mc, x2));
            x1 = Sel(hi1, 1.0, Sel(lo1, -1.0, x1));
            const bool hi2 = x2 > 1.0, lo2 = x2 < -1.0;
            const double y1 = Clamp(Sel(hi2, (u - mb) / ma, (u + mb) / ma), -1.0, 1.0);
            x1 = Sel(hi2 || lo2, y1, x1);
            x2 = Sel(hi2, 1.0, Sel(lo2, -1.0, x2));
          }
          {  // exactly parallel axes: midpoint of the overlap (see oracle/mjcpu/engine.c)
            const double lo = fmax(-1.0, (u - fabs(mb)) / ma), hi = fmin(1.0, (u + fabs(mb)) / ma);
            const double px1 = Sel(lo <= hi, 0.5 * (lo + hi), Sel(lo > 1.0, 1.0, -1.0));
            const double px2 = Clamp((v - mb * px1) / mc, -1.0, 1.0);
            x1 = Sel(reg, x1, px1);
            x2 = Sel(reg, x2, px2);
          }
          q1 = p1 + a1 * x1;
          q2 = p2 + a2 * x2;
        } else if constexpr (m.geom_type[g2] == kGeomCapsule) {  // sphere - capsule
          const Vec3 ax = {ga[g2][0], ga[g2][1], ga[g2][2]};
          q2 = p2 + ax * Clamp(Dot(ax, p1 - p2), -h2, h2);
        }
        Vec3 n, pos;
        const double dist = SphereSphere(q1, r1, q2, r2, &n, &pos);
        const bool on = near && dist < m.margin;
        SetBit(act, uni, kNLimit + c, on);
        if (WaveAny(on)) {
          w(kL.condist + c) = dist;
          w(kL.conpos + 3 * c) = pos.x;
          w(kL.conpos + 3 * c + 1) = pos.y;
          w(kL.conpos + 3 * c + 2) = pos.z;
          w(kL.connrm + 3 * c) = n.x;
          w(kL.connrm + 3 * c + 1) = n.y;
          w(kL.connrm + 3 * c + 2) = n.z;
        }
      }
    });
  }

  // ---- per-lane compaction of the constraint rows ------------------------------------------------
  // Which groups are active differs from lane to lane; visiting the wave's UNION of static rows
  // would cost every lane the work of all 64 (the union is 2-3x a lane's own set).  Rows are
  // therefore built into COMPACT slots: in iteration t of a phase (limits, floor contacts, geom
  // pairs) every lane takes ITS t-th active group of that phase, so the wave needs
  // max-over-lanes iterations.  Each lane still sees its own rows in MuJoCo's order (limits by
  // joint, contacts by body pair), which is what the order-dependent PGS sweep requires.  A lane
  // that has run out of groups writes an inert row (J = W = 0, f = 0).
  struct RowCount {
    int nl, nf, np;  // compact limit rows, floor contacts (4 rows each), pair rows: wave-uniform
    EPA_HD int rows() const { return nl + 4 * nf + np; }
  };
  EPA_HD static double Gather(Ws w, int slot_lane) {  // slot differs per lane
    return w.base[(size_t)slot_lane * kLaneStride + w.lane];
  }
  EPA_HD static double& GatherRef(Ws w, int slot_lane) {
    return w.base[(size_t)slot_lane * kLaneStride + w.lane];
  }
  // bits of the groups [lo, hi) inside mask word `wi`
  static EPA_HD unsigned long long RangeBits(int wi, int lo, int hi) {
    const int a = lo - 64 * wi, b = hi - 64 * wi;
    if (b <= 0 || a >= 64) return 0ull;
    const unsigned long long upto = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
    const unsigned long long from = a <= 0 ? ~0ull : (~0ull << a);
    return upto & from;
  }
  static EPA_HD unsigned long long WordOf(const GMask& act, int wi) {
    unsigned long long word = 0ull;
    static_for<0, kGW>([&](auto kc) { word = wi == decltype(kc)::value ? act.w[decltype(kc)::value] : word; });
    return word;
  }

  static EPA_HD double Impedance(double x_abs) {  // solimp (d0, dmax, width, 0.5, 2)
    constexpr TreeModel m = MP::kM;
    const double x = x_abs * (1.0 / m.sol_width);
    const double y = Sel(x <= 0.5, 2.0 * x * x, 1.0 - 2.0 * (1.0 - x) * (1.0 - x));
    return Sel(x >= 1.0, m.sol_dmax, m.sol_d0 + y * (m.sol_dmax - m.sol_d0));
  }

  // ---- mj_makeConstraint + mj_projectConstraint ------------------------------------------------
  // Pass 1 builds the Jacobian rows from cdof held in registers, with aref, R and the warm-start
  // force f = -D min(0, J a_warm - aref); pass 2 is one M^-1 solve per row: W_r = M^-1 J_r',
  // A_rr + R_r.  Also returns u = J' f_warm and csum = sum_r f_r (R_r f_r / 2 - aref_r).
  static EPA_HD RowCount MakeRows(Ws w0, const GMask& act, double* u, double* csum_out) {
    constexpr TreeModel m = MP::kM;
    static constexpr GroupTab gt = MakeGroupTab(MP::kM);
    RowCount rc{0, 0, 0};
    {
      double cd[NV][6], qv[NV], wm[NV];
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        qv[i] = w0(kL.qvel + i);
        wm[i] = w0(kL.warm + i);
        static_for<0, 6>([&](auto rc6) { cd[i][decltype(rc6)::value] = w0(kL.cdof + 6 * i + decltype(rc6)::value); });
      });
      const Vec3 com = {w0(kL.com), w0(kL.com + 1), w0(kL.com + 2)};
      EPA_TREE_FENCE();
      double csum = 0.0;
      static_for<0, NV>([&](auto ic) { u[decltype(ic)::value] = 0.0; });
      int row = 0;  // next compact row slot (wave-uniform)
      // phases: 0 limits, 1 floor contacts, 2 geom pairs
#pragma nounroll
      for (int phase = 0; phase < 3; ++phase) {
        const int glo = phase == 0 ? 0 : (phase == 1 ? kNLimit : kNLimit + kNFloor);
        const int ghi = phase == 0 ? kNLimit : (phase == 1 ? kNLimit + kNFloor : kNGroup);
        const int nsub = phase == 1 ? 4 : 1;
        int count = 0;
#pragma nounroll
        for (int wi = 0; wi < kGW; ++wi) {
          const unsigned long long bits = RangeBits(wi, glo, ghi);
          if (bits == 0ull) continue;
          unsigned long long rem = WordOf(act, wi) & bits;
          while (WaveAny(rem != 0ull)) {
            const Ws w = w0.Fresh();
            const bool on = rem != 0ull;
            const int g = on ? 64 * wi + Ctz64(rem) : glo;  // this lane's group (glo: inert dummy)
            rem &= rem - 1ull;
            const int c = phase == 0 ? 0 : g - kNLimit;  // contact index (floor, then pairs)
            double pos, lim_s = 0.0;
            Vec3 cpos = {0, 0, 0}, n = {0, 0, 1};
            if (phase == 0) {
              pos = Gather(w, kL.limd + g);
              lim_s = Gather(w, kL.lims + g);
            } else {
              pos = Gather(w, kL.condist + c) - m.margin;  // r = dist - includemargin
              cpos = {Gather(w, kL.conpos + 3 * c), Gather(w, kL.conpos + 3 * c + 1),
                      Gather(w, kL.conpos + 3 * c + 2)};
              if (phase == 2) {
                n = {Gather(w, kL.connrm + 3 * c), Gather(w, kL.connrm + 3 * c + 1),
                     Gather(w, kL.connrm + 3 * c + 2)};
              }
            }
            const unsigned m1 = phase == 0 ? 0u : gt.mask1[g], m2 = phase == 0 ? 0u : gt.mask2[g];
            const int ld = phase == 0 ? gt.dof[g] : -1;
            const double diag = gt.diag[g];
            EPA_TREE_FENCE();
            const Vec3 off = cpos - com;
            // impedance / regulariser of the group (mj_makeImpedance); pyramid rows share 2 mu^2 R
            const double imp = Impedance(fabs(pos));
            double R = fmax(kMinVal, (1.0 - imp) * diag / imp);
            if (phase == 1) R *= 2.0 * m.floor_mu * m.floor_mu;
            const double kimp = m.sol_K * imp * pos;
            if (phase != 0) {  // compact contact record for mj_rnePostConstraint
              const int t = phase == 1 ? count : kNFloor + count;
              w(kL.ccpos + 3 * t) = cpos.x;
              w(kL.ccpos + 3 * t + 1) = cpos.y;
              w(kL.ccpos + 3 * t + 2) = cpos.z;
              w(kL.ccnrm + 3 * t) = n.x;
              w(kL.ccnrm + 3 * t + 1) = n.y;
              w(kL.ccnrm + 3 * t + 2) = n.z;
              w(kL.ccbody + 2 * t) = (double)gt.b1[g];
              w(kL.ccbody + 2 * t + 1) = (double)gt.b2[g];
            }
#pragma nounroll
            for (int k = 0; k < nsub; ++k) {
              // row direction: n, or the pyramid edge n +- mu t (floor frame: n = z, t1 = y, t2 = -x)
              Vec3 dir = n;
              if (phase == 1) {
                dir = k == 0 ? Vec3{0, m.floor_mu, 1} : (k == 1 ? Vec3{0, -m.floor_mu, 1}
                             : (k == 2 ? Vec3{-m.floor_mu, 0, 1} : Vec3{m.floor_mu, 0, 1}));
              }
              const Vec3 mdir = Cross(off, dir);
              const int r = row + k;
              double vel = 0.0, jw = 0.0, J[NV];
              static_for<0, NV>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const double coef = (double)((int)((m2 >> i) & 1u) - (int)((m1 >> i) & 1u));
                const double jc = coef * (dir.x * cd[i][3] + dir.y * cd[i][4] + dir.z * cd[i][5] +
                                          mdir.x * cd[i][0] + mdir.y * cd[i][1] + mdir.z * cd[i][2]);
                J[i] = Sel(on, phase == 0 ? Sel(i == ld, lim_s, 0.0) : jc, 0.0);
                vel += J[i] * qv[i];
                jw += J[i] * wm[i];
                w.Pair(kL.rowJW, r * NV + i).x = J[i];
              });
              const double aref = -m.sol_B * vel - kimp;
              const double jar = jw - aref;
              const double f = (on && jar < 0.0) ? -jar / R : 0.0;
              w.Pair(kL.rowS, 3 * r).x = f;
              w.Pair(kL.rowS, 3 * r + 1).y = Sel(on, R, 0.0);
              w.Pair(kL.rowS, 3 * r + 2).x = Sel(on, aref, 0.0);
              csum += f * (0.5 * R * f - aref);
              static_for<0, NV>([&](auto ic) { u[decltype(ic)::value] += f * J[decltype(ic)::value]; });
            }
            row += nsub;
            ++count;
          }
        }
        if (phase == 0) rc.nl = count;
        if (phase == 1) rc.nf = count;
        if (phase == 2) rc.np = count;
      }
      *csum_out = csum;
    }
    EPA_TREE_FENCE();
    const int nrow = rc.rows();
#pragma nounroll
    for (int r = 0; r < nrow; ++r) {
      const Ws w = w0.Fresh();
      double x[NV];
      static_for<0, NV>([&](auto ic) { x[decltype(ic)::value] = w.Pair(kL.rowJW, r * NV + decltype(ic)::value).x; });
      const double R = w.Pair(kL.rowS, 3 * r + 1).y;
      {  // b_r = J_r qacc_smooth - aref_r, kept next to aref for the register-resident PGS
        double jb = -w.Pair(kL.rowS, 3 * r + 2).x;
        static_for<0, NV>([&](auto ic) { jb += x[decltype(ic)::value] * w(kL.accs + decltype(ic)::value); });
        w.Pair(kL.rowS, 3 * r + 2).y = jb;
      }
      EPA_TREE_FENCE();
      const double quad = SolveM(w, x);
      static_for<0, NV>([&](auto ic) { w.Pair(kL.rowJW, r * NV + decltype(ic)::value).y = x[decltype(ic)::value]; });
      const double arr = R + quad;  // 0 for an inert row
      w.Pair(kL.rowS, 3 * r + 1).x = arr;
      w.Pair(kL.rowS, 3 * r).y = arr > 0.0 ? 1.0 / arr : 0.0;
    }
    return rc;
  }

  // ---- mj_fwdConstraint with mj_solPGS ----------------------------------------------------------
  struct RowRegs {
    double J[NV], W[NV], f, arrinv, arr, R, aref;
  };
  static EPA_HD void LoadRow(Ws w, int r, RowRegs& t) {
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const D2 jw = w.Pair(kL.rowJW, r * NV + i);
      t.J[i] = jw.x;
      t.W[i] = jw.y;
    });
    const D2 s0 = w.Pair(kL.rowS, 3 * r), s1 = w.Pair(kL.rowS, 3 * r + 1);
    t.f = s0.x;
    t.arrinv = s0.y;
    t.arr = s1.x;
    t.R = s1.y;
    t.aref = w.Pair(kL.rowS, 3 * r + 2).x;
  }
  // one PGS row update (mj_solPGS, dim 1): returns the cost decrease.  An inert row (all zero)
  // yields delta = 0 by itself.
  static EPA_HD double VisitRow(Ws w, int r, const RowRegs& t, bool live, double* a) {
    double p0 = t.R * t.f - t.aref, p1 = 0.0, p2 = 0.0, p3 = 0.0;  // 4 chains: short latency
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i % 4 == 0) p0 += t.J[i] * a[i];
      if constexpr (i % 4 == 1) p1 += t.J[i] * a[i];
      if constexpr (i % 4 == 2) p2 += t.J[i] * a[i];
      if constexpr (i % 4 == 3) p3 += t.J[i] * a[i];
    });
    const double res = (p0 + p1) + (p2 + p3);
    const double fn = fmax(0.0, t.f - res * t.arrinv);
    double delta = fn - t.f;
    const double change = 0.5 * delta * delta * t.arr + delta * res;
    const bool keep = live && !(change > 1e-10);
    delta = Sel(keep, delta, 0.0);
    w.Pair(kL.rowS, 3 * r).x = t.f + delta;
    static_for<0, NV>([&](auto ic) { a[decltype(ic)::value] += delta * t.W[decltype(ic)::value]; });
    return Sel(keep, -change, 0.0);
  }
  // `u` = J' f_warm and `csum` = sum_r f_r (R_r f_r / 2 - aref_r) come from MakeRows.
  // `commit`: lanes that are only kept busy must not disturb their warm start.
  static EPA_HD void SolvePgs(Ws w0, int nrow, bool commit, double* u, double csum) {
    constexpr TreeModel m = MP::kM;
    double a[NV];
    static_for<0, NV>([&](auto ic) { a[decltype(ic)::value] = w0(kL.accs + decltype(ic)::value); });
    if (nrow > 0) {
      // dual cost of the warm-start forces: 1/2 f'(A+R)f + f'b with A f = J M^-1 J'f and
      // b = J qacc_smooth - aref; they are kept only if it is below the cost of f = 0
      double v[NV];
      double cost = csum;
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        v[i] = u[i];
        cost += u[i] * a[i];
      });
      SolveM(w0.Fresh(), v);
      static_for<0, NV>([&](auto ic) { cost += 0.5 * u[decltype(ic)::value] * v[decltype(ic)::value]; });
      const bool cold = cost > 0.0;
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        a[i] += Sel(cold, 0.0, v[i]);
      });
      if (WaveAny(cold)) {
#pragma nounroll
        for (int r = 0; r < nrow; ++r) {
          if (cold) w0.Pair(kL.rowS, 3 * r).x = 0.0;
        }
      }
      EPA_TREE_FENCE();
      const double scale = 1.0 / (m.meaninertia * (double)NV);
      bool done = false;
      for (int iter = 0; iter < m.iterations; ++iter) {
        // rows stream through a ring of kRing register buffers: a row is loaded kRing - 1 visits
        // before it is used, so the HBM latency overlaps the arithmetic of the rows in between
        constexpr int kRing = 3;
        double improvement = 0.0;
        RowRegs buf[kRing];
        // Lanes whose solve has converged skip the row traffic altogether (the one place with
        // lane-divergent control flow: per-env sweep counts are heavily skewed -- median 2, wave
        // maximum ~45 -- so late sweeps would otherwise stream 64 columns for a few live lanes).
        if (!done) {
          static_for<0, kRing - 1>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            LoadRow(w0.Fresh(), b < nrow ? b : nrow - 1, buf[b]);
          });
#pragma nounroll
          for (int r0 = 0; r0 < nrow; r0 += kRing) {
            static_for<0, kRing>([&](auto bc) {
              constexpr int b = decltype(bc)::value;
              constexpr int pre = (b + kRing - 1) % kRing;
              const int r = r0 + b;
              if (r < nrow) {
                const Ws w = w0.Fresh();
                // prefetch (always issued, clamped: the number of loads in flight is then the
                // same on every path and the wait below can leave them outstanding)
                LoadRow(w, r + kRing - 1 < nrow ? r + kRing - 1 : nrow - 1, buf[pre]);
                EPA_TREE_FENCE();
                improvement += VisitRow(w, r, buf[b], true, a);
                EPA_TREE_FENCE();
              }
            });
          }
        }
        done = done || improvement * scale < 1e-8;
        if (!WaveAny(!done)) break;
      }
      // qacc = qacc_smooth + M^-1 J' f is `a` itself (accumulated row update by row update)
    }
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      w0(kL.qacc + i) = a[i];
      if (commit) w0(kL.warm + i) = a[i];
    });
  }

  // ---- mj_solPGS, register-resident variant -----------------------------------------------------
  // When the wave's compact row count fits kRegRows, the dual problem's matrix A + R (symmetric,
  // kRegRows (kRegRows + 1) / 2 numbers per lane), b and f live in VGPRs and the sweeps touch no
  // memory at all (MuJoCo's own formulation: res_r = b_r + sum_c AR_rc f_c).  Streaming the
  // (J, W) rows instead re-reads ~50 numbers per row visit, ~45 sweeps deep for the slowest lane
  // of a wave: 10x the HBM traffic of everything else in a forward pass.
  static constexpr int kRegRows = 16;
  static EPA_HD constexpr int TriAR(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
  // A + R is split: the first kArReg entries (packed lower triangle) in VGPRs, the rest in a
  // lane-private LDS column.  All 136 in registers overflow the 256 architectural VGPRs into
  // AGPRs, and every use then costs two v_accvgpr_read: 2/3 of the sweep's instructions.
  static constexpr int kArAll = kRegRows * (kRegRows + 1) / 2;
  static constexpr int kArReg = kArAll / 2;
  static constexpr int kArLds = kArAll - kArReg;  // doubles per lane: 34 KB per wave
  struct ArStore {
    double reg[kArReg];
    double* lds;  // this lane's column: element k at lds[k * kLaneStride]
    template <int I>
    EPA_HD double Get() const {
      if constexpr (I < kArReg) return reg[I];
      else return lds[(I - kArReg) * kLaneStride];
    }
    template <int I>
    EPA_HD void Set(double v) {
      if constexpr (I < kArReg) reg[I] = v;
      else lds[(I - kArReg) * kLaneStride] = v;
    }
  };
  static EPA_HD void SolvePgsReg(Ws w0, int nrow, bool commit, double* lds_col) {
    constexpr TreeModel m = MP::kM;
    constexpr int R = kRegRows;
    ArStore AR;
    AR.lds = lds_col;
    double b[R], f[R], ainv[R];
    static_for<0, R>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      static_for<c, R>([&](auto rc) { AR.template Set<TriAR(decltype(rc)::value, c)>(0.0); });
      b[c] = f[c] = ainv[c] = 0.0;
    });
    // A_rc = J_r . W_c column by column: W_c stays in registers while the J rows below it
    // stream through (more columns at once would not fit next to the 136 numbers of A + R)
    static_for<0, R>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if (c < nrow) {  // scalar
        const Ws w = w0.Fresh();
        double Wc[NV];
        static_for<0, NV>([&](auto ic) { Wc[decltype(ic)::value] = w.Pair(kL.rowJW, c * NV + decltype(ic)::value).y; });
        const D2 s0 = w.Pair(kL.rowS, 3 * c), s1 = w.Pair(kL.rowS, 3 * c + 1);
        b[c] = w.Pair(kL.rowS, 3 * c + 2).y;
        f[c] = s0.x;
        ainv[c] = s0.y;
        AR.template Set<TriAR(c, c)>(s1.x);  // A_cc + R_c
        static_for<c + 1, R>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          if (r < nrow) {  // scalar
            double Jr[NV];
            static_for<0, NV>([&](auto ic) { Jr[decltype(ic)::value] = w.Pair(kL.rowJW, r * NV + decltype(ic)::value).x; });
            EPA_TREE_FENCE();
            double p0 = 0.0, p1 = 0.0;
            static_for<0, NV>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              if constexpr (i % 2 == 0) p0 += Jr[i] * Wc[i];
              else p1 += Jr[i] * Wc[i];
            });
            AR.template Set<TriAR(r, c)>(p0 + p1);
          }
        });
      }
    });
    EPA_TREE_FENCE();
    // dual cost of the warm-start forces 1/2 f'(A+R)f + f'b: kept only if below the cost of f = 0
    {
      double cost = 0.0;
      static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        double s = 0.0;
        static_for<0, R>([&](auto cc) { s += AR.template Get<TriAR(r, decltype(cc)::value)>() * f[decltype(cc)::value]; });
        cost += f[r] * (0.5 * s + b[r]);
      });
      const bool cold = cost > 0.0;
      static_for<0, R>([&](auto rc) { f[decltype(rc)::value] = Sel(cold, 0.0, f[decltype(rc)::value]); });
    }
    const double scale = 1.0 / (m.meaninertia * (double)NV);
    bool done = false;
    for (int iter = 0; iter < m.iterations; ++iter) {
      double improvement = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" ::: "memory");  // keep the LDS half of A + R in LDS: no hoisting out of the loop
#endif
      static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if (r < nrow) {  // scalar
          double p0 = b[r], p1 = 0.0, p2 = 0.0, p3 = 0.0;
          static_for<0, R>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const double t = AR.template Get<TriAR(r, c)>() * f[c];
            if constexpr (c % 4 == 0) p0 += t;
            if constexpr (c % 4 == 1) p1 += t;
            if constexpr (c % 4 == 2) p2 += t;
            if constexpr (c % 4 == 3) p3 += t;
          });
          const double res = (p0 + p1) + (p2 + p3);
          const double fn = fmax(0.0, f[r] - res * ainv[r]);
          double delta = fn - f[r];
          const double change = 0.5 * delta * delta * AR.template Get<TriAR(r, r)>() + delta * res;
          const bool keep = !done && !(change > 1e-10);
          f[r] += Sel(keep, delta, 0.0);
          improvement -= Sel(keep, change, 0.0);
        }
      });
      done = done || improvement * scale < 1e-8;
      if (!WaveAny(!done)) break;
    }
    EPA_TREE_FENCE();
    // dual finish: qacc = qacc_smooth + M^-1 J' f = qacc_smooth + sum_r f_r W_r
    double a[NV];
    static_for<0, NV>([&](auto ic) { a[decltype(ic)::value] = w0(kL.accs + decltype(ic)::value); });
    static_for<0, R>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      if (r < nrow) {  // scalar
        const Ws w = w0.Fresh();
        w.Pair(kL.rowS, 3 * r).x = f[r];  // efc_force, for mj_rnePostConstraint
        double Wr[NV];
        static_for<0, NV>([&](auto ic) { Wr[decltype(ic)::value] = w.Pair(kL.rowJW, r * NV + decltype(ic)::value).y; });
        EPA_TREE_FENCE();
        static_for<0, NV>([&](auto ic) { a[decltype(ic)::value] += f[r] * Wr[decltype(ic)::value]; });
      }
    });
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      w0(kL.qacc + i) = a[i];
      if (commit) w0(kL.warm + i) = a[i];
    });
  }

  // mj_forward
  // `lds_col`: this lane's column of a [kArLds][lanes] LDS block (SolvePgsReg)
  static EPA_HD RowCount Forward(Ws w, bool commit, double* lds_col) {
    // Fresh(): every stage recomputes its slot addresses locally (see Ws::Fresh)
    GMask act, uni;
    Kinematics(w.Fresh());
    EPA_TREE_FENCE();
    ComPos(w.Fresh());
    EPA_TREE_FENCE();
    CrbFactor(w.Fresh());
    EPA_TREE_FENCE();
    Detect(w.Fresh(), act, uni);
    EPA_TREE_FENCE();
    Velocity(w.Fresh());
    EPA_TREE_FENCE();
    double u[NV], csum;
    const RowCount rc = MakeRows(w.Fresh(), act, u, &csum);
    EPA_TREE_FENCE();
    if (rc.rows() <= kRegRows) {
      SolvePgsReg(w.Fresh(), rc.rows(), commit, lds_col);
    } else {
      SolvePgs(w.Fresh(), rc.rows(), commit, u, csum);
    }
    EPA_TREE_FENCE();
    return rc;
  }

  // mj_integratePos from the saved q0 with velocity slot `vel` scaled by `h`; result -> qpos
  static EPA_HD void IntegratePos(Ws w, int vel, double h, bool commit) {
    constexpr TreeModel m = MP::kM;
    static_for<0, NJ>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int qa = m.jnt_qadr[j], da = m.jnt_dadr[j];
      if constexpr (m.jnt_type[j] == kJntFree) {
        static_for<0, 3>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const double v = w(kL.x0q + qa + k) + h * w(vel + da + k);
          if (commit) w(kL.qpos + qa + k) = v;
        });
        const Vec3 om = {w(vel + da + 3), w(vel + da + 4), w(vel + da + 5)};
        const double nrm = ::sqrt(Dot(om, om));
        const bool rot = nrm * h > 0.0;
        const double inv = 1.0 / Sel(rot, nrm, 1.0);
        double sn, cs;
        SinCos(0.5 * nrm * h, &sn, &cs);
        const Quat q0 = {w(kL.x0q + qa + 3), w(kL.x0q + qa + 4), w(kL.x0q + qa + 5), w(kL.x0q + qa + 6)};
        const Quat q1 = QNormalize(QMul(q0, Quat{cs, om.x * inv * sn, om.y * inv * sn, om.z * inv * sn}));
        if (commit) {
          w(kL.qpos + qa + 3) = Sel(rot, q1.w, q0.w);
          w(kL.qpos + qa + 4) = Sel(rot, q1.x, q0.x);
          w(kL.qpos + qa + 5) = Sel(rot, q1.y, q0.y);
          w(kL.qpos + qa + 6) = Sel(rot, q1.z, q0.z);
        }
      } else {
        const double v = w(kL.x0q + qa) + h * w(vel + da);
        if (commit) w(kL.qpos + qa) = v;
      }
    });
  }

  // One stage boundary of mj_RungeKutta(4).  Called after the forward evaluation of stage
  // `stage` (0: the evaluation at the start state).  Stages 0..2 move the state to the next
  // stage point, stage 3 finishes the step.  `live`: lanes that really integrate.
  static EPA_HD void RkAdvance(Ws w, int stage, bool live) {
    constexpr TreeModel m = MP::kM;
    const double h = m.timestep;
    const double B = stage == 0 || stage == 3 ? 1.0 / 6.0 : 1.0 / 3.0;
    const double A = stage == 2 ? 1.0 : 0.5;
    if (stage == 0) {
      static_for<0, NQ>([&](auto ic) {
        if (live) w(kL.x0q + decltype(ic)::value) = w(kL.qpos + decltype(ic)::value);
      });
    }
    static_for<0, NV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const double v = w(kL.qvel + i), acc = w(kL.qacc + i);
      if (stage == 0) {
        if (live) {
          w(kL.x0v + i) = v;
          w(kL.accq + i) = B * v;
          w(kL.accv + i) = B * acc;
        }
      } else if (live) {
        w(kL.accq + i) += B * v;
        w(kL.accv + i) += B * acc;
      }
    });
    if (stage < 3) {
      // X[i+1] = X0 + h A (Xv[i], F[i]); qvel (the stage velocity) is still in place
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (live) w(kL.accs + i) = A * w(kL.qvel + i);  // scratch: dq
      });
      IntegratePos(w, kL.accs, h, live);
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (live) w(kL.qvel + i) = w(kL.x0v + i) + h * A * w(kL.qacc + i);
      });
    } else {
      static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (live) w(kL.qvel + i) = w(kL.x0v + i) + h * w(kL.accv + i);
      });
      IntegratePos(w, kL.accq, h, live);
    }
  }

  // mj_rnePostConstraint, cfrc_ext part: contact forces of the LAST forward evaluation as
  // spatial forces [torque; force] about the c-frame origin (mju_decodePyramid for the floor).
  // Walks the compact contact records MakeRows left behind; the bodies differ per lane.
  static EPA_HD void ContactWrench(Ws w, RowCount rc) {
    constexpr TreeModel m = MP::kM;
    static_for<0, 6 * NB>([&](auto kc) { w(kL.cext + decltype(kc)::value) = 0.0; });
    const Vec3 com = {w(kL.com), w(kL.com + 1), w(kL.com + 2)};
    const int ncon = rc.nf + rc.np;
#pragma nounroll
    for (int i = 0; i < ncon; ++i) {
      const bool is_floor = i < rc.nf;
      const int t = is_floor ? i : kNFloor + (i - rc.nf);
      Vec3 F;
      if (is_floor) {
        const int r = rc.nl + 4 * i;
        const double f0 = w.Pair(kL.rowS, 3 * r).x, f1 = w.Pair(kL.rowS, 3 * r + 3).x,
                     f2 = w.Pair(kL.rowS, 3 * r + 6).x, f3 = w.Pair(kL.rowS, 3 * r + 9).x;
        // frame rows n = z, t1 = y, t2 = -x
        F = {-(f2 - f3) * m.floor_mu, (f0 - f1) * m.floor_mu, f0 + f1 + f2 + f3};
      } else {
        const double f = w.Pair(kL.rowS, 3 * (rc.nl + 4 * rc.nf + (i - rc.nf))).x;
        F = Vec3{w(kL.ccnrm + 3 * t), w(kL.ccnrm + 3 * t + 1), w(kL.ccnrm + 3 * t + 2)} * f;
      }
      const Vec3 off = Vec3{w(kL.ccpos + 3 * t), w(kL.ccpos + 3 * t + 1), w(kL.ccpos + 3 * t + 2)} - com;
      const Vec3 tq = Cross(off, F);  // zero force (inert rows) => zero wrench
      const double w6[6] = {tq.x, tq.y, tq.z, F.x, F.y, F.z};
      const int b1 = (int)w(kL.ccbody + 2 * t), b2 = (int)w(kL.ccbody + 2 * t + 1);
      static_for<0, 6>([&](auto rc6) {
        constexpr int r = decltype(rc6)::value;
        GatherRef(w, kL.cext + 6 * b1 + r) -= w6[r];
        GatherRef(w, kL.cext + 6 * b2 + r) += w6[r];
      });
    }
  }
};

}  // namespace tree
}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_TREE_HIP_H_
