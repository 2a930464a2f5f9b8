// Build-time generator: runs the host model compilers (mj_cheetah_model.h,
// mj_ant_model.h) and prints their results as C++ aggregate literals, so the
// kernels see the MuJoCo model as compile-time constants (immediates / folded
// arithmetic) instead of ~150 kernel-argument scalars that have to live in —
// and spill out of — SGPRs.  Invoked by the Makefile: `gen cheetah` -> build/mj_cheetah_consts.inc,
// `gen walker` -> build/mj_walker_consts.inc, `gen ant` -> build/mj_ant_consts.inc,
// `gen humanoid` -> build/mj_humanoid_consts.inc, `gen pusher` -> build/mj_pusher_consts.inc,
// `gen chain` -> build/mj_pendulum_consts.inc (InvertedPendulum, InvertedDoublePendulum, Reacher, Swimmer).
#include <cstdio>

#include "mj_ant_model.h"
#include "mj_cheetah_model.h"
#include "mj_pendulum_model.h"
#include "mj_pusher_model.h"
#include "mj_tree_model.h"

static void Arr(const char* name, const double* v, int n) {
  std::printf("    /*%s*/ {", name);
  for (int i = 0; i < n; ++i) std::printf("%s%.17g", i ? ", " : "", v[i]);
  std::printf("},\n");
}
static void Arr2(const char* name, const double* v, int n, int m) {
  std::printf("    /*%s*/ {", name);
  for (int i = 0; i < n; ++i) {
    std::printf("%s{", i ? ", " : "");
    for (int j = 0; j < m; ++j) std::printf("%s%.17g", j ? ", " : "", v[i * m + j]);
    std::printf("}");
  }
  std::printf("},\n");
}
#define S(x) std::printf("    /*" #x "*/ %.17g,\n", m.x)
static void IArr(const char* name, const int* v, int n) {
  std::printf("    /*%s*/ {", name);
  for (int i = 0; i < n; ++i) std::printf("%s%d", i ? ", " : "", v[i]);
  std::printf("},\n");
}
static void UArr(const char* name, const unsigned* v, int n) {
  std::printf("    /*%s*/ {", name);
  for (int i = 0; i < n; ++i) std::printf("%s%uu", i ? ", " : "", v[i]);
  std::printf("},\n");
}
#define SI(x) std::printf("    /*" #x "*/ %d,\n", m.x)

// every field of tree::TreeModel, in declaration order
static void EmitTree(const char* name, const epa::mj::tree::TreeModel& m) {
  using namespace epa::mj::tree;
  std::printf("constexpr ::epa::mj::tree::TreeModel %s = {\n", name);
  SI(nbody); SI(njnt); SI(nq); SI(nv); SI(ngeom); SI(nu); SI(nlimit); SI(nfloor); SI(npair);
  SI(iterations); S(timestep); S(gravity); S(margin); S(floor_mu);
  S(sol_K); S(sol_B); S(sol_d0); S(sol_dmax); S(sol_width); S(meaninertia); S(total_mass);
  IArr("body_parent", m.body_parent, kMaxBody); IArr("body_jntadr", m.body_jntadr, kMaxBody);
  IArr("body_jntnum", m.body_jntnum, kMaxBody); IArr("body_dofadr", m.body_dofadr, kMaxBody);
  IArr("body_dofnum", m.body_dofnum, kMaxBody); UArr("body_dofmask", m.body_dofmask, kMaxBody);
  Arr2("body_pos", &m.body_pos[0][0], kMaxBody, 3); Arr2("body_quat", &m.body_quat[0][0], kMaxBody, 4);
  Arr2("body_ipos", &m.body_ipos[0][0], kMaxBody, 3); Arr("body_mass", m.body_mass, kMaxBody);
  Arr2("body_inertia", &m.body_inertia[0][0], kMaxBody, 6); Arr("body_invw", m.body_invw, kMaxBody);
  IArr("jnt_type", m.jnt_type, kMaxJnt); IArr("jnt_body", m.jnt_body, kMaxJnt);
  IArr("jnt_qadr", m.jnt_qadr, kMaxJnt); IArr("jnt_dadr", m.jnt_dadr, kMaxJnt);
  IArr("jnt_limited", m.jnt_limited, kMaxJnt);
  Arr2("jnt_pos", &m.jnt_pos[0][0], kMaxJnt, 3); Arr2("jnt_axis", &m.jnt_axis[0][0], kMaxJnt, 3);
  Arr("jnt_lo", m.jnt_lo, kMaxJnt); Arr("jnt_hi", m.jnt_hi, kMaxJnt); Arr("jnt_stiff", m.jnt_stiff, kMaxJnt);
  IArr("dof_parent", m.dof_parent, kMaxV); IArr("dof_body", m.dof_body, kMaxV);
  Arr("dof_arm", m.dof_arm, kMaxV); Arr("dof_damp", m.dof_damp, kMaxV); Arr("dof_invw", m.dof_invw, kMaxV);
  IArr("geom_type", m.geom_type, kMaxGeom); IArr("geom_body", m.geom_body, kMaxGeom);
  Arr2("geom_pos", &m.geom_pos[0][0], kMaxGeom, 3); Arr2("geom_axis", &m.geom_axis[0][0], kMaxGeom, 3);
  Arr("geom_rad", m.geom_rad, kMaxGeom); Arr("geom_hl", m.geom_hl, kMaxGeom);
  IArr("act_dof", m.act_dof, kMaxU); Arr("act_gear", m.act_gear, kMaxU); S(ctrl_lo); S(ctrl_hi);
  IArr("limit_jnt", m.limit_jnt, kMaxJnt);
  IArr("floor_geom", m.floor_geom, kMaxFloor); Arr("floor_sign", m.floor_sign, kMaxFloor);
  IArr("pair_g1", m.pair_g1, kMaxPair); IArr("pair_g2", m.pair_g2, kMaxPair);
  Arr("qpos0", m.qpos0, kMaxQ);
  std::printf("};\n");
}

template <int NL, int kBase>
static void PrintPend(const char* name, const epa::mj::pend::PendModel<double, NL, kBase>& m) {
  std::printf("constexpr ::epa::mj::pend::PendModel<double, %d, %d> %s = {\n", NL, kBase, name);
  S(cart_mass);
  Arr("mass", m.mass, NL); Arr("iyy", m.iyy, NL); Arr("cx", m.cx, NL); Arr("cz", m.cz, NL);
  Arr("lx", m.lx, NL); Arr("lz", m.lz, NL); Arr("damp", m.damp, NL + 2); Arr("arm", m.arm, NL + 2);
  S(grav_x); S(grav_z);
  Arr("gear", m.gear, NL + 2);
  S(ctrl_lo); S(ctrl_hi);
  IArr("limited", m.limited, NL + 2);
  Arr("lo", m.lo, NL + 2); Arr("hi", m.hi, NL + 2); Arr("margin", m.margin, NL + 2);
  Arr("dof_invw", m.dof_invw, NL + 2);
  S(lim_K); S(lim_B); S(lim_d0); S(lim_dmax); S(lim_width);
  S(fluid_density); S(fluid_viscosity);
  Arr2("box", &m.box[0][0], NL, 3);
  S(timestep); S(total_mass);
  std::printf("};\n");
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'h') {  // humanoid
    std::printf("// generated by gen_mj_consts.cpp -- do not edit\n");
    EmitTree("kHumanoidModelConst", epa::mj::tree::BuildHumanoidModel(false));
    EmitTree("kHumanoidStandupModelConst", epa::mj::tree::BuildHumanoidModel(true));
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'c' && argv[1][1] == 'h' && argv[1][2] == 'a') {  // chain families
    using namespace epa::mj::pend;
    std::printf("// generated by gen_mj_consts.cpp -- do not edit\n");
    PrintPend("kInvertedPendulumModelConst", BuildInvertedPendulum());
    PrintPend("kInvertedDoublePendulumModelConst", BuildInvertedDoublePendulum());
    PrintPend("kReacherModelConst", BuildReacher());
    PrintPend("kSwimmerModelConst", BuildSwimmer());
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'p') {  // pusher: pusher.xml and pusher_v5.xml
    using namespace epa::mj::pusher;
    std::printf("// generated by gen_mj_consts.cpp -- do not edit\n");
    const char* names[2] = {"kPusherModelConst", "kPusherV5ModelConst"};
    for (int v5 = 0; v5 < 2; ++v5) {
      const PusherModel<double> m = BuildPusherModel(v5 != 0);
      std::printf("constexpr ::epa::mj::pusher::PusherModel<double> %s = {\n", names[v5]);
      Arr2("off", &m.off[0][0], kNL, 3); Arr("mass", m.mass, kNL); Arr2("com", &m.com[0][0], kNL, 3);
      Arr2("inertia", &m.inertia[0][0], kNL, 6);
      Arr("lo", m.lo, kNL); Arr("hi", m.hi, kNL); Arr("damp", m.damp, kNV); Arr("arm", m.arm, kNV);
      Arr("dof_invw", m.dof_invw, kNL);
      Arr2("cap_p0", &m.cap_p0[0][0], kNCap, 3); Arr2("cap_p1", &m.cap_p1[0][0], kNCap, 3);
      S(cap_r); S(wrist_invw); S(obj_invw); S(table_z);
      Arr("obj_pos", m.obj_pos, 3); S(obj_mass); S(cyl_r); S(cyl_h);
      Arr("goal_pos", m.goal_pos, 3);
      S(margin); S(sol_K); S(sol_B); S(imp_d0); S(imp_dmax); S(imp_width);
      S(ctrl_lo); S(ctrl_hi); S(timestep);
      std::printf("};\n");
    }
    return 0;
  }
  const bool want_ant = argc > 1 && argv[1][0] == 'a';
  const bool want_walker = argc > 1 && argv[1][0] == 'w';
  if (!want_ant) {
    using namespace epa::mj;
    std::printf("// generated by gen_mj_consts.cpp -- do not edit\n");
    const char* names[4] = {"kCheetahModelConst", "kWalkerModelConst", "kWalkerV5ModelConst",
                            "kHopperModelConst"};
    for (int which = want_walker ? 1 : 0; which < (want_walker ? 4 : 1); ++which) {
      CheetahModel<double> m = which == 0 ? BuildCheetahModel()
                               : which == 3 ? BuildHopperModel()
                                            : BuildWalkerModel(which == 2);
      std::printf("constexpr ::epa::mj::CheetahModel<double> %s = {\n", names[which]);
      Arr("lx", m.lx, kNB); Arr("lz", m.lz, kNB); Arr("mass", m.mass, kNB); Arr("iyy", m.iyy, kNB);
      Arr("cx", m.cx, kNB); Arr("cz", m.cz, kNB); Arr("ex", m.ex, kNEnd); Arr("ez", m.ez, kNEnd);
      Arr("er", m.er, kNEnd);
      Arr("stiff", m.stiff, kNU); Arr("damp", m.damp, kNU); Arr("arm", m.arm, kNU);
      Arr("lo", m.lo, kNU); Arr("hi", m.hi, kNU); Arr("gear", m.gear, kNU);
      Arr("dof_invw", m.dof_invw, kNU); Arr("body_invw", m.body_invw, kNB);
      S(total_mass); Arr("bmu", m.bmu, kNB); S(con_K); S(con_B); S(con_d0); S(con_dmax); S(con_width);
      S(lim_K); S(lim_B); S(lim_d0); S(lim_dmax); S(lim_width); S(timestep); S(gravity);
      S(con_margin);
      std::printf("    /*n_pairs*/ %d,\n", m.n_pairs);
      std::printf("};\n");
    }
  }
  if (want_ant) {
    using namespace epa::mj::ant;
    std::printf("// generated by gen_mj_consts.cpp -- do not edit\n");
    AntModel<double> m = BuildAntModel();
    std::printf("constexpr ::epa::mj::ant::AntModel<double> kAntModelConst = {\n");
    Arr("mass", m.mass, kNB); Arr2("com", &m.com[0][0], kNB, 3); Arr2("inertia", &m.inertia[0][0], kNB, 6);
    Arr2("aux_pos", &m.aux_pos[0][0], kNLeg, 3); Arr2("foot_pos", &m.foot_pos[0][0], kNLeg, 3);
    Arr2("ankle_axis", &m.ankle_axis[0][0], kNLeg, 3);
    Arr2("sph", &m.sph[0][0], kNSph, 3); Arr("sph_r", m.sph_r, kNSph);
    Arr("geom_body_invw", m.geom_body_invw, kNGeomBody);
    Arr("lo", m.lo, kNU); Arr("hi", m.hi, kNU); Arr("dof_invw", m.dof_invw, kNU);
    Arr("damp", m.damp, kNU); Arr("arm", m.arm, kNU);
    S(gear); S(total_mass); S(mu); S(margin); S(con_K); S(con_B); S(imp_d0); S(imp_dmax);
    S(imp_width); S(timestep); S(gravity);
    std::printf("};\n");
  }
  return 0;
}
