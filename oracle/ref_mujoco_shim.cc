// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Bodies for the MuJoCo API subset of oracle/ref_shims_mujoco/mujoco.h, forwarding to the
// plain-C restatement oracle/mjcpu (engine.c / model.c / models.c, compiled as C with the same
// flags as oracle/_build/liboracle.so so that the arithmetic underneath is bit-identical).
// With these the reference's own task wrappers (envpool/mujoco/gym/*.h) link and run inside the
// reference's own AsyncEnvPool -- see ref_mujoco_driver.cc.
//
// What is NOT the reference here: the engine (mj_step / mj_forward / mj_rnePostConstraint ->
// mjc_step / mjc_forward / mjc_rne_post_constraint; "parity unpinned", see mjcpu/mjcpu.h) and
// the compiled models (mj_loadXML returns the hand-transcribed model of the requested file
// name; models.c cites the XML lines).  Body ids for mj_name2id are the XML document order of
// third_party/mujoco_gym_xml_patches/*.xml; tests/test_ref_mujoco.py re-derives them from
// the XML files where /root/reference exists.
#include <mujoco.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "envpool/mujoco/offscreen_renderer.h"

extern "C" {
#include "mjcpu/mjcpu.h"
}

namespace {

struct ModelImpl {
  mjc_model m;
  std::vector<std::string> body_names;  // index = body id ("" = unnamed)
  int tip_body{-1};                     // inverted_double_pendulum: body carrying site "tip"
  double tip_pos[3]{0, 0, 0};
};

struct DataImpl {
  mjData pub;
  mjc_data core;
  double site_xpos[3];
  const ModelImpl* model;
};

std::string Basename(const std::string& path) {
  auto slash = path.rfind('/');
  std::string b = slash == std::string::npos ? path : path.substr(slash + 1);
  // mujoco_env.h:50-58 prefers "<name>_envpool.xml" when that file exists: same model here
  const std::string tag = "_envpool.xml";
  if (b.size() > tag.size() && b.compare(b.size() - tag.size(), tag.size(), tag) == 0) {
    b = b.substr(0, b.size() - tag.size()) + ".xml";
  }
  return b;
}

std::unique_ptr<ModelImpl> Build(const std::string& file) {
  auto im = std::make_unique<ModelImpl>();
  mjc_model* m = &im->m;
  auto names = [&](std::initializer_list<std::pair<int, const char*>> l) {
    for (auto& p : l) im->body_names[p.first] = p.second;
  };
  if (file == "half_cheetah.xml") {
    mjc_build_half_cheetah(m);
  } else if (file == "ant.xml") {
    mjc_build_ant(m);
  } else if (file == "walker2d.xml") {
    mjc_build_walker2d(m, 0);
  } else if (file == "walker2d_v5.xml") {
    mjc_build_walker2d(m, 1);
  } else if (file == "hopper.xml") {
    mjc_build_hopper(m);
  } else if (file == "swimmer.xml") {
    mjc_build_swimmer(m);
  } else if (file == "reacher.xml") {
    mjc_build_reacher(m);
  } else if (file == "pusher.xml") {
    mjc_build_pusher(m, 0);
  } else if (file == "pusher_v5.xml") {
    mjc_build_pusher(m, 1);
  } else if (file == "inverted_pendulum.xml") {
    mjc_build_inverted_pendulum(m);
  } else if (file == "inverted_double_pendulum.xml") {
    mjc_build_inverted_double_pendulum(m);
  } else if (file == "humanoid.xml") {
    mjc_build_humanoid(m, 0);
  } else if (file == "humanoidstandup.xml") {
    mjc_build_humanoid(m, 1);
  } else {
    return nullptr;
  }
  im->body_names.assign(m->nbody, "");
  im->body_names[0] = "world";
  if (file == "ant.xml") {
    names({{1, "torso"}});  // ant_envpool.xml:37
  } else if (file == "reacher.xml") {
    // reacher_envpool.xml:33-45
    names({{1, "body0"}, {2, "body1"}, {3, "fingertip"}, {4, "target"}});
  } else if (file == "pusher.xml" || file == "pusher_v5.xml") {
    // pusher_envpool.xml:30-91
    names({{1, "r_shoulder_pan_link"}, {2, "r_shoulder_lift_link"},
           {3, "r_upper_arm_roll_link"}, {4, "r_upper_arm_link"},
           {5, "r_elbow_flex_link"}, {6, "r_forearm_roll_link"},
           {7, "r_forearm_link"}, {8, "r_wrist_flex_link"},
           {9, "r_wrist_roll_link"}, {10, "tips_arm"}, {11, "object"}, {12, "goal"}});
  } else if (file == "inverted_double_pendulum.xml") {
    // inverted_double_pendulum_envpool.xml:55: <site name="tip" pos="0 0 .6"/> on the last pole
    im->tip_body = m->nbody - 1;
    im->tip_pos[2] = 0.6;
  }
  return im;
}

// one compiled model per file name, shared by all envs of the process (mjModel is read-only
// for the wrappers); never freed
const ModelImpl* Lookup(const std::string& file) {
  static std::mutex mu;
  static std::map<std::string, std::unique_ptr<ModelImpl>> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(file);
  if (it == cache.end()) {
    it = cache.emplace(file, Build(file)).first;
  }
  return it->second.get();
}

void UpdateSites(DataImpl* d) {
  const ModelImpl* im = d->model;
  if (im->tip_body < 0) return;
  // mj_local2Global of a site at (0, 0, z) in its body frame; the x and y terms are exact zeros
  const double* R = d->core.xmat[im->tip_body];
  const double* p = d->core.xpos[im->tip_body];
  for (int r = 0; r < 3; ++r) d->site_xpos[r] = p[r] + R[3 * r + 2] * im->tip_pos[2];
}

}  // namespace

extern "C" {

mjModel* mj_loadXML(const char* filename, const void*, char* error, int error_sz) {
  const std::string file = Basename(filename);
  const ModelImpl* im = Lookup(file);
  if (im == nullptr) {
    if (error != nullptr) {
      std::snprintf(error, error_sz, "mujoco shim: no compiled-in model for '%s'", filename);
    }
    return nullptr;
  }
  auto* m = new mjModel();
  auto* cm = const_cast<mjc_model*>(&im->m);
  m->nq = cm->nq;
  m->nv = cm->nv;
  m->nu = cm->nu;
  m->na = 0;
  m->nbody = cm->nbody;
  m->ngeom = cm->ngeom;
  m->ncam = 0;
  m->opt.timestep = cm->timestep;
  m->stat.extent = 1.0;  // camera distance of the (stubbed) renderer only
  m->qpos0 = cm->qpos0;
  m->body_mass = cm->body_mass;
  m->impl = const_cast<ModelImpl*>(im);
  return m;
}

void mj_deleteModel(mjModel* m) { delete m; }

mjData* mj_makeData(const mjModel* m) {
  auto* d = new DataImpl();
  d->model = static_cast<const ModelImpl*>(m->impl);
  mjc_reset_data(&d->model->m, &d->core);
  std::memset(d->site_xpos, 0, sizeof(d->site_xpos));
  mjData* p = &d->pub;
  p->time = 0;
  p->qpos = d->core.qpos;
  p->qvel = d->core.qvel;
  p->qacc = d->core.qacc;
  p->ctrl = d->core.ctrl;
  p->xpos = &d->core.xpos[0][0];
  p->xipos = &d->core.xipos[0][0];
  p->cfrc_ext = &d->core.cfrc_ext[0][0];
  p->cinert = &d->core.cinert[0][0];
  p->cvel = &d->core.cvel[0][0];
  p->qfrc_actuator = d->core.qfrc_actuator;
  p->qfrc_constraint = d->core.qfrc_constraint;
  p->geom_xpos = &d->core.geom_xpos[0][0];
  p->site_xpos = d->site_xpos;
  p->subtree_com = &d->core.subtree_com[0][0];
  p->impl = d;
  return p;
}

void mj_deleteData(mjData* d) {
  if (d != nullptr) delete static_cast<DataImpl*>(d->impl);
}

void mj_resetData(const mjModel*, mjData* d) {
  auto* di = static_cast<DataImpl*>(d->impl);
  mjc_reset_data(&di->model->m, &di->core);
  std::memset(di->site_xpos, 0, sizeof(di->site_xpos));
  d->time = 0;
}

void mj_forward(const mjModel*, mjData* d) {
  auto* di = static_cast<DataImpl*>(d->impl);
  mjc_forward(&di->model->m, &di->core);
  UpdateSites(di);
}

void mj_step(const mjModel*, mjData* d) {
  auto* di = static_cast<DataImpl*>(d->impl);
  mjc_step(&di->model->m, &di->core);
  UpdateSites(di);  // positions of the last forward evaluation, as mjData keeps them
  d->time = di->core.time;
}

void mj_rnePostConstraint(const mjModel*, mjData* d) {
  auto* di = static_cast<DataImpl*>(d->impl);
  mjc_rne_post_constraint(&di->model->m, &di->core);
}

int mj_name2id(const mjModel* m, int type, const char* name) {
  if (type != mjOBJ_XBODY && type != mjOBJ_BODY) return -1;  // cameras: none compiled
  const auto* im = static_cast<const ModelImpl*>(m->impl);
  for (std::size_t i = 0; i < im->body_names.size(); ++i) {
    if (im->body_names[i] == name) return static_cast<int>(i);
  }
  return -1;
}

void mjv_defaultCamera(mjvCamera* cam) { std::memset(cam, 0, sizeof(*cam)); }

// tests: the body-name table of a model file
int ref_mujoco_body_id(const char* xml_file, const char* body) {
  const ModelImpl* im = Lookup(Basename(xml_file));
  if (im == nullptr) return -2;
  for (std::size_t i = 0; i < im->body_names.size(); ++i) {
    if (im->body_names[i] == body) return static_cast<int>(i);
  }
  return -1;
}
int ref_mujoco_nbody(const char* xml_file) {
  const ModelImpl* im = Lookup(Basename(xml_file));
  return im == nullptr ? -2 : im->m.nbody;
}

}  // extern "C"

// Rendering is out of scope: the renderer is declared by
// envpool/mujoco/offscreen_renderer.h, constructed lazily on the first Render() only.
namespace envpool::mujoco {
OffscreenRenderer::OffscreenRenderer(CameraPolicy camera_policy, bool, bool share_cgl_context,
                                     bool prefer_offline_cgl_context, bool resize_offscreen)
    : camera_policy_(camera_policy),
      share_cgl_context_(share_cgl_context),
      prefer_offline_cgl_context_(prefer_offline_cgl_context),
      resize_offscreen_(resize_offscreen) {}
OffscreenRenderer::~OffscreenRenderer() = default;
void OffscreenRenderer::Render(const mjModel*, mjData*, int width, int height, int,
                               unsigned char* rgb, const mjvCamera*, const mjvOption*) {
  std::memset(rgb, 0, static_cast<std::size_t>(width) * height * 3);
}
}  // namespace envpool::mujoco
