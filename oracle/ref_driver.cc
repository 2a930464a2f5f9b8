// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// oracle/_ref: drives the *reference's own* header-only runtime and env bodies
// (compiled in place from /root/reference, never copied) through a tiny C API
// so that Python tests / fixture generators can obtain reference rollouts.
//
// What is compiled from the reference (file:line are the classes driven):
//   envpool/core/async_envpool.h:42-238   AsyncEnvPool (thread pool + queues)
//   envpool/classic_control/cartpole.h:51, pendulum.h:49, acrobot.h:50,
//     mountain_car.h:49, mountain_car_continuous.h:49
//   envpool/toy_text/catch.h:49, frozen_lake.h:50, taxi.h:48, nchain.h:47,
//     cliffwalking.h:50, blackjack.h:49
// The driver follows the calling sequence of the reference's own C++ test
// (envpool/mujoco/gym/mujoco_gym_envpool_test.cc:27-56): Reset(ids) -> Recv()
// -> Send(vector<Array>{env_id, players.env_id, action}) -> Recv().
//
// Un-vendored third-party headers are replaced by oracle/ref_shims/*.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load the resulting library.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "envpool/classic_control/acrobot.h"
#include "envpool/classic_control/cartpole.h"
#include "envpool/classic_control/mountain_car.h"
#include "envpool/classic_control/mountain_car_continuous.h"
#include "envpool/classic_control/pendulum.h"
#include "envpool/toy_text/blackjack.h"
#include "envpool/toy_text/catch.h"
#include "envpool/toy_text/cliffwalking.h"
#include "envpool/toy_text/frozen_lake.h"
#include "envpool/toy_text/nchain.h"
#include "envpool/toy_text/taxi.h"

// The real renderers need OpenCV; rendering is out of scope.
namespace classic_control::rendering {
void RenderCartPole(double, double, int, int, std::uint8_t*) {}
void RenderPendulum(double, bool, double, int, int, std::uint8_t*) {}
void RenderMountainCar(double, double, int, int, std::uint8_t*) {}
void RenderAcrobot(double, double, int, int, std::uint8_t*) {}
}  // namespace classic_control::rendering

namespace {

// dtype codes shared with oracle/restate and the tests
// 0=int32 1=float32 2=float64 3=bool(uint8)
template <typename D>
constexpr int DtypeCode() {
  if (std::is_same_v<D, int>) return 0;
  if (std::is_same_v<D, float>) return 1;
  if (std::is_same_v<D, double>) return 2;
  if (std::is_same_v<D, bool>) return 3;
  return -1;
}

struct KeyInfo {
  std::string name;
  int dtype;
  int elems;      // elements per row (product of non-batch dims)
  int elem_size;  // bytes
};

struct IRef {
  virtual ~IRef() = default;
  std::vector<KeyInfo> state_keys;
  KeyInfo action;
  int num_envs{0};
  virtual void Reset(const int* ids, int k, void** out) = 0;
  virtual void Step(const int* ids, int k, const void* action, void** out) = 0;
  virtual double TimeSteps(int steps, const void* action) = 0;
};

template <typename Pool>
struct Ref : IRef {
  using Spec = typename Pool::Spec;
  std::unique_ptr<Spec> spec;
  std::unique_ptr<Pool> pool;
  std::vector<int> action_tail_shape;

  template <typename Configure>
  Ref(int n, int seed, int max_steps, int num_threads, Configure&& cfg) {
    auto config = Spec::kDefaultConfig;
    config["num_envs"_] = n;
    config["batch_size"_] = n;
    config["seed"_] = seed;
    config["num_threads"_] = num_threads;
    if (max_steps > 0) config["max_episode_steps"_] = max_steps;
    cfg(config);
    spec = std::make_unique<Spec>(config.AllValues());
    pool = std::make_unique<Pool>(*spec);
    num_envs = n;
    auto keys = Spec::StateSpec::AllKeys();
    std::size_t i = 0;
    std::apply(
        [&](auto&&... s) {
          (AddKey(keys[i++], s, &state_keys), ...);
        },
        spec->state_spec.AllValues());
    auto akeys = Spec::ActionSpec::AllKeys();
    std::vector<KeyInfo> ak;
    i = 0;
    std::apply([&](auto&&... s) { (AddKey(akeys[i++], s, &ak), ...); },
               spec->action_spec.AllValues());
    action = ak.back();
    auto shapes = spec->action_spec.template AllValues<ShapeSpec>();
    const auto& sh = shapes.back().shape;
    for (std::size_t j = 1; j < sh.size(); ++j) action_tail_shape.push_back(sh[j]);
  }

  template <typename S>
  static void AddKey(const std::string& name, const S& s,
                     std::vector<KeyInfo>* out) {
    int elems = 1;
    for (int d : s.shape) {
      if (d != -1) elems *= d;
    }
    out->push_back(KeyInfo{name, DtypeCode<typename S::dtype>(), elems,
                           static_cast<int>(sizeof(typename S::dtype))});
  }

  void CopyOut(const std::vector<Array>& arr, void** out) {
    for (std::size_t i = 0; i < arr.size(); ++i) {
      if (out[i] != nullptr) {
        std::memcpy(out[i], arr[i].Data(), arr[i].size * arr[i].element_size);
      }
    }
  }

  void Reset(const int* ids, int k, void** out) override {
    Array env_ids(::Spec<int>({k}));
    std::memcpy(env_ids.Data(), ids, sizeof(int) * k);
    pool->Reset(env_ids);
    CopyOut(pool->Recv(), out);
  }

  std::vector<Array> MakeAction(const int* ids, int k, const void* act) {
    std::vector<int> ashape = {k};
    for (int d : action_tail_shape) ashape.push_back(d);
    std::vector<Array> raw({Array(::Spec<int>({k})), Array(::Spec<int>({k})),
                            Array(ShapeSpec(action.elem_size, ashape))});
    std::memcpy(raw[0].Data(), ids, sizeof(int) * k);
    std::memcpy(raw[1].Data(), ids, sizeof(int) * k);
    std::memcpy(raw[2].Data(), act,
                static_cast<std::size_t>(k) * action.elems * action.elem_size);
    return raw;
  }

  void Step(const int* ids, int k, const void* act, void** out) override {
    pool->Send(MakeAction(ids, k, act));
    CopyOut(pool->Recv(), out);
  }

  // Timed sync loop: `steps` x (Send(all envs) ; Recv()).  A fresh action
  // batch Array is built per step, as the pybind shim does
  // (envpool/core/py_envpool.h:244-250).
  double TimeSteps(int steps, const void* act) override {
    std::vector<int> ids(num_envs);
    for (int i = 0; i < num_envs; ++i) ids[i] = i;
    auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; ++s) {
      pool->Send(MakeAction(ids.data(), num_envs, act));
      auto ret = pool->Recv();
      (void)ret;
    }
    std::chrono::duration<double> dt = std::chrono::steady_clock::now() - t0;
    return dt.count();
  }
};

double Extra(const double* extra, int n, int i, double dflt) {
  return (extra != nullptr && i < n) ? extra[i] : dflt;
}

}  // namespace

extern "C" {

void* orc_create(const char* task, int num_envs, int seed,
                 int max_episode_steps, const double* extra, int n_extra,
                 int num_threads) {
  std::string t(task);
  auto none = [](auto&) {};
  try {
    if (t == "CartPole") {
      return new Ref<classic_control::CartPoleEnvPool>(
          num_envs, seed, max_episode_steps, num_threads, none);
    }
    if (t == "Pendulum") {
      int version = static_cast<int>(Extra(extra, n_extra, 0, 0));
      return new Ref<classic_control::PendulumEnvPool>(
          num_envs, seed, max_episode_steps, num_threads,
          [&](auto& c) { c["version"_] = version; });
    }
    if (t == "MountainCar") {
      return new Ref<classic_control::MountainCarEnvPool>(
          num_envs, seed, max_episode_steps, num_threads, none);
    }
    if (t == "MountainCarContinuous") {
      return new Ref<classic_control::MountainCarContinuousEnvPool>(
          num_envs, seed, max_episode_steps, num_threads, none);
    }
    if (t == "Acrobot") {
      return new Ref<classic_control::AcrobotEnvPool>(
          num_envs, seed, max_episode_steps, num_threads, none);
    }
    if (t == "Catch") {
      int h = static_cast<int>(Extra(extra, n_extra, 0, 10));
      int w = static_cast<int>(Extra(extra, n_extra, 1, 5));
      return new Ref<toy_text::CatchEnvPool>(
          num_envs, seed, max_episode_steps, num_threads, [&](auto& c) {
            c["height"_] = h;
            c["width"_] = w;
          });
    }
    if (t == "FrozenLake") {
      int size = static_cast<int>(Extra(extra, n_extra, 0, 4));
      return new Ref<toy_text::FrozenLakeEnvPool>(
          num_envs, seed, max_episode_steps, num_threads,
          [&](auto& c) { c["size"_] = size; });
    }
    if (t == "Taxi") {
      return new Ref<toy_text::TaxiEnvPool>(num_envs, seed, max_episode_steps,
                                            num_threads, none);
    }
    if (t == "NChain") {
      return new Ref<toy_text::NChainEnvPool>(num_envs, seed, max_episode_steps,
                                              num_threads, none);
    }
    if (t == "CliffWalking") {
      bool slip = Extra(extra, n_extra, 0, 0) != 0;
      return new Ref<toy_text::CliffWalkingEnvPool>(
          num_envs, seed, max_episode_steps, num_threads,
          [&](auto& c) { c["is_slippery"_] = slip; });
    }
    if (t == "Blackjack") {
      bool natural = Extra(extra, n_extra, 0, 0) != 0;
      bool sab = Extra(extra, n_extra, 1, 1) != 0;
      return new Ref<toy_text::BlackjackEnvPool>(
          num_envs, seed, max_episode_steps, num_threads, [&](auto& c) {
            c["natural"_] = natural;
            c["sab"_] = sab;
          });
    }
  } catch (const std::exception& e) {
    std::cerr << "orc_create(" << t << "): " << e.what() << std::endl;
  }
  return nullptr;
}

int orc_num_state_keys(void* h) {
  return static_cast<int>(static_cast<IRef*>(h)->state_keys.size());
}

int orc_state_key(void* h, int i, char* name, int* dtype, int* elems) {
  auto* r = static_cast<IRef*>(h);
  if (i < 0 || i >= static_cast<int>(r->state_keys.size())) return -1;
  std::strncpy(name, r->state_keys[i].name.c_str(), 63);
  name[63] = 0;
  *dtype = r->state_keys[i].dtype;
  *elems = r->state_keys[i].elems;
  return 0;
}

int orc_action_info(void* h, int* dtype, int* elems) {
  auto* r = static_cast<IRef*>(h);
  *dtype = r->action.dtype;
  *elems = r->action.elems;
  return 0;
}

void orc_reset(void* h, const int* ids, int k, void** out) {
  static_cast<IRef*>(h)->Reset(ids, k, out);
}

void orc_step(void* h, const int* ids, int k, const void* action, void** out) {
  static_cast<IRef*>(h)->Step(ids, k, action, out);
}

double orc_time_steps(void* h, int steps, const void* action) {
  return static_cast<IRef*>(h)->TimeSteps(steps, action);
}

void orc_destroy(void* h) { delete static_cast<IRef*>(h); }

const char* orc_kind() { return "reference"; }

}  // extern "C"
