// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Generic part of the oracle/_ref drivers: wraps one of the reference's own
// `AsyncEnvPool<Env>` instantiations (compiled in place from /root/reference) behind
// the small orc_* C API that oracle/orc.py binds.  Each driver translation unit includes
// the reference headers of its family, this header, and defines orc_create().
// The calling sequence follows the reference's own C++ test
// (envpool/mujoco/gym/mujoco_gym_envpool_test.cc:27-56): Reset(ids) -> Recv()
// -> Send(vector<Array>{env_id, players.env_id, action}) -> Recv().
#ifndef ORACLE_REF_DRIVER_COMMON_H_
#define ORACLE_REF_DRIVER_COMMON_H_

#include <chrono>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

namespace {

// dtype codes shared with oracle/restate and the tests
// 0=int32 1=float32 2=float64 3=bool(uint8) 4=uint8
template <typename D>
constexpr int DtypeCode() {
  if (std::is_same_v<D, int>) return 0;
  if (std::is_same_v<D, float>) return 1;
  if (std::is_same_v<D, double>) return 2;
  if (std::is_same_v<D, bool>) return 3;
  if (std::is_same_v<D, std::uint8_t>) return 4;
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

#ifndef ORC_KIND
#define ORC_KIND "reference"
#endif
const char* orc_kind() { return ORC_KIND; }

}  // extern "C"

#endif  // ORACLE_REF_DRIVER_COMMON_H_
