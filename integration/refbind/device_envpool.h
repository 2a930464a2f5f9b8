// REFERENCE-SIDE BINDING (what a maintainer of sail-sg/envpool would add next to
// envpool/core/async_envpool.h) -- compiled against the reference's own headers
// in place, never shipped as product code.
//
// DeviceEnvPool<Spec> implements the reference's executor interface
// `EnvPool<Spec>` (envpool/core/envpool.h:29-56: Send / Recv / Reset on
// std::vector<Array>) on top of the C ABI of libenvpool_amd.so
// (include/envpool_amd.h).  Everything above the interface -- PyEnvPool<Pool>,
// the REGISTER macro (envpool/core/py_envpool.h:206-332), envpool/python/* --
// is the reference's unmodified code: `PyEnvPool<DeviceEnvPool-derived>` is
// what integration/refbind/refbind_module.cc registers.
//
// Recv is zero-copy on the host side: the batch lands in ONE pinned block
// (epa_recv_block) and every state key becomes an `Array` that aliases its
// section of the block and co-owns it through Array's shared_ptr<char>
// (envpool/core/array.h:30-60) -- the same ownership model as the numpy
// capsules of py_envpool.h:40-49, so arrays are never overwritten by later
// steps; the block goes back to a small free list when the last Array dies.
#ifndef INTEGRATION_REFBIND_DEVICE_ENVPOOL_H_
#define INTEGRATION_REFBIND_DEVICE_ENVPOOL_H_

#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "envpool/core/envpool.h"
#include "envpool_amd.h"

namespace envpool_amd_binding {

using Params = std::vector<std::pair<std::string, double>>;

// Every arithmetic entry of the reference Spec's config (bool / int / float / double), under
// the reference's own key name: libenvpool_amd's kernels read their parameters under exactly
// those names and ignore the keys they do not know (num_threads, reward_threshold, ...).
// String / vector entries (base_path, xml_file, env_seed) are skipped; an adapter adds what it
// derives from them (e.g. xml_v5 from xml_file).
template <typename Config>
Params NumericParams(const Config& config, Params extra = {}) {
  Params out;
  const std::vector<std::string> keys = Config::AllKeys();
  std::size_t i = 0;
  std::apply(
      [&](const auto&... value) {
        auto one = [&](const auto& v) {
          using V = std::decay_t<decltype(v)>;
          if constexpr (std::is_arithmetic_v<V>) out.emplace_back(keys[i], static_cast<double>(v));
          ++i;
        };
        (one(value), ...);
      },
      config.AllValues());
  for (auto& kv : extra) out.push_back(kv);
  return out;
}

// pinned host blocks shared by the Arrays of one batch
class BlockPool : public std::enable_shared_from_this<BlockPool> {
 public:
  ~BlockPool() {
    for (auto& f : free_) epa_host_free(f.second);
  }
  std::shared_ptr<char> Take(std::size_t bytes) {
    char* p = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (auto it = free_.begin(); it != free_.end(); ++it) {
        if (it->first == bytes) {
          p = static_cast<char*>(it->second);
          free_.erase(it);
          break;
        }
      }
    }
    if (p == nullptr) p = static_cast<char*>(epa_host_alloc(bytes));
    if (p == nullptr) throw std::runtime_error("epa_host_alloc failed");
    std::weak_ptr<BlockPool> weak = weak_from_this();
    return std::shared_ptr<char>(p, [weak, bytes](char* q) {
      if (auto self = weak.lock()) {
        std::lock_guard<std::mutex> lk(self->mu_);
        if (self->free_.size() < 4) {
          self->free_.emplace_back(bytes, q);
          return;
        }
      }
      epa_host_free(q);
    });
  }

 private:
  std::mutex mu_;
  std::vector<std::pair<std::size_t, void*>> free_;
};

inline void Check(int rc) {
  if (rc == EPA_OK) return;
  // error classes of the reference: std::invalid_argument -> ValueError,
  // std::runtime_error -> RuntimeError (pybind11's default translation)
  if (rc == EPA_ERR_INVALID) throw std::invalid_argument(epa_last_error());
  throw std::runtime_error(epa_last_error());
}

template <typename Spec>
class DeviceEnvPool : public EnvPool<Spec> {
 public:
  DeviceEnvPool(const char* family, const Spec& spec, const Params& params)
      : EnvPool<Spec>(spec),
        state_specs_(spec.state_spec.template AllValues<ShapeSpec>()),
        blocks_(std::make_shared<BlockPool>()) {
    std::vector<const char*> keys;
    std::vector<double> vals;
    for (auto& kv : params) {
      keys.push_back(kv.first.c_str());
      vals.push_back(kv.second);
    }
    const auto& env_seed = spec.config["env_seed"_];
    std::vector<int32_t> seeds(env_seed.begin(), env_seed.end());
    epa_config cfg{};
    cfg.num_envs = spec.config["num_envs"_];
    cfg.batch_size = spec.config["batch_size"_];
    cfg.seed = spec.config["seed"_];
    cfg.env_seed = seeds.empty() ? nullptr : seeds.data();
    cfg.max_episode_steps = spec.config["max_episode_steps"_];
    cfg.device = 0;
    cfg.env_id_offset = 0;
    cfg.n_params = static_cast<int32_t>(keys.size());
    cfg.param_keys = keys.data();
    cfg.param_values = vals.data();
    if (!seeds.empty() && static_cast<int>(seeds.size()) != cfg.num_envs) {
      throw std::invalid_argument("env_seed must have num_envs entries");
    }
    Check(epa_create(family, &cfg, &h_));
    sync_ = spec.config["batch_size"_] == spec.config["num_envs"_];
    // the C ABI's view of the state keys must agree with Spec::StateSpec
    std::vector<epa_key_info> info(state_specs_.size() + 1);
    int n = 0;
    Check(epa_describe_state(family, &cfg, info.data(), static_cast<int>(info.size()), &n));
    if (n != static_cast<int>(state_specs_.size())) {
      throw std::runtime_error("state key count differs between Spec and libenvpool_amd");
    }
    for (int i = 0; i < n; ++i) {
      std::size_t row = state_specs_[i].element_size;
      for (std::size_t d = 0; d < state_specs_[i].shape.size(); ++d) {
        int s = state_specs_[i].shape[d];
        if (d == 0 && s == -1) continue;  // the per-player dim: 1 for these envs
        row *= static_cast<std::size_t>(s);
      }
      if (row != static_cast<std::size_t>(info[i].row_bytes)) {
        throw std::runtime_error(std::string("row size of state key ") + info[i].name +
                                 " differs between Spec and libenvpool_amd");
      }
    }
  }
  ~DeviceEnvPool() override {
    if (h_ != nullptr) epa_destroy(h_);
  }
  DeviceEnvPool(const DeviceEnvPool&) = delete;
  DeviceEnvPool& operator=(const DeviceEnvPool&) = delete;

  // action = {env_id, players.env_id, <env action>}  (env_spec.h:32-35)
  void Send(const std::vector<Array>& action) override {
    int k = static_cast<int>(action[0].Shape(0));
    // A whole-pool step of a sync pool names the block of ITS batch now (epa_send_into): the reference allocates a
    // batch's output buffers before its workers write them too (state_buffer_queue.h:123-140), and here the step kernel
    // then writes its rows straight into the block Recv() will hand out.  Whether the library takes the offer is its
    // business: Recv passes the same block to epa_recv_block either way.
    std::shared_ptr<char> block;
    std::size_t total = 0;
    if (k == static_cast<int>(sync_ ? this->spec.config["num_envs"_] : this->spec.config["batch_size"_])) {
      const int n = static_cast<int>(state_specs_.size());
      std::vector<std::size_t> off(n);
      Check(epa_recv_layout(h_, k, off.data(), n, &total));
      if (total >= kPostBytes) block = blocks_->Take(total);
    }
    if (block) {
      Check(epa_send_into(h_, static_cast<const int32_t*>(action[0].Data()), k, action.back().Data(), block.get(),
                          total));
    } else {
      Check(epa_send(h_, static_cast<const int32_t*>(action[0].Data()), k, action.back().Data()));
    }
    if (k > 0) Push(k, std::move(block));
  }
  void Send(std::vector<Array>&& action) override { Send(action); }

  void Reset(const Array& env_ids) override {
    int k = static_cast<int>(env_ids.Shape(0));
    Check(epa_reset(h_, static_cast<const int32_t*>(env_ids.Data()), k));
    if (k > 0) Push(k);
  }

  std::vector<Array> Recv() override {
    int cap = this->spec.config["batch_size"_];
    if (sync_) {
      std::lock_guard<std::mutex> lk(mu_);
      if (!pending_.empty()) cap = pending_.front();
    }
    const int n = static_cast<int>(state_specs_.size());
    std::vector<std::size_t> off(n);
    std::size_t total = 0;
    Check(epa_recv_layout(h_, cap, off.data(), n, &total));
    std::shared_ptr<char> block;
    {  // the block Send named for exactly these rows, if it did (async: the oldest send is a whole, untouched batch)
      std::lock_guard<std::mutex> lk(mu_);
      if (!posted_.empty() && (sync_ || (!pending_.empty() && pending_.front() == cap))) block = posted_.front();
    }
    if (!block) block = blocks_->Take(total > 0 ? total : 256);
    int32_t k = 0;
    Check(epa_recv_block(h_, block.get(), total, off.data(), n, &k));
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (sync_) {
        if (!pending_.empty()) pending_.pop_front();
        if (!posted_.empty()) posted_.pop_front();
      } else {  // async: rows drain across the sends / resets in order
        int left = k;
        while (left > 0 && !pending_.empty()) {
          if (pending_.front() <= left) {
            left -= pending_.front();
            pending_.pop_front();
            if (!posted_.empty()) posted_.pop_front();
          } else {
            pending_.front() -= left;
            left = 0;
          }
        }
      }
    }
    std::vector<Array> out;
    out.reserve(n);
    for (int i = 0; i < n; ++i) {  // same key order as Spec::StateKeys
      std::vector<int> shape = {static_cast<int>(k)};
      const auto& s = state_specs_[i].shape;
      for (std::size_t d = 0; d < s.size(); ++d) {
        if (d == 0 && s[d] == -1) continue;  // [B * players] with one player: [B]
        shape.push_back(s[d]);
      }
      // the Array points into the block and co-owns it through its deleter
      out.emplace_back(ShapeSpec(state_specs_[i].element_size, std::move(shape)),
                       block.get() + off[i], [block](char* /*unused*/) {});
    }
    return out;
  }

 private:
  void Push(int k, std::shared_ptr<char> block = nullptr) {
    std::lock_guard<std::mutex> lk(mu_);
    pending_.push_back(k);
    posted_.push_back(std::move(block));
  }
  static constexpr std::size_t kPostBytes = 256 * 1024;  // smaller batches: epa_recv_block's own download is as fast
  epa_pool* h_{nullptr};
  std::vector<ShapeSpec> state_specs_;
  std::shared_ptr<BlockPool> blocks_;
  bool sync_{true};
  std::mutex mu_;
  std::deque<int> pending_;  // rows (left) of each outstanding Send / Reset
  std::deque<std::shared_ptr<char>> posted_;  // ... and the block Send named for it (null: Recv takes one)
};

}  // namespace envpool_amd_binding

#endif  // INTEGRATION_REFBIND_DEVICE_ENVPOOL_H_
