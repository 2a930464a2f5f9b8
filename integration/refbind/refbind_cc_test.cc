// TEST FIXTURE: drives DeviceEnvPool<Spec> from C++ through the reference's own types
// (Spec, Array, NamedVector Action/State) in the calling sequence of the reference's C++
// test, envpool/mujoco/gym/mujoco_gym_envpool_test.cc:27-112 -- Reset(ids) -> Recv() ->
// Send(Action{env_id, players.env_id, action}) -> Recv(), and its FrameStack checks -- with
// the pool class swapped from AsyncEnvPool<HalfCheetahEnv> to the device pool.  Also a
// CartPole pool for the classic family and an ownership check (arrays of an earlier Recv are
// not overwritten by later steps), and a producer / consumer pair of std::threads whose consumer
// is inside Recv() BEFORE the producer's Reset / Send (Recv blocks: async_envpool.h:169-181,
// state_buffer_queue.h:148-163), sync and async mode.  Exit code 0 = all checks passed; needs a GPU.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "envpool/classic_control/cartpole.h"
#include "envpool/mujoco/gym/half_cheetah.h"

#include "device_envpool.h"

namespace eab = envpool_amd_binding;

#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                         \
    }                                                                       \
  } while (0)

template <typename T>
static T At(const Array& a, int i) {
  return static_cast<const T*>(a.Data())[i];
}

struct DeviceHalfCheetahPool : eab::DeviceEnvPool<mujoco_gym::HalfCheetahEnvSpec> {
  explicit DeviceHalfCheetahPool(const mujoco_gym::HalfCheetahEnvSpec& s)
      : DeviceEnvPool("HalfCheetah", s,
                      {{"frame_skip", s.config["frame_skip"_]},
                       {"frame_stack", s.config["frame_stack"_]},
                       {"ctrl_cost_weight", s.config["ctrl_cost_weight"_]},
                       {"forward_reward_weight", s.config["forward_reward_weight"_]},
                       {"reset_noise_scale", s.config["reset_noise_scale"_]}}) {}
};
struct DeviceCartPolePool : eab::DeviceEnvPool<classic_control::CartPoleEnvSpec> {
  explicit DeviceCartPolePool(const classic_control::CartPoleEnvSpec& s)
      : DeviceEnvPool("CartPole", s, {}) {}
};

using MjcAction = typename DeviceHalfCheetahPool::Action;
using MjcState = typename DeviceHalfCheetahPool::State;

static void CheckAction() {  // mujoco_gym_envpool_test.cc:27-56
  auto config = mujoco_gym::HalfCheetahEnvSpec::kDefaultConfig;
  int num_envs = 128;
  config["num_envs"_] = num_envs;
  mujoco_gym::HalfCheetahEnvSpec spec(config);
  DeviceHalfCheetahPool envpool(spec);
  Array all_env_ids(Spec<int>({num_envs}));
  for (int i = 0; i < num_envs; ++i) all_env_ids[i] = i;
  envpool.Reset(all_env_ids);
  std::vector<Array> reset_vec = envpool.Recv();
  MjcState reset_state(&reset_vec);
  EXPECT(reset_state["obs"_].Shape() == std::vector<std::size_t>({128, 17}));
  EXPECT(reset_state["reward"_].Shape() == std::vector<std::size_t>({128}));
  std::vector<Array> raw_action({Array(Spec<int>({num_envs})), Array(Spec<int>({num_envs})),
                                 Array(Spec<double>({num_envs, 6}))});
  MjcAction action(&raw_action);
  for (int i = 0; i < num_envs; ++i) {
    action["env_id"_][i] = i;
    action["players.env_id"_][i] = i;
    for (int j = 0; j < 6; ++j) action["action"_][i][j] = (i + j + 1) / 100.0;
  }
  envpool.Send(action);
  std::vector<Array> state_vec = envpool.Recv();
  MjcState state(&state_vec);
  EXPECT(state["obs"_].Shape() == std::vector<std::size_t>({128, 17}));
  for (int i = 0; i < num_envs; ++i) {
    EXPECT(At<int>(state["info:env_id"_], i) == i);
    EXPECT(At<int>(state["elapsed_step"_], i) == 1);
    EXPECT(!At<bool>(state["done"_], i));
    // reward = forward_reward - ctrl_cost; ctrl cost = 0.1 * sum a^2 (half_cheetah.h:142-150)
    double ctrl = 0;
    for (int j = 0; j < 6; ++j) ctrl += 0.1 * ((i + j + 1) / 100.0) * ((i + j + 1) / 100.0);
    double rc = At<double>(state["info:reward_ctrl"_], i);
    EXPECT(std::fabs(rc + ctrl) < 1e-12);
    double rr = At<double>(state["info:reward_run"_], i);
    EXPECT(std::fabs(At<float>(state["reward"_], i) - static_cast<float>(rr + rc)) < 1e-6f);
  }
  // ownership: the reset batch is untouched by the step (fresh buffer per batch)
  const auto* r0 = static_cast<const double*>(reset_state["obs"_].Data());
  const auto* s0 = static_cast<const double*>(state["obs"_].Data());
  EXPECT(r0 != s0);
  bool differs = false;
  for (int j = 0; j < 17; ++j) differs = differs || r0[j] != s0[j];
  EXPECT(differs);
}

static void FrameStack() {  // mujoco_gym_envpool_test.cc:58-112
  auto config = mujoco_gym::HalfCheetahEnvSpec::kDefaultConfig;
  constexpr int num_envs = 1, frame_stack = 4, obs_dim = 17;
  config["num_envs"_] = num_envs;
  config["batch_size"_] = num_envs;
  config["seed"_] = 0;
  config["frame_stack"_] = frame_stack;
  mujoco_gym::HalfCheetahEnvSpec spec(config);
  EXPECT(spec.state_spec["obs"_].shape == std::vector<int>({frame_stack, obs_dim}));
  DeviceHalfCheetahPool envpool(spec);
  TArray<int> all_env_ids(Spec<int>({num_envs}));
  all_env_ids[0] = 0;
  envpool.Reset(all_env_ids);
  std::vector<Array> reset_vec = envpool.Recv();
  MjcState reset_state(&reset_vec);
  EXPECT(reset_state["obs"_].Shape() ==
         std::vector<std::size_t>({num_envs, frame_stack, obs_dim}));
  const auto reset_obs = TArray<mjtNum>(reset_state["obs"_][0]);
  const auto* reset_ptr = static_cast<const mjtNum*>(reset_obs.Data());
  for (int i = 1; i < frame_stack; ++i) {
    for (int j = 0; j < obs_dim; ++j) EXPECT(reset_ptr[j] == reset_ptr[i * obs_dim + j]);
  }
  std::vector<Array> raw_action({Array(Spec<int>({num_envs})), Array(Spec<int>({num_envs})),
                                 Array(Spec<double>({num_envs, 6}))});
  MjcAction action(&raw_action);
  action["env_id"_][0] = 0;
  action["players.env_id"_][0] = 0;
  for (int j = 0; j < 6; ++j) action["action"_][0][j] = 0.0;
  envpool.Send(action);
  std::vector<Array> step_vec = envpool.Recv();
  MjcState step_state(&step_vec);
  EXPECT(step_state["obs"_].Shape() ==
         std::vector<std::size_t>({num_envs, frame_stack, obs_dim}));
  const auto step_obs = TArray<mjtNum>(step_state["obs"_][0]);
  const auto* step_ptr = static_cast<const mjtNum*>(step_obs.Data());
  for (int i = 0; i < frame_stack - 1; ++i) {
    for (int j = 0; j < obs_dim; ++j) EXPECT(step_ptr[i * obs_dim + j] == reset_ptr[j]);
  }
  bool changed = false;
  for (int j = 0; j < obs_dim; ++j) {
    changed = changed || (step_ptr[(frame_stack - 1) * obs_dim + j] != reset_ptr[j]);
  }
  EXPECT(changed);
}

static void CartPoleEpisode() {  // the classic family through the same interface
  auto config = classic_control::CartPoleEnvSpec::kDefaultConfig;
  int num_envs = 64;
  config["num_envs"_] = num_envs;
  config["max_episode_steps"_] = 200;
  config["seed"_] = 3;
  classic_control::CartPoleEnvSpec spec(config);
  DeviceCartPolePool envpool(spec);
  using Action = typename DeviceCartPolePool::Action;
  using State = typename DeviceCartPolePool::State;
  Array ids(Spec<int>({num_envs}));
  for (int i = 0; i < num_envs; ++i) ids[i] = i;
  envpool.Reset(ids);
  std::vector<Array> st_vec = envpool.Recv();
  State st(&st_vec);
  EXPECT(st["obs"_].Shape() == std::vector<std::size_t>({64, 4}));
  int dones = 0;
  for (int t = 0; t < 300; ++t) {
    std::vector<Array> raw({Array(Spec<int>({num_envs})), Array(Spec<int>({num_envs})),
                            Array(Spec<int>({num_envs}))});
    Action action(&raw);
    for (int i = 0; i < num_envs; ++i) {
      action["env_id"_][i] = i;
      action["players.env_id"_][i] = i;
      action["action"_][i] = (i + t) & 1;
    }
    envpool.Send(action);
    std::vector<Array> s_vec = envpool.Recv();
    State s(&s_vec);
    for (int i = 0; i < num_envs; ++i) {
      bool done = At<bool>(s["done"_], i);
      dones += done;
      int el = At<int>(s["elapsed_step"_], i);
      EXPECT(el >= 0 && el <= 200);
      // trunc = done && elapsed >= max_episode_steps (env.h:241)
      EXPECT(At<bool>(s["trunc"_], i) == (done && el >= 200));
      EXPECT(At<float>(s["discount"_], i) == (done ? 0.0f : 1.0f));
    }
  }
  EXPECT(dones > 0);
  // invalid ids surface as std::invalid_argument like the reference's spec checks
  Array bad(Spec<int>({1}));
  bad[0] = num_envs + 5;
  bool threw = false;
  try {
    envpool.Reset(bad);
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  EXPECT(threw);
}

// Recv blocks until the producer has sent (async_envpool.h:169-181): the consumer thread enters
// Recv() first, the producer resets / steps afterwards.  Sync mode: T steps, batches arrive in
// send order.  Async mode (batch_size < num_envs): the consumer hands the env ids of each batch
// to the producer, which sends their next actions -- the actor loop of the reference's README.
static void ProducerConsumer(int num_envs, int batch_size, int steps) {
  auto config = classic_control::CartPoleEnvSpec::kDefaultConfig;
  config["num_envs"_] = num_envs;
  config["batch_size"_] = batch_size;
  config["max_episode_steps"_] = 1000000;
  config["seed"_] = 5;
  classic_control::CartPoleEnvSpec spec(config);
  DeviceCartPolePool envpool(spec);
  using Action = typename DeviceCartPolePool::Action;
  using State = typename DeviceCartPolePool::State;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::vector<int>> handed;  // env ids of received batches, consumer -> producer
  std::atomic<int> received{0};
  const int batches = steps * (num_envs / batch_size) + num_envs / batch_size;  // resets + steps
  std::vector<int> rows_of_env(num_envs, 0);
  std::thread consumer([&] {
    for (int b = 0; b < batches; ++b) {
      std::vector<Array> vec = envpool.Recv();  // the first call arrives before any Reset
      State st(&vec);
      const int k = static_cast<int>(st["info:env_id"_].Shape(0));
      EXPECT(k == batch_size);
      std::vector<int> ids(k);
      for (int i = 0; i < k; ++i) {
        ids[i] = At<int>(st["info:env_id"_], i);
        EXPECT(ids[i] >= 0 && ids[i] < num_envs);
        // row r of env e is its r-th result: elapsed_step counts them (no episode ends here
        // before the pole falls; a finished env restarts at 0)
        int el = At<int>(st["elapsed_step"_], i);
        EXPECT(el >= 0 && el <= rows_of_env[ids[i]]);
        ++rows_of_env[ids[i]];
      }
      received.fetch_add(1);
      std::lock_guard<std::mutex> lk(mu);
      handed.push_back(std::move(ids));
      cv.notify_one();
    }
  });
  std::this_thread::sleep_for(std::chrono::milliseconds(150));
  EXPECT(received.load() == 0);  // nothing was sent: the consumer is blocked inside Recv
  Array all_ids(Spec<int>({num_envs}));
  for (int i = 0; i < num_envs; ++i) all_ids[i] = i;
  envpool.Reset(all_ids);
  for (int b = 0; b < batches - num_envs / batch_size; ++b) {
    std::vector<int> ids;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !handed.empty(); });
      ids = std::move(handed.front());
      handed.pop_front();
    }
    const int k = static_cast<int>(ids.size());
    std::vector<Array> raw({Array(Spec<int>({k})), Array(Spec<int>({k})), Array(Spec<int>({k}))});
    Action action(&raw);
    for (int i = 0; i < k; ++i) {
      action["env_id"_][i] = ids[i];
      action["players.env_id"_][i] = ids[i];
      action["action"_][i] = (ids[i] + b) & 1;
    }
    envpool.Send(action);
  }
  consumer.join();
  EXPECT(received.load() == batches);
  for (int e = 0; e < num_envs; ++e) EXPECT(rows_of_env[e] == steps + 1);
}

int main() {
  ProducerConsumer(64, 64, 40);   // sync mode
  ProducerConsumer(64, 16, 40);   // async mode
  CheckAction();
  FrameStack();
  CartPoleEpisode();
  std::puts("refbind_cc_test: OK");
  return 0;
}
