// TEST FIXTURE: drives DeviceEnvPool<Spec> from C++ through the reference's own types
// (Spec, Array, NamedVector Action/State) in the calling sequence of the reference's C++
// test, envpool/mujoco/gym/mujoco_gym_envpool_test.cc:27-112 -- Reset(ids) -> Recv() ->
// Send(Action{env_id, players.env_id, action}) -> Recv(), and its FrameStack checks -- with
// the pool class swapped from AsyncEnvPool<HalfCheetahEnv> to the device pool.  Also a
// CartPole pool for the classic family and an ownership check (arrays of an earlier Recv are
// not overwritten by later steps).  Exit code 0 = all checks passed; needs a GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
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

int main() {
  CheckAction();
  FrameStack();
  CartPoleEpisode();
  std::puts("refbind_cc_test: OK");
  return 0;
}
