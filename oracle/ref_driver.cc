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

#include "ref_driver_common.h"

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


}  // extern "C"
