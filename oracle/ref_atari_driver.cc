// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// oracle/_ref/libref_atari.so: the reference's own AtariEnv / AtariEnvPool
// (envpool/atari/atari_env.h:97-349, compiled in place from /root/reference, never
// copied) driven through the orc_* C API.  Two un-vendored third parties are shimmed
// (oracle/ref_shims_atari): ALE -> the synthetic console of tests/synth_ale, OpenCV's
// cv::resize -> its plain-C restatement in oracle/atari/atari_post.c.  Everything else --
// noop / FIRE resets, the frame_skip loop, the two-frame max-pool, the frame stack,
// episodic life, reward clipping, zero_discount_on_life_loss, the elapsed_step / trunc /
// discount overrides of WriteState, and the whole AsyncEnvPool runtime under it -- is the
// reference's code.
//
// orc_create("Atari", ...) extra[]: 0 stack_num, 1 frame_skip, 2 noop_max,
// 3 zero_discount_on_life_loss, 4 episodic_life, 5 reward_clip, 6 use_fire_reset,
// 7 img_height, 8 img_width, 9 rom variant (0 synth_fire, 1 synth_nofire, 2 synth_fire_short),
// 10 mode, 11 difficulty, 12 full_action_space, 13 repeat_action_probability,
// 14 use_inter_area_resize, 15 gray_scale.
#include "envpool/atari/atari_env.h"

#include "ref_driver_common.h"

extern "C" {

void* orc_create(const char* task, int num_envs, int seed, int max_episode_steps,
                 const double* extra, int n_extra, int num_threads) {
  std::string t(task);
  if (t != "Atari") return nullptr;
  static const char* kRoms[] = {"synth_fire", "synth_nofire", "synth_fire_short"};
  try {
    return new Ref<atari::AtariEnvPool>(
        num_envs, seed, max_episode_steps, num_threads, [&](auto& c) {
          c["stack_num"_] = static_cast<int>(Extra(extra, n_extra, 0, 4));
          c["frame_skip"_] = static_cast<int>(Extra(extra, n_extra, 1, 4));
          c["noop_max"_] = static_cast<int>(Extra(extra, n_extra, 2, 30));
          c["zero_discount_on_life_loss"_] = Extra(extra, n_extra, 3, 0) != 0;
          c["episodic_life"_] = Extra(extra, n_extra, 4, 0) != 0;
          c["reward_clip"_] = Extra(extra, n_extra, 5, 0) != 0;
          c["use_fire_reset"_] = Extra(extra, n_extra, 6, 1) != 0;
          c["img_height"_] = static_cast<int>(Extra(extra, n_extra, 7, 84));
          c["img_width"_] = static_cast<int>(Extra(extra, n_extra, 8, 84));
          c["task"_] = std::string(kRoms[static_cast<int>(Extra(extra, n_extra, 9, 0)) % 3]);
          c["mode"_] = static_cast<int>(Extra(extra, n_extra, 10, -1));
          c["difficulty"_] = static_cast<int>(Extra(extra, n_extra, 11, -1));
          c["full_action_space"_] = Extra(extra, n_extra, 12, 0) != 0;
          c["repeat_action_probability"_] = static_cast<float>(Extra(extra, n_extra, 13, 0.0));
          c["use_inter_area_resize"_] = Extra(extra, n_extra, 14, 1) != 0;
          c["gray_scale"_] = Extra(extra, n_extra, 15, 1) != 0;
          c["base_path"_] = std::string("/synthetic");
        });
  } catch (const std::exception& e) {
    std::cerr << "orc_create(Atari): " << e.what() << std::endl;
  }
  return nullptr;
}

}  // extern "C"
