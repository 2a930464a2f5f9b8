// TEST INFRASTRUCTURE — shim, not product code.
// Stands in for ALE 0.11.2's <ale_interface.hpp> (un-vendored third party, fetched by Bazel:
// envpool/workspace0.bzl:239-283) so that the reference's OWN envpool/atari/atari_env.h
// compiles in place (oracle/_ref/libref_atari.so).  The emulated machine is the synthetic
// console of tests/synth_ale/synth_ale.h; only the members atari_env.h touches exist.
#ifndef ORACLE_REF_SHIMS_ATARI_ALE_INTERFACE_HPP_
#define ORACLE_REF_SHIMS_ATARI_ALE_INTERFACE_HPP_

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../tests/synth_ale/synth_ale.h"

namespace ale {

using reward_t = int;
enum Action : int { PLAYER_A_NOOP = 0, PLAYER_A_FIRE = 1 };
using ActionVect = std::vector<Action>;

struct Logger {
  enum mode { Info = 0, Warning = 1, Error = 2 };
  static void setMode(mode) {}
};

class ColourPalette {
 public:
  ColourPalette() { synth_ale::Console::Palette(gray_, rgb_); }
  void applyPaletteGrayscale(std::uint8_t* dst, const std::uint8_t* src, std::size_t n) const {
    for (std::size_t i = 0; i < n; ++i) dst[i] = gray_[src[i]];
  }
  void applyPaletteRGB(std::uint8_t* dst, const std::uint8_t* src, std::size_t n) const {
    for (std::size_t i = 0; i < n; ++i) {
      dst[3 * i] = rgb_[src[i]][0];
      dst[3 * i + 1] = rgb_[src[i]][1];
      dst[3 * i + 2] = rgb_[src[i]][2];
    }
  }

 private:
  std::uint8_t gray_[256], rgb_[256][3];
};
struct OSystem {
  ColourPalette pal;
  ColourPalette& colourPalette() { return pal; }
};
class ALEScreen {
 public:
  explicit ALEScreen(const synth_ale::Console* c) : c_(c) {}
  std::uint8_t* getArray() const { return const_cast<std::uint8_t*>(c_->Screen()); }

 private:
  const synth_ale::Console* c_;
};
class ALERAM {
 public:
  explicit ALERAM(const synth_ale::Console* c) : c_(c) {}
  const std::uint8_t* array() const { return c_->Ram(); }
  std::size_t size() const { return synth_ale::kRam; }

 private:
  const synth_ale::Console* c_;
};

class ALEInterface {
 public:
  std::unique_ptr<OSystem> theOSystem{new OSystem()};
  ALEInterface() : screen_(&c_), ram_(&c_) {}
  void setFloat(const std::string& key, float v) {
    if (key == "repeat_action_probability") c_.SetRepeatProb(v);
  }
  void setInt(const std::string& key, int v) {
    if (key == "random_seed") c_.SetSeed(v);
  }
  void loadROM(const std::string& path) {
    if (!c_.Load(path)) throw std::runtime_error("synth_ale: cannot load ROM " + path);
  }
  void setMode(int m) { c_.SetMode(m); }
  void setDifficulty(int d) { c_.SetDifficulty(d); }
  ActionVect getLegalActionSet() { return Conv(c_.LegalSet()); }
  ActionVect getMinimalActionSet() { return Conv(c_.MinimalSet()); }
  void reset_game() { c_.ResetGame(); }
  reward_t act(Action a) { return c_.Act(static_cast<int>(a)); }
  bool game_over() const { return c_.GameOver(); }
  int lives() const { return c_.Lives(); }
  const ALEScreen& getScreen() const { return screen_; }
  const ALERAM& getRAM() const { return ram_; }

 private:
  static ActionVect Conv(const std::vector<int>& v) {
    ActionVect r;
    for (int a : v) r.push_back(static_cast<Action>(a));
    return r;
  }
  synth_ale::Console c_;
  ALEScreen screen_;
  ALERAM ram_;
};

}  // namespace ale

#endif  // ORACLE_REF_SHIMS_ATARI_ALE_INTERFACE_HPP_
