// TEST INFRASTRUCTURE (not product): a deterministic synthetic game console with the
// observable surface of ALE that AtariEnv uses (envpool/atari/atari_env.h): palette-indexed
// 210x160 screen, 128 bytes of RAM, lives, rewards of both signs and magnitudes > 1,
// game over, a minimal and a legal action set (with or without FIRE), sticky actions.
// ALE 0.11.2 and its ROMs are un-vendored (envpool/workspace0.bzl:239-283) and absent
// offline, so BOTH sides of the Atari parity tests run on this console:
//   * oracle/_ref: the reference's own atari_env.h compiled in place, with
//     oracle/ref_shims/ale_interface.hpp wrapping this class as ale::ALEInterface;
//   * the product: tests/synth_ale/plugin.cc exposes it through include/envpool_amd_emulator.h.
// The game itself is arbitrary; what matters is that every call the reference makes has
// state-dependent, seed-dependent, frame-dependent behaviour so that a wrong call order,
// a missed frame or a wrong palette shows up in the outputs.
#ifndef TESTS_SYNTH_ALE_SYNTH_ALE_H_
#define TESTS_SYNTH_ALE_SYNTH_ALE_H_

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace synth_ale {

constexpr int kH = 210, kW = 160, kRam = 128;

class Console {
 public:
  // rom name (file stem of the path): "synth_fire*" => minimal action set contains FIRE (1),
  // anything else => it does not.  "*_short" ends an episode after 60 frames.
  bool Load(const std::string& path) {
    std::size_t s = path.find_last_of('/');
    std::string stem = s == std::string::npos ? path : path.substr(s + 1);
    std::size_t d = stem.find_last_of('.');
    if (d != std::string::npos) stem = stem.substr(0, d);
    if (stem.rfind("synth", 0) != 0) return false;  // "ROM not found"
    has_fire_ = stem.find("fire") != std::string::npos;
    frame_limit_ = stem.find("short") != std::string::npos ? 60 : 4000;
    loaded_ = true;
    ResetGame();
    return true;
  }
  void SetSeed(int seed) { rng_ = 0x9E3779B97F4A7C15ull ^ (std::uint64_t)(std::uint32_t)seed * 0xD1342543DE82EF95ull; }
  void SetRepeatProb(float p) { repeat_ = p; }
  void SetMode(int m) { mode_ = m; }
  void SetDifficulty(int d) { difficulty_ = d; }
  std::vector<int> MinimalSet() const {
    return has_fire_ ? std::vector<int>{0, 1, 3, 4, 11, 12} : std::vector<int>{0, 2, 5};
  }
  std::vector<int> LegalSet() const {
    std::vector<int> v;
    for (int i = 0; i < 18; ++i) v.push_back(i);
    return v;
  }
  void ResetGame() {
    frame_ = 0;
    lives_ = 3;
    over_ = false;
    paddle_ = 80;
    bx_ = 20 + (int)(Next() % 120);
    by_ = 30;
    vx_ = (Next() & 1) ? 2 : -2;
    vy_ = 3;
    last_action_ = 0;
    score_ = 0;
    Draw();
  }
  int Act(int action) {
    // sticky actions, ALE style: with probability `repeat` the previous action is kept
    if (repeat_ > 0.0f && (float)((Next() >> 11) & 0xFFFFFF) / 16777216.0f < repeat_) {
      action = last_action_;
    }
    last_action_ = action;
    if (over_) return 0;
    ++frame_;
    int reward = 0;
    const int dir = (action == 3 || action == 11 || action == 2) ? 1 : ((action == 4 || action == 12 || action == 5) ? -1 : 0);
    paddle_ += 4 * dir * (1 + (mode_ > 0 ? 1 : 0));
    if (paddle_ < 8) paddle_ = 8;
    if (paddle_ > kW - 8) paddle_ = kW - 8;
    bx_ += vx_;
    by_ += vy_;
    if (bx_ < 2 || bx_ > kW - 3) {
      vx_ = -vx_;
      bx_ += 2 * vx_;
    }
    if (by_ < 20) {
      vy_ = -vy_;
      by_ = 20;
      if ((Next() & 3) == 0) reward += 5;  // bonus: |reward| > 1 exercises reward_clip
    }
    if (by_ >= 190) {
      const int w = 10 - 2 * (difficulty_ > 0 ? 1 : 0);
      if (bx_ >= paddle_ - w && bx_ <= paddle_ + w) {
        vy_ = -vy_;
        by_ = 189;
        reward += 1;
        if (action == 1 || action == 11 || action == 12) reward += 1;  // FIRE on the hit
      } else {
        reward -= 1 + (int)(Next() & 1);
        --lives_;
        bx_ = 20 + (int)(Next() % 120);
        by_ = 30;
        if (lives_ == 0) over_ = true;
      }
    }
    if (frame_ >= frame_limit_) over_ = true;
    score_ += reward;
    Draw();
    return reward;
  }
  bool GameOver() const { return over_; }
  int Lives() const { return lives_; }
  const std::uint8_t* Screen() const { return screen_; }
  const std::uint8_t* Ram() const { return ram_; }
  // NTSC-like palette: arbitrary but non-monotonic in the index
  static void Palette(std::uint8_t gray[256], std::uint8_t rgb[256][3]) {
    for (int i = 0; i < 256; ++i) {
      const int r = (i * 37 + 11) & 255, g = (i * 101 + 7) & 255, b = (255 - i * 13) & 255;
      rgb[i][0] = (std::uint8_t)r;
      rgb[i][1] = (std::uint8_t)g;
      rgb[i][2] = (std::uint8_t)b;
      gray[i] = (std::uint8_t)((r * 77 + g * 150 + b * 29 + 128) >> 8);
    }
  }

 private:
  std::uint64_t Next() {
    rng_ = rng_ * 6364136223846793005ull + 1442695040888963407ull;
    return rng_ >> 17;
  }
  void Draw() {
    // background stripes that flicker with the frame parity (so max-pooling two frames
    // differs from taking the last one), score bar, ball, paddle
    const std::uint8_t par = (std::uint8_t)(frame_ & 1);
    for (int y = 0; y < kH; ++y) {
      const std::uint8_t base = (std::uint8_t)(((y >> 3) * 6 + (par ? 2 : 0)) & 255);
      std::memset(screen_ + y * kW, base, kW);
      if ((y & 7) == (int)(frame_ % 8)) {
        for (int x = (y * 7) % 16; x < kW; x += 16) screen_[y * kW + x] = (std::uint8_t)(200 + (x & 31));
      }
    }
    const int bar = ((score_ % 40) + 40) % 40;
    for (int x = 0; x < 4 * bar && x < kW; ++x) screen_[4 * kW + x] = screen_[5 * kW + x] = 0x46;
    for (int l = 0; l < lives_; ++l) {
      for (int x = 0; x < 6; ++x) screen_[10 * kW + 140 + 7 * l + x] = 0x1A;
    }
    for (int dy = -2; dy <= 2; ++dy) {
      for (int dx = -2; dx <= 2; ++dx) {
        const int y = by_ + dy, x = bx_ + dx;
        if (y >= 0 && y < kH && x >= 0 && x < kW) screen_[y * kW + x] = 0x0E;
      }
    }
    for (int y = 194; y < 198; ++y) {
      for (int x = paddle_ - 8; x < paddle_ + 8; ++x) {
        if (x >= 0 && x < kW) screen_[y * kW + x] = (std::uint8_t)(0x90 + (last_action_ & 15));
      }
    }
    for (int i = 0; i < kRam; ++i) ram_[i] = (std::uint8_t)((i * 31 + frame_ * (i & 7) + score_) & 255);
    ram_[0] = (std::uint8_t)lives_;
    ram_[1] = (std::uint8_t)bx_;
    ram_[2] = (std::uint8_t)by_;
    ram_[3] = (std::uint8_t)paddle_;
    ram_[4] = (std::uint8_t)(frame_ & 255);
    ram_[5] = (std::uint8_t)last_action_;
  }

  bool loaded_{false}, has_fire_{true}, over_{false};
  int frame_limit_{4000}, mode_{-1}, difficulty_{-1};
  float repeat_{0.0f};
  std::uint64_t rng_{0x1234567ull};
  int frame_{0}, lives_{3}, paddle_{80}, bx_{0}, by_{0}, vx_{2}, vy_{3}, last_action_{0}, score_{0};
  std::uint8_t screen_[kH * kW]{};
  std::uint8_t ram_[kRam]{};
};

}  // namespace synth_ale

#endif  // TESTS_SYNTH_ALE_SYNTH_ALE_H_
