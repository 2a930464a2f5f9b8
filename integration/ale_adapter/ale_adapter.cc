// Emulator plugin for the real Arcade Learning Environment (ALE 0.11.2, the version the
// reference pins: envpool/workspace0.bzl:239-283).  Implements include/envpool_amd_emulator.h
// with exactly the ALE calls envpool/atari/atari_env.h makes (cited per entry).
//
// ALE and its ROMs are un-vendored third parties and absent offline.  A deployer builds this file
// next to an ALE installation:
//     g++ -std=c++17 -O2 -fPIC -shared integration/ale_adapter/ale_adapter.cc
//         -I<ale>/include/ale -L<ale>/lib -lale -o libepa_ale.so
// The repository's own build compiles THE SAME SOURCE against the ALE-API shim it owns
// (oracle/ref_shims_atari/ale_interface.hpp: the members atari_env.h touches, over the synthetic
// console) -> integration/_build/libepa_ale_over_shim.so (`make -C integration adapter`), and
// tests/test_gpu_atari_env.py runs the reference-generated fixtures through that plugin.
// and points the pool at it: envpool_amd.make("Pong-v5", ..., emulator_lib="/path/libepa_ale.so",
// base_path=<dir holding atari/roms/pong.bin>)   (or EPA_ATARI_EMULATOR_LIB in the environment).
#include <cstring>
#include <memory>
#include <string>

#include "../../include/envpool_amd_emulator.h"
#include "ale_interface.hpp"

namespace {

thread_local std::string g_err;

struct Emu {
  std::unique_ptr<ale::ALEInterface> env;
};

const bool kQuiet = [] {  // TurnOffVerbosity, atari_env.h:36-41
  ale::Logger::setMode(ale::Logger::Error);
  return true;
}();

void* Create(const epa_emulator_config* cfg) {
  try {
    auto e = std::make_unique<Emu>();
    e->env = std::make_unique<ale::ALEInterface>();
    e->env->setFloat("repeat_action_probability", cfg->repeat_action_probability);  // :135
    e->env->setInt("random_seed", cfg->random_seed);                                // :137
    e->env->loadROM(cfg->rom_path);                                                 // :138
    if (cfg->mode >= 0) e->env->setMode(cfg->mode);                                 // :140
    if (cfg->difficulty >= 0) e->env->setDifficulty(cfg->difficulty);               // :143
    return e.release();
  } catch (const std::exception& ex) {
    g_err = ex.what();
    return nullptr;
  }
}
void Destroy(void* h) { delete static_cast<Emu*>(h); }
int32_t ActionSet(void* h, int32_t full, int32_t* codes, int32_t cap) {  // :146-150
  auto& env = *static_cast<Emu*>(h)->env;
  const ale::ActionVect v = full ? env.getLegalActionSet() : env.getMinimalActionSet();
  for (std::size_t i = 0; i < v.size() && (int32_t)i < cap; ++i) codes[i] = static_cast<int32_t>(v[i]);
  return static_cast<int32_t>(v.size());
}
void ResetGame(void* h) { static_cast<Emu*>(h)->env->reset_game(); }  // :172
int32_t Act(void* h, int32_t a) {                                      // :209
  return static_cast<int32_t>(static_cast<Emu*>(h)->env->act(static_cast<ale::Action>(a)));
}
int32_t GameOver(void* h) { return static_cast<Emu*>(h)->env->game_over() ? 1 : 0; }
int32_t Lives(void* h) { return static_cast<Emu*>(h)->env->lives(); }
const uint8_t* Screen(void* h) { return static_cast<Emu*>(h)->env->getScreen().getArray(); }  // :186
const uint8_t* Ram(void* h) { return static_cast<Emu*>(h)->env->getRAM().array(); }           // :278
// the tables behind applyPaletteGrayscale / applyPaletteRGB (:189-194): obtained by running
// the identity index ramp through ALE's own palette, so NTSC / PAL / SECAM ROMs are handled
// by ALE, not restated
void Palette(void* h, uint8_t gray[256], uint8_t rgb[256][3]) {
  auto& pal = static_cast<Emu*>(h)->env->theOSystem->colourPalette();
  uint8_t ramp[256];
  for (int i = 0; i < 256; ++i) ramp[i] = static_cast<uint8_t>(i);
  pal.applyPaletteGrayscale(gray, ramp, 256);
  pal.applyPaletteRGB(&rgb[0][0], ramp, 256);
}
const char* LastError() { return g_err.c_str(); }

const epa_emulator_api kApi = {EPA_EMULATOR_ABI, Create, Destroy, ActionSet, ResetGame, Act,
                               GameOver, Lives, Screen, Ram, Palette, LastError};
}  // namespace

extern "C" const epa_emulator_api* epa_emulator_get_api(void) { return &kApi; }
