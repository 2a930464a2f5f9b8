// TEST INFRASTRUCTURE (not product): the synthetic console of synth_ale.h behind the
// emulator plugin ABI (include/envpool_amd_emulator.h).  Built into
// tests/synth_ale/libsynth_ale.so by tests/atari_util.py; the product's Atari pool loads
// it like it would load the real ALE adapter (config key `emulator_lib`).
#include <string>

#include "../../include/envpool_amd_emulator.h"
#include "synth_ale.h"

namespace {
thread_local std::string g_err;

void* Create(const epa_emulator_config* cfg) {
  auto* c = new synth_ale::Console();
  c->SetSeed(cfg->random_seed);  // setInt / setFloat come before loadROM (atari_env.h:135-138)
  c->SetRepeatProb(cfg->repeat_action_probability);
  if (!c->Load(cfg->rom_path ? cfg->rom_path : "")) {
    g_err = std::string("synth_ale: cannot load ROM ") + (cfg->rom_path ? cfg->rom_path : "(null)");
    delete c;
    return nullptr;
  }
  if (cfg->mode >= 0) c->SetMode(cfg->mode);
  if (cfg->difficulty >= 0) c->SetDifficulty(cfg->difficulty);
  return c;
}
void Destroy(void* h) { delete static_cast<synth_ale::Console*>(h); }
int32_t ActionSet(void* h, int32_t full, int32_t* codes, int32_t cap) {
  auto* c = static_cast<synth_ale::Console*>(h);
  const std::vector<int> v = full ? c->LegalSet() : c->MinimalSet();
  for (int i = 0; i < (int)v.size() && i < cap; ++i) codes[i] = v[i];
  return (int32_t)v.size();
}
void ResetGame(void* h) { static_cast<synth_ale::Console*>(h)->ResetGame(); }
int32_t Act(void* h, int32_t a) { return static_cast<synth_ale::Console*>(h)->Act(a); }
int32_t GameOver(void* h) { return static_cast<synth_ale::Console*>(h)->GameOver() ? 1 : 0; }
int32_t Lives(void* h) { return static_cast<synth_ale::Console*>(h)->Lives(); }
const uint8_t* Screen(void* h) { return static_cast<synth_ale::Console*>(h)->Screen(); }
const uint8_t* Ram(void* h) { return static_cast<synth_ale::Console*>(h)->Ram(); }
void Palette(void*, uint8_t gray[256], uint8_t rgb[256][3]) { synth_ale::Console::Palette(gray, rgb); }
const char* LastError() { return g_err.c_str(); }

const epa_emulator_api kApi = {EPA_EMULATOR_ABI, Create, Destroy, ActionSet, ResetGame, Act,
                               GameOver, Lives, Screen, Ram, Palette, LastError};
}  // namespace

extern "C" const epa_emulator_api* epa_emulator_get_api(void) { return &kApi; }
