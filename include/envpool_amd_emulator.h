/* envpool_amd_emulator.h -- plugin ABI of the host-side Atari emulator.
 *
 * The reference's AtariEnv (envpool/atari/atari_env.h:97-293) owns an
 * `ale::ALEInterface` per env and calls a dozen of its methods.  The north star
 * keeps ALE on the host; this engine therefore talks to the emulator through
 * the small C table below, filled by a plugin shared library:
 *
 *   integration/ale_adapter/ale_adapter.cc   the real thing: ALE 0.11.2
 *                                            (envpool/workspace0.bzl:239-283; un-vendored,
 *                                            built by the deployer against libale)
 *   tests/synth_ale/plugin.cc                a deterministic synthetic console used by
 *                                            the test-suite and the benchmarks (ALE and
 *                                            its ROMs are not available offline)
 *
 * A plugin exports ONE symbol:
 *     const epa_emulator_api* epa_emulator_get_api(void);
 * and the pool loads it with dlopen (config key `emulator_lib`).  Every entry
 * cites the ALE call of atari_env.h it stands for.  All functions are called
 * from the pool's worker threads; calls on one handle are never concurrent.
 */
#ifndef ENVPOOL_AMD_EMULATOR_H_
#define ENVPOOL_AMD_EMULATOR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPA_EMULATOR_ABI 1
#define EPA_EMULATOR_SCREEN_H 210 /* kRawHeight, atari_env.h:99 */
#define EPA_EMULATOR_SCREEN_W 160 /* kRawWidth,  atari_env.h:100 */
#define EPA_EMULATOR_RAM 128      /* "info:ram", atari_env.h:72 */

typedef struct epa_emulator_config {
  const char* rom_path;            /* GetRomPath(base_path, task), atari_env.h:43-48 */
  int32_t random_seed;             /* setInt("random_seed", seed_)            :137 */
  float repeat_action_probability; /* setFloat("repeat_action_probability")   :135 */
  int32_t mode;                    /* setMode if >= 0                          :140 */
  int32_t difficulty;              /* setDifficulty if >= 0                    :143 */
} epa_emulator_config;

typedef struct epa_emulator_api {
  int32_t abi; /* EPA_EMULATOR_ABI */
  /* new ALEInterface + set* + loadROM (atari_env.h:121-145); NULL on failure, in which
   * case last_error() describes it */
  void* (*create)(const epa_emulator_config* cfg);
  void (*destroy)(void* h);
  /* getLegalActionSet (full != 0) / getMinimalActionSet (:146-150): writes up to `cap`
   * ALE action codes, returns the size of the set */
  int32_t (*action_set)(void* h, int32_t full, int32_t* codes, int32_t cap);
  void (*reset_game)(void* h);                 /* :172, :179 */
  int32_t (*act)(void* h, int32_t action_code); /* reward_t act(Action)  :177, :184, :209 */
  int32_t (*game_over)(void* h);               /* :170, :178, :210 */
  int32_t (*lives)(void* h);                   /* :197, :228-237 */
  /* getScreen().getArray(): 210 x 160 palette indices, valid until the next act / reset_game
   * (:186, :212) */
  const uint8_t* (*screen)(void* h);
  const uint8_t* (*ram)(void* h);              /* getRAM().array(), 128 bytes (:278-279) */
  /* theOSystem->colourPalette(): the 256-entry tables behind applyPaletteGrayscale /
   * applyPaletteRGB (:189-194).  Constant after create; the pool applies them on the GPU. */
  void (*palette)(void* h, uint8_t gray[256], uint8_t rgb[256][3]);
  const char* (*last_error)(void);
} epa_emulator_api;

typedef const epa_emulator_api* (*epa_emulator_get_api_fn)(void);

#ifdef __cplusplus
}
#endif
#endif /* ENVPOOL_AMD_EMULATOR_H_ */
