// placeholder: replaced by the Atari post-process kernel (K4)
#include "engine.h"
extern "C" {
int epa_atari_post_create(int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, epa_atari_post**) { return EPA_ERR_RUNTIME; }
int epa_atari_post_destroy(epa_atari_post*) { return EPA_ERR_RUNTIME; }
int epa_atari_post_push(epa_atari_post*, const int32_t*, int32_t, const uint8_t*, const uint8_t*, uint8_t*) { return EPA_ERR_RUNTIME; }
int epa_atari_post_push_device(epa_atari_post*, const int32_t*, int32_t, const uint8_t*, const uint8_t*, uint8_t*) { return EPA_ERR_RUNTIME; }
void* epa_atari_post_stream(epa_atari_post*) { return nullptr; }
}
