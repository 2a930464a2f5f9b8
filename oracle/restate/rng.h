/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Plain-C restatement of the random machinery every reference env owns
 * (envpool/core/env.h:78,109-117: `std::mt19937 gen_(seed_)`, seed_ =
 * seed + env_id or env_seed[env_id]) and of the libstdc++ (gcc 11)
 * distributions the env bodies call.  The reference does not vendor these:
 * they come from the C++ standard library it is built with, so the algorithms
 * are restated from /usr/include/c++/11/bits:
 *   mt19937             ISO C++ [rand.eng.mers] (32-bit MT, tempering consts)
 *   generate_canonical  random.tcc:3348-3380  (2 draws, sum in double, /2^64)
 *   uniform_real        random.h `(canonical * (b - a)) + a`
 *   uniform_int         uniform_int_dist.h:240-268,300-307 (Lemire, 64-bit)
 *   normal              random.tcc:1803-1835  (Marsaglia polar, saved value)
 * Must be compiled with -ffp-contract=off (the reference x86-64 build has no
 * FMA contraction).
 */
#ifndef ORACLE_RESTATE_RNG_H_
#define ORACLE_RESTATE_RNG_H_
#include <math.h>
#include <stdint.h>

typedef struct {
  uint32_t mt[624];
  int idx;
} orc_mt19937;

static inline void orc_mt_seed(orc_mt19937* g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i) {
    uint32_t x = g->mt[i - 1];
    g->mt[i] = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
  }
  g->idx = 624;
}

static inline void orc_mt_twist(orc_mt19937* g) {
  const uint32_t upper = 0x80000000u, lower = 0x7fffffffu;
  uint32_t* mt = g->mt;
  for (int k = 0; k < 624 - 397; ++k) {
    uint32_t y = (mt[k] & upper) | (mt[k + 1] & lower);
    mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  for (int k = 624 - 397; k < 623; ++k) {
    uint32_t y = (mt[k] & upper) | (mt[k + 1] & lower);
    mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  uint32_t y = (mt[623] & upper) | (mt[0] & lower);
  mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  g->idx = 0;
}

static inline uint32_t orc_mt_next(orc_mt19937* g) {
  if (g->idx >= 624) orc_mt_twist(g);
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* std::generate_canonical<double, 53>(mt19937) */
static inline double orc_canonical(orc_mt19937* g) {
  double sum = 0.0, tmp = 1.0;
  sum += (double)orc_mt_next(g) * tmp;
  tmp *= 4294967296.0;
  sum += (double)orc_mt_next(g) * tmp;
  tmp *= 4294967296.0;
  double ret = sum / tmp;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret;
}

/* std::uniform_real_distribution<double>(a, b)(gen) */
static inline double orc_uniform_real(orc_mt19937* g, double a, double b) {
  return (orc_canonical(g) * (b - a)) + a;
}

/* std::uniform_int_distribution<int>(a, b)(gen), range < 2^32 - 1 */
static inline int orc_uniform_int(orc_mt19937* g, int a, int b) {
  uint32_t range = (uint32_t)b - (uint32_t)a + 1u;
  uint64_t product = (uint64_t)orc_mt_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)orc_mt_next(g) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (int)((uint32_t)(product >> 32) + (uint32_t)a);
}

/* std::normal_distribution<double>: the saved value lives in the
 * distribution object, i.e. it persists across episodes. */
typedef struct {
  double saved;
  int saved_available;
} orc_normal_state;

static inline double orc_normal(orc_mt19937* g, orc_normal_state* st,
                                double mean, double stddev) {
  double ret;
  if (st->saved_available) {
    st->saved_available = 0;
    ret = st->saved;
  } else {
    double x, y, r2;
    do {
      x = 2.0 * orc_canonical(g) - 1.0;
      y = 2.0 * orc_canonical(g) - 1.0;
      r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    double mult = sqrt(-2 * log(r2) / r2);
    st->saved = x * mult;
    st->saved_available = 1;
    ret = y * mult;
  }
  return ret * stddev + mean;
}

#endif /* ORACLE_RESTATE_RNG_H_ */
