// Device-side helpers shared by every family kernel.
//
//  * Mt19937: the per-env `std::mt19937 gen_` of the reference
//    (envpool/core/env.h:78,109-117) kept in HBM as 624 words per env (`uint32 mt[624][N]`, or in
//    tiles of 2^k words per env, CommonDev::mt_shift) plus the position `int mti[N]`, regenerated one
//    word per draw, and the libstdc++-11 distributions the env bodies use
//    (generate_canonical / uniform_real / uniform_int(Lemire) / normal(polar)),
//    bit-exact with /usr/include/c++/11/bits/{random.tcc,uniform_int_dist.h}.
//    The floating-point distributions carry `#pragma clang fp contract(off)`
//    (the reference's x86-64 build does not fuse a*b+c) so they stay bit-exact
//    even in translation units built with fast contraction (mujoco_gym.hip).
//  * WriteCommon: the bookkeeping Env::Allocate does for every returned row
//    (envpool/core/env.h:224-256).
#ifndef ENVPOOL_AMD_CSRC_DEVICE_COMMON_HIP_H_
#define ENVPOOL_AMD_CSRC_DEVICE_COMMON_HIP_H_

#include <hip/hip_runtime.h>

#include <cstdint>

#include "engine.h"

namespace epa {

// indices of the common state keys (envpool/core/env_spec.h:37-43)
enum : int {
  kKeyEnvId = 0,
  kKeyPlayersEnvId = 1,
  kKeyElapsedStep = 2,
  kKeyDone = 3,
  kKeyReward = 4,
  kKeyDiscount = 5,
  kKeyStepType = 6,
  kKeyTrunc = 7,
  kKeyEnv0 = 8,
};

struct StepArgs {
  const int* ids;  // nullptr => row i is local env i
  int k;
  int force_reset;
  int max_episode_steps;
  int id_offset;
};

struct Mt19937 {
  uint32_t* mt;  // base of the pool's generator words (layout: MtAt)
  int e;         // this env's column
  int n;
  int sh;        // log2 of the tile width (CommonDev::mt_shift)
  int idx;
  int idx0;
  int* idx_slot;

  __device__ Mt19937(const CommonDev& c, int env)
      : mt(c.mt), e(env), n(c.n), sh(c.mt_shift), idx_slot(c.mti + env) {
    idx = idx0 = *idx_slot;
  }
  __device__ void Commit() {
    if (idx != idx0) *idx_slot = idx;
  }
  // Word j of this env.  Two layouts (CommonDev::mt_shift): 0 = the structure of arrays mt[j][N] -- every env
  // draws the same word in the same launch, a draw is one coalesced 4-byte column; 4 = tiles of 16 consecutive
  // words of ONE env are contiguous (tile t of all envs is one [N][16] slab) -- envs that reset at their own
  // times: the 8 .. 60 draws of a reset stay inside 1 .. 4 64-byte sectors instead of one sector per word.
  __device__ uint32_t& At(int j) {
    return mt[((((size_t)(j >> sh)) * n + e) << sh) | (size_t)(j & ((1 << sh) - 1))];
  }
  __device__ static uint32_t Twist1(uint32_t cur, uint32_t nxt, uint32_t partner) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return partner ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  // The 16 words [16 t, 16 t + 16) of the next block, in place (tiled layout): one 64-byte tile in and out as
  // uint4, the neighbour word of the next tile, 16 partner words out of two other tiles.
  __device__ void RegenTile16() { RegenTile16At(idx); }
  __device__ void RegenTile16At(int idx) {  // idx: the tile's first word, a multiple of 16
    const int t = idx >> 4;
    uint4* own = reinterpret_cast<uint4*>(&At(idx));
    uint32_t w[17];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 x = own[q];
      w[4 * q] = x.x;
      w[4 * q + 1] = x.y;
      w[4 * q + 2] = x.z;
      w[4 * q + 3] = x.w;
    }
    // the word after the tile: still the old block's, except behind the last tile, where it is word 0 of the
    // block being generated (libstdc++ reads _M_x[0] there too)
    w[16] = At(t == 38 ? 0 : idx + 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = idx + r;
      // partner k + 397 (old block) for k < 227, k - 227 (already regenerated: an earlier tile, or -- in the
      // tile that holds word 227 -- never one of this tile's own words) otherwise
      w[r] = Twist1(w[r], w[r + 1], At(k >= 227 ? k - 227 : k + 397));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) own[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
  }

  // std::mt19937::operator(): the generator is regenerated LAZILY, when a word is about to be consumed -- one word
  // at a time in the [624][N] layout, one 16-word tile at a time in the tiled one.  Word i of the next block is
  // s[i+397] ^ twist(s[i], s[i+1]) with the indices mod 624; at the moment word i (or its tile) is reached the
  // words after it are still the old block's and the words before it the new block's, which is exactly what
  // libstdc++'s block-wise _M_gen_rand reads for that word (bits/random.tcc:396-440): the same sequence, without
  // the 624-iteration loop that a single lane of a wave used to run while the other 63 waited (one wave per SIMD
  // in the MuJoCo kernels: the whole SIMD).  `idx` is the position of the next word, 0 .. 623, for ever.
  __device__ uint32_t Next() {
    const int i = idx;
    uint32_t y;
    if (sh == 0) {
      const int i1 = i == 623 ? 0 : i + 1;
      uint32_t& cur = At(i);
      y = Twist1(cur, At(i1), At(i >= 227 ? i - 227 : i + 397));
      cur = y;
    } else {
      if ((i & 15) == 0) RegenTile16();
      y = At(i);
    }
    idx = i == 623 ? 0 : i + 1;
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }

  // The next K outputs in ONE burst (K <= 16).  Tiled layout: Next() costs a dependent global round trip per word
  // (the compiler cannot move a load across the conditional tile regeneration in front of it), and a reset lane
  // runs them one after the other while the other 63 lanes of its wave wait; here the one tile boundary the K words
  // can cross is regenerated first (regenerating a tile before the rest of the previous one is consumed reads
  // nothing that consumption writes) and the K words are then K independent loads.  Same words, same order.
  template <int K>
  __device__ void NextWords(uint32_t (&y)[K]) {
    static_assert(K >= 1 && K <= 16, "one tile boundary at most");
    if (sh == 0) {
#pragma unroll
      for (int r = 0; r < K; ++r) y[r] = Next();
      return;
    }
    const int i0 = idx;
    const int d = (16 - (i0 & 15)) & 15;  // words in front of the next tile start (0: i0 starts a tile)
    if (d < K) {
      const int p = i0 + d;
      RegenTile16At(p >= 624 ? p - 624 : p);  // (624 = 39 tiles: the wrap lands on a tile start)
    }
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const int p = i0 + r;
      y[r] = At(p >= 624 ? p - 624 : p);
    }
    idx = i0 + K >= 624 ? i0 + K - 624 : i0 + K;
#pragma unroll
    for (int r = 0; r < K; ++r) {
      uint32_t t = y[r];
      t ^= (t >> 11);
      t ^= (t << 7) & 0x9d2c5680u;
      t ^= (t << 15) & 0xefc60000u;
      t ^= (t >> 18);
      y[r] = t;
    }
  }

  // std::generate_canonical<double, 53>: random.tcc:3348-3380
  __device__ static double CanonicalOf(uint32_t w0, uint32_t w1) {
#pragma clang fp contract(off)
    double sum = 0.0, tmp = 1.0;
    sum += (double)w0 * tmp;
    tmp *= 4294967296.0;
    sum += (double)w1 * tmp;
    tmp *= 4294967296.0;
    double ret = sum / tmp;
    if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;  // nextafter(1.0, 0.0)
    return ret;
  }
  __device__ double Canonical() {
    const uint32_t w0 = Next();
    const uint32_t w1 = Next();
    return CanonicalOf(w0, w1);
  }
  // K draws of std::uniform_real_distribution<double>(a, b), in order, as one burst of 2 K words (K <= 8)
  template <int K>
  __device__ void UniformReals(double a, double b, double* out) {
#pragma clang fp contract(off)
    uint32_t w[2 * K];
    NextWords<2 * K>(w);
#pragma unroll
    for (int r = 0; r < K; ++r) out[r] = (CanonicalOf(w[2 * r], w[2 * r + 1]) * (b - a)) + a;
  }
  // std::uniform_real_distribution<double>(a, b)
  __device__ double UniformReal(double a, double b) {
#pragma clang fp contract(off)
    return (Canonical() * (b - a)) + a;
  }
  // std::uniform_int_distribution<int>(a, b): uniform_int_dist.h:240-268
  __device__ int UniformInt(int a, int b) {
    uint32_t range = (uint32_t)b - (uint32_t)a + 1u;
    uint64_t product = (uint64_t)Next() * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
      uint32_t threshold = (0u - range) % range;
      while (low < threshold) {
        product = (uint64_t)Next() * (uint64_t)range;
        low = (uint32_t)product;
      }
    }
    return (int)((uint32_t)(product >> 32) + (uint32_t)a);
  }
  // std::normal_distribution<double>: random.tcc:1803-1835.  `saved`/`avail`
  // live in the distribution object of the env => persistent per-env state.
  __device__ double Normal(double mean, double stddev, double* saved,
                           int* avail) {
#pragma clang fp contract(off)
    double ret;
    if (*avail) {
      *avail = 0;
      ret = *saved;
    } else {
      double x, y, r2;
      do {
        x = 2.0 * Canonical() - 1.0;
        y = 2.0 * Canonical() - 1.0;
        r2 = x * x + y * y;
      } while (r2 > 1.0 || r2 == 0.0);
      double mult = sqrt(-2 * log(r2) / r2);
      *saved = x * mult;
      *avail = 1;
      ret = y * mult;
    }
    return ret * stddev + mean;
  }
};

// Env::Allocate (envpool/core/env.h:224-256) for one output row.
__device__ inline void WriteCommon(const OutPtrs& out, int row, int global_id,
                                   int cur_step, bool done, float reward,
                                   int max_episode_steps) {
  ((int*)out.p[kKeyEnvId])[row] = global_id;
  ((int*)out.p[kKeyPlayersEnvId])[row] = global_id;
  ((int*)out.p[kKeyElapsedStep])[row] = cur_step;
  ((unsigned char*)out.p[kKeyDone])[row] = done ? 1 : 0;
  ((float*)out.p[kKeyReward])[row] = reward;
  ((float*)out.p[kKeyDiscount])[row] = done ? 0.0f : 1.0f;
  int step_type = 1;  // dm_env.StepType.MID
  if (cur_step == 0) {
    step_type = 0;  // FIRST
  } else if (done) {
    step_type = 2;  // LAST
  }
  ((int*)out.p[kKeyStepType])[row] = step_type;
  ((unsigned char*)out.p[kKeyTrunc])[row] =
      (done && cur_step >= max_episode_steps) ? 1 : 0;
}

}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_DEVICE_COMMON_HIP_H_
