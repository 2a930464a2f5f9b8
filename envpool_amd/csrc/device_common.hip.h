// Device-side helpers shared by every family kernel.
//
//  * Mt19937: the per-env `std::mt19937 gen_` of the reference
//    (envpool/core/env.h:78,109-117) kept in HBM as SoA `uint32 mt[624][N]`
//    plus `int mti[N]`, and the libstdc++-11 distributions the env bodies use
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
  uint32_t* mt;  // base of this env's column: mt[j * n]
  int n;
  int idx;
  int idx0;
  int* idx_slot;

  __device__ Mt19937(const CommonDev& c, int e)
      : mt(c.mt + e), n(c.n), idx_slot(c.mti + e) {
    idx = idx0 = *idx_slot;
  }
  __device__ void Commit() {
    if (idx != idx0) *idx_slot = idx;
  }
  __device__ uint32_t& At(int j) { return mt[(size_t)j * n]; }

  __device__ void Twist() {
    const uint32_t upper = 0x80000000u, lower = 0x7fffffffu;
    uint32_t cur = At(0);
    const uint32_t first = cur;
    for (int k = 0; k < 624 - 397; ++k) {
      uint32_t nxt = At(k + 1);
      uint32_t y = (cur & upper) | (nxt & lower);
      At(k) = At(k + 397) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      cur = nxt;
    }
    for (int k = 624 - 397; k < 623; ++k) {
      uint32_t nxt = At(k + 1);
      uint32_t y = (cur & upper) | (nxt & lower);
      At(k) = At(k - 227) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      cur = nxt;
    }
    (void)first;
    uint32_t y = (cur & upper) | (At(0) & lower);
    At(623) = At(396) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    idx = 0;
  }

  __device__ uint32_t Next() {
    if (idx >= 624) Twist();
    uint32_t y = At(idx++);
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }

  // std::generate_canonical<double, 53>: random.tcc:3348-3380
  __device__ double Canonical() {
#pragma clang fp contract(off)
    double sum = 0.0, tmp = 1.0;
    sum += (double)Next() * tmp;
    tmp *= 4294967296.0;
    sum += (double)Next() * tmp;
    tmp *= 4294967296.0;
    double ret = sum / tmp;
    if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;  // nextafter(1.0, 0.0)
    return ret;
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
