// K2 — toy_text batched step kernels (one env per thread), bit-exact.
//
// Replaces, for the whole batch in one launch:
//   CatchEnv::{Reset,Step,WriteState}         envpool/toy_text/catch.h:62-93
//   FrozenLakeEnv::{Reset,Step}               envpool/toy_text/frozen_lake.h:74-111
//   TaxiEnv::{Reset,Step}                     envpool/toy_text/taxi.h:69-127
//   NChainEnv::{Reset,Step}                   envpool/toy_text/nchain.h:61-95
//   CliffWalkingEnv::{Reset,Step,SampleAction} envpool/toy_text/cliffwalking.h:63-111
//   BlackjackEnv::{Reset,Step,...}            envpool/toy_text/blackjack.h:65-152
// plus the runtime (async_envpool.h:118-132, env.h:184-256).
//
// Data layout (HBM): the whole discrete state of an env is packed into one
// int32 word (two for Blackjack) kept SoA `w0[N]`, `w1[N]`; the per-env
// std::mt19937 lives in CommonDev (device_common.hip.h).  All integer work;
// NChain's `uniform_real < 0.2` branch needs exact fp64 => -ffp-contract=off.
// HBM-bound: ~72 algorithmic bytes per env-step.
#include "device_common.hip.h"
#include "engine.h"

namespace epa {
namespace {

enum Kind : int {
  kCatch = 0,
  kFrozenLake,
  kTaxi,
  kNChain,
  kCliffWalking,
  kBlackjack
};

struct ToyDev {
  int* w0;
  int* w1;
};

struct ToyCfg {
  int size;         // FrozenLake
  int height, width;  // Catch
  int is_slippery;  // CliffWalking
  int natural, sab;  // Blackjack
};

// FrozenLake maps (frozen_lake.h:64-69) as bitmasks: bit (x*size+y)
// 4x4:  SFFF FHFH FFFH HFFG
__constant__ const unsigned long long kLake4Hole =
    (1ull << 5) | (1ull << 7) | (1ull << 11) | (1ull << 12);
__constant__ const unsigned long long kLake4Goal = (1ull << 15);
// 8x8: "SFFFFFFF","FFFFFFFF","FFFHFFFF","FFFFFHFF","FFFHFFFF","FHHFFFHF",
//      "FHFFHFHF","FFFHFFFG"
__constant__ const unsigned long long kLake8Hole =
    (1ull << 19) | (1ull << 29) | (1ull << 35) | (1ull << 41) | (1ull << 42) |
    (1ull << 46) | (1ull << 49) | (1ull << 52) | (1ull << 54) | (1ull << 59);
__constant__ const unsigned long long kLake8Goal = (1ull << 63);

// Taxi (taxi.h:62-66): map_[x][y+1]==':' / map_[x][y]==':' as bit tables.
//   "|:|::|","|:|::|","|::::|","||:|:|","||:|:|"   (columns 0..5)
__device__ inline bool TaxiColon(int x, int c) {
  const unsigned char rows[5] = {0b011010, 0b011010, 0b011110, 0b010100,
                                 0b010100};  // bit c set => map[x][c]==':'
  return (rows[x] >> c) & 1;
}
__device__ inline int TaxiLocX(int i) {
  const int v[4] = {0, 0, 4, 4};
  return v[i];
}
__device__ inline int TaxiLocY(int i) {
  const int v[4] = {0, 4, 0, 3};
  return v[i];
}
// loc_map_ (taxi.h:66): "0   1","     ","     ","     ","2  3 " -> -1 if ' '
__device__ inline int TaxiLocMap(int x, int y) {
  if (x == 0 && y == 0) return 0;
  if (x == 0 && y == 4) return 1;
  if (x == 4 && y == 0) return 2;
  if (x == 4 && y == 3) return 3;
  return -1;
}

// Blackjack hand folded to sum / any-ace / count(<=3) / first two cards:
// bits 0-9 sum, 10 ace, 11-12 count (saturating at 3), 13-16 card0, 17-20 card1
struct Hand {
  int sum, ace, cnt, c0, c1;
  __device__ static Hand Unpack(int w) {
    return {w & 1023, (w >> 10) & 1, (w >> 11) & 3, (w >> 13) & 15,
            (w >> 17) & 15};
  }
  __device__ int Pack() const {
    return sum | (ace << 10) | (cnt << 11) | (c0 << 13) | (c1 << 17);
  }
  __device__ void Push(int card) {  // player_.push_back(DrawCard())
    if (cnt == 0) c0 = card;
    if (cnt == 1) c1 = card;
    sum += card;
    if (card == 1) ace = 1;
    if (cnt < 3) ++cnt;
  }
  __device__ int SumHand() const {  // blackjack.h:123-132
    return (ace != 0 && sum + 10 <= 21) ? sum + 10 : sum;
  }
  __device__ int Score() const {  // :138-141
    int r = SumHand();
    return r > 21 ? 0 : r;
  }
  __device__ bool IsNatural() const {  // :143-146
    return cnt == 2 && ((c0 == 1 && c1 == 10) || (c0 == 10 && c1 == 1));
  }
};
__device__ inline int DrawCard(Mt19937& g) {  // blackjack.h:112
  int c = g.UniformInt(1, 13);
  return c < 10 ? c : 10;
}

template <int KIND>
__global__ __launch_bounds__(256) void ToyStepKernel(
    ToyDev dev, CommonDev cm, StepArgs a, const int* __restrict__ action,
    OutPtrs out, ToyCfg cfg) {
  __shared__ int sh_p1[256];
  __shared__ int sh_p2[256];
  for (int base = blockIdx.x * blockDim.x; base < a.k;
       base += gridDim.x * blockDim.x) {
    int row = base + threadIdx.x;
    bool active = row < a.k;
    int p1 = -1, p2 = -1;  // Catch one-hot positions
    if (active) {
      int e = a.ids ? a.ids[row] - a.id_offset : row;
      bool done = cm.done[e] != 0;
      int cur = cm.cur_step[e];
      bool reset = a.force_reset || done;  // async_envpool.h:127
      float reward = 0.0f;
      int w0 = 0, w1 = 0;
      if (!reset) {
        w0 = dev.w0[e];
        if constexpr (KIND == kBlackjack) w1 = dev.w1[e];
        ++cur;
      } else {
        cur = 0;
      }
      int act = reset ? 0 : action[row];
      Mt19937 g(cm, e);

      if constexpr (KIND == kCatch) {
        int x = w0 & 255, y = (w0 >> 8) & 255, paddle = (w0 >> 16) & 255;
        if (reset) {  // catch.h:62-68
          x = 0;
          y = g.UniformInt(0, cfg.width - 1);
          paddle = cfg.width / 2;
          done = false;
        } else {  // catch.h:70-85
          paddle += act - 1;
          if (paddle < 0) paddle = 0;
          if (paddle >= cfg.width) paddle = cfg.width - 1;
          if (++x == cfg.height - 1) {
            done = true;
            reward = y == paddle ? 1.0f : -1.0f;
          }
        }
        w0 = x | (y << 8) | (paddle << 16);
        p1 = x * cfg.width + y;
        p2 = (cfg.height - 1) * cfg.width + paddle;
      } else if constexpr (KIND == kFrozenLake) {
        int size = cfg.size;
        int x = w0 / size, y = w0 % size;
        if (reset) {  // frozen_lake.h:74-79
          x = y = 0;
          done = false;
        } else {  // frozen_lake.h:81-102
          done = cur >= a.max_episode_steps;
          act = (act + g.UniformInt(-1, 1) + 4) % 4;
          if (act == 0) {
            --y;
          } else if (act == 1) {
            ++x;
          } else if (act == 2) {
            ++y;
          } else {
            --x;
          }
          x = min(max(x, 0), size - 1);
          y = min(max(y, 0), size - 1);
          unsigned long long bit = 1ull << (x * size + y);
          unsigned long long hole = size != 8 ? kLake4Hole : kLake8Hole;
          unsigned long long goal = size != 8 ? kLake4Goal : kLake8Goal;
          if ((hole | goal) & bit) {
            done = true;
            reward = (goal & bit) ? 1.0f : 0.0f;
          }
        }
        w0 = x * size + y;
        ((int*)out.p[kKeyEnv0])[row] = w0;
      } else if constexpr (KIND == kTaxi) {
        int t = w0 & 3, s = (w0 >> 2) % 5, y = (w0 / 20) % 5, x = w0 / 100;
        if (reset) {  // taxi.h:69-77
          x = g.UniformInt(0, 4);
          y = g.UniformInt(0, 4);
          s = g.UniformInt(0, 3);
          t = g.UniformInt(0, 3);
          done = false;
        } else {  // taxi.h:79-119
          done = cur >= a.max_episode_steps;
          reward = -1.0f;
          if (act == 0) {
            if (x < 4) ++x;
          } else if (act == 1) {
            if (x > 0) --x;
          } else if (act == 2) {
            if (TaxiColon(x, y + 1)) ++y;
          } else if (act == 3) {
            if (TaxiColon(x, y)) --y;
          } else if (act == 4) {
            if (s < 4 && x == TaxiLocX(s) && y == TaxiLocY(s)) {
              s = 4;
            } else {
              reward = -10.0f;
            }
          } else {
            if (s == 4 && x == TaxiLocX(t) && y == TaxiLocY(t)) {
              s = t;
              done = true;
              reward = 20.0f;
            } else if (s == 4 && TaxiLocMap(x, y) >= 0) {
              s = TaxiLocMap(x, y);
            } else {
              reward = -10.0f;
            }
          }
        }
        w0 = ((x * 5 + y) * 5 + s) * 4 + t;
        ((int*)out.p[kKeyEnv0])[row] = w0;
      } else if constexpr (KIND == kNChain) {
        if (reset) {  // nchain.h:61-66
          w0 = 0;
          done = false;
        } else {  // nchain.h:68-84
          done = cur >= a.max_episode_steps;
          if (g.UniformReal(0, 1) < 0.2) act = 1 - act;
          if (act != 0) {
            reward = 2.0f;
            w0 = 0;
          } else if (w0 < 4) {
            ++w0;
          } else {
            reward = 10.0f;
          }
        }
        ((int*)out.p[kKeyEnv0])[row] = w0;
      } else if constexpr (KIND == kCliffWalking) {
        int x = w0 / 12, y = w0 % 12;
        float prob = 1.0f;
        if (reset) {  // cliffwalking.h:63-68
          x = 3;
          y = 0;
          done = false;
        } else {  // cliffwalking.h:70-94
          if (cfg.is_slippery) {  // SampleAction :97-104
            act = (act + (g.UniformInt(0, 2) - 1) + 4) % 4;
          }
          reward = -1.0f;
          if (act == 0) {
            --x;
          } else if (act == 1) {
            ++y;
          } else if (act == 2) {
            ++x;
          } else {
            --y;
          }
          x = min(3, max(0, x));
          y = min(11, max(0, y));
          if (x == 3 && y > 0 && y < 11) {
            reward = -100.0f;
            x = 3;
            y = 0;
          }
          if (x == 3 && y == 11) done = true;
          prob = cfg.is_slippery ? 1.0f / 3.0f : 1.0f;
        }
        w0 = x * 12 + y;
        ((int*)out.p[kKeyEnv0])[row] = w0;
        ((float*)out.p[kKeyEnv0 + 1])[row] = prob;
      } else {  // Blackjack
        Hand player = Hand::Unpack(w0), dealer = Hand::Unpack(w1);
        if (reset) {  // blackjack.h:65-74
          player = Hand{0, 0, 0, 0, 0};
          dealer = Hand{0, 0, 0, 0, 0};
          player.Push(DrawCard(g));
          player.Push(DrawCard(g));
          dealer.Push(DrawCard(g));
          dealer.Push(DrawCard(g));
          done = false;
        } else if (act != 0) {  // hit, blackjack.h:79-84
          player.Push(DrawCard(g));
          if (player.SumHand() > 21) {
            done = true;
            reward = -1.0f;
          }
        } else {  // stick, blackjack.h:85-99
          done = true;
          while (dealer.SumHand() < 17) dealer.Push(DrawCard(g));
          int ps = player.Score(), ds = dealer.Score();
          reward = (ps > ds ? 1.0f : 0.0f) - (ps < ds ? 1.0f : 0.0f);
          if (cfg.sab && player.IsNatural() && !dealer.IsNatural()) {
            reward = 1.0f;
          } else if (!cfg.sab && cfg.natural && player.IsNatural() &&
                     reward == 1.0f) {
            reward = 1.5f;
          }
        }
        w0 = player.Pack();
        w1 = dealer.Pack();
        int* o = (int*)out.p[kKeyEnv0] + (size_t)row * 3;  // :104-110
        o[0] = player.SumHand();
        o[1] = dealer.c0;
        o[2] = player.ace;
        dev.w1[e] = w1;
      }
      g.Commit();
      dev.w0[e] = w0;
      cm.done[e] = done ? 1 : 0;
      cm.cur_step[e] = cur;
      WriteCommon(out, row, e + a.id_offset, cur, done, reward,
                  a.max_episode_steps);
    }
    if constexpr (KIND == kCatch) {
      // obs [rows, H, W] one-hot floats written cooperatively by the block so
      // consecutive lanes hit consecutive addresses (catch.h:88-93 writes two
      // ones into a zero-initialised buffer).
      sh_p1[threadIdx.x] = p1;
      sh_p2[threadIdx.x] = p2;
      __syncthreads();
      // Round 6: 16-byte stores, (row, cell) of a thread's position carried along instead of divided out per element
      // -- the loop was 50 trips of a division, two LDS reads and a 4-byte store per thread: 12.0 us per launch at
      // num_envs = 65536, 1.4 TB/s (profiles/r6o_toy_kernel_trace.txt).  `base` is a multiple of the block size, so
      // the block's slab starts 16-byte aligned whatever H x W is.
      const int hw = cfg.height * cfg.width;
      const int rows_here = min((int)blockDim.x, a.k - base);
      float* obs = (float*)out.p[kKeyEnv0] + (size_t)base * hw;
      const int total = rows_here * hw, n4 = total >> 2;
      const int step = 4 * (int)blockDim.x, sr = step / hw, sc = step - sr * hw;  // (uniform)
      int r = (4 * (int)threadIdx.x) / hw, c = 4 * (int)threadIdx.x - r * hw;
      for (int i4 = threadIdx.x; i4 < n4; i4 += blockDim.x) {
        // (i4 < n4: all four cells lie inside the slab, so rr <= rows_here - 1)
        float v[4];
        int rr = r, cc = c, q1 = sh_p1[rr], q2 = sh_p2[rr];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = (cc == q1 || cc == q2) ? 1.0f : 0.0f;
          if (++cc == hw && j < 3) {
            cc = 0;
            ++rr;
            q1 = sh_p1[rr];
            q2 = sh_p2[rr];
          }
        }
        reinterpret_cast<float4*>(obs)[i4] = make_float4(v[0], v[1], v[2], v[3]);
        r += sr;
        c += sc;
        if (c >= hw) {
          c -= hw;
          ++r;
        }
      }
      for (int idx = 4 * n4 + threadIdx.x; idx < total; idx += blockDim.x) {  // (fewer than four cells)
        const int rt = idx / hw, ct = idx - rt * hw;
        obs[idx] = (ct == sh_p1[rt] || ct == sh_p2[rt]) ? 1.0f : 0.0f;
      }
      __syncthreads();
    }
  }
}

__global__ void ToyGetState(ToyDev dev, CommonDev cm, const int* ids, int k,
                            double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i];
  double* o = out + (size_t)i * 4;
  o[0] = dev.w0[e];
  o[1] = dev.w1 ? dev.w1[e] : 0;
  o[2] = cm.done[e];
  o[3] = cm.cur_step[e];
}
__global__ void ToySetState(ToyDev dev, CommonDev cm, const int* ids, int k,
                            const double* in) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i];
  const double* o = in + (size_t)i * 4;
  dev.w0[e] = (int)o[0];
  if (dev.w1) dev.w1[e] = (int)o[1];
  cm.done[e] = o[2] != 0.0;
  cm.cur_step[e] = (int)o[3];
}

struct FamilyInfo {
  const char* name;
  int kind;
};
const FamilyInfo kFamilies[] = {
    {"Catch", kCatch},   {"FrozenLake", kFrozenLake}, {"Taxi", kTaxi},
    {"NChain", kNChain}, {"CliffWalking", kCliffWalking},
    {"Blackjack", kBlackjack}};

const FamilyInfo* Find(const std::string& name) {
  for (auto& f : kFamilies) {
    if (name == f.name) return &f;
  }
  return nullptr;
}

ToyCfg MakeCfg(const Config& cfg) {
  ToyCfg c{};
  c.size = (int)cfg.Get("size", 4);
  c.height = (int)cfg.Get("height", 10);
  c.width = (int)cfg.Get("width", 5);
  c.is_slippery = cfg.Get("is_slippery", 0) != 0;
  c.natural = cfg.Get("natural", 0) != 0;
  c.sab = cfg.Get("sab", 1) != 0;
  return c;
}

std::vector<KeySpec> EnvKeys(int kind, const ToyCfg& c) {
  switch (kind) {
    case kCatch: return {{"obs", EPA_F32, {c.height, c.width}}};
    case kCliffWalking:
      return {{"obs", EPA_I32, {}}, {"info:prob", EPA_F32, {}}};
    case kBlackjack: return {{"obs", EPA_I32, {3}}};
    default: return {{"obs", EPA_I32, {}}};
  }
}

template <int KIND>
class ToyPool : public Pool {
 public:
  bool ConcurrentSafe() const override { return true; }  // per-env state + the launch's own block only
  explicit ToyPool(const Config& cfg)
      : Pool(cfg, EnvKeys(KIND, MakeCfg(cfg)), KeySpec{"action", EPA_I32, {}},
             /*needs_rng=*/true),
        tcfg_(MakeCfg(cfg)) {
    if (KIND == kCatch && (tcfg_.height < 2 || tcfg_.height > 255 ||
                           tcfg_.width < 1 || tcfg_.width > 255)) {
      throw std::invalid_argument("Catch: height/width out of range");
    }
    if (KIND == kFrozenLake && tcfg_.size != 4 && tcfg_.size != 8) {
      // the reference silently uses the 4x4 map with size_ clamps
      // (frozen_lake.h:64-69); only the two registered sizes are supported.
      throw std::invalid_argument("FrozenLake: size must be 4 or 8");
    }
    EPA_HIP(hipMalloc(&dev_.w0, sizeof(int) * cfg.num_envs));
    EPA_HIP(hipMemsetAsync(dev_.w0, 0, sizeof(int) * cfg.num_envs, stream_));
    if (KIND == kBlackjack) {
      EPA_HIP(hipMalloc(&dev_.w1, sizeof(int) * cfg.num_envs));
      EPA_HIP(hipMemsetAsync(dev_.w1, 0, sizeof(int) * cfg.num_envs, stream_));
    }
    // envs that draw at their own times (a reset row draws, or skips the step's draw): tiled generator words
    if (KIND == kFrozenLake || KIND == kTaxi || KIND == kBlackjack) mt_tile_default_ = 16;
    InitCommon();
  }
  ~ToyPool() override {
    if (dev_.w0) (void)hipFree(dev_.w0);
    if (dev_.w1) (void)hipFree(dev_.w1);
  }
  int StateDim() const override { return 4; }
  void GetState(const int* d_ids, int k, double* d_out) override {
    hipLaunchKernelGGL(ToyGetState, dim3((k + 255) / 256), dim3(256), 0,
                       stream_, dev_, common_, d_ids, k, d_out);
  }
  void SetState(const int* d_ids, int k, const double* d_in) override {
    hipLaunchKernelGGL(ToySetState, dim3((k + 255) / 256), dim3(256), 0,
                       stream_, dev_, common_, d_ids, k, d_in);
  }

 protected:
  void Launch(const int* d_ids, int k, const void* d_action, bool force_reset,
              const OutPtrs& out) override {
    StepArgs a{d_ids, k, force_reset ? 1 : 0, cfg_.max_episode_steps,
               cfg_.env_id_offset};
    int blocks = std::min((k + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(ToyStepKernel<KIND>, dim3(blocks), dim3(256), 0, stream_,
                       dev_, common_, a, static_cast<const int*>(d_action), out,
                       tcfg_);
  }

 private:
  ToyDev dev_{};
  ToyCfg tcfg_;
};

}  // namespace

bool DescribeToyText(const std::string& family, const Config& cfg,
                     std::vector<KeySpec>* state, KeySpec* action) {
  const FamilyInfo* fi = Find(family);
  if (!fi) return false;
  *state = EnvKeys(fi->kind, MakeCfg(cfg));
  *action = KeySpec{"action", EPA_I32, {}};
  return true;
}

Pool* MakeToyText(const std::string& family, const Config& cfg) {
  const FamilyInfo* fi = Find(family);
  if (!fi) return nullptr;
  switch (fi->kind) {
    case kCatch: return new ToyPool<kCatch>(cfg);
    case kFrozenLake: return new ToyPool<kFrozenLake>(cfg);
    case kTaxi: return new ToyPool<kTaxi>(cfg);
    case kNChain: return new ToyPool<kNChain>(cfg);
    case kCliffWalking: return new ToyPool<kCliffWalking>(cfg);
    default: return new ToyPool<kBlackjack>(cfg);
  }
}

}  // namespace epa
