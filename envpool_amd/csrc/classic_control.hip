// K1 — classic_control batched step kernels (one env per thread).
//
// Replaces, for the whole batch in one launch, the per-env bodies
//   CartPoleEnv::{Reset,Step,WriteState}   envpool/classic_control/cartpole.h:82-131
//   PendulumEnv::{...}                     envpool/classic_control/pendulum.h:77-135
//   MountainCarEnv::{...}                  envpool/classic_control/mountain_car.h:76-130
//   MountainCarContinuousEnv::{...}        envpool/classic_control/mountain_car_continuous.h:77-138
//   AcrobotEnv::{Reset,Step,Rk4,Derivs}    envpool/classic_control/acrobot.h:94-197
// and the runtime around them (async_envpool.h:118-132 worker loop,
// env.h:184-256 EnvStep/Allocate).
//
// Data layout (HBM): state is SoA in float64 exactly like the reference's
// `double x_, x_dot_, ...` members: s[j][N]; obs are written as float32 rows
// ([k,4] CartPole = one 16-byte store per thread).  Internal arithmetic is
// fp64 in the reference's operation order; this file is compiled with
// -ffp-contract=off so a*b+c is not fused (the reference's x86-64 build does
// not fuse), leaving device libm sin/cos as the only source of (<=1 ulp)
// differences.
//
// HBM-bound streaming kernel: algorithmic bytes per env-step:
// CartPole 128, Pendulum 92, Acrobot 144.
#include "device_common.hip.h"
#include "engine.h"

namespace epa {
namespace {

constexpr double kPi = 3.14159265358979323846;

enum Kind : int {
  kCartPole = 0,
  kPendulum,
  kMountainCar,
  kMountainCarContinuous,
  kAcrobot
};

struct ClassicDev {
  double* s[5];
};

template <int KIND>
struct Traits;
template <>
struct Traits<kCartPole> {
  static constexpr int kNumState = 4;
  using Act = int;
};
template <>
struct Traits<kPendulum> {
  static constexpr int kNumState = 2;
  using Act = float;
};
template <>
struct Traits<kMountainCar> {
  static constexpr int kNumState = 2;
  using Act = int;
};
template <>
struct Traits<kMountainCarContinuous> {
  static constexpr int kNumState = 2;
  using Act = float;
};
template <>
struct Traits<kAcrobot> {
  static constexpr int kNumState = 5;
  using Act = int;
};

// ---- reset bodies ---------------------------------------------------------
template <int KIND>
__device__ inline void ResetBody(double* s, Mt19937& g) {
  // (the draws of a reset as ONE burst of generator words: Mt19937::NextWords)
  if constexpr (KIND == kCartPole) {  // cartpole.h:82-90
    g.UniformReals<4>(-0.05, 0.05, s);
  } else if constexpr (KIND == kPendulum) {  // pendulum.h:77-85
#pragma clang fp contract(off)
    uint32_t w[4];
    g.NextWords<4>(w);
    s[0] = (Mt19937::CanonicalOf(w[0], w[1]) * (kPi - (-kPi))) + (-kPi);
    s[1] = (Mt19937::CanonicalOf(w[2], w[3]) * (1.0 - (-1.0))) + (-1.0);
  } else if constexpr (KIND == kMountainCar ||
                       KIND == kMountainCarContinuous) {  // mountain_car.h:76-82
    s[0] = g.UniformReal(-0.6, -0.4);
    s[1] = 0.0;
  } else {  // acrobot.h:94-103
    g.UniformReals<4>(-0.1, 0.1, s);
    s[4] = 0;
  }
}

// ---- Acrobot helpers (acrobot.h:148-178) ----------------------------------
struct V5 {
  double s0, s1, s2, s3, s4;
};
__device__ inline V5 Add(V5 a, V5 b) {
  return {a.s0 + b.s0, a.s1 + b.s1, a.s2 + b.s2, a.s3 + b.s3, a.s4 + b.s4};
}
__device__ inline V5 Mul(V5 a, double v) {
  return {a.s0 * v, a.s1 * v, a.s2 * v, a.s3 * v, a.s4 * v};
}
__device__ inline V5 Derivs(V5 s) {
  const double kG = 9.8, kL = 1.0, kM = 1.0, kLC = 0.5, kI = 1.0;
  double theta1 = s.s0, theta2 = s.s1, dtheta1 = s.s2, dtheta2 = s.s3;
  double a = s.s4;
  double d1 = kM * kLC * kLC +
              kM * (kL * kL + kLC * kLC + 2 * kL * kLC * cos(theta2)) + kI * 2;
  double d2 = kM * (kLC * kLC + kL * kLC * cos(theta2)) + kI;
  double phi2 = kM * kLC * kG * cos(theta1 + theta2 - kPi / 2);
  double phi1 =
      -(dtheta2 + 2 * dtheta1) * kM * kL * kLC * dtheta2 * sin(theta2) +
      kM * (kLC + kL) * kG * cos(theta1 - kPi / 2) + phi2;
  double ddtheta2 = (a + d2 / d1 * phi1 -
                     kM * kL * kLC * dtheta1 * dtheta1 * sin(theta2) - phi2) /
                    (kM * kLC * kLC + kI - d2 * d2 / d1);
  double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
  return {dtheta1, dtheta2, ddtheta1, ddtheta2, 0};
}
__device__ inline V5 Rk4(V5 y0) {
  const double kDt = 0.2;
  V5 k1 = Derivs(y0);
  V5 k2 = Derivs(Add(y0, Mul(k1, kDt / 2)));
  V5 k3 = Derivs(Add(y0, Mul(k2, kDt / 2)));
  V5 k4 = Derivs(Add(y0, Mul(k3, kDt)));
  return Add(y0,
             Mul(Add(Add(Add(k1, Mul(k2, 2)), Mul(k3, 2)), k4), kDt / 6.0));
}

// ---- step bodies: return reward, update s and done ------------------------
template <int KIND>
__device__ inline float StepBody(double* s, typename Traits<KIND>::Act act,
                                 bool* done, int version) {
  if constexpr (KIND == kCartPole) {  // cartpole.h:92-116
    const double kGravity = 9.8, kMassCart = 1.0, kMassPole = 0.1;
    const double kMassTotal = kMassCart + kMassPole, kLength = 0.5;
    const double kMassPoleLength = kMassPole * kLength, kForceMag = 10.0;
    const double kTau = 0.02;
    const double kThetaThresholdRadians = 12 * 2 * kPi / 360;
    const double kXThreshold = 2.4;
    double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
    double force = act == 1 ? kForceMag : -kForceMag;
    double costheta = cos(theta);
    double sintheta = sin(theta);
    double temp = (force + kMassPoleLength * theta_dot * theta_dot * sintheta) /
                  kMassTotal;
    double theta_acc =
        (kGravity * sintheta - costheta * temp) /
        (kLength * (4.0 / 3.0 - kMassPole * costheta * costheta / kMassTotal));
    double x_acc = temp - kMassPoleLength * theta_acc * costheta / kMassTotal;
    x += kTau * x_dot;
    x_dot += kTau * x_acc;
    theta += kTau * theta_dot;
    theta_dot += kTau * theta_acc;
    if (x < -kXThreshold || x > kXThreshold ||
        theta < -kThetaThresholdRadians || theta > kThetaThresholdRadians) {
      *done = true;
    }
    s[0] = x;
    s[1] = x_dot;
    s[2] = theta;
    s[3] = theta_dot;
    return 1.0f;
  } else if constexpr (KIND == kPendulum) {  // pendulum.h:87-122
    const double kMaxSpeed = 8, kMaxTorque = 2, kDt = 0.05, kGravity = 10;
    double theta = s[0], theta_dot = s[1];
    double u = act;
    if (act < -kMaxTorque) {
      u = -kMaxTorque;
    } else if (act > kMaxTorque) {
      u = kMaxTorque;
    }
    double cost = theta * theta + 0.1 * theta_dot * theta_dot + 0.001 * u * u;
    double new_theta_dot =
        theta_dot + 3 * (kGravity / 2 * sin(theta) + u) * kDt;
    if (version == 0) theta += new_theta_dot * kDt;
    theta_dot = new_theta_dot;
    if (new_theta_dot < -kMaxSpeed) {
      theta_dot = -kMaxSpeed;
    } else if (new_theta_dot > kMaxSpeed) {
      theta_dot = kMaxSpeed;
    }
    if (version == 1) theta += new_theta_dot * kDt;
    while (theta < -kPi) theta += kPi * 2;
    while (theta >= kPi) theta -= kPi * 2;
    s[0] = theta;
    s[1] = theta_dot;
    return static_cast<float>(-cost);
  } else if constexpr (KIND == kMountainCar) {  // mountain_car.h:84-107
    const double kMinPos = -1.2, kMaxPos = 0.6, kMaxSpeed = 0.07;
    const double kForce = 0.001, kGoalPos = 0.5, kGoalVel = 0;
    const double kGravity = 0.0025;
    double pos = s[0], vel = s[1];
    double a = act - 1;
    vel += a * kForce - cos(3 * pos) * kGravity;
    if (vel < -kMaxSpeed) {
      vel = -kMaxSpeed;
    } else if (vel > kMaxSpeed) {
      vel = kMaxSpeed;
    }
    pos += vel;
    if (pos < kMinPos) {
      pos = kMinPos;
    } else if (pos > kMaxPos) {
      pos = kMaxPos;
    }
    if (pos == kMinPos && vel < 0) vel = 0;
    if (pos >= kGoalPos && vel >= kGoalVel) *done = true;
    s[0] = pos;
    s[1] = vel;
    return -1.0f;
  } else if constexpr (KIND == kMountainCarContinuous) {
    // mountain_car_continuous.h:85-115
    const double kMinPos = -1.2, kMaxPos = 0.6, kMaxSpeed = 0.07;
    const double kPower = 0.0015, kGoalPos = 0.45, kGoalVel = 0;
    const double kGravity = 0.0025;
    double pos = s[0], vel = s[1];
    double a = act;
    double reward = -0.1 * a * a;
    if (a < -1) {
      a = -1;
    } else if (a > 1) {
      a = 1;
    }
    vel += a * kPower - cos(3 * pos) * kGravity;
    if (vel < -kMaxSpeed) {
      vel = -kMaxSpeed;
    } else if (vel > kMaxSpeed) {
      vel = kMaxSpeed;
    }
    pos += vel;
    if (pos < kMinPos) {
      pos = kMinPos;
    } else if (pos > kMaxPos) {
      pos = kMaxPos;
    }
    if (pos == kMinPos && vel < 0) vel = 0;
    if (pos >= kGoalPos && vel >= kGoalVel) {
      *done = true;
      reward += 100;
    }
    s[0] = pos;
    s[1] = vel;
    return static_cast<float>(reward);
  } else {  // Acrobot, acrobot.h:105-141
    const double kMaxVel1 = 4 * kPi, kMaxVel2 = 9 * kPi;
    float reward = -1.0f;
    V5 v = {s[0], s[1], s[2], s[3], s[4]};
    v.s4 = act - 1;
    v = Rk4(v);
    while (v.s0 < -kPi) v.s0 += kPi * 2;
    while (v.s1 < -kPi) v.s1 += kPi * 2;
    while (v.s0 >= kPi) v.s0 -= kPi * 2;
    while (v.s1 >= kPi) v.s1 -= kPi * 2;
    if (v.s2 < -kMaxVel1) v.s2 = -kMaxVel1;
    if (v.s3 < -kMaxVel2) v.s3 = -kMaxVel2;
    if (v.s2 > kMaxVel1) v.s2 = kMaxVel1;
    if (v.s3 > kMaxVel2) v.s3 = kMaxVel2;
    if (-cos(v.s0) - cos(v.s0 + v.s1) > 1) {
      *done = true;
      reward = 0.0f;
    }
    s[0] = v.s0;
    s[1] = v.s1;
    s[2] = v.s2;
    s[3] = v.s3;
    s[4] = v.s4;
    return reward;
  }
}

template <int KIND>
__device__ inline void WriteObs(const OutPtrs& out, int row, const double* s) {
  if constexpr (KIND == kCartPole) {  // cartpole.h:123-131
    float4 o = make_float4((float)s[0], (float)s[1], (float)s[2], (float)s[3]);
    ((float4*)out.p[kKeyEnv0])[row] = o;
  } else if constexpr (KIND == kPendulum) {  // pendulum.h:128-135
    float* o = (float*)out.p[kKeyEnv0] + (size_t)row * 3;
    o[0] = (float)cos(s[0]);
    o[1] = (float)sin(s[0]);
    o[2] = (float)s[1];
  } else if constexpr (KIND == kAcrobot) {  // acrobot.h:180-193
    float2* o = (float2*)out.p[kKeyEnv0] + (size_t)row * 3;
    o[0] = make_float2((float)cos(s[0]), (float)sin(s[0]));
    o[1] = make_float2((float)cos(s[1]), (float)sin(s[1]));
    o[2] = make_float2((float)s[2], (float)s[3]);
    ((float2*)out.p[kKeyEnv0 + 1])[row] = make_float2((float)s[0], (float)s[1]);
  } else {  // mountain_car.h:123-129
    ((float2*)out.p[kKeyEnv0])[row] = make_float2((float)s[0], (float)s[1]);
  }
}

// kEarly: every input of the row -- state, action, generator position -- is read together with `done`, in front of the
// reset branch.  At BASELINE config 2's size (num_envs = 65536: one wave per SIMD) a step is a chain of dependent
// round trips to memory, `done` -> state -> compute -> stores, and for a wave with a reset row (4.5 % of CartPole's
// rows reset per step: every wave has one) `done` -> generator position -> words -> stores; read early, both are one
// round trip shorter.  The price is one generator position (4 bytes) per row and step that only reset rows need.
// Measured (rocprofv3 kernel trace, profiles/r6o_classic_early_kernel_trace.txt, num_envs = 65536): CartPole 5.98 ->
// 5.60 us (4 M rows: 156.6 -> 150.3), Pendulum / MountainCar unchanged (3.7 - 4.0 us: their waves rarely hold a reset
// row), Acrobot 9.5 -> 9.9: on for CartPole only ("classic_early" = 0 / 1 overrides).
template <int KIND, bool kEarly>
__global__ __launch_bounds__(256) void ClassicStepKernel(
    ClassicDev dev, CommonDev cm, StepArgs a,
    const typename Traits<KIND>::Act* __restrict__ action, OutPtrs out,
    int version) {
  constexpr int NS = Traits<KIND>::kNumState;
  using Act = typename Traits<KIND>::Act;
  static_assert(sizeof(Act) == 4, "the stand-in address below");
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < a.k;
       row += gridDim.x * blockDim.x) {
    int e = a.ids ? a.ids[row] - a.id_offset : row;
    const unsigned char done_in = cm.done[e];
    int cur = cm.cur_step[e];
    double s[NS];
    Act act{};
    int position = 0;
    if constexpr (kEarly) {
#pragma unroll
      for (int j = 0; j < NS; ++j) s[j] = dev.s[j][e];
      // (a launch that resets every row has no actions: any readable word stands in, the value is never used)
      const Act* src = action != nullptr ? action + row : reinterpret_cast<const Act*>(cm.cur_step + e);
      act = *src;
      position = cm.mti[e];
#pragma unroll
      for (int j = 0; j < NS; ++j) asm volatile("" : "+v"(s[j]));  // (issued here, not sunk into the branches)
      asm volatile("" : "+v"(act), "+v"(position));
    }
    bool done = done_in != 0;
    // async_envpool.h:127: reset = force_reset || env->IsDone()
    bool reset = a.force_reset || done;
    float reward = 0.0f;
    if (reset) {
      cur = 0;  // env.h:211-212
      Mt19937 g(cm, e);  // (kEarly: its read of the position is the one above, or hits the line that one brought in)
      if constexpr (kEarly) g.idx = g.idx0 = position;
      ResetBody<KIND>(s, g);
      g.Commit();
      done = false;
    } else {
      ++cur;  // env.h:214
      if constexpr (!kEarly) {
#pragma unroll
        for (int j = 0; j < NS; ++j) s[j] = dev.s[j][e];
        act = action[row];
      }
      // `done_ = (++elapsed_step_ >= max_episode_steps_)`: elapsed_step_ and
      // current_step_ coincide once an env has been reset.
      done = cur >= a.max_episode_steps;
      reward = StepBody<KIND>(s, act, &done, version);
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) dev.s[j][e] = s[j];
    cm.done[e] = done ? 1 : 0;
    cm.cur_step[e] = cur;
    WriteObs<KIND>(out, row, s);
    WriteCommon(out, row, e + a.id_offset, cur, done, reward,
                a.max_episode_steps);
  }
}

// flat state vector for tests: [s..., done, cur_step]
template <int NS>
__global__ void GetStateKernel(ClassicDev dev, CommonDev cm, const int* ids,
                               int k, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i];
  double* o = out + (size_t)i * (NS + 2);
  for (int j = 0; j < NS; ++j) o[j] = dev.s[j][e];
  o[NS] = cm.done[e];
  o[NS + 1] = cm.cur_step[e];
}
template <int NS>
__global__ void SetStateKernel(ClassicDev dev, CommonDev cm, const int* ids,
                               int k, const double* in) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  int e = ids[i];
  const double* o = in + (size_t)i * (NS + 2);
  for (int j = 0; j < NS; ++j) dev.s[j][e] = o[j];
  cm.done[e] = o[NS] != 0.0;
  cm.cur_step[e] = (int)o[NS + 1];
}

struct FamilyInfo {
  const char* name;
  int kind;
  std::vector<KeySpec> keys;
  KeySpec action;
};

const std::vector<FamilyInfo>& Families() {
  static const std::vector<FamilyInfo> f = {
      {"CartPole", kCartPole, {{"obs", EPA_F32, {4}}}, {"action", EPA_I32, {}}},
      {"Pendulum", kPendulum, {{"obs", EPA_F32, {3}}}, {"action", EPA_F32, {1}}},
      {"MountainCar", kMountainCar, {{"obs", EPA_F32, {2}}},
       {"action", EPA_I32, {}}},
      {"MountainCarContinuous", kMountainCarContinuous, {{"obs", EPA_F32, {2}}},
       {"action", EPA_F32, {1}}},
      {"Acrobot", kAcrobot,
       {{"obs", EPA_F32, {6}}, {"info:state", EPA_F32, {2}}},
       {"action", EPA_I32, {}}},
  };
  return f;
}

const FamilyInfo* Find(const std::string& name) {
  for (auto& f : Families()) {
    if (name == f.name) return &f;
  }
  return nullptr;
}

template <int KIND>
class ClassicPool : public Pool {
 public:
  bool ConcurrentSafe() const override { return true; }  // per-env state + the launch's own block only
  ClassicPool(const Config& cfg, const FamilyInfo& fi)
      : Pool(cfg, fi.keys, fi.action, /*needs_rng=*/true),
        version_((int)cfg.Get("version", 0)) {
    for (int j = 0; j < NS; ++j) {
      EPA_HIP(hipMalloc(&dev_.s[j], sizeof(double) * cfg.num_envs));
      EPA_HIP(hipMemsetAsync(dev_.s[j], 0, sizeof(double) * cfg.num_envs,
                             stream_));
    }
    // CartPole / Acrobot episodes end at their own times under any policy: tiled generator words
    // (engine.h: mt_tile_default_); Pendulum and the MountainCars run to the step limit together
    if (KIND == kCartPole || KIND == kAcrobot) mt_tile_default_ = 16;
    block_ = (int)cfg.Get("classic_block", 0);
    rows_ = (int)cfg.Get("classic_rows", 1);
    early_ = (int)cfg.Get("classic_early", -1);
    if ((block_ != 0 && block_ != 64 && block_ != 128 && block_ != 256) || rows_ < 1 || rows_ > 8) {
      throw std::invalid_argument("classic_block must be 64, 128 or 256 and classic_rows 1 .. 8");
    }
    InitCommon();
  }
  ~ClassicPool() override {
    for (int j = 0; j < NS; ++j) {
      if (dev_.s[j]) (void)hipFree(dev_.s[j]);
    }
  }
  int StateDim() const override { return NS + 2; }
  void GetState(const int* d_ids, int k, double* d_out) override {
    hipLaunchKernelGGL(GetStateKernel<NS>, dim3((k + 255) / 256), dim3(256), 0,
                       stream_, dev_, common_, d_ids, k, d_out);
  }
  void SetState(const int* d_ids, int k, const double* d_in) override {
    hipLaunchKernelGGL(SetStateKernel<NS>, dim3((k + 255) / 256), dim3(256), 0,
                       stream_, dev_, common_, d_ids, k, d_in);
  }

 protected:
  void Launch(const int* d_ids, int k, const void* d_action, bool force_reset,
              const OutPtrs& out) override {
    StepArgs a{d_ids, k, force_reset ? 1 : 0, cfg_.max_episode_steps,
               cfg_.env_id_offset};
    // "classic_block" threads per block (64 / 128 / 256; 0 = default), "classic_rows" rows per thread of the
    // grid-stride loop: A/B keys.  At num_envs = 65536 (BASELINE config 2) a step is one wave per SIMD and one
    // dependent load -> store chain: an EMPTY kernel takes 1.54 us per back-to-back launch on this machine and a
    // kernel with nothing but this step's loads and stores 2.85 us (tools/probes/launch_floor_probe.hip); the step
    // takes 4.0 - 5.7 us.  64-thread blocks are ~10 % faster there for the short bodies (MountainCar 4.71 -> 4.11,
    // Pendulum 4.44 -> 4.04 us), not for CartPole / Acrobot; two or more rows per thread are slower for every family
    // (profiles/r6j_classic_launch_shape_ab.txt).
    const bool short_body = KIND != kCartPole && KIND != kAcrobot;
    const int block = block_ > 0 ? block_ : (short_body && k <= 131072 ? 64 : 256);
    const int per = block * rows_;
    int blocks = std::min((k + per - 1) / per, 256 * 8 * (256 / block));
    const bool early = early_ >= 0 ? early_ != 0 : KIND == kCartPole;
    if (early) {
      hipLaunchKernelGGL((ClassicStepKernel<KIND, true>), dim3(blocks), dim3(block), 0, stream_, dev_, common_, a,
                         static_cast<const typename Traits<KIND>::Act*>(d_action), out, version_);
    } else {
      hipLaunchKernelGGL((ClassicStepKernel<KIND, false>), dim3(blocks), dim3(block), 0, stream_, dev_, common_, a,
                         static_cast<const typename Traits<KIND>::Act*>(d_action), out, version_);
    }
  }

 private:
  static constexpr int NS = Traits<KIND>::kNumState;
  ClassicDev dev_{};
  int version_;
  int block_{0}, rows_{1}, early_{-1};
};

}  // namespace

bool DescribeClassicControl(const std::string& family, const Config& cfg,
                            std::vector<KeySpec>* state, KeySpec* action) {
  (void)cfg;
  const FamilyInfo* fi = Find(family);
  if (!fi) return false;
  *state = fi->keys;
  *action = fi->action;
  return true;
}

Pool* MakeClassicControl(const std::string& family, const Config& cfg) {
  const FamilyInfo* fi = Find(family);
  if (!fi) return nullptr;
  switch (fi->kind) {
    case kCartPole: return new ClassicPool<kCartPole>(cfg, *fi);
    case kPendulum: return new ClassicPool<kPendulum>(cfg, *fi);
    case kMountainCar: return new ClassicPool<kMountainCar>(cfg, *fi);
    case kMountainCarContinuous:
      return new ClassicPool<kMountainCarContinuous>(cfg, *fi);
    default: return new ClassicPool<kAcrobot>(cfg, *fi);
  }
}

}  // namespace epa
