// Host side of the engine: Pool (stream, staging, batch FIFO) and the C ABI.
// See engine.h / include/envpool_amd.h for what each piece replaces in the
// reference.
#include "engine.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "device_common.hip.h"

namespace epa {

namespace {
thread_local std::string g_last_error;

size_t Align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
// host-path transfers below this size stay on the kernel stream (see Pool::Send)
constexpr size_t kSplitStreamBytes = 256 * 1024;

__global__ void InitCommonKernel(CommonDev c, int seed, const int* env_seed,
                                 int id_offset, int with_rng) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= c.n) return;
  c.cur_step[e] = -1;  // Env::current_step_{-1}, env.h:86
  c.done[e] = 1;       // XxxEnv::done_{true}
  if (with_rng) {
    // std::mt19937(seed_): seed_ = env_seed[env_id] or seed + env_id
    // (envpool/core/env.h:101-117)
    uint32_t s = (uint32_t)(env_seed ? env_seed[e] : seed + id_offset + e);
    uint32_t* col = c.mt + e;
    uint32_t x = s;
    col[0] = x;
    for (int i = 1; i < 624; ++i) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
      col[(size_t)i * c.n] = x;
    }
    c.mti[e] = 624;
  }
}
}  // namespace

void SetLastError(const std::string& msg) { g_last_error = msg; }

void LaunchInitCommon(CommonDev c, int seed, const int* d_env_seed,
                      int id_offset, bool with_rng, hipStream_t s) {
  int threads = 256, blocks = (c.n + threads - 1) / threads;
  hipLaunchKernelGGL(InitCommonKernel, dim3(blocks), dim3(threads), 0, s, c,
                     seed, d_env_seed, id_offset, with_rng ? 1 : 0);
  EPA_HIP(hipGetLastError());
}

Config Config::From(const epa_config* c) {
  Config r;
  if (c == nullptr) throw std::invalid_argument("null config");
  r.num_envs = c->num_envs;
  if (r.num_envs < 1) throw std::invalid_argument("num_envs must be >= 1");
  r.batch_size = c->batch_size;
  r.seed = c->seed;
  if (c->env_seed != nullptr) {
    r.env_seed.assign(c->env_seed, c->env_seed + c->num_envs);
  }
  r.max_episode_steps = c->max_episode_steps > 0 ? c->max_episode_steps : INT_MAX;
  r.device = c->device;
  r.env_id_offset = c->env_id_offset;
  for (int i = 0; i < c->n_params; ++i) {
    r.params[c->param_keys[i]] = c->param_values[i];
  }
  // EnvSpec ctor, envpool/core/env_spec.h:75-83
  if (r.batch_size > r.num_envs) {
    throw std::invalid_argument(
        "It is required that batch_size <= num_envs, got num_envs = " +
        std::to_string(r.num_envs) +
        ", batch_size = " + std::to_string(r.batch_size));
  }
  if (r.batch_size <= 0) r.batch_size = r.num_envs;
  return r;
}

std::vector<KeySpec> CommonStateKeys() {
  return {{"info:env_id", EPA_I32, {}},  {"info:players.env_id", EPA_I32, {}},
          {"elapsed_step", EPA_I32, {}}, {"done", EPA_BOOL, {}},
          {"reward", EPA_F32, {}},       {"discount", EPA_F32, {}},
          {"step_type", EPA_I32, {}},    {"trunc", EPA_BOOL, {}}};
}

Pool::Pool(const Config& cfg, std::vector<KeySpec> env_state_keys,
           KeySpec action, bool needs_rng)
    : cfg_(cfg), action_(std::move(action)), needs_rng_(needs_rng) {
  keys_ = CommonStateKeys();
  for (auto& k : env_state_keys) keys_.push_back(std::move(k));
  if (keys_.size() > kMaxKeys) throw std::runtime_error("too many state keys");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    throw DeviceError(
        "no HIP device available: envpool_amd has no CPU fallback "
        "(hipGetDeviceCount: " +
        std::string(hipGetErrorString(e)) + ")");
  }
  if (cfg_.device < 0 || cfg_.device >= ndev) {
    throw std::invalid_argument("device ordinal out of range");
  }
  EPA_HIP(hipSetDevice(cfg_.device));
  EPA_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  compute_.push_back(stream_);
  // "compute_streams" (extension key, default 4): async mode rotates successive batches over this
  // many streams; 1 = every kernel on one stream (the behaviour up to round 2)
  const bool async_mode = cfg_.batch_size > 0 && cfg_.batch_size < cfg_.num_envs;
  const int want = async_mode ? std::max(1, std::min(16, (int)cfg_.Get("compute_streams", 4))) : 1;
  for (int i = 1; i < want; ++i) {
    hipStream_t s;
    EPA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    compute_.push_back(s);
  }
  if (compute_.size() > 1) busy_.assign((size_t)cfg_.num_envs, 0);
  free_.resize(compute_.size());
  EPA_HIP(hipStreamCreateWithFlags(&h2d_stream_, hipStreamNonBlocking));
  EPA_HIP(hipStreamCreateWithFlags(&d2h_stream_, hipStreamNonBlocking));
  staging_.resize(3);
}

void Pool::InitCommon() {
  EPA_HIP(hipSetDevice(cfg_.device));
  int n = cfg_.num_envs;
  common_.n = n;
  EPA_HIP(hipMalloc(&common_.cur_step, sizeof(int) * n));
  EPA_HIP(hipMalloc(&common_.done, n));
  int* d_env_seed = nullptr;
  if (needs_rng_) {
    EPA_HIP(hipMalloc(&common_.mt, sizeof(uint32_t) * 624 * (size_t)n));
    EPA_HIP(hipMalloc(&common_.mti, sizeof(int) * n));
    if (!cfg_.env_seed.empty()) {
      EPA_HIP(hipMalloc(&d_env_seed, sizeof(int) * n));
      EPA_HIP(hipMemcpy(d_env_seed, cfg_.env_seed.data(), sizeof(int) * n,
                        hipMemcpyHostToDevice));
    }
  }
  LaunchInitCommon(common_, cfg_.seed, d_env_seed, cfg_.env_id_offset,
                   needs_rng_, stream_);
  EPA_HIP(hipStreamSynchronize(stream_));
  if (d_env_seed) EPA_HIP(hipFree(d_env_seed));
}

Pool::~Pool() {
  (void)hipSetDevice(cfg_.device);
  if (h2d_stream_) (void)hipStreamSynchronize(h2d_stream_);
  for (hipStream_t s : compute_) (void)hipStreamSynchronize(s);
  if (d2h_stream_) (void)hipStreamSynchronize(d2h_stream_);
  for (hipEvent_t e : join_ev_) (void)hipEventDestroy(e);
  for (auto& b : all_) {
    if (b->dbuf) (void)hipFree(b->dbuf);
    if (b->done) (void)hipEventDestroy(b->done);
  }
  for (auto& s : staging_) {
    if (s.h) (void)hipHostFree(s.h);
    if (s.d) (void)hipFree(s.d);
    if (s.free_ev) (void)hipEventDestroy(s.free_ev);
    if (s.h2d_ev) (void)hipEventDestroy(s.h2d_ev);
  }
  for (auto& t : timers_) {
    (void)hipEventDestroy(t.first);
    (void)hipEventDestroy(t.second);
  }
  for (auto& t : timer_pool_) (void)hipEventDestroy(t);
  if (win0_) (void)hipEventDestroy(win0_);
  if (win1_) (void)hipEventDestroy(win1_);
  if (recv_stage_) (void)hipHostFree(recv_stage_);
  if (order_ev_) (void)hipEventDestroy(order_ev_);
  if (common_.cur_step) (void)hipFree(common_.cur_step);
  if (common_.done) (void)hipFree(common_.done);
  if (stack_ring_) (void)hipFree(stack_ring_);
  if (stack_head_) (void)hipFree(stack_head_);
  if (stack_tmp_) (void)hipFree(stack_tmp_);
  if (common_.mt) (void)hipFree(common_.mt);
  if (common_.mti) (void)hipFree(common_.mti);
  for (hipStream_t s : compute_) (void)hipStreamDestroy(s);
  if (h2d_stream_) (void)hipStreamDestroy(h2d_stream_);
  if (d2h_stream_) (void)hipStreamDestroy(d2h_stream_);
}

void WaveTrace::Init(const char* env, size_t n_waves, hipStream_t s) {
  const char* f = getenv(env);
  if (!f || !*f) return;
  file = f;
  waves = n_waves;
  EPA_HIP(hipMalloc(&d, sizeof(long long) * 6 * waves));
  EPA_HIP(hipMemsetAsync(d, 0, sizeof(long long) * 6 * waves, s));
}
void WaveTrace::DumpAndFree() {
  if (!d) return;
  std::vector<long long> h(6 * waves);
  if (hipMemcpy(h.data(), d, sizeof(long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
    if (FILE* f = fopen(file.c_str(), "wb")) {
      fwrite(h.data(), sizeof(long long), h.size(), f);
      fclose(f);
    }
  }
  (void)hipFree(d);
  d = nullptr;
}

// ---- generic observation frame stack ---------------------------------------------
std::vector<int> StackedObsShape(const Config& cfg, int nobs) {
  int s = (int)cfg.Get("frame_stack", 1);
  if (s < 1) throw std::invalid_argument("frame_stack must be greater than 0");
  if (s == 1) return {nobs};
  return {s, nobs};
}

// FrameStackBuffer::Commit (frame_stack.h:109-135) as a ring: a reset fills every
// slot with the new frame, a step overwrites the oldest one.
__global__ void ObsStackKernel(const double* __restrict__ tmp, double* __restrict__ out,
                               const int* __restrict__ elapsed, const int* __restrict__ ids,
                               int id_offset, int k, int nobs, int S, double* ring,
                               const int* __restrict__ head) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k * nobs) return;
  const int row = t / nobs, i = t - row * nobs;
  const int e = ids ? ids[row] - id_offset : row;
  const double x = tmp[t];
  double* r = ring + (size_t)e * S * nobs;
  double* o = out + (size_t)row * S * nobs;
  if (elapsed[row] == 0) {  // first observation of an episode
    for (int f = 0; f < S; ++f) {
      r[f * nobs + i] = x;
      o[f * nobs + i] = x;
    }
  } else {
    const int h = head[e];
    r[h * nobs + i] = x;
    for (int f = 0; f < S - 1; ++f) o[f * nobs + i] = r[((h + 1 + f) % S) * nobs + i];
    o[(S - 1) * nobs + i] = x;
  }
}
__global__ void ObsStackAdvanceKernel(const int* __restrict__ elapsed, const int* __restrict__ ids,
                                      int id_offset, int k, int S, int* head) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= k) return;
  const int e = ids ? ids[row] - id_offset : row;
  head[e] = elapsed[row] == 0 ? 0 : (head[e] + 1) % S;
}

void Pool::EnableObsStack() {
  const KeySpec& obs = keys_[kNumCommonKeys];
  if (obs.name != "obs" || obs.dtype != EPA_F64) {
    throw std::runtime_error("EnableObsStack: first env key must be a float64 `obs`");
  }
  if (obs.shape.size() != 2) return;  // frame_stack == 1: nothing to do
  stack_s_ = obs.shape[0];
  stack_nobs_ = obs.shape[1];
  size_t n = cfg_.num_envs;
  EPA_HIP(hipMalloc(&stack_ring_, sizeof(double) * n * stack_s_ * stack_nobs_));
  EPA_HIP(hipMemsetAsync(stack_ring_, 0, sizeof(double) * n * stack_s_ * stack_nobs_, stream_));
  EPA_HIP(hipMalloc(&stack_head_, sizeof(int) * n));
  EPA_HIP(hipMemsetAsync(stack_head_, 0, sizeof(int) * n, stream_));
  EPA_HIP(hipMalloc(&stack_tmp_, sizeof(double) * n * stack_nobs_));
}

Batch* Pool::AcquireBatch(int k) {
  Batch* b = nullptr;
  // a block is always written by kernels of ONE compute stream (the one it was created for): the
  // next launch into a recycled block is stream-ordered behind everything that stream did with it
  auto& fl = free_[rr_];
  if (!fl.empty()) {
    b = fl.back();
    fl.pop_back();
  } else {
    all_.push_back(std::make_unique<Batch>());
    b = all_.back().get();
    b->cap_rows = cfg_.num_envs;
    size_t total = 0;
    for (auto& key : keys_) total += Align(b->cap_rows * key.row_bytes());
    EPA_HIP(hipMalloc(&b->dbuf, total));
    EPA_HIP(hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
    b->stream_idx = (int)rr_;
  }
  b->k = k;
  b->consumed = 0;
  b->offsets.resize(keys_.size());
  size_t off = 0;
  for (size_t i = 0; i < keys_.size(); ++i) {
    b->offsets[i] = off;
    off += Align((size_t)k * keys_[i].row_bytes());
  }
  return b;
}

void Pool::ReleaseBatch(Batch* b) { free_[(size_t)b->stream_idx].push_back(b); }

OutPtrs Pool::PtrsOf(const Batch& b) const {
  OutPtrs o{};
  for (size_t i = 0; i < keys_.size(); ++i) o.p[i] = b.dbuf + b.offsets[i];
  return o;
}

void Pool::Enqueue(const int* d_ids, int k, const void* d_action, bool force) {
  Batch* b = AcquireBatch(k);
  hipEvent_t t0 = nullptr, t1 = nullptr;
  if (timing_ == 2) {  // window timing: no event between the launches (an event pair per launch keeps
    // consecutive step kernels ~12 us apart on this runtime)
    if (!win_open_) {
      if (win0_ == nullptr) {
        EPA_HIP(hipEventCreate(&win0_));
        EPA_HIP(hipEventCreate(&win1_));
      }
      EPA_HIP(hipEventRecord(win0_, stream_));
      win_open_ = true;
      win_launches_ = 0;
    }
    ++win_launches_;
  } else if (timing_ == 1) {
    auto get = [&]() {
      hipEvent_t ev;
      if (!timer_pool_.empty()) {
        ev = timer_pool_.back();
        timer_pool_.pop_back();
      } else {
        EPA_HIP(hipEventCreate(&ev));
      }
      return ev;
    };
    t0 = get();
    t1 = get();
    EPA_HIP(hipEventRecord(t0, stream_));
  }
  OutPtrs out = PtrsOf(*b);
  void* stacked_obs = out.p[kNumCommonKeys];
  if (stack_s_ > 1) out.p[kNumCommonKeys] = stack_tmp_;  // the kernel writes one frame per row
  Launch(d_ids, k, d_action, force, out);
  EPA_HIP(hipGetLastError());
  if (stack_s_ > 1) {
    const int* elapsed = static_cast<const int*>(out.p[2]);  // "elapsed_step": 0 on a reset
    const int total = k * stack_nobs_;
    hipLaunchKernelGGL(ObsStackKernel, dim3((total + 255) / 256), dim3(256), 0, stream_,
                       stack_tmp_, static_cast<double*>(stacked_obs), elapsed, d_ids,
                       cfg_.env_id_offset, k, stack_nobs_, stack_s_, stack_ring_, stack_head_);
    hipLaunchKernelGGL(ObsStackAdvanceKernel, dim3((k + 255) / 256), dim3(256), 0, stream_,
                       elapsed, d_ids, cfg_.env_id_offset, k, stack_s_, stack_head_);
    EPA_HIP(hipGetLastError());
  }
  if (timing_ == 1) {
    EPA_HIP(hipEventRecord(t1, stream_));
    timers_.emplace_back(t0, t1);
  }
  EPA_HIP(hipEventRecord(b->done, stream_));
  b->stream = stream_;
  b->host_ids.clear();
  if (!busy_.empty()) {
    if (next_identity_) {
      b->host_ids.resize((size_t)k);
      for (int i = 0; i < k; ++i) b->host_ids[i] = i;
    } else if (next_host_ids_ != nullptr) {
      b->host_ids.resize((size_t)k);
      for (int i = 0; i < k; ++i) b->host_ids[i] = next_host_ids_[i] - cfg_.env_id_offset;
    }
  }
  next_host_ids_ = nullptr;
  next_identity_ = false;
  pending_.push_back(b);
}

// ---- concurrent batches ---------------------------------------------------------------
// The reference's worker threads step every queued slice in parallel (async_envpool.h:116-132):
// with batch_size < num_envs several batches are in flight and none depends on another, because an
// env is either queued / stepping or waiting to be received (an action can only be sent for an env
// that recv handed out).  Here an async-mode pool owns several compute streams, each an
// independent pipeline with its own result blocks, and a launch is placed so that stream order
// alone gives it what it depends on -- the previous step of ITS OWN envs:
//  * device path: the env ids of a send are (normally) the `info:env_id` array of a batch that
//    recv_device handed out; the launch goes on THAT batch's stream, behind the kernel that produced
//    the ids and the state, and ahead of whatever will recycle the block.  No event is waited for.
//    Ids from anywhere else: round robin, ordered behind every handed-out batch whose kernel has not
//    completed (`frontier_`);
//  * host path: recv returned the rows, so their kernel has completed (the D2H was synchronised):
//    round robin.  A caller that breaks the rule (sends an env again before receiving it; the
//    reference would race) is caught by the per-env `busy_` flags, and that launch is ordered behind
//    EVERYTHING enqueued so far -- the single-stream behaviour.
void Pool::JoinCompute(hipStream_t into) {
  if (compute_.size() < 2) return;
  while (join_ev_.size() < compute_.size()) {
    hipEvent_t e;
    EPA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    join_ev_.push_back(e);
  }
  for (size_t i = 0; i < compute_.size(); ++i) {
    if (compute_[i] == into) continue;
    EPA_HIP(hipEventRecord(join_ev_[i], compute_[i]));
    EPA_HIP(hipStreamWaitEvent(into, join_ev_[i], 0));
  }
}

void Pool::SyncCompute() {
  for (hipStream_t s : compute_) EPA_HIP(hipStreamSynchronize(s));
}

void Pool::PickStream(const int32_t* host_ids, int k, bool device_path, const void* d_env_id) {
  next_host_ids_ = nullptr;
  next_identity_ = false;
  // one stream for families with per-launch scratch, and while the generic observation stack is on
  // (its un-stacked frame buffer is indexed by the row of the launch)
  if (compute_.size() < 2 || !ConcurrentSafe() || stack_s_ > 1) {
    picked_ = false;
    return;
  }
  Batch* cont = nullptr;  // the handed-out batch this send continues
  if (device_path && d_env_id != nullptr) {
    const char* p = static_cast<const char*>(d_env_id);
    for (Batch* b : lent_) {
      if (b != nullptr && p >= b->dbuf && p < b->dbuf + b->offsets[1]) cont = b;  // inside its info:env_id
    }
  }
  const bool was_picked = picked_;
  const size_t picked_rr = rr_;
  picked_ = false;
  if (cont != nullptr) {
    rr_ = (size_t)cont->stream_idx;
  } else if (!was_picked) {
    rr_ = (rr_ + 1) % compute_.size();
  }
  stream_ = compute_[rr_];
  // WaitStream ordered the stream IT chose behind the producer; if the launch goes elsewhere, again
  if (was_picked && rr_ != picked_rr) EPA_HIP(hipStreamWaitEvent(stream_, order_ev_, 0));
  bool join = false;
  if (cont == nullptr) {
    // batches handed out by recv_device whose kernels may still be running, on other streams
    size_t keep = 0;
    for (Batch* b : frontier_) {
      if (hipEventQuery(b->done) == hipSuccess) continue;
      if (b->stream != stream_) EPA_HIP(hipStreamWaitEvent(stream_, b->done, 0));
      frontier_[keep++] = b;
    }
    frontier_.resize(keep);
    if (device_path) {
      // ... and batches recv_device has NOT handed out yet (still in pending_): a device-path send whose ids
      // do not continue a handed-out batch (identity ids, ids from elsewhere) may name their envs
      for (Batch* b : pending_) {
        if (b->stream != stream_ && hipEventQuery(b->done) != hipSuccess) {
          EPA_HIP(hipStreamWaitEvent(stream_, b->done, 0));
        }
      }
    }
    (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is not an error
  }
  if (!device_path) {
    const bool identity = host_ids == nullptr;
    for (int i = 0; i < k; ++i) {
      const int e = identity ? i : host_ids[i] - cfg_.env_id_offset;
      if (busy_[(size_t)e] != 0) join = true;  // rows of this env are still outstanding
      ++busy_[(size_t)e];
    }
    next_host_ids_ = host_ids;
    next_identity_ = identity;
  }
  if (join) JoinCompute(stream_);
}

void Pool::MarkIdle(Batch* b, int first, int count) {
  if (busy_.empty() || b->host_ids.empty()) return;
  // one outstanding row less: an env sent twice before a recv stays marked until BOTH rows were received,
  // so the send after the first recv is still joined behind the second launch
  for (int i = first; i < first + count; ++i) {
    uint32_t& c = busy_[(size_t)b->host_ids[(size_t)i]];
    if (c > 0) --c;
  }
}

Pool::Staging& Pool::NextStaging(size_t bytes) {
  Staging& s = staging_[staging_next_];
  staging_next_ = (staging_next_ + 1) % staging_.size();
  if (s.in_use) {
    EPA_HIP(hipEventSynchronize(s.free_ev));
    s.in_use = false;
  }
  if (s.bytes < bytes) {
    if (s.h) EPA_HIP(hipHostFree(s.h));
    if (s.d) EPA_HIP(hipFree(s.d));
    size_t cap = std::max(bytes, Align((size_t)cfg_.num_envs * 4) +
                                     Align((size_t)cfg_.num_envs *
                                           action_.row_bytes()));
    EPA_HIP(hipHostMalloc(&s.h, cap, hipHostMallocDefault));
    EPA_HIP(hipMalloc(&s.d, cap));
    s.bytes = cap;
  }
  if (!s.free_ev) {
    EPA_HIP(hipEventCreateWithFlags(&s.free_ev, hipEventDisableTiming));
    EPA_HIP(hipEventCreateWithFlags(&s.h2d_ev, hipEventDisableTiming));
  }
  return s;
}

void Pool::CheckIds(const int32_t* ids, int k) const {
  if (k < 0 || k > cfg_.num_envs) {
    throw std::invalid_argument("batch of " + std::to_string(k) +
                                " rows exceeds num_envs");
  }
  if (ids == nullptr) return;
  int lo = cfg_.env_id_offset, hi = lo + cfg_.num_envs;
  for (int i = 0; i < k; ++i) {
    if (ids[i] < lo || ids[i] >= hi) {
      throw std::invalid_argument("env_id " + std::to_string(ids[i]) +
                                  " out of range");
    }
  }
}

static bool IsIdentity(const int32_t* ids, int k, int offset, int n) {
  if (k != n) return false;
  for (int i = 0; i < k; ++i) {
    if (ids[i] != offset + i) return false;
  }
  return true;
}

void Pool::Send(const int32_t* env_id, int k, const void* action) {
  if (env_id == nullptr || action == nullptr) {
    throw std::invalid_argument("send: null env_id/action");
  }
  CheckIds(env_id, k);
  if (k == 0) return;
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  size_t id_bytes = Align((size_t)k * 4);
  size_t act_bytes = (size_t)k * action_.row_bytes();
  Staging& s = NextStaging(id_bytes + Align(act_bytes));
  bool identity = IsIdentity(env_id, k, cfg_.env_id_offset, cfg_.num_envs);
  PickStream(identity ? nullptr : env_id, k, false);
  size_t copy_from = identity ? id_bytes : 0;
  if (!identity) std::memcpy(s.h, env_id, (size_t)k * 4);
  std::memcpy(s.h + id_bytes, action, act_bytes);
  // upload on its own stream: it overlaps the kernel of the previous batch; the
  // step kernel only waits for this slot's upload.  Tiny batches stay on the kernel
  // stream: there the extra event round trip costs more than the overlap gains.
  const size_t up_bytes = id_bytes + act_bytes - copy_from;
  if (up_bytes >= kSplitStreamBytes) {
    EPA_HIP(hipMemcpyAsync(s.d + copy_from, s.h + copy_from, up_bytes, hipMemcpyHostToDevice,
                           h2d_stream_));
    EPA_HIP(hipEventRecord(s.h2d_ev, h2d_stream_));
    EPA_HIP(hipStreamWaitEvent(stream_, s.h2d_ev, 0));
  } else {
    EPA_HIP(hipMemcpyAsync(s.d + copy_from, s.h + copy_from, up_bytes, hipMemcpyHostToDevice,
                           stream_));
  }
  Enqueue(identity ? nullptr : reinterpret_cast<const int*>(s.d), k,
          s.d + id_bytes, false);
  EPA_HIP(hipEventRecord(s.free_ev, stream_));
  s.in_use = true;
}

void Pool::Reset(const int32_t* env_ids, int k) {
  if (env_ids == nullptr) throw std::invalid_argument("reset: null env_ids");
  CheckIds(env_ids, k);
  if (k == 0) return;
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  bool identity = IsIdentity(env_ids, k, cfg_.env_id_offset, cfg_.num_envs);
  PickStream(identity ? nullptr : env_ids, k, false);
  if (identity) {
    Enqueue(nullptr, k, nullptr, true);
    return;
  }
  size_t id_bytes = Align((size_t)k * 4);
  Staging& s = NextStaging(id_bytes);
  std::memcpy(s.h, env_ids, (size_t)k * 4);
  EPA_HIP(hipMemcpyAsync(s.d, s.h, (size_t)k * 4, hipMemcpyHostToDevice, stream_));
  Enqueue(reinterpret_cast<const int*>(s.d), k, nullptr, true);
  EPA_HIP(hipEventRecord(s.free_ev, stream_));
  s.in_use = true;
}

void Pool::SendDevice(const int32_t* d_env_id, int k, const void* d_action,
                      hipEvent_t wait_event) {
  CheckIds(nullptr, k);
  if (k == 0) return;
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  // the producer of d_action / d_env_id (a learner on another stream) recorded
  // `wait_event` after writing them: the step kernel is ordered behind it
  // (the analogue of the XLA custom call's stream ordering, core/xla.h:151-169)
  PickStream(nullptr, k, true, d_env_id);
  if (wait_event != nullptr) EPA_HIP(hipStreamWaitEvent(stream_, wait_event, 0));
  Enqueue(d_env_id, k, d_action, d_action == nullptr);
}

void Pool::WaitStream(hipStream_t producer) {
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  if (!order_ev_) EPA_HIP(hipEventCreateWithFlags(&order_ev_, hipEventDisableTiming));
  EPA_HIP(hipEventRecord(order_ev_, producer));
  if (compute_.size() > 1 && !picked_ && ConcurrentSafe() && stack_s_ == 1) {  // the stream the NEXT launch will use
    rr_ = (rr_ + 1) % compute_.size();
    stream_ = compute_[rr_];
    picked_ = true;
  }
  EPA_HIP(hipStreamWaitEvent(stream_, order_ev_, 0));
}

void Pool::ConsumerWait(hipStream_t consumer) {
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  if (lent_[0] == nullptr) {
    throw std::runtime_error("consumer_wait: no batch handed out by recv_device yet");
  }
  EPA_HIP(hipStreamWaitEvent(consumer, lent_[0]->done, 0));
}

int Pool::PendingRows() {
  std::lock_guard<std::mutex> lk(mu_);
  int rows = 0;
  for (Batch* b : pending_) rows += b->k - b->consumed;
  return rows;
}

int Pool::WantRows() {
  if (pending_.empty()) {
    throw std::runtime_error(
        "recv: nothing pending (the reference would block forever: call "
        "send/reset/async_reset first)");
  }
  bool sync_mode = cfg_.batch_size == cfg_.num_envs;
  int want = sync_mode ? pending_.front()->k - pending_.front()->consumed
                       : cfg_.batch_size;
  int avail = 0;
  for (Batch* b : pending_) avail += b->k - b->consumed;
  if (avail < want) {
    throw std::runtime_error("recv: only " + std::to_string(avail) +
                             " rows pending, batch_size is " +
                             std::to_string(want));
  }
  return want;
}

size_t Pool::RecvLayout(int rows, size_t* offsets, int n_keys) const {
  if (n_keys < (int)keys_.size()) {
    throw std::invalid_argument("recv_layout: need one offset per state key");
  }
  size_t total = 0;
  for (size_t i = 0; i < keys_.size(); ++i) {
    offsets[i] = total;
    total += Align((size_t)rows * keys_[i].row_bytes());
  }
  return total;
}

// Device -> host copy of the next `want` rows into a host block laid out by
// RecvLayout(want); waits for completion.
void Pool::CopyRowsToHost(char* dst, const std::vector<size_t>& off, int want) {
  size_t total = off.back() + Align((size_t)want * keys_.back().row_bytes());
  // small batches of a single-stream pool: copy on the kernel stream itself (no event round trip).  With
  // several compute streams `stream_` is whichever stream the NEWEST launch went to, not the producer of the
  // rows being received: the copy would queue behind an unrelated step kernel, so those pools always use the
  // download stream (ordered behind each batch's own `done` event below)
  hipStream_t cs = (total >= kSplitStreamBytes || compute_.size() > 1) ? d2h_stream_ : stream_;
  int got = 0;
  while (got < want) {
    Batch* b = pending_.front();
    int take = std::min(want - got, b->k - b->consumed);
    // the copies go on the download stream, behind the kernel that produced the
    // rows (its `done` event) but NOT behind kernels enqueued after it
    if (cs != b->stream) EPA_HIP(hipStreamWaitEvent(cs, b->done, 0));
    MarkIdle(b, b->consumed, take);
    if (got == 0 && take == want && b->consumed == 0 && take == b->k) {
      // whole batch: one D2H of the packed block (offsets coincide)
      EPA_HIP(hipMemcpyAsync(dst, b->dbuf, total, hipMemcpyDeviceToHost, cs));
    } else {
      for (size_t i = 0; i < keys_.size(); ++i) {
        size_t rb = keys_[i].row_bytes();
        EPA_HIP(hipMemcpyAsync(dst + off[i] + (size_t)got * rb,
                               b->dbuf + b->offsets[i] + (size_t)b->consumed * rb,
                               (size_t)take * rb, hipMemcpyDeviceToHost, cs));
      }
    }
    b->consumed += take;
    got += take;
    if (b->consumed == b->k) {
      pending_.pop_front();
      ReleaseBatch(b);  // safe: the copy is complete (synchronised below) before any
                        // later Send can hand the buffer to another kernel (mu_ is held)
    }
  }
  EPA_HIP(hipStreamSynchronize(cs));
}

int Pool::Recv(void* const* out_ptrs, int n_ptrs, int cap_rows) {
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  if (n_ptrs < (int)keys_.size()) {
    throw std::invalid_argument("recv: need one output pointer per state key");
  }
  int want = WantRows();
  if (cap_rows < want) {
    throw std::invalid_argument("recv: output buffers too small");
  }
  // pinned landing block laid out like a batch of `want` rows
  std::vector<size_t> off(keys_.size());
  size_t total = RecvLayout(want, off.data(), (int)off.size());
  if (recv_stage_bytes_ < total) {
    if (recv_stage_) EPA_HIP(hipHostFree(recv_stage_));
    size_t cap = 0;
    for (auto& key : keys_) cap += Align((size_t)cfg_.num_envs * key.row_bytes());
    cap = std::max(cap, total);
    EPA_HIP(hipHostMalloc(&recv_stage_, cap, hipHostMallocDefault));
    recv_stage_bytes_ = cap;
  }
  CopyRowsToHost(recv_stage_, off, want);
  for (size_t i = 0; i < keys_.size(); ++i) {
    if (out_ptrs[i] != nullptr) {
      std::memcpy(out_ptrs[i], recv_stage_ + off[i],
                  (size_t)want * keys_[i].row_bytes());
    }
  }
  return want;
}

int Pool::RecvBlock(void* block, size_t block_bytes, size_t* offsets, int n_keys) {
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  if (block == nullptr) throw std::invalid_argument("recv_block: null block");
  if (n_keys < (int)keys_.size()) {
    throw std::invalid_argument("recv_block: need one offset per state key");
  }
  int want = WantRows();
  std::vector<size_t> off(keys_.size());
  size_t total = RecvLayout(want, off.data(), (int)off.size());
  if (block_bytes < total) {
    throw std::invalid_argument("recv_block: block too small (" +
                                std::to_string(block_bytes) + " < " +
                                std::to_string(total) + " bytes)");
  }
  CopyRowsToHost(static_cast<char*>(block), off, want);
  for (size_t i = 0; i < off.size(); ++i) offsets[i] = off[i];
  return want;
}

// Per-key device -> host copies straight into caller-owned buffers (no landing block, no
// host memcpy): what a multi-GPU gather needs, where GPU g's rows land directly in ITS
// slice of one host batch (SURVEY 8e "host gather").
int Pool::RecvInto(void* const* out_ptrs, int n_ptrs, int cap_rows) {
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  if (n_ptrs < (int)keys_.size()) {
    throw std::invalid_argument("recv_into: need one output pointer per state key");
  }
  int want = WantRows();
  if (cap_rows < want) throw std::invalid_argument("recv_into: output buffers too small");
  hipStream_t cs = d2h_stream_;
  int got = 0;
  while (got < want) {
    Batch* b = pending_.front();
    int take = std::min(want - got, b->k - b->consumed);
    EPA_HIP(hipStreamWaitEvent(cs, b->done, 0));
    MarkIdle(b, b->consumed, take);
    for (size_t i = 0; i < keys_.size(); ++i) {
      if (out_ptrs[i] == nullptr) continue;
      size_t rb = keys_[i].row_bytes();
      EPA_HIP(hipMemcpyAsync(static_cast<char*>(out_ptrs[i]) + (size_t)got * rb,
                             b->dbuf + b->offsets[i] + (size_t)b->consumed * rb,
                             (size_t)take * rb, hipMemcpyDeviceToHost, cs));
    }
    b->consumed += take;
    got += take;
    if (b->consumed == b->k) {
      pending_.pop_front();
      ReleaseBatch(b);
    }
  }
  EPA_HIP(hipStreamSynchronize(cs));
  return want;
}

int Pool::RecvDevice(void** d_out_ptrs, int n_ptrs) {
  std::lock_guard<std::mutex> lk(mu_);
  if (n_ptrs < (int)keys_.size()) {
    throw std::invalid_argument("recv_device: need one pointer per state key");
  }
  if (pending_.empty()) throw std::runtime_error("recv_device: nothing pending");
  Batch* b = pending_.front();
  if (b->consumed != 0) {
    throw std::runtime_error("recv_device: batch partially consumed by recv");
  }
  pending_.pop_front();
  MarkIdle(b, 0, b->k);
  if (compute_.size() > 1) {
    // (a block is recycled two recv_device calls later: drop the entry it may still have)
    frontier_.erase(std::remove(frontier_.begin(), frontier_.end(), b), frontier_.end());
    frontier_.push_back(b);
  }
  for (size_t i = 0; i < keys_.size(); ++i) {
    d_out_ptrs[i] = b->dbuf + b->offsets[i];
  }
  if (lent_[1]) ReleaseBatch(lent_[1]);
  lent_[1] = lent_[0];
  lent_[0] = b;
  return b->k;
}

void Pool::Synchronize() {
  EPA_HIP(hipSetDevice(cfg_.device));
  EPA_HIP(hipStreamSynchronize(h2d_stream_));
  SyncCompute();
  EPA_HIP(hipStreamSynchronize(d2h_stream_));
}

void Pool::SetTiming(int mode) {
  std::lock_guard<std::mutex> lk(mu_);
  timing_ = mode;
  win_open_ = false;
  win_launches_ = 0;
}

void Pool::KernelTime(double* avg_ms, int* launches) {
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  if (win_open_) {  // mode 2: (last launch's end - first launch's start) / launches, gaps included
    JoinCompute(stream_);  // several compute streams: the window closes behind all of them
    EPA_HIP(hipEventRecord(win1_, stream_));
    EPA_HIP(hipStreamSynchronize(stream_));
    float ms = 0;
    EPA_HIP(hipEventElapsedTime(&ms, win0_, win1_));
    *launches = win_launches_;
    *avg_ms = win_launches_ > 0 ? (double)ms / win_launches_ : 0.0;
    win_open_ = false;
    win_launches_ = 0;
    return;
  }
  SyncCompute();
  double tot = 0;
  for (auto& t : timers_) {
    float ms = 0;
    EPA_HIP(hipEventElapsedTime(&ms, t.first, t.second));
    tot += ms;
    timer_pool_.push_back(t.first);
    timer_pool_.push_back(t.second);
  }
  *launches = (int)timers_.size();
  *avg_ms = timers_.empty() ? 0.0 : tot / timers_.size();
  timers_.clear();
}

void Pool::GetStateHost(const int32_t* ids, int k, double* out) {
  if (ids == nullptr || out == nullptr) throw std::invalid_argument("get_state: null argument");
  CheckIds(ids, k);
  if (k == 0) return;
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  int dim = StateDim();
  int* d_ids;
  double* d_buf;
  std::vector<int> local(ids, ids + k);
  for (auto& v : local) v -= cfg_.env_id_offset;
  if (compute_.size() > 1) SyncCompute();  // the state of envs stepping on other streams
  EPA_HIP(hipMalloc(&d_ids, sizeof(int) * k));
  EPA_HIP(hipMalloc(&d_buf, sizeof(double) * k * dim));
  EPA_HIP(hipMemcpyAsync(d_ids, local.data(), sizeof(int) * k,
                         hipMemcpyHostToDevice, stream_));
  GetState(d_ids, k, d_buf);
  EPA_HIP(hipMemcpyAsync(out, d_buf, sizeof(double) * k * dim,
                         hipMemcpyDeviceToHost, stream_));
  EPA_HIP(hipStreamSynchronize(stream_));
  EPA_HIP(hipFree(d_ids));
  EPA_HIP(hipFree(d_buf));
}

void Pool::SetStateHost(const int32_t* ids, int k, const double* in) {
  if (ids == nullptr || in == nullptr) throw std::invalid_argument("set_state: null argument");
  CheckIds(ids, k);
  if (k == 0) return;
  std::lock_guard<std::mutex> lk(mu_);
  EPA_HIP(hipSetDevice(cfg_.device));
  int dim = StateDim();
  int* d_ids;
  double* d_buf;
  std::vector<int> local(ids, ids + k);
  for (auto& v : local) v -= cfg_.env_id_offset;
  if (compute_.size() > 1) SyncCompute();
  EPA_HIP(hipMalloc(&d_ids, sizeof(int) * k));
  EPA_HIP(hipMalloc(&d_buf, sizeof(double) * k * dim));
  EPA_HIP(hipMemcpyAsync(d_ids, local.data(), sizeof(int) * k,
                         hipMemcpyHostToDevice, stream_));
  EPA_HIP(hipMemcpyAsync(d_buf, in, sizeof(double) * k * dim,
                         hipMemcpyHostToDevice, stream_));
  SetState(d_ids, k, d_buf);
  EPA_HIP(hipStreamSynchronize(stream_));
  EPA_HIP(hipFree(d_ids));
  EPA_HIP(hipFree(d_buf));
}

// ---------------------------------------------------------------------------
// family registry
// ---------------------------------------------------------------------------
const std::vector<std::string>& FamilyNames() {
  static const std::vector<std::string> names = {
      "CartPole", "Pendulum", "MountainCar", "MountainCarContinuous", "Acrobot",
      "Catch", "FrozenLake", "Taxi", "NChain", "CliffWalking", "Blackjack",
      "HalfCheetah", "Ant", "Walker2d", "InvertedPendulum", "InvertedDoublePendulum", "Reacher", "Swimmer", "Hopper", "Humanoid", "HumanoidStandup", "Pusher"};
  return names;
}

static bool Describe(const std::string& family, const Config& cfg,
                     std::vector<KeySpec>* state, KeySpec* action) {
  std::vector<KeySpec> env_keys;
  bool ok = DescribeClassicControl(family, cfg, &env_keys, action) ||
            DescribeToyText(family, cfg, &env_keys, action) ||
            DescribeMujoco(family, cfg, &env_keys, action);
  if (!ok) return false;
  *state = CommonStateKeys();
  for (auto& k : env_keys) state->push_back(k);
  return true;
}

static Pool* Make(const std::string& family, const Config& cfg) {
  Pool* p = MakeClassicControl(family, cfg);
  if (!p) p = MakeToyText(family, cfg);
  if (!p) p = MakeMujoco(family, cfg);
  return p;
}

}  // namespace epa

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
struct epa_pool {
  std::unique_ptr<epa::Pool> impl;
};

namespace {
template <typename F>
int Guard(F&& f) {
  try {
    f();
    return EPA_OK;
  } catch (const std::invalid_argument& e) {
    epa::g_last_error = e.what();
    return EPA_ERR_INVALID;
  } catch (const epa::DeviceError& e) {
    epa::g_last_error = e.what();
    return EPA_ERR_DEVICE;
  } catch (const std::exception& e) {
    epa::g_last_error = e.what();
    return EPA_ERR_RUNTIME;
  }
}

thread_local std::vector<std::string> g_key_names;

void FillKeyInfo(const std::vector<epa::KeySpec>& keys, epa_key_info* out,
                 int cap, int* n) {
  *n = (int)keys.size();
  g_key_names.clear();
  for (auto& k : keys) g_key_names.push_back(k.name);
  for (int i = 0; i < (int)keys.size() && i < cap; ++i) {
    out[i].name = g_key_names[i].c_str();
    out[i].dtype = keys[i].dtype;
    out[i].ndim = (int)keys[i].shape.size();
    for (int d = 0; d < 4; ++d) {
      out[i].shape[d] = d < out[i].ndim ? keys[i].shape[d] : 0;
    }
    out[i].row_elems = keys[i].row_elems();
    out[i].row_bytes = keys[i].row_bytes();
  }
}

// Describe must not require a valid num_envs/device.
epa::Config LenientConfig(const epa_config* c) {
  epa::Config r;
  if (c != nullptr) {
    for (int i = 0; i < c->n_params; ++i) {
      r.params[c->param_keys[i]] = c->param_values[i];
    }
    if (c->num_envs > 0) r.num_envs = c->num_envs;
  }
  return r;
}
}  // namespace

extern "C" {

int epa_num_families(void) { return (int)epa::FamilyNames().size(); }
const char* epa_family_name(int i) {
  auto& n = epa::FamilyNames();
  return (i >= 0 && i < (int)n.size()) ? n[i].c_str() : nullptr;
}

int epa_describe_state(const char* family, const epa_config* cfg,
                       epa_key_info* keys, int cap, int* n) {
  return Guard([&] {
    std::vector<epa::KeySpec> st;
    epa::KeySpec act;
    if (!epa::Describe(family, LenientConfig(cfg), &st, &act)) {
      throw std::invalid_argument(std::string("unknown env family: ") + family);
    }
    FillKeyInfo(st, keys, cap, n);
  });
}

int epa_describe_action(const char* family, const epa_config* cfg,
                        epa_key_info* keys, int cap, int* n) {
  return Guard([&] {
    std::vector<epa::KeySpec> st;
    epa::KeySpec act;
    if (!epa::Describe(family, LenientConfig(cfg), &st, &act)) {
      throw std::invalid_argument(std::string("unknown env family: ") + family);
    }
    // common_action_spec, envpool/core/env_spec.h:32-35
    std::vector<epa::KeySpec> a = {{"env_id", EPA_I32, {}},
                                   {"players.env_id", EPA_I32, {}},
                                   act};
    FillKeyInfo(a, keys, cap, n);
  });
}

int epa_create(const char* family, const epa_config* cfg, epa_pool** out) {
  return Guard([&] {
    if (out == nullptr || family == nullptr) {
      throw std::invalid_argument("epa_create: null argument");
    }
    epa::Config c = epa::Config::From(cfg);
    epa::Pool* p = epa::Make(family, c);
    if (p == nullptr) {
      throw std::invalid_argument(std::string("unknown env family: ") + family);
    }
    *out = new epa_pool{std::unique_ptr<epa::Pool>(p)};
  });
}

int epa_destroy(epa_pool* pool) {
  return Guard([&] { delete pool; });
}

int epa_send(epa_pool* pool, const int32_t* env_id, int32_t k,
             const void* action) {
  return Guard([&] { pool->impl->Send(env_id, k, action); });
}

int epa_reset(epa_pool* pool, const int32_t* env_ids, int32_t k) {
  return Guard([&] { pool->impl->Reset(env_ids, k); });
}

int epa_recv(epa_pool* pool, void* const* out_ptrs, int32_t n_ptrs,
             int32_t cap_rows, int32_t* k_out) {
  return Guard([&] { *k_out = pool->impl->Recv(out_ptrs, n_ptrs, cap_rows); });
}

int epa_recv_layout(epa_pool* pool, int32_t rows, size_t* offsets, int32_t n_keys,
                    size_t* total_bytes) {
  return Guard([&] {
    if (rows < 0) throw std::invalid_argument("recv_layout: rows < 0");
    *total_bytes = pool->impl->RecvLayout(rows, offsets, n_keys);
  });
}

int epa_recv_block(epa_pool* pool, void* block, size_t block_bytes,
                   size_t* offsets, int32_t n_keys, int32_t* k_out) {
  return Guard([&] { *k_out = pool->impl->RecvBlock(block, block_bytes, offsets, n_keys); });
}

int epa_recv_into(epa_pool* pool, void* const* out_ptrs, int32_t n_ptrs,
                  int32_t cap_rows, int32_t* k_out) {
  return Guard([&] { *k_out = pool->impl->RecvInto(out_ptrs, n_ptrs, cap_rows); });
}

int epa_pending_rows(epa_pool* pool, int32_t* rows) {
  return Guard([&] { *rows = pool->impl->PendingRows(); });
}

int epa_send_device(epa_pool* pool, const int32_t* d_env_id, int32_t k,
                    const void* d_action, void* wait_event) {
  return Guard([&] {
    pool->impl->SendDevice(d_env_id, k, d_action, static_cast<hipEvent_t>(wait_event));
  });
}

int epa_wait_stream(epa_pool* pool, void* producer_stream) {
  return Guard([&] { pool->impl->WaitStream(static_cast<hipStream_t>(producer_stream)); });
}

int epa_consumer_wait(epa_pool* pool, void* consumer_stream) {
  return Guard([&] { pool->impl->ConsumerWait(static_cast<hipStream_t>(consumer_stream)); });
}

int epa_recv_device(epa_pool* pool, void** d_out_ptrs, int32_t n_ptrs,
                    int32_t* k_out) {
  return Guard([&] { *k_out = pool->impl->RecvDevice(d_out_ptrs, n_ptrs); });
}

void* epa_stream(epa_pool* pool) { return (void*)pool->impl->stream(); }

int epa_synchronize(epa_pool* pool) {
  return Guard([&] { pool->impl->Synchronize(); });
}

int epa_set_timing(epa_pool* pool, int32_t enabled) {
  return Guard([&] { pool->impl->SetTiming(enabled < 0 ? 0 : (enabled > 2 ? 1 : (int)enabled)); });
}

int epa_kernel_time_ms(epa_pool* pool, double* avg_ms, int32_t* launches) {
  return Guard([&] { pool->impl->KernelTime(avg_ms, launches); });
}

int epa_state_dim(epa_pool* pool, int32_t* dim) {
  return Guard([&] { *dim = pool->impl->StateDim(); });
}

int epa_get_state(epa_pool* pool, const int32_t* env_ids, int32_t k,
                  double* out) {
  return Guard([&] { pool->impl->GetStateHost(env_ids, k, out); });
}

int epa_set_state(epa_pool* pool, const int32_t* env_ids, int32_t k,
                  const double* in) {
  return Guard([&] { pool->impl->SetStateHost(env_ids, k, in); });
}

int epa_atari_create(const epa_atari_config* cfg, epa_pool** out) {
  return Guard([&] {
    if (cfg == nullptr || out == nullptr || cfg->rom_path == nullptr) {
      throw std::invalid_argument("epa_atari_create: null argument");
    }
    epa::Config c = epa::Config::From(&cfg->base);
    epa::Pool* p = epa::MakeAtari(c, cfg->rom_path, cfg->emulator_lib ? cfg->emulator_lib : "");
    *out = new epa_pool{std::unique_ptr<epa::Pool>(p)};
  });
}

int epa_atari_num_actions(const epa_atari_config* cfg, int32_t* n) {
  return Guard([&] {
    if (cfg == nullptr || n == nullptr || cfg->rom_path == nullptr) {
      throw std::invalid_argument("epa_atari_num_actions: null argument");
    }
    *n = epa::AtariNumActions(LenientConfig(&cfg->base), cfg->rom_path,
                              cfg->emulator_lib ? cfg->emulator_lib : "");
  });
}

int epa_pool_state_keys(epa_pool* pool, epa_key_info* keys, int cap, int* n) {
  return Guard([&] { FillKeyInfo(pool->impl->state_keys(), keys, cap, n); });
}

int epa_pool_action_keys(epa_pool* pool, epa_key_info* keys, int cap, int* n) {
  return Guard([&] {
    std::vector<epa::KeySpec> a = {{"env_id", EPA_I32, {}},
                                   {"players.env_id", EPA_I32, {}},
                                   pool->impl->action_key()};
    FillKeyInfo(a, keys, cap, n);
  });
}

const char* epa_last_error(void) { return epa::g_last_error.c_str(); }
const char* epa_version(void) { return "envpool_amd 0.1 (gfx950)"; }

int epa_device_count(int32_t* n) {
  return Guard([&] {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) c = 0;
    *n = c;
  });
}

void* epa_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void epa_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

}  // extern "C"
