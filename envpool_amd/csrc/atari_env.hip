// Atari end to end: host-side emulator workers + GPU observation post-process behind the
// same C ABI as every other family (epa_send / epa_recv / epa_reset on an epa_pool).
//
// Replaces, for the whole pool:
//   AtariEnv::{Reset,Step,WriteState}     envpool/atari/atari_env.h:167-293
//   Env::{EnvStep,PreProcess,Allocate}    envpool/core/env.h:184-256 (bookkeeping rows)
//   AsyncEnvPool worker loop / queues     envpool/core/async_envpool.h:42-238
// The north star keeps ALE on the host: emulation cannot move to the GPU, so this family
// is the one place where the engine still runs a thread pool.  What changed against the
// reference:
//   * the emulator is reached through a plugin table (include/envpool_amd_emulator.h);
//   * a worker only EMULATES: it runs the frame_skip loop, keeps the last two raw screens
//     (palette indices, 1 byte per pixel) and the scalars of the step.  The colour palette,
//     the two-frame max-pool, cv::resize, the transpose of the colour planes and the frame
//     stack -- PushStack + Resize + the obs part of WriteState, atari_env.h:283-346 -- run as
//     ONE HIP kernel per batch (atari_post.hip) behind pinned hipMemcpyAsync copies that
//     are chunked so upload, kernel and download overlap;
//   * rows are claimed in a per-batch pinned block (first-come in async mode, send order in
//     sync mode: state_buffer_queue.h:123-163), so the upload needs no gather.
// Semantics kept: first step of an env is a reset; a step on a finished env resets it;
// `reset = force_reset || IsDone()` (async_envpool.h:127); noop / FIRE resets; episodic
// life; reward clipping; zero_discount_on_life_loss; the elapsed_step / trunc / discount
// overrides of WriteState; seeds `seed + env_id` or env_seed[] (env.h:101-110) for both
// the noop RNG and the emulator.
#include <dlfcn.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <random>
#include <thread>

#include "../../include/envpool_amd_emulator.h"
#include "engine.h"

namespace epa {
namespace {

constexpr int kRawH = EPA_EMULATOR_SCREEN_H, kRawW = EPA_EMULATOR_SCREEN_W;
constexpr int kRawSize = kRawH * kRawW;
constexpr int kRam = EPA_EMULATOR_RAM;

struct Plugin {
  void* dl{nullptr};
  const epa_emulator_api* api{nullptr};
  explicit Plugin(const std::string& path) {
    if (path.empty()) {
      throw std::invalid_argument(
          "Atari: no emulator plugin given (config key `emulator_lib`): ALE is not part of this "
          "repository -- build integration/ale_adapter against ALE 0.11.2, see INTEGRATION.md");
    }
    dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dl) throw std::invalid_argument(std::string("Atari: cannot load emulator plugin: ") + dlerror());
    auto get = reinterpret_cast<epa_emulator_get_api_fn>(dlsym(dl, "epa_emulator_get_api"));
    if (!get) throw std::invalid_argument("Atari: " + path + " does not export epa_emulator_get_api");
    api = get();
    if (!api || api->abi != EPA_EMULATOR_ABI) {
      throw std::invalid_argument("Atari: emulator plugin ABI mismatch");
    }
  }
  ~Plugin() {
    if (dl) dlclose(dl);
  }
};

struct AtariCfg {
  int stack_num, frame_skip, noop_max, img_h, img_w, mode, difficulty;
  bool zero_discount_on_life_loss, episodic_life, reward_clip, use_fire_reset;
  bool full_action_space, use_inter_area, gray_scale;
  float repeat_action_probability;
  int num_threads;
  static AtariCfg From(const Config& c) {  // AtariEnvFns::DefaultConfig, atari_env.h:52-63
    AtariCfg a;
    a.stack_num = (int)c.Get("stack_num", 4);
    a.frame_skip = (int)c.Get("frame_skip", 4);
    a.noop_max = (int)c.Get("noop_max", 30);
    a.zero_discount_on_life_loss = c.Get("zero_discount_on_life_loss", 0) != 0;
    a.episodic_life = c.Get("episodic_life", 0) != 0;
    a.reward_clip = c.Get("reward_clip", 0) != 0;
    a.use_fire_reset = c.Get("use_fire_reset", 1) != 0;
    a.img_h = (int)c.Get("img_height", 84);
    a.img_w = (int)c.Get("img_width", 84);
    a.mode = (int)c.Get("mode", -1);
    a.difficulty = (int)c.Get("difficulty", -1);
    a.full_action_space = c.Get("full_action_space", 0) != 0;
    a.repeat_action_probability = (float)c.Get("repeat_action_probability", 0.0);
    a.use_inter_area = c.Get("use_inter_area_resize", 1) != 0;
    a.gray_scale = c.Get("gray_scale", 1) != 0;
    a.num_threads = (int)c.Get("num_threads", 0);
    if (a.stack_num < 1 || a.frame_skip < 1 || a.noop_max < 1) {
      throw std::invalid_argument("Atari: stack_num, frame_skip and noop_max must be >= 1");
    }
    return a;
  }
};

std::vector<KeySpec> AtariKeys(const AtariCfg& a) {  // StateSpec, atari_env.h:64-75
  return {{"obs", EPA_U8, {a.stack_num * (a.gray_scale ? 1 : 3), a.img_h, a.img_w}},
          {"info:lives", EPA_I32, {}},
          {"info:reward", EPA_F32, {}},
          {"info:terminated", EPA_I32, {}},
          {"info:ram", EPA_U8, {kRam}}};
}

epa_emulator_config EmuCfg(const std::string& rom, int seed, const AtariCfg& a) {
  epa_emulator_config e;
  e.rom_path = rom.c_str();
  e.random_seed = seed;
  e.repeat_action_probability = a.repeat_action_probability;
  e.mode = a.mode;
  e.difficulty = a.difficulty;
  return e;
}

// one env: the host half of AtariEnv
struct Env {
  void* emu{nullptr};
  std::mt19937 gen;                            // Env::gen_, env.h:78
  std::uniform_int_distribution<> dist_noop;   // atari_env.h:117,134
  int elapsed_step{0};                         // max_episode_steps + 1 until the first reset (:125)
  int current_step{-1};                        // Env::current_step_, env.h:86
  bool done{true};                             // :107
  int lives{0};
  std::vector<uint8_t> frame[2];               // raw screens behind maxpool_buf_[0/1]
};

// one output batch: rows are claimed by the workers, then post-processed and handed out
struct OutBatch {
  int rows{0};                 // rows this batch will hold
  std::atomic<int> claimed{0}, finished{0};
  // pinned staging, row major
  uint8_t* frames{nullptr};    // [cap][2][kRawSize]
  uint8_t* flags{nullptr};     // [cap] PushStack flags (include/envpool_amd.h)
  int32_t* env_id{nullptr};
  int32_t *elapsed{nullptr}, *step_type{nullptr}, *lives{nullptr}, *terminated{nullptr};
  uint8_t *done{nullptr}, *trunc{nullptr};
  float *reward{nullptr}, *discount{nullptr}, *info_reward{nullptr};
  uint8_t* ram{nullptr};       // [cap][128]
  int32_t* local_id{nullptr};  // env index inside the pool (ring of the frame stack)
  double t_send{0}, t_done{0};
};

struct Task {
  int env;
  int action;
  bool force_reset;
  OutBatch* batch;  // sync mode: the batch of this send; async: nullptr (claimed at completion)
  int row;          // sync mode: position in the send
};

class AtariPool : public Pool {
 public:
  AtariPool(const Config& cfg, const std::string& rom, const std::string& emulator_lib)
      : Pool(cfg, AtariKeys(AtariCfg::From(cfg)), KeySpec{"action", EPA_I32, {}}, false),
        a_(AtariCfg::From(cfg)),
        plugin_(emulator_lib),
        rom_(rom) {
    const int n = cfg.num_envs;
    sync_ = cfg.batch_size <= 0 || cfg.batch_size >= n;
    batch_size_ = sync_ ? n : cfg.batch_size;
    envs_.resize(n);
    // num_threads = 0: the reference's rule, min(batch_size, hardware_concurrency)
    // (async_envpool.h:115-117) -- with a real emulator (ALE: ~150 us per frame) the emulate phase is
    // the step and wants every core.  An emulator that costs ~1 us per frame (the synthetic
    // console of the tests) is better served by fewer workers (emulate phase of a 1024-env step:
    // 0.9 ms with 32 workers, 4.3 ms with 128, 5.0 ms with 256 on a 256-thread host,
    // profiles/archive/r2e): set num_threads, or EPA_ATARI_THREADS for a process-wide default.
    int nthreads = a_.num_threads;
    if (nthreads <= 0) {
      const char* ev = getenv("EPA_ATARI_THREADS");
      nthreads = ev ? atoi(ev) : 0;
    }
    if (nthreads <= 0) nthreads = std::max(1, (int)std::thread::hardware_concurrency());
    nthreads = std::max(1, std::min(nthreads, batch_size_));
    // emulators (ROM loading is the slow part: in parallel)
    std::atomic<int> next{0};
    std::string first_error;
    std::mutex err_mu;
    auto load = [&] {
      for (int i = next++; i < n; i = next++) {
        Env& e = envs_[i];
        const int seed = cfg.env_seed.empty() ? cfg.seed + cfg.env_id_offset + i : cfg.env_seed[i];
        epa_emulator_config ec = EmuCfg(rom_, seed, a_);
        e.emu = plugin_.api->create(&ec);
        if (!e.emu) {
          std::lock_guard<std::mutex> lk(err_mu);
          if (first_error.empty()) first_error = plugin_.api->last_error();
          continue;
        }
        e.gen.seed((unsigned)seed);
        e.dist_noop = std::uniform_int_distribution<>(0, a_.noop_max - 1);
        e.elapsed_step = cfg.max_episode_steps + 1;
        e.frame[0].assign(kRawSize, 0);
        e.frame[1].assign(kRawSize, 0);
      }
    };
    {
      std::vector<std::thread> th;
      for (int t = 0; t < nthreads; ++t) th.emplace_back(load);
      for (auto& t : th) t.join();
    }
    if (!first_error.empty()) {
      DestroyEmus();
      throw std::invalid_argument("Atari: emulator create failed: " + first_error);
    }
    // action set (atari_env.h:146-159)
    int32_t codes[64];
    const int na = plugin_.api->action_set(envs_[0].emu, a_.full_action_space ? 1 : 0, codes, 64);
    action_set_.assign(codes, codes + std::min(na, 64));
    if (a_.use_fire_reset) {
      for (int c : action_set_) fire_reset_ = fire_reset_ || c == 1;
    }
    // colour palette -> device post-process
    uint8_t gray[256], rgb[256][3], lut[3 * 256];
    plugin_.api->palette(envs_[0].emu, gray, rgb);
    if (a_.gray_scale) {
      std::memcpy(lut, gray, 256);
    } else {
      for (int i = 0; i < 256; ++i) {
        lut[i] = rgb[i][0];
        lut[256 + i] = rgb[i][1];
        lut[512 + i] = rgb[i][2];
      }
    }
    if (epa_atari_post_create_ex(n, a_.stack_num, kRawH, kRawW, a_.img_h, a_.img_w,
                                 a_.use_inter_area ? 1 : 0, a_.gray_scale ? 1 : 0, lut, cfg.device,
                                 &post_) != EPA_OK) {
      DestroyEmus();
      throw std::invalid_argument(std::string("Atari: ") + epa_last_error());
    }
    ring_.resize((size_t)4 * n);
    slot_free_.reset(new std::atomic<uint64_t>[ring_.size()]);
    for (size_t i = 0; i < ring_.size(); ++i) slot_free_[i].store(i, std::memory_order_relaxed);
    for (int t = 0; t < nthreads; ++t) workers_.emplace_back([this] { WorkerLoop(); });
  }

  ~AtariPool() override {
    if (getenv("EPA_ATARI_PROFILE") && prof_batches_ > 0) {
      fprintf(stderr, "atari pool: %ld batches, per batch: emulate %.3f ms (send -> last row), "
              "recv called %.3f ms after that, post-process + copies %.3f ms; %zu workers\n",
              prof_batches_, 1e3 * prof_emulate_ / prof_batches_, 1e3 * prof_wait_ / prof_batches_,
              1e3 * prof_deliver_ / prof_batches_, workers_.size());
    }
    stop_.store(true, std::memory_order_release);
    sleepers_.fetch_add(1);  // force the wake-up syscall
    WakeWorkers();
    for (auto& t : workers_) t.join();
    if (post_) epa_atari_post_destroy(post_);
    for (auto& b : all_batches_) FreeBatch(b.get());
    DestroyEmus();
  }

  int num_actions() const { return (int)action_set_.size(); }

  // ---- Pool interface -------------------------------------------------------------
  void Send(const int32_t* env_id, int k, const void* action) override {
    if (k < 0 || (k > 0 && (!env_id || !action))) throw std::invalid_argument("send: null argument");
    CheckIds(env_id, k);
    const int32_t* act = static_cast<const int32_t*>(action);
    for (int i = 0; i < k; ++i) {
      if (act[i] < 0 || act[i] >= (int)action_set_.size()) {
        throw std::invalid_argument("send: action out of range");
      }
    }
    Enqueue(env_id, k, act, false);
  }
  void Reset(const int32_t* env_ids, int k) override {
    if (k < 0 || (k > 0 && !env_ids)) throw std::invalid_argument("reset: null argument");
    CheckIds(env_ids, k);
    Enqueue(env_ids, k, nullptr, true);
  }
  void SendInto(const int32_t* env_id, int k, const void* action, void*, size_t) override {
    Send(env_id, k, action);  // (the host executor writes its own result blocks)
  }
  void SendDevice(const int32_t*, int, const void*, hipEvent_t) override {
    throw std::runtime_error("Atari: the emulator runs on the host, there is no device-resident send");
  }
  int RecvDevice(void**, int) override {
    throw std::runtime_error("Atari: recv_device is not available (host-side emulator)");
  }
  int PendingRows() override {
    std::lock_guard<std::mutex> lk(b_mu_);
    if (!sync_) return batch_size_;
    if (out_queue_.empty()) throw std::runtime_error("recv: nothing pending");
    return out_queue_.front()->rows;
  }
  int Recv(void* const* out_ptrs, int n_ptrs, int cap_rows) override {
    return RecvInto(out_ptrs, n_ptrs, cap_rows);
  }
  int RecvInto(void* const* out_ptrs, int n_ptrs, int cap_rows) override {
    if (n_ptrs < (int)keys_.size()) {
      throw std::invalid_argument("recv: need one output pointer per state key");
    }
    OutBatch* b = WaitFront();
    if (cap_rows < b->rows) throw std::invalid_argument("recv: output buffers too small");
    Deliver(b, out_ptrs);
    return PopFront();
  }
  int RecvBlock(void* block, size_t block_bytes, size_t* offsets, int n_keys) override {
    if (n_keys < (int)keys_.size()) {
      throw std::invalid_argument("recv_block: need one offset slot per state key");
    }
    OutBatch* b = WaitFront();
    std::vector<size_t> off(keys_.size());
    const size_t total = RecvLayout(b->rows, off.data(), (int)off.size());
    if (block_bytes < total) throw std::invalid_argument("recv_block: block too small");
    std::vector<void*> ptrs(keys_.size());
    for (size_t i = 0; i < off.size(); ++i) {
      ptrs[i] = static_cast<char*>(block) + off[i];
      offsets[i] = off[i];
    }
    Deliver(b, ptrs.data());
    return PopFront();
  }
  void Synchronize() override {
    Pool::Synchronize();
    if (post_) EPA_HIP(hipStreamSynchronize((hipStream_t)epa_atari_post_stream(post_)));
  }

  int StateDim() const override { return 0; }
  void GetState(const int*, int, double*) override {
    throw std::runtime_error("Atari: emulator state is opaque (no get_state)");
  }
  void SetState(const int*, int, const double*) override {
    throw std::runtime_error("Atari: emulator state is opaque (no set_state)");
  }

 protected:
  void Launch(const int*, int, const void*, bool, const OutPtrs&) override {
    throw std::runtime_error("Atari: no stream-ordered launch (host-side emulator)");
  }

 private:
  void DestroyEmus() {
    for (Env& e : envs_) {
      if (e.emu) plugin_.api->destroy(e.emu);
      e.emu = nullptr;
    }
  }

  // ---- batches ----------------------------------------------------------------------
  OutBatch* NewBatch(int rows) {  // b_mu_ held
    OutBatch* b = nullptr;
    if (!free_batches_.empty()) {
      b = free_batches_.back();
      free_batches_.pop_back();
    } else {
      all_batches_.emplace_back(new OutBatch());
      b = all_batches_.back().get();
      EPA_HIP(hipSetDevice(cfg_.device));
      const size_t cap = cfg_.num_envs;
      auto pin = [&](size_t bytes) {
        void* p = nullptr;
        EPA_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        return p;
      };
      b->frames = (uint8_t*)pin(cap * 2 * kRawSize);
      b->flags = (uint8_t*)pin(cap);
      b->env_id = (int32_t*)pin(cap * 4);
      b->local_id = (int32_t*)pin(cap * 4);
      b->elapsed = (int32_t*)pin(cap * 4);
      b->step_type = (int32_t*)pin(cap * 4);
      b->lives = (int32_t*)pin(cap * 4);
      b->terminated = (int32_t*)pin(cap * 4);
      b->done = (uint8_t*)pin(cap);
      b->trunc = (uint8_t*)pin(cap);
      b->reward = (float*)pin(cap * 4);
      b->discount = (float*)pin(cap * 4);
      b->info_reward = (float*)pin(cap * 4);
      b->ram = (uint8_t*)pin(cap * kRam);
    }
    b->rows = rows;
    b->claimed = 0;
    b->finished = 0;
    return b;
  }
  void FreeBatch(OutBatch* b) {
    for (void* p : {(void*)b->frames, (void*)b->flags, (void*)b->env_id, (void*)b->local_id,
                    (void*)b->elapsed, (void*)b->step_type, (void*)b->lives, (void*)b->terminated,
                    (void*)b->done, (void*)b->trunc, (void*)b->reward, (void*)b->discount,
                    (void*)b->info_reward, (void*)b->ram}) {
      if (p) (void)hipHostFree(p);
    }
  }

  void Enqueue(const int32_t* ids, int k, const int32_t* act, bool force) {
    if (k == 0) return;
    OutBatch* b = nullptr;
    {
      std::lock_guard<std::mutex> lk(b_mu_);
      inflight_ += k;
      if (sync_) {  // rows come back in send order (state_buffer.h:94-97)
        b = NewBatch(k);
        out_queue_.push_back(b);
      }
      // async: rows of successive batch_size-row batches are claimed as envs finish
    }
    // publish k tickets on the ring (single producer: send_mu_), then wake sleepers
    std::lock_guard<std::mutex> sl(send_mu_);
    if (b) b->t_send = Now();
    uint64_t tail = tail_.load(std::memory_order_relaxed);
    for (int i = 0; i < k; ++i) {
      // the slot is free once the worker that owned its previous ticket (tail - R) has copied the
      // task out -- gated per SLOT: a count of copied tickets would let a later lap overwrite the
      // slot of a worker that has claimed its ticket but not read it yet
      while (slot_free_[tail % ring_.size()].load(std::memory_order_acquire) != tail) {
        std::this_thread::yield();  // ring full: more than 4 x num_envs steps outstanding
      }
      ring_[tail % ring_.size()] = Task{ids[i] - cfg_.env_id_offset, act ? act[i] : 0, force, b, i};
      ++tail;
    }
    tail_.store(tail, std::memory_order_release);
    WakeWorkers();
  }

  // sleeping workers park on a futex word (no mutex to re-acquire on wake-up: a condition
  // variable serialises a few hundred woken threads on its mutex, 17 ms per 1024-env step
  // with 256 workers instead of 3 ms)
  void WakeWorkers() {
    wake_seq_.fetch_add(1, std::memory_order_acq_rel);
    if (sleepers_.load(std::memory_order_acquire) > 0) {
      syscall(SYS_futex, reinterpret_cast<uint32_t*>(&wake_seq_), FUTEX_WAKE_PRIVATE, INT_MAX,
              nullptr, nullptr, 0);
    }
  }

  OutBatch* WaitFront() {
    std::unique_lock<std::mutex> lk(b_mu_);
    auto ready = [&] {
      return !out_queue_.empty() && out_queue_.front()->finished.load() >= out_queue_.front()->rows;
    };
    // blocks like the reference's Recv (async_envpool.h:169-181): a consumer thread may arrive before the
    // producer's send; engine key "recv_timeout_ms" as in Pool::WantRows (engine.hip)
    const int timeout = (int)cfg_.Get("recv_timeout_ms", -1);
    const bool can_finish = sync_ ? !out_queue_.empty() : inflight_ >= batch_size_;
    if (timeout < 0 || can_finish) {
      done_cv_.wait(lk, ready);
    } else if (timeout == 0 || !done_cv_.wait_for(lk, std::chrono::milliseconds(timeout), ready)) {
      throw std::runtime_error(sync_ ? "recv: nothing pending"
                                     : "recv: fewer than batch_size envs in flight");
    }
    return out_queue_.front();
  }
  int PopFront() {
    std::lock_guard<std::mutex> lk(b_mu_);
    OutBatch* b = out_queue_.front();
    out_queue_.pop_front();
    const int rows = b->rows;
    inflight_ -= rows;
    free_batches_.push_back(b);
    return rows;
  }

  // post-process on the GPU and copy everything into the caller's arrays (state key order)
  static double Now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  void Deliver(OutBatch* b, void* const* out) {
    const int k = b->rows;
    const double t_begin = Now();
    struct Acc {  // EPA_ATARI_PROFILE=1: phase times on stderr when the pool is destroyed
      AtariPool* p;
      OutBatch* b;
      double t0;
      ~Acc() {
        p->prof_emulate_ += b->t_done - b->t_send;
        p->prof_wait_ += t0 - b->t_done;
        p->prof_deliver_ += Now() - t0;
        ++p->prof_batches_;
      }
    } acc{this, b, t_begin};
    EPA_HIP(hipSetDevice(cfg_.device));
    enum { kEnvId = 0, kPlayers, kElapsed, kDone, kReward, kDiscount, kStepType, kTrunc, kObs,
           kLives, kInfoReward, kTerminated, kRamKey };
    // obs: frames up, kernel, observations down -- straight into the caller's (pinned) array
    std::vector<uint8_t> scratch;
    uint8_t* obs = static_cast<uint8_t*>(out[kObs]);
    if (!obs) {
      scratch.resize((size_t)k * keys_[kObs].row_bytes());
      obs = scratch.data();
    }
    if (epa_atari_post_push(post_, b->local_id, k, b->frames, b->flags, obs) != EPA_OK) {
      throw std::runtime_error(std::string("Atari post-process: ") + epa_last_error());
    }
    auto put = [&](int key, const void* src, size_t bytes) {
      if (out[key]) std::memcpy(out[key], src, bytes);
    };
    put(kEnvId, b->env_id, (size_t)k * 4);
    put(kPlayers, b->env_id, (size_t)k * 4);
    put(kElapsed, b->elapsed, (size_t)k * 4);
    put(kDone, b->done, k);
    put(kReward, b->reward, (size_t)k * 4);
    put(kDiscount, b->discount, (size_t)k * 4);
    put(kStepType, b->step_type, (size_t)k * 4);
    put(kTrunc, b->trunc, k);
    put(kLives, b->lives, (size_t)k * 4);
    put(kInfoReward, b->info_reward, (size_t)k * 4);
    put(kTerminated, b->terminated, (size_t)k * 4);
    put(kRamKey, b->ram, (size_t)k * kRam);
  }

  // ---- the env bodies -------------------------------------------------------------
  void Capture(Env& e, int which) {
    std::memcpy(e.frame[which].data(), plugin_.api->screen(e.emu), kRawSize);
  }

  // one EnvStep (env.h:184-196) of env `t.env`; returns the PushStack flags of the row
  struct RowOut {
    float reward, discount, info_reward;
    uint8_t flags;
  };
  RowOut RunEnv(Env& e, const Task& t) {
    const epa_emulator_api* A = plugin_.api;
    const int max_steps = cfg_.max_episode_steps;
    const bool reset = t.force_reset || e.done;  // async_envpool.h:127
    e.current_step = reset ? 0 : e.current_step + 1;  // PreProcess, env.h:204-214
    RowOut r{0.0f, 1.0f, 0.0f, 0};
    if (reset) {  // AtariEnv::Reset, atari_env.h:167-199
      int noop = e.dist_noop(e.gen) + 1 - (fire_reset_ ? 1 : 0);
      bool push_all = false;
      if (t.force_reset || !a_.episodic_life || A->game_over(e.emu) || e.elapsed_step >= max_steps) {
        A->reset_game(e.emu);
        e.elapsed_step = 0;
        push_all = true;
      }
      while ((noop--) != 0) {
        A->act(e.emu, 0);
        if (A->game_over(e.emu)) {
          A->reset_game(e.emu);
          push_all = true;
        }
      }
      if (fire_reset_) A->act(e.emu, 1);
      Capture(e, 0);
      r.flags = push_all ? 1 : 2;  // PushStack(push_all, false)
      e.done = false;
      e.lives = A->lives(e.emu);
      return r;  // WriteState(0.0, 1.0, 0.0)
    }
    // AtariEnv::Step, atari_env.h:201-250
    float reward = 0.0f;
    e.done = false;
    int skip_id = a_.frame_skip;
    const int pooled = std::min(a_.frame_skip, 2);
    bool captured0 = false;
    for (; skip_id > 0 && !e.done; --skip_id) {
      reward += (float)A->act(e.emu, action_set_[t.action]);
      e.done = A->game_over(e.emu) != 0;
      if (skip_id <= pooled) {
        Capture(e, pooled - skip_id);
        captured0 = captured0 || pooled - skip_id == 0;
      }
    }
    // PushStack(false, frame_skip > 1 && skip_id == 0): without the max-pool the reference
    // resizes maxpool_buf_[0] -- which is this step's first pooled screen if one was
    // captured, and otherwise still the image pushed last time
    const bool maxpool = a_.frame_skip > 1 && skip_id == 0;
    r.flags = maxpool ? 0 : (captured0 ? 2 : 4);
    ++e.elapsed_step;
    e.done = e.done || e.elapsed_step >= max_steps;
    const int lives_now = A->lives(e.emu);
    if (a_.episodic_life && 0 < lives_now && lives_now < e.lives) e.done = true;
    if (a_.zero_discount_on_life_loss) {
      r.discount = (float)(e.lives == lives_now && !e.done);
    } else {
      r.discount = 1.0f - (float)e.done;
    }
    r.info_reward = reward;
    if (a_.reward_clip) {
      if (reward > 0) {
        reward = 1;
      } else if (reward < 0) {
        reward = -1;
      }
    }
    r.reward = reward;
    e.lives = lives_now;
    return r;
  }

  void WorkerLoop() {
    const epa_emulator_api* A = plugin_.api;
    for (;;) {
      // lock-free ticket (the reference's ActionBufferQueue, action_buffer_queue.h:59-80, is the
      // same idea): a worker owns ticket i and waits until the producer has published it --
      // a short spin (a step's tickets arrive together), then a condition variable
      const uint64_t ticket = head_.fetch_add(1, std::memory_order_acq_rel);
      int spins = 0;
      while (tail_.load(std::memory_order_acquire) <= ticket) {
        if (stop_.load(std::memory_order_acquire)) return;
        if (++spins < 4000) {
          __builtin_ia32_pause();
          continue;
        }
        const uint32_t seq = wake_seq_.load(std::memory_order_acquire);
        sleepers_.fetch_add(1, std::memory_order_acq_rel);
        if (tail_.load(std::memory_order_acquire) <= ticket && !stop_.load(std::memory_order_acquire)) {
          syscall(SYS_futex, reinterpret_cast<uint32_t*>(&wake_seq_), FUTEX_WAIT_PRIVATE, seq,
                  nullptr, nullptr, 0);  // returns at once if wake_seq_ moved on
        }
        sleepers_.fetch_sub(1, std::memory_order_acq_rel);
      }
      const Task t = ring_[ticket % ring_.size()];
      // the slot may be reused by ticket + R
      slot_free_[ticket % ring_.size()].store(ticket + ring_.size(), std::memory_order_release);
      Env& e = envs_[t.env];
      const RowOut r = RunEnv(e, t);
      // claim the row (Allocate, state_buffer_queue.h:123-141) and write it (WriteState)
      OutBatch* b = t.batch;
      int row = t.row;
      if (!b) {  // async mode: first come, first served (state_buffer_queue.h:123-141)
        std::lock_guard<std::mutex> lk(b_mu_);
        if (!claim_ || claim_->claimed.load() >= claim_->rows) {
          claim_ = NewBatch(batch_size_);
          out_queue_.push_back(claim_);
        }
        b = claim_;
        row = b->claimed.fetch_add(1);
      }
      const int max_steps = cfg_.max_episode_steps;
      if (r.flags != 4) {
        std::memcpy(b->frames + (size_t)row * 2 * kRawSize, e.frame[0].data(), kRawSize);
        if (r.flags == 0) {
          std::memcpy(b->frames + (size_t)row * 2 * kRawSize + kRawSize, e.frame[1].data(), kRawSize);
        }
      }
      b->flags[row] = r.flags;
      b->local_id[row] = t.env;
      b->env_id[row] = t.env + cfg_.env_id_offset;
      b->done[row] = e.done ? 1 : 0;
      b->step_type[row] = e.current_step == 0 ? 0 : (e.done ? 2 : 1);  // Allocate, env.h:224-240
      b->reward[row] = r.reward;
      b->discount[row] = r.discount;
      b->trunc[row] = (e.done && e.elapsed_step >= max_steps) ? 1 : 0;  // atari_env.h:273
      b->elapsed[row] = e.elapsed_step;                                  // :283 (Issue #179)
      b->lives[row] = e.lives;
      b->info_reward[row] = r.info_reward;
      b->terminated[row] = A->game_over(e.emu);
      std::memcpy(b->ram + (size_t)row * kRam, A->ram(e.emu), kRam);
      if (b->finished.fetch_add(1) + 1 >= b->rows) {
        b->t_done = Now();
        std::lock_guard<std::mutex> lk(b_mu_);
        done_cv_.notify_all();
      }
    }
  }

  AtariCfg a_;
  Plugin plugin_;
  std::string rom_;
  bool sync_{true};
  int batch_size_{0};
  std::vector<Env> envs_;
  std::vector<int> action_set_;
  bool fire_reset_{false};
  epa_atari_post* post_{nullptr};
  // task ring: tickets [head_, tail_) are published and unclaimed
  std::vector<Task> ring_;
  std::unique_ptr<std::atomic<uint64_t>[]> slot_free_;  // per slot: the ticket that may write it next
  std::atomic<uint64_t> head_{0}, tail_{0};
  std::atomic<int> sleepers_{0};
  std::atomic<uint32_t> wake_seq_{0};
  std::atomic<bool> stop_{false};
  std::mutex send_mu_;
  std::vector<std::thread> workers_;
  // output batches
  std::mutex b_mu_;
  std::condition_variable done_cv_;
  std::deque<OutBatch*> out_queue_;    // creation order; front = next recv
  OutBatch* claim_{nullptr};           // async mode: the batch whose rows are being claimed
  std::vector<OutBatch*> free_batches_;
  std::vector<std::unique_ptr<OutBatch>> all_batches_;
  long long inflight_{0};              // rows sent / reset and not yet received
  double prof_emulate_{0}, prof_wait_{0}, prof_deliver_{0};
  long prof_batches_{0};
};

}  // namespace

Pool* MakeAtari(const Config& cfg, const std::string& rom_path, const std::string& emulator_lib) {
  return new AtariPool(cfg, rom_path, emulator_lib);
}

// AtariEnvFns::ActionSpec (atari_env.h:76-90): the size of the action set needs the ROM
int AtariNumActions(const Config& cfg, const std::string& rom_path, const std::string& emulator_lib) {
  Plugin plugin(emulator_lib);
  const AtariCfg a = AtariCfg::From(cfg);
  epa_emulator_config ec = EmuCfg(rom_path, 0, a);
  void* h = plugin.api->create(&ec);
  if (!h) throw std::invalid_argument(std::string("Atari: emulator create failed: ") + plugin.api->last_error());
  int32_t codes[64];
  const int n = plugin.api->action_set(h, a.full_action_space ? 1 : 0, codes, 64);
  plugin.api->destroy(h);
  return n;
}

}  // namespace epa
