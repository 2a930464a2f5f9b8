// Internal C++ side of the C ABI in include/envpool_amd.h.
//
// A `Pool` owns: device-resident SoA env state (family specific), a HIP stream
// for the step kernels plus one each for action uploads and result downloads
// (so that in async mode -- several batches in flight, async_envpool.h's whole
// point -- the copies of one batch overlap the kernel of the next, and recv of
// the oldest batch does not wait for work enqueued after it), pinned action
// staging, and a FIFO of result batches.  It replaces
// AsyncEnvPool + ActionBufferQueue + StateBufferQueue of the reference
// (envpool/core/async_envpool.h, action_buffer_queue.h, state_buffer_queue.h):
// "enqueue" = launch one batched step kernel on the stream, "state buffer" = a
// packed device block holding every state key for the k rows of that launch.
#ifndef ENVPOOL_AMD_CSRC_ENGINE_H_
#define ENVPOOL_AMD_CSRC_ENGINE_H_

#include <hip/hip_runtime.h>

#include <climits>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/envpool_amd.h"

namespace epa {

struct DeviceError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define EPA_HIP(expr)                                                        \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      throw ::epa::DeviceError(std::string(#expr) + ": " +                   \
                               hipGetErrorString(_e));                       \
    }                                                                        \
  } while (0)

inline int DtypeBytes(int dt) {
  switch (dt) {
    case EPA_I32: return 4;
    case EPA_F32: return 4;
    case EPA_F64: return 8;
    default: return 1;
  }
}

struct KeySpec {
  std::string name;
  int dtype;
  std::vector<int> shape;  // per row
  int row_elems() const {
    int n = 1;
    for (int d : shape) n *= d;
    return n;
  }
  int row_bytes() const { return row_elems() * DtypeBytes(dtype); }
};

// Parsed epa_config.
struct Config {
  int num_envs{1};
  int batch_size{0};
  int seed{42};
  std::vector<int> env_seed;
  int max_episode_steps{INT_MAX};
  int device{0};
  int env_id_offset{0};
  std::map<std::string, double> params;
  double Get(const std::string& k, double dflt) const {
    auto it = params.find(k);
    return it == params.end() ? dflt : it->second;
  }
  static Config From(const epa_config* c);
};

// Common state keys of every env (envpool/core/env_spec.h:37-43).
std::vector<KeySpec> CommonStateKeys();
constexpr int kNumCommonKeys = 8;
constexpr int kMaxKeys = 24;

// Output pointers handed to a step kernel: out.p[key] is the base of that
// key's [k, ...] array inside the batch block.
struct OutPtrs {
  void* p[kMaxKeys];
};

struct Batch {
  // where the batch's kernels write and recv reads: the block's own device allocation (`dev_buf`), or -- a DIRECT
  // step, Pool::SendInto -- the caller's pinned host block, where the results then already are when recv wants them
  char* dbuf{nullptr};
  char* dev_buf{nullptr};
  bool direct{false};
  size_t cap_rows{0};
  int k{0};
  int consumed{0};                // rows already handed to recv (async mode)
  std::vector<size_t> offsets;    // byte offset of each key's section
  hipEvent_t done{nullptr};
  // single-stream pools record `done` only when somebody needs it (Pool::EnsureDone): an event record behind
  // every launch keeps consecutive step kernels further apart than the launch path alone
  bool done_recorded{false};
  // A whole-pool host-path step of a sync pool may be cut into TWO launches (Pool::Send, "step_pipeline"): rows
  // [0, part_rows) are complete at `part_ev`, so their download overlaps the second launch
  int part_rows{0};               // 0: one launch
  hipEvent_t part_ev{nullptr};
  hipStream_t stream{nullptr};    // the compute stream its kernel was launched on ...
  int stream_idx{0};              // ... and always will be: blocks are recycled per stream
  // async mode with several compute streams: the local env of every row (host-path sends / resets;
  // empty for device-path batches, whose ids the host never sees), so that recv can mark them idle
  std::vector<int32_t> host_ids;
};

// Per-env bookkeeping shared by all families, SoA on device:
//   cur_step  Env::current_step_ (env.h:86)      init -1
//   done      XxxEnv::done_                       init 1 (first step resets)
//   mt/mti    std::mt19937 gen_ (env.h:78)        624 words per env + the position of the next word.
//             Layout: mt_shift = 0 the plain [624][N] structure of arrays, mt_shift = 4 tiles of 16 consecutive
//             words of one env, tile t of all envs = one [N][16] slab (device_common.hip.h: Mt19937::At)
struct CommonDev {
  int* cur_step;
  unsigned char* done;
  uint32_t* mt;
  int* mti;
  int n;
  int mt_shift;
};

// A few helper threads that copy one big host buffer into the pinned staging slot in pieces, together with the
// calling thread: a 3 MB action batch takes one thread 0.12 ms (25 GB/s), and in the sync step() that copy is in front
// of everything else (the reference hands its workers POINTERS into the caller's array, py_envpool.h:89-101; a
// device cannot read pageable memory).  Pieces are claimed with an atomic counter; the caller watches the in-order
// frontier of finished pieces and uploads behind it.  Workers poll for ~0.3 ms after a job (the next step's send
// is that close in a step loop) and sleep on a condition variable otherwise.
// CPUs of the NUMA node of `device` (empty if unknown), engine.hip
std::vector<int> DeviceLocalCpus(int device);

class HostCopier {
 public:
  static constexpr size_t kPiece = 256u << 10;
  // `cpus`: where the helpers may run (the device's NUMA node, DeviceLocalCpus; empty = anywhere)
  HostCopier(int threads, const std::vector<int>& cpus);
  ~HostCopier();
  HostCopier(const HostCopier&) = delete;
  HostCopier& operator=(const HostCopier&) = delete;
  void Start(char* dst, const char* src, size_t bytes);
  // The caller copies pieces too until bytes [0, upto) are in place; returns the bytes in place (a multiple of
  // kPiece, or the total), which may be more.
  size_t Advance(size_t upto);

 private:
  // One copy job; immutable but for the claim counter and the per-piece flags.  A helper that is late for a job finds
  // every piece claimed and leaves it alone: the next job never has to wait for stragglers.
  struct Job {
    char* dst;
    const char* src;
    size_t bytes, pieces;
    std::atomic<size_t> next{0};
    std::unique_ptr<std::atomic<uint8_t>[]> done;
  };
  void Work();
  static bool CopyOne(Job& j);
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<bool> stop_{false};
  std::shared_ptr<Job> job_;   // guarded by mu_
  std::shared_ptr<Job> mine_;  // the caller's handle on the current job
  size_t frontier_{0};         // pieces [0, frontier_) are complete (the caller's view)
};

class Pool {
 public:
  Pool(const Config& cfg, std::vector<KeySpec> env_state_keys, KeySpec action,
       bool needs_rng);
  virtual ~Pool();

  const Config& cfg() const { return cfg_; }
  const std::vector<KeySpec>& state_keys() const { return keys_; }
  const KeySpec& action_key() const { return action_; }
  hipStream_t stream() const { return stream_; }

  // The host-facing entry points are virtual: a family whose env bodies run on the HOST
  // (Atari: ALE stays on the CPU, north star) replaces the stream-ordered execution with
  // its own executor and keeps the C ABI (atari_env.hip).
  virtual void Send(const int32_t* env_id, int k, const void* action);
  // Send whose results may land straight in `block` (see epa_send_into); families with their own executor ignore the block
  virtual void SendInto(const int32_t* env_id, int k, const void* action, void* block, size_t block_bytes);
  virtual void Reset(const int32_t* env_ids, int k);
  virtual void SendDevice(const int32_t* d_env_id, int k, const void* d_action,
                          hipEvent_t wait_event = nullptr);
  // stream_ waits for everything enqueued so far on `producer` (device path)
  void WaitStream(hipStream_t producer);
  // `consumer` waits for the kernel of the batch RecvDevice handed out last
  void ConsumerWait(hipStream_t consumer);
  virtual int Recv(void* const* out_ptrs, int n_ptrs, int cap_rows);
  // zero-copy host recv into a caller-owned block (see epa_recv_block)
  size_t RecvLayout(int rows, size_t* offsets, int n_keys) const;
  virtual int RecvBlock(void* block, size_t block_bytes, size_t* offsets, int n_keys);
  virtual int RecvInto(void* const* out_ptrs, int n_ptrs, int cap_rows);
  virtual int RecvDevice(void** d_out_ptrs, int n_ptrs);
  virtual int PendingRows();
  virtual void Synchronize();
  void SetTiming(int mode);  // 0 off, 1 an event pair around every launch, 2 one pair around the whole window
  void KernelTime(double* avg_ms, int* launches);

  // May launches of this family run concurrently on several streams (async mode)?  True only if
  // a launch touches nothing but the per-env state of ITS rows and its own result block: no
  // per-launch scratch shared through the pool (the Humanoid kernels' HBM workspace and sort
  // buffer are indexed by the wave of the launch: they say no and keep one compute stream).
  virtual bool ConcurrentSafe() const { return false; }
  virtual int StateDim() const = 0;
  // family hooks: flat double state <-> device SoA, for the listed local ids
  virtual void GetState(const int* d_ids, int k, double* d_out) = 0;
  virtual void SetState(const int* d_ids, int k, const double* d_in) = 0;
  void GetStateHost(const int32_t* ids, int k, double* out);
  void SetStateHost(const int32_t* ids, int k, const double* in);

 protected:
  // Launch the family's batched step kernel for k rows on stream_.
  // d_ids == nullptr means rows 0..k-1 map to local envs 0..k-1.
  virtual void Launch(const int* d_ids, int k, const void* d_action,
                      bool force_reset, const OutPtrs& out) = 0;
  void InitCommon();  // allocates + initialises CommonDev (after derived ctor)
  // Generic TypedFrameStackBuffer (envpool/mujoco/frame_stack.h:74-146) for families
  // whose step kernel writes one un-stacked float64 observation per row: the first
  // env key must be "obs" declared with shape [S, nobs] (StackedObsShape).  The
  // kernel then writes into a scratch [k, nobs] block and a second small kernel
  // maintains a per-env ring and emits the stacked rows.  No-op for S == 1.
  void EnableObsStack();

  Config cfg_;
  std::vector<KeySpec> keys_;
  KeySpec action_;
  bool needs_rng_;
  // words of one env's generator kept contiguous (1 or 16; engine key "mt_tile" overrides): 1 for
  // families whose envs all draw at the same launches (the word a wave reads is one coalesced column), 16 for
  // families whose envs reset at their own times (a reset's draws then stay inside a few 64-byte sectors)
  int mt_tile_default_{1};
  // default of "step_pipeline" (see Pool::SendPipelined): 0 unless the family's constructor says otherwise -- it pays
  // where a step kernel takes about as long as its results need for the way down (the planar MuJoCo tasks: +18 %)
  // and costs where the kernel dominates (Ant: two half launches have two tails, -10 %)
  int pipeline_default_{0};
  // default of "direct_out" (Pool::SendInto): whether a whole-pool host-path step writes its results straight into
  // the block the caller named at send time (1), and also reads its action rows in place out of the pinned staging
  // slot instead of an uploaded copy (2: every family but the Ant, whose units would read them five times over)
  int direct_default_{2};
  // The stream the NEXT step kernel goes on.  Sync mode (batch_size == num_envs): always
  // compute_[0].  Async mode: successive batches rotate over the compute streams so that
  // independent in-flight batches run concurrently, like the reference's workers run every queued
  // slice in parallel (envpool/core/async_envpool.h:116-132); see PickStream.
  hipStream_t stream_{nullptr};
  hipStream_t h2d_stream_{nullptr};   // action uploads of the host path
  hipStream_t d2h_stream_{nullptr};   // result downloads of the host path
  // a pipelined step's big sections by DMA: first half on stream 2, second half on stream 3 (the small ones by a
  // gather kernel on d2h_stream_ / the kernel stream)
  hipStream_t d2h_stream2_{nullptr}, d2h_stream3_{nullptr};
  CommonDev common_{};

 private:
  // Rows the next Recv returns.  BLOCKS (mu_ released) until that many rows are pending -- AsyncEnvPool::Recv /
  // StateBufferQueue::Wait of the reference block on a semaphore (envpool/core/async_envpool.h:169-181,
  // state_buffer_queue.h:148-163), so a consumer thread may call recv before the producer's send.  Engine key
  // "recv_timeout_ms": < 0 (default) wait forever like the reference, 0 raise at once, > 0 raise after that long.
  int WantRows(std::unique_lock<std::mutex>& lk);
  void CopyRowsToHost(char* dst, const std::vector<size_t>& off, int want, std::unique_lock<std::mutex>& lk);
  Batch* AcquireBatch(int k);
  void ReleaseBatch(Batch* b);
  OutPtrs PtrsOf(const Batch& b) const;
  void Enqueue(const int* d_ids, int k, const void* d_action, bool force);
  Batch* BeginBatch(int k);
  // one launch of the family's step kernel for rows [row0, row0 + kp) of batch b
  void LaunchPart(Batch* b, const int* d_ids, int row0, int kp, const void* d_action, bool force);
  void FinishBatch(Batch* b);
  struct Staging;
  void SendPipelined(Staging& s, int k, const void* action, size_t id_bytes, size_t act_bytes,
                     char* direct_block);
  void SendImpl(const int32_t* env_id, int k, const void* action, char* block, size_t block_bytes);
  // a direct batch's rows [consumed, consumed + take) for recv: waits for its kernel, copies only if `dst` is not the
  // block the kernel wrote (CopyRowsToHost / RecvInto)
  void TakeDirect(Batch* b, int take, int got, char* dst, const size_t* dst_off, void* const* dst_ptrs,
                  std::unique_lock<std::mutex>& lk);
  int direct_out_{-1};  // "direct_out" (-1: not read yet)
  bool HostBlockVisible(const void* p);
  // chooses stream_ for the next launch and orders it behind what it may depend on
  void PickStream(const int32_t* host_ids, int k, bool device_path, const void* d_env_id = nullptr);
  void EnsureDone(Batch* b);  // records b->done on the batch's stream if nobody has yet
  void JoinCompute(hipStream_t into);  // `into` waits for everything enqueued on every compute stream
  void SyncCompute();                  // host waits for every compute stream
  void MarkIdle(Batch* b, int first, int count);
  struct Staging {
    char* h{nullptr};
    char* d{nullptr};
    size_t bytes{0};
    hipEvent_t free_ev{nullptr};   // the kernel that read this slot has finished
    hipEvent_t h2d_ev{nullptr};    // the upload into this slot has finished
    bool in_use{false};
  };
  Staging& NextStaging(size_t bytes);

 protected:
  void CheckIds(const int32_t* ids, int k) const;

 private:
  // generic observation frame stack (EnableObsStack)
  int stack_s_{1}, stack_nobs_{0};
  double* stack_ring_{nullptr};  // [N][S][nobs]
  int* stack_head_{nullptr};     // [N] slot holding the oldest frame
  double* stack_tmp_{nullptr};   // [N][nobs] un-stacked obs of the current launch
  std::mutex mu_;
  std::mutex recv_mu_;               // recv is single-consumer (state_buffer_queue.h:143-147): callers are serialised
  std::condition_variable pending_cv_;  // signalled by Enqueue; WantRows waits on it
  int recv_timeout_ms_{-1};
  int pipeline_rows_{-1};            // "step_pipeline": whole-pool host-path steps of at least this many rows (0: never)
  int* iota_dev_{nullptr};           // [num_envs] global env ids in order (the second half's id list)
  int zero_copy_small_{-1};          // "small_zero_copy" (-1: not read yet)
  bool ZeroCopySmall() {
    if (zero_copy_small_ < 0) zero_copy_small_ = cfg_.Get("small_zero_copy", 1) != 0 ? 1 : 0;
    return zero_copy_small_ == 1;
  }
  std::unique_ptr<HostCopier> copier_;  // "copy_threads" helpers (default 2, 0 = none) for pipelined steps
  std::deque<Batch*> pending_;
  std::vector<std::vector<Batch*>> free_;  // per compute stream
  std::vector<std::unique_ptr<Batch>> all_;
  Batch* lent_[2]{nullptr, nullptr};  // batches handed out by RecvDevice
  std::vector<Staging> staging_;
  size_t staging_next_{0};
  char* recv_stage_{nullptr};  // pinned D2H landing block
  size_t recv_stage_bytes_{0};
  hipEvent_t order_ev_{nullptr};  // WaitStream's producer marker
  // concurrent batches (async mode)
  std::vector<hipStream_t> compute_;     // compute_[0] is the sync-mode stream
  std::vector<hipEvent_t> join_ev_;      // one per compute stream (JoinCompute)
  size_t rr_{0};
  bool picked_{false};                   // WaitStream already chose the stream of the next launch
  std::vector<uint32_t> busy_;           // host path: rows of this env launched and not received yet (a count)
  std::vector<Batch*> frontier_;         // device path: batches handed out whose kernels may still run
  const int32_t* next_host_ids_{nullptr};  // ids of the launch being enqueued (for Batch::host_ids)
  bool next_identity_{false};
  // timing
  int timing_{0};
  hipEvent_t win0_{nullptr}, win1_{nullptr};  // timing mode 2: first launch .. KernelTime()
  bool win_open_{false};
  int win_launches_{0};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timers_;
  std::vector<hipEvent_t> timer_pool_;
};

// Diagnostic per-wave trace of a family's step kernel: when the environment variable
// `env` names a file, `d` is a device buffer of 6 x int64 per wave that the kernel fills
// (wall clock begin / end at 100 MHz, core clock begin / end, two family-specific words)
// and the buffer of the LAST launch is written to that file when the pool is destroyed
// (tools/ant_trace_stats.py, tools/planar_trace_stats.py).  Off (d == nullptr) otherwise.
struct WaveTrace {
  long long* d{nullptr};
  size_t waves{0};
  std::string file;
  void Init(const char* env, size_t n_waves, hipStream_t s);
  void DumpAndFree();
};

// obs key shape with the optional leading frame_stack dimension (StackSpec,
// envpool/mujoco/frame_stack.h:42-71); throws like the reference on frame_stack < 1
std::vector<int> StackedObsShape(const Config& cfg, int nobs);

// family factories (defined next to the kernels)
Pool* MakeClassicControl(const std::string& family, const Config& cfg);
bool DescribeClassicControl(const std::string& family, const Config& cfg,
                            std::vector<KeySpec>* state, KeySpec* action);
Pool* MakeToyText(const std::string& family, const Config& cfg);
bool DescribeToyText(const std::string& family, const Config& cfg,
                     std::vector<KeySpec>* state, KeySpec* action);
Pool* MakeMujoco(const std::string& family, const Config& cfg);
bool DescribeMujoco(const std::string& family, const Config& cfg,
                    std::vector<KeySpec>* state, KeySpec* action);
// Atari (atari_env.hip): needs two strings the numeric epa_config cannot carry
Pool* MakeAtari(const Config& cfg, const std::string& rom_path, const std::string& emulator_lib);
int AtariNumActions(const Config& cfg, const std::string& rom_path, const std::string& emulator_lib);

const std::vector<std::string>& FamilyNames();
void SetLastError(const std::string& msg);  // thread-local epa_last_error()

// Launch helpers shared by the family files.
void LaunchInitCommon(CommonDev c, int seed, const int* d_env_seed,
                      int id_offset, bool with_rng, hipStream_t s);

}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_ENGINE_H_
