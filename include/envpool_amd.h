/*
 * envpool_amd — C ABI of the MI355X-native batched-step engine.
 *
 * This is the drop-in boundary for ONE path of sail-sg/envpool: the batched
 * Send()/Recv()/Reset() execution path.  In the reference that path is the
 * C++ virtual class `EnvPool<Spec>` (envpool/core/envpool.h:29-56) implemented
 * by `AsyncEnvPool<Env>` (envpool/core/async_envpool.h:42-238: thread pool +
 * ActionBufferQueue + StateBufferQueue) and bound to Python by
 * `PyEnvPool<Pool>` (envpool/core/py_envpool.h:206-288).  Here the thread pool
 * and the queues are replaced by device-resident SoA env state and one batched
 * HIP kernel per env family; everything above (`PySend/PyRecv/PyReset`, the
 * Python adaptors) can stay and bind to the functions below
 * (see INTEGRATION.md for the pybind11 / ctypes stubs).
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure;
 *    `epa_last_error()` then holds a message (thread local).  Error classes
 *    follow the reference: EPA_ERR_INVALID  <-> std::invalid_argument
 *    (-> Python ValueError, envpool/core/env_spec.h:75-80), EPA_ERR_RUNTIME <->
 *    std::runtime_error (-> RuntimeError).
 *  - plain pointers and sizes only; no C++/torch types.
 *  - arrays are C-contiguous, row-major, one row per env in the batch.
 *  - key order is the reference's (envpool/core/env_spec.h:32-43):
 *      actions: "env_id", "players.env_id", <env action keys...>
 *      states : "info:env_id", "info:players.env_id", "elapsed_step", "done",
 *               "reward", "discount", "step_type", "trunc", <env state keys...>
 */
#ifndef ENVPOOL_AMD_H_
#define ENVPOOL_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPA_OK 0
#define EPA_ERR_INVALID 1 /* std::invalid_argument in the reference */
#define EPA_ERR_RUNTIME 2 /* std::runtime_error in the reference */
#define EPA_ERR_DEVICE 3  /* HIP failure (no reference analogue) */

/* element types of state / action arrays */
#define EPA_I32 0
#define EPA_F32 1
#define EPA_F64 2
#define EPA_BOOL 3 /* 1 byte, numpy bool_ */
#define EPA_U8 4

typedef struct epa_pool epa_pool;

/*
 * Pool configuration = the reference's `common_config`
 * (envpool/core/env_spec.h:26-31) restricted to what the step path reads, plus
 * per-family numeric options passed as (key, value) pairs named exactly like
 * the reference's `XxxEnvFns::DefaultConfig()` keys (e.g. "version" for
 * Pendulum, "size" for FrozenLake, "frame_skip", "ctrl_cost_weight", ...).
 * Unknown keys are ignored.  Engine extensions (not reference keys):
 *   "precision"   Ant: 1 fp64 (default), 0 fp32 arithmetic (meets 1e-5).  HalfCheetah / Walker2d / Hopper: only 1
 *                 (their fp32 mode was removed in round 4: outside 1e-5 and slower than the fp64 kernel)
 *   "xml_v5"      Walker2d / Pusher: 1 selects the *_v5 model (the reference's xml_file)
 *   "planar_spread" HalfCheetah / Walker2d / Hopper / Pusher: 1 (default) a batch of 16 .. 64 (Pusher: 32 .. 64) envs
 *                 per SIMD is spread over all SIMDs with 16 / 32 / 48 envs per wave; 0 always 64 envs per wave
 *   "hum_layout"  Humanoid / HumanoidStandup: 1 one env per lane quad (default), 0 one env per lane -- the superseded
 *                 kernel, built only into lib/libenvpool_amd_alt.so (`make -C envpool_amd/csrc EPA_ALT_KERNELS=1`, the
 *                 cross-check tests load it through ENVPOOL_AMD_LIB); the product library refuses 0
 *   "hum_sort"    quad layout: 1 cost-sorted waves (default), 0 rows in send order
 *   "hum_debug"   quad layout: stages switched off (bits 1 2 4 8), solver statistics (16) or cycles per stage
 *                 (32 64 128 256) routed into the info keys; accepted by the diagnostic build only
 *                 (-DEPA_HUM_DEBUG: tools/build_trace_lib.sh, tools/build_alt_hum4.sh), the product library refuses it
 *   "planar_layout" HalfCheetah / Walker2d, fp64: lanes per env of the step kernel -- 2 or 4 (one env per
 *                 lane group, mujoco_planar_lg.hip), 1 (one env per lane, mujoco_gym.hip), 0 (default)
 *                 chosen once per pool: 2 above 16384 rows, else 4 (rows = num_envs in sync mode,
 *                 min(num_envs, 4 x batch_size) in async mode: what is in flight on the compute streams).
 *                 The two layouts sum the contact rows in different orders: with the default, an env's
 *                 low-order bits therefore depend on the pool's num_envs / batch_size (never on the rows
 *                 of a particular send); set the key explicitly where pools of different shape must agree
 *                 bit for bit.  Each layout is within 1e-9 of the oracle per env-step.
 *                 Hopper: 0 (default) the lane-group kernel with a group of ONE lane (the lane's 6 dofs are
 *                 the robot), 1 the one-env-per-lane kernel on the 9-dof tree with a ghost leg; 2 / 4 refused.
 *   "planar_waves" lane-group kernel with 4 lanes per env: register budget for 1 (default) or 2 waves per SIMD
 *                 (A/B switch; with 1 or 2 lanes per env LDS allows one wave and the key has no effect)
 *   "planar_lpt"  lane-group kernel: 1 whole-pool launches serve the chunks of envs slowest first, by their
 *                 duration in the previous launch (default for Walker2d / Hopper); 0 index order (default for
 *                 HalfCheetah since round 5).  Never changes results.
 *   "step_pipeline" sync pools, host path (epa_send): a whole-pool send of at least this many rows runs as TWO launches
 *                 over the two halves of the rows, and epa_recv* downloads the first half while the second computes (the
 *                 rows still arrive as ONE batch in send order, every value bit-identical to the single launch).
 *                 Default 32768 for HalfCheetah / Walker2d pools with 2 lanes per env (where half the rows take half
 *                 the time: numpy API +20 %), 0 = off for every other family (measured slower there); 0 switches it off.
 *   "copy_threads" helper threads (default 2, 0 .. 8) that copy the action rows of a pipelined step into the pinned
 *                 staging slot together with the calling thread; they poll ~0.3 ms after a step, then sleep
 *   "direct_out"  what epa_send_into does for a whole-pool step of a sync pool: 0 = it is epa_send; 1 = the step
 *                 kernel writes its rows straight into the caller's block; 2 (default; the Ant: 1) = that, and the
 *                 action rows are read in place out of the pinned staging slot (no upload): numpy step of HalfCheetah
 *                 N = 65536 0.46 -> 0.43 ms, Hopper 0.48 -> 0.39, Pusher 0.87 -> 0.65.  Never changes results.
 *   "small_zero_copy" 1 (default): host-path batches of up to 64 KB go without DMA commands -- the step kernel reads ids
 *                 and action rows straight out of the pinned staging slot, epa_recv's landing block is filled by a copy
 *                 kernel on the kernel stream (CartPole num_envs = 64: send + recv 32.4 -> 29.4 us); 0 = DMA as for
 *                 bigger batches.  Never changes results.
 *   "numa_bind"   1 (default): those helper threads run on the CPUs of the device's NUMA node (where this runtime
 *                 allocates pinned memory); 0 leaves them to the scheduler.  The CALLING thread is never moved by the
 *                 library: envpool_amd.bind_host_to_device() (Python) does that for a process that wants it, as the
 *                 reference's benchmark/numa_test.sh does with numactl
 *   "ant_sub"     Ant: mj_steps per unit of the step kernel's work queue (default 1: an env-step of a 16-env chunk is
 *                 frame_skip units, the chunk's state goes through HBM between them; frame_skip = one unit per chunk,
 *                 the schedule of rounds 2-5).  Never changes results.
 *   "selftest"    MuJoCo families with a self-test table (HalfCheetah, Walker2d, Hopper, Ant, Humanoid,
 *                 HumanoidStandup): 0 skips the load-time self-test for this pool (see epa_create); the environment
 *                 variable EPA_SELFTEST=0 skips it for the process
 *   "classic_block", "classic_rows" classic_control: threads per block (64 / 128 / 256, default by family and size) and
 *                 rows per thread of the step kernel (A/B keys; never change results)
 *   "classic_early" classic_control: 1 = the step kernel reads state, action and generator position together with
 *                 `done`, in front of the reset branch (default: CartPole only, where every wave holds a reset row:
 *                 -6 % kernel time at num_envs = 65536); never changes results
 *   "recv_timeout_ms" every family: how long epa_recv* waits for rows that have not been sent yet (see epa_recv):
 *                 < 0 forever (default, the reference's behaviour), 0 not at all, > 0 milliseconds
 *   "compute_streams" async mode (batch_size < num_envs): successive batches run on this many
 *                 compute streams (default 4, 1 = one stream), like the reference's worker threads
 *                 step all queued slices in parallel (core/async_envpool.h:116-132).  Pools with the generic
 *                 frame stack (frame_stack > 1) and the one-env-per-lane Humanoid kernel (hum_layout = 0, one
 *                 shared workspace) keep one stream.  Device memory of the Humanoid quad pools' per-launch scratch:
 *                 compute_streams x W(batch_size) + W(num_envs), W(rows) = 1.9 MB per 16 rows (the second term only
 *                 once a send exceeds batch_size rows, e.g. the reset of all envs; such sends share one copy and are
 *                 ordered behind each other).  A device-path send whose env ids do not continue a handed-out batch
 *                 (identity ids, ids from elsewhere) is ordered behind EVERY batch still executing or pending.
 *                 The rule for the caller is the reference's: an env may be sent
 *                 again only after recv handed it out (host path: a violation is detected and that
 *                 launch is ordered behind everything enqueued; device path: the env ids of a send
 *                 should be the `info:env_id` array of a batch recv_device returned -- the launch then
 *                 continues that batch's stream -- any other pointer is ordered behind every batch
 *                 still executing).
 */
typedef struct epa_config {
  int32_t num_envs;          /* common_config "num_envs" */
  int32_t batch_size;        /* "batch_size"; 0 => num_envs (env_spec.h:81-83) */
  int32_t seed;              /* "seed": env i is seeded seed + i (env.h:109) */
  const int32_t* env_seed;   /* "env_seed": NULL or num_envs explicit seeds */
  int32_t max_episode_steps; /* "max_episode_steps"; <=0 => INT_MAX */
  int32_t device;            /* HIP device ordinal (extension) */
  int32_t env_id_offset;     /* global id of local env 0 when a pool is one
                                shard of a multi-GPU pool (extension) */
  int32_t n_params;
  const char* const* param_keys;
  const double* param_values;
} epa_config;

/* One state or action key. `shape` excludes the leading batch dimension
 * (the reference's -1 player dimension is dropped: all hot-path envs are
 * single-player). */
typedef struct epa_key_info {
  const char* name;
  int32_t dtype;
  int32_t ndim;
  int32_t shape[4];
  int32_t row_elems; /* product of shape (1 for scalars) */
  int32_t row_bytes;
} epa_key_info;

/* ---- spec queries: no GPU needed ------------------------------------- */

/* Number of env families compiled in and their names ("CartPole", ...). */
int epa_num_families(void);
const char* epa_family_name(int i);

/* Describe the state/action keys a family would produce for `cfg`
 * (replaces EnvSpec<Fns>::{state_spec,action_spec}, env_spec.h:48-85).
 * Writes up to `cap` entries; returns the total number through *n. */
int epa_describe_state(const char* family, const epa_config* cfg,
                       epa_key_info* keys, int cap, int* n);
int epa_describe_action(const char* family, const epa_config* cfg,
                        epa_key_info* keys, int cap, int* n);

/* ---- pool lifetime ---------------------------------------------------- */

/* Replaces AsyncEnvPool<Env>::AsyncEnvPool(spec) (async_envpool.h:90-149):
 * allocates SoA state for num_envs envs on `cfg->device`, seeds every env's
 * mt19937 (env.h:101-117) and marks every env done so that the first step is a
 * reset (cartpole.h:67 `done_{true}`, async_envpool.h:127).
 * The FIRST pool of a MuJoCo family in a process (per device) also runs the library's self-test: fixed states are
 * stepped once in every kernel variant of the family and compared with the same arithmetic evaluated on the host at
 * build time (envpool_amd/csrc/gen_selftest.cpp), and the launch is repeated and must be bit-identical; a library that
 * was not built the way envpool_amd/csrc/Makefile builds it is refused with EPA_ERR_DEVICE (~25 ms; EPA_SELFTEST=0 or
 * the engine key "selftest" = 0 skip it). */
int epa_create(const char* family, const epa_config* cfg, epa_pool** out);

/* Replaces ~AsyncEnvPool (async_envpool.h:151-162). */
int epa_destroy(epa_pool* pool);

/* ---- host path: the reference's Send / Recv / Reset -------------------- */

/* Replaces AsyncEnvPool::Send(vector<Array>) (async_envpool.h:59-82,163-167).
 * `env_id[k]` int32, `action` = k rows of the family's action key in its
 * reference dtype.  Copies both into pinned staging (the caller may free its
 * buffers on return), enqueues H2D + one batched step kernel on the pool's
 * stream and returns immediately.  Row i of the resulting batch belongs to
 * env_id[i] (sync-mode ordering, state_buffer.h:94-97). */
int epa_send(epa_pool* pool, const int32_t* env_id, int32_t k,
             const void* action);

/* Replaces AsyncEnvPool::Reset(env_ids) (async_envpool.h:224-237): forced
 * reset of the listed envs; the result arrives through epa_recv. */
int epa_reset(epa_pool* pool, const int32_t* env_ids, int32_t k);

/* Replaces AsyncEnvPool::Recv() (async_envpool.h:169-181).  Blocks until the
 * oldest pending rows are computed, copies them device->host into
 * `out_ptrs[key]` (one buffer per state key, each with room for `cap_rows`
 * rows) and returns the number of rows through *k_out.
 *   sync  mode (batch_size == num_envs): returns the whole oldest send/reset
 *         batch (k rows, possibly < num_envs for a partial env_id send).
 *   async mode (batch_size <  num_envs): returns exactly batch_size rows in
 *         completion (= submission) order, a legal schedule of
 *         state_buffer_queue.h:123-163.
 * BLOCKS, like the reference (StateBufferQueue::Wait sits on a semaphore, state_buffer_queue.h:148-163; the binding
 * releases the GIL around it, py_envpool.h:255-262): a consumer thread may call epa_recv BEFORE the producer thread's
 * epa_send / epa_reset -- it returns once enough rows have been enqueued and computed.  send / reset from other
 * threads are not held up by a waiting or downloading consumer.  recv itself is single-consumer
 * (state_buffer_queue.h:143-147): concurrent calls are serialised.  Extension key "recv_timeout_ms" (epa_config
 * params): < 0 (default) wait forever; 0 EPA_ERR_RUNTIME at once when fewer rows are pending than a batch holds (for
 * single-threaded callers that would otherwise hang); > 0 EPA_ERR_RUNTIME after that many milliseconds.
 * epa_recv_block / epa_recv_into / epa_recv_device / epa_step_device wait the same way. */
int epa_recv(epa_pool* pool, void* const* out_ptrs, int32_t n_ptrs,
             int32_t cap_rows, int32_t* k_out);

/* Zero-copy variant of epa_recv: the reference hands numpy arrays that OWN
 * their memory and are never overwritten by later steps (py_envpool.h:40-49,
 * state_buffer_queue.h:149-163: a fresh buffer per batch).  Here the caller
 * provides one host block per batch (pinned memory from epa_host_alloc gives the
 * full PCIe rate), the batch lands in it with ONE device->host copy and no host
 * memcpy, and the per-key arrays are views at `offsets[key]`.
 * epa_recv_layout: section offsets (256-B aligned) and total size of a block
 * holding `rows` rows of every state key.
 * epa_recv_block: same blocking / batching semantics as epa_recv; `offsets`
 * (n_keys entries) is filled for the *k_out rows actually returned. */
int epa_recv_layout(epa_pool* pool, int32_t rows, size_t* offsets, int32_t n_keys,
                    size_t* total_bytes);
int epa_recv_block(epa_pool* pool, void* block, size_t block_bytes,
                   size_t* offsets, int32_t n_keys, int32_t* k_out);

/* epa_send for a caller that already knows WHERE the results shall go (the reference's StateBufferQueue allocates the
 * batch's output buffers before the workers write them, state_buffer_queue.h:123-140): `block` is a pinned host block
 * (epa_host_alloc) with room for k rows laid out by epa_recv_layout(k), which stays the caller's but must live until
 * the epa_recv_block that returns this batch.  For a batch that recv will return as a whole -- every env of a sync pool,
 * or batch_size rows of an async pool -- the
 * step kernel then writes its rows STRAIGHT into the block -- they cross the link as the kernel's own stores, while
 * it runs -- and epa_recv_block with the same block only waits for the kernel (any other recv call copies out of the
 * block).  By default it also reads the action rows in place out of the pinned staging slot: no DMA command at all in
 * such a step.  A send whose block is too small or not pinned, or with the extension key "direct_out" = 0, behaves
 * exactly like epa_send and ignores the block; rows that a recv returns in pieces or together with other batches'
 * rows are copied out of the block.  Results are the same bytes either way. */
int epa_send_into(epa_pool* pool, const int32_t* env_id, int32_t k, const void* action, void* block,
                  size_t block_bytes);

/* Same blocking / batching semantics as epa_recv, but every state key is copied
 * device->host DIRECTLY into `out_ptrs[key]` (no landing block, no host memcpy;
 * pinned destinations get the full PCIe rate, NULL skips a key).  This is the
 * building block of the multi-GPU host gather: GPU g's rows land
 * in ITS row range of one host batch shared by all GPUs. */
int epa_recv_into(epa_pool* pool, void* const* out_ptrs, int32_t n_ptrs,
                  int32_t cap_rows, int32_t* k_out);

/* Rows currently computed-or-in-flight and not yet received. */
int epa_pending_rows(epa_pool* pool, int32_t* rows);

/* ---- device path (zero-copy; the analogue of envpool/core/xla.h:116-213
 *      without the host staging the reference does there) ---------------- */

/* Like epa_send but `d_env_id` (may be NULL = all envs in order) and
 * `d_action` (NULL = forced reset of the listed envs) are device pointers on
 * the pool's device.  The step kernel is enqueued on the pool's stream behind
 * `wait_event` (a hipEvent_t as void*, may be NULL) -- the event the producer
 * of the action buffer recorded on ITS stream after writing it.  This is the
 * stream-ordering half of the reference's XLA custom call
 * (envpool/core/xla.h:151-169), minus its host staging.  With NULL the caller
 * must have made the buffers visible some other way (epa_wait_stream below, or
 * a device synchronise). */
int epa_send_device(epa_pool* pool, const int32_t* d_env_id, int32_t k,
                    const void* d_action, void* wait_event);

/* Convenience for callers that have a stream rather than an event (e.g.
 * torch.cuda.current_stream().cuda_stream): everything enqueued on
 * `producer_stream` (hipStream_t as void*; NULL = the legacy default stream)
 * so far happens-before every kernel the pool enqueues from now on. */
int epa_wait_stream(epa_pool* pool, void* producer_stream);

/* Hands out device pointers (one per state key) to the oldest pending batch.
 * The pointers stay valid until the second next epa_recv_device call on this
 * pool (batches are double buffered).  Does not synchronise the host: work
 * enqueued on epa_stream() after this call is ordered after the step kernel. */
int epa_recv_device(epa_pool* pool, void** d_out_ptrs, int32_t n_ptrs,
                    int32_t* k_out);

/* epa_send_device followed by epa_recv_device in ONE call -- the device-path form of the reference's sync `step()`
 * (envpool/python/envpool.py:345-349: send, then recv).  At the sizes where a step kernel takes a few microseconds
 * (classic_control / toy_text at num_envs = 65536) the two calls of a binding are most of a step's time. */
int epa_step_device(epa_pool* pool, const int32_t* d_env_id, int32_t k, const void* d_action,
                    void* wait_event, void** d_out_ptrs, int32_t n_ptrs, int32_t* k_out);

/* The mirror of epa_wait_stream for the outputs: `consumer_stream` waits for the
 * step kernel of the batch the LAST epa_recv_device handed out (a consumer on
 * epa_stream() itself needs no call).  The consumer must be done with a batch's
 * buffers before the second next epa_recv_device, when they are recycled. */
int epa_consumer_wait(epa_pool* pool, void* consumer_stream);

/* hipStream_t of the pool, as void*. */
void* epa_stream(epa_pool* pool);
int epa_synchronize(epa_pool* pool);

/* Average duration in ms of the step kernels launched since the last call,
 * measured with HIP events on the pool's stream; *launches = how many.
 * epa_set_timing(pool, 1): an event pair around every launch (exact per-launch
 * durations; the events keep consecutive launches ~12 us apart).
 * epa_set_timing(pool, 2): one event before the first launch and one when
 * epa_kernel_time_ms is called: (elapsed / launches), inter-launch gaps included,
 * nothing inserted between the launches -- what bench.py uses for its timed region.
 * epa_set_timing(pool, 0): off. */
int epa_set_timing(epa_pool* pool, int32_t enabled);
int epa_kernel_time_ms(epa_pool* pool, double* avg_ms, int32_t* launches);

/* Test hooks: read / overwrite the persistent state of the listed envs as the
 * family's flat double vector (CartPole: x,x_dot,theta,theta_dot; HalfCheetah:
 * qpos[9],qvel[9],qacc_warmstart[9],time ...).  Mirrors the reference's
 * ENVPOOL_TEST-only `info:qpos0/qvel0` state sync
 * (envpool/mujoco/gym/half_cheetah.h:50-53,112-115). */
int epa_state_dim(epa_pool* pool, int32_t* dim);
int epa_get_state(epa_pool* pool, const int32_t* env_ids, int32_t k,
                  double* out);
int epa_set_state(epa_pool* pool, const int32_t* env_ids, int32_t k,
                  const double* in);

/* ---- Atari post-process (K4): max-pool of the last two ALE frames, resize
 *      to 84x84, push into the frame stack (replaces AtariEnv::PushStack,
 *      envpool/atari/atari_env.h:308-346 + envpool/utils/image_process.h:27-36).
 *      use_inter_area: the config key `use_inter_area_resize` (atari_env.h:61):
 *      1 = cv::INTER_AREA (default), 0 = cv::INTER_LINEAR (what the reference's
 *      benchmark/test_envpool.py:92 selects). */
typedef struct epa_atari_post epa_atari_post;
int epa_atari_post_create(int32_t num_envs, int32_t stack_num, int32_t in_h,
                          int32_t in_w, int32_t out_h, int32_t out_w,
                          int32_t use_inter_area, int32_t device,
                          epa_atari_post** out);
/* As above, plus the colour handling of atari_env.h:189-194 / 213-219 / 320-335:
 *   palette != NULL: the frames handed to push are ALE palette INDICES (the emulator's
 *     screen, 1 byte per pixel) and `palette` is the table behind applyPaletteGrayscale
 *     ([256], gray_scale = 1) or applyPaletteRGB ([3][256] planar, gray_scale = 0); it is
 *     applied on the device, per frame before the max-pool.
 *   gray_scale = 0 (needs the palette): a stacked frame is three planes [3, out_h, out_w]
 *     (the transpose of atari_env.h:320-335), observations are
 *     [k, stack_num * 3, out_h, out_w]. */
int epa_atari_post_create_ex(int32_t num_envs, int32_t stack_num, int32_t in_h,
                             int32_t in_w, int32_t out_h, int32_t out_w,
                             int32_t use_inter_area, int32_t gray_scale,
                             const uint8_t* palette, int32_t device,
                             epa_atari_post** out);
int epa_atari_post_destroy(epa_atari_post* p);
/* frames: [k, 2, in_h, in_w] u8 (the two max-pool buffers, host memory);
 * reset_mask[k] u8 (NULL = all 0), per-row flags of AtariEnv::PushStack (atari_env.h:308-346):
 *   0  max-pool the two frames, push                       (Step, frame_skip loop completed)
 *   1  frame 0 only, replicated into every stack slot      (Reset with push_all)
 *   2  frame 0 only, pushed like a step                    (Reset of an episodic-life env,
 *                                                           frame_skip = 1, early game over)
 *   4  no new screen: push the newest stacked frame again  (game over before a capture)
 * obs_out: [k, stack_num * (gray_scale ? 1 : 3), out_h, out_w] u8 host. */
int epa_atari_post_push(epa_atari_post* p, const int32_t* env_id, int32_t k,
                        const uint8_t* frames, const uint8_t* reset_mask,
                        uint8_t* obs_out);
/* device-resident variant: all pointers are device pointers. */
int epa_atari_post_push_device(epa_atari_post* p, const int32_t* d_env_id,
                               int32_t k, const uint8_t* d_frames,
                               const uint8_t* d_reset_mask, uint8_t* d_obs_out);
void* epa_atari_post_stream(epa_atari_post* p);

/* ---- Atari end to end --------------------------------------------------- *
 * AtariEnvPool = AsyncEnvPool<AtariEnv> (envpool/atari/atari_env.h:348) as an epa_pool: the
 * emulator runs on host worker threads behind the plugin table of
 * include/envpool_amd_emulator.h, palette + max-pool + resize + frame stack run as one HIP
 * kernel per batch.  The pool is driven with the generic epa_send / epa_recv / epa_recv_block
 * / epa_reset / epa_destroy (the device-resident entry points raise: the emulator is on the
 * host).  Numeric config keys of AtariEnvFns::DefaultConfig (atari_env.h:52-63) travel in
 * base.param_*: stack_num frame_skip noop_max zero_discount_on_life_loss episodic_life
 * reward_clip use_fire_reset img_height img_width mode difficulty full_action_space
 * repeat_action_probability use_inter_area_resize gray_scale, plus num_threads. */
typedef struct epa_atari_config {
  epa_config base;
  const char* rom_path;     /* GetRomPath(base_path, task), atari_env.h:43-48 */
  const char* emulator_lib; /* emulator plugin, see envpool_amd_emulator.h */
} epa_atari_config;
int epa_atari_create(const epa_atari_config* cfg, epa_pool** out);
/* size of the action set (AtariEnvFns::ActionSpec loads the ROM for it, atari_env.h:76-90) */
int epa_atari_num_actions(const epa_atari_config* cfg, int32_t* n);
/* state / action keys of an existing pool (same order and meaning as epa_describe_*) */
int epa_pool_state_keys(epa_pool* pool, epa_key_info* keys, int cap, int* n);
int epa_pool_action_keys(epa_pool* pool, epa_key_info* keys, int cap, int* n);

/* ---- misc -------------------------------------------------------------- */
const char* epa_last_error(void);
const char* epa_version(void);
int epa_device_count(int32_t* n);
/* pinned host memory for zero-copy numpy hand-off (py_envpool.h:40-49) */
void* epa_host_alloc(size_t bytes);
void epa_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* ENVPOOL_AMD_H_ */
