// K4 — Atari observation post-process: max-pool of the last two ALE frames,
// INTER_AREA (default) or INTER_LINEAR resize to img_height x img_width, push into the
// frame stack.
//
// Replaces, for a whole batch of envs in one launch:
//   AtariEnv::PushStack   envpool/atari/atari_env.h:308-346
//   Resize                envpool/utils/image_process.h:27-36 (cv::resize INTER_AREA)
//   the obs part of AtariEnv::WriteState   atari_env.h:283-287
// ALE emulation itself stays on the host (north star): the caller hands over
// the two grayscale frames `maxpool_buf_[0/1]` of every env (210x160 u8 each).
//
// HBM-streaming kernel, one 256-thread block per env:
//   in : 2 x 33,600 B frames (coalesced 16-B loads), 3 x 7,056 B old stack frames
//   out: 7,056 B ring slot + 4 x 7,056 B observation            (~123.6 KB/env)
// The pooled frame and both area tables are staged in LDS (33.6 + 4.7 KB/block =>
// 4 blocks/CU); every thread then produces 4 horizontally adjacent destination
// pixels from <= 4x3 LDS taps each and stores them as one 32-bit word.  (A first
// version read the tap tables from global memory inside the inner loop: ~340
// dependent cache hits per thread made the kernel latency-bound at 20 % of HBM.)
// Arithmetic follows OpenCV's generic area resize bit for bit (float taps from
// computeResizeAreaTab, horizontal then vertical accumulation in table order,
// cvRound saturate) — compiled with -ffp-contract=off, so the result equals
// oracle/atari/atari_post.c exactly.  The frame stack is a per-env ring in HBM
// (`head` = oldest slot): a push overwrites one slot instead of shifting three.
// use_inter_area_resize=false (the reference's benchmark setting, benchmark/test_envpool.py:92)
// selects cv::INTER_LINEAR: OpenCV's 8-bit fixed-point path, pure integer arithmetic -- 2x2
// taps with short coefficients scaled by 2^11 and the vertical pass
// ((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2 (kernel instantiation <2, 2, true>;
// the tables carry the coefficients as exactly representable floats).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "engine.h"

namespace epa {
namespace {

constexpr int kMaxTap = 6;
constexpr int kPostBlock = 256;

struct AreaTab {  // per destination index: first source index, #taps, weights
  short* ofs;
  short* cnt;
  float* alpha;  // [dsize][kMaxTap]
};

struct TabEntry {  // one destination index of an area table, as staged in LDS
  short ofs, cnt;
  float alpha[kMaxTap];
};

struct PostDev {
  unsigned char* ring;  // [N][S][C][dh*dw]
  int* head;            // [N] oldest slot
  AreaTab xt, yt;
  int n, s, sh, sw, dh, dw;
  int chan;                   // planes per frame: 1 (gray_scale) or 3 (RGB, atari_env.h:320-335)
  const unsigned char* lut;   // [chan][256] colour palette, or nullptr: frames are pixel values
};

// per-byte unsigned max of two packed words (SWAR; no packed u8 max on CDNA)
__device__ __forceinline__ unsigned int MaxU8x4(unsigned int a, unsigned int b) {
  // bytes where a >= b: the carry-out of (a | 0x80) - (b & 0x7f) corrected by the top bits
  const unsigned int H = 0x80808080u;
  const unsigned int d = ((a | H) - (b & ~H));          // bit 7 of each byte: (a&0x7f) >= (b&0x7f)
  const unsigned int ge = (((a & ~b) | (~(a ^ b) & d)) & H) >> 7;  // 1 per byte where a >= b
  const unsigned int m = ge * 255u;                      // 0xff per such byte
  return (a & m) | (b & ~m);
}

// XT / YT: compile-time upper bounds on the taps per destination column / row
// (3 / 4 for 210x160 -> 84x84); taps beyond a row's count carry weight 0, which
// adds exactly +0.0f, so the result is bit-identical to OpenCV's variable loops
// while every LDS read of a pixel can be issued before the first one is used.
// kChan / d.lut: the frames may be ALE palette INDICES (what the emulator's screen holds):
// the colour palette -- applyPaletteGrayscale / applyPaletteRGB of atari_env.h:189-194,
// 213-219 -- is then applied here, per frame BEFORE the max (the max of two grey values is
// not the grey value of the larger index), and for gray_scale=False the three colour planes
// are produced one after the other from the same 1-byte-per-pixel upload (a third of the
// PCIe bytes of shipping RGB) and stored already transposed to [3, h, w] (atari_env.h:320-335).
template <int XT, int YT, bool kLinear = false, int kChan = 1>
__global__ __launch_bounds__(kPostBlock) void AtariPostKernel(
    PostDev d, const int* __restrict__ env_id, int k,
    const unsigned char* __restrict__ frames,
    const unsigned char* __restrict__ reset_mask, unsigned char* __restrict__ obs) {
  extern __shared__ __align__(16) unsigned char pooled[];
  const int row = blockIdx.x;
  if (row >= k) return;
  const int e = env_id ? env_id[row] : row;
  const int fsz = d.sh * d.sw, osz = d.dh * d.dw;
  // per-row flags (include/envpool_amd.h): 1 reset (single frame, shown in every stack slot),
  // 2 single frame without replication, 4 repeat the newest stacked frame (no new screen)
  const unsigned int flags = reset_mask != nullptr ? reset_mask[row] : 0u;
  const bool push_all = (flags & 1u) != 0, rst = (flags & 3u) != 0, repeat = (flags & 4u) != 0;
  // 0. area tables (and the palette) -> LDS, behind the pooled frame
  TabEntry* xtab = reinterpret_cast<TabEntry*>(pooled + (fsz + 15) / 16 * 16);
  TabEntry* ytab = xtab + d.dw;
  unsigned char* lut = reinterpret_cast<unsigned char*>(ytab + d.dh);  // [kChan][256]
  const bool indexed = d.lut != nullptr;
  for (int i = threadIdx.x; i < d.dw + d.dh; i += kPostBlock) {
    const bool isx = i < d.dw;
    const int j = isx ? i : i - d.dw;
    const AreaTab& t = isx ? d.xt : d.yt;
    TabEntry en;
    en.ofs = t.ofs[j];
    en.cnt = t.cnt[j];
#pragma unroll
    for (int a = 0; a < kMaxTap; ++a) en.alpha[a] = t.alpha[j * kMaxTap + a];
    (isx ? xtab : ytab)[j] = en;
  }
  if (indexed) {
    for (int i = threadIdx.x; i < kChan * 256; i += kPostBlock) lut[i] = d.lut[i];
  }
  const unsigned char* f0 = frames + (size_t)row * 2 * fsz;
  const unsigned char* f1 = f0 + fsz;
  const int head = d.head[e];
  const size_t fr = (size_t)kChan * osz;  // bytes of one stacked frame (kChan planes)
  unsigned char* ring_e = d.ring + (size_t)e * d.s * fr;
  unsigned char* obs_e = obs + (size_t)row * d.s * fr;
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < kChan; ++c) {
    if (repeat) {
      // AtariEnv::Step left the frame_skip loop before a screen was captured (game over in
      // the first sub-frames, atari_env.h:208-221): PushStack resizes the unchanged
      // maxpool_buf_[0] again, i.e. the newest stacked frame is pushed once more
      const unsigned char* prev = ring_e + (size_t)((head + d.s - 1) % d.s) * fr + (size_t)c * osz;
      unsigned char* slot = ring_e + (size_t)head * fr + (size_t)c * osz;
      unsigned char* newest = obs_e + (size_t)(d.s - 1) * fr + (size_t)c * osz;
      for (int i = threadIdx.x; i < osz; i += kPostBlock) {
        const unsigned char v = prev[i];
        newest[i] = v;
        if (d.s > 1) slot[i] = v;
      }
      continue;
    }
    // 1. max-pool the two frames into LDS (atari_env.h:310-315); on reset there
    //    is only one observation (maxpool = false)
    const unsigned char* lc = lut + c * 256;
    auto look4 = [&](unsigned int w) -> unsigned int {  // palette lookup of 4 packed indices
      return (unsigned int)lc[w & 255u] | ((unsigned int)lc[(w >> 8) & 255u] << 8) |
             ((unsigned int)lc[(w >> 16) & 255u] << 16) | ((unsigned int)lc[w >> 24] << 24);
    };
    const int nvec = fsz / 16;
    constexpr int kBatch = 6;  // independent 16-B loads in flight per thread and frame
    for (int base = 0; base < nvec; base += kBatch * kPostBlock) {
      uint4 a[kBatch], b[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int i = base + u * kPostBlock + threadIdx.x;
        if (i < nvec) {
          a[u] = reinterpret_cast<const uint4*>(f0)[i];
          if (!rst) b[u] = reinterpret_cast<const uint4*>(f1)[i];
        }
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int i = base + u * kPostBlock + threadIdx.x;
        if (i < nvec) {
          uint4 r = a[u];
          if (indexed) {
            r.x = look4(r.x);
            r.y = look4(r.y);
            r.z = look4(r.z);
            r.w = look4(r.w);
          }
          if (!rst) {
            uint4 q = b[u];
            if (indexed) {
              q.x = look4(q.x);
              q.y = look4(q.y);
              q.z = look4(q.z);
              q.w = look4(q.w);
            }
            r.x = MaxU8x4(r.x, q.x);
            r.y = MaxU8x4(r.y, q.y);
            r.z = MaxU8x4(r.z, q.z);
            r.w = MaxU8x4(r.w, q.w);
          }
          reinterpret_cast<uint4*>(pooled)[i] = r;
        }
      }
    }
    for (int i = nvec * 16 + threadIdx.x; i < fsz; i += kPostBlock) {
      unsigned char a = f0[i], b = f1[i];
      if (indexed) {
        a = lc[a];
        b = lc[b];
      }
      pooled[i] = rst ? a : (a > b ? a : b);
    }
    __syncthreads();
    // 2. area resize of this plane
    unsigned char* slot = ring_e + (size_t)head * fr + (size_t)c * osz;
    unsigned char* newest = obs_e + (size_t)(d.s - 1) * fr + (size_t)c * osz;
    // OpenCV's order: per source row the horizontal sum in tap order, then the
    // vertical accumulation in tap order, then cvRound + saturate
    auto pixel = [&](const TabEntry& ty, int dx) -> unsigned int {
      const TabEntry tx = xtab[dx];
      if constexpr (kLinear) {  // cv::INTER_LINEAR, 8UC1 fixed point
        const int sy0 = ty.ofs * d.sw, sy1 = min(ty.ofs + 1, d.sh - 1) * d.sw;
        const int sx0 = tx.ofs, sx1 = min(tx.ofs + 1, d.sw - 1);
        const int a0 = (int)tx.alpha[0], a1 = (int)tx.alpha[1];
        const int b0 = (int)ty.alpha[0], b1 = (int)ty.alpha[1];
        const int S0 = pooled[sy0 + sx0] * a0 + pooled[sy0 + sx1] * a1;
        const int S1 = pooled[sy1 + sx0] * a0 + pooled[sy1 + sx1] * a1;
        return (unsigned int)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
      }
      unsigned char px[YT][XT];
#pragma unroll
      for (int yi = 0; yi < YT; ++yi) {  // clamped: padded taps have weight 0
        const int sy = min(ty.ofs + yi, d.sh - 1) * d.sw;
#pragma unroll
        for (int xi = 0; xi < XT; ++xi) px[yi][xi] = pooled[sy + min(tx.ofs + xi, d.sw - 1)];
      }
      float sum = 0.0f;
#pragma unroll
      for (int yi = 0; yi < YT; ++yi) {
        float buf = 0.0f;
#pragma unroll
        for (int xi = 0; xi < XT; ++xi) buf += (float)px[yi][xi] * tx.alpha[xi];
        float t = ty.alpha[yi] * buf;
        sum = yi == 0 ? t : sum + t;
      }
      int r = __float2int_rn(sum);  // cvRound
      return (unsigned int)(r < 0 ? 0 : (r > 255 ? 255 : r));
    };
    if ((d.dw & 3) == 0) {  // 4 pixels of one row per thread, one 32-bit store
      const int qw = d.dw >> 2, nq = osz >> 2;
      for (int q = threadIdx.x; q < nq; q += kPostBlock) {
        const int dy = q / qw, dx = (q - dy * qw) << 2;
        const TabEntry ty = ytab[dy];
        unsigned int w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) w |= pixel(ty, dx + j) << (8 * j);
        const int p = q << 2;
        *reinterpret_cast<unsigned int*>(newest + p) = w;
        if (push_all) {  // every stack slot shows the new frame (atari_env.h:338-345)
          for (int s = 0; s < d.s; ++s) {
            *reinterpret_cast<unsigned int*>(ring_e + (size_t)s * fr + (size_t)c * osz + p) = w;
          }
          for (int s = 0; s < d.s - 1; ++s) {
            *reinterpret_cast<unsigned int*>(obs_e + (size_t)s * fr + (size_t)c * osz + p) = w;
          }
        } else {
          *reinterpret_cast<unsigned int*>(slot + p) = w;
        }
      }
    } else {
      for (int p = threadIdx.x; p < osz; p += kPostBlock) {
        const int dy = p / d.dw, dx = p - dy * d.dw;
        const unsigned char v = (unsigned char)pixel(ytab[dy], dx);
        newest[p] = v;
        if (push_all) {
          for (int s = 0; s < d.s; ++s) ring_e[(size_t)s * fr + (size_t)c * osz + p] = v;
          for (int s = 0; s < d.s - 1; ++s) obs_e[(size_t)s * fr + (size_t)c * osz + p] = v;
        } else {
          slot[p] = v;
        }
      }
    }
    if (kChan > 1) __syncthreads();  // the next plane reuses the pooled frame
  }
  // 3. older frames: obs[j] = ring[(head + 1 + j) % S], j = 0..S-2
  if (!push_all) {
    const size_t ovec = fr / 16;  // 84*84 = 441 * 16
    for (int j = 0; j < d.s - 1; ++j) {
      const unsigned char* src = ring_e + (size_t)((head + 1 + j) % d.s) * fr;
      unsigned char* dst = obs_e + (size_t)j * fr;
      if ((fr & 15) == 0) {
        for (size_t i = threadIdx.x; i < ovec; i += kPostBlock) {
          reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        }
      } else {
        for (size_t i = threadIdx.x; i < fr; i += kPostBlock) dst[i] = src[i];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) d.head[e] = push_all ? 0 : (head + 1) % d.s;
}

// cv::computeResizeAreaTab restated (opencv imgproc/src/resize.cpp)
bool BuildAreaTab(int ssize, int dsize, std::vector<short>* ofs,
                  std::vector<short>* cnt, std::vector<float>* alpha) {
  double scale = (double)ssize / dsize;
  ofs->assign(dsize, 0);
  cnt->assign(dsize, 0);
  alpha->assign((size_t)dsize * kMaxTap, 0.0f);
  for (int dx = 0; dx < dsize; dx++) {
    double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    double cell = std::min(scale, ssize - fsx1);
    int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    int n = 0, first = -1;
    auto push = [&](int si, float a) {
      if (n == 0) first = si;
      if (n < kMaxTap) (*alpha)[(size_t)dx * kMaxTap + n] = a;
      ++n;
    };
    if (sx1 - fsx1 > 1e-3) push(sx1 - 1, (float)((sx1 - fsx1) / cell));
    for (int sx = sx1; sx < sx2; sx++) push(sx, (float)(1.0 / cell));
    if (fsx2 - sx2 > 1e-3) {
      push(sx2, (float)(std::min(std::min(fsx2 - sx2, 1.0), cell) / cell));
    }
    if (n > kMaxTap || n == 0) return false;
    (*ofs)[dx] = (short)first;
    (*cnt)[dx] = (short)n;
  }
  return true;
}

// one axis of cv::resize's INTER_LINEAR tables (imgproc/src/resize.cpp): source index and the two
// weights saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE), c computed in float
void BuildLinearTab(int ssize, int dsize, std::vector<short>* ofs, std::vector<short>* cnt,
                    std::vector<float>* alpha) {
  const double scale = (double)ssize / dsize;
  ofs->assign(dsize, 0);
  cnt->assign(dsize, 2);
  alpha->assign((size_t)dsize * kMaxTap, 0.0f);
  for (int dx = 0; dx < dsize; dx++) {
    float fx = (float)((dx + 0.5) * scale - 0.5);
    int sx = (int)std::floor(fx);
    fx -= (float)sx;
    if (sx < 0) {
      fx = 0;
      sx = 0;
    }
    if (sx >= ssize - 1) {
      fx = 0;
      sx = ssize - 1;
    }
    (*ofs)[dx] = (short)sx;
    (*alpha)[(size_t)dx * kMaxTap] = (float)std::lrint((1.f - fx) * 2048.f);
    (*alpha)[(size_t)dx * kMaxTap + 1] = (float)std::lrint(fx * 2048.f);
  }
}

}  // namespace
}  // namespace epa

struct epa_atari_post {
  epa::PostDev d{};
  int device{0};
  hipStream_t stream{nullptr};
  std::mutex mu;
  // staging for the host path
  unsigned char* d_frames{nullptr};
  unsigned char* d_obs{nullptr};
  unsigned char* d_mask{nullptr};
  int* d_ids{nullptr};
  int cap{0};
  int max_xtap{epa::kMaxTap}, max_ytap{epa::kMaxTap};
  bool linear{false};  // use_inter_area_resize = false
  unsigned char* d_lut{nullptr};  // [chan][256] colour palette (frames are palette indices)
  // host path pipeline: frames go up, kernels run and observations come down on
  // three streams, in chunks linked by events, so the two PCIe directions and the
  // kernel overlap
  static constexpr int kChunks = 4;
  hipStream_t h2d{nullptr}, d2h{nullptr};
  hipEvent_t up[kChunks]{}, done[kChunks]{};
};

namespace {
template <typename F>
int PostGuard(F&& f) {
  try {
    f();
    return EPA_OK;
  } catch (const std::invalid_argument& e) {
    epa::SetLastError(e.what());
    return EPA_ERR_INVALID;
  } catch (const epa::DeviceError& e) {
    epa::SetLastError(e.what());
    return EPA_ERR_DEVICE;
  } catch (const std::exception& e) {
    epa::SetLastError(e.what());
    return EPA_ERR_RUNTIME;
  }
}

void UploadTab(const std::vector<short>& ofs, const std::vector<short>& cnt,
               const std::vector<float>& alpha, epa::AreaTab* t) {
  EPA_HIP(hipMalloc(&t->ofs, ofs.size() * sizeof(short)));
  EPA_HIP(hipMalloc(&t->cnt, cnt.size() * sizeof(short)));
  EPA_HIP(hipMalloc(&t->alpha, alpha.size() * sizeof(float)));
  EPA_HIP(hipMemcpy(t->ofs, ofs.data(), ofs.size() * sizeof(short), hipMemcpyHostToDevice));
  EPA_HIP(hipMemcpy(t->cnt, cnt.data(), cnt.size() * sizeof(short), hipMemcpyHostToDevice));
  EPA_HIP(hipMemcpy(t->alpha, alpha.data(), alpha.size() * sizeof(float), hipMemcpyHostToDevice));
}

void LaunchPost(epa_atari_post* p, const int* d_ids, int k,
                const unsigned char* d_frames, const unsigned char* d_mask,
                unsigned char* d_obs) {
  size_t lds = (size_t)p->d.sh * p->d.sw;
  lds = (lds + 15) / 16 * 16 + sizeof(epa::TabEntry) * (size_t)(p->d.dw + p->d.dh) +
        (size_t)p->d.chan * 256;
#define EPA_POST_LAUNCH(XT, YT, LIN, CH)                                                     \
  hipLaunchKernelGGL((epa::AtariPostKernel<XT, YT, LIN, CH>), dim3(k), dim3(epa::kPostBlock), \
                     lds, p->stream, p->d, d_ids, k, d_frames, d_mask, d_obs)
  const bool rgb = p->d.chan == 3;
  if (p->linear) {
    if (rgb) EPA_POST_LAUNCH(2, 2, true, 3); else EPA_POST_LAUNCH(2, 2, true, 1);
  } else if (p->max_xtap <= 3 && p->max_ytap <= 4) {  // the Atari default 210x160 -> 84x84
    if (rgb) EPA_POST_LAUNCH(3, 4, false, 3); else EPA_POST_LAUNCH(3, 4, false, 1);
  } else {
    if (rgb) EPA_POST_LAUNCH(epa::kMaxTap, epa::kMaxTap, false, 3);
    else EPA_POST_LAUNCH(epa::kMaxTap, epa::kMaxTap, false, 1);
  }
#undef EPA_POST_LAUNCH
  EPA_HIP(hipGetLastError());
}
}  // namespace

extern "C" {

int epa_atari_post_create(int32_t num_envs, int32_t stack_num, int32_t in_h,
                          int32_t in_w, int32_t out_h, int32_t out_w,
                          int32_t use_inter_area, int32_t device,
                          epa_atari_post** out) {
  return epa_atari_post_create_ex(num_envs, stack_num, in_h, in_w, out_h, out_w, use_inter_area,
                                  1, nullptr, device, out);
}

int epa_atari_post_create_ex(int32_t num_envs, int32_t stack_num, int32_t in_h,
                             int32_t in_w, int32_t out_h, int32_t out_w,
                             int32_t use_inter_area, int32_t gray_scale,
                             const uint8_t* palette, int32_t device,
                             epa_atari_post** out) {
  return PostGuard([&] {
    if (!gray_scale && palette == nullptr) {
      throw std::invalid_argument(
          "atari_post: gray_scale = 0 needs the colour palette (frames are palette indices)");
    }
    if (num_envs < 1 || stack_num < 1 || out_h > in_h || out_w > in_w ||
        out_h < 1 || out_w < 1 || (size_t)in_h * in_w > 60000) {
      throw std::invalid_argument("atari_post: bad dimensions");
    }
    if (!use_inter_area && in_h == 2 * out_h && in_w == 2 * out_w) {
      throw std::invalid_argument(
          "atari_post: cv::INTER_LINEAR with an exact 2x2 reduction takes OpenCV's area-fast "
          "path, which is not restated");
    }
    if (use_inter_area && in_h % out_h == 0 && in_w % out_w == 0) {
      throw std::invalid_argument(
          "atari_post: integer scale factors take OpenCV's resizeAreaFast path, "
          "which is not restated");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
      throw epa::DeviceError("no HIP device available: envpool_amd has no CPU fallback");
    }
    if (device < 0 || device >= ndev) throw std::invalid_argument("device out of range");
    std::vector<short> xo, xc, yo, yc;
    std::vector<float> xa, ya;
    if (!use_inter_area) {
      epa::BuildLinearTab(in_w, out_w, &xo, &xc, &xa);
      epa::BuildLinearTab(in_h, out_h, &yo, &yc, &ya);
    } else if (!epa::BuildAreaTab(in_w, out_w, &xo, &xc, &xa) ||
               !epa::BuildAreaTab(in_h, out_h, &yo, &yc, &ya)) {
      throw std::invalid_argument("atari_post: scale factor too large");
    }
    auto* p = new epa_atari_post();
    p->linear = !use_inter_area;
    p->device = device;
    EPA_HIP(hipSetDevice(device));
    EPA_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    EPA_HIP(hipStreamCreateWithFlags(&p->h2d, hipStreamNonBlocking));
    EPA_HIP(hipStreamCreateWithFlags(&p->d2h, hipStreamNonBlocking));
    for (int c = 0; c < epa_atari_post::kChunks; ++c) {
      EPA_HIP(hipEventCreateWithFlags(&p->up[c], hipEventDisableTiming));
      EPA_HIP(hipEventCreateWithFlags(&p->done[c], hipEventDisableTiming));
    }
    p->d.n = num_envs;
    p->d.s = stack_num;
    p->d.sh = in_h;
    p->d.sw = in_w;
    p->d.dh = out_h;
    p->d.dw = out_w;
    p->d.chan = gray_scale ? 1 : 3;
    if (palette != nullptr) {
      EPA_HIP(hipMalloc(&p->d_lut, (size_t)p->d.chan * 256));
      EPA_HIP(hipMemcpy(p->d_lut, palette, (size_t)p->d.chan * 256, hipMemcpyHostToDevice));
      p->d.lut = p->d_lut;
    }
    size_t ring = (size_t)num_envs * stack_num * p->d.chan * out_h * out_w;
    EPA_HIP(hipMalloc(&p->d.ring, ring));
    EPA_HIP(hipMemset(p->d.ring, 0, ring));
    EPA_HIP(hipMalloc(&p->d.head, sizeof(int) * num_envs));
    EPA_HIP(hipMemset(p->d.head, 0, sizeof(int) * num_envs));
    UploadTab(xo, xc, xa, &p->d.xt);
    UploadTab(yo, yc, ya, &p->d.yt);
    p->max_xtap = *std::max_element(xc.begin(), xc.end());
    p->max_ytap = *std::max_element(yc.begin(), yc.end());
    *out = p;
  });
}

int epa_atari_post_destroy(epa_atari_post* p) {
  return PostGuard([&] {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipStreamSynchronize(p->stream);
    (void)hipFree(p->d.ring);
    (void)hipFree(p->d.head);
    for (epa::AreaTab* t : {&p->d.xt, &p->d.yt}) {
      (void)hipFree(t->ofs);
      (void)hipFree(t->cnt);
      (void)hipFree(t->alpha);
    }
    if (p->d_lut) (void)hipFree(p->d_lut);
    if (p->d_frames) (void)hipFree(p->d_frames);
    if (p->d_obs) (void)hipFree(p->d_obs);
    if (p->d_mask) (void)hipFree(p->d_mask);
    if (p->d_ids) (void)hipFree(p->d_ids);
    (void)hipStreamDestroy(p->stream);
    if (p->h2d) (void)hipStreamDestroy(p->h2d);
    if (p->d2h) (void)hipStreamDestroy(p->d2h);
    for (int c = 0; c < epa_atari_post::kChunks; ++c) {
      if (p->up[c]) (void)hipEventDestroy(p->up[c]);
      if (p->done[c]) (void)hipEventDestroy(p->done[c]);
    }
    delete p;
  });
}

int epa_atari_post_push(epa_atari_post* p, const int32_t* env_id, int32_t k,
                        const uint8_t* frames, const uint8_t* reset_mask,
                        uint8_t* obs_out) {
  return PostGuard([&] {
    if (k < 0 || k > p->d.n || !env_id || !frames || !obs_out) {
      throw std::invalid_argument("atari_post_push: bad arguments");
    }
    for (int i = 0; i < k; ++i) {
      if (env_id[i] < 0 || env_id[i] >= p->d.n) {
        throw std::invalid_argument("atari_post_push: env_id out of range");
      }
    }
    if (k == 0) return;
    std::lock_guard<std::mutex> lk(p->mu);
    EPA_HIP(hipSetDevice(p->device));
    size_t fsz = (size_t)2 * p->d.sh * p->d.sw;
    size_t osz = (size_t)p->d.s * p->d.chan * p->d.dh * p->d.dw;
    if (p->cap < p->d.n) {
      EPA_HIP(hipMalloc(&p->d_frames, fsz * p->d.n));
      EPA_HIP(hipMalloc(&p->d_obs, osz * p->d.n));
      EPA_HIP(hipMalloc(&p->d_mask, p->d.n));
      EPA_HIP(hipMalloc(&p->d_ids, sizeof(int) * p->d.n));
      p->cap = p->d.n;
    }
    // the kernel stream may still hold device-path work of the caller
    EPA_HIP(hipStreamSynchronize(p->stream));
    EPA_HIP(hipMemcpyAsync(p->d_ids, env_id, sizeof(int) * k, hipMemcpyHostToDevice, p->h2d));
    if (reset_mask) {
      EPA_HIP(hipMemcpyAsync(p->d_mask, reset_mask, k, hipMemcpyHostToDevice, p->h2d));
    }
    const int nchunk = k >= 64 ? epa_atari_post::kChunks : 1;
    for (int c = 0; c < nchunk; ++c) {
      const int r0 = (int)((long long)k * c / nchunk), r1 = (int)((long long)k * (c + 1) / nchunk);
      if (r1 == r0) continue;
      EPA_HIP(hipMemcpyAsync(p->d_frames + fsz * r0, frames + fsz * r0, fsz * (r1 - r0),
                             hipMemcpyHostToDevice, p->h2d));
      EPA_HIP(hipEventRecord(p->up[c], p->h2d));
      EPA_HIP(hipStreamWaitEvent(p->stream, p->up[c], 0));
      LaunchPost(p, p->d_ids + r0, r1 - r0, p->d_frames + fsz * r0,
                 reset_mask ? p->d_mask + r0 : nullptr, p->d_obs + osz * r0);
      EPA_HIP(hipEventRecord(p->done[c], p->stream));
      EPA_HIP(hipStreamWaitEvent(p->d2h, p->done[c], 0));
      EPA_HIP(hipMemcpyAsync(obs_out + osz * r0, p->d_obs + osz * r0, osz * (r1 - r0),
                             hipMemcpyDeviceToHost, p->d2h));
    }
    EPA_HIP(hipStreamSynchronize(p->d2h));
  });
}

int epa_atari_post_push_device(epa_atari_post* p, const int32_t* d_env_id,
                               int32_t k, const uint8_t* d_frames,
                               const uint8_t* d_reset_mask, uint8_t* d_obs_out) {
  return PostGuard([&] {
    if (k < 0 || k > p->d.n || !d_frames || !d_obs_out) {
      throw std::invalid_argument("atari_post_push_device: bad arguments");
    }
    if (k == 0) return;
    std::lock_guard<std::mutex> lk(p->mu);
    EPA_HIP(hipSetDevice(p->device));
    LaunchPost(p, d_env_id, k, d_frames, d_reset_mask, d_obs_out);
  });
}

void* epa_atari_post_stream(epa_atari_post* p) { return (void*)p->stream; }

}  // extern "C"
