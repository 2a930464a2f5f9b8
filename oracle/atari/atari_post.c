/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement of the Atari observation post-process of the reference:
 *   AtariEnv::PushStack   envpool/atari/atari_env.h:308-346
 *     ptr[i] = max(maxpool_buf_[0][i], maxpool_buf_[1][i])        (:310-315)
 *     Resize(maxpool_buf_[0], &resize_img_, use_inter_area_resize_) (:316)
 *     pop the oldest frame, push the new one, optionally replicate (:317-345)
 *   Resize                envpool/utils/image_process.h:27-36 -> cv::resize
 *     (..., INTER_AREA) on an 8UC1 210x160 image to 84x84.
 *
 * PARITY UNPINNED: cv::resize lives in OpenCV 4.13.0 (pinned at
 * envpool/workspace0.bzl, un-vendored, not installed here) and the only
 * reference test (envpool/utils/image_process_test.cc:23-40) checks shapes.
 * The algorithm restated is OpenCV's generic area resize for non-integer
 * scale factors (imgproc/src/resize.cpp: computeResizeAreaTab +
 * ResizeArea_Invoker<uchar, float>): float taps, horizontal then vertical
 * accumulation in table order, saturate_cast<uchar> = round-half-even.
 * Compile with -ffp-contract=off (x86-64 OpenCV does not fuse).
 * use_inter_area_resize=false (what the reference's benchmark/test_envpool.py:92 selects)
 * is cv::INTER_LINEAR on 8UC1: OpenCV's fixed-point path -- coefficients as shorts scaled by
 * 2^11 (saturate_cast<short>(c * 2048), c in float), horizontal pass into int, vertical pass
 * ((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2 (the uchar specialisation of
 * VResizeLinear; its SIMD twin VResizeLinearVec_32s8u computes the same expression).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int si, di;
  float alpha;
} tap;

static int area_tab(int ssize, int dsize, double scale, tap* tab) {
  int k = 0;
  for (int dx = 0; dx < dsize; dx++) {
    double fsx1 = dx * scale;
    double fsx2 = fsx1 + scale;
    double cell = fmin(scale, ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    if (sx2 > ssize - 1) sx2 = ssize - 1;
    if (sx1 > sx2) sx1 = sx2;
    if (sx1 - fsx1 > 1e-3) {
      tab[k].di = dx;
      tab[k].si = sx1 - 1;
      tab[k++].alpha = (float)((sx1 - fsx1) / cell);
    }
    for (int sx = sx1; sx < sx2; sx++) {
      tab[k].di = dx;
      tab[k].si = sx;
      tab[k++].alpha = (float)(1.0 / cell);
    }
    if (fsx2 - sx2 > 1e-3) {
      tab[k].di = dx;
      tab[k].si = sx2;
      tab[k++].alpha = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell);
    }
  }
  return k;
}

static unsigned char sat_u8(float v) {
  long r = lrintf(v); /* cvRound: round half to even */
  return (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

/* cv::resize(src[sh x sw] 8UC1, dst[dh x dw], INTER_AREA), non-integer scale */
void orc_resize_area_u8(const unsigned char* src, int sh, int sw,
                        unsigned char* dst, int dh, int dw) {
  double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
  tap* xtab = (tap*)malloc(sizeof(tap) * sw * 2);
  tap* ytab = (tap*)malloc(sizeof(tap) * sh * 2);
  int xn = area_tab(sw, dw, scale_x, xtab);
  int yn = area_tab(sh, dh, scale_y, ytab);
  float* buf = (float*)malloc(sizeof(float) * dw);
  float* sum = (float*)malloc(sizeof(float) * dw);
  for (int dx = 0; dx < dw; dx++) sum[dx] = 0;
  int prev_dy = ytab[0].di;
  for (int j = 0; j < yn; j++) {
    float beta = ytab[j].alpha;
    int dy = ytab[j].di, sy = ytab[j].si;
    const unsigned char* S = src + (size_t)sy * sw;
    for (int dx = 0; dx < dw; dx++) buf[dx] = 0;
    for (int k = 0; k < xn; k++) buf[xtab[k].di] += S[xtab[k].si] * xtab[k].alpha;
    if (dy != prev_dy) {
      unsigned char* D = dst + (size_t)prev_dy * dw;
      for (int dx = 0; dx < dw; dx++) {
        D[dx] = sat_u8(sum[dx]);
        sum[dx] = beta * buf[dx];
      }
      prev_dy = dy;
    } else {
      for (int dx = 0; dx < dw; dx++) sum[dx] += beta * buf[dx];
    }
  }
  unsigned char* D = dst + (size_t)prev_dy * dw;
  for (int dx = 0; dx < dw; dx++) D[dx] = sat_u8(sum[dx]);
  free(xtab);
  free(ytab);
  free(buf);
  free(sum);
}

/* one axis of cv::resize's INTER_LINEAR tables: source index and the two fixed-point weights */
static void linear_tab(int ssize, int dsize, int* ofs, short* coef) {
  double scale = (double)ssize / dsize;
  for (int dx = 0; dx < dsize; dx++) {
    float fx = (float)((dx + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) {
      fx = 0;
      sx = 0;
    }
    if (sx >= ssize - 1) {
      fx = 0;
      sx = ssize - 1;
    }
    ofs[dx] = sx;
    long c0 = lrintf((1.f - fx) * 2048.f), c1 = lrintf(fx * 2048.f); /* saturate_cast<short> */
    coef[2 * dx] = (short)c0;
    coef[2 * dx + 1] = (short)c1;
  }
}

/* cv::resize(src[sh x sw] 8UC1, dst[dh x dw], INTER_LINEAR) */
void orc_resize_linear_u8(const unsigned char* src, int sh, int sw, unsigned char* dst, int dh,
                          int dw) {
  int* xofs = (int*)malloc(sizeof(int) * dw);
  int* yofs = (int*)malloc(sizeof(int) * dh);
  short* xa = (short*)malloc(sizeof(short) * 2 * dw);
  short* yb = (short*)malloc(sizeof(short) * 2 * dh);
  linear_tab(sw, dw, xofs, xa);
  linear_tab(sh, dh, yofs, yb);
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = yofs[dy], sy1 = sy0 + 1 < sh ? sy0 + 1 : sh - 1;
    const unsigned char *r0 = src + (size_t)sy0 * sw, *r1 = src + (size_t)sy1 * sw;
    int b0 = yb[2 * dy], b1 = yb[2 * dy + 1];
    for (int dx = 0; dx < dw; dx++) {
      int sx0 = xofs[dx], sx1 = sx0 + 1 < sw ? sx0 + 1 : sw - 1;
      int a0 = xa[2 * dx], a1 = xa[2 * dx + 1];
      int S0 = r0[sx0] * a0 + r0[sx1] * a1, S1 = r1[sx0] * a0 + r1[sx1] * a1;
      dst[(size_t)dy * dw + dx] =
          (unsigned char)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
    }
  }
  free(xofs);
  free(yofs);
  free(xa);
  free(yb);
}

/* Frame-stack pool: stacks[N][S][dh*dw], logical order oldest..newest. */
typedef struct {
  int n, s, sh, sw, dh, dw;
  int linear; /* use_inter_area_resize = false */
  unsigned char* stacks;
} post_pool;

void* orc_atari_post_create(int n, int s, int sh, int sw, int dh, int dw) {
  post_pool* p = (post_pool*)calloc(1, sizeof(post_pool));
  p->n = n; p->s = s; p->sh = sh; p->sw = sw; p->dh = dh; p->dw = dw;
  p->stacks = (unsigned char*)calloc((size_t)n * s * dh * dw, 1);
  return p;
}
void orc_atari_post_set_linear(void* h, int linear) { ((post_pool*)h)->linear = linear; }
void orc_atari_post_destroy(void* h) {
  post_pool* p = (post_pool*)h;
  free(p->stacks);
  free(p);
}
/* frames [k][2][sh][sw]; reset_mask[k]: 1 => push_all=true, maxpool=false
 * (atari_env.h:330-337 / reset path), 0 => push_all=false, maxpool=true. */
void orc_atari_post_push(void* h, const int* env_id, int k,
                         const unsigned char* frames,
                         const unsigned char* reset_mask, unsigned char* obs) {
  post_pool* p = (post_pool*)h;
  size_t fsz = (size_t)p->sh * p->sw, osz = (size_t)p->dh * p->dw;
  unsigned char* pooled = (unsigned char*)malloc(fsz);
  unsigned char* resized = (unsigned char*)malloc(osz);
  for (int i = 0; i < k; ++i) {
    const unsigned char* f0 = frames + (size_t)i * 2 * fsz;
    const unsigned char* f1 = f0 + fsz;
    int rst = reset_mask ? reset_mask[i] : 0;
    for (size_t j = 0; j < fsz; ++j) {
      pooled[j] = rst ? f0[j] : (f0[j] > f1[j] ? f0[j] : f1[j]);
    }
    if (p->linear) {
      orc_resize_linear_u8(pooled, p->sh, p->sw, resized, p->dh, p->dw);
    } else {
      orc_resize_area_u8(pooled, p->sh, p->sw, resized, p->dh, p->dw);
    }
    unsigned char* st = p->stacks + (size_t)env_id[i] * p->s * osz;
    memmove(st, st + osz, (size_t)(p->s - 1) * osz);
    memcpy(st + (size_t)(p->s - 1) * osz, resized, osz);
    if (rst) {
      for (int s = 0; s < p->s - 1; ++s) memcpy(st + (size_t)s * osz, resized, osz);
    }
    memcpy(obs + (size_t)i * p->s * osz, st, (size_t)p->s * osz);
  }
  free(pooled);
  free(resized);
}
