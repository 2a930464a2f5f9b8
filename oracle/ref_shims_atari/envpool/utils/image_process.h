// TEST INFRASTRUCTURE — shim, not product code.
// Shadows the reference's envpool/utils/image_process.h (which needs OpenCV 4.13, un-vendored:
// envpool/workspace0.bzl) when its atari_env.h is compiled in place: `Resize` keeps the
// reference's signature and channel handling (image_process.h:27-36) and delegates the pixel
// arithmetic to the plain-C restatement of cv::resize in oracle/atari/atari_post.c
// (INTER_AREA generic path / INTER_LINEAR 8-bit fixed point; channels of an interleaved image
// are resized independently with the same tables, as OpenCV does for CV_8UC(n)).
#ifndef ORACLE_REF_SHIMS_ATARI_IMAGE_PROCESS_H_
#define ORACLE_REF_SHIMS_ATARI_IMAGE_PROCESS_H_

#include <cstdint>
#include <vector>

#include "envpool/core/array.h"

extern "C" {
void orc_resize_area_u8(const unsigned char* src, int sh, int sw, unsigned char* dst, int dh, int dw);
void orc_resize_linear_u8(const unsigned char* src, int sh, int sw, unsigned char* dst, int dh, int dw);
}

inline void Resize(const Array& src, Array* tgt, bool use_inter_area = true) {
  const int channel = static_cast<int>(src.Shape(2));
  const int sh = static_cast<int>(src.Shape(0)), sw = static_cast<int>(src.Shape(1));
  const int dh = static_cast<int>(tgt->Shape(0)), dw = static_cast<int>(tgt->Shape(1));
  const auto* s = static_cast<const std::uint8_t*>(src.Data());
  auto* d = static_cast<std::uint8_t*>(tgt->Data());
  std::vector<std::uint8_t> plane(static_cast<std::size_t>(sh) * sw), out(static_cast<std::size_t>(dh) * dw);
  for (int c = 0; c < channel; ++c) {
    for (int i = 0; i < sh * sw; ++i) plane[i] = s[i * channel + c];
    if (use_inter_area) {
      orc_resize_area_u8(plane.data(), sh, sw, out.data(), dh, dw);
    } else {
      orc_resize_linear_u8(plane.data(), sh, sw, out.data(), dh, dw);
    }
    for (int i = 0; i < dh * dw; ++i) d[i * channel + c] = out[i];
  }
}

#endif  // ORACLE_REF_SHIMS_ATARI_IMAGE_PROCESS_H_
