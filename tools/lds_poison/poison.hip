// DIAGNOSTIC (not product): fills the LDS of every CU with NaN bit patterns so that a kernel which
// reads an LDS slot before writing it (and multiplies it by a zero weight) shows up as NaN output
// deterministically instead of depending on what the previous kernel on that CU happened to leave.
// Built by tools/lds_poison/build.sh into tools/lds_poison/libpoison.so; used by
// tools/hum_poison_check.py.
#include <hip/hip_runtime.h>

__global__ void PoisonLds(unsigned* sink, unsigned pattern) {
  extern __shared__ unsigned buf[];
  const int n = 65536 / 4;
  for (int i = threadIdx.x; i < n; i += blockDim.x) buf[i] = pattern;
  __syncthreads();
  // keep the stores alive
  if (buf[(threadIdx.x * 7) % n] != pattern) sink[0] = 1;
  // stay resident for a while so that later blocks land on other LDS offsets of the same CU
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);
}

extern "C" int poison_lds(void* stream, unsigned pattern) {
  static unsigned* sink = nullptr;
  if (!sink && hipMalloc(&sink, 4) != hipSuccess) return 1;
  (void)hipFuncSetAttribute((const void*)PoisonLds, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(PoisonLds, dim3(256 * 8), dim3(256), 65536, (hipStream_t)stream, sink, pattern);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
