// Probe (not product): how fast can a KERNEL write a result batch into pinned host memory, next to the DMA engine?
// A pipelined sync step downloads the first half's rows of 13 state keys while the second half computes; 13 small
// hipMemcpyAsync per half measured slower than one big copy of everything at the end (profiles/r6f_*), so: one
// gather kernel per half, 16-byte stores straight into the pinned block.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/d2h_kernel_probe.hip -o tools/probes/d2h_kernel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Seg { const char* src; char* dst; size_t bytes; };
struct Segs { Seg s[16]; int n; size_t total16; };
__global__ void GatherToHost(Segs g) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.total16; c += stride) {
    size_t off = c * 16;
    int i = 0;
    while (off >= g.s[i].bytes) { off -= g.s[i].bytes; ++i; }
    *reinterpret_cast<uint4*>(g.s[i].dst + off) = *reinterpret_cast<const uint4*>(g.s[i].src + off);
  }
}
__global__ void Busy(double* x, int iters) {  // one wave per SIMD, register heavy: a stand-in for the step kernel
  double a = x[threadIdx.x], b = 1.0000001;
  for (int i = 0; i < iters; ++i) a = a * b + 1e-9;
  x[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
  const int rows = 32768;
  const size_t rb[13] = {4, 4, 4, 1, 4, 4, 4, 1, 136, 8, 8, 8, 8};  // HalfCheetah-v4 state keys
  size_t total = 0; for (size_t r : rb) total += r * rows;
  char *d, *h; CK(hipMalloc(&d, total)); CK(hipHostMalloc(&h, total, hipHostMallocDefault));
  CK(hipMemset(d, 1, total));
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](const char* name, auto&& f, int reps) {
    for (int i = 0; i < 3; ++i) f();
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.1f us  %6.1f GB/s\n", name, 1e3 * ms / reps, total / (ms / reps) / 1e6);
  };
  printf("half a HalfCheetah batch: %d rows, %zu bytes, 13 keys\n", rows, total);
  time_it("one hipMemcpyAsync D2H", [&] { hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, s); }, 50);
  time_it("13 hipMemcpyAsync D2H (one per key)", [&] {
    size_t off = 0;
    for (size_t r : rb) { hipMemcpyAsync(h + off, d + off, r * rows, hipMemcpyDeviceToHost, s); off += r * rows; }
  }, 50);
  Segs g{}; size_t off = 0; g.n = 13;
  for (int i = 0; i < 13; ++i) { g.s[i] = {d + off, h + off, rb[i] * rows}; off += rb[i] * rows; }
  g.total16 = total / 16;
  for (int blocks : {32, 64, 128, 256, 512, 1024}) {
    char name[64]; snprintf(name, sizeof name, "gather kernel -> pinned host, %d blocks x 256", blocks);
    time_it(name, [&] { hipLaunchKernelGGL(GatherToHost, dim3(blocks), dim3(256), 0, s, g); }, 50);
  }
  // the same while a register-heavy kernel occupies every SIMD on another stream
  double* x; CK(hipMalloc(&x, sizeof(double) * 1024 * 64)); CK(hipMemset(x, 0, sizeof(double) * 1024 * 64));
  for (int blocks : {64, 256}) {
    hipLaunchKernelGGL(Busy, dim3(1024), dim3(64), 0, s2, x, 4000000);  // ~ tens of ms
    char name[80]; snprintf(name, sizeof name, "gather kernel, %d blocks, under a busy kernel", blocks);
    time_it(name, [&] { hipLaunchKernelGGL(GatherToHost, dim3(blocks), dim3(256), 0, s, g); }, 20);
    hipStreamSynchronize(s2);
  }
  hipLaunchKernelGGL(Busy, dim3(1024), dim3(64), 0, s2, x, 4000000);
  time_it("one hipMemcpyAsync D2H, under a busy kernel", [&] { hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, s); }, 20);
  hipStreamSynchronize(s2);
  return 0;
}
