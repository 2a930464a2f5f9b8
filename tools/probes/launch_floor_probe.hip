// Probe (not product): what does a launch cost on this machine when the kernel does (a) nothing, (b) ONE dependent
// load -> store per thread over 65536 threads -- the shape of a classic_control step at num_envs = 65536 (config 2)?
// Back-to-back launches on one stream, one event pair around the window.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/launch_floor_probe.hip -o tools/probes/launch_floor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void Empty() {}
__global__ void LoadStore(const double* __restrict__ in, double* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * 1.0000001 + 1.0;
}
// the classic step's memory shape: 4 state doubles + done + cur_step + action in, the same + 42 B of outputs back
__global__ void StepShape(double* s0, double* s1, double* s2, double* s3, unsigned char* done, int* cur, const int* act,
                          float4* obs, float* rew, int* el, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = s0[i], b = s1[i], c = s2[i], d = s3[i];
  const int k = cur[i] + 1;
  const double f = act[i] ? 10.0 : -10.0;
  a += 0.02 * b; b += 0.02 * f; c += 0.02 * d; d += 0.02 * (f - c);
  s0[i] = a; s1[i] = b; s2[i] = c; s3[i] = d;
  done[i] = (a > 2.4 || k >= 500) ? 1 : 0; cur[i] = k;
  obs[i] = make_float4((float)a, (float)b, (float)c, (float)d); rew[i] = 1.0f; el[i] = k;
}
int main() {
  const int n = 65536, reps = 2000;
  double *s[4], *in, *out; unsigned char* done; int *cur, *act, *el; float4* obs; float* rew;
  for (auto& p : s) { hipMalloc(&p, 8 * n); hipMemset(p, 0, 8 * n); }
  hipMalloc(&in, 8 * n); hipMalloc(&out, 8 * n); hipMalloc(&done, n); hipMalloc(&cur, 4 * n); hipMalloc(&act, 4 * n);
  hipMalloc(&el, 4 * n); hipMalloc(&obs, 16 * n); hipMalloc(&rew, 4 * n);
  hipMemset(in, 0, 8 * n); hipMemset(cur, 0, 4 * n); hipMemset(act, 0, 4 * n);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto&& f) {
    for (int i = 0; i < 50; ++i) f();
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %6.2f us per launch (back to back)\n", name, 1e3 * ms / reps);
  };
  for (int block : {64, 256}) {
    char nm[96];
    snprintf(nm, sizeof nm, "empty kernel, %d blocks x %d", n / block, block);
    run(nm, [&] { hipLaunchKernelGGL(Empty, dim3(n / block), dim3(block), 0, st); });
    snprintf(nm, sizeof nm, "one load -> store per thread, %d x %d", n / block, block);
    run(nm, [&] { hipLaunchKernelGGL(LoadStore, dim3(n / block), dim3(block), 0, st, in, out, n); });
    snprintf(nm, sizeof nm, "classic-step memory shape (7 loads, 9 stores), %d x %d", n / block, block);
    run(nm, [&] { hipLaunchKernelGGL(StepShape, dim3(n / block), dim3(block), 0, st, s[0], s[1], s[2], s[3], done, cur, act, obs, rew, el, n); });
  }
  run("empty kernel, 1 block x 64", [&] { hipLaunchKernelGGL(Empty, dim3(1), dim3(64), 0, st); });
  return 0;
}
