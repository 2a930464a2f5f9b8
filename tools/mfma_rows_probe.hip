// A/B probe (not product code): the one GEMM-shaped piece of the Humanoid constraint stage --
// the dual matrix  A = Y D^-1 Y^T  of an env's constraint rows (K3d, mj_hum4.hip.h: rows x 23, rows <= 32
// on chip) -- formed two ways on gfx950, same inputs, same outputs:
//
//   quad   the product kernel's form: one env per lane quad, a row's 23 numbers distributed over the four
//          lanes (7 per lane: 9 trunk entries split 3/2/2/2 + the lane's 4 limb entries, zero padded), an
//          entry A_rc = 7 FMAs per lane + ONE quad reduction (two DPP quad_perm butterflies); four
//          register-resident columns at a time, the other rows streamed past them; lower triangle only.
//   mfma   v_mfma_f64_16x16x4_f64: 32 x 24 per env as two 16-row operand tiles, three 16x16 output tiles
//          (the symmetric fourth skipped), 6 k-steps each = 18 MFMAs per env, accumulators (4 f64 per lane
//          and tile) where the compiler puts them (AGPRs are legal for MFMA C/D); operands straight from
//          global memory in [env][row][k] layout -- the BEST case for MFMA: in the step kernel the rows are
//          produced in the quad-distributed form, 16 envs at a time, and would first have to be transposed
//          through LDS (16 envs x 32 x 24 doubles = 98 KB per wave; the kernel has 40 KB).
//
// fp64 matrix peak = fp64 vector peak on MI355X (78.6 TFLOP/s: 32 flop/clk/SIMD either way), so no rate
// win is expected; the question the round-3 review asked is whether the MFMA form is at least not slower,
// because its accumulators could live in AGPRs.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_rows_probe.hip -o gpurun_out/mfma_rows_probe
//   gpurun_out/mfma_rows_probe            (prints us per launch, cycles per env, max |difference| vs the CPU)
//   rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU --kernel-trace ... -- gpurun_out/mfma_rows_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                   \
    }                                                                                 \
  } while (0)

constexpr int kR = 32;   // rows per env
constexpr int kK = 24;   // 23 dofs, padded
constexpr int kShare = 7;  // numbers of a row a lane of the quad holds

using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int CTRL>
__device__ __forceinline__ double DppD(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double QuadSum(double x) {
  x += DppD<0xB1>(x);  // quad_perm [1,0,3,2]
  x += DppD<0x4E>(x);  // quad_perm [2,3,0,1]
  return x;
}

// ---- quad form.  yq: [env][row][lane of quad][kShare]; out: [env][kR][kR] (lower triangle written)
__global__ __launch_bounds__(64) void QuadKernel(const double* __restrict__ yq, double* __restrict__ out, int n_env) {
  const int lane = threadIdx.x, l = lane & 3;
  const int env = blockIdx.x * 16 + (lane >> 2);
  if (env >= n_env) return;
  const double* y = yq + ((size_t)env * kR * 4 + l) * kShare;
  double* o = out + (size_t)env * kR * kR;
  for (int c0 = 0; c0 < kR; c0 += 4) {  // four resident columns
    double col[4][kShare];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int i = 0; i < kShare; ++i) col[c][i] = y[(size_t)(c0 + c) * 4 * kShare + i];
    }
    for (int r = c0; r < kR; ++r) {  // rows streamed past them
      double row[kShare];
#pragma unroll
      for (int i = 0; i < kShare; ++i) row[i] = y[(size_t)r * 4 * kShare + i];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < kShare; ++i) s = __builtin_fma(row[i], col[c][i], s);
        s = QuadSum(s);
        if (l == c && r >= c0 + c) o[(size_t)r * kR + c0 + c] = s;
      }
    }
  }
}

// ---- MFMA form.  z: [env][row][kK] (= Y D^-1/2); out as above (tiles 00, 10, 11)
__global__ __launch_bounds__(64) void MfmaKernel(const double* __restrict__ z, double* __restrict__ out, int n_env) {
  const int lane = threadIdx.x;
  const int i16 = lane & 15, k4 = lane >> 4;
  for (int e = 0; e < 16; ++e) {  // the 16 envs of a wave of the step kernel, one after the other
    const int env = blockIdx.x * 16 + e;
    if (env >= n_env) return;
    const double* zz = z + (size_t)env * kR * kK;
    f64x4 d00 = {0, 0, 0, 0}, d10 = {0, 0, 0, 0}, d11 = {0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < kK / 4; ++kk) {
      const double a0 = zz[(size_t)i16 * kK + 4 * kk + k4];         // rows 0..15:  A[i][k] and B[k][j]
      const double a1 = zz[(size_t)(16 + i16) * kK + 4 * kk + k4];  // rows 16..31
      d00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, d00, 0, 0, 0);
      d10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a0, d10, 0, 0, 0);
      d11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, d11, 0, 0, 0);
    }
    double* o = out + (size_t)env * kR * kR;
#pragma unroll
    for (int v = 0; v < 4; ++v) {  // C/D: col = lane & 15, row = (lane >> 4) + 4 * v
      const int row = k4 + 4 * v, colj = i16;
      if (row >= colj) o[(size_t)row * kR + colj] = d00[v];
      o[(size_t)(16 + row) * kR + colj] = d10[v];
      if (row >= colj) o[(size_t)(16 + row) * kR + 16 + colj] = d11[v];
    }
  }
}

int main(int argc, char** argv) {
  const int n_env = argc > 1 ? std::atoi(argv[1]) : 65536;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 20;
  std::vector<double> z((size_t)n_env * kR * kK), yq((size_t)n_env * kR * 4 * kShare, 0.0);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
  };
  // the quad's split of a 23-vector: trunk entries 0..8 over the lanes as 3/2/2/2, then limb l's 4 entries
  auto owner = [](int k, int* l, int* pos) {
    if (k < 9) {
      static const int start[5] = {0, 3, 5, 7, 9};
      for (int q = 0; q < 4; ++q) {
        if (k < start[q + 1]) { *l = q; *pos = k - start[q]; return; }
      }
    }
    *l = (k - 9) / 4; *pos = 3 + (k - 9) % 4;  // limb entries 9..24 -> lane (k-9)/4; (lane 3's 4th: the pad)
    if (*l > 3) { *l = 3; *pos = 6; }
  };
  for (int e = 0; e < n_env; ++e) {
    for (int r = 0; r < kR; ++r) {
      for (int k = 0; k < kK; ++k) {
        const double v = k < 23 ? rnd() : 0.0;
        z[((size_t)e * kR + r) * kK + k] = v;
        int l, pos;
        owner(k, &l, &pos);
        if (k < 23) yq[(((size_t)e * kR + r) * 4 + l) * kShare + pos] = v;
      }
    }
  }
  double *dz, *dyq, *da, *db;
  const size_t ob = sizeof(double) * (size_t)n_env * kR * kR;
  CK(hipMalloc(&dz, z.size() * 8)); CK(hipMalloc(&dyq, yq.size() * 8));
  CK(hipMalloc(&da, ob)); CK(hipMalloc(&db, ob));
  CK(hipMemcpy(dz, z.data(), z.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dyq, yq.data(), yq.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(da, 0, ob)); CK(hipMemset(db, 0, ob));
  const int blocks = (n_env + 15) / 16;
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  float ms_q = 0, ms_m = 0;
  for (int w = 0; w < 2; ++w) {  // first round warms up
    CK(hipEventRecord(t0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(QuadKernel, dim3(blocks), dim3(64), 0, 0, dyq, da, n_env);
    CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms_q, t0, t1));
    CK(hipEventRecord(t0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(MfmaKernel, dim3(blocks), dim3(64), 0, 0, dz, db, n_env);
    CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms_m, t0, t1));
  }
  std::vector<double> a((size_t)n_env * kR * kR), b(a.size());
  CK(hipMemcpy(a.data(), da, ob, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), db, ob, hipMemcpyDeviceToHost));
  double eq = 0, em = 0;
  for (int e = 0; e < n_env; e += 997) {
    for (int r = 0; r < kR; ++r) {
      for (int c = 0; c <= r; ++c) {
        double ref = 0;
        for (int k = 0; k < 23; ++k) ref += z[((size_t)e * kR + r) * kK + k] * z[((size_t)e * kR + c) * kK + k];
        eq = std::fmax(eq, std::fabs(a[((size_t)e * kR + r) * kR + c] - ref));
        em = std::fmax(em, std::fabs(b[((size_t)e * kR + r) * kR + c] - ref));
      }
    }
  }
  int clk_khz = 0;
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  const double waves_per_simd = (double)blocks / 1024.0;
  auto report = [&](const char* name, float ms, double err) {
    const double us = 1e3 * ms / reps;
    // a wave = 16 envs; 1024 SIMDs; cycles a SIMD spends per wave if the launch is evenly spread
    const double cyc_wave = us * 1e-6 * clk_khz * 1e3 / waves_per_simd;
    std::printf("%-5s %8.1f us per launch of %d envs   %7.0f SIMD cycles per 16-env wave   %5.0f per env   max |err| %.2e\n",
                name, us, n_env, cyc_wave, cyc_wave / 16, err);
  };
  std::printf("A = Y D^-1 Y^T, %d rows x 23 per env, lower triangle, fp64, clock %d MHz\n", kR, clk_khz / 1000);
  report("quad", ms_q, eq);
  report("mfma", ms_m, em);
  std::printf("flops per env (triangle): %d   quad issues 7 FMA + 2 DPP sums per entry; mfma 18 x 16x16x4 f64 per env\n",
              kR * (kR + 1) / 2 * 23 * 2);
  return (eq < 1e-12 && em < 1e-12) ? 0 : 2;
}
