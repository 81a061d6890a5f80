// developer microbenchmark: VALU issued between a wave's own MFMAs, by MFMA shape  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NT, int SHAPE>      // SHAPE 0: none, 1: 16x16x32 f16, 2: 32x32x16 f16
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  h16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 big[2];
  for (int i = 0; i < 16; ++i) { big[0][i] = 0; big[1][i] = 0; }
  float v[8], t[4];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  for (int i = 0; i < 4; ++i) t[i] = threadIdx.x * 0.001f + i;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (SHAPE == 1) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
      if (SHAPE == 2) big[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[m & 1], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < NV; ++x) v[(m * NV + x) & 7] = __builtin_fmaf(v[(m * NV + x) & 7], 1.0001f, 0.5f);
#pragma unroll
      for (int x = 0; x < NT; ++x) t[(m * NT + x) & 3] = __builtin_amdgcn_exp2f(t[(m * NT + x) & 3]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += t[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += big[0][i] + big[1][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NV, int NT, int SHAPE>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NV, NT, SHAPE>), dim3(1), dim3(256), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-30s %.1f clk per group (shape %d + %d valu + %d trans)\n", name, (double)c / (iters * 8.0), SHAPE, NV, NT);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0, 0, 1>("16x16x32 only");
  run<0, 0, 2>("32x32x16 only");
  run<4, 0, 0>("4 valu only");
  run<8, 0, 0>("8 valu only");
  run<12, 0, 0>("12 valu only");
  run<4, 0, 1>("16x16x32 + 4 valu");
  run<4, 0, 2>("32x32x16 + 4 valu");
  run<8, 0, 2>("32x32x16 + 8 valu");
  run<12, 0, 2>("32x32x16 + 12 valu");
  run<0, 2, 2>("32x32x16 + 2 trans");
  run<6, 2, 2>("32x32x16 + 6 valu + 2 trans");
  return 0;
}
