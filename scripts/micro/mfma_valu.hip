// developer microbenchmark: does VALU work issued between a wave's own MFMAs run under them?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, int NT, bool MF>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  h16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8], t[4];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  for (int i = 0; i < 4; ++i) t[i] = threadIdx.x * 0.001f + i;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MF) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
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
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NV, int NT, bool MF>
void run(const char* name, int waves) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NV, NT, MF>), dim3(1), dim3(64 * waves), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s waves %d: %.1f clk per group (1 mfma=%d + %d valu + %d trans)\n", name, waves, (double)c / (iters * 8.0), (int)MF, NV, NT);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {1, 4, 8}) {
    run<0, 0, true>("mfma only", w);
    run<3, 0, false>("3 valu only", w);
    run<3, 0, true>("mfma + 3 valu", w);
    run<6, 0, true>("mfma + 6 valu", w);
    run<0, 2, false>("2 trans only", w);
    run<0, 2, true>("mfma + 2 trans", w);
    run<2, 1, true>("mfma + 2 valu + 1 trans", w);
    run<8, 0, false>("8 valu only", w);
    run<8, 0, true>("mfma + 8 valu", w);
  }
  return 0;
}
