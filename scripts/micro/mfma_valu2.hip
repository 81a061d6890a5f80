// developer microbenchmark: two waves on one SIMD, one issuing MFMAs, the other VALU / transcendentals: max or sum?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode bit 0: waves 0..3 run the MFMA loop; bit 1: waves 4..7 run the VALU loop (NV fma + NT exp per group)
template <int NV, int NT>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, unsigned long long* cyc) {
  const int wv = threadIdx.x >> 6;
  h16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8], t[4];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  for (int i = 0; i < 4; ++i) t[i] = threadIdx.x * 0.001f + i;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  if (wv < 4) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
          for (int x = 0; x < NV; ++x) v[(m * NV + x) & 7] = __builtin_fmaf(v[(m * NV + x) & 7], 1.0001f, 0.5f);
#pragma unroll
          for (int x = 0; x < NT; ++x) t[(m * NT + x) & 3] = __builtin_amdgcn_exp2f(t[(m * NT + x) & 3]);
        }
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += t[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wv] = t1 - t0;
}
template <int NV, int NT>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
  const int iters = 2000;
  for (int mode = 1; mode <= 3; ++mode) {
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NV, NT>), dim3(1), dim3(512), 0, 0, out, iters, mode, cyc);
    hipDeviceSynchronize();
    unsigned long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-26s mode %d: mfma wave %.1f clk / mfma, valu wave %.1f clk / group (%d valu + %d trans)\n", name, mode,
           (double)c[0] / (iters * 8.0), (double)c[4] / (iters * 8.0), NV, NT);
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<4, 0>("4 valu");
  run<8, 0>("8 valu");
  run<0, 2>("2 trans");
  run<4, 1>("4 valu + 1 trans");
  run<6, 2>("6 valu + 2 trans");
  return 0;
}
