// developer probe: what ds_read_b64_tr_b16 returns for per-lane addresses with a row stride (gfx950)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 scripts/probes/probe_tr16.hip -o /tmp/probe_tr16 && /tmp/probe_tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int rowstride, int groupstride) {
  __shared__ short L[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) L[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  const short* p = L + g * groupstride + (i >> 2) * rowstride + (i & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  for (int j = 0; j < 4; ++j) out[4 * l + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 512);
  const int cfg[3][2] = {{16, 64}, {264, 1056}, {264, 8}};
  for (auto& c : cfg) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      const int cc = l & 15, g = l >> 4;
      const int want = g * c[1] + j * c[0] + cc;        // element (row j, column cc) of the group's 4 x 16 block
      if (h[4 * l + j] != (short)want) ++bad;
    }
    printf("rowstride %d groupstride %d: %d mismatches; lane 0: %d %d %d %d  lane 5: %d %d %d %d  lane 21: %d %d %d %d\n", c[0], c[1], bad,
           h[0], h[1], h[2], h[3], h[20], h[21], h[22], h[23], h[84], h[85], h[86], h[87]);
  }
  return 0;
}
