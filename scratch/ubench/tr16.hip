// ds_read_b64_tr_b16 semantics check: 16-lane group, lane 4q+p supplies row q / columns 4p..4p+3; lane i receives column i of the 4 rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const __bf16* in, __bf16* out) {
  __shared__ __attribute__((aligned(16))) __bf16 tile[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) tile[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  const __bf16* addr = tile + (4 * (g >> 1) + q) * 64 + 16 * (g & 1) + 4 * p;
  auto v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)addr);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  __bf16 h[16 * 64], *d, *o, r[256];
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 64; ++j) h[i * 64 + j] = (__bf16)(float)(mode ? j : i);
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf(mode ? "cols:\n" : "rows:\n");
    for (int l = 0; l < 64; l += (l % 16 == 3 ? 13 : 1)) { printf(" lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %3.0f", (float)r[l * 4 + j]); printf("\n"); }
  }
  return 0;
}
