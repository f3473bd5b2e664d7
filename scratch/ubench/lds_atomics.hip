// microbenchmark: LDS atomic throughput per CU (f32 add, u32 add, rtn u32, plain write, plain RMW,
// and -- not yet run: next round -- u64 add and packed bf16 / f16 add: two channels per LDS atomic.
// A 64-bit integer add carries TWO 32-bit fixed-point channels exactly: add (hi << 32) + (int64)lo, decode
// lo = (int32)sum, hi = (int32)((sum - lo) >> 32); if ds_add_u64 issues at the ds_add_u32 rate per
// wave instruction the windowed grad_value kernel's main loop (32 ds_add_u32 per (query, head)) halves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int stride) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int addr = (lane * stride + wave * 977) & 16383;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) atomicAdd(&lds[addr], 1.0f);
    if (MODE == 1) atomicAdd((unsigned*)&lds[addr], 1u);
    if (MODE == 2) acc += atomicAdd((unsigned*)&lds[addr], 1u);
    if (MODE == 3) lds[addr] = (float)i;
    if (MODE == 4) { float v = lds[addr]; lds[addr] = v + 1.0f; }
    if (MODE == 5) atomicMax((unsigned*)&lds[addr], (unsigned)i);
    if (MODE == 6) atomicAdd((unsigned long long*)&lds[addr & ~1], 0x100000001ull);
    if (MODE == 7) {
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      (void)__builtin_amdgcn_ds_atomic_fadd_v2bf16((__attribute__((address_space(3))) bf16x2_t*)&lds[addr],
                                                   bf16x2_t{(__bf16)1.0f, (__bf16)1.0f});
    }
    if (MODE == 8) {
      typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
      (void)__builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) f16x2_t*)&lds[addr],
                                                  f16x2_t{(_Float16)1.0f, (_Float16)1.0f});
    }
    addr = (addr + 64 * stride + 1) & 16383;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[5] + acc;
}
template <int MODE> void run(const char* name, int stride) {
  float* d; hipMalloc(&d, 4096);
  const int iters = 4096, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d, iters, stride);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d, iters, stride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per CU: 16 waves x iters wave-instructions
  double cyc = ms * 1e-3 * 2.4e9 / (16.0 * iters);
  printf("%-28s stride %d: %.3f ms  -> %.1f cycles per wave-instruction per CU\n", name, stride, ms, cyc);
}
int main() {
  for (int stride : {1, 33}) {
    run<0>("ds_add_f32", stride); run<1>("ds_add_u32", stride); run<2>("ds_add_rtn_u32", stride);
    run<3>("ds_write_b32", stride); run<4>("read+add+write", stride); run<5>("ds_max_u32", stride);
    run<6>("ds_add_u64", stride); run<7>("ds_pk_add_bf16", stride); run<8>("ds_pk_add_f16", stride);
  }
  return 0;
}
