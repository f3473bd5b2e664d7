// Shared helpers of the gfx950 hot-path library (internal; the ABI is include/dskd_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dskd_hip.h"

namespace dskd {

// Thread-local error text behind dskd_last_error().
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DSKD_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DSKD_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set it once on every device a process launches on
// (`done`: a zero-initialised static array of 64 flags owned by the call site).
inline bool reserve_lds(const void* kernel, int bytes, bool* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (done[dev]) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  done[dev] = true;
  return true;
}

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// Blocks with equal blockIdx % 8 share an XCD (and its 4 MiB L2) under the observed
// round-robin dispatch; give each XCD one contiguous range of work items so that
// neighbouring tiles hit the same L2.  Bijective for every grid size; placement is a
// speed hint only (MI355X_MICROARCH.md, "Workgroup dispatch").
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int per = nblk >> 3, rem = nblk & 7;
  const int x = bid & 7, slot = bid >> 3;
  return x * per + (x < rem ? x : rem) + slot;
}

// Sum over aligned groups of 4 / 8 lanes with DPP (no LDS crossbar traffic).
__device__ __forceinline__ float dpp_quad_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
      0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
      0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_half_mirror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
      0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
}
__device__ __forceinline__ float group4_sum(float v) {
  v += dpp_quad_xor1(v);
  v += dpp_quad_xor2(v);
  return v;
}
__device__ __forceinline__ float group8_sum(float v) {
  v = group4_sum(v);
  v += dpp_half_mirror(v);
  return v;
}

__device__ __forceinline__ float group8_max(float v) {
  v = fmaxf(v, dpp_quad_xor1(v));
  v = fmaxf(v, dpp_quad_xor2(v));
  v = fmaxf(v, dpp_half_mirror(v));
  return v;
}

__device__ __forceinline__ float dpp_row_mirror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
}
// reductions over an aligned group of 16 lanes (one DPP row); every lane gets the result
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_quad_xor1(v);
  v += dpp_quad_xor2(v);
  v += dpp_half_mirror(v);
  v += dpp_row_mirror(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_quad_xor1(v));
  v = fmaxf(v, dpp_quad_xor2(v));
  v = fmaxf(v, dpp_half_mirror(v));
  v = fmaxf(v, dpp_row_mirror(v));
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Orders this wave's LDS traffic: lanes of ONE wave exchange data through LDS without a
// workgroup barrier (DS operations of a wave complete in order); the fences only stop the
// compiler from moving accesses across the hand-off.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Zero fill as a KERNEL.  hipMemsetAsync must not be used on the launch path: captured into a
// hipGraph (ROCm 7.0 runtime of this image) the memset node replays with a garbage fill value
// from the second replay on (scratch/graph_memset_repro.py), which silently corrupts results.
template <int kUnused = 0>   // template: one definition shared by every translation unit
__global__ void zero_fill_kernel(u32x4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    p[i] = u32x4{0u, 0u, 0u, 0u};
}
// bytes must be a multiple of 16 and p 16-byte aligned.
inline void zero_fill(void* p, size_t bytes, hipStream_t st) {
  const size_t n16 = bytes / 16;
  if (n16 == 0) return;
  const unsigned blocks = (unsigned)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
  hipLaunchKernelGGL(zero_fill_kernel<0>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<u32x4*>(p), n16);
}

}  // namespace dskd
