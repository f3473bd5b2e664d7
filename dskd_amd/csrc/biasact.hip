// Epilogue of a folded convolution, one pass in place:   y = act(x + bias[c] (+ identity))
//
// ResNet's Bottleneck (reference mmdet/models/backbones/resnet.py:271-303) is
// conv -> BN -> ReLU three times with `out += identity` before the last ReLU.  With the frozen
// BatchNorm folded into the convolution the BN leaves a per-channel bias behind, and PyTorch
// runs MIOpen's convolution, a separate bias add, the residual add and the ReLU as up to four
// launches that each stream the whole activation (137 MB per tensor at 256 x 200 x 334, B=4):
// measured on MI355X the bias + ReLU passes cost as much as the convolutions themselves.
// (MIOpen's own fused conv+bias+activation API falls back to 10-60 ms kernels for these shapes:
// scratch/conv_fused_relu.py.)  This is the single streaming pass that replaces them.
//
// channels_last memory ([N, H, W, C], C innermost), 16 bytes per lane and step; HBM-bound.
#include "common.h"

namespace dskd {
namespace {

template <typename T> struct Vec;
template <> struct Vec<float> { static constexpr int n = 4; };
template <> struct Vec<__bf16> { static constexpr int n = 8; };

template <typename T>
__device__ __forceinline__ void load_vec(const T* p, float* f) {
  if constexpr (sizeof(T) == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    // (bit_cast of a vector ELEMENT reads element 0 with this compiler: go through scalars)
    const u32x4 v = *reinterpret_cast<const u32x4*>(p);
    const unsigned a = v.x, b = v.y, c = v.z, d = v.w;
    f[0] = __builtin_bit_cast(float, a << 16); f[1] = __builtin_bit_cast(float, a & 0xFFFF0000u);
    f[2] = __builtin_bit_cast(float, b << 16); f[3] = __builtin_bit_cast(float, b & 0xFFFF0000u);
    f[4] = __builtin_bit_cast(float, c << 16); f[5] = __builtin_bit_cast(float, c & 0xFFFF0000u);
    f[6] = __builtin_bit_cast(float, d << 16); f[7] = __builtin_bit_cast(float, d & 0xFFFF0000u);
  }
}

template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float* f) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
  } else {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const bf16x8 v = {(__bf16)f[0], (__bf16)f[1], (__bf16)f[2], (__bf16)f[3],
                      (__bf16)f[4], (__bf16)f[5], (__bf16)f[6], (__bf16)f[7]};
    *reinterpret_cast<bf16x8*>(p) = v;
  }
}

template <typename T, bool HAS_ID, bool RELU>
__global__ __launch_bounds__(256) void bias_act_kernel(T* __restrict__ x, const T* __restrict__ bias,
                                                       const T* __restrict__ identity, long long nvec, int C) {
  constexpr int V = Vec<T>::n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * V;
    const int c = (int)(e % C);                    // C is a multiple of V: a vector never straddles a pixel
    float xv[V], bv[V];
    load_vec(x + e, xv);
    load_vec(bias + c, bv);
#pragma unroll
    for (int k = 0; k < V; ++k) xv[k] += bv[k];
    if constexpr (HAS_ID) {
      float iv[V];
      load_vec(identity + e, iv);
#pragma unroll
      for (int k = 0; k < V; ++k) xv[k] += iv[k];
    }
    if constexpr (RELU) {
#pragma unroll
      for (int k = 0; k < V; ++k) xv[k] = fmaxf(xv[k], 0.f);
    }
    store_vec(x + e, xv);
  }
}

template <typename T>
void launch(T* x, const T* bias, const T* identity, long long n, int C, bool relu, hipStream_t st) {
  const long long nvec = n / Vec<T>::n;
  const long long want = (nvec + 255) / 256;
  const dim3 grid((unsigned)(want < 8192 ? want : 8192)), block(256);
  if (identity) {
    if (relu) hipLaunchKernelGGL((bias_act_kernel<T, true, true>), grid, block, 0, st, x, bias, identity, nvec, C);
    else hipLaunchKernelGGL((bias_act_kernel<T, true, false>), grid, block, 0, st, x, bias, identity, nvec, C);
  } else {
    if (relu) hipLaunchKernelGGL((bias_act_kernel<T, false, true>), grid, block, 0, st, x, bias, identity, nvec, C);
    else hipLaunchKernelGGL((bias_act_kernel<T, false, false>), grid, block, 0, st, x, bias, identity, nvec, C);
  }
}

// The stem's tail where nothing needs a gradient (the teacher, and the student's frozen stem): conv1 -> folded BN -> ReLU ->
// MaxPool(3, stride 2, padding 1) (reference resnet.py:633-640) leaves a bias + ReLU pass over [B, 64, 400, 667] (137 MB read
// and written) and the pooling (137 MB read, 34 MB written).  Rounding and ReLU are monotonic and the bias is constant over
// a window, so max(relu(round(x + b))) = relu(round(max(x) + b)): one pass, 137 MB read + 34 MB written, bit-identical.
// thread = (output pixel, 16-byte channel group); the windows of neighbouring outputs overlap in L1 / L2.
template <typename T>
__global__ __launch_bounds__(256) void bias_relu_maxpool_kernel(const T* __restrict__ x, const T* __restrict__ bias,
                                                                T* __restrict__ y, int H, int W, int Ho, int Wo, int C,
                                                                long long nvec) {
  constexpr int V = Vec<T>::n;
  const int cv = C / V;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * V;
    long long p = i / cv;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const long long b = p / Ho;
    float m[V];
#pragma unroll
    for (int k = 0; k < V; ++k) m[k] = -INFINITY;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int hi = 2 * ho + dy;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int wi = 2 * wo + dx;
        if ((unsigned)wi >= (unsigned)W) continue;
        float v[V];
        load_vec(x + ((b * H + hi) * W + wi) * C + c, v);
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = fmaxf(m[k], v[k]);
      }
    }
    float bv[V];
    load_vec(bias + c, bv);
    if constexpr (sizeof(T) == 2) {
      // round(x + b) first (what the unfused pass stores), then ReLU
      float r[V];
#pragma unroll
      for (int k = 0; k < V; ++k) r[k] = fmaxf((float)(__bf16)(m[k] + bv[k]), 0.f);
      store_vec(y + i * V, r);
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) m[k] = fmaxf(m[k] + bv[k], 0.f);
      store_vec(y + i * V, m);
    }
  }
}

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int dskd_bias_relu_maxpool(const void* x, const void* bias, void* y, int B, int H, int W, int C, int dtype,
                                      void* stream) {
  if (dtype != DSKD_DTYPE_F32 && dtype != DSKD_DTYPE_BF16)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_bias_relu_maxpool: unknown dtype %d", dtype);
  const int V = dtype == DSKD_DTYPE_F32 ? 4 : 8;
  if (!x || !bias || !y || B < 0 || H <= 0 || W <= 0 || C <= 0 || C % V != 0)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_bias_relu_maxpool: need C %% %d == 0 and positive sizes (B=%d, H=%d, W=%d, C=%d)", V,
                B, H, W, C);
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(y)) & 15)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_bias_relu_maxpool: pointers must be 16-byte aligned");
  if (B == 0) return DSKD_OK;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long nvec = (long long)B * Ho * Wo * (C / V);
  const long long want = (nvec + 255) / 256;
  const dim3 grid((unsigned)(want < 65536 ? want : 65536)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DSKD_DTYPE_F32)
    hipLaunchKernelGGL(bias_relu_maxpool_kernel<float>, grid, block, 0, st, (const float*)x, (const float*)bias, (float*)y, H, W,
                       Ho, Wo, C, nvec);
  else
    hipLaunchKernelGGL(bias_relu_maxpool_kernel<__bf16>, grid, block, 0, st, (const __bf16*)x, (const __bf16*)bias, (__bf16*)y, H,
                       W, Ho, Wo, C, nvec);
  return check_launch("dskd_bias_relu_maxpool");
}

extern "C" int dskd_bias_act(void* x, const void* bias, const void* identity, int64_t n, int C, int relu,
                             int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_F32 && dtype != DSKD_DTYPE_BF16)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_bias_act: unknown dtype %d", dtype);
  const int V = dtype == DSKD_DTYPE_F32 ? 4 : 8;
  if (!x || !bias || n < 0 || C <= 0 || C % V != 0 || n % C != 0)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_bias_act: need n %% C == 0 and C %% %d == 0 (n=%lld, C=%d)", V,
                (long long)n, C);
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(identity)) & 15)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_bias_act: pointers must be 16-byte aligned");
  if (n == 0) return DSKD_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DSKD_DTYPE_F32)
    launch<float>((float*)x, (const float*)bias, (const float*)identity, n, C, relu != 0, st);
  else
    launch<__bf16>((__bf16*)x, (const __bf16*)bias, (const __bf16*)identity, n, C, relu != 0, st);
  return check_launch("dskd_bias_act");
}
