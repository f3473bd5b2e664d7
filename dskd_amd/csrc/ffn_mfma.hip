// The encoder FFN as ONE MFMA kernel per direction (bf16, d_model 256, hidden 1024).
//
// ext-mmcv FFN = Linear(256,1024) -> ReLU -> Dropout -> Linear(1024,256) (mmcv/cnn/bricks/transformer.py,
// run per encoder layer by mmdet/models/utils/transformer.py:454-483) over T = B * 22 223 = 88 892 tokens at B=4.
// As library GEMMs the 1024-wide hidden activation H [T, 1024] (182 MB) is written by the first GEMM, rewritten by
// the dropout pass and read by the second GEMM; the backward walks it (and its gradient) as often again.  Here a
// wave keeps 32 tokens for the whole layer:
//
//   forward    Y^T[256, 32] = W2 . dropout(relu(W1 . X^T + b1)) + b2      (H leaves the chip once, for the backward)
//   backward   dX^T[256, 32] = W1^T . g1^T,  g1^T = (W2^T . dY^T) * [H > 0] / (1 - p)   (g1 leaves once, for dW1)
//
// Both are the same loop over 32 tiles of 32 hidden units: GEMM-1 (K = 256, 16 x v_mfma_f32_32x32x16_bf16) gives
// a [32 hidden, 32 token] f32 tile whose column sits on the lane and whose rows sit in the 16 accumulator
// registers; after the elementwise step it is converted to bf16 IN PLACE and is the B operand of GEMM-2
// (K = those 32 hidden units, 8 output tiles x 2 MFMAs), which sums over exactly that register index -- no LDS
// round trip and no lane movement (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand").
// The A operands are the weights, pre-packed once per step into MFMA fragment order (ffn_pack_kernel: 2 x 1 MB) so
// that a workgroup stages a tile's 32 KB with sixteen-byte LDS-DMA loads (global_load_lds), double buffered, one
// barrier per tile.  Rows of an A tile are permuted (pi below) so that a lane's 16 accumulator registers are 16
// CONSECUTIVE hidden units / output features: H, g1, Y, dX are read and written as 32-byte runs per lane
// (64 bytes per token with the partner lane), plain row-major tensors for the weight-gradient GEMMs that follow.
//
// MFMA work per launch 93 GFLOP (37 us at the 2.5 PFLOP/s dense peak); HBM 272 MB forward / 454 MB backward.
// Dropout is the mask of dskd_dropout_fwd (Philox4x32-10 on element index / 8, 16-bit fields), never stored: the
// backward reads it off H like dskd_relu_dropout_bwd.
#include "common.h"

namespace dskd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = 256;            // d_model
constexpr int kF = 1024;           // hidden
constexpr int kTiles = kF / 32;    // hidden tiles
constexpr int kTileBytes = 32768;  // 16 KB GEMM-1 fragments + 16 KB GEMM-2 fragments
constexpr int kFragsPerTile = kTileBytes / 16;

// MFMA row slot r of a 32-row A tile carries row pi(r) of the matrix: with the C/D map
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) this makes accumulator register i of lane half h row 16 h + i.
__device__ __forceinline__ int pi_row(int r) { return (r & 3) + 4 * (r >> 3) + 16 * ((r >> 2) & 1); }

// Fragment f of direction `dir` (0 forward, 1 backward), 16 bytes each:
//   tile ht | part 0: GEMM-1, k-step s (16)          lane (r, h) element j: k = 128 h + 8 s + j, row = 32 ht + pi(r)
//           | part 1: GEMM-2, output tile ot, step s  lane (r, h) element j: k = 32 ht + 16 h + 8 s + j, row = 32 ot + pi(r)
// forward: GEMM-1 A = W1 [hidden][d], GEMM-2 A = W2 [out][hidden]; backward: GEMM-1 A = W2^T, GEMM-2 A = W1^T.
__global__ __launch_bounds__(256) void ffn_pack_kernel(const __bf16* __restrict__ W1, const __bf16* __restrict__ W2,
                                                       __bf16* __restrict__ fwdp, __bf16* __restrict__ bwdp) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  const int dir = blockIdx.y;
  __bf16* dst = dir ? bwdp : fwdp;
  if (!dst) return;
  const int ht = f / kFragsPerTile, rem = f % kFragsPerTile;
  const int part = rem >> 10, q = rem & 1023, lane = q & 63, blk = q >> 6;
  const int r = lane & 31, h = lane >> 5;
  bf16x8 v;
  if (part == 0) {
    const int hid = 32 * ht + pi_row(r), k0 = 128 * h + 8 * blk;
    if (dir == 0) {
      v = *reinterpret_cast<const bf16x8*>(W1 + (size_t)hid * kD + k0);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = W2[(size_t)(k0 + j) * kF + hid];
    }
  } else {
    const int ot = blk >> 1, s = blk & 1;
    const int row = 32 * ot + pi_row(r), hid0 = 32 * ht + 16 * h + 8 * s;
    if (dir == 0) {
      v = *reinterpret_cast<const bf16x8*>(W2 + (size_t)row * kF + hid0);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = W1[(size_t)(hid0 + j) * kD + row];
    }
  }
  *reinterpret_cast<bf16x8*>(dst + (size_t)f * 8) = v;
}

__device__ __forceinline__ u32x4 philox8(unsigned long long idx, unsigned long long seed, unsigned long long offset) {
  unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

struct FfnArgs {
  const __bf16* in;      // X (forward) / dY (backward)  [T, 256]
  const __bf16* wp;      // packed weights of this direction [32 tiles][32 KB]
  const __bf16* b1;      // forward: bias of the first Linear [1024]
  const __bf16* b2;      // forward: bias of the second Linear [256]
  const __bf16* h_in;    // backward: H = dropout(relu(.)) [T, 1024]
  __bf16* h_out;         // forward (training): H;  backward: g1  [T, 1024]
  __bf16* out;           // Y / dX [T, 256]
  long long T;
  float scale;           // 1 / (1 - p)
  unsigned thresh16;     // drop when the 16-bit field < thresh16 (0: no dropout)
  unsigned long long seed, offset;
  const unsigned long long* epoch;
};

enum { kFwdTrain = 0, kFwdEval = 1, kBwd = 2 };

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ffn_fused_kernel(const FfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];       // 2 x 32 KB weight tiles | 4 KB b1 as f32
  float* const s_b1 = reinterpret_cast<float*>(smem + 2 * kTileBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const long long tok = (long long)blockIdx.x * (WAVES * 32) + wave * 32 + r;
  const bool live = tok < a.T;
  const long long tk = live ? tok : a.T - 1;

  auto stage = [&](int ht, int buf) {
    const char* src = reinterpret_cast<const char*>(a.wp) + (size_t)ht * kTileBytes;
    char* dst = smem + buf * kTileBytes;
#pragma unroll
    for (int p = 0; p < 32 / WAVES; ++p) {
      const int blk = p * WAVES + wave;                              // 1 KB per wave instruction
      __builtin_amdgcn_global_load_lds(
          reinterpret_cast<const __attribute__((address_space(1))) void*>(src + blk * 1024 + lane * 16),
          reinterpret_cast<__attribute__((address_space(3))) void*>(dst + blk * 1024), 16, 0, 0);
    }
  };

  stage(0, 0);
  if (MODE != kBwd)
    for (int i = threadIdx.x; i < kF; i += WAVES * 64) s_b1[i] = (float)a.b1[i];

  bf16x8 xf[16];                                                     // this lane's half row: k = 128 h + 8 s + j
  {
    const __bf16* xrow = a.in + tk * kD + 128 * h;
#pragma unroll
    for (int s = 0; s < 16; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xrow + 8 * s);
  }
  f32x16 yacc[8];
#pragma unroll
  for (int ot = 0; ot < 8; ++ot)
#pragma unroll
    for (int i = 0; i < 16; ++i) yacc[ot][i] = 0.f;

  unsigned long long offset = 0;
  if (MODE == kFwdTrain) offset = a.offset + (a.epoch ? *a.epoch : 0ull);
  __syncthreads();

  for (int ht = 0; ht < kTiles; ++ht) {
    if (ht + 1 < kTiles) stage(ht + 1, (ht + 1) & 1);
    const char* wb = smem + (ht & 1) * kTileBytes + lane * 16;
    const long long hoff = tk * kF + 32 * ht + 16 * h;               // this lane's 16 hidden units of the tile

    bf16x8 hin[2];
    if (MODE == kBwd) {
      hin[0] = *reinterpret_cast<const bf16x8*>(a.h_in + hoff);
      hin[1] = *reinterpret_cast<const bf16x8*>(a.h_in + hoff + 8);
    }
    f32x16 acc;
    if (MODE != kBwd) {
      const f32x4* bp = reinterpret_cast<const f32x4*>(s_b1 + 32 * ht + 16 * h);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = bp[g];
        acc[4 * g] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const bf16x8 wa = *reinterpret_cast<const bf16x8*>(wb + s * 1024);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xf[s], acc, 0, 0, 0);
    }

    bf16x8 hp[2];
    if (MODE == kBwd) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float hv = (float)hin[i >> 3][i & 7];
        hp[i >> 3][i & 7] = (__bf16)(hv != 0.f ? acc[i] * a.scale : 0.f);
      }
    } else if (MODE == kFwdTrain && a.thresh16) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const u32x4 rnd = philox8((unsigned long long)(hoff >> 3) + half, a.seed, offset);
        const unsigned w[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned field = (w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
          const float v = fmaxf(acc[8 * half + k], 0.f);
          hp[half][k] = (__bf16)(field < a.thresh16 ? 0.f : v * a.scale);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) hp[i >> 3][i & 7] = (__bf16)fmaxf(acc[i], 0.f);
    }
    if (MODE != kFwdEval && live) {
      *reinterpret_cast<bf16x8*>(a.h_out + hoff) = hp[0];
      *reinterpret_cast<bf16x8*>(a.h_out + hoff + 8) = hp[1];
    }

#pragma unroll
    for (int ot = 0; ot < 8; ++ot)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 wa = *reinterpret_cast<const bf16x8*>(wb + 16384 + (ot * 2 + s) * 1024);
        yacc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, hp[s], yacc[ot], 0, 0, 0);
      }
    __syncthreads();
  }

  if (live) {
    __bf16* orow = a.out + tok * kD + 16 * h;
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
      bf16x8 o[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = yacc[ot][i];
        if (MODE != kBwd) v += (float)a.b2[32 * ot + 16 * h + i];
        o[i >> 3][i & 7] = (__bf16)v;
      }
      *reinterpret_cast<bf16x8*>(orow + 32 * ot) = o[0];
      *reinterpret_cast<bf16x8*>(orow + 32 * ot + 8) = o[1];
    }
  }
}

int ffn_waves() {
  static const int w = [] {
    const char* e = getenv("DSKD_FFN_WAVES");
    const int v = e ? atoi(e) : 4;
    return v == 8 ? 8 : 4;
  }();
  return w;
}

template <int MODE>
int launch_ffn(const FfnArgs& a, hipStream_t st) {
  const size_t lds = 2 * kTileBytes + kF * sizeof(float);
  if (ffn_waves() == 8) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel<MODE, 8>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return fail(DSKD_ERR_LAUNCH, "dskd_ffn: LDS attribute: %s", hipGetErrorString(attr));
    const long long grid = (a.T + 255) / 256;
    hipLaunchKernelGGL((ffn_fused_kernel<MODE, 8>), dim3((unsigned)grid), dim3(512), lds, st, a);
  } else {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel<MODE, 4>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return fail(DSKD_ERR_LAUNCH, "dskd_ffn: LDS attribute: %s", hipGetErrorString(attr));
    const long long grid = (a.T + 127) / 128;
    hipLaunchKernelGGL((ffn_fused_kernel<MODE, 4>), dim3((unsigned)grid), dim3(256), lds, st, a);
  }
  return check_launch("dskd_ffn");
}

bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int64_t dskd_ffn_packed_bytes(int d_model, int hidden) {
  if (d_model != kD || hidden != kF) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn: d_model 256 / hidden 1024 only (got %d / %d)", d_model, hidden) < 0 ? -1 : -1;
  return (int64_t)kTiles * kTileBytes;
}

extern "C" int dskd_ffn_pack(const void* w1, const void* w2, void* packed_fwd, void* packed_bwd, int d_model,
                             int hidden, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_pack: bf16 only");
  if (d_model != kD || hidden != kF)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_pack: d_model 256 / hidden 1024 only (got %d / %d)", d_model, hidden);
  if (!w1 || !w2 || (!packed_fwd && !packed_bwd)) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_pack: null pointer");
  if (misaligned(w1) || misaligned(w2) || misaligned(packed_fwd) || misaligned(packed_bwd))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_pack: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(kTiles * kFragsPerTile / 256, 2), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)w1, (const __bf16*)w2, (__bf16*)packed_fwd, (__bf16*)packed_bwd);
  return check_launch("dskd_ffn_pack");
}

extern "C" int dskd_ffn_fwd(const void* x, const void* packed_fwd, const void* b1, const void* b2, void* h_out, void* y,
                            int64_t tokens, int d_model, int hidden, float p, uint64_t seed, uint64_t offset,
                            const uint64_t* epoch, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_fwd: bf16 only");
  if (d_model != kD || hidden != kF)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_fwd: d_model 256 / hidden 1024 only (got %d / %d)", d_model, hidden);
  if (!x || !packed_fwd || !b1 || !b2 || !y || tokens < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_fwd: null pointer or negative token count");
  if (misaligned(x) || misaligned(packed_fwd) || misaligned(h_out) || misaligned(y))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_fwd: pointers must be 16-byte aligned");
  if (!(p >= 0.f) || p >= 1.f) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_fwd: p=%f", p);
  if (p > 0.f && !h_out) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_fwd: dropout needs h_out (the mask is read off H by the backward)");
  if (tokens == 0) return DSKD_OK;
  FfnArgs a{};
  a.in = (const __bf16*)x; a.wp = (const __bf16*)packed_fwd; a.b1 = (const __bf16*)b1; a.b2 = (const __bf16*)b2;
  a.h_out = (__bf16*)h_out; a.out = (__bf16*)y; a.T = tokens;
  a.scale = 1.0f / (1.0f - p);
  const unsigned t = (unsigned)((double)p * 65536.0 + 0.5);
  a.thresh16 = p > 0.f ? (t < 1 ? 1u : t) : 0u;
  a.seed = seed; a.offset = offset; a.epoch = reinterpret_cast<const unsigned long long*>(epoch);
  return h_out ? launch_ffn<kFwdTrain>(a, (hipStream_t)stream) : launch_ffn<kFwdEval>(a, (hipStream_t)stream);
}

extern "C" int dskd_ffn_bwd(const void* grad_y, const void* h, const void* packed_bwd, void* grad_h, void* grad_x,
                            int64_t tokens, int d_model, int hidden, float p, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: bf16 only");
  if (d_model != kD || hidden != kF)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: d_model 256 / hidden 1024 only (got %d / %d)", d_model, hidden);
  if (!grad_y || !h || !packed_bwd || !grad_h || !grad_x || tokens < 0)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: null pointer or negative token count");
  if (misaligned(grad_y) || misaligned(h) || misaligned(packed_bwd) || misaligned(grad_h) || misaligned(grad_x))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: pointers must be 16-byte aligned");
  if (!(p >= 0.f) || p >= 1.f) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: p=%f", p);
  if (tokens == 0) return DSKD_OK;
  FfnArgs a{};
  a.in = (const __bf16*)grad_y; a.wp = (const __bf16*)packed_bwd; a.h_in = (const __bf16*)h;
  a.h_out = (__bf16*)grad_h; a.out = (__bf16*)grad_x; a.T = tokens; a.scale = 1.0f / (1.0f - p);
  return launch_ffn<kBwd>(a, (hipStream_t)stream);
}
