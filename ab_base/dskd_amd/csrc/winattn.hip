// Window attention of the Swin backbone (BASELINE.json configs[3], "MFMA window-attn path"), forward and backward, as
// hand-written MFMA kernels for gfx950.
//
// Replaces WindowMSA.forward of the reference (mmdet/models/backbones/swin.py:81-126: q k^T * scale + relative position
// bias (+ the shifted-window mask of ShiftWindowMSA, :180-286) -> softmax -> attn @ v) between its qkv Linear and its
// output projection; windows of 7 x 7 = 49 tokens, head dimension 32.  One WAVE owns one (window, head):
//   S^T = K Q^T            weights K as the A operand, queries as the B operand -> a lane holds ONE query and, in its
//                          accumulator registers, that query's scores against 64 (49 + padding) keys: the softmax is
//                          lane-local plus one exchange between the two lane halves (cdna_hip_programming.md T12)
//   O^T = V^T P^T          the converted accumulators ARE the B operand (guide section 3, "An accumulator tile as the next
//                          MFMA's operand"); V^T comes out of the row-major V image with ds_read_b64_tr_b16 (T10)
// so the 49 x 49 score matrix never leaves registers.  q / k / v are read straight from the qkv projection's output
// [windows, 49, 3, heads, 32] and the result is written as [windows, 49, heads * 32] (what the projection reads): none of
// the permute / contiguous copies around F.scaled_dot_product_attention remain.  The additive term (bias + mask) comes as
// a table [mask types][heads][64 keys][64 queries] (f32, -30000 on the padded keys): a shifted layer has at most four
// distinct masks (interior, last row, last column, corner of the window grid), an unshifted one none.
//
// Backward (same wave ownership, P recomputed): dP^T = V dO^T, dS^T = P^T o (dP^T - delta), dQ^T = K^T dS^T from registers
// as above; dV^T = dO^T P and dK^T = Q^T dS sum over the QUERY index, i.e. over lanes: P^T and dS^T are written to LDS as
// bf16 [query][key] images and read back transposed (ds_read_b64_tr_b16).  The bias gradient is accumulated in registers
// over all the windows a wave handles (a wave keeps one head) and leaves as one atomic add per entry and wave.
#include "common.h"

namespace dskd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kN = 49;          // tokens per window
constexpr int kD = 32;          // head dimension
constexpr int kNP = 64;         // padded tokens
constexpr int kImg = kNP * kD * 2;      // bytes of one [64][32] bf16 image

__device__ __forceinline__ int pi_row(int r) { return (r & 3) + 4 * (r >> 3) + 16 * ((r >> 2) & 1); }

// byte offset of element (row, col) of a [64][32] bf16 image whose 16-byte chunks are swizzled for conflict-free
// ds_read_b128 operand reads (chunk c of row r at position c ^ ((r >> 2) & 3), as in gemm_nt.hip)
__device__ __forceinline__ int img_off(int row, int col) {
  return row * 64 + ((((col >> 3) ^ ((row >> 2) & 3))) << 4) + (col & 7) * 2;
}

__device__ __forceinline__ bf16x8 lds_read16(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x4 lds_read_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// rows [0, 49) of one of q / k / v / dO (row stride `ld` elements, 32 elements each) -> swizzled LDS image; rows 49..63 = 0
__device__ __forceinline__ void load_image(char* img, const __bf16* src, long long ld, int lane) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = it * 64 + lane;             // 16-byte chunk: row c >> 2, part c & 3
    const int row = c >> 2, part = c & 3;
    bf16x8 v = {};
    if (row < kN) v = *reinterpret_cast<const bf16x8*>(src + (long long)row * ld + part * 8);
    *reinterpret_cast<bf16x8*>(img + row * 64 + ((part ^ ((row >> 2) & 3)) << 4)) = v;
  }
}

// operand of a product that sums over the 32 channels: row `row` of an image, k-step s (16 channels), lane half h
__device__ __forceinline__ bf16x8 row_frag(const char* img, int row, int s, int h) {
  return lds_read16(img + row * 64 + (((2 * s + h) ^ ((row >> 2) & 3)) << 4));
}

// A operand X^T[d slot][k = token] of a product that sums over TOKENS (rows of the image `img` [token][d]), through the
// transposing read.  Slot r of the operand carries channel pi(r), so that the product's accumulator registers are 16
// consecutive channels.  acc_order: the token order inside the k-step is that of an accumulator tile used as the other
// operand (element j of lane half h = token 16 s + 8 (j >> 2) + 4 h + (j & 3)); otherwise natural (16 s + 8 h + j).
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int tok0, int s, int lane, bool acc_order) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = lane >> 5;
  const int d0 = 8 * (g & 1) + 4 * (p >> 1) + 16 * (p & 1);            // pi(16 (g & 1) + 4 p)
  const int t0 = tok0 + 16 * s + (acc_order ? 4 * h : 8 * h) + q;
  const int t1 = t0 + (acc_order ? 8 : 4);
  return cat4(lds_read_tr(img + img_off(t0, d0)), lds_read_tr(img + img_off(t1, d0)));
}

// accumulator tile (32 keys x 32 queries, f32) -> the two bf16 B operands of the following product (k-steps 0, 1)
__device__ __forceinline__ void acc_to_frags(const f32x16& a, bf16x8& f0, bf16x8& f1) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { f0[i] = (__bf16)a[i]; f1[i] = (__bf16)a[8 + i]; }
}

struct WinArgs {
  const __bf16* qkv;     // [windows, 49, 3, heads, 32]
  const float* table;    // [types][heads][64 keys][64 queries] additive term
  const int* wtype;      // [nW] mask type of window (index within the image), or null (type 0)
  __bf16* out;           // forward: [windows, 49, heads * 32]
  const __bf16* dout;    // backward: same layout
  __bf16* dqkv;          // backward: [windows, 49, 3, heads, 32]
  float* dtable;         // backward: [heads][64 keys][64 queries], += (zeroed by the caller)
  int windows, heads, nW;
  float scale;
};

// S^T = scale * K Q^T + table, then P^T = softmax over the keys (registers of a lane + the other lane half)
// st[kt][qt]: 32 keys (tile kt) x 32 queries (tile qt)
__device__ __forceinline__ void scores_softmax(const char* qs, const char* ks, const float* tab, float scale, int lane,
                                               f32x16 (&st)[2][2]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(ks, 32 * kt + r, s, h), row_frag(qs, 32 * qt + r, s, h), acc, 0, 0, 0);
      st[kt][qt] = acc;
    }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int query = 32 * qt + r;
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h;
        const float v = fmaf(st[kt][qt][i], scale, tab[key * kNP + query]);
        st[kt][qt][i] = v;
        m = fmaxf(m, v);
      }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e = __expf(st[kt][qt][i] - m);
        st[kt][qt][i] = e;
        l += e;
      }
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) st[kt][qt][i] *= inv;
  }
}

// X^T[d][query] = sum over keys of (image [key][d])^T times the accumulator tiles at[kt][qt] (keys x queries)
__device__ __forceinline__ void keys_product(const char* img, const f32x16 (&at)[2][2], int lane, f32x16 (&o)[2]) {
  bf16x8 vf[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) vf[kt][s] = tr_frag(img, 32 * kt, s, lane, true);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      bf16x8 p0, p1;
      acc_to_frags(at[kt][qt], p0, p1);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt][0], p0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt][1], p1, acc, 0, 0, 0);
    }
    o[qt] = acc;
  }
}

// a lane's 16 consecutive channels (16 h ..) of token `tok` -> dst row (32 elements per head), scaled
__device__ __forceinline__ void store_row16(__bf16* dst, const f32x16& a, float mul) {
  bf16x8 o0, o1;
#pragma unroll
  for (int i = 0; i < 8; ++i) { o0[i] = (__bf16)(a[i] * mul); o1[i] = (__bf16)(a[8 + i] * mul); }
  *reinterpret_cast<bf16x8*>(dst) = o0;
  *reinterpret_cast<bf16x8*>(dst + 8) = o1;
}

constexpr int kFwdWaves = 4;

__global__ __launch_bounds__(kFwdWaves * 64) void winattn_fwd_kernel(const WinArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* qs = smem + wave * 3 * kImg;
  char* ks = qs + kImg;
  char* vs = ks + kImg;
  const long long task = (long long)blockIdx.x * kFwdWaves + wave;       // (window, head), head fastest
  if (task >= (long long)a.windows * a.heads) return;                    // whole waves only: EXEC stays full below
  const int w = (int)(task / a.heads), hd = (int)(task - (long long)w * a.heads);
  const long long ld = 3LL * a.heads * kD;
  const __bf16* base = a.qkv + (long long)w * kN * ld + hd * kD;
  load_image(qs, base, ld, lane);
  load_image(ks, base + a.heads * kD, ld, lane);
  load_image(vs, base + 2 * a.heads * kD, ld, lane);
  const int type = a.wtype ? a.wtype[w % a.nW] : 0;
  const float* tab = a.table + ((long long)type * a.heads + hd) * (kNP * kNP);
  wave_lds_sync();

  f32x16 st[2][2];
  scores_softmax(qs, ks, tab, a.scale, lane, st);
  f32x16 o[2];
  keys_product(vs, st, lane, o);
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int query = 32 * qt + r;
    if (query < kN) store_row16(a.out + ((long long)w * kN + query) * (a.heads * kD) + hd * kD + 16 * h, o[qt], 1.f);
  }
}

// B operand X[k = query][col = key] of a product that sums over QUERIES, read (transposed) from a [64 query][64 key] bf16
// image with 128-byte rows: k-step s4 of 4 (16 queries, natural order 16 s4 + 8 h + j), key tile kt
__device__ __forceinline__ bf16x8 tr_frag_qk(const char* img, int s4, int kt, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = lane >> 5;
  const int key0 = 32 * kt + 16 * (g & 1) + 4 * p;
  const int t0 = 16 * s4 + 8 * h + q;
  return cat4(lds_read_tr(img + t0 * 128 + key0 * 2), lds_read_tr(img + (t0 + 4) * 128 + key0 * 2));
}

// accumulator tiles (keys x queries; the lane's query fixed) -> bf16 image [query][key]: 4 consecutive keys per store
__device__ __forceinline__ void tiles_to_image(char* img, const f32x16 (&t)[2][2], int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const bf16x4 v = {(__bf16)t[kt][qt][4 * rg], (__bf16)t[kt][qt][4 * rg + 1], (__bf16)t[kt][qt][4 * rg + 2],
                          (__bf16)t[kt][qt][4 * rg + 3]};
        *reinterpret_cast<bf16x4*>(img + (32 * qt + r) * 128 + (32 * kt + 8 * rg + 4 * h) * 2) = v;
      }
}

constexpr int kBwdWaves = 2;
constexpr int kBwdLdsPerWave = 4 * kImg + 2 * (kNP * kNP * 2);      // q, k, v, dO images + P and dS images: 32 KB

__global__ __launch_bounds__(kBwdWaves * 64) void winattn_bwd_kernel(const WinArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* qs = smem + wave * kBwdLdsPerWave;
  char* ks = qs + kImg;
  char* vs = ks + kImg;
  char* dos = vs + kImg;
  char* pimg = dos + kImg;
  char* dsimg = pimg + kNP * kNP * 2;
  const int hd = blockIdx.y;
  const int r = lane & 31, h = lane >> 5;
  const long long ld = 3LL * a.heads * kD;
  f32x16 dbacc[2][2];                       // bias gradient of this wave's windows (keys x queries, as S^T)
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int i = 0; i < 16; ++i) dbacc[kt][qt][i] = 0.f;

  for (int w = blockIdx.x * kBwdWaves + wave; w < a.windows; w += gridDim.x * kBwdWaves) {      // wave-uniform: EXEC full
    const __bf16* base = a.qkv + (long long)w * kN * ld + hd * kD;
    load_image(qs, base, ld, lane);
    load_image(ks, base + a.heads * kD, ld, lane);
    load_image(vs, base + 2 * a.heads * kD, ld, lane);
    load_image(dos, a.dout + (long long)w * kN * (a.heads * kD) + hd * kD, (long long)a.heads * kD, lane);
    const int type = a.wtype ? a.wtype[w % a.nW] : 0;
    const float* tab = a.table + ((long long)type * a.heads + hd) * (kNP * kNP);
    wave_lds_sync();

    f32x16 st[2][2], dpt[2][2];
    scores_softmax(qs, ks, tab, a.scale, lane, st);                   // P^T
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {                                // dP^T = V dO^T
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(vs, 32 * kt + r, s, h), row_frag(dos, 32 * qt + r, s, h), acc, 0, 0, 0);
        dpt[kt][qt] = acc;
      }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {                                  // dS^T = P^T o (dP^T - sum_k P dP)
      float dl = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dl = fmaf(st[kt][qt][i], dpt[kt][qt][i], dl);
      dl += __shfl_xor(dl, 32);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float ds = st[kt][qt][i] * (dpt[kt][qt][i] - dl);
          dpt[kt][qt][i] = ds;
          dbacc[kt][qt][i] += ds;
        }
    }
    tiles_to_image(pimg, st, lane);
    tiles_to_image(dsimg, dpt, lane);
    {                                                                 // dQ^T = K^T dS^T (from registers)
      f32x16 o[2];
      keys_product(ks, dpt, lane, o);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int query = 32 * qt + r;
        if (query < kN) store_row16(a.dqkv + ((long long)w * kN + query) * ld + hd * kD + 16 * h, o[qt], a.scale);
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int which = 0; which < 2; ++which) {                         // dV^T = dO^T P ; dK^T = Q^T dS (sums over queries)
      const char* aimg = which == 0 ? dos : qs;
      const char* bimg = which == 0 ? pimg : dsimg;
      bf16x8 af[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) af[s4] = tr_frag(aimg, 0, s4, lane, false);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s4], tr_frag_qk(bimg, s4, kt, lane), acc, 0, 0, 0);
        const int key = 32 * kt + r;
        if (key < kN)
          store_row16(a.dqkv + ((long long)w * kN + key) * ld + (which == 0 ? 2 : 1) * a.heads * kD + hd * kD + 16 * h, acc,
                      which == 0 ? 1.f : a.scale);
      }
    }
    wave_lds_sync();      // the next window overwrites the images
  }
  float* dt = a.dtable + (long long)hd * (kNP * kNP);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * h, query = 32 * qt + r;
        if (key < kN && query < kN) atomicAdd(dt + key * kNP + query, dbacc[kt][qt][i]);
      }
}

}  // namespace
}  // namespace dskd

using namespace dskd;

static int winattn_check(const char* who, const void* qkv, const void* table, int windows, int heads, int nW, int dtype) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "%s: bf16 only", who);
  if (!qkv || !table) return fail(DSKD_ERR_INVALID_ARG, "%s: null pointer", who);
  if (windows < 0 || heads < 1 || nW < 1 || (windows % nW) != 0)
    return fail(DSKD_ERR_INVALID_ARG, "%s: bad sizes (windows=%d heads=%d windows per image=%d)", who, windows, heads, nW);
  if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(table) & 15))
    return fail(DSKD_ERR_INVALID_ARG, "%s: pointers must be 16-byte aligned", who);
  return DSKD_OK;
}

extern "C" int dskd_winattn_fwd(const void* qkv, const float* table, const int32_t* wtype, void* out, int windows, int heads,
                                int nW, int tokens, int head_dim, float scale, int dtype, void* stream) {
  if (tokens != kN || head_dim != kD)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_winattn_fwd: built for 49-token windows and head dimension 32 (got %d, %d)", tokens, head_dim);
  if (int rc = winattn_check("dskd_winattn_fwd", qkv, table, windows, heads, nW, dtype)) return rc;
  if (!out || (reinterpret_cast<uintptr_t>(out) & 15)) return fail(DSKD_ERR_INVALID_ARG, "dskd_winattn_fwd: bad output pointer");
  if (windows == 0) return DSKD_OK;
  WinArgs a = {};
  a.qkv = (const __bf16*)qkv; a.table = table; a.wtype = wtype; a.out = (__bf16*)out;
  a.windows = windows; a.heads = heads; a.nW = nW; a.scale = scale;
  const long long tasks = (long long)windows * heads;
  const size_t lds = (size_t)kFwdWaves * 3 * kImg;
  hipLaunchKernelGGL(winattn_fwd_kernel, dim3((unsigned)((tasks + kFwdWaves - 1) / kFwdWaves)), dim3(kFwdWaves * 64), lds,
                     (hipStream_t)stream, a);
  return check_launch("dskd_winattn_fwd");
}

extern "C" int dskd_winattn_bwd(const void* qkv, const float* table, const int32_t* wtype, const void* dout, void* dqkv,
                                float* dtable, int windows, int heads, int nW, int tokens, int head_dim, float scale,
                                int dtype, void* stream) {
  if (tokens != kN || head_dim != kD)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_winattn_bwd: built for 49-token windows and head dimension 32 (got %d, %d)", tokens, head_dim);
  if (int rc = winattn_check("dskd_winattn_bwd", qkv, table, windows, heads, nW, dtype)) return rc;
  if (!dout || !dqkv || !dtable || (reinterpret_cast<uintptr_t>(dout) & 15) || (reinterpret_cast<uintptr_t>(dqkv) & 15))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_winattn_bwd: null or misaligned pointer");
  if (windows == 0) return DSKD_OK;
  WinArgs a = {};
  a.qkv = (const __bf16*)qkv; a.table = table; a.wtype = wtype; a.dout = (const __bf16*)dout; a.dqkv = (__bf16*)dqkv;
  a.dtable = dtable; a.windows = windows; a.heads = heads; a.nW = nW; a.scale = scale;
  constexpr int lds = kBwdWaves * kBwdLdsPerWave;
  int dev = 0;
  static bool done[64] = {};
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!done[dev]) {
    if (hipFuncSetAttribute((const void*)winattn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return fail(DSKD_ERR_LAUNCH, "dskd_winattn_bwd: cannot reserve %d bytes of LDS", lds);
    done[dev] = true;
  }
  // a wave keeps one head and walks several windows (its bias gradient stays in registers): ~2 000 waves in all
  int gx = (windows + kBwdWaves - 1) / kBwdWaves;
  const int cap = (1024 + heads - 1) / heads;
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(winattn_bwd_kernel, dim3((unsigned)gx, (unsigned)heads), dim3(kBwdWaves * 64), lds, (hipStream_t)stream, a);
  return check_launch("dskd_winattn_bwd");
}
