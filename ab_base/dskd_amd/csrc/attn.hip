// Multi-head self-attention of the decoder's 300 object queries, forward and backward, as hand-written MFMA kernels (gfx950).
//
// Replaces the core of ext-mmcv `MultiheadAttention` (an nn.MultiheadAttention with 8 heads of 32 channels, attention dropout
// 0.1; configured at configs/deformable_detr/*_il.py:82-87 and run as the first sub-layer of every decoder layer by
// DeformableDetrTransformerDecoder.forward, mmdet/models/utils/transformer.py:639-709) between its input projection and its
// output projection: softmax(q k^T / sqrt(32)) -> dropout -> @ v per (image, head).  PyTorch ran it as
// scaled_dot_product_attention -> AOTriton attn_fwd / bwd_preprocess / bwd_kernel_dq / bwd_kernel_dk_dv (18 + 22 + 15 us per
// layer on the decoder's serial launch chain) plus the layout copies around them.
//
// Shapes: L = 300 queries = keys (up to 320), head dimension 32.  One WAVE owns 32 queries (forward, dQ) or 32 keys (dK, dV)
// of one (image, head) and walks the other axis in tiles of 32:
//   forward, per wave (b, h, query tile):   S^T = K Q^T   (keys x queries: a lane holds ONE query, its scores in registers)
//                                           P^T = softmax over the keys (registers + one exchange between the lane halves),
//                                           dropout mask from a counter hash of (b, h, query, key)
//                                           O^T = V^T P^T (the converted accumulators ARE the B operand; V^T out of the
//                                           row-major V image with ds_read_b64_tr_b16, as in winattn.hip)
//   backward, query-tile waves:             P^T recomputed from the saved row statistics, dP^T = V dO^T,
//                                           dS^T = P^T o (mask dP^T / (1 - p) - delta), dQ^T = K^T dS^T from registers
//   backward, key-tile waves:               the same tiles in the other orientation (S = Q K^T: queries x keys, a lane holds
//                                           ONE key), so that dV^T = dO^T P and dK^T = Q^T dS sum over the accumulator ROWS:
//                                           no transposes through LDS, no atomics, no reduction between waves.
// delta[q] = sum_d dO[q, d] O[q, d] comes from a small pre-pass.  q / k are read in place from the joint projection output
// [.., 2 E] and v from [.., E]; the result is written [.., E] as the output projection reads it, the gradients straight
// into d(qk) [.., 2 E] and d(v): none of the permute / split / cat copies around SDPA remain.
//
// Dropout: element (b, h, q, k) is dropped when the upper 16 bits of lowbias32(index * 0x9E3779B1 + key) fall below
// p * 2^16 -- a counter hash instead of the Philox stream of the other kernels, because the two backward orientations visit
// the elements in different groupings (a lane needs single elements, not runs of 8); key = f(seed, offset + *epoch) as in
// dskd_dropout_fwd, so a hipGraph replay draws new masks.
#include "common.h"

namespace dskd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = 32;            // head dimension
constexpr int kMaxT = 10;         // tiles of 32 queries / keys: L <= 320
constexpr int kTileImg = 32 * kD * 2;      // bytes of one [32][32] bf16 image

// byte offset of element (row, col) of a [rows][32] bf16 image whose 16-byte chunks are swizzled for conflict-free operand
// reads (chunk c of row r at position c ^ ((r >> 2) & 3), as in gemm_nt.hip / winattn.hip)
__device__ __forceinline__ int img_off(int row, int col) {
  return row * 64 + ((((col >> 3) ^ ((row >> 2) & 3))) << 4) + (col & 7) * 2;
}
__device__ __forceinline__ bf16x8 lds_read16(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x4 lds_read_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) { return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// rows [row0, row0 + 32) of a [L][32]-per-head operand (row stride `rs` elements) -> swizzled 32-row LDS image; rows >= L: 0
__device__ __forceinline__ void load_tile(char* img, const __bf16* src, long long rs, int row0, int L, int lane) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = it * 64 + lane;             // 16-byte chunk: row c >> 2, part c & 3
    const int row = c >> 2, part = c & 3;
    bf16x8 v = {};
    if (row0 + row < L) v = *reinterpret_cast<const bf16x8*>(src + (long long)(row0 + row) * rs + part * 8);
    *reinterpret_cast<bf16x8*>(img + row * 64 + ((part ^ ((row >> 2) & 3)) << 4)) = v;
  }
}
// operand of a product that sums over the 32 channels: row `row` of a tile image, k-step s (16 channels), lane half h
__device__ __forceinline__ bf16x8 row_frag(const char* img, int row, int s, int h) {
  return lds_read16(img + row * 64 + (((2 * s + h) ^ ((row >> 2) & 3)) << 4));
}
// the same operand straight from memory: 16 bytes of row `p` (already offset to the head's 32 channels)
__device__ __forceinline__ bf16x8 row_frag_g(const __bf16* p, int s, int h) {
  return *reinterpret_cast<const bf16x8*>(p + 16 * s + 8 * h);
}
// A operand X^T[d slot][k = token] of a product that sums over the 32 TOKENS of a tile image [token][d], through the
// transposing read; slot r carries channel pi(r) so that the product's 16 accumulator registers are 16 consecutive channels;
// the token order inside the k-step is that of an accumulator tile used as the other operand (element j of lane half h =
// token 16 s + 8 (j >> 2) + 4 h + (j & 3)): cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand".
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int s, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = lane >> 5;
  const int d0 = 8 * (g & 1) + 4 * (p >> 1) + 16 * (p & 1);            // pi(16 (g & 1) + 4 p)
  const int t0 = 16 * s + 4 * h + q;
  return cat4(lds_read_tr(img + img_off(t0, d0)), lds_read_tr(img + img_off(t0 + 8, d0)));
}
__device__ __forceinline__ void acc_to_frags(const f32x16& a, bf16x8& f0, bf16x8& f1) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { f0[i] = (__bf16)a[i]; f1[i] = (__bf16)a[8 + i]; }
}
__device__ __forceinline__ void store_row16(__bf16* dst, const f32x16& a, float mul) {
  bf16x8 o0, o1;
#pragma unroll
  for (int i = 0; i < 8; ++i) { o0[i] = (__bf16)(a[i] * mul); o1[i] = (__bf16)(a[8 + i] * mul); }
  *reinterpret_cast<bf16x8*>(dst) = o0;
  *reinterpret_cast<bf16x8*>(dst + 8) = o1;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ unsigned lowbias32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

struct AttnArgs {
  const __bf16 *q, *k, *v;       // heads side by side in the row: channel = head * 32 + d
  __bf16* out;                   // forward
  const __bf16 *o, *dout;        // backward: the forward's output and its gradient (layout of out)
  __bf16 *dq, *dk, *dv;          // backward (layouts of q / k / v)
  float* stats;                  // [B, H, L, 2]: row maximum (of the scaled scores), 1 / row sum -- forward writes, backward reads
  float* delta;                  // [B, H, L]: backward pre-pass writes, backward reads
  long long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;      // batch / row strides in elements (dq / dk / dv / dout: same)
  int B, H, L, T;                // T = ceil(L / 32)
  float scale, drop_scale;
  unsigned thresh;               // drop when (hash >> 16) < thresh; 0: no dropout
  unsigned long long seed, offset;
  const unsigned long long* epoch;
};

__device__ __forceinline__ unsigned drop_key(const AttnArgs& a) {
  const unsigned long long off = a.offset + (a.epoch ? *a.epoch : 0ull);
  return lowbias32((unsigned)a.seed ^ lowbias32((unsigned)off ^ 0x85ebca6bu)) ^ lowbias32((unsigned)(off >> 32) + (unsigned)(a.seed >> 32));
}
// element index of (query, key) of the (image, head) whose first element is `base`
__device__ __forceinline__ bool kept(unsigned base, int query, int key, unsigned dkey, unsigned thresh) {
  return (lowbias32((base + (unsigned)query * 512u + (unsigned)key) * 0x9E3779B1u + dkey) >> 16) >= thresh;
}

constexpr int kWaves = 4;

// ---------------------------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(kWaves * 64) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long task = (long long)blockIdx.x * kWaves + wave;       // (image, head, query tile), tile fastest
  if (task >= (long long)a.B * a.H * a.T) return;                     // whole waves only: EXEC stays full below
  const int qt = (int)(task % a.T);
  const int bh = (int)(task / a.T), hd = bh % a.H, b = bh / a.H;
  const int r = lane & 31, hh = lane >> 5;
  char* vimg = smem + wave * (kMaxT * kTileImg);
  const __bf16* vb = a.v + (long long)b * a.v_bs + hd * kD;
  const __bf16* kb = a.k + (long long)b * a.k_bs + hd * kD;
#pragma unroll
  for (int kt = 0; kt < kMaxT; ++kt)
    if (kt < a.T) load_tile(vimg + kt * kTileImg, vb, a.v_rs, 32 * kt, a.L, lane);
  const int query = 32 * qt + r;
  const int qi = query < a.L ? query : a.L - 1;
  const __bf16* qp = a.q + (long long)b * a.q_bs + (long long)qi * a.q_rs + hd * kD;
  const bf16x8 qf0 = row_frag_g(qp, 0, hh), qf1 = row_frag_g(qp, 1, hh);
  wave_lds_sync();

  f32x16 st[kMaxT];
#pragma unroll
  for (int kt = 0; kt < kMaxT; ++kt) {
    st[kt] = zero16();
    if (kt < a.T) {
      const int ki = 32 * kt + r;
      const __bf16* kp = kb + (long long)(ki < a.L ? ki : a.L - 1) * a.k_rs;
      st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_g(kp, 0, hh), qf0, st[kt], 0, 0, 0);
      st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_g(kp, 1, hh), qf1, st[kt], 0, 0, 0);
    }
  }
  float m = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < kMaxT; ++kt)
    if (kt < a.T) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * hh;
        const float v = key < a.L ? st[kt][i] * a.scale : -3.0e38f;
        st[kt][i] = v;
        m = fmaxf(m, v);
      }
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < kMaxT; ++kt)
    if (kt < a.T) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * hh;
        const float e = key < a.L ? __expf(st[kt][i] - m) : 0.f;
        st[kt][i] = e;
        l += e;
      }
    }
  l += __shfl_xor(l, 32);
  const float inv = 1.f / l;
  if (a.stats && hh == 0 && query < a.L) {
    float* sp = a.stats + ((long long)bh * a.L + query) * 2;
    sp[0] = m; sp[1] = inv;
  }
  const unsigned dkey = a.thresh ? drop_key(a) : 0u;
  const unsigned ebase = (unsigned)bh * (unsigned)a.L * 512u;
  const float keep_mul = inv * a.drop_scale;
  f32x16 o = zero16();
#pragma unroll
  for (int kt = 0; kt < kMaxT; ++kt)
    if (kt < a.T) {
      if (a.thresh) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * hh;
          st[kt][i] = kept(ebase, query, key, dkey, a.thresh) ? st[kt][i] * keep_mul : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) st[kt][i] *= inv;
      }
      bf16x8 p0, p1;
      acc_to_frags(st[kt], p0, p1);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vimg + kt * kTileImg, 0, lane), p0, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vimg + kt * kTileImg, 1, lane), p1, o, 0, 0, 0);
    }
  if (query < a.L) store_row16(a.out + (long long)b * a.o_bs + (long long)query * a.o_rs + hd * kD + 16 * hh, o, 1.f);
}

// ------------------------------------------------------------------------------------------- backward: delta = sum_d dO O
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // (image, head, query), query fastest
  if (i >= (long long)a.B * a.H * a.L) return;
  const int query = (int)(i % a.L);
  const int bh = (int)(i / a.L), hd = bh % a.H, b = bh / a.H;
  const long long off = (long long)b * a.o_bs + (long long)query * a.o_rs + hd * kD;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(a.o + off + 8 * c), y = *reinterpret_cast<const bf16x8*>(a.dout + off + 8 * c);
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf((float)x[j], (float)y[j], s);
  }
  a.delta[i] = s;
}

// ---------------------------------------------------------------------------------------------------------------- backward
// tasks [0, B H T): key tile w -> dK, dV;  [B H T, 2 B H T): query tile -> dQ
__global__ __launch_bounds__(kWaves * 64) void attn_bwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long nt = (long long)a.B * a.H * a.T;
  long long task = (long long)blockIdx.x * kWaves + wave;
  if (task >= 2 * nt) return;
  const bool qside = task >= nt;
  if (qside) task -= nt;
  const int tile = (int)(task % a.T);
  const int bh = (int)(task / a.T), hd = bh % a.H, b = bh / a.H;
  const int r = lane & 31, hh = lane >> 5;
  char* img = smem + wave * (kMaxT * kTileImg);
  const __bf16* qb = a.q + (long long)b * a.q_bs + hd * kD;
  const __bf16* kb = a.k + (long long)b * a.k_bs + hd * kD;
  const __bf16* vb = a.v + (long long)b * a.v_bs + hd * kD;
  const __bf16* dob = a.dout + (long long)b * a.o_bs + hd * kD;
  const float* stb = a.stats + (long long)bh * a.L * 2;
  const float* dlb = a.delta + (long long)bh * a.L;
  const unsigned dkey = a.thresh ? drop_key(a) : 0u;
  const unsigned ebase = (unsigned)bh * (unsigned)a.L * 512u;

  if (qside) {
    // ================================================================= 32 queries: dQ^T = K^T dS^T, keys in tiles
#pragma unroll
    for (int kt = 0; kt < kMaxT; ++kt)
      if (kt < a.T) load_tile(img + kt * kTileImg, kb, a.k_rs, 32 * kt, a.L, lane);
    const int query = 32 * tile + r;
    const int qi = query < a.L ? query : a.L - 1;
    const __bf16* qp = qb + (long long)qi * a.q_rs;
    const __bf16* dop = dob + (long long)qi * a.o_rs;
    const bf16x8 qf0 = row_frag_g(qp, 0, hh), qf1 = row_frag_g(qp, 1, hh);
    const bf16x8 df0 = row_frag_g(dop, 0, hh), df1 = row_frag_g(dop, 1, hh);
    const float m = stb[qi * 2], inv = stb[qi * 2 + 1], dl = dlb[qi];
    wave_lds_sync();
    f32x16 dq = zero16();
#pragma unroll
    for (int kt = 0; kt < kMaxT; ++kt)
      if (kt < a.T) {
        const int ki = 32 * kt + r;
        const long long kr = ki < a.L ? ki : a.L - 1;
        const __bf16* kp = kb + kr * a.k_rs;
        const __bf16* vp = vb + kr * a.v_rs;
        f32x16 s = zero16(), dp = zero16();
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_g(kp, 0, hh), qf0, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_g(kp, 1, hh), qf1, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_g(vp, 0, hh), df0, dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_g(vp, 1, hh), df1, dp, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * hh;
          const float p = key < a.L ? __expf(s[i] * a.scale - m) * inv : 0.f;
          const bool keep = a.thresh == 0 || kept(ebase, query, key, dkey, a.thresh);
          s[i] = p * ((keep ? dp[i] * a.drop_scale : 0.f) - dl);          // dS^T
        }
        bf16x8 p0, p1;
        acc_to_frags(s, p0, p1);
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(img + kt * kTileImg, 0, lane), p0, dq, 0, 0, 0);
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(img + kt * kTileImg, 1, lane), p1, dq, 0, 0, 0);
      }
    if (query < a.L) store_row16(a.dq + (long long)b * a.q_bs + (long long)query * a.q_rs + hd * kD + 16 * hh, dq, a.scale);
    return;
  }
  // =================================================================== 32 keys: dV^T = dO^T P, dK^T = Q^T dS, queries in tiles
  char* qimg = img;
  char* doimg = img + kTileImg;
  const int key = 32 * tile + r;
  const long long kr = key < a.L ? key : a.L - 1;
  const __bf16* kp = kb + kr * a.k_rs;
  const __bf16* vp = vb + kr * a.v_rs;
  const bf16x8 kf0 = row_frag_g(kp, 0, hh), kf1 = row_frag_g(kp, 1, hh);      // B operand K^T[d][key]
  const bf16x8 vf0 = row_frag_g(vp, 0, hh), vf1 = row_frag_g(vp, 1, hh);
  f32x16 dv = zero16(), dk = zero16();
  for (int qt = 0; qt < a.T; ++qt) {
    load_tile(qimg, qb, a.q_rs, 32 * qt, a.L, lane);
    load_tile(doimg, dob, a.o_rs, 32 * qt, a.L, lane);
    wave_lds_sync();
    f32x16 s = zero16(), dp = zero16();
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(qimg, r, 0, hh), kf0, s, 0, 0, 0);       // S = Q K^T: queries x keys
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(qimg, r, 1, hh), kf1, s, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(doimg, r, 0, hh), vf0, dp, 0, 0, 0);    // dP = dO V^T
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(doimg, r, 1, hh), vf1, dp, 0, 0, 0);
    f32x16 pd;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int query = 32 * qt + (i & 3) + 8 * (i >> 2) + 4 * hh;
      const int qi = query < a.L ? query : a.L - 1;
      const float m = stb[qi * 2], inv = stb[qi * 2 + 1], dl = dlb[qi];
      const float p = (query < a.L && key < a.L) ? __expf(s[i] * a.scale - m) * inv : 0.f;
      const bool keep = a.thresh == 0 || kept(ebase, query, key, dkey, a.thresh);
      pd[i] = keep ? p * a.drop_scale : 0.f;                                 // dropped P (what multiplied V)
      s[i] = p * ((keep ? dp[i] * a.drop_scale : 0.f) - dl);                 // dS
    }
    bf16x8 p0, p1, s0, s1;
    acc_to_frags(pd, p0, p1);
    acc_to_frags(s, s0, s1);
    dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(doimg, 0, lane), p0, dv, 0, 0, 0);
    dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(doimg, 1, lane), p1, dv, 0, 0, 0);
    dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(qimg, 0, lane), s0, dk, 0, 0, 0);
    dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(qimg, 1, lane), s1, dk, 0, 0, 0);
    wave_lds_sync();          // the next query tile overwrites the images
  }
  if (key < a.L) {
    store_row16(a.dv + (long long)b * a.v_bs + (long long)key * a.v_rs + hd * kD + 16 * hh, dv, 1.f);
    store_row16(a.dk + (long long)b * a.k_bs + (long long)key * a.k_rs + hd * kD + 16 * hh, dk, a.scale);
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int attn_fill(AttnArgs* a, const char* who, const void* q, const void* k, const void* v, int B, int H, int L, int head_dim,
              const int64_t* strides, float scale, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* epoch,
              int dtype) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "%s: bf16 only", who);
  if (head_dim != kD || L < 1 || L > 32 * kMaxT || B < 0 || H < 1)
    return fail(DSKD_ERR_INVALID_ARG, "%s: built for head dimension 32 and 1..320 tokens (got %d, %d)", who, head_dim, L);
  if (!q || !k || !v || !strides || !al16(q) || !al16(k) || !al16(v))
    return fail(DSKD_ERR_INVALID_ARG, "%s: null or misaligned pointer", who);
  for (int i = 0; i < 8; ++i)
    if (strides[i] < 0 || (strides[i] & 7)) return fail(DSKD_ERR_INVALID_ARG, "%s: strides must be non-negative multiples of 8 elements", who);
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return fail(DSKD_ERR_INVALID_ARG, "%s: drop_p=%f", who, drop_p);
  if ((long long)B * H * L * 512ll > 0xFFFFFFFFll) return fail(DSKD_ERR_INVALID_ARG, "%s: B H L too large for the dropout index", who);
  *a = AttnArgs{};
  a->q = (const __bf16*)q; a->k = (const __bf16*)k; a->v = (const __bf16*)v;
  a->q_bs = strides[0]; a->q_rs = strides[1]; a->k_bs = strides[2]; a->k_rs = strides[3];
  a->v_bs = strides[4]; a->v_rs = strides[5]; a->o_bs = strides[6]; a->o_rs = strides[7];
  a->B = B; a->H = H; a->L = L; a->T = (L + 31) / 32;
  a->scale = scale;
  const double t = (double)drop_p * 65536.0 + 0.5;
  a->thresh = drop_p > 0.f ? (unsigned)(t < 1.0 ? 1.0 : t) : 0u;
  a->drop_scale = 1.0f / (1.0f - drop_p);
  a->seed = seed; a->offset = offset; a->epoch = reinterpret_cast<const unsigned long long*>(epoch);
  return DSKD_OK;
}

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int dskd_attn_fwd(const void* q, const void* k, const void* v, void* out, float* stats, int B, int H, int L,
                             int head_dim, const int64_t* strides, float scale, float drop_p, uint64_t seed, uint64_t offset,
                             const uint64_t* epoch, int dtype, void* stream) {
  AttnArgs a;
  if (int rc = attn_fill(&a, "dskd_attn_fwd", q, k, v, B, H, L, head_dim, strides, scale, drop_p, seed, offset, epoch, dtype)) return rc;
  if (!out || !al16(out) || (stats && !al16(stats))) return fail(DSKD_ERR_INVALID_ARG, "dskd_attn_fwd: bad output pointer");
  if (B == 0) return DSKD_OK;
  a.out = (__bf16*)out; a.stats = stats;
  constexpr int lds = kWaves * kMaxT * kTileImg;
  static bool done[64] = {};
  if (!reserve_lds((const void*)attn_fwd_kernel, lds, done)) return fail(DSKD_ERR_LAUNCH, "dskd_attn_fwd: cannot reserve %d bytes of LDS", lds);
  const long long tasks = (long long)B * H * a.T;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)((tasks + kWaves - 1) / kWaves)), dim3(kWaves * 64), lds, (hipStream_t)stream, a);
  return check_launch("dskd_attn_fwd");
}

extern "C" int dskd_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* stats,
                             float* delta, void* dq, void* dk, void* dv, int B, int H, int L, int head_dim,
                             const int64_t* strides, float scale, float drop_p, uint64_t seed, uint64_t offset,
                             const uint64_t* epoch, int dtype, void* stream) {
  AttnArgs a;
  if (int rc = attn_fill(&a, "dskd_attn_bwd", q, k, v, B, H, L, head_dim, strides, scale, drop_p, seed, offset, epoch, dtype)) return rc;
  if (!out || !dout || !stats || !delta || !dq || !dk || !dv || !al16(out) || !al16(dout) || !al16(dq) || !al16(dk) || !al16(dv))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_attn_bwd: null or misaligned pointer");
  if (B == 0) return DSKD_OK;
  a.o = (const __bf16*)out; a.dout = (const __bf16*)dout; a.stats = const_cast<float*>(stats); a.delta = delta;
  a.dq = (__bf16*)dq; a.dk = (__bf16*)dk; a.dv = (__bf16*)dv;
  hipStream_t st = (hipStream_t)stream;
  const long long rows = (long long)B * H * L;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, a);
  constexpr int lds = kWaves * kMaxT * kTileImg;
  static bool done[64] = {};
  if (!reserve_lds((const void*)attn_bwd_kernel, lds, done)) return fail(DSKD_ERR_LAUNCH, "dskd_attn_bwd: cannot reserve %d bytes of LDS", lds);
  const long long tasks = 2ll * B * H * a.T;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)((tasks + kWaves - 1) / kWaves)), dim3(kWaves * 64), lds, st, a);
  return check_launch("dskd_attn_bwd");
}
