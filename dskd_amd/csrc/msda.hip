// Multi-scale deformable attention (sampling + aggregation), forward and backward,
// hand-written for gfx950 (MI355X).
//
// Replaces ext-mmcv `MultiScaleDeformableAttnFunction` (mmcv-full 1.3.17..1.6.2), which
// the reference imports at mmdet/models/utils/transformer.py:22-29 and reaches from the
// encoder (transformer.py:985-995) and the decoder (transformer.py:1032-1043).
// Semantics: per (b, q, h)
//   out[b,q,h,:] = sum_{l,p} attn[b,q,h,l,p] * S_l(loc[b,q,h,l,p])
// with x = loc_x*W_l - 0.5, y = loc_y*H_l - 0.5 and S_l the zero-padded bilinear sample of
// value[b, start_l + y*W_l + x, h, :].
//
// MI355X mapping (not mmcv's one-thread-per-output-channel):
//  * value rows are [heads*32] contiguous, so one corner of one head is a 128-B (f32) or
//    64-B (bf16) line.  A wave owns whole queries: 8 heads x (8 | 4) lanes, 16 B per lane,
//    so every gather instruction moves complete lines and the output row is one coalesced
//    1-KiB (f32) / 512-B (bf16) store.  No cross-lane reduction in the forward at all.
//  * the 128 sampling points of a query (8 heads x 16) are turned into (byte offset, weight)
//    quadruples ONCE, cooperatively (2 points per lane), parked in the wave's private LDS
//    slice and re-read as broadcast ds_read_b128 -- instead of every lane redoing the
//    floor/weight/bounds arithmetic for its head.
//  * gathers are raw buffer loads: 32-bit offsets against a per-image descriptor, and an
//    out-of-image corner is encoded as an out-of-range offset, which the hardware returns
//    as zeros (exactly grid_sample's zero padding; no clamped re-read, no NaN*0).
//  * work items are dealt to XCDs in contiguous chunks (xcd_remap) so that raster
//    neighbours, which sample overlapping windows, share one 4 MiB L2.
//  * backward: d(out)/d(value) is a scatter; contributions are added with no-return
//    global_atomic_add_f32 shaped as two full 128-B segments per wave instruction (the
//    full-rate shape, MI355X_MICROARCH.md "Global float atomics"); grad_loc / grad_attn
//    come from per-head dot products reduced with DPP inside 8- / 4-lane groups.
#include "common.h"
#include "msda_internal.h"
#include "msda_geom.h"
#include <stdlib.h>

namespace dskd {
namespace {

constexpr int kWaves = 4;       // waves per workgroup
constexpr int kMaxLP = 16;      // levels * points

// Per-lane selection among four scalars.  Arrays inside kernel-argument structs must only be
// indexed with compile-time constants (a runtime index sends the whole struct to scratch).
__device__ __forceinline__ int sel4v(int a0, int a1, int a2, int a3, int i) {
  return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
}
#define SEL4(arr, i) sel4v((arr)[0], (arr)[1], (arr)[2], (arr)[3], (i))
__device__ __forceinline__ float sel4f(float a0, float a1, float a2, float a3, int i) {
  return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
}
#define SEL4F(arr, i) sel4f((arr)[0], (arr)[1], (arr)[2], (arr)[3], (i))

template <typename T>
struct Traits;
template <>
struct Traits<float> {
  static constexpr int QPW = 1;    // queries per wave pass
  static constexpr int LPH = 8;    // lanes per head
  static constexpr int ROWB = 1024;  // bytes per value row (8 heads x 32 ch)
  static constexpr int NACC = 4;   // channels per lane
};
template <>
struct Traits<__bf16> {
  static constexpr int QPW = 2;
  static constexpr int LPH = 4;
  static constexpr int ROWB = 512;
  static constexpr int NACC = 8;
};

// One sampling point -> four corner byte offsets (row * ROWB, or kOOB) and the four
// bilinear weights.  aux = (lx, ly, attn, level) for the backward.
// Per-level geometry is looked up in a tiny LDS table indexed by the lane's level: chains of
// selects over the kernel-argument arrays were lowered to divergent branch trees by the compiler.
__device__ __forceinline__ void fill_level_table(i32x4* tab, const LevelGeom& g) {
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l)
    if (threadIdx.x == l) tab[l] = i32x4{g.H[l], g.W[l], g.start[l], 0};
}

template <int ROWB>
__device__ __forceinline__ void point_params(float lx_n, float ly_n, float a, int lvl,
                                             const i32x4* tab, i32x4& off, f32x4& w,
                                             f32x4& aux) {
  const i32x4 lt = tab[lvl];
  const int H = lt.x, W = lt.y, st = lt.z;
  const float x = lx_n * (float)W - 0.5f;
  const float y = ly_n * (float)H - 0.5f;
  off = i32x4{kOOB, kOOB, kOOB, kOOB};
  w = f32x4{0.f, 0.f, 0.f, 0.f};
  aux = f32x4{0.f, 0.f, a, (float)lvl};
  // Same acceptance test as the reference op: strictly inside (-1, size).  NaN fails it.
  if (x > -1.f && y > -1.f && x < (float)W && y < (float)H) {
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf;
    const float lx = x - xf, ly = y - yf;
    const float hx = 1.f - lx, hy = 1.f - ly;
    const bool vx0 = x0 >= 0, vx1 = x0 + 1 <= W - 1;
    const bool vy0 = y0 >= 0, vy1 = y0 + 1 <= H - 1;
    const int r00 = (st + y0 * W + x0) * ROWB;
    off.x = (vy0 && vx0) ? r00 : kOOB;
    off.y = (vy0 && vx1) ? r00 + ROWB : kOOB;
    off.z = (vy1 && vx0) ? r00 + W * ROWB : kOOB;
    off.w = (vy1 && vx1) ? r00 + W * ROWB + ROWB : kOOB;
    w = f32x4{hy * hx, hy * lx, ly * hx, ly * lx};
    aux.x = lx;
    aux.y = ly;
  }
}

__device__ __forceinline__ void unpack_bf16x8(const u32x4& v, float* f) {
  f[0] = as_f32(v.x << 16);
  f[1] = as_f32(v.x & 0xFFFF0000u);
  f[2] = as_f32(v.y << 16);
  f[3] = as_f32(v.y & 0xFFFF0000u);
  f[4] = as_f32(v.z << 16);
  f[5] = as_f32(v.z & 0xFFFF0000u);
  f[6] = as_f32(v.w << 16);
  f[7] = as_f32(v.w & 0xFFFF0000u);
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 p = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, p);
}

template <typename T>
__device__ __forceinline__ void load_vals(__amdgpu_buffer_rsrc_t rsrc, int voff, float* f) {
  // the builtin returns a GCC vector_size type: cast explicitly
  const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
  if constexpr (sizeof(T) == 4) {
    f[0] = as_f32(v.x);
    f[1] = as_f32(v.y);
    f[2] = as_f32(v.z);
    f[3] = as_f32(v.w);
  } else {
    unpack_bf16x8(v, f);
  }
}

// Ablation builds of the FORWARD kernels (tools/prof/msda_fwd_ablation.sh, profiles/r04_msda_fwd_ablation.txt;
// timing only, results are garbage): -DDSKD_FWD_ABLATE=1 keeps every corner load (texture path / LDS) and drops the unpack +
// FMA work (one xor chain + one FMA per sample); =2 keeps staging + arithmetic and drops the corner loads (operands made from
// the offsets).  0 = the product.
#ifndef DSKD_FWD_ABLATE
#define DSKD_FWD_ABLATE 0
#endif
__device__ __forceinline__ void fwd_fma4_bf16(float (&acc)[8], const u32x4& r0, const u32x4& r1, const u32x4& r2,
                                              const u32x4& r3, const f32x4& w) {
#if DSKD_FWD_ABLATE == 1
  const u32x4 x = r0 ^ r1 ^ r2 ^ r3;
  acc[0] = fmaf(w.x, as_f32((x.x ^ x.y ^ x.z ^ x.w) & 0x3FFFFFFFu), acc[0]);
#else
  float v0[8], v1[8], v2[8], v3[8];
  unpack_bf16x8(r0, v0);
  unpack_bf16x8(r1, v1);
  unpack_bf16x8(r2, v2);
  unpack_bf16x8(r3, v3);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = fmaf(w.x, v0[i], acc[i]);
    acc[i] = fmaf(w.y, v1[i], acc[i]);
    acc[i] = fmaf(w.z, v2[i], acc[i]);
    acc[i] = fmaf(w.w, v3[i], acc[i]);
  }
#endif
}
__device__ __forceinline__ u32x4 fwd_fake_operand(int off) {
  const unsigned u = (unsigned)off;
  return u32x4{u * 0x9E3779B1u, u ^ 0x3F803F80u, u + 0x3F003F00u, ~u};
}

// r4: the dot product of a corner's 8 bf16 channels with the 8 bf16 channels of grad_out as four v_dot2c_f32_bf16 (bf16 x bf16
// products are exact in f32; f32 accumulation) instead of 16 unpack + 8 FMA instructions: the gather kernels of the backward
// are bound by their vector instruction stream (profiles/r04_msda_fwd_ablation.txt shows it for the forward's same loop), and the
// four corners of a sample were 48-64 of their ~110 instructions per sample and lane.  The forward cannot use it (its weights
// are f32).  Both gather kernels use this helper, so they stay bit-identical to each other.
// Inline asm, not __builtin_amdgcn_fdot2_f32_bf16: with the builtin on bit-cast vector elements this compiler (ROCm 7.2
// clang) emitted all four instructions on the FIRST element pair (found by the oracle tests; isolated in a 20-line kernel).
__device__ __forceinline__ float dot8_bf16(const u32x4& v, const u32x4& g) {
  // All four links are the VOP3P form (the first with the constant 0 as its addend: no zeroing move).  Mixing the forms --
  // VOP3P first, then three VOP2 v_dot2c -- gave WRONG sums in 2 % of the entries: inline asm is opaque to the compiler's hazard
  // recognizer and a DOT result handed to a DIFFERENT opcode is not forwarded; same-opcode chains are (oracle tests).
  float d;
  asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(d) : "v"(v.x), "v"(g.x));
  asm("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(v.y), "v"(g.y));
  asm("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(v.z), "v"(g.z));
  asm("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(v.w), "v"(g.w));
  return d;
}

// A DOT result read by a non-DOT instruction needs 3 wait states (LLVM's hazard recognizer inserts them for instructions it
// sees; an asm block it does not): one s_nop behind the four chains of a sample, tied to the values so that it stays between
// the dot products and their first use.
__device__ __forceinline__ void dot8_settle(float& a, float& b, float& c, float& d) {
  asm volatile("s_nop 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// Cooperative parameter pass shared by forward and backward.
//
// PH > 1: the samples of a query are staged and consumed in PH phases of LP / PH samples per head
// (phase ``ph`` holds samples [ph * LP/PH, (ph+1) * LP/PH)), so a wave's LDS slice shrinks by PH.
// With two bf16 queries per wave the single-phase slices (34.9 KB per workgroup forward, 52.3 KB
// backward) cap the CU at 4 / 3 waves per SIMD although the kernels need only 64 VGPRs; the
// gathers are latency-bound, so resident waves are what hides them.  The accumulation order over
// the samples does not change: results are bit-identical for every PH.
// Head stride in 16-byte slots: LP/PH samples + 1 pad -> 17 / 9 / 5, which keeps the 16 (query,
// head) groups of a wave on distinct LDS banks for the broadcast ds_read_b128 of the consumer.
template <int PH>
struct Phased {
  static_assert(PH == 1 || PH == 2 || PH == 4, "phases");
  static constexpr int HS = kMaxLP / PH + 1;
};

template <typename T, bool WITH_AUX, int PH, bool KEEP_WT>
__device__ __forceinline__ void stage_points(const float* __restrict__ loc,
                                             const float* __restrict__ attn,
                                             const i32x4* g, int b, int Nq, int q0,
                                             int q_end, int LP, int points, int ph, int lane,
                                             i32x4* s_off, f32x4* s_wt, f32x4* s_aux) {
  using TR = Traits<T>;
  constexpr int HS = Phased<PH>::HS;
  const int LPS = LP / PH;                 // samples per head in this phase
  const int npts = TR::QPW * kHeads * LPS;
  for (int pi = lane; pi < npts; pi += 64) {
    const int qs = pi / (kHeads * LPS);
    const int r = pi - qs * (kHeads * LPS);
    const int h = r / LPS;
    const int sl = r - h * LPS;
    const int s = ph * LPS + sl;
    const int q = q0 + qs;
    i32x4 off = i32x4{kOOB, kOOB, kOOB, kOOB};
    f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 aux = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < q_end) {
      const size_t base = ((size_t)b * Nq + q) * (size_t)(kHeads * LP) + h * LP + s;
      const f32x2 xy = *reinterpret_cast<const f32x2*>(loc + base * 2);
      const float a = attn[base];
      point_params<TR::ROWB>(xy.x, xy.y, a, s / points, g, off, w, aux);
    }
    const int slot = (qs * kHeads + h) * HS + sl;
    s_off[slot] = off;
    if constexpr (WITH_AUX) {
      if constexpr (KEEP_WT) s_wt[slot] = w;  // raw bilinear weights; attn lives in aux.z
      s_aux[slot] = aux;
    } else {
      s_wt[slot] = w * aux.z;  // attn folded in
    }
  }
}

// The same pass when the MODULE prologue is folded in (no-grad forward: frozen teacher, inference):
// the points come straight from the projection output ``both`` ([.., heads*16*2] offsets then
// [.., heads*16] logits, as written by the fused offsets/logits GEMM) and the reference points --
// softmax over the 16 (level, point) logits of a head with DPP row reductions, loc = ref + off /
// (W, H) -- with the arithmetic of msda_prep.hip, so the result is bit-identical to prologue
// kernel + sampling kernel while loc / attn (1.5 KB per query, f32) are never written or read.
// PH == 2: 8 consecutive lanes hold one half of a head's 16 points and fetch the other half's logit
// as well; max is exact, and the sum is formed as (own half) + (other half) with the same three
// DPP steps per half as row16_sum's first three, i.e. the same additions in the same order.
template <typename T, int PH>
__device__ __forceinline__ void stage_points_fused(const T* __restrict__ both, const float* __restrict__ ref,
                                                   const i32x4* g, int b, int Nq, int q0, int q_end,
                                                   int levels, int points, int ph, int lane, i32x4* s_off,
                                                   f32x4* s_wt) {
  static_assert(PH == 1 || PH == 2, "fused prologue: one or two phases");
  using TR = Traits<T>;
  constexpr int HS = Phased<PH>::HS;
  constexpr int PQ = kHeads * 16;                  // points per query (levels * points == 16)
  constexpr int LPS = 16 / PH;
  constexpr int npts = TR::QPW * kHeads * LPS;
  for (int pi = lane; pi < npts; pi += 64) {       // 16 / PH consecutive lanes = the staged points of one (query, head)
    const int qs = pi / (kHeads * LPS);
    const int rr = pi - qs * (kHeads * LPS);
    const int h = rr / LPS, s = ph * LPS + (rr - h * LPS);
    const int r = h * 16 + s;
    const int q = q0 + qs;
    const int qc = q < q_end ? q : q_end - 1;
    const T* row = both + ((size_t)b * Nq + qc) * (size_t)(PQ * 3);
    const float ox = (float)row[r * 2], oy = (float)row[r * 2 + 1];
    const float lg = (float)row[PQ * 2 + r];
    float a;
    if constexpr (PH == 1) {
      const float mx = row16_max(lg);
      const float e = __expf(lg - mx);
      a = e / row16_sum(e);
    } else {
      const float lg2 = (float)row[PQ * 2 + (r ^ 8)];
      const float mx = group8_max(fmaxf(lg, lg2));
      const float e = __expf(lg - mx), e2 = __expf(lg2 - mx);
      a = e / (group8_sum(e) + group8_sum(e2));
    }
    const int lvl = s / points;
    const i32x4 lt = g[lvl];                        // {H, W, start, 0}
    const f32x2 rf = *reinterpret_cast<const f32x2*>(ref + (((size_t)b * Nq + qc) * levels + lvl) * 2);
    i32x4 off = i32x4{kOOB, kOOB, kOOB, kOOB};
    f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 aux = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < q_end) point_params<TR::ROWB>(rf.x + ox / (float)lt.y, rf.y + oy / (float)lt.x, a, lvl, g, off, w, aux);
    const int slot = (qs * kHeads + h) * HS + (s - ph * LPS);
    s_off[slot] = off;
    s_wt[slot] = w * aux.z;
  }
}

// ------------------------------------------------------------------ forward
template <typename T, bool FUSED, int PH>
__global__ __launch_bounds__(kWaves * 64) void msda_fwd_kernel(
    const T* __restrict__ value, const float* __restrict__ loc,
    const float* __restrict__ attn, const T* __restrict__ both, const float* __restrict__ ref,
    T* __restrict__ out, LevelGeom g, int Nv, int Nq, int LP, int points, int qpb,
    int blocks_per_img) {
  using TR = Traits<T>;
  constexpr int HS = Phased<PH>::HS;
  constexpr int SLOTS = TR::QPW * kHeads * HS;
  __shared__ i32x4 s_off_all[kWaves][SLOTS];
  __shared__ f32x4 s_wt_all[kWaves][SLOTS];
  __shared__ i32x4 s_lvl[kMaxLevels];
  fill_level_table(s_lvl, g);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  i32x4* s_off = s_off_all[wave];
  f32x4* s_wt = s_wt_all[wave];

  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int b = vb / blocks_per_img;
  const int blk = vb - b * blocks_per_img;
  const int q_begin = blk * qpb;
  const int q_end = min(q_begin + qpb, Nq);

  const T* vbase = value + (size_t)b * Nv * (kHeads * kCh);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(vbase), 0, Nv * TR::ROWB, 0x00020000);

  // lane -> (query slot, head, 16-byte part of the head's line)
  const int qs = lane / (kHeads * TR::LPH);
  const int h = (lane / TR::LPH) & (kHeads - 1);
  const int part = lane & (TR::LPH - 1);
  const int hb = h * (kCh * (int)sizeof(T)) + part * 16;
  const i32x4* my_off = s_off + (qs * kHeads + h) * HS;
  const f32x4* my_wt = s_wt + (qs * kHeads + h) * HS;
  const int LPS = LP / PH;      // samples per head and phase

  for (int q0 = q_begin + wave * TR::QPW; q0 < q_end; q0 += kWaves * TR::QPW) {
    float acc[TR::NACC];
#pragma unroll
    for (int i = 0; i < TR::NACC; ++i) acc[i] = 0.f;

#pragma unroll 1
    for (int ph = 0; ph < PH; ++ph) {
      if constexpr (FUSED)
        stage_points_fused<T, PH>(both, ref, s_lvl, b, Nq, q0, q_end, LP / points, points, ph, lane, s_off, s_wt);
      else
        stage_points<T, false, PH, true>(loc, attn, s_lvl, b, Nq, q0, q_end, LP, points, ph, lane, s_off, s_wt,
                                         nullptr);
      wave_lds_sync();

#pragma unroll 4
      for (int s = 0; s < LPS; ++s) {
        const i32x4 o = my_off[s];
        const f32x4 w = my_wt[s];
#if DSKD_FWD_ABLATE
        if constexpr (sizeof(T) == 2) {
#if DSKD_FWD_ABLATE == 2
          fwd_fma4_bf16(acc, fwd_fake_operand(o.x + hb), fwd_fake_operand(o.y + hb), fwd_fake_operand(o.z + hb),
                        fwd_fake_operand(o.w + hb), w);
#else
          fwd_fma4_bf16(acc, __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + hb, 0, 0)),
                        __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + hb, 0, 0)),
                        __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + hb, 0, 0)),
                        __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + hb, 0, 0)), w);
#endif
          continue;
        }
#endif
        float v0[TR::NACC], v1[TR::NACC], v2[TR::NACC], v3[TR::NACC];
        load_vals<T>(rsrc, o.x + hb, v0);
        load_vals<T>(rsrc, o.y + hb, v1);
        load_vals<T>(rsrc, o.z + hb, v2);
        load_vals<T>(rsrc, o.w + hb, v3);
#pragma unroll
        for (int i = 0; i < TR::NACC; ++i) {
          acc[i] = fmaf(w.x, v0[i], acc[i]);
          acc[i] = fmaf(w.y, v1[i], acc[i]);
          acc[i] = fmaf(w.z, v2[i], acc[i]);
          acc[i] = fmaf(w.w, v3[i], acc[i]);
        }
      }
      wave_lds_sync();  // the next staging pass overwrites this wave's LDS slice
    }

    const int q = q0 + qs;
    if (q < q_end) {
      T* orow = out + ((size_t)b * Nq + q) * (kHeads * kCh) + h * kCh;
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(orow + part * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
      } else {
        u32x4 p;
        p.x = pack_bf16x2(acc[0], acc[1]);
        p.y = pack_bf16x2(acc[2], acc[3]);
        p.z = pack_bf16x2(acc[4], acc[5]);
        p.w = pack_bf16x2(acc[6], acc[7]);
        *reinterpret_cast<u32x4*>(orow + part * 8) = p;
      }
    }
  }
}

// ------------------------------------------------------------------ backward
// PH > 1 (grad_loc / grad_attn only, i.e. !WITH_VALUE): phased staging as in the forward; the raw
// bilinear weights are then not parked in LDS at all -- the one lane per group that finishes a
// sample recomputes them from (lx, ly) with the same multiplications (for a rejected location
// they are (1,0,0,0) instead of zeros, against dot products that are exactly zero) -- and every
// phase writes its finished gradients out as coalesced runs of LP/PH points per (query, head).
template <typename T, bool WITH_VALUE, int PH>
__global__ __launch_bounds__(kWaves * 64) void msda_bwd_kernel(
    const T* __restrict__ value, const float* __restrict__ loc,
    const float* __restrict__ attn, const T* __restrict__ grad_out,
    float* __restrict__ grad_value, float* __restrict__ grad_loc,
    float* __restrict__ grad_attn, LevelGeom g, int Nv, int Nq, int LP, int points,
    int qpb, int blocks_per_img) {
  static_assert(PH == 1 || !WITH_VALUE, "the scatter phase needs every sample of the query staged");
  using TR = Traits<T>;
  constexpr bool KEEP_WT = (PH == 1);
  constexpr int HS = Phased<PH>::HS;
  constexpr int SLOTS = TR::QPW * kHeads * HS;
  __shared__ i32x4 s_off_all[kWaves][SLOTS];
  __shared__ f32x4 s_wt_all[kWaves][KEEP_WT ? SLOTS : 1];
  __shared__ f32x4 s_aux_all[kWaves][SLOTS];
  __shared__ i32x4 s_lvl[kMaxLevels];
  fill_level_table(s_lvl, g);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  i32x4* s_off = s_off_all[wave];
  f32x4* s_wt = s_wt_all[wave];
  f32x4* s_aux = s_aux_all[wave];

  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int b = vb / blocks_per_img;
  const int blk = vb - b * blocks_per_img;
  const int q_begin = blk * qpb;
  const int q_end = min(q_begin + qpb, Nq);

  const T* vbase = value + (size_t)b * Nv * (kHeads * kCh);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(vbase), 0, Nv * TR::ROWB, 0x00020000);
  float* gvb = grad_value + (size_t)b * Nv * (kHeads * kCh);

  const int qs = lane / (kHeads * TR::LPH);
  const int h = (lane / TR::LPH) & (kHeads - 1);
  const int part = lane & (TR::LPH - 1);
  const int hb = h * (kCh * (int)sizeof(T)) + part * 16;
  const int myslot = (qs * kHeads + h) * HS;
  const int LPS = LP / PH;      // samples per head and phase

  for (int q0 = q_begin + wave * TR::QPW; q0 < q_end; q0 += kWaves * TR::QPW) {
    // ---- phase A: grad_attn / grad_loc.  Lane = (query slot, head, 16-B part).
    const int q = q0 + qs;
    const bool qv = q < q_end;
    float go[TR::NACC];
    u32x4 go_pk = u32x4{0u, 0u, 0u, 0u};      // bf16: the packed slice for the dot products
    {
      const int qc = qv ? q : q_end - 1;
      const T* grow = grad_out + ((size_t)b * Nq + qc) * (kHeads * kCh) + h * kCh;
      if constexpr (sizeof(T) == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(grow + part * 4);
        go[0] = t.x; go[1] = t.y; go[2] = t.z; go[3] = t.w;
      } else {
        go_pk = *reinterpret_cast<const u32x4*>(grow + part * 8);
        unpack_bf16x8(go_pk, go);
      }
    }
#pragma unroll 1
    for (int ph = 0; ph < PH; ++ph) {
      stage_points<T, true, PH, KEEP_WT>(loc, attn, s_lvl, b, Nq, q0, q_end, LP, points, ph, lane, s_off, s_wt,
                                         s_aux);
      wave_lds_sync();
      for (int s = 0; s < LPS; ++s) {
        const i32x4 o = s_off[myslot + s];
        const f32x4 ax = s_aux[myslot + s];
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        if constexpr (sizeof(T) == 2) {
          const u32x4 r0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + hb, 0, 0));
          const u32x4 r1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + hb, 0, 0));
          const u32x4 r2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + hb, 0, 0));
          const u32x4 r3 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + hb, 0, 0));
          d0 = dot8_bf16(r0, go_pk); d1 = dot8_bf16(r1, go_pk); d2 = dot8_bf16(r2, go_pk); d3 = dot8_bf16(r3, go_pk);
          dot8_settle(d0, d1, d2, d3);
        } else {
          float v0[TR::NACC], v1[TR::NACC], v2[TR::NACC], v3[TR::NACC];
          load_vals<T>(rsrc, o.x + hb, v0);
          load_vals<T>(rsrc, o.y + hb, v1);
          load_vals<T>(rsrc, o.z + hb, v2);
          load_vals<T>(rsrc, o.w + hb, v3);
#pragma unroll
          for (int i = 0; i < TR::NACC; ++i) {
            d0 = fmaf(v0[i], go[i], d0);
            d1 = fmaf(v1[i], go[i], d1);
            d2 = fmaf(v2[i], go[i], d2);
            d3 = fmaf(v3[i], go[i], d3);
          }
        }
        if constexpr (TR::LPH == 8) {
          d0 = group8_sum(d0); d1 = group8_sum(d1); d2 = group8_sum(d2); d3 = group8_sum(d3);
        } else {
          d0 = group4_sum(d0); d1 = group4_sum(d1); d2 = group4_sum(d2); d3 = group4_sum(d3);
        }
        // One lane of the group finishes this sample's gradients and parks them in the sample's
        // (now dead) aux slot; they leave for HBM after the loop as coalesced rows.  Per-lane
        // 4-/8-byte stores from here measured 3.9x write amplification (WRITE_SIZE 525 MB for
        // 136 MB of gradients at B=4).
        if (part == (s & (TR::LPH - 1))) {
          const float lx = ax.x, ly = ax.y, a = ax.z;
          const int lvl = (int)ax.w;
          const i32x4 lt = s_lvl[lvl];
          const float Wf = (float)lt.y, Hf = (float)lt.x;
          const float hx = 1.f - lx, hy = 1.f - ly;
          f32x4 w;
          if constexpr (KEEP_WT) w = s_wt[myslot + s];
          else w = f32x4{hy * hx, hy * lx, ly * hx, ly * lx};
          const float ga = w.x * d0 + w.y * d1 + w.z * d2 + w.w * d3;
          const float gx = Wf * a * (hy * (d1 - d0) + ly * (d3 - d2));
          const float gy = Hf * a * (hx * (d2 - d0) + lx * (d3 - d1));
          s_aux[myslot + s] = f32x4{gx, gy, a, ga};   // .z (attn) stays for the scatter phase
        }
      }
      wave_lds_sync();
      {
        const int npts = TR::QPW * kHeads * LPS;
        for (int pi = lane; pi < npts; pi += 64) {
          const int qs2 = pi / (kHeads * LPS);
          const int r = pi - qs2 * (kHeads * LPS);
          const int h2 = r / LPS;
          const int sl = r - h2 * LPS;
          const int q2 = q0 + qs2;
          if (q2 < q_end) {
            const f32x4 res = s_aux[(qs2 * kHeads + h2) * HS + sl];
            const size_t base = ((size_t)b * Nq + q2) * (size_t)(kHeads * LP) + h2 * LP + ph * LPS + sl;
            grad_attn[base] = res.w;
            *reinterpret_cast<f32x2*>(grad_loc + base * 2) = f32x2{res.x, res.y};
          }
        }
      }
      if constexpr (!WITH_VALUE) wave_lds_sync();   // the next staging pass overwrites the slice
    }

    if constexpr (WITH_VALUE) {
      // ---- phase B: grad_value scatter.  Lane = (corner parity, channel): one wave
      // instruction adds two complete 128-B head lines.
      const int ch = lane & 31;
      const int cpar = lane >> 5;
#pragma unroll 1
      for (int qq = 0; qq < TR::QPW; ++qq) {
        const int q2 = q0 + qq;
        if (q2 >= q_end) break;
        const T* grow = grad_out + ((size_t)b * Nq + q2) * (kHeads * kCh);
        float gch[kHeads];
#pragma unroll
        for (int hh = 0; hh < kHeads; ++hh) gch[hh] = (float)grow[hh * kCh + ch];
#pragma unroll
        for (int hh = 0; hh < kHeads; ++hh) {
          const int slot = (qq * kHeads + hh) * HS;
          const int hoff = hh * kCh + ch;
          for (int s = 0; s < LP; ++s) {
            const i32x4 o = s_off[slot + s];
            const f32x4 w = s_wt[slot + s];
            const float a = s_aux[slot + s].z;
            const float ga = a * gch[hh];
            // corners (0,1) then (2,3); each half-wave takes one corner
            const int oa = cpar ? o.y : o.x;
            const float wa = cpar ? w.y : w.x;
            const int ob = cpar ? o.w : o.z;
            const float wb = cpar ? w.w : w.z;
            if (oa != kOOB)
              atomicAdd(gvb + (size_t)(oa / TR::ROWB) * (kHeads * kCh) + hoff, wa * ga);
            if (ob != kOOB)
              atomicAdd(gvb + (size_t)(ob / TR::ROWB) * (kHeads * kCh) + hoff, wb * ga);
          }
        }
      }
      wave_lds_sync();
    }
  }
}


// ------------------------------------------------------------------ backward, grad_value (encoder)
// In the encoder the queries ARE the pixels (Nq == Nv) and every sampling location is the
// pixel's own centre plus a few-pixel offset, so the scatter of d(out)/d(value) is spatially
// local.  v1 above pays one global float atomic per (corner, channel): 1.46 GB of atomic bytes
// per image and call against a chip-wide rate of ~1.3 TB/s (MI355X_MICROARCH.md "Global float
// atomics") -> ~1.04 ms per image, 10x everything else in this op, with thousands of adders
// per row on the coarse levels.  Here a workgroup owns a REGION of the image (<= 32x32 level-0
// pixels, the queries of all four levels whose centre falls inside it), one head and a
// 16- or 32-channel group, and accumulates into LDS windows (region footprint + margin on
// the levels it handles, ~100 KB); only the windows' non-zero entries reach HBM as atomics
// (~20x fewer atomic bytes).  Samples that leave the window (large learned offsets) fall back to a direct global
// atomic, so the result is correct for ANY sampling locations.
//  * LDS float atomics are not usable for this on gfx950: measured (scratch/ubench/
//    lds_atomics.hip) ds_add_f32 = 193 cycles per wave instruction per CU against 5.2 for
//    ds_add_u32 -- the f32 form is serialised per lane.  The windows therefore accumulate in
//    32-bit FIXED POINT with integer atomics: scale = 2^30 / (max|grad_out| * sum|attn|) over
//    the workgroup's own queries (a proven bound of any cell's magnitude, so no overflow),
//    i.e. >= 19 bits below that bound per contribution at ~1000 queries; the sum itself is
//    exact, hence independent of the arrival order (bitwise reproducible inside a window).
//  * lane = (point parity, corner, channel lane): ONE ds_add per two sampling points and
//    channel, and with the window width == 2 (mod 4) and the channel-plane stride == 4 (mod 32)
//    the 32 lanes of a half-wave (4 corners x 8 channel lanes) hit 32 distinct banks.
//  * the per-point arithmetic (floor, weights, window address) is done once per point by one
//    lane and broadcast through the wave's LDS slice.
// One launch per LEVEL GROUP (variant).  A workgroup owns (image, region, head, channel group)
// and only the sampling points of its levels, so the per-point arithmetic (floor, bilinear
// weights, window address) is done 20 times per (query, head) over the three launches instead
// of 64 with one 8-channel slice per workgroup.  At 100x167 / 50x84 / 25x42 / 13x21 (regions of
// 25x28 level-0 pixels, all the same size so the workgroups of a launch are balanced):
//   variant 0: level 0, 16 channels (two groups)    window 42x37        -> 100 KB, 16 waves
//   variant 1: level 1, 32 channels                 window 30x25        ->  99 KB, 12 waves
//   variant 2: levels 2+3, 32 channels              windows 22x19+18x16 ->  95 KB, 16 waves
// The main loop is bound by the LDS pipeline: 2048 integer atomic lane-adds per (query, head)
// = 32 ds_add_u32 wave instructions at ~5.2 cycles each, plus one 8-byte record read per K adds.
struct VarGeom {
  int lv0, nlv;              // first level, number of levels handled
  int base[kMaxLevels];      // first window position of each handled level
  int npos, NP;              // window positions, channel-plane stride (incl. 8 dummy slots)
  int waves;                 // workgroup size in waves (host side only)
};

// NE (4 or 8) consecutive grad_out elements as floats, with vector loads (rows are 16-byte
// aligned: 256 elements per (query) row, channel offsets multiples of NE)
template <typename T, int NE>
__device__ __forceinline__ void load_row(const T* p, float* f) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int k = 0; k < NE / 4; ++k) {
      const f32x4 v = reinterpret_cast<const f32x4*>(p)[k];
      f[4 * k] = v.x; f[4 * k + 1] = v.y; f[4 * k + 2] = v.z; f[4 * k + 3] = v.w;
    }
  } else if constexpr (NE == 8) {
    unpack_bf16x8(*reinterpret_cast<const u32x4*>(p), f);
  } else {
    const i32x2 v = *reinterpret_cast<const i32x2*>(p);
    const unsigned lo = (unsigned)v.x, hi = (unsigned)v.y;
    f[0] = as_f32(lo << 16); f[1] = as_f32(lo & 0xFFFF0000u);
    f[2] = as_f32(hi << 16); f[3] = as_f32(hi & 0xFFFF0000u);
  }
}

#ifdef DSKD_VALUE_PROFILE
__device__ unsigned long long g_vprof[64];
#define VPROF(slot) do { if (tid == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_vprof[vg.lv0 * 8 + (slot)], t_ - t_prev); t_prev = t_; } } while (0)
#else
#define VPROF(slot)
#endif

// FUSED (the levels-2+3 launch, bf16): the kernel also forms the grad_loc / grad_attn dot products of ITS samples, so the
// gather kernel keeps the fine levels only.  The scatter loop is paced by the LDS atomic pipe and leaves the texture path
// idle: the four corner rows of a sample (64 B per head, L2-resident: levels 2+3 are 1 323 pixels) are fetched with
// buffer loads -- lane = (point parity, corner, 8-byte part of the row), i.e. the SAME lane roles as the ds_add loop, with
// the lane's channels now contiguous (4 cl + k instead of cl + 8 k; channel-plane stride == 1 (mod 32) keeps the 32 lanes
// of a half-wave on 32 banks) -- issued before the ds_add loop of the pass and consumed behind it: dot over the lane's
// 4 channels, DPP sum over the 8 part-lanes, the 4 x 8 corner results of a pass parked in LDS; then lane = (query, point)
// finishes (gx, gy, ga) with the (lx, ly, attn) it still holds from phase 1 and writes its 12 bytes; the 8 points of a
// (query, head) are 8 consecutive lanes: whole 32-B / 64-B runs.
// stats != nullptr: the fixed-point bound comes from the gather kernel's by-product (max |grad_out|, sum |attn| per
// (16 x 16 region, head)) instead of a pre-pass over the region's rows.
template <typename T, int NCH, int PPQ, int NW, bool FUSED>
__global__ __launch_bounds__(NW * 64) void msda_bwd_value_kernel(
    const T* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn,
    const T* __restrict__ grad_out, float* __restrict__ grad_value, float* __restrict__ grad_loc,
    float* __restrict__ grad_attn, const float* __restrict__ stats, int sRX, int sRY, int sEX, int sEY, ValueGeom g,
    VarGeom vg, int Nq, int LP, int points) {
  static_assert(!FUSED || (sizeof(T) == 2 && NCH == kCh), "fused dot products: bf16, whole heads");
  constexpr int NQW = 64 / PPQ;           // queries per wave pass
  constexpr int NG = kCh / NCH;           // channel groups per head
  extern __shared__ float smem[];
  int* win = reinterpret_cast<int*>(smem);                              // [NCH][NP] fixed point
  int* s_rec = win + NCH * vg.NP;                                       // [waves][64][4]{off, w}
  float* s_g = reinterpret_cast<float*>(s_rec + NW * 64 * 8);           // [waves][NQW][NCH]
  float* s_red = s_g + NW * NQW * NCH;                             // [2 * waves]
  i32x4* s_tab = reinterpret_cast<i32x4*>(s_red + 2 * NW);              // [4][4] lookup rows
  int* s_geo = reinterpret_cast<int*>(s_tab + 4 * kMaxLevels);     // [24] region geometry
  i32x4* s_voff = reinterpret_cast<i32x4*>(s_geo + 6 * kMaxLevels + 8);   // FUSED: [waves][64] global corner offsets

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef DSKD_VALUE_PROFILE
  unsigned long long t_prev = wall_clock64();
#endif

  int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int cg = vb % NG; vb /= NG;
  const int h = vb & 7; vb >>= 3;
  const int rx = vb % g.RX; vb /= g.RX;
  const int ry = vb % g.RY;
  const int b = vb / g.RY;
  const int chbase = h * kCh + cg * NCH;

  // Region geometry.  The 24 integer divisions behind it (4 levels x {query range begin / end
  // in x and y, window origin in x and y}) run on 24 lanes of wave 0 at once instead of 24 times
  // in sequence on every wave; the rows of the lookup table are built from the results.
  //   s_tab[4l+0] = {cum, qxa, qya, qdx}   [4l+1] = {1/qdx, start, W, H}
  //   s_tab[4l+2] = {ww, wh, wx0, wy0}     [4l+3] = {window base, nq, 0, 0}
  if (wave == 0) {
    if (lane < 6 * kMaxLevels) {
      const int l = lane / 6, kind = lane - 6 * l;   // 0,1: x begin/end  2,3: y begin/end  4,5: origin x/y
      const bool xaxis = kind == 0 || kind == 1 || kind == 4;
      int Sl = 1;
#pragma unroll
      for (int k = 0; k < kMaxLevels; ++k)
        if (l == k) Sl = xaxis ? g.W[k] : g.H[k];
      const int S0 = xaxis ? g.W[0] : g.H[0], E = xaxis ? g.EX : g.EY;
      const int r = (xaxis ? rx : ry) + ((kind == 1 || kind == 3) ? 1 : 0);
      const int q = floor_div(2 * E * r * Sl - S0 + (kind < 4 ? 2 * S0 - 1 : 0), 2 * S0);
      s_geo[lane] = kind < 4 ? (q < 0 ? 0 : (q > Sl ? Sl : q)) : q - kMarginLo;   // region_begin | win_origin
    }
    wave_lds_sync();
    int tot = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < kMaxLevels; ++k) {
      if (k == lane) mine = tot;
      if (k < g.levels) tot += (s_geo[6 * k + 1] - s_geo[6 * k]) * (s_geo[6 * k + 3] - s_geo[6 * k + 2]);
    }
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l)
      if (lane == l) {
        const int qxa = s_geo[6 * l], qya = s_geo[6 * l + 2];
        const int qdx = l < g.levels ? s_geo[6 * l + 1] - qxa : 0;
        s_tab[4 * l + 0] = i32x4{mine, qxa, qya, qdx};
        s_tab[4 * l + 1] = i32x4{as_i32(1.0f / (float)(qdx > 0 ? qdx : 1)), g.start[l], g.W[l], g.H[l]};
        s_tab[4 * l + 2] = i32x4{g.ww[l], g.wh[l], s_geo[6 * l + 4], s_geo[6 * l + 5]};
        s_tab[4 * l + 3] = i32x4{vg.base[l], tot, 0, 0};
      }
  }
  for (int i = tid * 4; i < NCH * vg.NP; i += NW * 64 * 4)
    *reinterpret_cast<i32x4*>(win + i) = i32x4{0, 0, 0, 0};
  __syncthreads();
  int cum[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) cum[l] = s_tab[4 * l].x;
  const int nq = s_tab[3].y;

  VPROF(0);
  float* gvb = grad_value + (size_t)b * Nq * (kHeads * kCh) + chbase;
  const T* gob = grad_out + (size_t)b * Nq * (kHeads * kCh) + chbase;
  const int s_first = vg.lv0 * points;          // first of the PPQ sample indices handled here

  // ---- fixed-point scale: bound of any window cell = max|grad_out| * sum|attn| over the
  // region's queries (this channel group, these levels)
  float gmax = 0.f, asum = 0.f;
  if (stats) {
    // from the gather kernel: the (16 x 16-pixel region, head) cells that cover this region's level-0 pixel range (a
    // query's region is its centre's level-0 pixel / edge in both kernels, so the union contains every query of ours)
    if (wave == 0) {
      const int px0 = rx * g.EX, px1 = min((rx + 1) * g.EX, g.W[0]) - 1;
      const int py0 = ry * g.EY, py1 = min((ry + 1) * g.EY, g.H[0]) - 1;
      const int gx0 = px0 / sEX, gx1 = min(px1 / sEX, sRX - 1), gy0 = py0 / sEY, gy1 = min(py1 / sEY, sRY - 1);
      const int nx = gx1 - gx0 + 1, ncell = nx * (gy1 - gy0 + 1);
      for (int i = lane; i < ncell; i += 64) {
        const int iy = i / nx, ix = i - iy * nx;
        const f32x4 v = *reinterpret_cast<const f32x4*>(
            stats + ((((size_t)b * sRY + gy0 + iy) * sRX + gx0 + ix) * kHeads + h) * 4);
        gmax = fmaxf(gmax, v.x);
        asum += vg.lv0 == 1 ? v.y : v.z;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        gmax = fmaxf(gmax, __shfl_xor(gmax, o));
        asum += __shfl_xor(asum, o);
      }
      if (lane == 0) { s_red[0] = gmax; s_red[1] = asum; }
    }
    __syncthreads();
    gmax = s_red[0]; asum = s_red[1];
  } else {
  for (int i0 = tid; i0 < nq * (NCH / 8); i0 += 4 * NW * 64) {   // 8 channels per lane and step,
#pragma unroll                                                    // 4 steps of loads in flight
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * NW * 64;
      const bool ok = i < nq * (NCH / 8);            // out of range: load element 0, ignore it
      const int ii = ok ? i : 0;
      const int qi = ii / (NCH / 8), c8 = ii - qi * (NCH / 8);
      const int qg = region_query(s_tab, cum, qi);
      float f[8];
      load_row<T, 8>(gob + (size_t)qg * (kHeads * kCh) + c8 * 8, f);
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(
          attn + (((size_t)b * Nq + qg) * kHeads + h) * (size_t)LP + s_first + (c8 < PPQ / 4 ? c8 : 0) * 4);
      float m = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(f[k]));
      gmax = ok ? fmaxf(gmax, m) : gmax;
      asum += (ok && c8 < PPQ / 4) ? fabsf(a4.x) + fabsf(a4.y) + fabsf(a4.z) + fabsf(a4.w) : 0.f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    gmax = fmaxf(gmax, __shfl_xor(gmax, o));
    asum += __shfl_xor(asum, o);
  }
  if (lane == 0) { s_red[wave] = gmax; s_red[NW + wave] = asum; }
  __syncthreads();
  gmax = 0.f; asum = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < NW; ++w2) { gmax = fmaxf(gmax, s_red[w2]); asum += s_red[NW + w2]; }
  }
  const float bound = gmax * asum;
  VPROF(1);
  // nothing to scatter from this region (uniform across the workgroup); the fused launch still owes its gradients
  const bool no_scatter = bound == 0.f;
  if (!FUSED && no_scatter) return;
  // NaN / inf gradients: accumulate nothing here and let them through the float fallback below
  const bool fx_ok = bound > 0.f && bound < 3.0e38f;
  const float fx_scale = fx_ok ? 1.0e9f / bound : 0.f;        // 1e9 < 2^30: headroom for rounding
  const float fx_inv = fx_ok ? bound * 1.0e-9f : 0.f;

  int* my_rec = s_rec + wave * 64 * 8;
  float* my_g = s_g + wave * NQW * NCH;
  // phase-2 lane roles: (point parity, corner, channel lane); a lane owns the K channels
  // cl + 8k, so ONE 8-byte record read feeds K ds_add.  A half-wave is one point x 4 corners x
  // 8 channel lanes: window width == 2 (mod 4) spreads the corners over the 4 bank residues,
  // plane stride == 4 (mod 32) spreads the channel lanes -> 32 distinct banks.
  constexpr int K = NCH / 8;
  const int pp = lane >> 5, crn = (lane >> 3) & 3, cl = lane & 7;
  // the lane's K channels: cl + 8 k (plane stride == 4 mod 32), or -- FUSED -- the contiguous 4 cl + k (== 1 mod 32)
  constexpr int CHM = FUSED ? K : 1, CHS = FUSED ? 1 : 8;      // channel of (cl, k) = CHM * cl + CHS * k
  char* win_cl = reinterpret_cast<char*>(win + CHM * cl * vg.NP);
  const int plane8 = vg.NP * 4 * CHS;             // bytes between the planes of the lane's consecutive channels
  const unsigned dummy = (unsigned)(vg.npos + crn + 4 * pp) * 4u;
  const int* rec_lane = my_rec + pp * 8 + crn * 2;

  // Software pipeline: the global loads of pass i+1 (sampling location, attention weight and
  // the query's grad_out slice) are issued before phase 2 of pass i, so their latency overlaps
  // the LDS work instead of heading every pass.
  constexpr int NE = NCH / PPQ;   // grad_out channels staged per lane
  f32x2 n_xy = f32x2{0.f, 0.f};
  float n_a = 0.f, n_g[NE];
  int n_qg = -1;
  auto fetch = [&](int qbase) {
    const int qi = qbase + lane / PPQ;
    n_qg = -1;
    if (qi < nq) {
      n_qg = region_query(s_tab, cum, qi);
      const size_t base = (((size_t)b * Nq + n_qg) * kHeads + h) * (size_t)LP + s_first + (lane & (PPQ - 1));
      n_xy = *reinterpret_cast<const f32x2*>(loc + base * 2);
      n_a = attn[base];
      load_row<T, NE>(gob + (size_t)n_qg * (kHeads * kCh) + (lane & (PPQ - 1)) * NE, n_g);
    }
  };
  fetch(wave * NQW);

  // FUSED: value rows through the texture path; what phase 3 needs of phase 1 stays in registers
  __amdgpu_buffer_rsrc_t vrsrc;
  if constexpr (FUSED)
    vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(value + (size_t)b * Nq * (kHeads * kCh)), 0,
                                              Nq * (kHeads * kCh * (int)sizeof(T)), 0x00020000);
  // dot-product lane roles: (row of the instruction, 16-byte part of the head's 64-byte row); a pass has 64 points x
  // 4 corners = 256 rows = 16 load instructions, the rows of instruction i belong to query i / 2
  const int dpart = lane & 3, drow = lane >> 2;
  const int hb16 = h * (kCh * (int)sizeof(T)) + dpart * 16;
  i32x4* my_voff = s_voff + wave * 64;
  // corner dot products [64 points][4] over the pass's records, which are dead by then; accessed as int like the
  // records themselves (type-based alias analysis would otherwise let the compiler move float stores across int loads)
  int* my_dot = s_rec + wave * 64 * 8;

  for (int qbase = wave * NQW; qbase < nq; qbase += NW * NQW) {
    unsigned long long fbmask;   // points of this pass that left the window
    float p_lx = 0.f, p_ly = 0.f, p_a = 0.f, p_wf = 0.f, p_hf = 0.f;      // FUSED: this lane's point, for phase 3
    const int p_qg = n_qg;
    // ---- phase 1: one lane per sampling point (NQW queries x PPQ points)
    {
      const int s = s_first + (lane & (PPQ - 1));
      // LDS byte offsets; kSkip (and the negative fallback codes) are clamped onto the consumer
      // lane's dummy slot by an unsigned min in phase 2
      i32x4 off = i32x4{kSkip, kSkip, kSkip, kSkip};
      f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
      i32x4 voff = i32x4{kOOB, kOOB, kOOB, kOOB};
      bool is_fb = false;
      const int qg = n_qg;
      if (qg >= 0) {
        const f32x2 xy = n_xy;
        const float a = n_a;
        const int lvl = s / points;
        const i32x4 lb = s_tab[4 * lvl + 1], lc = s_tab[4 * lvl + 2];
        const int H = lb.w, W = lb.z, st = lb.y;
        const float x = xy.x * (float)W - 0.5f;
        const float y = xy.y * (float)H - 0.5f;
        p_a = a; p_wf = (float)W; p_hf = (float)H;
        if (x > -1.f && y > -1.f && x < (float)W && y < (float)H) {
          const float xf = floorf(x), yf = floorf(y);
          const int x0 = (int)xf, y0 = (int)yf;
          const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
          const bool vx0 = x0 >= 0, vx1 = x0 + 1 <= W - 1, vy0 = y0 >= 0, vy1 = y0 + 1 <= H - 1;
          if constexpr (FUSED) {
            p_lx = lx; p_ly = ly;
            const int r00 = (st + y0 * W + x0) * (kHeads * kCh * (int)sizeof(T));
            const int rb = kHeads * kCh * (int)sizeof(T);
            voff = i32x4{(vy0 && vx0) ? r00 : kOOB, (vy0 && vx1) ? r00 + rb : kOOB, (vy1 && vx0) ? r00 + W * rb : kOOB,
                         (vy1 && vx1) ? r00 + W * rb + rb : kOOB};
          }
          w = f32x4{(vy0 && vx0) ? hy * hx * a : 0.f, (vy0 && vx1) ? hy * lx * a : 0.f,
                    (vy1 && vx0) ? ly * hx * a : 0.f, (vy1 && vx1) ? ly * lx * a : 0.f};
          const int wwl = lc.x, whl = lc.y;
          const int wx = x0 - lc.z, wy = y0 - lc.w;
          if (fx_ok && wx >= 0 && wx + 1 < wwl && wy >= 0 && wy + 1 < whl) {
            const int pb = (s_tab[4 * lvl + 3].x + wy * wwl + wx) * 4;
            off = i32x4{pb, pb + 4, pb + wwl * 4, pb + wwl * 4 + 4};
          } else {  // outside the LDS window: direct global atomics, rows encoded as -(2+row)
            const int r00 = st + y0 * W + x0;
            off = i32x4{(vy0 && vx0) ? -(2 + r00) : -1, (vy0 && vx1) ? -(2 + r00 + 1) : -1,
                        (vy1 && vx0) ? -(2 + r00 + W) : -1, (vy1 && vx1) ? -(2 + r00 + W + 1) : -1};
            is_fb = true;
          }
        }
      }
      fbmask = __ballot(is_fb);
      i32x4* rec = reinterpret_cast<i32x4*>(my_rec + lane * 8);   // {off0, w0, off1, w1} {off2, w2, off3, w3}
      rec[0] = i32x4{off.x, as_i32(w.x), off.y, as_i32(w.y)};
      rec[1] = i32x4{off.z, as_i32(w.z), off.w, as_i32(w.w)};
      if constexpr (FUSED) my_voff[lane] = voff;
      // the pass's grad_out rows (this channel group), NE channels per lane
      float* gdst = my_g + (lane / PPQ) * NCH + (lane & (PPQ - 1)) * NE;
#pragma unroll
      for (int e = 0; e < NE; e += 4)
        *reinterpret_cast<f32x4*>(gdst + e) = qg >= 0 ? f32x4{n_g[e], n_g[e + 1], n_g[e + 2], n_g[e + 3]}
                                                       : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    fetch(qbase + NW * NQW);
    wave_lds_sync();

    // ---- FUSED: request this lane's 16 bytes of every (point, corner) row of the pass; consumed behind the ds_add loop
    u32x4 vv[FUSED ? 16 : 1];
    if constexpr (FUSED) {
      const int* vo = reinterpret_cast<const int*>(my_voff) + drow;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        vv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, vo[i * 16] + hb16, 0, 0));
    }
    // ---- phase 2: branch-free accumulation.  Per pair of points: one 8-byte LDS read
    // {offset, weight}, one unsigned min (skipped / out-of-window corners land on the lane's
    // dummy slot), and per channel one multiply, one rounding convert, one ds_add.
    if (!no_scatter)
#pragma unroll
    for (int qk = 0; qk < NQW; ++qk) {
      float gs[K];
#pragma unroll
      for (int k = 0; k < K; ++k) gs[k] = my_g[qk * NCH + CHM * cl + CHS * k] * fx_scale;
      i32x2 r[PPQ / 2];
#pragma unroll
      for (int j = 0; j < PPQ / 2; ++j)
        r[j] = *reinterpret_cast<const i32x2*>(rec_lane + (qk * PPQ + 2 * j) * 8);
#pragma unroll
      for (int j = 0; j < PPQ / 2; ++j) {
        const unsigned o = min((unsigned)r[j].x, dummy);
        const float wj = as_f32((unsigned)r[j].y);
#pragma unroll
        for (int k = 0; k < K; ++k)
          atomicAdd(reinterpret_cast<int*>(win_cl + k * plane8 + o), cvt_round(wj * gs[k]));
      }
    }
    while (fbmask) {   // samples that left the window -> direct global atomics, point by point
      const int pt = __builtin_ctzll(fbmask);
      fbmask &= fbmask - 1;
      const int o = my_rec[pt * 8 + crn * 2];
      if (o <= -2) {
        const float wj = as_f32((unsigned)my_rec[pt * 8 + crn * 2 + 1]);
        for (int k = pp; k < K; k += 2)
          atomicAdd(gvb + (size_t)(-(o + 2)) * (kHeads * kCh) + CHM * cl + CHS * k,
                    wj * my_g[(pt / PPQ) * NCH + CHM * cl + CHS * k]);
      }
    }
    if constexpr (FUSED) {
      // ---- dot products <value[corner row], grad_out[query, head]>: 8 channels per lane as four v_dot2c_f32_bf16 (the
      // products of two bf16 are exact in f32), summed over the row's 4 part-lanes with DPP
#pragma unroll
      for (int qk = 0; qk < NQW; ++qk) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(my_g + qk * NCH + 8 * dpart);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(my_g + qk * NCH + 8 * dpart + 4);
        const bf16x2 g0 = {(__bf16)ga.x, (__bf16)ga.y}, g1 = {(__bf16)ga.z, (__bf16)ga.w};
        const bf16x2 g2 = {(__bf16)gb.x, (__bf16)gb.y}, g3 = {(__bf16)gb.z, (__bf16)gb.w};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const u32x4 v = vv[2 * qk + e];
          // (bit casts of vector ELEMENTS go through a by-value helper: see the note at as_f32)
          float d = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(v.x), g0, 0.f, false);
          d = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(v.y), g1, d, false);
          d = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(v.z), g2, d, false);
          d = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(v.w), g3, d, false);
          d = group4_sum(d);
          if (dpart == 0) my_dot[(2 * qk + e) * 16 + drow] = as_i32(d);
        }
      }
      wave_lds_sync();
      // ---- phase 3: lane = (query, point) again
      if (p_qg >= 0) {
        const i32x4 di = *reinterpret_cast<const i32x4*>(my_dot + lane * 4);
        const f32x4 d = f32x4{as_f32((unsigned)di.x), as_f32((unsigned)di.y), as_f32((unsigned)di.z), as_f32((unsigned)di.w)};
        const float hx = 1.f - p_lx, hy = 1.f - p_ly;
        const float ga = (hy * hx) * d.x + (hy * p_lx) * d.y + (p_ly * hx) * d.z + (p_ly * p_lx) * d.w;
        const float gx = p_wf * p_a * (hy * (d.y - d.x) + p_ly * (d.w - d.z));
        const float gy = p_hf * p_a * (hx * (d.z - d.x) + p_lx * (d.w - d.y));
        const size_t base = (((size_t)b * Nq + p_qg) * kHeads + h) * (size_t)LP + s_first + (lane & (PPQ - 1));
        grad_attn[base] = ga;
        *reinterpret_cast<f32x2*>(grad_loc + base * 2) = f32x2{gx, gy};
      }
    }
    wave_lds_sync();
  }
  if (FUSED && no_scatter) return;
  VPROF(2);
  __syncthreads();
  VPROF(3);

  // ---- flush the non-zero window entries: consecutive lanes = consecutive channels, so every
  // global atomic instruction adds NCH*4-byte contiguous segments
  const int ch = tid & (NCH - 1);
  for (int p = tid / NCH; p < vg.npos; p += (NW * 64) / NCH) {
    const int iv = win[ch * vg.NP + p];
    if (iv == 0) continue;
    const float v = (float)iv * fx_inv;
    int l = vg.lv0;
#pragma unroll
    for (int k = 1; k < kMaxLevels; ++k) l += (k > vg.lv0 && k < vg.lv0 + vg.nlv && p >= vg.base[k]) ? 1 : 0;
    const i32x4 lb = s_tab[4 * l + 1], lc = s_tab[4 * l + 2];
    const int rel = p - s_tab[4 * l + 3].x;
    const int wwl = lc.x;
    const int wy = (int)(((float)rel + 0.5f) / (float)wwl), wx = rel - wy * wwl;
    const int gx = lc.z + wx, gy = lc.w + wy;
    if (gx >= 0 && gx < lb.z && gy >= 0 && gy < lb.w)
      atomicAdd(gvb + (size_t)(lb.y + gy * lb.z + gx) * (kHeads * kCh) + ch, v);
  }
  VPROF(4);
}

// Host side of the windowed kernels: per-level window extents, the three level-group variants
// and their LDS budgets.  Returns false when the geometry does not fit (then the plain-atomics
// kernel is used).
constexpr int kNumVar = 3;
constexpr int kVarNch[kNumVar] = {16, 32, 32};
constexpr int kVarPpq[kNumVar] = {4, 4, 8};
constexpr size_t kMaxLds = 160 * 1024;

inline size_t value_lds_bytes(int nch, int ppq, int nw, int NP, bool fused) {
  const int nqw = 64 / ppq;
  return sizeof(int) * (size_t)nch * NP + sizeof(int) * nw * 64 * 8 +
         sizeof(float) * nw * nqw * nch + sizeof(float) * 2 * nw +
         sizeof(int) * 4 * 4 * kMaxLevels + sizeof(int) * (6 * kMaxLevels + 8) +
         (fused ? (size_t)nw * 64 * 16 : 0);       // global corner offsets of a pass
}

// fuse23: the levels-2+3 launch also forms its samples' grad_loc / grad_attn (channel planes then == 1 mod 32)
bool make_value_geom(const LevelGeom& lg, int levels, int points, int Nq, ValueGeom* g,
                     VarGeom* var, size_t* lds_bytes, bool fuse23 = false) {
  if (levels != 4 || points != 4) return false;
  int tot = 0;
  for (int l = 0; l < levels; ++l) tot += lg.H[l] * lg.W[l];
  if (tot != Nq) return false;
  for (int l = 0; l < levels; ++l)
    if (lg.start[l] != (l == 0 ? 0 : lg.start[l - 1] + lg.H[l - 1] * lg.W[l - 1])) return false;
  const int W0 = lg.W[0], H0 = lg.H[0];
  g->levels = levels;
  g->RX = (W0 + kRegion - 1) / kRegion;
  g->RY = (H0 + kRegion - 1) / kRegion;
  // equal-sized regions: every workgroup of a launch gets the same number of queries
  g->EX = (W0 + g->RX - 1) / g->RX;
  g->EY = (H0 + g->RY - 1) / g->RY;
  if (g->RX > kMaxReg || g->RY > kMaxReg) return false;
  for (int l = 0; l < levels; ++l) {
    if (lg.W[l] > W0 || lg.H[l] > H0) return false;   // level 0 must be the finest
    g->H[l] = lg.H[l]; g->W[l] = lg.W[l]; g->start[l] = lg.start[l];
    g->ww[l] = (g->EX * lg.W[l] + W0 - 1) / W0 + 1 + kMarginLo + kMarginHi;
    while ((g->ww[l] & 3) != 2) ++g->ww[l];             // bank spread of the 4 corners
    g->wh[l] = (g->EY * lg.H[l] + H0 - 1) / H0 + 1 + kMarginLo + kMarginHi;
  }
  const int lv0[kNumVar] = {0, 1, 2}, nlv[kNumVar] = {1, 1, 2};
  for (int v = 0; v < kNumVar; ++v) {
    VarGeom& vg = var[v];
    vg.lv0 = lv0[v]; vg.nlv = nlv[v];
    int npos = 0;
    for (int l = 0; l < kMaxLevels; ++l) {
      const bool mine = l >= lv0[v] && l < lv0[v] + nlv[v];
      vg.base[l] = mine ? npos : 0x3FFFFFFF;
      if (mine) npos += g->ww[l] * g->wh[l];
    }
    vg.npos = npos;
    // channel-plane stride (incl. 8 dummy slots): == 4 (mod 32), see the kernel's lane roles (fused launch: == 1)
    const bool fused = fuse23 && v == 2;
    int NP = npos + 8;
    while ((NP & 31) != (fused ? 1 : 4)) ++NP;
    vg.NP = NP;
    // as many waves as the LDS left beside the window allows (each wave owns a parameter slice)
    vg.waves = 0;
    for (int nw : {16, 12, 8}) {
      lds_bytes[v] = (value_lds_bytes(kVarNch[v], kVarPpq[v], nw, NP, fused) + 15) & ~(size_t)15;
      if (lds_bytes[v] <= kMaxLds) { vg.waves = nw; break; }
    }
    if (vg.waves == 0) return false;
  }
  return true;
}

// What the fused / statistics-fed launches need beyond the scatter's own arguments.
struct ValueExtra {
  const void* value;      // fused launch: the value tensor
  float* grad_loc;
  float* grad_attn;
  const float* stats;     // per (gather region, head) {max |grad_out|, sum |attn| level 1, levels 2+3, -} or null
  int sRX, sRY, sEX, sEY; // the gather kernel's region grid
  bool fuse23;
};

template <typename T, int V, int NW, bool FUSED>
int launch_value_nw(const float* loc, const float* attn, const T* grad_out, float* grad_value,
                    const ValueGeom& g, const VarGeom& vg, size_t lds, int B, int Nq, int LP,
                    int points, const ValueExtra& ex, hipStream_t st) {
  constexpr int NCH = kVarNch[V], PPQ = kVarPpq[V];
  auto kern = msda_bwd_value_kernel<T, NCH, PPQ, NW, FUSED>;
  int dev = 0;
  static bool done[64] = {};              // the attribute is per device: set it once on each
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!done[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds) != hipSuccess)
      return fail(DSKD_ERR_LAUNCH, "dskd_msda_bwd: cannot reserve LDS");
    done[dev] = true;
  }
  const dim3 grid((unsigned)(B * g.RY * g.RX * kHeads * (kCh / NCH))), block(NW * 64);
  const float* stats = V == 0 ? nullptr : ex.stats;       // the by-product has no level-0 sum
  hipLaunchKernelGGL(kern, grid, block, lds, st, (const T*)ex.value, loc, attn, grad_out, grad_value, ex.grad_loc,
                     ex.grad_attn, stats, ex.sRX, ex.sRY, ex.sEX, ex.sEY, g, vg, Nq, LP, points);
  return DSKD_OK;
}

template <typename T, int V>
int launch_value_variant(const float* loc, const float* attn, const T* grad_out, float* grad_value,
                         const ValueGeom& g, const VarGeom& vg, size_t lds, int B, int Nq, int LP,
                         int points, const ValueExtra& ex, hipStream_t st) {
  if constexpr (V == 2 && sizeof(T) == 2) {
    if (ex.fuse23) switch (vg.waves) {
      case 16: return launch_value_nw<T, V, 16, true>(loc, attn, grad_out, grad_value, g, vg, lds, B, Nq, LP, points, ex, st);
      case 12: return launch_value_nw<T, V, 12, true>(loc, attn, grad_out, grad_value, g, vg, lds, B, Nq, LP, points, ex, st);
      default: return launch_value_nw<T, V, 8, true>(loc, attn, grad_out, grad_value, g, vg, lds, B, Nq, LP, points, ex, st);
    }
  }
  switch (vg.waves) {
    case 16: return launch_value_nw<T, V, 16, false>(loc, attn, grad_out, grad_value, g, vg, lds, B, Nq, LP, points, ex, st);
    case 12: return launch_value_nw<T, V, 12, false>(loc, attn, grad_out, grad_value, g, vg, lds, B, Nq, LP, points, ex, st);
    default: return launch_value_nw<T, V, 8, false>(loc, attn, grad_out, grad_value, g, vg, lds, B, Nq, LP, points, ex, st);
  }
}

// variants: bit v set = launch level group v (0: level 0, 1: level 1, 2: levels 2+3)
template <typename T>
int launch_value(const float* loc, const float* attn, const T* grad_out, float* grad_value,
                 const ValueGeom& g, const VarGeom* var, const size_t* lds, int B, int Nq, int LP,
                 int points, const ValueExtra& ex, hipStream_t st, int variants = 7) {
  if (variants & 1)
    if (int rc = launch_value_variant<T, 0>(loc, attn, grad_out, grad_value, g, var[0], lds[0], B, Nq, LP, points, ex, st)) return rc;
  if (variants & 2)
    if (int rc = launch_value_variant<T, 1>(loc, attn, grad_out, grad_value, g, var[1], lds[1], B, Nq, LP, points, ex, st)) return rc;
  if (variants & 4)
    return launch_value_variant<T, 2>(loc, attn, grad_out, grad_value, g, var[2], lds[2], B, Nq, LP, points, ex, st);
  return DSKD_OK;
}

// rows [row0, row0 + nrows) of every image of grad_value ([B, Nv, 256] f32) = 0
__global__ void zero_rows_kernel(float* __restrict__ gv, int Nv, int row0, int nrows, unsigned* __restrict__ hdr) {
  // hdr: the stray-list header of the workspace, zeroed at the START of every backward (a launch that failed half-way
  // must not leave a count behind for the next one)
  if (hdr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 16) hdr[threadIdx.x] = 0u;
  u32x4* base = reinterpret_cast<u32x4*>(gv + ((size_t)blockIdx.y * Nv + row0) * (kHeads * kCh));
  const size_t n16 = (size_t)nrows * (kHeads * kCh) / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    base[i] = u32x4{0u, 0u, 0u, 0u};
}

// Levels whose grad_value rows the tiled pull kernel (msda_pull.hip) produces; the others stay on the windowed
// LDS-accumulation kernels.  Default: level 0 (75 % of the pixels, ~21 contributions per cell -> gather form: 124 us
// against 197 us at B=4); from level 1 on a cell sums 85 / 340 / 1 300 contributions and accumulating in LDS windows
// (scatter form) is faster (level 1: 175 us windowed, 209 us pulled -- profiles/r02_msda_bwd_pull_ab.txt).
// DSKD_MSDA_PULL_LEVELS=<digits> overrides ("" or "none": no pull; levels 2 and 3 only together).
// Levels on the matrix-core kernel (msda_mm.hip).  DSKD_MSDA_MM=<digits> overrides for A/B runs ("0": none).
inline int mm_level_mask() {
  int mask = 2 | 4;
  if (const char* e = getenv("DSKD_MSDA_MM")) {
    mask = 0;
    for (const char* c = e; *c; ++c) {
      if (*c == '1') mask |= 2;
      if (*c == '2' || *c == '3') mask |= 4;
    }
  }
  return mask;
}

inline int pull_level_mask() {
  int mask = 1;
  if (const char* e = getenv("DSKD_MSDA_PULL_LEVELS")) {
    mask = 0;
    for (const char* c = e; *c; ++c)
      if (*c >= '0' && *c <= '3') mask |= 1 << (*c - '0');
  }
  if (((mask >> 2) & 3) != 0 && ((mask >> 2) & 3) != 3) mask &= 3;
  return mask;
}

// ------------------------------------------------------------------ forward, windowed (encoder, bf16)
// The plain forward above runs at the chip's L2 -> L1 gather rate (profiles/r01_msda_phased_staging_ab.txt,
// r01_msda_layout_ab.txt: ~2.4 shader cycles per 64-B segment and CU whether it hits L1 or not, 16 TB/s
// chip-wide): every (query, head) pulls 64 corners x 64 B = 4 KB through the texture path.  In the
// encoder the queries are the pixels and sample near their own position, so here a workgroup owns
// (image, region of <= 16x16 level-0 pixels, ONE head), copies that head's value windows of all four
// levels (region footprint + the 5/6-pixel margins of the windowed backward) into LDS ONCE --
// ~5 pixels of 64 B per (query, head) instead of 64 segments -- and gathers from LDS (256 B/clk/CU
// against ~27 B/clk/CU).  Window pixels outside the image are zeros (= the zero padding); a sample
// that leaves the window takes the plain buffer-load path, so the result is exact for ANY location;
// a rejected sample reads a zero slot.  Same weights, same order of the 16 samples, same FMAs as the
// plain kernel: bit-identical output.
// Measured at B=4, 100x167 (profiles/r01_msda_fwd_windowed_ab.json): with ALL four levels in LDS the 104-KB
// window leaves one workgroup per CU and the kernel is slower than the plain one (177-219 vs 162 us);
// holding only the coarse levels (lv0 = 2: levels 2+3, 31 KB) and leaving the fine ones on the
// buffer-load path lets two 8-wave workgroups share a CU and the texture path and the LDS work side by
// side: 138 us, -15 %.  Experimental (DSKD_MSDA_FWD=win); not the default yet.
//   lane = (query of the pass, 16-B part of the head's 64-B line): 16 queries per wave pass;
//   the 16 samples are staged level by level (4 points per query: one per lane), so the level
//   geometry of a staging step is wave-uniform.
struct FwdWinGeom {
  int base[kMaxLevels];   // first window pixel of each level
  int npos;               // window pixels over all levels (the zero slot follows)
  int lv0;                // first level held in LDS; levels below it (the finest, with the largest windows) stay on
                          // the buffer-load path, which then runs beside the LDS gather instead of idling
  int waves;              // workgroup size in waves (host side)
  int lv_end;             // backward gather only: levels [0, lv_end) are processed here (the coarser ones have their dot
                          // products formed inside the windowed grad_value kernel, which holds their windows anyway)
};
constexpr int kRegionF = 16;
constexpr int kFwdHS = 5;       // staging slots per query: 4 points + 1 pad
constexpr int kFwdSign = (int)0x80000000;

template <typename T>
__global__ __launch_bounds__(1024) void msda_fwd_win_kernel(
    const T* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn,
    T* __restrict__ out, ValueGeom g, FwdWinGeom fw, int Nq, int points) {
  static_assert(sizeof(T) == 2, "windowed forward: bf16 only");
  constexpr int PIXB = kCh * (int)sizeof(T);     // 64 B: one head of one pixel
  constexpr int ROWB = kHeads * PIXB;            // 512 B: one pixel, all heads
  constexpr int LP = 16;
  extern __shared__ float smem[];
  char* win = reinterpret_cast<char*>(smem);                                    // [npos + 1][PIXB]
  const int NW = blockDim.x >> 6;
  i32x4* s_off_all = reinterpret_cast<i32x4*>(win + (size_t)(fw.npos + 1) * PIXB);   // [NW][16 * kFwdHS]
  f32x4* s_wt_all = reinterpret_cast<f32x4*>(s_off_all + NW * 16 * kFwdHS);
  i32x4* s_tab = reinterpret_cast<i32x4*>(s_wt_all + NW * 16 * kFwdHS);             // [4][4] lookup rows
  int* s_geo = reinterpret_cast<int*>(s_tab + 4 * kMaxLevels);                       // [24] region geometry

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int h = vb & 7; vb >>= 3;
  const int rx = vb % g.RX; vb /= g.RX;
  const int ry = vb % g.RY;
  const int b = vb / g.RY;

  // Region geometry: as in msda_bwd_value_kernel (same tables, same integer arithmetic).
  //   s_tab[4l+0] = {cum, qxa, qya, qdx}   [4l+1] = {1/qdx, start, W, H}
  //   s_tab[4l+2] = {ww, wh, wx0, wy0}     [4l+3] = {window base, nq, 0, 0}
  if (wave == 0) {
    if (lane < 6 * kMaxLevels) {
      const int l = lane / 6, kind = lane - 6 * l;   // 0,1: x begin/end  2,3: y begin/end  4,5: origin x/y
      const bool xaxis = kind == 0 || kind == 1 || kind == 4;
      int Sl = 1;
#pragma unroll
      for (int k = 0; k < kMaxLevels; ++k)
        if (l == k) Sl = xaxis ? g.W[k] : g.H[k];
      const int S0 = xaxis ? g.W[0] : g.H[0], E = xaxis ? g.EX : g.EY;
      const int r = (xaxis ? rx : ry) + ((kind == 1 || kind == 3) ? 1 : 0);
      const int q = floor_div(2 * E * r * Sl - S0 + (kind < 4 ? 2 * S0 - 1 : 0), 2 * S0);
      s_geo[lane] = kind < 4 ? (q < 0 ? 0 : (q > Sl ? Sl : q)) : q - kMarginLo;   // region_begin | win_origin
    }
    wave_lds_sync();
    int tot = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < kMaxLevels; ++k) {
      if (k == lane) mine = tot;
      if (k < g.levels) tot += (s_geo[6 * k + 1] - s_geo[6 * k]) * (s_geo[6 * k + 3] - s_geo[6 * k + 2]);
    }
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l)
      if (lane == l) {
        const int qxa = s_geo[6 * l], qya = s_geo[6 * l + 2];
        const int qdx = l < g.levels ? s_geo[6 * l + 1] - qxa : 0;
        s_tab[4 * l + 0] = i32x4{mine, qxa, qya, qdx};
        s_tab[4 * l + 1] = i32x4{as_i32(1.0f / (float)(qdx > 0 ? qdx : 1)), g.start[l], g.W[l], g.H[l]};
        s_tab[4 * l + 2] = i32x4{g.ww[l], g.wh[l], s_geo[6 * l + 4], s_geo[6 * l + 5]};
        s_tab[4 * l + 3] = i32x4{fw.base[l], tot, 0, 0};
      }
  }
  __syncthreads();
  int cum[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) cum[l] = s_tab[4 * l].x;
  const int nq = s_tab[3].y;

  const T* vbase = value + (size_t)b * Nq * (kHeads * kCh);      // Nq == Nv
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(vbase), 0, Nq * ROWB, 0x00020000);
  const int part = lane & 3;
  const int hb = h * PIXB + part * 16;

  // ---- window fill: 4 lanes per window pixel, 16 B each; outside the image (and the zero slot): zeros.
  // kFill loads are in flight per lane before the first LDS store (a load -> store loop pays one
  // memory latency per iteration: ~10 iterations here).
  {
    constexpr int kFill = 5;
    const int step = blockDim.x >> 2;
    for (int p0 = tid >> 2; p0 <= fw.npos; p0 += kFill * step) {
      u32x4 v[kFill];
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const int p = p0 + u * step;
        int l = fw.lv0;
#pragma unroll
        for (int k = 1; k < kMaxLevels; ++k) l += (k > fw.lv0 && p >= fw.base[k]) ? 1 : 0;
        const i32x4 lb = s_tab[4 * l + 1], lc = s_tab[4 * l + 2];
        const int rel = p - s_tab[4 * l + 3].x;
        const int wwl = lc.x;
        const int wy = (int)(((float)rel + 0.5f) / (float)wwl), wx = rel - wy * wwl;
        const int gx = lc.z + wx, gy = lc.w + wy;
        const bool in = p < fw.npos && gx >= 0 && gx < lb.z && gy >= 0 && gy < lb.w;
        const int goff = in ? (lb.y + gy * lb.z + gx) * ROWB + hb : kOOB;
        v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const int p = p0 + u * step;
        if (p <= fw.npos) *reinterpret_cast<u32x4*>(win + (size_t)p * PIXB + part * 16) = v[u];
      }
    }
  }
  __syncthreads();

  i32x4* s_off = s_off_all + wave * 16 * kFwdHS;
  f32x4* s_wt = s_wt_all + wave * 16 * kFwdHS;
  const int ql = lane >> 2;                  // query of the pass
  const int zero_slot = fw.npos * PIXB;

  // Software pipeline: the sampling locations / attention weights of the NEXT pass (this lane's point on
  // each of the four levels) are requested before the current pass is consumed.
  f32x2 n_xy[kMaxLevels];
  float n_a[kMaxLevels];
  int n_qg = -1;
  auto fetch = [&](int qb) {
    const int qi = qb + ql;
    n_qg = qi < nq ? region_query(s_tab, cum, qi) : -1;
    if (n_qg >= 0) {
      const size_t base = (((size_t)b * Nq + n_qg) * kHeads + h) * (size_t)LP + part;
#pragma unroll
      for (int l = 0; l < kMaxLevels; ++l) {
        n_xy[l] = *reinterpret_cast<const f32x2*>(loc + (base + l * points) * 2);
        n_a[l] = attn[base + l * points];
      }
    }
  };
  fetch(wave * 16);

  for (int qbase = wave * 16; qbase < nq; qbase += NW * 16) {
    const int qg = n_qg;
    f32x2 c_xy[kMaxLevels];
    float c_a[kMaxLevels];
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l) { c_xy[l] = n_xy[l]; c_a[l] = n_a[l]; }
    fetch(qbase + NW * 16);
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;

#pragma unroll
    for (int lvl = 0; lvl < kMaxLevels; ++lvl) {
      // ---- stage the 4 points of this level: lane = (query, point)
      bool any_fb;
      {
        i32x4 off = i32x4{zero_slot, zero_slot, zero_slot, zero_slot};
        f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
        if (qg >= 0) {
          const f32x2 xy = c_xy[lvl];
          const float a = c_a[lvl];
          const i32x4 lb = s_tab[4 * lvl + 1], lc = s_tab[4 * lvl + 2];
          const int H = lb.w, W = lb.z, st = lb.y;
          const float x = xy.x * (float)W - 0.5f;
          const float y = xy.y * (float)H - 0.5f;
          if (x > -1.f && y > -1.f && x < (float)W && y < (float)H) {   // as point_params
            const float xf = floorf(x), yf = floorf(y);
            const int x0 = (int)xf, y0 = (int)yf;
            const float lx = x - xf, ly = y - yf;
            const float hx = 1.f - lx, hy = 1.f - ly;
            w = f32x4{hy * hx, hy * lx, ly * hx, ly * lx} * a;
            const int wwl = lc.x, whl = lc.y;
            const int wx = x0 - lc.z, wy = y0 - lc.w;
            if (lvl >= fw.lv0 && wx >= 0 && wx + 1 < wwl && wy >= 0 && wy + 1 < whl) {
              const int pb = (s_tab[4 * lvl + 3].x + wy * wwl + wx) * PIXB;
              off = i32x4{pb, pb + PIXB, pb + wwl * PIXB, pb + wwl * PIXB + PIXB};
            } else {      // left the window: global byte offsets, flagged by the sign bit
              const bool vx0 = x0 >= 0, vx1 = x0 + 1 <= W - 1;
              const bool vy0 = y0 >= 0, vy1 = y0 + 1 <= H - 1;
              const int r00 = (st + y0 * W + x0) * ROWB;
              off.x = ((vy0 && vx0) ? r00 : kOOB) | kFwdSign;
              off.y = ((vy0 && vx1) ? r00 + ROWB : kOOB) | kFwdSign;
              off.z = ((vy1 && vx0) ? r00 + W * ROWB : kOOB) | kFwdSign;
              off.w = ((vy1 && vx1) ? r00 + W * ROWB + ROWB : kOOB) | kFwdSign;
            }
          }
        }
        s_off[ql * kFwdHS + part] = off;
        s_wt[ql * kFwdHS + part] = w;
        any_fb = __ballot(off.x < 0) != 0ull;
      }
      wave_lds_sync();
      // ---- consume: lane = (query, 16-B part)
      if (!any_fb) {    // wave-uniform: every sample of this step is inside the windows -> LDS only, no lane branches
#pragma unroll 2
        for (int sl = 0; sl < 4; ++sl) {
          const i32x4 o = s_off[ql * kFwdHS + sl];
          const f32x4 w = s_wt[ql * kFwdHS + sl];
#if DSKD_FWD_ABLATE == 2
          const u32x4 r0 = fwd_fake_operand(o.x + part), r1 = fwd_fake_operand(o.y + part), r2 = fwd_fake_operand(o.z + part),
                      r3 = fwd_fake_operand(o.w + part);
#else
          const u32x4 r0 = *reinterpret_cast<const u32x4*>(win + o.x + part * 16);
          const u32x4 r1 = *reinterpret_cast<const u32x4*>(win + o.y + part * 16);
          const u32x4 r2 = *reinterpret_cast<const u32x4*>(win + o.z + part * 16);
          const u32x4 r3 = *reinterpret_cast<const u32x4*>(win + o.w + part * 16);
#endif
          fwd_fma4_bf16(acc, r0, r1, r2, r3, w);
        }
      } else
#pragma unroll 2
      for (int sl = 0; sl < 4; ++sl) {
        const i32x4 o = s_off[ql * kFwdHS + sl];
        const f32x4 w = s_wt[ql * kFwdHS + sl];
        u32x4 r0, r1, r2, r3;
        if (o.x < 0) {
          r0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (o.x & 0x7FFFFFFF) + hb, 0, 0));
          r1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (o.y & 0x7FFFFFFF) + hb, 0, 0));
          r2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (o.z & 0x7FFFFFFF) + hb, 0, 0));
          r3 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (o.w & 0x7FFFFFFF) + hb, 0, 0));
        } else {
          r0 = *reinterpret_cast<const u32x4*>(win + o.x + part * 16);
          r1 = *reinterpret_cast<const u32x4*>(win + o.y + part * 16);
          r2 = *reinterpret_cast<const u32x4*>(win + o.z + part * 16);
          r3 = *reinterpret_cast<const u32x4*>(win + o.w + part * 16);
        }
#if DSKD_FWD_ABLATE == 2
        r0 = fwd_fake_operand(o.x + part); r1 = fwd_fake_operand(o.y + part); r2 = fwd_fake_operand(o.z + part);
        r3 = fwd_fake_operand(o.w + part);
#endif
        fwd_fma4_bf16(acc, r0, r1, r2, r3, w);
      }
      wave_lds_sync();   // the next level's staging overwrites the slots
    }

    if (qg >= 0) {
      T* orow = out + ((size_t)b * Nq + qg) * (kHeads * kCh) + h * kCh;
      u32x4 p;
      p.x = pack_bf16x2(acc[0], acc[1]);
      p.y = pack_bf16x2(acc[2], acc[3]);
      p.z = pack_bf16x2(acc[4], acc[5]);
      p.w = pack_bf16x2(acc[6], acc[7]);
      *reinterpret_cast<u32x4*>(orow + part * 8) = p;
    }
  }
}

// Host side: regions of <= kRegionF level-0 pixels, all four windows of one head in LDS.
bool make_fwd_win_geom(const LevelGeom& lg, int levels, int points, int Nq, int lv0, int nw_req, ValueGeom* g,
                       FwdWinGeom* fw, size_t* lds_bytes, int lv_end = kMaxLevels) {
  if (levels != 4 || points != 4) return false;
  int tot = 0;
  for (int l = 0; l < levels; ++l) tot += lg.H[l] * lg.W[l];
  if (tot != Nq) return false;
  for (int l = 0; l < levels; ++l)
    if (lg.start[l] != (l == 0 ? 0 : lg.start[l - 1] + lg.H[l - 1] * lg.W[l - 1])) return false;
  const int W0 = lg.W[0], H0 = lg.H[0];
  g->levels = levels;
  g->RX = (W0 + kRegionF - 1) / kRegionF;
  g->RY = (H0 + kRegionF - 1) / kRegionF;
  g->EX = (W0 + g->RX - 1) / g->RX;
  g->EY = (H0 + g->RY - 1) / g->RY;
  int npos = 0, nq_max = 0;
  for (int l = 0; l < levels; ++l) {
    if (lg.W[l] > W0 || lg.H[l] > H0) return false;   // level 0 must be the finest
    g->H[l] = lg.H[l]; g->W[l] = lg.W[l]; g->start[l] = lg.start[l];
    g->ww[l] = (g->EX * lg.W[l] + W0 - 1) / W0 + 1 + kMarginLo + kMarginHi;
    g->wh[l] = (g->EY * lg.H[l] + H0 - 1) / H0 + 1 + kMarginLo + kMarginHi;
    fw->base[l] = npos;
    if (l >= lv0 && l < lv_end) npos += g->ww[l] * g->wh[l];
  }
  fw->npos = npos;
  fw->lv0 = lv0;
  fw->lv_end = lv_end;
  // queries of the largest region (the kernel's own integer arithmetic)
  auto edge = [](int E, int r, int Sl, int S0) {
    const int q = floor_div(2 * E * r * Sl - S0 + 2 * S0 - 1, 2 * S0);
    return q < 0 ? 0 : (q > Sl ? Sl : q);
  };
  for (int ry = 0; ry < g->RY; ++ry)
    for (int rx = 0; rx < g->RX; ++rx) {
      int nq = 0;
      for (int l = 0; l < levels; ++l)
        nq += (edge(g->EX, rx + 1, lg.W[l], W0) - edge(g->EX, rx, lg.W[l], W0)) *
              (edge(g->EY, ry + 1, lg.H[l], H0) - edge(g->EY, ry, lg.H[l], H0));
      if (nq > nq_max) nq_max = nq;
    }
  // as few rounds of 16-query wave passes as possible, then as few waves as cover them
  const int passes = (nq_max + 15) / 16;
  const int rounds = (passes + 15) / 16;
  int nw = (passes + rounds - 1) / rounds;
  if (nw_req > 0) nw = nw_req;
  if (nw < 4) nw = 4;
  if (nw > 16) nw = 16;
  fw->waves = nw;
  *lds_bytes = ((size_t)(npos + 1) * (kCh * 2) + (size_t)nw * 16 * kFwdHS * 32 + 16 * 4 * kMaxLevels +
                sizeof(int) * 6 * kMaxLevels + 15) & ~(size_t)15;
  return *lds_bytes <= kMaxLds;
}

int launch_fwd_win(const __bf16* value, const float* loc, const float* attn, __bf16* out, const ValueGeom& g,
                   const FwdWinGeom& fw, size_t lds, int B, int Nq, int points, hipStream_t st) {
  auto kern = msda_fwd_win_kernel<__bf16>;
  static bool done[64] = {};
  if (!reserve_lds((const void*)kern, (int)kMaxLds, done)) return fail(DSKD_ERR_LAUNCH, "dskd_msda_fwd: cannot reserve LDS");
  const dim3 grid((unsigned)(B * g.RY * g.RX * kHeads)), block(fw.waves * 64);
  hipLaunchKernelGGL(kern, grid, block, lds, st, value, loc, attn, out, g, fw, Nq, points);
  return DSKD_OK;
}

// ------------------------------------------------------------------ backward, grad_loc / grad_attn, windowed (encoder, bf16)
// The gather half of the backward reads the same 64 corner segments per (query, head) as the forward and is paced by
// the same texture path (msda_bwd_kernel: 263 us at B=4).  This is the forward's MIXED windowed scheme applied to it:
// a workgroup owns (image, region of <= 16x16 level-0 pixels, one head), holds that head's value windows of the
// coarse levels (>= lv0) in LDS and leaves the fine levels on the buffer-load path, so both pipes work side by side.
//   lane = (query of the pass, 16-B part of the head's 64-B line); per sample the four corner dot products with the
//   query's grad_out slice are formed per lane over its 8 channels and reduced over the 4 part-lanes with DPP --
//   the same channels per lane and the same reduction as msda_bwd_kernel<bf16>, hence bit-identical gradients;
//   the 16 (gx, gy, ga) triples of a (query, head) are parked in LDS and leave as whole 128-B / 64-B runs.
template <typename T>
__global__ __launch_bounds__(1024) void msda_bwd_win_kernel(
    const T* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn,
    const T* __restrict__ grad_out, float* __restrict__ grad_loc, float* __restrict__ grad_attn, float* __restrict__ stats,
    ValueGeom g, FwdWinGeom fw, int Nq, int points) {
  static_assert(sizeof(T) == 2, "windowed gather: bf16 only");
  constexpr int PIXB = kCh * (int)sizeof(T);     // 64 B: one head of one pixel
  constexpr int ROWB = kHeads * PIXB;            // 512 B: one pixel, all heads
  constexpr int LP = 16;
  extern __shared__ float smem[];
  char* win = reinterpret_cast<char*>(smem);                                    // [npos + 1][PIXB]
  const int NW = blockDim.x >> 6;
  i32x4* s_off_all = reinterpret_cast<i32x4*>(win + (size_t)(fw.npos + 1) * PIXB);   // [NW][16 * kFwdHS]
  f32x4* s_aux_all = reinterpret_cast<f32x4*>(s_off_all + NW * 16 * kFwdHS);         // {lx, ly, attn, 0}
  f32x2* s_gl_all = reinterpret_cast<f32x2*>(s_aux_all + NW * 16 * kFwdHS);          // [NW][16 queries][16 samples]
  float* s_ga_all = reinterpret_cast<float*>(s_gl_all + NW * 16 * LP);               // [NW][16][16]
  i32x4* s_tab = reinterpret_cast<i32x4*>(s_ga_all + NW * 16 * LP);                  // [4][4] lookup rows
  int* s_geo = reinterpret_cast<int*>(s_tab + 4 * kMaxLevels);                       // [24] region geometry
  float* s_st = reinterpret_cast<float*>(s_geo + 6 * kMaxLevels);                    // [NW][4] statistics partials

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int h = vb & 7; vb >>= 3;
  const int rx = vb % g.RX; vb /= g.RX;
  const int ry = vb % g.RY;
  const int b = vb / g.RY;

  // Region geometry: as in msda_fwd_win_kernel (same tables, same integer arithmetic).
  if (wave == 0) {
    if (lane < 6 * kMaxLevels) {
      const int l = lane / 6, kind = lane - 6 * l;
      const bool xaxis = kind == 0 || kind == 1 || kind == 4;
      int Sl = 1;
#pragma unroll
      for (int k = 0; k < kMaxLevels; ++k)
        if (l == k) Sl = xaxis ? g.W[k] : g.H[k];
      const int S0 = xaxis ? g.W[0] : g.H[0], E = xaxis ? g.EX : g.EY;
      const int r = (xaxis ? rx : ry) + ((kind == 1 || kind == 3) ? 1 : 0);
      const int q = floor_div(2 * E * r * Sl - S0 + (kind < 4 ? 2 * S0 - 1 : 0), 2 * S0);
      s_geo[lane] = kind < 4 ? (q < 0 ? 0 : (q > Sl ? Sl : q)) : q - kMarginLo;
    }
    wave_lds_sync();
    int tot = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < kMaxLevels; ++k) {
      if (k == lane) mine = tot;
      if (k < g.levels) tot += (s_geo[6 * k + 1] - s_geo[6 * k]) * (s_geo[6 * k + 3] - s_geo[6 * k + 2]);
    }
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l)
      if (lane == l) {
        const int qxa = s_geo[6 * l], qya = s_geo[6 * l + 2];
        const int qdx = l < g.levels ? s_geo[6 * l + 1] - qxa : 0;
        s_tab[4 * l + 0] = i32x4{mine, qxa, qya, qdx};
        s_tab[4 * l + 1] = i32x4{as_i32(1.0f / (float)(qdx > 0 ? qdx : 1)), g.start[l], g.W[l], g.H[l]};
        s_tab[4 * l + 2] = i32x4{g.ww[l], g.wh[l], s_geo[6 * l + 4], s_geo[6 * l + 5]};
        s_tab[4 * l + 3] = i32x4{fw.base[l], tot, 0, 0};
      }
  }
  __syncthreads();
  int cum[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) cum[l] = s_tab[4 * l].x;
  const int nq = s_tab[3].y;

  const T* vbase = value + (size_t)b * Nq * (kHeads * kCh);      // Nq == Nv
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(vbase), 0, Nq * ROWB, 0x00020000);
  const int part = lane & 3;
  const int hb = h * PIXB + part * 16;

  // ---- window fill (as in the forward)
  {
    constexpr int kFill = 5;
    const int step = blockDim.x >> 2;
    for (int p0 = tid >> 2; p0 <= fw.npos; p0 += kFill * step) {
      u32x4 v[kFill];
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const int p = p0 + u * step;
        int l = fw.lv0;
#pragma unroll
        for (int k = 1; k < kMaxLevels; ++k) l += (k > fw.lv0 && p >= fw.base[k]) ? 1 : 0;
        const i32x4 lb = s_tab[4 * l + 1], lc = s_tab[4 * l + 2];
        const int rel = p - s_tab[4 * l + 3].x;
        const int wwl = lc.x;
        const int wy = (int)(((float)rel + 0.5f) / (float)wwl), wx = rel - wy * wwl;
        const int gx = lc.z + wx, gy = lc.w + wy;
        const bool in = p < fw.npos && gx >= 0 && gx < lb.z && gy >= 0 && gy < lb.w;
        const int goff = in ? (lb.y + gy * lb.z + gx) * ROWB + hb : kOOB;
        v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const int p = p0 + u * step;
        if (p <= fw.npos) *reinterpret_cast<u32x4*>(win + (size_t)p * PIXB + part * 16) = v[u];
      }
    }
  }
  __syncthreads();

  i32x4* s_off = s_off_all + wave * 16 * kFwdHS;
  f32x4* s_aux = s_aux_all + wave * 16 * kFwdHS;
  f32x2* s_gl = s_gl_all + wave * 16 * LP;
  float* s_ga = s_ga_all + wave * 16 * LP;
  const int ql = lane >> 2;                  // query of the pass
  const int zero_slot = fw.npos * PIXB;

  f32x2 n_xy[kMaxLevels] = {};
  float n_a[kMaxLevels];
  u32x4 n_go = u32x4{0u, 0u, 0u, 0u};
  int n_qg = -1;
  auto fetch = [&](int qb) {
    const int qi = qb + ql;
    n_qg = qi < nq ? region_query(s_tab, cum, qi) : -1;
    if (n_qg >= 0) {
      const size_t base = (((size_t)b * Nq + n_qg) * kHeads + h) * (size_t)LP + part;
#pragma unroll
      for (int l = 0; l < kMaxLevels; ++l) {
        if (l < fw.lv_end) n_xy[l] = *reinterpret_cast<const f32x2*>(loc + (base + l * points) * 2);
        n_a[l] = attn[base + l * points];      // every level's weight: the statistics need them (same 64-byte line)
      }
      n_go = *reinterpret_cast<const u32x4*>(grad_out + ((size_t)b * Nq + n_qg) * (kHeads * kCh) + h * kCh + part * 8);
    }
  };
  fetch(wave * 16);

  // By-product for the windowed grad_value kernels (their fixed-point scale): max |grad_out| and the sums of |attn| of
  // levels 1 and 2+3 over this workgroup's (region, head) -- the data is in registers here anyway, and those kernels
  // no longer walk their region's grad_out / attn rows a second time (that pre-pass was 8-17 % of their run time).
  unsigned st_gbits = 0u;      // bf16 bit pattern of max |grad_out| (NaN patterns compare above every number: kept)
  float st_a1 = 0.f, st_a23 = 0.f;
  for (int qbase = wave * 16; qbase < nq; qbase += NW * 16) {
    const int qg = n_qg;
    f32x2 c_xy[kMaxLevels];
    float c_a[kMaxLevels];
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l) { c_xy[l] = n_xy[l]; c_a[l] = n_a[l]; }
    const u32x4 c_go = n_go;          // this pass's grad_out slice, packed (the dot products read it as it is)
    if (qg >= 0) {
      // |bf16| compares like its bit pattern: the max over the 8 packed channels without unpacking them
      const unsigned ax = n_go.x & 0x7FFF7FFFu, ay = n_go.y & 0x7FFF7FFFu, az = n_go.z & 0x7FFF7FFFu, aw = n_go.w & 0x7FFF7FFFu;
      const unsigned hi = max(max(ax >> 16, ay >> 16), max(az >> 16, aw >> 16));
      const unsigned lo = max(max(ax & 0xFFFFu, ay & 0xFFFFu), max(az & 0xFFFFu, aw & 0xFFFFu));
      st_gbits = max(st_gbits, max(hi, lo));
      st_a1 += fabsf(c_a[1]);
      st_a23 += fabsf(c_a[2]) + fabsf(c_a[3]);
    }
    fetch(qbase + NW * 16);

#pragma unroll
    for (int lvl = 0; lvl < kMaxLevels; ++lvl) {
      if (lvl >= fw.lv_end) break;          // wave-uniform: the coarser levels belong to the grad_value kernel
      // ---- stage the 4 points of this level: lane = (query, point)
      {
        i32x4 off = i32x4{zero_slot, zero_slot, zero_slot, zero_slot};
        f32x4 aux = f32x4{0.f, 0.f, 0.f, 0.f};
        if (qg >= 0) {
          const f32x2 xy = c_xy[lvl];
          const i32x4 lb = s_tab[4 * lvl + 1], lc = s_tab[4 * lvl + 2];
          const int H = lb.w, W = lb.z, st = lb.y;
          const float x = xy.x * (float)W - 0.5f;
          const float y = xy.y * (float)H - 0.5f;
          aux.z = c_a[lvl];
          if (x > -1.f && y > -1.f && x < (float)W && y < (float)H) {   // as point_params
            const float xf = floorf(x), yf = floorf(y);
            const int x0 = (int)xf, y0 = (int)yf;
            aux.x = x - xf;
            aux.y = y - yf;
            const int wwl = lc.x, whl = lc.y;
            const int wx = x0 - lc.z, wy = y0 - lc.w;
            // (selects, not a lane branch: with the two forms in separate blocks the allocator spilled eight registers of the
            // staging arithmetic -- 45 MB of scratch written back per launch, profiles/r04_msda_pmc_hbm_B4_bf16.json)
            const bool inw = lvl >= fw.lv0 && wx >= 0 && wx + 1 < wwl && wy >= 0 && wy + 1 < whl;
            const int pb = (s_tab[4 * lvl + 3].x + wy * wwl + wx) * PIXB;
            const bool vx0 = x0 >= 0, vx1 = x0 + 1 <= W - 1;
            const bool vy0 = y0 >= 0, vy1 = y0 + 1 <= H - 1;
            const int r00 = (st + y0 * W + x0) * ROWB;       // left the window: global byte offsets, flagged by the sign bit
            off.x = inw ? pb : (((vy0 && vx0) ? r00 : kOOB) | kFwdSign);
            off.y = inw ? pb + PIXB : (((vy0 && vx1) ? r00 + ROWB : kOOB) | kFwdSign);
            off.z = inw ? pb + wwl * PIXB : (((vy1 && vx0) ? r00 + W * ROWB : kOOB) | kFwdSign);
            off.w = inw ? pb + wwl * PIXB + PIXB : (((vy1 && vx1) ? r00 + W * ROWB + ROWB : kOOB) | kFwdSign);
          }
        }
        s_off[ql * kFwdHS + part] = off;
        s_aux[ql * kFwdHS + part] = aux;
      }
      wave_lds_sync();
      const i32x4 lt = s_tab[4 * lvl + 1];
      const float Wf = (float)lt.z, Hf = (float)lt.w;
      // ---- consume: lane = (query, 16-B part)
      // r4: the corner loads of sample sl + 1 are requested BEFORE sample sl is consumed (the ISA of the plain loop had each
      // sample issue its four loads and wait for them on the spot -- the lane branch between the LDS and the buffer path kept
      // the compiler from hoisting them: one LDS / L2 latency per sample and wave).
      auto fetch4 = [&](int sl, u32x4& a0, u32x4& a1, u32x4& a2, u32x4& a3) {
        const i32x4 o = s_off[ql * kFwdHS + sl];
        if (o.x < 0) {
          a0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (o.x & 0x7FFFFFFF) + hb, 0, 0));
          a1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (o.y & 0x7FFFFFFF) + hb, 0, 0));
          a2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (o.z & 0x7FFFFFFF) + hb, 0, 0));
          a3 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (o.w & 0x7FFFFFFF) + hb, 0, 0));
        } else {
          a0 = *reinterpret_cast<const u32x4*>(win + o.x + part * 16);
          a1 = *reinterpret_cast<const u32x4*>(win + o.y + part * 16);
          a2 = *reinterpret_cast<const u32x4*>(win + o.z + part * 16);
          a3 = *reinterpret_cast<const u32x4*>(win + o.w + part * 16);
        }
      };
      u32x4 n0, n1, n2, n3;
      fetch4(0, n0, n1, n2, n3);
      float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const u32x4 r0 = n0, r1 = n1, r2 = n2, r3 = n3;
        if (sl < 3) fetch4(sl + 1, n0, n1, n2, n3);
        float d0 = dot8_bf16(r0, c_go), d1 = dot8_bf16(r1, c_go), d2 = dot8_bf16(r2, c_go), d3 = dot8_bf16(r3, c_go);
        dot8_settle(d0, d1, d2, d3);
        d0 = group4_sum(d0); d1 = group4_sum(d1); d2 = group4_sum(d2); d3 = group4_sum(d3);
        // lane `part` of the group keeps the sums of sample `part` and finishes it ONCE behind the loop (r4: the ~25
        // instructions of the finishing arithmetic ran on all four lanes for every sample, three of them for nothing --
        // the kernel is bound by its vector instruction count, not by the gathers)
        if (part == sl) { k0 = d0; k1 = d1; k2 = d2; k3 = d3; }
      }
      {
        const f32x4 ax = s_aux[ql * kFwdHS + part];
        const float lx = ax.x, ly = ax.y, a = ax.z;
        const float hx = 1.f - lx, hy = 1.f - ly;
        const float ga = (hy * hx) * k0 + (hy * lx) * k1 + (ly * hx) * k2 + (ly * lx) * k3;
        const float gx = Wf * a * (hy * (k1 - k0) + ly * (k3 - k2));
        const float gy = Hf * a * (hx * (k2 - k0) + lx * (k3 - k1));
        s_gl[ql * LP + lvl * 4 + part] = f32x2{gx, gy};
        s_ga[ql * LP + lvl * 4 + part] = ga;
      }
      wave_lds_sync();   // the next level's staging overwrites the slots
    }
    // ---- write the pass out: 16 queries x 16 samples, whole (query, head) runs
    {
      const int nvalid = min(16, nq - qbase);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane;          // (query of the pass, sample)
        const int q2 = idx >> 4, smp = idx & 15;
        const int qg2 = __shfl(qg, q2 * 4);      // the query's global index lives in its first lane (all lanes take part:
        if (q2 < nvalid && smp < fw.lv_end * 4) {   // a shuffle reads nothing from a lane that sits out a branch)
          const size_t base = (((size_t)b * Nq + qg2) * kHeads + h) * (size_t)LP + smp;
          grad_attn[base] = s_ga[q2 * LP + smp];
          *reinterpret_cast<f32x2*>(grad_loc + base * 2) = s_gl[q2 * LP + smp];
        }
      }
    }
    wave_lds_sync();
  }
  if (stats) {
    float st_gmax = as_f32(st_gbits << 16);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      st_gmax = fmaxf(st_gmax, __shfl_xor(st_gmax, o));
      st_a1 += __shfl_xor(st_a1, o);
      st_a23 += __shfl_xor(st_a23, o);
    }
    // every lane of a (query) quad saw the same grad_out slice only in part (8 of 32 channels): the max over the quad is
    // in the wave max; attn: each lane one point per level -> the sums count every sample once
    if (lane == 0) { s_st[wave * 4 + 0] = st_gmax; s_st[wave * 4 + 1] = st_a1; s_st[wave * 4 + 2] = st_a23; }
    __syncthreads();
    if (tid == 0) {
      float gm = 0.f, a1 = 0.f, a23 = 0.f;
      for (int w2 = 0; w2 < NW; ++w2) { gm = fmaxf(gm, s_st[w2 * 4]); a1 += s_st[w2 * 4 + 1]; a23 += s_st[w2 * 4 + 2]; }
      float* dst = stats + ((((size_t)b * g.RY + ry) * g.RX + rx) * kHeads + h) * 4;
      *reinterpret_cast<f32x4*>(dst) = f32x4{gm, a1, a23, 0.f};
    }
  }
}

int launch_bwd_win(const __bf16* value, const float* loc, const float* attn, const __bf16* grad_out, float* grad_loc,
                   float* grad_attn, float* stats, const ValueGeom& g, const FwdWinGeom& fw, size_t lds, int B, int Nq,
                   int points, hipStream_t st) {
  auto kern = msda_bwd_win_kernel<__bf16>;
  static bool done[64] = {};
  if (!reserve_lds((const void*)kern, (int)kMaxLds, done)) return fail(DSKD_ERR_LAUNCH, "dskd_msda_bwd: cannot reserve LDS");
  const dim3 grid((unsigned)(B * g.RY * g.RX * kHeads)), block(fw.waves * 64);
  hipLaunchKernelGGL(kern, grid, block, lds, st, value, loc, attn, grad_out, grad_loc, grad_attn, stats, g, fw, Nq, points);
  return DSKD_OK;
}

int fill_geom(const int64_t* spatial_shapes, const int64_t* level_start, int levels, int Nv,
              LevelGeom* g) {
  int64_t covered = 0;
  for (int l = 0; l < kMaxLevels; ++l) {
    if (l < levels) {
      const int64_t H = spatial_shapes[2 * l], W = spatial_shapes[2 * l + 1];
      const int64_t st = level_start[l];
      if (H <= 0 || W <= 0 || st < 0 || st + H * W > Nv)
        return fail(DSKD_ERR_INVALID_ARG, "msda: level %d (H=%lld W=%lld start=%lld) exceeds Nv=%d",
                    l, (long long)H, (long long)W, (long long)st, Nv);
      g->H[l] = (int)H; g->W[l] = (int)W; g->start[l] = (int)st;
      covered += H * W;
    } else {
      g->H[l] = 1; g->W[l] = 1; g->start[l] = 0;
    }
  }
  (void)covered;
  return DSKD_OK;
}

int check_shapes(const char* who, int B, int Nv, int Nq, int heads, int ch, int levels,
                 int points, int dtype) {
  if (heads != kHeads || ch != kCh)
    return fail(DSKD_ERR_INVALID_ARG, "%s: only heads=8, ch=32 supported (got %d, %d)", who, heads, ch);
  if (levels < 1 || levels > kMaxLevels || points < 1 || levels * points > kMaxLP)
    return fail(DSKD_ERR_INVALID_ARG, "%s: need levels<=4 and levels*points<=16 (got %d, %d)", who, levels, points);
  if (B < 0 || Nv <= 0 || Nq < 0)
    return fail(DSKD_ERR_INVALID_ARG, "%s: bad sizes B=%d Nv=%d Nq=%d", who, B, Nv, Nq);
  if ((int64_t)Nv * 1024 >= (int64_t)kOOB)
    return fail(DSKD_ERR_INVALID_ARG, "%s: Nv=%d too large for 32-bit row offsets", who, Nv);
  if (dtype != DSKD_DTYPE_F32 && dtype != DSKD_DTYPE_BF16)
    return fail(DSKD_ERR_INVALID_ARG, "%s: unknown dtype %d", who, dtype);
  return DSKD_OK;
}

constexpr int kQPB = 32;  // queries per workgroup (8 per wave) when the grid is large

// Few queries (decoder cross-attention: 300 per image) would leave most CUs idle at 32 queries
// per workgroup; shrink the workgroup's share down to one wave pass so the grid covers the chip.
inline int pick_qpb(int B, int Nq, int dtype) {
  const int min_qpb = kWaves * (dtype == DSKD_DTYPE_BF16 ? 2 : 1);
  int qpb = kQPB;
  while (qpb > min_qpb && (long long)B * ((Nq + qpb - 1) / qpb) < 1024) qpb >>= 1;
  return qpb < min_qpb ? min_qpb : qpb;
}

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int dskd_msda_fwd(const void* value, const int64_t* spatial_shapes,
                             const int64_t* level_start, const float* loc,
                             const float* attn, void* out, int B, int Nv, int Nq,
                             int heads, int ch, int levels, int points, int dtype,
                             void* stream) {
  if (int rc = check_shapes("dskd_msda_fwd", B, Nv, Nq, heads, ch, levels, points, dtype)) return rc;
  if (!value || !loc || !attn || !out || !spatial_shapes || !level_start)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_fwd: null pointer");
  if (B == 0 || Nq == 0) return DSKD_OK;
  LevelGeom g;
  if (int rc = fill_geom(spatial_shapes, level_start, levels, Nv, &g)) return rc;
  const int qpb = pick_qpb(B, Nq, dtype);
  const int bpi = (Nq + qpb - 1) / qpb;
  const dim3 grid((unsigned)(B * bpi)), block(kWaves * 64);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DSKD_DTYPE_F32) {
    hipLaunchKernelGGL((msda_fwd_kernel<float, false, 1>), grid, block, 0, st, (const float*)value, loc, attn,
                       (const float*)nullptr, (const float*)nullptr, (float*)out, g, Nv, Nq, levels * points,
                       points, qpb, bpi);
  } else {
#define DSKD_FWD_BF16(PH)                                                                                       \
  hipLaunchKernelGGL((msda_fwd_kernel<__bf16, false, PH>), grid, block, 0, st, (const __bf16*)value, loc, attn, \
                     (const __bf16*)nullptr, (const float*)nullptr, (__bf16*)out, g, Nv, Nq, levels * points,   \
                     points, qpb, bpi)
    // Encoder shape (queries == pixels, 4 levels x 4 points): windowed forward in MIXED mode -- the coarse levels
    // 2+3 of one head in LDS, the fine levels on the buffer-load path, 8 waves per workgroup (two workgroups per
    // CU): bit-identical to the plain kernel and 13-15 % faster (other splits / wave counts measured
    // slower in round 1-2 and removed).
    if (Nq == Nv) {
      ValueGeom vg;
      FwdWinGeom fw;
      size_t lds = 0;
      if (make_fwd_win_geom(g, levels, points, Nq, 2, 8, &vg, &fw, &lds)) {
        if (int rc = launch_fwd_win((const __bf16*)value, loc, attn, (__bf16*)out, vg, fw, lds, B, Nq, points, st))
          return rc;
        return check_launch("dskd_msda_fwd");
      }
    }
    DSKD_FWD_BF16(1);
#undef DSKD_FWD_BF16
  }
  return check_launch("dskd_msda_fwd");
}

extern "C" int dskd_msda_fwd_fused(const void* value, const int64_t* spatial_shapes,
                                   const int64_t* level_start, const void* both, const float* ref,
                                   void* out, int B, int Nv, int Nq, int heads, int ch, int levels,
                                   int points, int dtype, void* stream) {
  if (int rc = check_shapes("dskd_msda_fwd_fused", B, Nv, Nq, heads, ch, levels, points, dtype)) return rc;
  if (levels * points != 16)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_fwd_fused: needs levels*points == 16 (got %d x %d)", levels, points);
  if (!value || !both || !ref || !out || !spatial_shapes || !level_start)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_fwd_fused: null pointer");
  if (B == 0 || Nq == 0) return DSKD_OK;
  LevelGeom g;
  if (int rc = fill_geom(spatial_shapes, level_start, levels, Nv, &g)) return rc;
  const int qpb = pick_qpb(B, Nq, dtype);
  const int bpi = (Nq + qpb - 1) / qpb;
  const dim3 grid((unsigned)(B * bpi)), block(kWaves * 64);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DSKD_DTYPE_F32)
    hipLaunchKernelGGL((msda_fwd_kernel<float, true, 1>), grid, block, 0, st, (const float*)value,
                       (const float*)nullptr, (const float*)nullptr, (const float*)both, ref, (float*)out, g, Nv,
                       Nq, 16, points, qpb, bpi);
  else
    hipLaunchKernelGGL((msda_fwd_kernel<__bf16, true, 1>), grid, block, 0, st, (const __bf16*)value,
                       (const float*)nullptr, (const float*)nullptr, (const __bf16*)both, ref, (__bf16*)out, g,
                       Nv, Nq, 16, points, qpb, bpi);
  return check_launch("dskd_msda_fwd_fused");
}

namespace dskd {
namespace {
// workspace != nullptr: the caller's grad_value is NOT assumed zeroed (this function zeroes what its atomics need)
// and the fine levels may go through the pull kernel.
int msda_bwd_impl(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const float* loc,
                  const float* attn, const void* grad_out, float* grad_value, float* grad_loc, float* grad_attn,
                  int B, int Nv, int Nq, int heads, int ch, int levels, int points, int dtype, void* workspace,
                  size_t workspace_bytes, void* stream) {
  if (int rc = check_shapes("dskd_msda_bwd", B, Nv, Nq, heads, ch, levels, points, dtype)) return rc;
  if (!value || !loc || !attn || !grad_out || !grad_value || !grad_loc || !grad_attn ||
      !spatial_shapes || !level_start)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_bwd: null pointer");
  if (B == 0 || Nq == 0) return DSKD_OK;
  LevelGeom g;
  if (int rc = fill_geom(spatial_shapes, level_start, levels, Nv, &g)) return rc;
  const int qpb = pick_qpb(B, Nq, dtype);
  const int bpi = (Nq + qpb - 1) / qpb;
  const dim3 grid((unsigned)(B * bpi)), block(kWaves * 64);
  hipStream_t st = (hipStream_t)stream;
  const int LP = levels * points;

  // Encoder self-attention shape (queries == pixels): windowed LDS accumulation of grad_value.
  ValueGeom vg;
  VarGeom var[kNumVar];
  size_t lds[kNumVar];
  bool windowed = Nq == Nv && make_value_geom(g, levels, points, Nq, &vg, var, lds);
  int pull_mask = 0;
  int mm = 0;               // bit 1: level 1, bit 2: levels 2+3: grad_value on the matrix-core kernel (msda_mm.hip)
  MsdaLevels ml;
  // The bf16 workspace path: grad_value of the coarse levels on the matrix-core kernel, every level's grad_loc / grad_attn
  // in the gather kernel.  Where the matrix-core kernel does not take levels 2+3, the windowed levels-2+3 launch forms
  // its own samples' gradients (bound from the gather kernel's statistics) and the gather keeps levels 0+1.
  ValueGeom wg;
  FwdWinGeom fw;
  size_t wl = 0;
  bool gather_win = false, fuse23 = false;
  size_t stats_bytes = 0;
  if (workspace && windowed) {
    for (int l = 0; l < kMaxLevels; ++l) { ml.H[l] = g.H[l]; ml.W[l] = g.W[l]; ml.start[l] = g.start[l]; }
    pull_mask = pull_level_mask();
    // every level's tile geometry is validated here, before the first launch of this call
    if (pull_mask && !pull_supported(ml, levels, points, Nv, Nq, dtype, pull_mask)) pull_mask = 0;
    if (dtype == DSKD_DTYPE_BF16) {
      mm = mm_level_mask();
      if (pull_mask & 2) mm &= ~2;
      if (pull_mask >> 2) mm &= ~4;
      if ((mm & 2) && !mm_supported(ml, levels, points, Nv, Nq, dtype, 1, 1)) mm &= ~2;
      if ((mm & 4) && !mm_supported(ml, levels, points, Nv, Nq, dtype, 2, 2)) mm &= ~4;
    }
  }
  if (windowed && dtype == DSKD_DTYPE_BF16) {
    const int gnw = 8;      // waves per gather workgroup (4: 175 us, 8: 152-159 us, 12 / 16: 200-206 us)
    if (mm) {
      // the gather forms every level's grad_loc / grad_attn (levels 2+3 of one head in LDS, the fine levels on the
      // buffer-load path) and the statistics
      gather_win = make_fwd_win_geom(g, levels, points, Nq, 2, gnw, &wg, &fw, &wl);
      if (gather_win) {
        wl += (size_t)fw.waves * 16 * 16 * 12 + (size_t)fw.waves * 16;
        gather_win = wl <= kMaxLds;
      }
      stats_bytes = (size_t)B * wg.RY * wg.RX * kHeads * 16;      // the f16 scale of the matrix-core kernel comes from them
      if (!gather_win || workspace_bytes < kPullWsHeader + stats_bytes + kPullWsEntry) { mm = 0; stats_bytes = 0; }
    }
  }
  if (windowed && dtype == DSKD_DTYPE_BF16 && !mm) {
    fuse23 = workspace && (pull_mask >> 2) == 0;     // levels 2+3 on the windowed kernel (the default split)
    const int gnw = 8;
    // fused path: the gather keeps levels 0+1 -- level 1's windows (23 KB) in LDS, level 0 on the buffer-load path
    // (measured: 152 us against 168 us with no window and 203 us with both levels' windows)
    gather_win = make_fwd_win_geom(g, levels, points, Nq, fuse23 ? 1 : 2, gnw, &wg, &fw, &wl, fuse23 ? 2 : kMaxLevels);
    if (gather_win) {
      wl += (size_t)fw.waves * 16 * 16 * 12 + (size_t)fw.waves * 16;   // parked (gx, gy, ga) triples; statistics partials
      gather_win = wl <= kMaxLds;
    }
    if (fuse23) {
      stats_bytes = (size_t)B * wg.RY * wg.RX * kHeads * 16;
      VarGeom var2[kNumVar];
      size_t lds2[kNumVar];
      fuse23 = gather_win && workspace_bytes >= kPullWsHeader + stats_bytes + kPullWsEntry &&
               make_value_geom(g, levels, points, Nq, &vg, var2, lds2, true);
      if (fuse23) {
        for (int v = 0; v < kNumVar; ++v) { var[v] = var2[v]; lds[v] = lds2[v]; }
      } else {                                    // back to the unfused launches (the gather then takes every level)
        stats_bytes = 0;
        windowed = make_value_geom(g, levels, points, Nq, &vg, var, lds);
        gather_win = windowed && make_fwd_win_geom(g, levels, points, Nq, 2, 8, &wg, &fw, &wl);
        if (gather_win) {
          wl += (size_t)fw.waves * 16 * 16 * 12 + (size_t)fw.waves * 16;
          gather_win = wl <= kMaxLds;
        }
      }
    }
  }
  // statistics at the END of the workspace; the stray list of the pull kernel keeps the front
  float* stats = (fuse23 || mm) ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ((workspace_bytes - stats_bytes) & ~(size_t)15))
                        : nullptr;
  if (fuse23 || mm) stats_bytes = workspace_bytes - (size_t)(reinterpret_cast<char*>(stats) - reinterpret_cast<char*>(workspace));
  if (workspace) {
    // zero what the atomics of the remaining kernels add into, and the stray-list header
    unsigned* hdr = reinterpret_cast<unsigned*>(workspace);
    if (pull_mask == 0) {
      zero_fill(workspace, kPullWsHeader, st);
      zero_fill(grad_value, sizeof(float) * (size_t)B * Nv * (kHeads * kCh), st);
    } else {
      // runs of consecutive un-pulled levels (their rows are contiguous) share one launch
      for (int l = 0; l < levels;) {
        if (pull_mask & (1 << l)) { ++l; continue; }
        int e = l, nrows = 0;
        while (e < levels && !(pull_mask & (1 << e)) && g.start[e] == g.start[l] + nrows) nrows += g.H[e] * g.W[e], ++e;
        const int bx = (nrows * 64 + 255) / 256;
        hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)(bx < 1024 ? bx : 1024), (unsigned)B), dim3(256), 0, st,
                           grad_value, Nv, g.start[l], nrows, hdr);
        hdr = nullptr;
        l = e;
      }
      if (hdr) zero_fill(workspace, kPullWsHeader, st);       // every level pulled: nothing else zeroes it
    }
  }
  const int variants = (pull_mask & 1 ? 0 : 1) | ((pull_mask & 2) || (mm & 2) ? 0 : 2) | ((pull_mask >> 2) == 3 || (mm & 4) ? 0 : 4);
  if (windowed) {
    ValueExtra ex;
    ex.value = value; ex.grad_loc = grad_loc; ex.grad_attn = grad_attn; ex.stats = stats;
    ex.sRX = wg.RX; ex.sRY = wg.RY; ex.sEX = wg.EX; ex.sEY = wg.EY; ex.fuse23 = fuse23;
    if (dtype == DSKD_DTYPE_F32) {
      ex.stats = nullptr;
      hipLaunchKernelGGL((msda_bwd_kernel<float, false, 1>), grid, block, 0, st, (const float*)value, loc, attn,
                         (const float*)grad_out, grad_value, grad_loc, grad_attn, g, Nv, Nq, LP, points, qpb,
                         bpi);
      if (int rc = launch_value<float>(loc, attn, (const float*)grad_out, grad_value, vg, var, lds, B, Nq, LP,
                                       points, ex, st, variants)) return rc;
    } else {
      // grad_loc / grad_attn: the windowed gather kernel (fused path: levels 0+1 on the buffer-load path, no windows;
      // otherwise levels 2+3 of one head in LDS), else the plain gather kernel
      if (gather_win) {
        if (int rc = launch_bwd_win((const __bf16*)value, loc, attn, (const __bf16*)grad_out, grad_loc, grad_attn, stats,
                                    wg, fw, wl, B, Nq, points, st)) return rc;
      } else {
        hipLaunchKernelGGL((msda_bwd_kernel<__bf16, false, 1>), grid, block, 0, st, (const __bf16*)value, loc, attn,
                           (const __bf16*)grad_out, grad_value, grad_loc, grad_attn, g, Nv, Nq, LP, points, qpb, bpi);
      }
      if (int rc = launch_value<__bf16>(loc, attn, (const __bf16*)grad_out, grad_value, vg, var, lds, B, Nq, LP,
                                        points, ex, st, variants)) return rc;
    }
    const int sgrid[4] = {wg.RX, wg.RY, wg.EX, wg.EY};
    if (mm)
      if (int rc = launch_bwd_mm(loc, attn, grad_out, grad_value, stats, sgrid, ml, mm, B, Nq, points, st)) return rc;
    if (pull_mask)
      if (int rc = launch_pull(loc, attn, grad_out, grad_value, ml, pull_mask, B, Nq, dtype, workspace,
                               workspace_bytes - stats_bytes, st)) return rc;
    return check_launch("dskd_msda_bwd");
  }
  if (dtype == DSKD_DTYPE_F32)
    hipLaunchKernelGGL((msda_bwd_kernel<float, true, 1>), grid, block, 0, st, (const float*)value, loc,
                       attn, (const float*)grad_out, grad_value, grad_loc, grad_attn, g, Nv,
                       Nq, LP, points, qpb, bpi);
  else
    hipLaunchKernelGGL((msda_bwd_kernel<__bf16, true, 1>), grid, block, 0, st, (const __bf16*)value, loc,
                       attn, (const __bf16*)grad_out, grad_value, grad_loc, grad_attn, g, Nv,
                       Nq, LP, points, qpb, bpi);
  return check_launch("dskd_msda_bwd");
}
}  // namespace
}  // namespace dskd

extern "C" int dskd_msda_bwd(const void* value, const int64_t* spatial_shapes,
                             const int64_t* level_start, const float* loc,
                             const float* attn, const void* grad_out, float* grad_value,
                             float* grad_loc, float* grad_attn, int B, int Nv, int Nq,
                             int heads, int ch, int levels, int points, int dtype,
                             void* stream) {
  return msda_bwd_impl(value, spatial_shapes, level_start, loc, attn, grad_out, grad_value, grad_loc, grad_attn, B, Nv,
                       Nq, heads, ch, levels, points, dtype, nullptr, 0, stream);
}

extern "C" int64_t dskd_msda_bwd_workspace(int B, int Nv, int Nq, int heads, int levels, int points) {
  (void)Nv;
  if (B < 0 || Nq < 0 || heads < 0 || levels < 0 || points < 0) return -1;
  // header + room for 1/16 of all (query, head, point, corner) contributions as stray entries + the gather kernel's
  // statistics: 16 bytes per (16 x 16-pixel region, head); regions <= Nq / 256 + Nq / 16 + 2 for any level-0 shape
  int64_t entries = (int64_t)B * Nq * heads * levels * points * 4 / 16;
  if (entries < 4096) entries = 4096;
  const int64_t stats = (int64_t)B * ((int64_t)Nq / 256 + Nq / 16 + 2) * heads * 16 + 16;
  return (int64_t)kPullWsHeader + entries * (int64_t)kPullWsEntry + stats;
}

extern "C" int dskd_msda_bwd_ws(const void* value, const int64_t* spatial_shapes,
                                const int64_t* level_start, const float* loc,
                                const float* attn, const void* grad_out, float* grad_value,
                                float* grad_loc, float* grad_attn, int B, int Nv, int Nq,
                                int heads, int ch, int levels, int points, int dtype,
                                void* workspace, int64_t workspace_bytes, void* stream) {
  if (!workspace || workspace_bytes < (int64_t)(kPullWsHeader + kPullWsEntry) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_bwd_ws: need a 16-byte aligned workspace of at least %d bytes",
                (int)(kPullWsHeader + kPullWsEntry));
  return msda_bwd_impl(value, spatial_shapes, level_start, loc, attn, grad_out, grad_value, grad_loc, grad_attn, B, Nv,
                       Nq, heads, ch, levels, points, dtype, workspace, (size_t)workspace_bytes, stream);
}

#ifdef DSKD_VALUE_PROFILE
extern "C" int dskd_debug_value_prof(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dskd::g_vprof), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
  if (reset) { unsigned long long z[64] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(dskd::g_vprof), z, sizeof(z)); }
  return 0;
}
#endif
