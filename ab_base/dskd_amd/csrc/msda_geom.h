// Shared by msda.hip (gather / windowed kernels, dispatch) and msda_mm.hip (matrix-core backward): constants of the op,
// the region decomposition of the encoder shape and its lookup tables.  Internal; every definition has internal linkage.
#pragma once
#include "common.h"

namespace dskd {
namespace {

constexpr int kHeads = 8;
constexpr int kCh = 32;
constexpr int kMaxLevels = 4;
constexpr int kOOB = 0x7F000000;  // byte offset beyond every descriptor range

struct LevelGeom {
  int H[kMaxLevels];
  int W[kMaxLevels];
  int start[kMaxLevels];
};

// NOTE: __builtin_bit_cast applied directly to a vector ELEMENT lvalue (v.y, v[1]) reads
// element 0 with this compiler (hipcc 7.2); always go through a scalar by-value helper.
__device__ __forceinline__ float as_f32(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ int as_i32(float f) { return __builtin_bit_cast(int, f); }
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x2 as_bf16x2(unsigned u) { return __builtin_bit_cast(bf16x2, u); }
using i32x2 = __attribute__((ext_vector_type(2))) int;

constexpr int kRegion = 32;     // largest region edge, level-0 pixels (edges are balanced: ceil(S0 / ceil(S0/32)))
constexpr int kMarginLo = 5;    // window margin below / above the region footprint
constexpr int kMarginHi = 6;
constexpr int kMaxReg = 16;     // regions per axis
constexpr int kSkip = 0x7FFFFFF0;

struct ValueGeom {
  int H[kMaxLevels], W[kMaxLevels], start[kMaxLevels];
  int ww[kMaxLevels], wh[kMaxLevels];
  int RX, RY, EX, EY, levels;
};

// floor(x + 0.5) in ONE VALU instruction (v_cvt_rpi_i32_f32; checked on gfx950).  Plain
// truncation would bias every contribution towards zero, which shows on the coarse levels
// where a cell sums hundreds of them.
__device__ __forceinline__ int cvt_round(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

__host__ __device__ inline int floor_div(int a, int b) {  // b > 0
  const int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}

// Region r of an axis (edge E level-0 pixels, finest extent S0) on a level of extent Sl:
//  * the query x belongs to region floor(((2x+1) * S0) / (2 * Sl)) / E, so its query range starts
//    at ceil((2 * E * r * Sl - S0) / (2 * S0)), clamped to [0, Sl];
//  * its window starts at floor(r * E * Sl / S0 - 0.5) - kMarginLo.
// Both are evaluated in exact integer arithmetic in the kernel prologue.

// Global query index of the qi-th query of a region (queries ordered level by level, row by row
// inside the region's footprint on that level); rows of the lookup table as filled below.
__device__ __forceinline__ int region_query(const i32x4* s_tab, const int* cum, int qi) {
  int lq = 0;
#pragma unroll
  for (int l = 1; l < kMaxLevels; ++l) lq += qi >= cum[l];
  const i32x4 qa = s_tab[4 * lq], qb = s_tab[4 * lq + 1];
  const int rem = qi - qa.x;
  // exact: the fractional part of (rem + 0.5) / dx is at least 0.5/dx away from an integer
  const int yy = (int)(((float)rem + 0.5f) * as_f32((unsigned)qb.x));
  return qb.y + (qa.z + yy) * qb.z + qa.y + (rem - yy * qa.w);
}


// Region tables of the workgroup that owns region (rx, ry): executed by wave 0 (every lane), followed by the caller's
// __syncthreads().  s_geo: 6 * kMaxLevels ints of scratch.  BASE(l): first window position of level l in the caller's
// window image (its own meaning).  A macro, not a function: ``g`` is a kernel-argument struct, and passing it on by
// reference makes the compiler keep a private (scratch) copy of its arrays.
//   s_tab[4l+0] = {cum, qxa, qya, qdx}   [4l+1] = {1/qdx, start, W, H}
//   s_tab[4l+2] = {ww, wh, wx0, wy0}     [4l+3] = {window base, nq (total queries of the region), 0, 0}
#define DSKD_REGION_TABLES(g, rx, ry, lane, BASE, s_tab, s_geo)                                                        \
  do {                                                                                                                 \
    if ((lane) < 6 * kMaxLevels) {                                                                                     \
      const int l_ = (lane) / 6, kind_ = (lane) - 6 * l_; /* 0,1: x begin/end  2,3: y begin/end  4,5: origin x/y */    \
      const bool xaxis_ = kind_ == 0 || kind_ == 1 || kind_ == 4;                                                      \
      int Sl_ = 1;                                                                                                     \
      _Pragma("unroll") for (int k_ = 0; k_ < kMaxLevels; ++k_)                                                        \
        if (l_ == k_) Sl_ = xaxis_ ? (g).W[k_] : (g).H[k_];                                                            \
      const int S0_ = xaxis_ ? (g).W[0] : (g).H[0], E_ = xaxis_ ? (g).EX : (g).EY;                                     \
      const int r_ = (xaxis_ ? (rx) : (ry)) + ((kind_ == 1 || kind_ == 3) ? 1 : 0);                                    \
      const int q_ = floor_div(2 * E_ * r_ * Sl_ - S0_ + (kind_ < 4 ? 2 * S0_ - 1 : 0), 2 * S0_);                      \
      (s_geo)[lane] = kind_ < 4 ? (q_ < 0 ? 0 : (q_ > Sl_ ? Sl_ : q_)) : q_ - kMarginLo; /* region_begin | origin */   \
    }                                                                                                                  \
    wave_lds_sync();                                                                                                   \
    int tot_ = 0, mine_ = 0;                                                                                           \
    _Pragma("unroll") for (int k_ = 0; k_ < kMaxLevels; ++k_) {                                                        \
      if (k_ == (lane)) mine_ = tot_;                                                                                  \
      if (k_ < (g).levels)                                                                                             \
        tot_ += ((s_geo)[6 * k_ + 1] - (s_geo)[6 * k_]) * ((s_geo)[6 * k_ + 3] - (s_geo)[6 * k_ + 2]);                 \
    }                                                                                                                  \
    _Pragma("unroll") for (int l_ = 0; l_ < kMaxLevels; ++l_)                                                          \
      if ((lane) == l_) {                                                                                              \
        const int qxa_ = (s_geo)[6 * l_], qya_ = (s_geo)[6 * l_ + 2];                                                  \
        const int qdx_ = l_ < (g).levels ? (s_geo)[6 * l_ + 1] - qxa_ : 0;                                             \
        (s_tab)[4 * l_ + 0] = i32x4{mine_, qxa_, qya_, qdx_};                                                          \
        (s_tab)[4 * l_ + 1] = i32x4{as_i32(1.0f / (float)(qdx_ > 0 ? qdx_ : 1)), (g).start[l_], (g).W[l_], (g).H[l_]}; \
        (s_tab)[4 * l_ + 2] = i32x4{(g).ww[l_], (g).wh[l_], (s_geo)[6 * l_ + 4], (s_geo)[6 * l_ + 5]};                 \
        (s_tab)[4 * l_ + 3] = i32x4{BASE(l_), tot_, 0, 0};                                                             \
      }                                                                                                                \
  } while (0)

}  // namespace
}  // namespace dskd
