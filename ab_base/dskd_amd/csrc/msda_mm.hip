// MSDeformAttn backward of the encoder shape, grad_value of the COARSE levels (1-3) on the matrix cores (bf16 path).
//
// Replaces, for those levels, the grad_value half of the backward of ext-mmcv `MultiScaleDeformableAttnFunction` reached
// from the encoder at mmdet/models/utils/transformer.py:985-995 (same semantics as msda.hip).
//
// With S[q, x] = attn * bilinear weight of value pixel x for query q (one (head, level)),
//     grad_value[x, :] = sum_q S[q, x] * grad_out[q, :]          (S^T G)
// The scatter form (msda_bwd_value_kernel) pays one LDS atomic per (corner, channel) -- 2 048 lane-adds per
// (query, head) over the four levels, issue-bound at 5.2 cycles per 64: 117 us for level 1 and 128 us for levels 2+3 at
// B=4.  In the encoder a region's queries sample a small window of each coarse level (region footprint + margins), so S
// restricted to (16 queries) x (window) is a few per cent dense and S^T G is one MFMA per 32 window pixels and 16
// queries: ~0.3 GFLOP per (region, head) of padded work on the otherwise idle matrix pipe, while the LDS sees 4 integer
// atomics per sample instead of 128.
//
//   workgroup = (image, region of <= 32 x 32 level-0 pixels, head), 8 waves.  The window pixels of the handled levels
//   are cut into 32-pixel tiles owned by the PRODUCT waves: a tile's grad_value [32 px x 32 ch] accumulates in 16
//   registers per lane over all queries of the region and is added to HBM once at the end (float atomics: the windows of
//   neighbouring regions overlap; 32-pixel regions because that volume -- 62-66 MB per launch at B=4 against a chip-wide
//   atomic rate of ~1.3 TB/s -- is the launch's floor; 16-pixel regions measured 152 MB = 117 of 247 us).
//   Queries go through in chunks of 16, software-pipelined over two S images with ONE barrier per chunk:
//     sampler wave(s) (one per handled level; lane = (query, point)), chunk c+1: bilinear weights x attn -> entries of the
//         S image [px][16 q], accumulated as 16-bit FIXED POINT with no-return ds_add_u32 on the containing word (points
//         of a query that share a corner just add up; a dependent 2-byte read-modify-write chain per corner cost 2 400
//         cycles per chunk);
//     stager (one wave), chunk c+1: the chunk's grad_out rows -> [ch][q] f16 image;
//     product waves, chunk c: per owned tile, S tile -> f16 fragment, v_mfma_f32_32x32x16_f16 into the tile's
//         accumulators, S tile zeroed behind the read.
//   The loads of a role are unconditional and eight chunks ahead (conditional loads make the compiler wait for the
//   NEWEST outstanding load before the oldest may be used: one HBM latency per chunk).
//   A sample outside the window (learned offsets beyond the margin), or whose quad of attention weights is not in
//   [0, 1.5] (see below), is added by its lane alone with global atomics -- exact for ANY input.
//   grad_loc / grad_attn of every level stay with the gather kernel (msda_bwd_win_kernel): forming the dot products
//   here as a second product D = V G^T was built and measured -- 45 KB of LDS writes per chunk for 1 KB of use, 312 us per
//   launch -- and dropped.
//
// Numerics: the product runs in F16.  An entry of S is the fixed-point sum (quantum 2^-15: the four attention weights
// of a (query, head, level) must be >= 0 and sum to <= 1.5 -- softmax outputs do) converted to f16, i.e. weights rounded
// to 11 bits (bf16 weights -- 8 bits -- were measured just outside the bf16 path's tolerance on the coarsest level, where
// a cell sums > 1 000 contributions); grad_out, scaled by a power of two per (region, head) so that its largest element
// sits at 2^13 (from the gather kernel's statistics), converts exactly (bf16 has fewer mantissa bits than f16; elements
// below 2^-27 of the region's maximum lose bits, far below the f32 accumulator's own rounding).  Accumulation in f32.
// The f32 path keeps the windowed fixed-point kernels.
#include "common.h"
#include "msda_internal.h"
#include "msda_geom.h"
#include <type_traits>

namespace dskd {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMmWaves = 8;
constexpr int kMmQ = 16;             // queries per chunk
constexpr int kMmRegion = 32;        // region edge, level-0 pixels
constexpr int kMmTpw = 4;            // tiles per product wave
constexpr int kMmPre = 8;            // chunks whose loads are in flight (HBM latency under this access pattern: ~2 us)
constexpr size_t kMmMaxLds = 160 * 1024;

struct MmGeom {
  int lv0, nlv;              // handled levels [lv0, lv0 + nlv)
  int base[kMaxLevels];      // first window position of each handled level (others: beyond every position)
  int npos;                  // window positions of the handled levels
  int ntiles;                // 32-position tiles
};

// byte offset of S[px][q] / G^T[ch][q] (rows of 16 two-byte entries = 32 B; the two 16-byte halves of a row are swapped
// on every other group of 8 rows, so that the 16 lanes of one ds_read_b128 phase -- rows r .. r+15, one half -- cover 64 banks)
__device__ __forceinline__ int img16_off(int row, int q) {
  return row * 32 + ((((q >> 3) ^ (row >> 3)) & 1) << 4) + (q & 7) * 2;
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned u) { return as_f32(u << 16); }

#ifdef DSKD_MM_PROFILE
// per-phase shader-clock totals: [role: 0 = wave 0 (sampler), 16 = wave 4 (products only)][slot]
__device__ unsigned long long g_mmprof[32];
#define MMPROF(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); t_acc[slot] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define MMPROF(slot)
#endif

__global__ __launch_bounds__(kMmWaves * 64, 4) void msda_bwd_mm_kernel(
    const float* __restrict__ loc, const float* __restrict__ attn, const __bf16* __restrict__ grad_out,
    float* __restrict__ grad_value, const float* __restrict__ stats, int sRX, int sRY, int sEX, int sEY, ValueGeom g,
    MmGeom mga, MmGeom mgb, int ngroups, int Nq, int points) {
  constexpr int LP = 16;
  constexpr int NW = kMmWaves;
  constexpr int TPW = kMmTpw;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef DSKD_MM_PROFILE
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

  int vb = xcd_remap(blockIdx.x, gridDim.x);
  // one launch serves both level groups (level 1 | levels 2+3): 2 x 768 workgroups fill 512 slots in 3 rounds where two
  // launches of 768 took 2 x 2.  The group's geometry becomes scalars here (kernel-argument arrays must not be
  // indexed at run time).
  const bool second = ngroups > 1 && (vb & 1);
  if (ngroups > 1) vb >>= 1;
  struct { int lv0, nlv, npos, ntiles, base[kMaxLevels]; } mg;
  mg.lv0 = second ? mgb.lv0 : mga.lv0;
  mg.nlv = second ? mgb.nlv : mga.nlv;
  mg.npos = second ? mgb.npos : mga.npos;
  mg.ntiles = second ? mgb.ntiles : mga.ntiles;
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) mg.base[l] = second ? mgb.base[l] : mga.base[l];
  const int NPX = mg.ntiles * 32;
  char* s_S = smem;                                             // [2][NPX][16 q] u16 fixed point, swizzled (img16_off)
  char* s_gt = s_S + 2 * NPX * 32;                              // [2][32 ch][16 q] f16 (scaled), swizzled
  int* s_row = reinterpret_cast<int*>(s_gt + 2 * kCh * 32);     // [NPX] window position -> value row of the image | -1
  i32x4* s_tab = reinterpret_cast<i32x4*>(s_row + NPX);         // [4][4] lookup rows (msda_geom.h)
  int* s_geo = reinterpret_cast<int*>(s_tab + 4 * kMaxLevels);  // [24]

  const int h = vb & 7; vb >>= 3;
  const int rx = vb % g.RX; vb /= g.RX;
  const int ry = vb % g.RY;
  const int b = vb / g.RY;

#define MM_BASE(l) mg.base[l]
  if (wave == 0) DSKD_REGION_TABLES(g, rx, ry, lane, MM_BASE, s_tab, s_geo);
#undef MM_BASE
  for (int i = tid; i < NPX * 4; i += NW * 64) reinterpret_cast<u32x4*>(s_S)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  int cum[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) cum[l] = s_tab[4 * l].x;
  const int nq = s_tab[3].y;

  // ---- window position -> value row (pixels outside the image: -1; their S entries are never flushed)
  for (int p = tid; p < NPX; p += NW * 64) {
    int row = -1;
    if (p < mg.npos) {
      int l = mg.lv0;
#pragma unroll
      for (int k = 1; k < kMaxLevels; ++k) l += (k > mg.lv0 && p >= mg.base[k]) ? 1 : 0;
      const i32x4 lb = s_tab[4 * l + 1], lc = s_tab[4 * l + 2];
      const int rel = p - s_tab[4 * l + 3].x;
      const int wwl = lc.x;
      const int wy = (int)(((float)rel + 0.5f) / (float)wwl), wx = rel - wy * wwl;
      const int gx = lc.z + wx, gy = lc.w + wy;
      if (gx >= 0 && gx < lb.z && gy >= 0 && gy < lb.w) row = lb.y + gy * lb.z + gx;
    }
    s_row[p] = row;
  }

  // power-of-two scale of grad_out for the f16 product: the region's max |grad_out| -> 2^13.  The maximum comes from the
  // (16 x 16-pixel region, head) cells of the gather kernel that cover this region's level-0 pixel range (a query's region
  // is its centre's level-0 pixel / edge in both kernels, so their union contains every query of ours); every wave that
  // needs the scale reads those few cells itself, where the load's latency hides behind its wait for the first chunk.
  float go_scale = 1.f, go_inv = 1.f;
  auto set_scale = [&]() {
    const int px0 = rx * g.EX, px1 = min((rx + 1) * g.EX, g.W[0]) - 1;
    const int py0 = ry * g.EY, py1 = min((ry + 1) * g.EY, g.H[0]) - 1;
    const int gx0 = px0 / sEX, gx1 = min(px1 / sEX, sRX - 1), gy0 = py0 / sEY, gy1 = min(py1 / sEY, sRY - 1);
    const int nx = gx1 - gx0 + 1, ncell = nx * (gy1 - gy0 + 1);
    float m = 0.f;
    for (int i = lane; i < ncell; i += 64) {
      const int iy = i / nx, ix = i - iy * nx;
      const float v = stats[((((size_t)b * sRY + gy0 + iy) * sRX + gx0 + ix) * kHeads + h) * 4];
      m = (v != v) ? v : fmaxf(m, v);            // keep a NaN
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(m, o);
      m = (m != m) ? m : ((m2 != m2) ? m2 : fmaxf(m, m2));
    }
    const int eb = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu);          // biased exponent
    if (eb > 0 && eb < 255) {                                                       // finite, normal
      int se = 127 + 13 - (eb - 127);
      se = se < 27 ? 27 : (se > 227 ? 227 : se);
      go_scale = as_f32((unsigned)se << 23);
      go_inv = as_f32((unsigned)(254 - se) << 23);
    }
  };

  const int nlv = mg.nlv;
  const bool sampler = wave < nlv;               // wave-uniform roles
  const bool stager = wave == nlv;
  const int npw = NW - nlv - 1;                  // product waves: nlv + 1 .. NW - 1
  const int lvl = mg.lv0 + (sampler ? wave : 0);
  const int ql = lane >> 2, pt = lane & 3;       // (query of the chunk, point | 16-byte part of the grad_out row)
  // column of query ql in the S / G^T images: its 4 bits reversed.  Neighbouring queries of a row sample the same pixel
  // of a coarse level (2, 4 or 8 of them), and with consecutive columns their entries would share 32-bit words: LDS
  // atomics of one instruction on one address are serialised
  const int qcol = ((ql & 1) << 3) | ((ql & 2) << 1) | ((ql & 4) >> 1) | ((ql & 8) >> 3);
  float* gvb = grad_value + (size_t)b * Nq * (kHeads * kCh) + h * kCh;
  const __bf16* gob = grad_out + (size_t)b * Nq * (kHeads * kCh) + h * kCh;
  const int nchunks = (nq + kMmQ - 1) / kMmQ;
  MMPROF(0);                 // prologue

  if (sampler) {
    // =================================================================== sampler: S image of chunk c + 1 during chunk c
    // One wave per level carries ~200 dependent-ish vector instructions per chunk and is the workgroup's critical path
    // (the product waves wait for it at the barrier): its instructions go first on the SIMD it shares with them.
    __builtin_amdgcn_s_setprio(3);
    int p_qg[kMmPre];
    f32x2 p_xy[kMmPre];
    float p_a[kMmPre];
    // The loads are issued as asm and waited for with COUNTED s_waitcnt (2 loads per chunk, kMmPre chunks in flight): left
    // to the compiler, the loop header waits for vmcnt(0) -- the newest load -- in every iteration (~1 800 cycles of HBM
    // latency per chunk).  No other vector-memory operation is issued by this wave inside the loop.
    auto fetch = [&](int c, int& n_qg, f32x2& n_xy, float& n_a) {
      const int qi = c * kMmQ + ql;
      const int qc = region_query(s_tab, cum, qi < nq ? qi : nq - 1);
      const size_t base = (((size_t)b * Nq + qc) * kHeads + h) * (size_t)LP + lvl * points + pt;
      asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(n_xy) : "v"(loc + base * 2) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=&v"(n_a) : "v"(attn + base) : "memory");
      n_qg = qi < nq ? qc : -1;
    };
#pragma unroll
    for (int i = 0; i < kMmPre; ++i) fetch(i, p_qg[i], p_xy[i], p_a[i]);
    const i32x4 lb = s_tab[4 * lvl + 1], lc = s_tab[4 * lvl + 2];
    const int H = lb.w, W = lb.z, st = lb.y;
    const float Wf = (float)W, Hf = (float)H;
    const int wwl = lc.x, whl = lc.y, wbase = s_tab[4 * lvl + 3].x;
    // One sample: window pass (FB == false) -> its four fixed-point entries into S, returns true when the sample has to
    // take the per-lane path instead; fallback pass (FB == true) -> exactly those samples, straight to HBM.
    auto sample = [&](auto fb_pass, int qg, f32x2 xy, float a, bool quad_ok, char* S) -> bool {
      constexpr bool FB = decltype(fb_pass)::value;
      if (qg < 0) return false;
      const float x = xy.x * Wf - 0.5f;
      const float y = xy.y * Hf - 0.5f;
      if (!(x > -1.f && y > -1.f && x < Wf && y < Hf)) return false;   // as point_params (msda.hip); NaN fails
      const float xf = floorf(x), yf = floorf(y);
      const int x0 = (int)xf, y0 = (int)yf;
      const float lx = x - xf, ly = y - yf;
      const float hx = 1.f - lx, hy = 1.f - ly;
      const bool vx0 = x0 >= 0, vx1 = x0 + 1 <= W - 1, vy0 = y0 >= 0, vy1 = y0 + 1 <= H - 1;
      const f32x4 w = f32x4{(vy0 && vx0) ? hy * hx * a : 0.f, (vy0 && vx1) ? hy * lx * a : 0.f,
                            (vy1 && vx0) ? ly * hx * a : 0.f, (vy1 && vx1) ? ly * lx * a : 0.f};
      const int wx = x0 - lc.z, wy = y0 - lc.w;
      const bool inwin = quad_ok && wx >= 0 && wx + 1 < wwl && wy >= 0 && wy + 1 < whl;
      if constexpr (!FB) {
        if (!inwin) return true;
        const int pb = wbase + wy * wwl + wx;
        const int sh = (qcol & 1) * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float wk = k == 0 ? w.x : (k == 1 ? w.y : (k == 2 ? w.z : w.w));
          const int pk = pb + (k & 1) + (k >> 1) * wwl;
          const unsigned fx = (unsigned)cvt_round(wk * 32768.f);
          if (fx != 0u) atomicAdd(reinterpret_cast<unsigned*>(S + (img16_off(pk, qcol) & ~3)), fx << sh);
        }
        return false;
      } else {
        if (inwin) return false;
        const int r00 = st + y0 * W + x0;
        const __bf16* gor = gob + (size_t)qg * (kHeads * kCh);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float wk = k == 0 ? w.x : (k == 1 ? w.y : (k == 2 ? w.z : w.w));
          const bool ok = k == 0 ? (vy0 && vx0) : (k == 1 ? (vy0 && vx1) : (k == 2 ? (vy1 && vx0) : (vy1 && vx1)));
          if (!ok) continue;
          float* dst = gvb + (size_t)(r00 + (k & 1) + (k >> 1) * W) * (kHeads * kCh);
          for (int ch = 0; ch < kCh; ++ch) atomicAdd(dst + ch, wk * (float)gor[ch]);
        }
        return true;
      }
    };
    // fixed-point entries: the 4 weights of a (query, head, level) -- one quad of lanes -- must be >= 0 and sum to <= 1.5
    // (then an entry stays below 2^16 whatever corners coincide); otherwise (and for NaN) the per-lane path
    auto quad_fits = [](float a) {
      const float s_abs = group4_sum(fabsf(a)), s_sgn = group4_sum(a);
      return s_abs == s_sgn && s_abs <= 1.5f;
    };
    bool any_fb = false;
    // iteration c builds chunk c; the last one (c == nchunks) only syncs.  Unrolled by the prefetch depth so that the
    // slots are addressed statically: rotating them through registers makes every iteration wait for its newest load.
    // The per-lane path is NOT in this loop (a possible path with > 63 memory operations makes the compiler's wait-count
    // insertion give up counting: every iteration then waited for its newest load, ~1 800 cycles): such samples are only
    // flagged here and handled by a second walk below.
    for (int c0 = 0; c0 <= nchunks; c0 += kMmPre)
#pragma unroll
    for (int u = 0; u < kMmPre; ++u) {
      const int c = c0 + u;
      if (c > nchunks) break;                    // uniform over the workgroup
      if (c < nchunks) {
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(p_xy[u].x), "+v"(p_xy[u].y), "+v"(p_a[u]) : "n"(2 * (kMmPre - 1)) : "memory");
        const int qg = p_qg[u];
        const f32x2 xy = p_xy[u];
        const float a = qg >= 0 ? p_a[u] : 0.f;
        MMPROF(3);             // (profile) prefetched data in registers
        any_fb |= sample(std::false_type{}, qg, xy, a, quad_fits(a), s_S + (c & 1) * NPX * 32);
        MMPROF(4);             // (profile) weights + LDS atomics (+ fallback)
        fetch(c + kMmPre, p_qg[u], p_xy[u], p_a[u]);
        MMPROF(5);             // (profile) next loads issued
      }
      MMPROF(1);
      __syncthreads();
      MMPROF(2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead loads of the last iterations land in dead registers
    if (__builtin_amdgcn_ballot_w64(any_fb) != 0ull) {
      // ---- samples outside the window / with an unusual weight quad: a second walk over the region's chunks
      for (int c = 0; c < nchunks; ++c) {
        int qg; f32x2 xy; float a;
        fetch(c, qg, xy, a);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xy.x), "+v"(xy.y), "+v"(a) :: "memory");
        a = qg >= 0 ? a : 0.f;
        const bool took = sample(std::true_type{}, qg, xy, a, quad_fits(a), nullptr);
#ifdef DSKD_MM_PROFILE
        if (took) atomicAdd(&g_mmprof[30], 1ull);
#else
        (void)took;
#endif
      }
    }
  } else if (stager) {
    // =================================================================== stager: grad_out image of chunk c during chunk c - 1
    set_scale();
    int p_qg[kMmPre];
    u32x4 p_go[kMmPre];
    auto fetch = [&](int c, int& n_qg, u32x4& n_go) {      // asm load + counted wait, as in the sampler
      const int qi = c * kMmQ + ql;
      const int qc = region_query(s_tab, cum, qi < nq ? qi : nq - 1);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(n_go) : "v"(gob + (size_t)qc * (kHeads * kCh) + pt * 8) : "memory");
      n_qg = qi < nq ? qc : -1;
    };
#pragma unroll
    for (int i = 0; i < kMmPre; ++i) fetch(i, p_qg[i], p_go[i]);
    for (int c0 = 0; c0 <= nchunks; c0 += kMmPre)
#pragma unroll
    for (int u = 0; u < kMmPre; ++u) {
      const int c = c0 + u;
      if (c > nchunks) break;                    // uniform over the workgroup
      if (c < nchunks) {
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(p_go[u].x), "+v"(p_go[u].y), "+v"(p_go[u].z), "+v"(p_go[u].w) : "n"(kMmPre - 1) : "memory");
        const u32x4 go4 = p_qg[u] >= 0 ? p_go[u] : u32x4{0u, 0u, 0u, 0u};
        char* GT = s_gt + (c & 1) * kCh * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ch = pt * 8 + j;
          const unsigned wd = (j >> 1) == 0 ? go4.x : ((j >> 1) == 1 ? go4.y : ((j >> 1) == 2 ? go4.z : go4.w));
          const float gf = bf16_bits_to_f32((j & 1) ? (wd >> 16) : (wd & 0xFFFFu));
          *reinterpret_cast<_Float16*>(GT + img16_off(ch, qcol)) = (_Float16)(gf * go_scale);
        }
        fetch(c + kMmPre, p_qg[u], p_go[u]);
      }
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead loads of the last iterations land in dead registers
  } else {
    // =================================================================== product waves
    set_scale();
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int pw = wave - nlv - 1;               // tile t = pw + npw * i
    for (int c = 0; c <= nchunks; ++c) {         // iteration c multiplies chunk c - 1
      MMPROF(1);
      if (c > 0) {
        char* S = s_S + ((c - 1) & 1) * NPX * 32;
        const f16x8 gt = *reinterpret_cast<const f16x8*>(s_gt + ((c - 1) & 1) * kCh * 32 + img16_off(lane & 31, (lane >> 5) * 8));
        // all of the wave's S tiles are requested before the first is zeroed or used (a zeroing store between two reads
        // would pin their order: the compiler cannot tell the tiles apart)
        u32x4 sr[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int t = pw + npw * i;
          sr[i] = u32x4{0u, 0u, 0u, 0u};
          if (t < mg.ntiles) sr[i] = *reinterpret_cast<const u32x4*>(S + img16_off(t * 32 + (lane & 31), (lane >> 5) * 8));
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int t = pw + npw * i;
          if (t < mg.ntiles) {                   // wave-uniform
            // every 16-byte half of the tile belongs to exactly one lane: zeroed behind its read
            *reinterpret_cast<u32x4*>(S + img16_off(t * 32 + (lane & 31), (lane >> 5) * 8)) = u32x4{0u, 0u, 0u, 0u};
            // grad_value tile [m = px][n = ch] += S^T[px][q] G[q][ch]: lane (m = l & 31, hh = l >> 5) holds q = 8 hh ..
            f16x8 sf;                             // fixed point -> f16 (v_cvt_f16_u16); the 2^-15 is folded into the flush
            sf[0] = (_Float16)(unsigned short)(sr[i].x & 0xFFFFu); sf[1] = (_Float16)(unsigned short)(sr[i].x >> 16);
            sf[2] = (_Float16)(unsigned short)(sr[i].y & 0xFFFFu); sf[3] = (_Float16)(unsigned short)(sr[i].y >> 16);
            sf[4] = (_Float16)(unsigned short)(sr[i].z & 0xFFFFu); sf[5] = (_Float16)(unsigned short)(sr[i].z >> 16);
            sf[6] = (_Float16)(unsigned short)(sr[i].w & 0xFFFFu); sf[7] = (_Float16)(unsigned short)(sr[i].w >> 16);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sf, gt, acc[i], 0, 0, 0);
          }
        }
      }
      MMPROF(3);
      __syncthreads();
      MMPROF(2);
    }
    MMPROF(6);
    // ---- flush: tile [px][ch] -> grad_value rows; lane = channel (l & 31), 16 positions per lane:
    // every atomic instruction adds two whole 128-byte runs
    go_inv *= 1.f / 32768.f;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int t = pw + npw * i;
      if (t < mg.ntiles) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int row = s_row[p];
          const float v = acc[i][r];
          if (row >= 0 && v != 0.f) atomicAdd(gvb + (size_t)row * (kHeads * kCh) + (lane & 31), v * go_inv);
        }
      }
    }
    MMPROF(7);
  }
#ifdef DSKD_MM_PROFILE
  if (lane == 0 && (wave == 0 || wave == 4))
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_mmprof[(wave == 0 ? 0 : 16) + i], t_acc[i]);
#endif
}

bool make_mm_geom(const MsdaLevels& lg, int levels, int points, int Nq, int lv0, int nlv, ValueGeom* g, MmGeom* mg,
                  size_t* lds_bytes) {
  if (levels != 4 || points != 4 || lv0 < 1 || nlv < 1 || nlv > 2 || lv0 + nlv > levels) return false;
  int tot = 0;
  for (int l = 0; l < levels; ++l) tot += lg.H[l] * lg.W[l];
  if (tot != Nq) return false;
  for (int l = 0; l < levels; ++l)
    if (lg.start[l] != (l == 0 ? 0 : lg.start[l - 1] + lg.H[l - 1] * lg.W[l - 1])) return false;
  const int W0 = lg.W[0], H0 = lg.H[0];
  g->levels = levels;
  g->RX = (W0 + kMmRegion - 1) / kMmRegion;
  g->RY = (H0 + kMmRegion - 1) / kMmRegion;
  g->EX = (W0 + g->RX - 1) / g->RX;
  g->EY = (H0 + g->RY - 1) / g->RY;
  int npos = 0;
  mg->lv0 = lv0; mg->nlv = nlv;
  for (int l = 0; l < levels; ++l) {
    if (lg.W[l] > W0 || lg.H[l] > H0) return false;   // level 0 must be the finest
    g->H[l] = lg.H[l]; g->W[l] = lg.W[l]; g->start[l] = lg.start[l];
    g->ww[l] = (g->EX * lg.W[l] + W0 - 1) / W0 + 1 + kMarginLo + kMarginHi;
    g->wh[l] = (g->EY * lg.H[l] + H0 - 1) / H0 + 1 + kMarginLo + kMarginHi;
    const bool mine = l >= lv0 && l < lv0 + nlv;
    mg->base[l] = mine ? npos : 0x3FFFFFFF;
    if (mine) npos += g->ww[l] * g->wh[l];
  }
  mg->npos = npos;
  mg->ntiles = (npos + 31) / 32;
  if (mg->ntiles > (kMmWaves - nlv - 1) * kMmTpw) return false;
  const size_t NPX = (size_t)mg->ntiles * 32;
  *lds_bytes = 2 * NPX * 32 + 2 * kCh * 32 + NPX * 4 + 16 * 4 * kMaxLevels + sizeof(int) * 6 * kMaxLevels + 16 + 16 + 16;
  return *lds_bytes <= kMmMaxLds;
}

}  // namespace

bool mm_supported(const MsdaLevels& lg, int levels, int points, int Nv, int Nq, int dtype, int lv0, int nlv) {
  if (dtype != DSKD_DTYPE_BF16 || Nv != Nq) return false;
  ValueGeom g;
  MmGeom mg;
  size_t lds;
  return make_mm_geom(lg, levels, points, Nq, lv0, nlv, &g, &mg, &lds);
}

int launch_bwd_mm(const float* loc, const float* attn, const void* grad_out, float* grad_value, const float* stats,
                  const int* sgrid, const MsdaLevels& lg, int mask, int B, int Nq, int points, hipStream_t st) {
  if (!stats || !sgrid) return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_bwd: matrix-core backward needs the gather kernel's statistics");
  // mask: bit 1 = level 1, bit 2 = levels 2+3; both groups go out as ONE launch (interleaved workgroups)
  ValueGeom g;
  MmGeom mg[2];
  size_t lds[2] = {0, 0};
  int ng = 0;
  if (mask & 2) {
    if (!make_mm_geom(lg, 4, points, Nq, 1, 1, &g, &mg[ng], &lds[ng]))
      return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_bwd: matrix-core backward does not take level 1");
    ++ng;
  }
  if (mask & 4) {
    if (!make_mm_geom(lg, 4, points, Nq, 2, 2, &g, &mg[ng], &lds[ng]))
      return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_bwd: matrix-core backward does not take levels 2+3");
    ++ng;
  }
  if (ng == 0) return DSKD_OK;
  if (ng == 1) mg[1] = mg[0];
  const size_t lds_max = lds[0] > lds[1] ? lds[0] : lds[1];
  auto kern = msda_bwd_mm_kernel;
  static bool done[64] = {};
  if (!reserve_lds((const void*)kern, (int)kMmMaxLds, done)) return fail(DSKD_ERR_LAUNCH, "dskd_msda_bwd: cannot reserve LDS");
  const dim3 grid((unsigned)(B * g.RY * g.RX * kHeads * ng)), block(kMmWaves * 64);
  hipLaunchKernelGGL(kern, grid, block, lds_max, st, loc, attn, (const __bf16*)grad_out, grad_value, stats, sgrid[0], sgrid[1],
                     sgrid[2], sgrid[3], g, mg[0], mg[1], ng, Nq, points);
  return DSKD_OK;
}

}  // namespace dskd

#ifdef DSKD_MM_PROFILE
extern "C" int dskd_debug_mm_prof(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dskd::g_mmprof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(dskd::g_mmprof), z, sizeof(z)); }
  return 0;
}
#endif
