// Multi-scale deformable attention backward, d(out)/d(value) for the ENCODER shape (queries ==
// pixels), as a tiled PULL: no atomics, no fixed point, no pre-zeroed output.
//
// Replaces the scatter half of ext-mmcv `MultiScaleDeformableAttnFunction.backward` (called from
// mmdet/models/utils/transformer.py:985-995 through autograd).  grad_value[cell] = sum over every
// (query, point, corner) that touches the cell of attn * bilinear weight * grad_out[query]: a
// sparse matrix (64 non-zeros per query and head) transposed.  The scatter form pays one atomic
// per (corner, channel) -- 2 048 lane-adds per (query, head); the windowed LDS kernel in msda.hip
// runs at the ds_add_u32 issue rate and needs fixed point because gfx950's LDS float atomics are
// serialised (msda.hip, "LDS float atomics").  On the fine levels a cell receives only ~20 (level
// 0) / ~85 (level 1) contributions, so here the transposition is done explicitly instead:
//
//   workgroup = (image, head, level, tile of 2^tws x 2^ths cells), 1 024 threads
//   candidates = the queries (of all four query levels) whose NOMINAL cell on this level -- the cell
//                under the query's own centre -- lies within M cells of the tile: four rectangles
//   scan       one lane per (candidate, point): bilinear weights as in the forward; every corner
//              that falls into the tile takes a slot in its cell's list with ONE returning LDS
//              counter add (counting sort: count -> prefix sum -> place), record = {byte offset
//              of the query's grad_out row, weight}
//   reduce     a group of LPR lanes (one 16-B part of the head's row each) owns a cell, walks the
//              cell's records: one 8-byte LDS read, one 16-byte buffer load of grad_out, FMAs
//              into registers (f32, same arithmetic as the oracle up to summation order);
//              R groups share a cell on coarser levels (longer lists) and are summed with DPP
//   store      every cell of the tile is written once, whole 128-B (cell, head) lines
//
// A corner in ANOTHER tile is that tile's business -- unless the sample strayed more than M cells
// from its query, so that the owning tile does not scan this query at all.  The query's HOME tile
// (the one holding its nominal cell) sees that and appends {row, query, weight, head} to a list in
// the caller's workspace; a small apply kernel adds those few with global float atomics AFTER the
// plain stores (kernel boundary).  The result is exact for ANY sampling locations; if the list
// overflows, the apply kernel re-scans everything and adds the strays directly.
#include "msda_internal.h"

#include <stdlib.h>

namespace dskd {
namespace {

constexpr int kHeads = 8;
constexpr int kCh = 32;
constexpr int kLP = 16;          // levels * points
constexpr int kPts = 4;
constexpr int kMaxThreads = 1024;
constexpr int kTabMax = 160;     // candidate extent per axis and query level
constexpr int kOOBg = 0x7F000000;

struct PullGeom {
  int H[kMsdaMaxLevels], W[kMsdaMaxLevels], start[kMsdaMaxLevels];
  int tl;               // target level of this launch
  int tws, ths;         // log2 of the tile width / height in cells
  int TX, TY;           // tiles per axis
  int M;                // candidate margin, cells of the target level
  int Nq;
  int nblocks;          // B * TY * TX * heads
};

struct FbHeader { unsigned count, overflow, ticket, pad; };
struct FbEntry { int row, gq; float w; int h; };
static_assert(sizeof(FbEntry) == kPullWsEntry, "fallback entry layout");

template <typename T> struct PT;
template <> struct PT<float> {
  static constexpr int LPRS = 3;     // log2(lanes per record): 8 lanes x 16 B = one 128-B head row
  static constexpr int NACC = 4;
  static constexpr int ROWB = 1024;
};
template <> struct PT<__bf16> {
  static constexpr int LPRS = 2;     // 4 lanes x 16 B = one 64-B head row
  static constexpr int NACC = 8;
  static constexpr int ROWB = 512;
};

__device__ __forceinline__ float as_f32u(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ int as_i32f(float f) { return __builtin_bit_cast(int, f); }

__host__ __device__ inline int floor_div_i(int a, int b) {   // b > 0
  const int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}
// Nominal cell of query coordinate xq (level extent Sj) on a level of extent Sl, clamped into the level:
// floor((xq + 0.5) * Sl / Sj - 0.5).  Monotone in xq.
__host__ __device__ inline int nominal_cell(int xq, int Sj, int Sl) {
  const int v = floor_div_i((2 * xq + 1) * Sl - Sj, 2 * Sj);
  return v < 0 ? 0 : (v > Sl - 1 ? Sl - 1 : v);
}

using i32x2 = __attribute__((ext_vector_type(2))) int;

template <typename T>
__device__ __forceinline__ void load_g(__amdgpu_buffer_rsrc_t rsrc, int off, float* f) {
  const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
  if constexpr (sizeof(T) == 4) {
    f[0] = as_f32u(v.x); f[1] = as_f32u(v.y); f[2] = as_f32u(v.z); f[3] = as_f32u(v.w);
  } else {
    const unsigned a = v.x, b = v.y, c = v.z, d = v.w;
    f[0] = as_f32u(a << 16); f[1] = as_f32u(a & 0xFFFF0000u);
    f[2] = as_f32u(b << 16); f[3] = as_f32u(b & 0xFFFF0000u);
    f[4] = as_f32u(c << 16); f[5] = as_f32u(c & 0xFFFF0000u);
    f[6] = as_f32u(d << 16); f[7] = as_f32u(d & 0xFFFF0000u);
  }
}

#ifdef DSKD_PULL_PROFILE
__device__ unsigned long long g_pprof[32];
#define PPROF(slot) do { if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_pprof[g.tl * 8 + (slot)], t_ - t_prev); t_prev = t_; } } while (0)
#else
#define PPROF(slot)
#endif

// Static LDS of a tile pass (both the pull and the re-scan use it).
// Per scan pass NT / 2 candidates are looked at (two lanes each, 2 points per lane); NT hit samples are sorted and
// reduced together (one per lane).  NT = threads per workgroup.
template <int NT>
struct TileTables {
  i32x4 lv[kMsdaMaxLevels];                      // {H, W, start, 0}
  int bnd[16];                                   // per query level: x lo, x hi, y lo, y hi of the candidate rectangle
  i32x4 cand[kMsdaMaxLevels];                    // {cum, qx0, qy0, qw}
  float inv_qw[kMsdaMaxLevels];
  int ncand;
  unsigned qn;                                   // this pass: hit samples (low 16 bits), candidates with hits (high 16)
  unsigned short tabx[kMsdaMaxLevels][kTabMax];  // nominal cell (x) of candidate column i of query level j
  unsigned short taby[kMsdaMaxLevels][kTabMax];
  unsigned queue[NT / 2 * kPts];                 // hit samples of the pass: slot << 16 | candidate-of-pass << 2 | point
  unsigned short slot2cand[NT / 2];              // candidates with hits, in queue order
};

// The tables, built cooperatively.  Needs a barrier before the first use.
template <int NT>
__device__ __forceinline__ void build_tables(TileTables<NT>& tt, const PullGeom& g, int X0, int Y0, int TW, int TH) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int l = 0; l < kMsdaMaxLevels; ++l)
    if (tid == l) tt.lv[l] = i32x4{g.H[l], g.W[l], g.start[l], 0};
  __syncthreads();
  const i32x4 lt = tt.lv[g.tl];
  const int Hl = lt.x, Wl = lt.y;
  if (tid < 16) {
    const int j = tid >> 2, kind = tid & 3;
    const bool xaxis = kind < 2;
    const i32x4 lj = tt.lv[j];
    const int Sj = xaxis ? lj.y : lj.x, Sl = xaxis ? Wl : Hl;
    const int A = (xaxis ? X0 : Y0) - g.M;
    const int Bv = (xaxis ? X0 + TW - 1 : Y0 + TH - 1) + g.M;
    const int target = (kind & 1) ? Bv + 1 : A;      // smallest coordinate whose nominal cell is >= target
    int lo = 0, hi = Sj;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (nominal_cell(mid, Sj, Sl) >= target) hi = mid; else lo = mid + 1;
    }
    tt.bnd[tid] = lo;
  }
  __syncthreads();
  if (tid == 0) {
    int cum = 0;
#pragma unroll
    for (int j = 0; j < kMsdaMaxLevels; ++j) {
      const int qx0 = tt.bnd[4 * j], qw = tt.bnd[4 * j + 1] - qx0;
      const int qy0 = tt.bnd[4 * j + 2], qh = tt.bnd[4 * j + 3] - qy0;
      tt.cand[j] = i32x4{cum, qx0, qy0, qw};
      tt.inv_qw[j] = 1.0f / (float)(qw > 0 ? qw : 1);
      cum += (qw > 0 && qh > 0) ? qw * qh : 0;
    }
    tt.ncand = cum;
  }
  __syncthreads();
  for (int i = tid; i < 2 * kMsdaMaxLevels * kTabMax; i += NT) {
    const int j = i / (2 * kTabMax), r = i - j * (2 * kTabMax);
    const bool xaxis = r < kTabMax;
    const int k = xaxis ? r : r - kTabMax;
    const i32x4 lj = tt.lv[j];
    const int ext = xaxis ? tt.bnd[4 * j + 1] - tt.bnd[4 * j] : tt.bnd[4 * j + 3] - tt.bnd[4 * j + 2];
    if (k < ext) {
      const int c0 = xaxis ? tt.bnd[4 * j] : tt.bnd[4 * j + 2];
      const int v = nominal_cell(c0 + k, xaxis ? lj.y : lj.x, xaxis ? Wl : Hl);
      if (xaxis) tt.tabx[j][k] = (unsigned short)v; else tt.taby[j][k] = (unsigned short)v;
    }
  }
}

// One tile.  RESCAN == false: the pull proper.  RESCAN == true (list overflow only): no records; the
// home lane of a stray corner adds its 32 channels with global atomics itself.
//
// Everything here is paced by the texture path (one access per 64-B segment and lane group, ~2.4 cycles
// each per CU), so the tile is walked in passes of 512 candidates that touch global memory as little as possible:
//   A  two lanes per candidate read its 32 B of sampling locations on this level (ONE segment) and keep only the
//      samples that can matter -- a corner inside the tile, or a sample of a HOME query that strayed beyond the
//      margin -- in a queue; candidates with a hit get a slot.  The scan is M-margin redundant (a 16 x 16 tile looks
//      at 28 x 28 cells' worth of queries): this cheap test runs on everything, the rest only on the ~1/3 that hit.
//   S  (bf16) the grad_out row of every candidate with a hit is copied into LDS ONCE per pass (one segment per
//      candidate); the reduce phase then gathers from LDS.  A sample's four corners land in four cells and a
//      query has four points per level: from global memory the same 64 bytes would be fetched ~14 times.
//   B  per queued sample: weights as in the forward, a slot in each hit cell's list (one returning LDS add),
//      strays to the global list
//   sort (prefix sum, place) and reduce as described at the top of the file.
// Experiment build (-DDSKD_PULL_COMPACT): the tile kernel reads loc / attn of ITS level from a compact
// level- and head-major copy [B][heads][Nq][4 points] (32 + 16 bytes per (query, head), neighbouring candidates in
// neighbouring bytes) instead of the interleaved [B][Nq][heads][16] lines (128 + 64 bytes, 32 + 16 of them used) -- the
// best case of the "level-major record" for this kernel; the copy is made by pull_compact_kernel in front of the launch.
// Measured: profiles/r04_msda_pull_compact_ab.txt.  Not part of the product build.
#ifdef DSKD_PULL_COMPACT
#define DSKD_PULL_SB(b, q, h, tl, pt) (RESCAN ? ((((size_t)(b) * g.Nq + (q)) * kHeads + (h)) * (size_t)kLP + (tl) * kPts + (pt)) \
                                             : ((((size_t)(b) * kHeads + (h)) * g.Nq + (q)) * (size_t)kPts + (pt)))
#else
#define DSKD_PULL_SB(b, q, h, tl, pt) ((((size_t)(b) * g.Nq + (q)) * kHeads + (h)) * (size_t)kLP + (tl) * kPts + (pt))
#endif
template <typename T, int NT, int R, bool RESCAN>
__device__ __forceinline__ void pull_tile(int vb, const float* __restrict__ loc, const float* __restrict__ attn,
                                          const T* __restrict__ grad_out, float* __restrict__ grad_value,
                                          const PullGeom& g, FbHeader* hdr, FbEntry* fb, unsigned fb_cap,
                                          TileTables<NT>& tt, int* s_cnt, i32x2* s_rec, char* s_gst) {
  using P = PT<T>;
  constexpr int LPR = 1 << P::LPRS;
  constexpr int NC = NT / (LPR * R);             // cells per tile
  constexpr int kPassCand = NT / 2, kSlice = NT;
  constexpr bool STAGE = sizeof(T) == 2;         // grad_out rows of the pass staged in LDS (64 B per candidate)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int h = vb & 7; vb >>= 3;
  const int tx = vb % g.TX; vb /= g.TX;
  const int ty = vb % g.TY;
  const int b = vb / g.TY;
  const int TW = 1 << g.tws, TH = 1 << g.ths;
  const int X0 = tx << g.tws, Y0 = ty << g.ths;
  const int tl = g.tl;
  const int M = g.M;
#ifdef DSKD_PULL_PROFILE
  unsigned long long t_prev = clock64();
#endif

  build_tables<NT>(tt, g, X0, Y0, TW, TH);
  __syncthreads();
  const i32x4 lt = tt.lv[tl];
  const int Hl = lt.x, Wl = lt.y, stl = lt.z;
  PPROF(0);
  const int ncand = tt.ncand;
  int cum[kMsdaMaxLevels];
#pragma unroll
  for (int j = 0; j < kMsdaMaxLevels; ++j) cum[j] = tt.cand[j].x;

  // candidate ci -> query row, nominal cell
  auto decode = [&](int ci, int& q, int& ncx, int& ncy) {
    const int j = (ci >= cum[1]) + (ci >= cum[2]) + (ci >= cum[3]);
    const i32x4 cj = tt.cand[j];
    const int rem = ci - cj.x;
    // exact: the fractional part of (rem + 0.5) / qw is at least 0.5 / qw away from an integer
    const int yy = (int)(((float)rem + 0.5f) * tt.inv_qw[j]);
    const int xx = rem - yy * cj.w;
    const i32x4 lj = tt.lv[j];
    q = lj.z + (cj.z + yy) * lj.y + cj.y + xx;
    ncx = tt.tabx[j][xx];
    ncy = tt.taby[j][yy];
  };

  const T* gob = grad_out + (size_t)b * g.Nq * (kHeads * kCh);
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(gob), 0, g.Nq * P::ROWB, 0x00020000);
  float* gvb = grad_value + (size_t)b * g.Nq * (kHeads * kCh);      // Nv == Nq

  // reduce-phase lane roles
  const int part = tid & (LPR - 1);
  const int grp = tid >> P::LPRS;
  const int sub = grp & (R - 1);
  const int cell = grp / R;                           // < NC
  const int hb = h * (kCh * (int)sizeof(T)) + part * 16;
  float acc[P::NACC];
#pragma unroll
  for (int i = 0; i < P::NACC; ++i) acc[i] = 0.f;

  for (int cbase = 0; cbase < ncand; cbase += kPassCand) {
    if (tid == 0) tt.qn = 0u;
    __syncthreads();
    // ---- phase A: which of this candidate's samples can matter?  lane = (candidate, pair of points)
    {
      const int cp = tid >> 1, half = tid & 1;
      const int ci = cbase + cp;
      unsigned hit = 0;
      if (ci < ncand) {
        int q, ncx, ncy;
        decode(ci, q, ncx, ncy);
        const bool home = (ncx >> g.tws) == tx && (ncy >> g.ths) == ty;
        const size_t sb = DSKD_PULL_SB(b, q, h, tl, half * 2);
        const f32x4 l0 = *reinterpret_cast<const f32x4*>(loc + sb * 2);
        const float xs[2] = {l0.x, l0.z}, ys[2] = {l0.y, l0.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const float x = xs[p] * (float)Wl - 0.5f, y = ys[p] * (float)Hl - 0.5f;
          if (x > -1.f && y > -1.f && x < (float)Wl && y < (float)Hl) {
            const int x0 = (int)floorf(x), y0 = (int)floorf(y);
            const bool in_tile = x0 + 1 >= X0 && x0 < X0 + TW && y0 + 1 >= Y0 && y0 < Y0 + TH;
            // both corner columns / rows within M of the nominal cell: every corner's owner tile scans this query
            const bool near = x0 - ncx >= -M && x0 - ncx < M && y0 - ncy >= -M && y0 - ncy < M;
            if ((!RESCAN && in_tile) || (home && !near)) hit |= 1u << p;
          }
        }
      }
      const unsigned pair_hit = hit | (unsigned)__shfl_xor((int)hit, 1);
      const int nh = __popc(hit);
      const int mine = nh | ((half == 0 && pair_hit) ? 1 << 16 : 0);      // hits | candidates with hits
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const int total = __shfl(incl, 63);
      unsigned wbase = 0;
      if (lane == 63 && total) wbase = atomicAdd(&tt.qn, (unsigned)total);
      wbase = (unsigned)__shfl((int)wbase, 63);
      const unsigned excl = wbase + (unsigned)(incl - mine);
      // the pair's slot: the even lane's exclusive candidate count (the odd lane reads it from its partner)
      unsigned slot = excl >> 16;
      const unsigned slot_even = (unsigned)__shfl((int)slot, lane & ~1);
      slot = half ? slot_even : slot;
      if (half == 0 && pair_hit) tt.slot2cand[slot] = (unsigned short)cp;
      unsigned pos = excl & 0xFFFFu;
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if (hit & (1u << p)) tt.queue[pos++] = (slot << 16) | (unsigned)((cp << 2) | (half * 2 + p));
    }
    __syncthreads();
    const unsigned qnp = tt.qn;
    const int qn = (int)(qnp & 0xFFFFu);
    PPROF(1);
    // ---- S: stage the grad_out rows (this head's 64 B) of the candidates with hits
    if constexpr (STAGE && !RESCAN) {
      const int nslots = (int)(qnp >> 16);
      for (int sl = tid >> 2; sl < nslots; sl += NT / 4) {
        int q, ncx, ncy;
        decode(cbase + tt.slot2cand[sl], q, ncx, ncy);
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(grsrc, q * P::ROWB + hb, 0, 0));
        *reinterpret_cast<u32x4*>(s_gst + sl * 64 + part * 16) = v;
      }
    }

    // ---- phase B on slices of the queue: weights, cell slots, strays; then sort and reduce
    for (int e0 = 0; e0 < qn; e0 += kSlice) {
      if constexpr (!RESCAN) {
        for (int i = tid; i <= NC; i += NT) s_cnt[i] = 0;
        __syncthreads();
      }
      int pk[4];
      float wv[4];
      int goff = kOOBg;
#pragma unroll
      for (int c = 0; c < 4; ++c) { pk[c] = -1; wv[c] = 0.f; }
      const int e = e0 + tid;
      if (e < qn) {
        const unsigned ent = tt.queue[e];
        const int p = ent & 3;
        int q, ncx, ncy;
        decode(cbase + (int)((ent & 0xFFFFu) >> 2), q, ncx, ncy);
        const size_t sb = DSKD_PULL_SB(b, q, h, tl, p);
        const f32x2 xy = *reinterpret_cast<const f32x2*>(loc + sb * 2);
        const float a = attn[sb];
        // same arithmetic as the forward: phase A accepted the location (strictly inside (-1, size))
        const float x = xy.x * (float)Wl - 0.5f;
        const float y = xy.y * (float)Hl - 0.5f;
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf;
        const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
        const float w4[4] = {hy * hx * a, hy * lx * a, ly * hx * a, ly * lx * a};
        const bool home = (ncx >> g.tws) == tx && (ncy >> g.ths) == ty;
        goff = STAGE ? (int)(ent >> 16) * 64 : q * P::ROWB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int cx = x0 + (c & 1), cy = y0 + (c >> 1);
          if (cx < 0 || cx > Wl - 1 || cy < 0 || cy > Hl - 1) continue;    // zero padding
          const int tcx = cx - X0, tcy = cy - Y0;
          if ((unsigned)tcx < (unsigned)TW && (unsigned)tcy < (unsigned)TH) {
            if constexpr (!RESCAN) {
              const int cl = (tcy << g.tws) + tcx;
              const int rank = atomicAdd(&s_cnt[cl], 1);
              pk[c] = (cl << 16) | rank;
              wv[c] = w4[c];
            }
          } else if (home) {
            // the corner's owner tile scans this query iff the query's nominal cell is within M of that tile
            const int ox0 = (cx >> g.tws) << g.tws, oy0 = (cy >> g.ths) << g.ths;
            const bool owned = ncx >= ox0 - M && ncx <= ox0 + TW - 1 + M && ncy >= oy0 - M && ncy <= oy0 + TH - 1 + M;
            if (!owned) {
              const int row = stl + cy * Wl + cx;
              if constexpr (!RESCAN) {
                const unsigned idx = atomicAdd(&hdr->count, 1u);
                if (idx < fb_cap) fb[idx] = FbEntry{b * g.Nq + row, b * g.Nq + q, w4[c], h};
                else atomicOr(&hdr->overflow, 1u);
              } else {
                const T* grow = gob + (size_t)q * (kHeads * kCh) + h * kCh;
                float* dst = gvb + (size_t)row * (kHeads * kCh) + h * kCh;
                for (int ch = 0; ch < kCh; ++ch) atomicAdd(dst + ch, w4[c] * (float)grow[ch]);
              }
            }
          }
        }
      }
      if constexpr (!RESCAN) {
        __syncthreads();
        PPROF(2);
        // ---- exclusive prefix sum of the NC counters (wave 0), total in s_cnt[NC]
        if (wave == 0) {
          constexpr int PER = NC >= 64 ? NC / 64 : 1;
          int v[PER], sum = 0;
#pragma unroll
          for (int i = 0; i < PER; ++i) {
            const int idx = lane * PER + i;
            v[i] = idx < NC ? s_cnt[idx] : 0;
            sum += v[i];
          }
          int incl = sum;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
          }
          int run = incl - sum;
#pragma unroll
          for (int i = 0; i < PER; ++i) {
            const int idx = lane * PER + i;
            if (idx < NC) s_cnt[idx] = run;
            run += v[i];
          }
          if (lane == 63) s_cnt[NC] = incl;
        }
        __syncthreads();
        PPROF(3);
        // ---- place the records
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (pk[c] >= 0) s_rec[s_cnt[pk[c] >> 16] + (pk[c] & 0xFFFF)] = i32x2{goff, as_i32f(wv[c])};
        __syncthreads();
        PPROF(4);
        // ---- reduce: this group's share of its cell's list, two records in flight
        {
          auto fetch = [&](int off, float* f) {
            if constexpr (STAGE) {
              const u32x4 v = *reinterpret_cast<const u32x4*>(s_gst + off + part * 16);
              const unsigned a = v.x, bb = v.y, c = v.z, d = v.w;
              f[0] = as_f32u(a << 16); f[1] = as_f32u(a & 0xFFFF0000u);
              f[2] = as_f32u(bb << 16); f[3] = as_f32u(bb & 0xFFFF0000u);
              f[4] = as_f32u(c << 16); f[5] = as_f32u(c & 0xFFFF0000u);
              f[6] = as_f32u(d << 16); f[7] = as_f32u(d & 0xFFFF0000u);
            } else {
              load_g<T>(grsrc, off + hb, f);
            }
          };
          const int end = s_cnt[cell + 1];
          int r = s_cnt[cell] + sub;
          for (; r + R < end; r += 2 * R) {
            const i32x2 r0 = s_rec[r], r1 = s_rec[r + R];
            float g0[P::NACC], g1[P::NACC];
            fetch(r0.x, g0);
            fetch(r1.x, g1);
            const float w0 = as_f32u((unsigned)r0.y), w1 = as_f32u((unsigned)r1.y);
#pragma unroll
            for (int i = 0; i < P::NACC; ++i) acc[i] = fmaf(w0, g0[i], acc[i]);
#pragma unroll
            for (int i = 0; i < P::NACC; ++i) acc[i] = fmaf(w1, g1[i], acc[i]);
          }
          if (r < end) {
            const i32x2 r0 = s_rec[r];
            float g0[P::NACC];
            fetch(r0.x, g0);
            const float w0 = as_f32u((unsigned)r0.y);
#pragma unroll
            for (int i = 0; i < P::NACC; ++i) acc[i] = fmaf(w0, g0[i], acc[i]);
          }
        }
        __syncthreads();     // the next slice resets the counters and overwrites the records
        PPROF(5);
      }
    }
    __syncthreads();         // the next pass overwrites the queue and the staged rows
  }

  if constexpr (!RESCAN) {
    // ---- the R groups of a cell are LPR lanes apart inside one wave
#pragma unroll
    for (int o = LPR; o < LPR * R; o <<= 1) {
#pragma unroll
      for (int i = 0; i < P::NACC; ++i) acc[i] += __shfl_xor(acc[i], o);
    }
    const int tcx = cell & (TW - 1), tcy = cell >> g.tws;
    const int cx = X0 + tcx, cy = Y0 + tcy;
    const bool inimg = cx < Wl && cy < Hl;
    float* dst = gvb + (size_t)(stl + cy * Wl + cx) * (kHeads * kCh) + h * kCh;
    if constexpr (sizeof(T) == 4) {
      if (sub == 0 && inimg) *reinterpret_cast<f32x4*>(dst + part * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    } else {
      // lane `part` holds channels 8*part .. 8*part+7 = the quads 2*part, 2*part+1; regroup so that each store
      // instruction writes 64 contiguous bytes per cell: lane `part` stores quad `part` and quad 4 + part
      const int lbase = lane & ~3;
      float lo[4], hi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a0 = __shfl(acc[i], lbase + (part >> 1)), a1 = __shfl(acc[4 + i], lbase + (part >> 1));
        const float b0 = __shfl(acc[i], lbase + 2 + (part >> 1)), b1 = __shfl(acc[4 + i], lbase + 2 + (part >> 1));
        lo[i] = (part & 1) ? a1 : a0;
        hi[i] = (part & 1) ? b1 : b0;
      }
      if (sub == 0 && inimg) {
        *reinterpret_cast<f32x4*>(dst + part * 4) = f32x4{lo[0], lo[1], lo[2], lo[3]};
        *reinterpret_cast<f32x4*>(dst + 16 + part * 4) = f32x4{hi[0], hi[1], hi[2], hi[3]};
      }
    }
  }
}

// 8 waves per SIMD = two 16-wave workgroups per CU (64 VGPRs, ~75 KB of LDS each)
template <typename T, int NT, int R>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void msda_bwd_pull_kernel(
    const float* __restrict__ loc, const float* __restrict__ attn, const T* __restrict__ grad_out,
    float* __restrict__ grad_value, PullGeom g, FbHeader* hdr, FbEntry* fb, unsigned fb_cap) {
  using P = PT<T>;
  constexpr int NC = NT / ((1 << P::LPRS) * R);
  __shared__ TileTables<NT> tt;
  extern __shared__ int dyn[];
  int* s_cnt = dyn;                                                   // [NC + 1], padded to 16 bytes
  i32x2* s_rec = reinterpret_cast<i32x2*>(dyn + ((NC + 4) & ~3));     // [NT * 4]
  char* s_gst = reinterpret_cast<char*>(s_rec + NT * 4);              // [NT / 2 * 64] (bf16 only)
  pull_tile<T, NT, R, false>(xcd_remap(blockIdx.x, gridDim.x), loc, attn, grad_out, grad_value, g, hdr, fb, fb_cap,
                         tt, s_cnt, s_rec, s_gst);
}

struct PullGeomSet {
  PullGeom g[kMsdaMaxLevels];
  int n;
};

// After the pull launches (kernel boundary: every plain store has landed): add the strays.
template <typename T>
__global__ __launch_bounds__(kMaxThreads) void msda_bwd_pull_apply_kernel(
    const float* __restrict__ loc, const float* __restrict__ attn, const T* __restrict__ grad_out,
    float* __restrict__ grad_value, PullGeomSet gs, FbHeader* hdr, const FbEntry* fb, unsigned fb_cap) {
  __shared__ TileTables<kMaxThreads> tt;
  const unsigned count = hdr->count, overflow = hdr->overflow;
  if (!overflow) {
    const unsigned n = count < fb_cap ? count : fb_cap;
    const int ch = threadIdx.x & 31;
    for (unsigned e = blockIdx.x * (kMaxThreads / 32) + (threadIdx.x >> 5); e < n; e += gridDim.x * (kMaxThreads / 32)) {
      const FbEntry en = fb[e];
      const float gv = (float)grad_out[(size_t)en.gq * (kHeads * kCh) + en.h * kCh + ch];
      atomicAdd(grad_value + (size_t)en.row * (kHeads * kCh) + en.h * kCh + ch, en.w * gv);
    }
  } else if (count != 0) {
    // the list was too small: ignore it, walk every tile again and add the strays directly
    for (int li = 0; li < gs.n; ++li) {
      PullGeom g = gs.g[0];
#pragma unroll
      for (int k = 1; k < kMsdaMaxLevels; ++k)
        if (li == k) g = gs.g[k];
      for (int vb = blockIdx.x; vb < g.nblocks; vb += gridDim.x) {
        pull_tile<T, kMaxThreads, 1, true>(vb, loc, attn, grad_out, grad_value, g, hdr, nullptr, 0u, tt, nullptr, nullptr, nullptr);
        __syncthreads();
      }
    }
  }
  // leave the header zeroed for the next call: the last workgroup to arrive resets it
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned t = atomicAdd(&hdr->ticket, 1u);
    if (t == gridDim.x - 1) {
      hdr->count = 0u;
      hdr->overflow = 0u;
      __threadfence();
      hdr->ticket = 0u;
    }
  }
}

// Tile shape and list sharing per level.  Level 0 (~21 contributions per cell): one group per cell,
// 16 x 16 cells (bf16).  Coarser levels: a cell's list is ~4x longer per level, R = 4 groups per cell.
struct LevelPlan { int R, tws, ths, nt; };

// Threads per workgroup.  A tile is a chain of dependent steps (scan -> stage -> slots -> sort -> reduce, a global
// round trip in three of them, barriers between all): what hides that latency is the NUMBER of workgroups resident on
// a CU, so smaller workgroups (more of them per CU at the same LDS and wave budget) win although their tiles re-scan a
// larger margin.  Measured in round 2: 512 threads 124 us, 256: 135 us, 1 024: 160 us.
inline int pull_threads() { return 512; }

inline LevelPlan plan_level(int level, int dtype) {
  const int lprs = dtype == DSKD_DTYPE_BF16 ? 2 : 3;
  const int kThreads = pull_threads();
  // one group per cell on the fine levels (lists are cut into slices of 2 048 hit samples anyway: ~10 records per
  // cell and slice); R = 4 with 8 x 8 cells where a level is too small to fill the chip with 16 x 16 tiles
  const int R = level <= 1 ? 1 : 4;
  int cells = kThreads >> lprs;
  int rs = 0;
  while ((1 << rs) < R) ++rs;
  cells >>= rs;
  int cs = 0;
  while ((1 << cs) < cells) ++cs;
  LevelPlan p;
  p.R = R;
  p.nt = kThreads;
  p.tws = (cs + 1) / 2;
  p.ths = cs - p.tws;
  return p;
}

inline int pull_margin() { return 5; }      // candidate margin in cells of the target level (the module's initial offsets reach 4)

inline bool make_pull_geom(const MsdaLevels& lg, int level, int dtype, int B, int Nq, int M, PullGeom* g) {
  const LevelPlan p = plan_level(level, dtype);
  for (int l = 0; l < kMsdaMaxLevels; ++l) { g->H[l] = lg.H[l]; g->W[l] = lg.W[l]; g->start[l] = lg.start[l]; }
  g->tl = level; g->tws = p.tws; g->ths = p.ths; g->M = M; g->Nq = Nq;
  g->TX = (lg.W[level] + (1 << p.tws) - 1) >> p.tws;
  g->TY = (lg.H[level] + (1 << p.ths) - 1) >> p.ths;
  g->nblocks = B * g->TY * g->TX * kHeads;
  // candidate extents must fit the nominal-cell tables: a range of n cells covers at most
  // ceil(n * Sj / Sl) + 1 query coordinates of a level of extent Sj
  for (int j = 0; j < kMsdaMaxLevels; ++j) {
    const long long nx = (1 << p.tws) + 2 * M + 1, ny = (1 << p.ths) + 2 * M + 1;
    const long long ex = (nx * lg.W[j] + lg.W[level] - 1) / lg.W[level] + 2;
    const long long ey = (ny * lg.H[j] + lg.H[level] - 1) / lg.H[level] + 2;
    if ((ex < lg.W[j] ? ex : lg.W[j]) > kTabMax || (ey < lg.H[j] ? ey : lg.H[j]) > kTabMax) return false;
  }
  return true;
}

#ifdef DSKD_PULL_COMPACT
__global__ __launch_bounds__(256) void pull_compact_kernel(const float* __restrict__ loc, const float* __restrict__ attn,
                                                           float* __restrict__ cloc, float* __restrict__ cattn, int B, int Nq,
                                                           int tl) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // (b, q, h), h fastest: coalesced reads of 32-B pieces
  if (i >= (long long)B * Nq * kHeads) return;
  const int h = (int)(i % kHeads);
  const long long bq = i / kHeads;
  const int q = (int)(bq % Nq), b = (int)(bq / Nq);
  const size_t src = (size_t)i * kLP + tl * kPts, dst = (((size_t)b * kHeads + h) * Nq + q) * kPts;
  const f32x4 a = *reinterpret_cast<const f32x4*>(loc + src * 2), c = *reinterpret_cast<const f32x4*>(loc + src * 2 + 4);
  *reinterpret_cast<f32x4*>(cloc + dst * 2) = a;
  *reinterpret_cast<f32x4*>(cloc + dst * 2 + 4) = c;
  *reinterpret_cast<f32x4*>(cattn + dst) = *reinterpret_cast<const f32x4*>(attn + src);
}
#endif

template <typename T, int NT, int R>
int launch_pull_level(const float* loc, const float* attn, const T* grad_out, float* grad_value, const PullGeom& g,
                      FbHeader* hdr, FbEntry* fb, unsigned cap, hipStream_t st) {
  using P = PT<T>;
  constexpr int NC = NT / ((1 << P::LPRS) * R);
  constexpr size_t lds = sizeof(int) * ((NC + 4) & ~3) + sizeof(i32x2) * (size_t)NT * 4 +
                         (sizeof(T) == 2 ? (size_t)(NT / 2) * 64 : 0);
  auto kern = msda_bwd_pull_kernel<T, NT, R>;
  int dev = 0;
  static bool done[64] = {};              // the attribute is per device: set it once on each
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!done[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(DSKD_ERR_LAUNCH, "dskd_msda_bwd: cannot reserve LDS for the pull kernel");
    done[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)g.nblocks), dim3(NT), lds, st, loc, attn, grad_out, grad_value, g, hdr,
                     fb, cap);
  return DSKD_OK;
}

template <typename T>
int launch_pull_t(const float* loc, const float* attn, const T* grad_out, float* grad_value, const MsdaLevels& lg,
                  int level_mask, int B, int Nq, int dtype, void* workspace, size_t workspace_bytes, hipStream_t st) {
  FbHeader* hdr = reinterpret_cast<FbHeader*>(workspace);
  FbEntry* fb = reinterpret_cast<FbEntry*>(reinterpret_cast<char*>(workspace) + kPullWsHeader);
  const size_t cap64 = (workspace_bytes - kPullWsHeader) / kPullWsEntry;
  const unsigned cap = (unsigned)(cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : cap64);
  const int M = pull_margin();
  // every level's geometry first: nothing is launched unless all of them fit (the apply kernel must follow every pull
  // launch, or the stray list would be left behind)
  PullGeomSet gs;
  gs.n = 0;
  for (int l = 0; l < kMsdaMaxLevels; ++l) {
    if (!(level_mask & (1 << l))) continue;
    if (!make_pull_geom(lg, l, dtype, B, Nq, M, &gs.g[gs.n]))
      return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_bwd: level %d does not fit the pull kernel's tables", l);
    ++gs.n;
  }
  for (int i = 0; i < gs.n; ++i) {
    const PullGeom& g = gs.g[i];
    int rc;
    const LevelPlan pl = plan_level(g.tl, dtype);
    const float* ploc = loc;
    const float* pattn = attn;
#ifdef DSKD_PULL_COMPACT
    {
      static float* cbuf = nullptr;
      static size_t cbytes = 0;
      const size_t need = (size_t)B * Nq * kHeads * kPts * 3 * sizeof(float);
      if (cbytes < need) {
        if (cbuf) (void)hipFree(cbuf);
        if (hipMalloc(&cbuf, need) != hipSuccess) return fail(DSKD_ERR_LAUNCH, "pull experiment: hipMalloc");
        cbytes = need;
      }
      float* cloc = cbuf;
      float* cattn = cbuf + (size_t)B * Nq * kHeads * kPts * 2;
      const long long n = (long long)B * Nq * kHeads;
      hipLaunchKernelGGL(pull_compact_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, loc, attn, cloc, cattn, B, Nq, g.tl);
      ploc = cloc; pattn = cattn;
    }
#endif
#define DSKD_PULL_LAUNCH(NT_)                                                                                      \
  (pl.R == 1 ? launch_pull_level<T, NT_, 1>(ploc, pattn, grad_out, grad_value, g, hdr, fb, cap, st)                \
             : launch_pull_level<T, NT_, 4>(ploc, pattn, grad_out, grad_value, g, hdr, fb, cap, st))
    if (pl.nt == 1024) rc = DSKD_PULL_LAUNCH(1024);
    else if (pl.nt == 512) rc = DSKD_PULL_LAUNCH(512);
    else rc = DSKD_PULL_LAUNCH(256);
#undef DSKD_PULL_LAUNCH
    if (rc) return rc;
  }
  if (gs.n == 0) return DSKD_OK;
  hipLaunchKernelGGL(msda_bwd_pull_apply_kernel<T>, dim3(256), dim3(kMaxThreads), 0, st, loc, attn, grad_out, grad_value,
                     gs, hdr, fb, cap);
  return DSKD_OK;
}

}  // namespace

bool pull_supported(const MsdaLevels& lg, int levels, int points, int Nv, int Nq, int dtype, int level_mask) {
  if (levels != 4 || points != 4 || Nq != Nv) return false;
  int tot = 0;
  for (int l = 0; l < levels; ++l) {
    if (lg.start[l] != tot) return false;
    if (lg.W[l] > lg.W[0] || lg.H[l] > lg.H[0]) return false;
    tot += lg.H[l] * lg.W[l];
  }
  if (tot != Nq) return false;
  if ((long long)Nq * 1024 >= (long long)kOOBg) return false;
  PullGeom g;
  for (int l = 0; l < levels; ++l)
    if ((level_mask & (1 << l)) && !make_pull_geom(lg, l, dtype, 1, Nq, pull_margin(), &g)) return false;
  return true;
}

#ifdef DSKD_PULL_PROFILE
}  // namespace dskd
extern "C" int dskd_debug_pull_prof(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dskd::g_pprof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(dskd::g_pprof), z, sizeof(z)); }
  return 0;
}
namespace dskd {
#endif

int launch_pull(const float* loc, const float* attn, const void* grad_out, float* grad_value,
                const MsdaLevels& lg, int level_mask, int B, int Nq, int dtype, void* workspace,
                size_t workspace_bytes, hipStream_t st) {
  if (!workspace || workspace_bytes < kPullWsHeader + kPullWsEntry)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_bwd_ws: workspace too small");
  if (dtype == DSKD_DTYPE_F32)
    return launch_pull_t<float>(loc, attn, (const float*)grad_out, grad_value, lg, level_mask, B, Nq, dtype, workspace,
                                workspace_bytes, st);
  return launch_pull_t<__bf16>(loc, attn, (const __bf16*)grad_out, grad_value, lg, level_mask, B, Nq, dtype, workspace,
                               workspace_bytes, st);
}

}  // namespace dskd
