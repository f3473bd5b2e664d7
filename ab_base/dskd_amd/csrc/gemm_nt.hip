// Y[M, N] = act(X[M, K] W[N, K]^T + bias[N] (+ R[M, N])) in bf16 with f32 accumulation on the matrix cores of gfx950 --
// the 1x1 convolutions of the ResNet trunk and of the ChannelMapper on channels_last activations, where the activation
// IS the [B*H*W, Cin] matrix and the folded convolution weight [Cout, Cin, 1, 1] IS W.
//
// Replaces, for every Bottleneck (mmdet/models/backbones/resnet.py:271-303: conv1 -> bn1 -> relu, conv3 -> bn3 ->
// (+ identity) -> relu, downsample conv -> bn) and for ChannelMapper's lateral convolutions
// (mmdet/models/necks/channel_mapper.py:90-100), the library convolution + the separate bias / residual / ReLU pass
// (csrc/biasact.hip): the folded-BN shift, the residual and the ReLU are applied to the f32 accumulators and the
// output is written once.  The same kernel with the transposed weight is the input-gradient GEMM dX = dY W.
//
// Why hand-written: these GEMMs are short in K and N (64 .. 2048) and very tall (M = 4 200 .. 267 200 at B = 4,
// 800 x 1333), i.e. mostly memory-bound streaming of the activation; the library's implicit-GEMM kernels run them at
// ~270 TFLOP/s with a second pass over every output for the epilogue (profiles/r02_step_breakdown.txt).
//
// Structure (cdna_hip_programming.md section 5):
//  * workgroup = 4 waves, tile 64 tokens x 128 outputs (128 x 64 when N is not a multiple of 128); every wave owns 64
//    outputs x 32 tokens = 2 accumulator tiles of v_mfma_f32_32x32x16_bf16.  The WEIGHT is the A operand and the
//    activation the B operand, so a lane ends up with ONE token and -- rows of the A tile permuted by pi (as in
//    ffn_mfma.hip) -- 16 CONSECUTIVE output channels per accumulator tile: 32 contiguous bytes per lane, bias / residual
//    / ReLU in registers, two 16-byte stores.
//  * both operands are K-contiguous (16 bytes = the 8 k of one lane's fragment), so they go global -> LDS with 16-byte
//    LDS-DMA (global_load_lds), no register staging and no pack kernel.  K is walked in stages of 64: two panels of
//    [rows][32 k] = 64-byte rows per operand; the 16-byte chunk c of row r sits at chunk position c ^ ((r >> 2) & 3) --
//    applied on the SOURCE address, the LDS image of one DMA instruction stays lane-linear (16 rows x 64 B) -- which
//    makes every ds_read_b128 fragment read conflict-free (the four 16-lane groups of that instruction cover rows
//    {0-3, 12-15, 20-27}, ...: four different (r >> 2) & 3 for every r & 3; pi permutes rows inside those groups).
//  * two LDS stages (48 | 64 KB per workgroup -> 3 | 2 workgroups per CU cover each other's waits); the DMA of stage
//    k + 1 is issued before stage k is consumed and retired with a COUNTED s_waitcnt vmcnt; raw s_barrier.
//  * a strided 1x1 convolution (the downsample branch: stride 2) only changes the row address of the activation.
//  * blockIdx -> tile: XCD-contiguous (xcd_remap), output tiles of one token tile adjacent, so the activation tile is
//    fetched from HBM once and re-read from that XCD's L2.
#include "common.h"

namespace dskd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int pi_row(int r) { return (r & 3) + 4 * (r >> 3) + 16 * ((r >> 2) & 1); }

struct GemmArgs {
  const __bf16* x;       // activation rows (see the row map below)
  const __bf16* w;       // [N, K]
  const __bf16* bias;    // [N] or null
  const __bf16* res;     // [M, N] or null: added before the activation
  const __bf16* gate;    // [M, N] or null: outputs where gate <= 0 are zeroed (the ReLU mask of a backward pass)
  __bf16* y;             // [M, N]
  long long M;
  int N, K, relu;
  // row map: output row m = (img, ho, wo) reads the activation row ((img * Hi + s * ho) * Wi + s * wo); s == 0: row m
  int s, HoWo, Wo, Hi, Wi;
  // 3 x 3 convolution (padding 1): K = 9 * C with k = tap * C + c (the channels_last weight [N][ky][kx][C] as it lies);
  // tap (ky, kx) reads the pixel (s ho + ky - 1, s wo + kx - 1), rows outside the image come from a page of zeros
  int C, cshift;         // channels per tap, log2(C / 64)
};

__device__ __attribute__((aligned(256))) char g_zero_page[256];

#ifdef DSKD_GEMM_PROFILE
// timing-only diagnostic build (scratch/r04_gemm_prof.py): wave 0 of every workgroup stamps its phases into a buffer of its
// own (8 words per workgroup) that nothing else reads
__device__ long long* g_gemm_prof;
__device__ __forceinline__ void prof_stamp(int slot) {
  if (g_gemm_prof && threadIdx.x == 0) {
    long long* p = g_gemm_prof + (long long)blockIdx.x * 8;
    p[slot] = (long long)__builtin_amdgcn_s_memtime();
    if (slot == 0) {
      p[6] = (long long)__builtin_amdgcn_s_memrealtime();
      unsigned id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      p[5] = ((long long)xcc << 32) | id;
    }
    if (slot == 4) p[7] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}
#define PROF(slot) prof_stamp(slot)
#else
#define PROF(slot)
#endif

__device__ __forceinline__ unsigned lds_offset(const void* p) {
  return (unsigned)(unsigned long)((const __attribute__((address_space(3))) char*)p);
}
__device__ __forceinline__ bf16x8 frag_read(unsigned lds_addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr));
  return v;
}

// Fragment reads of one K stage (64): 16 ds_read_b128, written as asm (hipcc would sink each read down to its use:
// ds_read -> s_waitcnt 0 -> MFMA), issued in k-step order so that counted waits can release the MFMAs step by step.
template <int MT>
struct Frags {
  bf16x8 w[4][2], x[4][MT];
};
template <int MT, int PX, int PW>
__device__ __forceinline__ void read_stage(Frags<MT>& f, const unsigned (&xa)[MT][2], const unsigned (&wa)[2][2], unsigned so) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) f.w[ks][nt] = frag_read(wa[nt][ks & 1] + so + (ks >> 1) * PW);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) f.x[ks][mt] = frag_read(xa[mt][ks & 1] + so + (ks >> 1) * PX);
  }
}
// the 16 MFMAs of a stage; COUNTED: the fragments were read just before (LDS returns in order: before k-step ks all
// but the (3 - ks) * (2 + MT) youngest reads are done); otherwise they are already complete
template <int MT, bool COUNTED>
__device__ __forceinline__ void mfma_stage(f32x16 (&acc)[MT][2], Frags<MT>& f) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if constexpr (COUNTED && MT == 2) {
      if (ks == 0) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.x[0][0]), "+v"(f.x[0][1]));
      if (ks == 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f.w[1][0]), "+v"(f.w[1][1]), "+v"(f.x[1][0]), "+v"(f.x[1][1]));
      if (ks == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f.w[2][0]), "+v"(f.w[2][1]), "+v"(f.x[2][0]), "+v"(f.x[2][1]));
      if (ks == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.w[3][0]), "+v"(f.w[3][1]), "+v"(f.x[3][0]), "+v"(f.x[3][1]));
    } else if constexpr (COUNTED) {
      if (ks == 0) asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.x[0][0]));
      if (ks == 1) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f.w[1][0]), "+v"(f.w[1][1]), "+v"(f.x[1][0]));
      if (ks == 2) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f.w[2][0]), "+v"(f.w[2][1]), "+v"(f.x[2][0]));
      if (ks == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.w[3][0]), "+v"(f.w[3][1]), "+v"(f.x[3][0]));
    }
    // r4: without this fence hipcc moves the later waits up in front of the first MFMA (the asm statements only keep their
    // order among themselves): lgkmcnt(12), (8), (4) back to back, i.e. every fragment read of the stage has to land before
    // any MFMA issues (seen in the ISA of the r3 build)
    if constexpr (COUNTED) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[ks][nt], f.x[ks][mt], acc[mt][nt], 0, 0, 0);
    if constexpr (COUNTED) __builtin_amdgcn_sched_barrier(0);
  }
}
struct FragK {
  bf16x8 w[2], x[2];
};
// One k-step (16) of a 64 x 64 wave tile: the MFMAs only (the caller has waited for / waits for the fragments);
// KS < 3: behind a counted wait that assumes the stage's 16 fragment reads are the wave's youngest LDS operations.
template <int KS>
__device__ __forceinline__ void mfma_ks(f32x16 (&acc)[2][2], Frags<2>& f) {
  if constexpr (KS == 0) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.x[0][0]), "+v"(f.x[0][1]));
  if constexpr (KS == 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f.w[1][0]), "+v"(f.w[1][1]), "+v"(f.x[1][0]), "+v"(f.x[1][1]));
  if constexpr (KS == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f.w[2][0]), "+v"(f.w[2][1]), "+v"(f.x[2][0]), "+v"(f.x[2][1]));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[KS][nt], f.x[KS][mt], acc[mt][nt], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}
// Epilogue: lane = one token (m_first + 32 mt), 16 consecutive outputs (n_first + 32 nt ..) per accumulator tile.
template <int MT>
__device__ __forceinline__ void store_tile(const GemmArgs& a, const f32x16 (&acc)[MT][2], long long m_first, int n_first) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long long m = m_first + mt * 32;
    if (m >= a.M) continue;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = n_first + nt * 32;
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = acc[mt][nt][i];
      if (a.bias) {
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(a.bias + n), b1 = *reinterpret_cast<const bf16x8*>(a.bias + n + 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] += (float)b0[i]; v[8 + i] += (float)b1[i]; }
      }
      if (a.res) {
        const __bf16* rp = a.res + m * a.N + n;
        const bf16x8 r0 = *reinterpret_cast<const bf16x8*>(rp), r1 = *reinterpret_cast<const bf16x8*>(rp + 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] += (float)r0[i]; v[8 + i] += (float)r1[i]; }
      }
      if (a.gate) {
        const __bf16* gp = a.gate + m * a.N + n;
        const bf16x8 g0 = *reinterpret_cast<const bf16x8*>(gp), g1 = *reinterpret_cast<const bf16x8*>(gp + 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[i] = (float)g0[i] > 0.f ? v[i] : 0.f;
          v[8 + i] = (float)g1[i] > 0.f ? v[8 + i] : 0.f;
        }
      }
      bf16x8 o0, o1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float p = v[i], q = v[8 + i];
        if (a.relu) { p = fmaxf(p, 0.f); q = fmaxf(q, 0.f); }
        o0[i] = (__bf16)p;
        o1[i] = (__bf16)q;
      }
      __bf16* yp = a.y + m * a.N + n;
      *reinterpret_cast<bf16x8*>(yp) = o0;
      *reinterpret_cast<bf16x8*>(yp + 8) = o1;
    }
  }
}

// ---- epilogue through LDS (r4): whole 128-byte lines instead of 32 rows x 16 B per store instruction
__device__ __forceinline__ void lds_write4(unsigned addr, const f32x16& v, int q) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}) : "memory");
}

// rows [row0, row0 + 32) x 64 outputs of one wave: acc[nt][i] = token (lane & 31), output 32 nt + 16 (lane >> 5) + i.
// `scr`: this wave's 32 x 272-byte LDS scratch.  PARTIAL: f32 to `dst_f32` (row stride ldp floats), no epilogue.
// residual / gate of the 32 x 64 block a wave stores, in the layout store_rows32 consumes them (lane = 16 B of one row):
// requested BEFORE the K loop so that their latency runs under the loop instead of after it -- the tall thin layers of
// ResNet stage 1-2 (K = 64 .. 256: one to four stages) are HBM-bound and spent a third of a tile's time in the epilogue
// waiting for exactly these reads.
#ifndef DSKD_EPI_PRE
#define DSKD_EPI_PRE 1       // -DDSKD_EPI_PRE=0: the A/B build that reads them in the epilogue
#endif
struct EpiPre {
  bf16x8 res[4], gate[4];
};
__device__ __forceinline__ void load_epi(const GemmArgs& a, long long m_first, int n_first, int lane, EpiPre& p) {
  const int row = lane >> 3, n = n_first + (lane & 7) * 8;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    long long m = m_first + it * 8 + row;
    if (m >= a.M) m = a.M - 1;
    p.res[it] = bf16x8{};
    p.gate[it] = bf16x8{};
    if (a.res) p.res[it] = *reinterpret_cast<const bf16x8*>(a.res + m * a.N + n);
    if (a.gate) p.gate[it] = *reinterpret_cast<const bf16x8*>(a.gate + m * a.N + n);
  }
}

template <bool PARTIAL, bool PRE = false>
__device__ __forceinline__ void store_rows32(const GemmArgs& a, const f32x16 (&acc)[2], unsigned scr, long long m_first,
                                             int n_first, float* dst_f32, int ldp, int lane, const EpiPre* pre = nullptr) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) lds_write4(scr + r * 272 + (nt * 32 + 16 * h + 4 * q) * 4, acc[nt], q);
  const int row = lane >> 3, ch = lane & 7;
  const int n = n_first + ch * 8;
  float bias[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bias[i] = 0.f;
  if (!PARTIAL && a.bias) {
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(a.bias + n);
#pragma unroll
    for (int i = 0; i < 8; ++i) bias[i] = (float)b[i];
  }
  f32x4 v0[4], v1[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const unsigned ad = scr + (it * 8 + row) * 272 + ch * 32;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v0[it]) : "v"(ad) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(v1[it]) : "v"(ad) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0[0]), "+v"(v0[1]), "+v"(v0[2]), "+v"(v0[3]), "+v"(v1[0]), "+v"(v1[1]), "+v"(v1[2]),
               "+v"(v1[3])::"memory");
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long long m = m_first + it * 8 + row;
    if (m >= a.M) continue;
    float v[8] = {v0[it].x, v0[it].y, v0[it].z, v0[it].w, v1[it].x, v1[it].y, v1[it].z, v1[it].w};
    if constexpr (PARTIAL) {
      float* dp = dst_f32 + (long long)(it * 8 + row) * ldp + ch * 8;
      *reinterpret_cast<f32x4*>(dp) = v0[it];
      *reinterpret_cast<f32x4*>(dp + 4) = v1[it];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += bias[i];
      if (a.res) {
        bf16x8 rr;
        if constexpr (PRE) rr = pre->res[it];
        else rr = *reinterpret_cast<const bf16x8*>(a.res + m * a.N + n);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += (float)rr[i];
      }
      if (a.gate) {
        bf16x8 gg;
        if constexpr (PRE) gg = pre->gate[it];
        else gg = *reinterpret_cast<const bf16x8*>(a.gate + m * a.N + n);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)gg[i] > 0.f ? v[i] : 0.f;
      }
      bf16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (__bf16)(a.relu ? fmaxf(v[i], 0.f) : v[i]);
      *reinterpret_cast<bf16x8*>(a.y + m * a.N + n) = o;
    }
  }
}

// Two LDS stages, the DMA of stage k + 1 issued before stage k is consumed; 48 | 64 KB per workgroup, so two or three
// workgroups share a CU and cover each other's waits.  Measured and NOT kept (scratch/r03_conv1x1.py, every variant
// green on the parity tests, each slower on all 19 layer shapes of the trunk): four stages with the DMA three ahead;
// a fifth, DMA-only wave with three stages; four stages + register double-buffered fragments + one barrier per stage;
// a per-tile rotation of the K order.  What they have in common is one workgroup per CU (96 - 128 KB of LDS): the
// layers where a deeper pipeline should pay (K >= 512) have 4 200 .. 16 800 tokens, i.e. 264 .. 1 056 tiles -- about
// one round of the chip -- and there tile quantisation (264 = 256 + 8) and the second resident workgroup matter more
// than the per-stage latency.  SQ counters of the K = 1024, N = 256, 16 800-token layer: no LDS bank conflicts, MFMA
// pipe busy 24 % of the wave's lifetime, 30 % in s_waitcnt / barrier.
template <int BN, int MT, bool CONV3, bool LDS_EPI = false, bool EPI_PRE = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmArgs a) {
  constexpr int NS = 2;
  constexpr int WN = BN / 64;              // waves along the outputs
  constexpr int WM = 4 / WN;               // waves along the tokens
  constexpr int BM = WM * MT * 32;         // 128, or 64 for <128, 1>
  constexpr int XRB = BM / 64;             // 16-row activation blocks whose DMA this wave issues
  constexpr int PX = BM * 64, PW = BN * 64;          // bytes of one 32-k panel
  constexpr int STAGE = 2 * (PX + PW);
  constexpr int WRB = BN / 64;             // 16-row weight blocks whose DMA this wave issues (activation: always 2)
  constexpr int LOADS = 2 * (XRB + WRB);   // LDS-DMA instructions per wave and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];      // NS stages

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;

  const int tiles_n = a.N / BN;
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = vb % tiles_n;
  const long long tm = vb / tiles_n;
  const long long m0 = tm * BM;
  const int n0 = tn * BN;
  const int nk = a.K >> 6;

  // ---- per-lane source pointers of the LDS-DMA: lane i of an instruction fills row (i >> 2), chunk position (i & 3)
  const int lr = lane >> 2;
  const int csw = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;      // logical chunk held at that position (bytes)
  const char* xp[XRB];
  const char* wp[WRB];
  unsigned vmask[XRB];                                          // CONV3: taps whose pixel lies inside the image
#pragma unroll
  for (int j = 0; j < XRB; ++j) {
    long long m = m0 + (wave * XRB + j) * 16 + lr;
    if (m >= a.M) m = a.M - 1;                                // rows past the end: any valid row, never stored
    long long row = m;
    vmask[j] = 0x1FFu;
    if (a.s) {
      const long long img = m / a.HoWo;
      const int rem = (int)(m - img * a.HoWo);
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      const int hi = a.s * ho, wi = a.s * wo;
      row = (img * a.Hi + hi) * a.Wi + wi;
      if constexpr (CONV3) {
        const unsigned rowm = (hi > 0 ? 1u : 0u) | 2u | (hi + 1 < a.Hi ? 4u : 0u);      // ky = 0, 1, 2
        const unsigned colm = (wi > 0 ? 1u : 0u) | 2u | (wi + 1 < a.Wi ? 4u : 0u);      // kx = 0, 1, 2
        vmask[j] = ((rowm & 1u) ? colm : 0u) | ((rowm & 2u) ? colm << 3 : 0u) | ((rowm & 4u) ? colm << 6 : 0u);
      }
    }
    xp[j] = reinterpret_cast<const char*>(a.x) + row * (CONV3 ? a.C : a.K) * 2 + csw;
  }
  const char* const zp = g_zero_page + csw;
#pragma unroll
  for (int j = 0; j < WRB; ++j)
    wp[j] = reinterpret_cast<const char*>(a.w) + (long long)(n0 + (wave * WRB + j) * 16 + lr) * a.K * 2 + csw;

  auto issue = [&](int kt) {
    char* sx = smem + (kt % NS) * STAGE;
    char* sw = sx + 2 * PX;
    const int kb = kt * 128;                                  // bytes along K
    int tap = 0, xoff = kb;
    if constexpr (CONV3) {                                    // stage kt = 64 channels of ONE tap (C is a multiple of 64)
      tap = kt >> a.cshift;
      const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;       // tap / 3 for tap < 9
      xoff = ((ky - 1) * a.Wi + (kx - 1)) * a.C * 2 + (kt - (tap << a.cshift)) * 128;
    }
#pragma unroll
    for (int j = 0; j < XRB; ++j) {
      const char* src = xp[j] + xoff;
      if constexpr (CONV3) src = ((vmask[j] >> tap) & 1u) ? src : zp;
#pragma unroll
      for (int p = 0; p < 2; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 64),
                                         (__attribute__((address_space(3))) void*)(sx + p * PX + (wave * XRB + j) * 1024),
                                         16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < WRB; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp[j] + kb + p * 64),
                                         (__attribute__((address_space(3))) void*)(sw + p * PW + (wave * WRB + j) * 1024),
                                         16, 0, 0);
  };

  // ---- fragment addresses inside a stage (k-step ks of 4: panel ks >> 1, logical chunk 2 (ks & 1) + h)
  const unsigned base = lds_offset(smem);
  unsigned xa[MT][2], wa[2][2];           // [tile][ks & 1]; the panel adds PX / PW
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int c = 2 * e + h;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = (wm * MT + mt) * 32 + r;
      xa[mt][e] = base + row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int row = (wn * 2 + nt) * 32 + pi_row(r);
      wa[nt][e] = base + 2 * PX + row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
    }
  }
  f32x16 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  constexpr bool PRE = LDS_EPI && EPI_PRE;
  EpiPre pre[PRE ? MT : 1];
  {
    PROF(0);
    issue(0);
    if constexpr (PRE) {       // older than every later DMA: the loop's first counted wait covers them too
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) load_epi(a, m0 + (wm * MT + mt) * 32, n0 + wn * 64, lane, pre[mt]);
    }
    Frags<MT> f;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) {
        issue(kt + 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");      // stage kt has landed (mine); kt + 1 in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                                       // ... and everybody's
      if (kt == 0) PROF(1);
      read_stage<MT, PX, PW>(f, xa, wa, (kt % NS) * STAGE);
      mfma_stage<MT, true>(acc, f);
      __builtin_amdgcn_s_barrier();       // every wave has read this stage before the next DMA overwrites it
    }
  }
  PROF(2);
  if constexpr (LDS_EPI) {       // the loop's last barrier is behind every wave's fragment reads: the stage buffers are free
    const unsigned scr = base + wave * (32 * 272);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      store_rows32<false, PRE>(a, acc[mt], scr, m0 + (wm * MT + mt) * 32, n0 + wn * 64, nullptr, 0, lane, &pre[PRE ? mt : 0]);
  } else {
    store_tile<MT>(a, acc, m0 + wm * MT * 32 + r, n0 + wn * 64 + 16 * h);
  }
#ifdef DSKD_GEMM_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PROF(4);
#endif
}

template <int BN, int MT, bool CONV3, bool LDS_EPI = false, bool EPI_PRE = false>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
  constexpr int BM = (4 / (BN / 64)) * MT * 32;
  constexpr int LDS = 2 * 2 * (BM * 64 + BN * 64);
  static_assert(LDS >= 4 * 32 * 272, "epilogue scratch");
  auto kern = gemm_nt_kernel<BN, MT, CONV3, LDS_EPI, EPI_PRE>;
  int dev = 0;
  static bool done[64] = {};              // the attribute is per device: set it once on each
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!done[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return fail(DSKD_ERR_LAUNCH, "dskd_gemm_nt: cannot reserve %d bytes of LDS", LDS);
    done[dev] = true;
  }
  const long long tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
  if (tiles > 0x7FFFFFFFll) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_nt: too many tiles");
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), LDS, st, a);
  return check_launch("dskd_gemm_nt");
}

// ------------------------------------------------------------------------------------------------------------------
// The MFMA-bound shapes (K >= 256: ResNet stages 3-4, the ChannelMapper, every 3x3 convolution): big tiles.
//
// What bounds gemm_nt_kernel there (profiles/r03_conv1x1_microbench.txt, r04_gemm_big_microbench.txt): a 64 x 128 tile
// pulls (64 + 128) rows x 128 B = 24 KB through LDS-DMA per K stage for 1.05 MFLOP; at 510-580 TFLOP/s that is 12-15 TB/s of
// L2 -> LDS fill, i.e. the chip's gather-into-LDS rate (MI355X_MICROARCH.md "Indexed rows": 16.8-18.8 TB/s), not the
// matrix pipe (busy 24 %).  Fill bytes per FLOP fall with the tile: 256 x 128 needs 11.7 KB / MFLOP, 256 x 256 7.8.
//   * workgroup = WM x WN waves, every wave 64 tokens x 64 outputs (2 x 2 accumulator tiles, 1 KB of fragment reads per
//     MFMA), tile 64 WM x 64 WN; NS LDS stages of K = 64 with the LDS-DMA NS - 1 stages ahead and ONE barrier per stage
//     (the barrier that publishes stage k also retires the reads of stage k - 1, whose buffer the next DMA overwrites);
//   * a big tile means few tiles (132 .. 1 056 for 256 CUs): the last, partial round of the grid would idle most of the
//     chip.  The host cuts the tile list at a multiple of the resident workgroups: the first `full` tiles are whole
//     workgroups, each remaining tile is split along K over `splits` workgroups that store f32 partial tiles into the
//     caller's scratch; gemm_fixup_kernel sums them and applies the epilogue (no inter-workgroup hand-off inside a
//     launch, no atomics: deterministic);
//   * epilogue through LDS: the accumulators (lane = token, 16 channels) are turned in a per-wave f32 scratch so that
//     8 lanes cover the 128 contiguous bytes of one token's 64 outputs -- residual / gate are read and the result is
//     stored as whole 128-B lines (gemm_nt_kernel's lanes touch 32 rows x 16 B per instruction).
struct BigPlan {
  int full;          // work items [0, full): whole tiles (a multiple of 8, or all)
  int splits;        // each tile >= full is cut into `splits` K ranges (1: stored directly)
  float* planes;     // [tiles - full][splits][BM * BN] f32 partial tiles (splits > 1)
};

template <int WM, int WN, int NS, bool CONV3>
__global__ __launch_bounds__(WM * WN * 64) void gemm_big_kernel(const GemmArgs a, const BigPlan p) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int XRB = 4 / WN, WRB = 4 / WM;          // 16-row blocks per wave and operand
  static_assert(XRB * WN == 4 && WRB * WM == 4, "WM, WN in {1, 2, 4}");
  constexpr int PX = BM * 64, PW = BN * 64;          // bytes of one 32-k panel
  constexpr int STAGE = 2 * (PX + PW);
  constexpr int LOADS = 2 * (XRB + WRB);             // LDS-DMA instructions per wave and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;
  const int tiles_n = a.N / BN;
  const int nk = a.K >> 6;

  int tile, ks0 = 0, ks1 = nk, sp = 0;
  const int b = blockIdx.x;
  const bool partial = p.splits > 1 && b >= p.full;
  if (b < p.full) {
    tile = xcd_remap(b, p.full);
  } else {
    const int rb = b - p.full;
    tile = p.full + rb / p.splits;
    sp = rb - (rb / p.splits) * p.splits;
    ks0 = (int)((long long)sp * nk / p.splits);
    ks1 = (int)((long long)(sp + 1) * nk / p.splits);
  }
  const int tn = tile % tiles_n;
  const long long tm = tile / tiles_n;
  const long long m0 = tm * BM;
  const int n0 = tn * BN;

  // ---- per-lane source pointers of the LDS-DMA: lane i of an instruction fills row (i >> 2), chunk position (i & 3)
  const int lr = lane >> 2;
  const int csw = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
  const char* xp[XRB];
  const char* wp[WRB];
  unsigned vmask[XRB];
#pragma unroll
  for (int j = 0; j < XRB; ++j) {
    long long m = m0 + (wave * XRB + j) * 16 + lr;
    if (m >= a.M) m = a.M - 1;
    long long row = m;
    vmask[j] = 0x1FFu;
    if (a.s) {
      const long long img = m / a.HoWo;
      const int rem = (int)(m - img * a.HoWo);
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      const int hi = a.s * ho, wi = a.s * wo;
      row = (img * a.Hi + hi) * a.Wi + wi;
      if constexpr (CONV3) {
        const unsigned rowm = (hi > 0 ? 1u : 0u) | 2u | (hi + 1 < a.Hi ? 4u : 0u);
        const unsigned colm = (wi > 0 ? 1u : 0u) | 2u | (wi + 1 < a.Wi ? 4u : 0u);
        vmask[j] = ((rowm & 1u) ? colm : 0u) | ((rowm & 2u) ? colm << 3 : 0u) | ((rowm & 4u) ? colm << 6 : 0u);
      }
    }
    xp[j] = reinterpret_cast<const char*>(a.x) + row * (CONV3 ? a.C : a.K) * 2 + csw;
  }
  const char* const zp = g_zero_page + csw;
#pragma unroll
  for (int j = 0; j < WRB; ++j)
    wp[j] = reinterpret_cast<const char*>(a.w) + (long long)(n0 + (wave * WRB + j) * 16 + lr) * a.K * 2 + csw;

  // LDS-DMA instructions [lo, hi) of the LOADS that bring in K stage kt (the bounds are constants after inlining)
  auto issue = [&](int kt, int lo, int hi) {
    char* sx = smem + ((kt - ks0) % NS) * STAGE;
    char* sw = sx + 2 * PX;
    const int kb = kt * 128;
    int tap = 0, xoff = kb;
    if constexpr (CONV3) {
      tap = kt >> a.cshift;
      const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
      xoff = ((ky - 1) * a.Wi + (kx - 1)) * a.C * 2 + (kt - (tap << a.cshift)) * 128;
    }
#pragma unroll
    for (int j = 0; j < XRB; ++j) {
      const char* src = xp[j] + xoff;
      if constexpr (CONV3) src = ((vmask[j] >> tap) & 1u) ? src : zp;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (2 * j + q >= lo && 2 * j + q < hi)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 64),
                                           (__attribute__((address_space(3))) void*)(sx + q * PX + (wave * XRB + j) * 1024),
                                           16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < WRB; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (2 * XRB + 2 * j + q >= lo && 2 * XRB + 2 * j + q < hi)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp[j] + kb + q * 64),
                                           (__attribute__((address_space(3))) void*)(sw + q * PW + (wave * WRB + j) * 1024),
                                           16, 0, 0);
  };
  // all but my `n` youngest K stages have landed
  auto wait_stages = [&](int n) {
    if (n >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LOADS < 63 ? 3 * LOADS : 63) : "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  const unsigned base = lds_offset(smem);
  unsigned xa[2][2], wa[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int c = 2 * e + h;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = (wm * 2 + mt) * 32 + r;
      xa[mt][e] = base + row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int row = (wn * 2 + nt) * 32 + pi_row(r);
      wa[nt][e] = base + 2 * PX + row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
    }
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  // ---- K loop.  The fragments of stage k + 1 are requested (second register set) BEFORE the last MFMA group of stage k,
  // right behind the one barrier of the stage, and the LDS-DMA instructions of stage k + NS go out in three parts between
  // the MFMA groups: first cut had barrier -> 6 DMA issues -> 16 fragment reads -> MFMAs in a row with both waves of a SIMD
  // in the same phase, 2 200 - 2 400 cycles per stage for 1 024 cycles of MFMA (profiles/r04_gemm_big_phases.txt).
  constexpr int P1 = LOADS / 3, P2 = 2 * LOADS / 3;
  PROF(0);
#pragma unroll
  for (int i = 0; i < NS; ++i)
    if (ks0 + i < ks1) issue(ks0 + i, 0, LOADS);
  // fragment registers: k-steps 0 .. 2 of the stage in f (re-used by the next stage: its reads are requested when those
  // MFMAs have been issued), k-step 3 alternates between t0 / t1 (still needed while the next stage's reads land)
  Frags<2> f;
  FragK t0, t1;
  auto read_next = [&](FragK& t, unsigned so) {
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) f.w[ks][nt] = frag_read(wa[nt][ks & 1] + so + (ks >> 1) * PW);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) f.x[ks][mt] = frag_read(xa[mt][ks & 1] + so + (ks >> 1) * PX);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) t.w[nt] = frag_read(wa[nt][1] + so + PW);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) t.x[mt] = frag_read(xa[mt][1] + so + PX);
  };
  if (ks0 < ks1) {
    wait_stages(min(NS - 1, ks1 - 1 - ks0));
    __builtin_amdgcn_s_barrier();
    PROF(1);
    read_next(t0, 0);
  }
  auto body = [&](FragK& cur, FragK& nxt, int kt) {
    const bool tail_dma = kt > ks0 && kt - 1 + NS < ks1;      // stage kt - 1 + NS: first part issued in the previous body
    mfma_ks<0>(acc, f);
    if (tail_dma) issue(kt - 1 + NS, P1, P2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_ks<1>(acc, f);
    if (tail_dma) issue(kt - 1 + NS, P2, LOADS);
    __builtin_amdgcn_sched_barrier(0);
    mfma_ks<2>(acc, f);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur.w[0]), "+v"(cur.w[1]), "+v"(cur.x[0]), "+v"(cur.x[1])::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < ks1) {
      wait_stages(min(NS - 2, ks1 - 2 - kt));          // my part of stage kt + 1 has landed
      __builtin_amdgcn_s_barrier();                    // everybody's has; everybody has read stage kt
      read_next(nxt, ((kt + 1 - ks0) % NS) * STAGE);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.w[nt], cur.x[mt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NS < ks1) issue(kt + NS, 0, P1);          // ... into the buffer of stage kt (every wave is past the barrier)
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int kt = ks0; kt < ks1; kt += 2) {
    body(t0, t1, kt);
    if (kt + 1 < ks1) body(t1, t0, kt + 1);
  }
  PROF(2);
  __builtin_amdgcn_s_barrier();           // the stage buffers become the epilogue's scratch
  PROF(3);
  const unsigned scr = base + wave * (32 * 272);
  const long long m_w = m0 + wm * 64;
  const int n_w = n0 + wn * 64;
  if (partial) {
    float* pl = p.planes + ((long long)(tile - p.full) * p.splits + sp) * (BM * BN) + (long long)(wm * 64) * BN + wn * 64;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) store_rows32<true>(a, acc[mt], scr, m_w + mt * 32, n_w, pl + (long long)mt * 32 * BN, BN, lane);
  } else {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) store_rows32<false>(a, acc[mt], scr, m_w + mt * 32, n_w, nullptr, 0, lane);
  }
#ifdef DSKD_GEMM_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PROF(4);
#endif
}

// Sum of the `splits` partial tiles of every split tile + the epilogue; one thread = 8 consecutive outputs of one token.
__global__ __launch_bounds__(256) void gemm_fixup_kernel(const GemmArgs a, const BigPlan p, int BM, int BN) {
  const int per_row = BN >> 3;
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= BM * per_row) return;
  const int row = e / per_row, c8 = e - row * per_row;
  const int tile = p.full + blockIdx.x;
  const int tiles_n = a.N / BN;
  const int tn = tile % tiles_n;
  const long long m = (long long)(tile / tiles_n) * BM + row;
  if (m >= a.M) return;
  const int n = tn * BN + c8 * 8;
  const float* pl = p.planes + (long long)blockIdx.x * p.splits * (BM * BN) + (long long)row * BN + c8 * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.splits; ++s) {
    const f32x4 u0 = *reinterpret_cast<const f32x4*>(pl + (long long)s * BM * BN);
    const f32x4 u1 = *reinterpret_cast<const f32x4*>(pl + (long long)s * BM * BN + 4);
    v[0] += u0.x; v[1] += u0.y; v[2] += u0.z; v[3] += u0.w;
    v[4] += u1.x; v[5] += u1.y; v[6] += u1.z; v[7] += u1.w;
  }
  if (a.bias) {
    const bf16x8 bb = *reinterpret_cast<const bf16x8*>(a.bias + n);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += (float)bb[i];
  }
  if (a.res) {
    const bf16x8 rr = *reinterpret_cast<const bf16x8*>(a.res + m * a.N + n);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += (float)rr[i];
  }
  if (a.gate) {
    const bf16x8 gg = *reinterpret_cast<const bf16x8*>(a.gate + m * a.N + n);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)gg[i] > 0.f ? v[i] : 0.f;
  }
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (__bf16)(a.relu ? fmaxf(v[i], 0.f) : v[i]);
  *reinterpret_cast<bf16x8*>(a.y + m * a.N + n) = o;
}

// Tile configurations of gemm_big_kernel the host may pick (index = the `cfg` of dskd_gemm_debug_config).
struct BigCfg {
  int wm, wn, ns, occ;      // waves along tokens / outputs, LDS stages, resident workgroups per CU
};
constexpr int kBigCfgs = 6;
__host__ inline const BigCfg& big_cfg(int i) {
  static const BigCfg t[kBigCfgs] = {
      {2, 2, 2, 2},      // 1: 128 x 128, 4 waves, 64 KB
      {4, 2, 3, 1},      // 2: 256 x 128, 8 waves, 144 KB
      {2, 4, 3, 1},      // 3: 128 x 256, 8 waves, 144 KB
      {4, 1, 2, 2},      // 4: 256 x 64, 4 waves, 80 KB
      {2, 2, 3, 1},      // 5: 128 x 128, 4 waves, 96 KB
      {4, 2, 2, 1},      // 6: 256 x 128, 8 waves, 96 KB
  };
  return t[i];
}

template <int WM, int WN, int NS, bool CONV3>
int launch_big(const GemmArgs& a, const BigPlan& p, long long items, hipStream_t st) {
  constexpr int BM = 64 * WM, BN = 64 * WN, WAVES = WM * WN;
  constexpr int stage = 2 * (BM * 64 + BN * 64);
  constexpr int lds_pipe = NS * stage, lds_epi = WAVES * 32 * 272;
  constexpr int LDS = lds_pipe > lds_epi ? lds_pipe : lds_epi;
  static bool done[64] = {};
  if (!reserve_lds((const void*)gemm_big_kernel<WM, WN, NS, CONV3>, LDS, done))
    return fail(DSKD_ERR_LAUNCH, "dskd_gemm_nt: cannot reserve %d bytes of LDS", LDS);
  hipLaunchKernelGGL((gemm_big_kernel<WM, WN, NS, CONV3>), dim3((unsigned)items), dim3(WAVES * 64), LDS, st, a, p);
  return check_launch("dskd_gemm_nt/big");
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradients: C[N, K] += G[M, N]^T X[M, K]  (dW = dY^T X of a Linear layer / 1x1 convolution), bf16 in, f32 out.
// The reduction runs over the ROWS (tokens) of both operands, so neither is K-contiguous for the matrix cores: the
// [tokens][channels] tiles go global -> LDS as they lie (256-byte row pieces, LDS-DMA) and the fragments come out of them
// with ds_read_b64_tr_b16 (4 tokens x 16 channels per 16-lane group, delivered token-contiguous: cdna_hip_programming.md
// T10).  The output is tiny (64 .. 2048 squared) and M is 4 200 .. 267 200: the tokens are split over `splits`
// workgroups per output tile, each adding its 128 x 128 f32 tile with global atomics into the zero-filled result (the
// atomic volume is splits x N x K x 4 bytes: the host picks the split so that it stays ~16 MB).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#ifndef DSKD_TN_TOK
#define DSKD_TN_TOK 64
#endif
#ifndef DSKD_TN_NS
#define DSKD_TN_NS 3
#endif
#ifndef DSKD_TN_ATOMIC_MB
#define DSKD_TN_ATOMIC_MB 16.0
#endif

struct TnArgs {
  const __bf16* g;       // [M, ldg]: columns [0, N) used
  const __bf16* x;       // [M, ldx]: columns [0, K) used
  float* c;              // [N, K] f32, += (zero-filled by the caller)
  long long M;
  int N, K, ldg, ldx;
  int splits;            // workgroups along M per output tile
  long long chunk;       // tokens per split (a multiple of 32)
  int tn;                // 128-row groups of the output tile (1: 128 x 128, 2: 256 x 128)
  float* db;             // null | [splits][N] f32 planes: column sums of g (the bias gradient), formed by the k-tile-0 waves
  // CONV (3x3 weight gradient): x is the INPUT image [B, Hi, Wi, C = ldx], row m of g is output pixel (img, ho, wo); column
  // tap * C + c of the virtual x operand is channel c of input pixel (s ho + ky - 1, s wo + kx - 1), zeros outside
  int Hi, Wi, Wo, HoWo, s;
  float inv_howo, inv_wo;
};

// m / d for 0 <= m < 2^24 (exact in f32) with inv = 1 / d: the float quotient is off by at most one
__device__ __forceinline__ int fast_div(int m, int d, float inv) {
  int q = (int)((float)m * inv);
  const int r = m - q * d;
  q += r >= d ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}

__device__ __forceinline__ bf16x8 tr_pair(unsigned a0, unsigned a1) {      // tokens t .. t+3 (a0) and t+4 .. t+7 (a1) of one channel column
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(unsigned long)a0);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(unsigned long)a1);
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <bool ATOMIC, int TN, bool CONV = false>
__global__ __launch_bounds__((4 * TN + 4) * 64) void gemm_tn_kernel(const TnArgs a) {
  // TN = 2 (r4): a 256 (N) x 128 (K) output tile, eight multiplying waves (two per SIMD) + four producers: per stage 48 KB of
  // operands for 4.2 MFLOP instead of 32 KB for 2.1 -- with the 128 x 128 tile the matrix pipe (512 cycles per stage), the
  // vector-memory path (32 KB at 64 B/clk) and the LDS (32 KB written + 64 KB read) were all "about 500 cycles" each and the
  // stage took ~1 400.
  constexpr int ROWB = 256;                  // bytes of one token's 128 channels
  constexpr int TOK = DSKD_TN_TOK;           // tokens per stage
  constexpr int TILE = TOK * ROWB;           // one [tokens][128 channels] image of a stage
  constexpr int STAGE = (TN + 1) * TILE;     // TN images of g, one of x
  constexpr int NS = DSKD_TN_NS;             // stages in LDS; the DMA runs NS - 1 stages ahead
  constexpr int LD = (TN + 1) * TOK / 16;    // LDS-DMA instructions per producer wave and stage
  constexpr int NC = 4 * TN;                 // multiplying waves
  extern __shared__ __attribute__((aligned(16))) char smem[];      // NS stages
  const int lane = threadIdx.x & 63;
  // r4: waves 0-3 multiply (one per SIMD), waves 4-7 only issue the LDS-DMA -- every global_load_lds costs the issuing wave
  // ~100-180 cycles, and with the loads issued by the multiplying waves themselves (r3: one wave per SIMD doing both) those
  // stalls came straight out of the MFMA stream: 1 970 cycles per stage in every layer shape, of which 512 are MFMA and only
  // ~220 waiting for data, barrier or fragments (profiles/r04_gemm_tn_phases.txt).
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wave_all >= NC;
  const int wave = producer ? wave_all - NC : wave_all;      // producer: 0 .. 3 (token rows it fills); consumer: 0 .. NC - 1
  const int wn = wave >> 1, wk = wave & 1;                   // consumer: 64 output rows wn, 64 output columns wk

  const int tiles_k = a.K >> 7, tiles_n = a.N / (128 * TN);
  // r4: the TILES of one token range are neighbours (tile index fastest), so that the workgroups an XCD receives (a
  // contiguous range of vb) share their operand rows through its L2: with the split index fastest (r3) the 32 workgroups
  // of an XCD read 32 different token ranges and every g / x row was pulled from beyond L2 by up to tiles_k + tiles_n XCDs
  // (FFN dW1: 713 MB of fill for 227 MB of operands, i.e. the launch ran at the Infinity Cache's ~8 TB/s)
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = tiles_k * tiles_n;
  const int sp = vb / tiles, t_ = vb - sp * tiles;
  const int tk = t_ % tiles_k, tn = t_ / tiles_k;
  const long long m_begin = (long long)sp * a.chunk;
  const long long m_end = m_begin + a.chunk < a.M ? m_begin + a.chunk : a.M;
  if (m_begin >= m_end) return;
  const int nst = (int)((m_end - m_begin + TOK - 1) / TOK);

  // LDS-DMA: one instruction = 4 token rows x 256 B; wave w fills rows (TOK / 4) w .. of both operands
  // The 16-byte chunk ch of token row r sits at chunk position ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) (guide T10, image (b):
  // without it the four rows of a transposing read fall on the same banks) -- applied on the SOURCE address, the LDS
  // image of one DMA instruction stays lane-linear.
  const int lrow = lane >> 4;
  const char* gp = reinterpret_cast<const char*>(a.g) + (long long)tn * 256 * TN;
  // CONV: a 128-column tile of the virtual operand lies inside ONE tap (C is a multiple of 128)
  const int tap = CONV ? (tk * 128) / a.ldx : 0;
  const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;              // tap / 3 for tap < 9
  const char* xp = reinterpret_cast<const char*>(a.x) + (CONV ? (long long)(tk * 128 - tap * a.ldx) * 2 : (long long)tk * 256);
  const char* const zp = g_zero_page;
  auto issue = [&](int st, int j0, int j1) {          // the LDS-DMA instructions of token groups [j0, j1) of stage st
    char* sg = smem + (st % NS) * STAGE;
    char* sx = sg + TN * TILE;
#pragma unroll
    for (int j = 0; j < TOK / 16; ++j) {
      if (j < j0 || j >= j1) continue;
      const int row = wave * (TOK / 4) + j * 4 + lrow;
      const long long m = m_begin + (long long)st * TOK + row;
      const bool ok = m < m_end;                               // rows past this split's tokens contribute zeros
      const int lcol = ((lane & 15) ^ ((lrow << 2) | ((wave * (TOK / 16) + j) & 3))) * 16;
#pragma unroll
      for (int im = 0; im < TN; ++im)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ok ? gp + m * a.ldg * 2 + im * 256 + lcol : zp),
                                         (__attribute__((address_space(3))) void*)(sg + im * TILE + (wave * (TOK / 4) + j * 4) * ROWB), 16, 0, 0);
      const char* xsrc = zp;
      if constexpr (CONV) {
        if (ok) {
          const int mi = (int)m;
          const int img = fast_div(mi, a.HoWo, a.inv_howo), rem = mi - img * a.HoWo;
          const int ho = fast_div(rem, a.Wo, a.inv_wo), wo = rem - ho * a.Wo;
          const int hi = a.s * ho + ky - 1, wi = a.s * wo + kx - 1;
          if ((unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi)
            xsrc = xp + (((long long)img * a.Hi + hi) * a.Wi + wi) * a.ldx * 2 + lcol;
        }
      } else if (ok) {
        xsrc = xp + m * a.ldx * 2 + lcol;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xsrc,
                                       (__attribute__((address_space(3))) void*)(sx + (wave * (TOK / 4) + j * 4) * ROWB), 16, 0, 0);
    }
  };

  // fragment addresses: 16-lane group g reads tokens 16 s + 8 h + q (+ 4) at the 16 channels 16 (g & 1) .. of its tile
  const int q = (lane >> 2) & 3, p = lane & 3, h = lane >> 5, g1 = (lane >> 4) & 1;
  const unsigned base = lds_offset(smem);
  // rows 8 h + q (+ 4, + 16 s): row & 3 = q, (row >> 2) & 3 = 2 h (+ 1 for the second read of a pair; + 0 for the k-steps,
  // which are 16 rows apart) -> the swizzle of the pair's second read differs in one bit: two addresses per fragment
  unsigned ga[2][2], xa[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int row = 8 * h + q + 4 * e;
      const int f = ((row & 3) << 2) | ((row >> 2) & 3);
      const int cg = (wn & 1) * 64 + t * 32 + 16 * g1 + 4 * p, cx = wk * 64 + t * 32 + 16 * g1 + 4 * p;
      ga[t][e] = base + (wn >> 1) * TILE + row * ROWB + ((((cg >> 3) ^ f)) << 4) + ((cg & 7) << 1);
      xa[t][e] = base + TN * TILE + row * ROWB + ((((cx >> 3) ^ f)) << 4) + ((cx & 7) << 1);
    }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // r4 loop: ONE barrier per stage, placed in front of the stage's LAST MFMA group; the first fragments of stage s + 1 are
  // requested right behind it (while that group runs), and the LDS-DMA instructions of stage s + NS are issued between the
  // MFMA groups.  The r3 loop ran barrier -> 8 DMA issues -> fragment reads -> MFMAs in a row on one wave per SIMD:
  // ~1 400 cycles per stage for 512 of MFMA (what the phase stamps of the same structure in gemm_big_kernel showed,
  // profiles/r04_gemm_phases.txt).
  auto wait_stages = [&](int n) {           // all but my n youngest stages have landed
    if (n >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LD) : "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // Barrier k (k = 0 .. nst - 1) means: stage k has landed (the producers waited for their loads) AND stage k - 1 has been
  // read (the consumers waited for their fragments) -- so behind it the producers refill the buffer of stage k - 1 with
  // stage k - 1 + NS.  Both roles pass exactly nst barriers.
  if (producer) {
#pragma unroll
    for (int i = 0; i < NS; ++i)
      if (i < nst) issue(i, 0, TOK / 16);
    wait_stages(min(NS - 1, nst - 1));
    __builtin_amdgcn_s_barrier();                          // barrier 0
    for (int st = 0; st + 1 < nst; ++st) {
      wait_stages(min(NS - 2, nst - 2 - st));              // my part of stage st + 1 has landed
      __builtin_amdgcn_s_barrier();                        // barrier st + 1
      if (st + NS < nst) issue(st + NS, 0, TOK / 16);      // ... into the buffer of stage st
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // Consumer stream.  One wave per SIMD: whatever the wave issues besides its MFMAs has to fit into the gaps behind them
  // (an MFMA holds the issue port for 8 of its 32 cycles), so (1) the 16 fragment reads of k-step s + 1 are interleaved with
  // the four MFMAs of k-step s instead of standing in front of them, (2) their addresses are 8 per-stage bases + immediate
  // offsets (64 v_add per stage before), (3) the stage's barrier sits in front of its last MFMA group, whose gaps take the
  // first reads of the next stage.
  bf16x4 fr[2][8];                          // [buffer][g0.lo, g0.hi, g1.lo, g1.hi, x0.lo, x0.hi, x1.lo, x1.hi]
  const bool bias_wave = !ATOMIC && a.db != nullptr && tk == 0 && wk == 0;      // wave-uniform
  const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
  f32x16 accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
  unsigned sa[8];                           // this stage's fragment bases: g (t, e) = sa[2 t + e], x (t, e) = sa[4 + 2 t + e]
  auto set_bases = [&](int st) {
    const unsigned o = (st % NS) * STAGE;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 2; ++e) { sa[2 * t + e] = ga[t][e] + o; sa[4 + 2 * t + e] = xa[t][e] + o; }
  };
#define TN_READ(buf, idx, KS) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fr[buf][idx]) : "v"(sa[idx]), "n"((KS) * 16 * ROWB))
#define TN_READS(BUF, KS, F) TN_READ(BUF, F, KS); TN_READ(BUF, F + 1, KS); TN_READ(BUF, F + 2, KS); TN_READ(BUF, F + 3, KS)
  auto frag = [&](int b, int i) { return bf16x8{fr[b][i][0], fr[b][i][1], fr[b][i][2], fr[b][i][3], fr[b][i + 1][0], fr[b][i + 1][1], fr[b][i + 1][2], fr[b][i + 1][3]}; };
#define TN_MFMA(B, I, J) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(B, 2 * (I)), frag(B, 4 + 2 * (J)), acc[I][J], 0, 0, 0)
#define TN_FENCE() __builtin_amdgcn_sched_barrier(0)
#define TN_WAIT0(B) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[B][0]), "+v"(fr[B][1]), "+v"(fr[B][2]), "+v"(fr[B][3]), "+v"(fr[B][4]), \
                                 "+v"(fr[B][5]), "+v"(fr[B][6]), "+v"(fr[B][7])::"memory")
  // one k-step: the 4 MFMAs on buffer B with the 8 reads of (buffer NB, k-step NK) in their gaps
  // r4 (late): the bias gradient = column sums of g rides along as one more product per g fragment, g^T x ONES, on the
  // waves that own output columns 0-63 of k-tile 0 (each row block of the result exactly once): the colsum + hand-over
  // launches of every Linear layer's backward (~150 launches of 3-10 us per step) disappear into the dW launch and its reduce.
#define TN_BIAS(B) if (bias_wave) { accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(B, 0), ones, accb[0], 0, 0, 0); \
                                    accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(B, 2), ones, accb[1], 0, 0, 0); }
#define TN_GROUP(B, NB, NK)                                                   \
  TN_MFMA(B, 0, 0); TN_FENCE(); TN_READS(NB, NK, 0); TN_FENCE();              \
  TN_MFMA(B, 0, 1); TN_FENCE(); TN_READS(NB, NK, 4); TN_FENCE();              \
  TN_MFMA(B, 1, 0); TN_MFMA(B, 1, 1); TN_BIAS(B) TN_FENCE();
#ifdef DSKD_GEMM_PROFILE
  long long t_wait = 0, t_bar = 0, t_lgkm = 0, t_mark;
#define TN_T0() t_mark = (long long)__builtin_amdgcn_s_memtime()
#define TN_ACC(v) do { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); v += n_ - t_mark; t_mark = n_; } while (0)
#else
#define TN_T0()
#define TN_ACC(v)
#endif
  PROF(0);
  __builtin_amdgcn_s_barrier();                            // barrier 0: stage 0 has landed
  PROF(1);
  set_bases(0);
  TN_READS(0, 0, 0); TN_READS(0, 0, 4);
  for (int st = 0; st < nst; ++st) {
    TN_T0(); TN_WAIT0(0); TN_ACC(t_lgkm); TN_FENCE();
    TN_GROUP(0, 1, 1)                                     // k-step 0 (buffer 0), reads of k-step 1 -> buffer 1
    TN_T0(); TN_WAIT0(1); TN_ACC(t_lgkm); TN_FENCE();
    TN_GROUP(1, 0, 2)                                     // k-step 1, reads of k-step 2 -> buffer 0
    TN_T0(); TN_WAIT0(0); TN_ACC(t_lgkm); TN_FENCE();
    TN_GROUP(0, 1, 3)                                     // k-step 2, reads of k-step 3 -> buffer 1
    TN_T0(); TN_WAIT0(1); TN_ACC(t_lgkm); TN_FENCE();      // every read of this stage has returned
    if (st + 1 < nst) {
      TN_T0();
      __builtin_amdgcn_s_barrier();                        // barrier st + 1: stage st + 1 has landed; stage st is read
      TN_ACC(t_bar);
      set_bases(st + 1);
    }
    TN_FENCE();
    // k-step 3 (buffer 1) with the reads of the next stage's k-step 0 -> buffer 0 in its gaps.  After the last stage the same
    // reads go to this stage's buffer again and are never used: ONE copy of the group keeps the accumulators in place (an
    // if / else pair made the compiler hold a second set of 64 registers)
    TN_GROUP(1, 0, 0)
    TN_FENCE();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the look-ahead reads of the last stage land in dead registers
#undef TN_GROUP
#undef TN_BIAS
#undef TN_WAIT0
#undef TN_FENCE
#undef TN_MFMA
#undef TN_READS
#undef TN_READ
  PROF(2);
#ifdef DSKD_GEMM_PROFILE
  if (g_gemm_prof && threadIdx.x == 0) {
    long long* pp = g_gemm_prof + (long long)blockIdx.x * 8;
    pp[3] = t_wait; pp[5] = ((long long)nst << 40) | (t_bar & 0xFFFFFFFFFFll); pp[7] = t_lgkm;
  }
#endif
  // acc[i][j]: rows = n (register index), column = k (lane): 32 consecutive k per half-wave -> 128-byte atomic rows
  const int r = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // ATOMIC: every split adds into the one [N, K] result.  Otherwise split sp owns plane sp of a [splits, N, K] scratch
      // and writes it with plain stores (the chip's float-atomic rate is ~1.3 TB/s: the 16 MB flush of a launch was 12 us
      // of its 26-90); reduce_cvt_kernel sums the planes and hands the result over in the parameter's dtype.
      float* cp = a.c + (ATOMIC ? 0ll : (long long)sp * a.N * a.K) +
                  (long long)(tn * 128 * TN + wn * 64 + i * 32 + 4 * h) * a.K + tk * 128 + wk * 64 + j * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if constexpr (ATOMIC) atomicAdd(cp + (long long)((e & 3) + 8 * (e >> 2)) * a.K, acc[i][j][e]);
        else cp[(long long)((e & 3) + 8 * (e >> 2)) * a.K] = acc[i][j][e];
      }
    }
  if (bias_wave && r == 0) {          // every column of accb holds the same sums: lanes 0 and 32 write their 16 rows each
    float* bp = a.db + (long long)sp * a.N + tn * 128 * TN + wn * 64 + 4 * h;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) bp[i * 32 + (e & 3) + 8 * (e >> 2)] = accb[i][e];
  }
}

// dst (bf16) = src (f32), src = 0: hands a weight gradient over in the parameter's dtype and leaves the accumulator that
// gemm_tn_kernel adds into zeroed for its next use -- one launch instead of a zero fill before and a cast after.
__global__ __launch_bounds__(256) void cvt_clear_kernel(float* __restrict__ src, __bf16* __restrict__ dst, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<f32x4*>(src)[i];
    typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
    reinterpret_cast<bf16x4v*>(dst)[i] = bf16x4v{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    reinterpret_cast<f32x4*>(src)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// dst[i] (bf16) = sum over the planes of part[p][i]: the split-K partial products of gemm_tn_kernel<false> summed in a fixed
// order (deterministic, unlike the atomic form) and handed over in the parameter's dtype.
// (r4: a second, short segment -- the bias-gradient planes [planes][n2] behind the product planes -- is summed by the blocks
// behind the first segment's: i in [n_pad, n_pad + n2), n_pad = n rounded up to the block size)
__global__ __launch_bounds__(256) void reduce_cvt_kernel(const float* __restrict__ part, int planes, long long n,
                                                         __bf16* __restrict__ dst, const float* __restrict__ part2 = nullptr,
                                                         long long n2 = 0, __bf16* __restrict__ dst2 = nullptr) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n_pad = (n + 255) / 256 * 256;
  if (i >= n_pad) { i -= n_pad; part = part2; n = n2; dst = dst2; }
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = 0;
  for (; p + 3 < planes; p += 4) {
    s0 += part[(long long)p * n + i];
    s1 += part[(long long)(p + 1) * n + i];
    s2 += part[(long long)(p + 2) * n + i];
    s3 += part[(long long)(p + 3) * n + i];
  }
  for (; p < planes; ++p) s0 += part[(long long)p * n + i];
  dst[i] = (__bf16)((s0 + s1) + (s2 + s3));
}

// The same sum for MANY planes of a small result (64 planes of 256 x 256: the thin layers): 32 outputs per workgroup, 8 lanes
// of planes each (independent loads), one LDS step -- a thread of the form above walks its 64 planes alone.
__global__ __launch_bounds__(256) void reduce_cvt_wide_kernel(const float* __restrict__ part, int planes, long long n,
                                                              __bf16* __restrict__ dst, const float* __restrict__ part2 = nullptr,
                                                              long long n2 = 0, __bf16* __restrict__ dst2 = nullptr) {
  __shared__ float s_part[8][33];
  const int cl = threadIdx.x & 31, kg = threadIdx.x >> 5;
  long long i = (long long)blockIdx.x * 32 + cl;
  const long long n_pad = (n + 31) / 32 * 32;
  if ((long long)blockIdx.x * 32 >= n_pad) { i -= n_pad; part = part2; n = n2; dst = dst2; }      // block-uniform
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int p = kg;
    for (; p + 8 < planes; p += 16) {
      s0 += part[(long long)p * n + i];
      s1 += part[(long long)(p + 8) * n + i];
    }
    if (p < planes) s0 += part[(long long)p * n + i];
  }
  s_part[kg][cl] = s0 + s1;
  __syncthreads();
  if (kg == 0 && i < n) {
    float t = s_part[0][cl];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += s_part[k][cl];
    dst[i] = (__bf16)t;
  }
}

}  // namespace
}  // namespace dskd

using namespace dskd;

// ---- tile choice ---------------------------------------------------------------------------------------------------
// cfg 0 = gemm_nt_kernel (64 x 128 / 128 x 64 tiles, 3 workgroups per CU; epilogue by M), 1 .. kBigCfgs = big_cfg(cfg - 1),
// kBigCfgs + 1 / + 2 / + 3 = gemm_nt_kernel with the register / the LDS epilogue / the LDS epilogue without the early
// residual + gate reads forced.
// dskd_gemm_nt_tune: a tuning hook for microbenchmarks and tests (scratch/r04_gemm_big.py) -- cfg < 0: automatic
// (default); splits: 0 automatic, 1 never split, > 1 forced (clamped to the K stages and the scratch).
static int g_tune_cfg = -1, g_tune_splits = 0;
static int g_tn_force = 0;      // tuning hook (dskd_gemm_nt_tune with cfg -2 / -3): force the 128 x 128 / 256 x 128 dW tile
static int g_tn_splits = 0;     // ... and, with splits > 0, the number of token splits of the dW kernels
extern "C" int dskd_gemm_nt_tune(int cfg, int splits) {
  if (cfg == -2 || cfg == -3) {      // dW tile: 128 x 128 / 256 x 128
    g_tn_force = cfg == -2 ? 1 : 2;
    g_tn_splits = splits > 0 ? splits : 0;
    return DSKD_OK;
  }
  if (cfg == -1) g_tn_force = g_tn_splits = 0;
  if (cfg > kBigCfgs + 3) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_nt_tune: cfg %d > %d", cfg, kBigCfgs + 3);
  g_tune_cfg = cfg;
  g_tune_splits = splits;
  return DSKD_OK;
}
extern "C" int64_t dskd_gemm_nt_scratch_bytes(void) { return (int64_t)32 << 20; }
#ifdef DSKD_GEMM_PROFILE
extern "C" int dskd_gemm_nt_profile(void* buf) {      // 8 x int64 per workgroup of the next launches (NULL: off)
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

struct BigChoice {
  int cfg;           // 0: gemm_nt_kernel
  BigPlan plan;
  long long items;   // workgroups of the main launch
  int rem;           // split tiles (fixup grid), 0: no fixup launch
  double t;          // modelled time, us
};

// Model of one launch (us).  Constants fitted to profiles/r04_gemm_big_microbench.txt: a CU sustains ~5.2 MFLOP/us of bf16
// MFMA in these loops and ~62 KB/us of L2 -> LDS fill, whichever is slower paces a K stage; every tile pays a pipeline
// fill (~1.6 us: the first stage's HBM round trip) and its epilogue at the CU's share of HBM bandwidth (~22 KB/us).
static BigChoice model_cfg(int cfg, long long M, int N, int K, int epi_ops, bool conv3, int64_t scratch_bytes) {
  BigChoice c{};
  c.cfg = cfg;
  c.t = 1e30;
  int BM, BN, occ;
  if (cfg == 0 || cfg > kBigCfgs) {      // kBigCfgs + 1 / + 2: the small tile with the register / the LDS epilogue forced
    BN = (N % 128) ? 64 : 128; BM = (N % 128) ? 128 : 64; occ = 3;
  } else {
    const BigCfg& g = big_cfg(cfg - 1);
    BM = 64 * g.wm; BN = 64 * g.wn; occ = g.occ;
  }
  if (N % BN) return c;
  const long long tiles = ((M + BM - 1) / BM) * (N / BN);
  if (tiles > 0x3FFFFFFFll) return c;
  const int nk = K >> 6;
  const long long slots = 256ll * occ;
  const double t_stage = occ * fmax(BM * (double)BN * 128.0 / 5.2e6, (BM + BN) * 128.0 / 62.0e3);
  const double t_epi = occ * (BM * (double)BN * 2.0 * (1 + epi_ops)) / 22.0e3;
  const double t_tile = 1.6 + nk * t_stage + t_epi;
  const long long rounds = tiles / slots, R = tiles - rounds * slots;
  c.plan.full = (int)tiles; c.plan.splits = 1; c.plan.planes = nullptr; c.items = tiles; c.rem = 0;
  if (R == 0) { c.t = rounds * t_tile; return c; }
  // last, partial round: R tiles on `slots` slots
  const double alone = R <= 256 ? t_tile / occ * 1.15 : t_tile;      // fewer workgroups than CUs: each has a CU to itself
  c.t = rounds * t_tile + alone;
  if (cfg == 0 || cfg > kBigCfgs || g_tune_splits == 1) return c;
  long long S = g_tune_splits > 1 ? g_tune_splits : slots / R;
  if (S > nk) S = nk;
  if (S > 32) S = 32;
  while (S > 1 && R * S * BM * BN * 4ll > scratch_bytes) --S;
  if (S <= 1) return c;
  const double t_split = 1.6 + (double)((nk + S - 1) / S) * t_stage * (R * S <= 256 ? 1.0 / occ * 1.15 : 1.0) +
                         occ * BM * (double)BN * 4.0 / 22.0e3 + 3.0;      // f32 partial tile out + the fixup launch
  {      // measured: with R * 2 <= slots the split wins on every shape where a big tile is chosen at all
    c.t = rounds * t_tile + t_split;
    c.plan.full = (int)(rounds * slots); c.plan.splits = (int)S; c.items = rounds * slots + R * S; c.rem = (int)R;
  }
  return c;
}

static BigChoice choose_cfg(long long M, int N, int K, int epi_ops, bool conv3, int64_t scratch_bytes) {
  if (g_tune_cfg >= 0) return model_cfg(g_tune_cfg, M, N, K, epi_ops, conv3, scratch_bytes);
  // Measured (profiles/r04_gemm_big_microbench.txt, all 46 layer shapes of the step x 6 big tiles x with / without the split-K
  // remainder): the big tiles win only where K is very long and the small tile has too few tiles to hide its own
  // pipeline fill -- the 3x3 convolutions of ResNet stage 4 (K = 4 608, 4 200 tokens: 39.2 against 45.8 us with 128 x 256
  // tiles + the split remainder).  Everywhere else the 64 x 128 kernel's three resident workgroups per CU are faster.
  if (conv3 && K >= 4608 && N % 256 == 0 && M <= 8192) {
    const BigChoice c = model_cfg(3, M, N, K, epi_ops, conv3, scratch_bytes);
    if (c.t < 1e30) return c;
  }
  return model_cfg(0, M, N, K, epi_ops, conv3, scratch_bytes);
}

template <bool CONV3>
static int launch_choice(const GemmArgs& a, BigChoice c, void* scratch, hipStream_t st) {
  if (c.cfg == 0 || c.cfg > kBigCfgs) {
    // 64 tokens x 128 outputs per workgroup (each wave 32 x 64): measured faster than 128 x 128 of the same kernel on every
    // layer shape of the trunk (scratch/r03_conv1x1.py) -- three workgroups per CU instead of two.  Epilogue through LDS
    // (whole 128-byte lines) for the tall layers: 3 - 10 % on the write-heavy ones (l1.conv3 82.6 -> 78.4 us, l1.down 42.8 ->
    // 38.2, l2.conv1 dX 46.1 -> 41.3), a wash or a small loss on the 4 200-row layers (profiles/r04_gemm_big_microbench.txt)
    // r4 (tools/prof/gemm_tiles_bench.py epi, interleaved): with the residual / gate reads requested before the K loop the
    // LDS epilogue gains 13 - 16 % on the HBM-bound stage-1 layers that read a residual (l1.conv3 80 -> 70 us, l1.conv1 dX
    // 105 -> 88); input gradients that only read a gate are 1 - 3 % faster with the register epilogue on every shape
    const bool lds_epi = c.cfg == 0 ? (a.M >= 8192 && !(a.gate && !a.res)) : c.cfg >= kBigCfgs + 2;
    // residual / gate requested before the K loop (EpiPre) where the epilogue reads one: +36 VGPRs, nothing for the others
    const bool pre = lds_epi && DSKD_EPI_PRE && (a.res || a.gate) && c.cfg != kBigCfgs + 3;
    if (a.N % 128)
      return pre ? launch_gemm<64, 1, CONV3, true, true>(a, st)
                 : lds_epi ? launch_gemm<64, 1, CONV3, true>(a, st) : launch_gemm<64, 1, CONV3, false>(a, st);
    return pre ? launch_gemm<128, 1, CONV3, true, true>(a, st)
               : lds_epi ? launch_gemm<128, 1, CONV3, true>(a, st) : launch_gemm<128, 1, CONV3, false>(a, st);
  }
  c.plan.planes = (float*)scratch;
  int rc;
  switch (c.cfg) {
    case 1: rc = launch_big<2, 2, 2, CONV3>(a, c.plan, c.items, st); break;
    case 2: rc = launch_big<4, 2, 3, CONV3>(a, c.plan, c.items, st); break;
    case 3: rc = launch_big<2, 4, 3, CONV3>(a, c.plan, c.items, st); break;
    case 4: rc = launch_big<4, 1, 2, CONV3>(a, c.plan, c.items, st); break;
    case 5: rc = launch_big<2, 2, 3, CONV3>(a, c.plan, c.items, st); break;
    default: rc = launch_big<4, 2, 2, CONV3>(a, c.plan, c.items, st); break;
  }
  if (rc || c.rem == 0) return rc;
  const BigCfg& g = big_cfg(c.cfg - 1);
  const int BM = 64 * g.wm, BN = 64 * g.wn;
  hipLaunchKernelGGL(gemm_fixup_kernel, dim3((unsigned)c.rem, (unsigned)((BM * (BN >> 3) + 255) / 256)), dim3(256), 0, st, a,
                     c.plan, BM, BN);
  return check_launch("dskd_gemm_nt/fixup");
}

static int gemm_nt_impl(const void* x, const void* w, const void* bias, const void* res, const void* gate, void* y,
                        int64_t M, int N, int K, int relu, int stride, int Ho, int Wo, int Hi, int Wi, int dtype,
                        void* scratch, int64_t scratch_bytes, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_nt: bf16 only");
  if (!x || !w || !y || M < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_nt: null pointer or negative row count");
  if (N <= 0 || K <= 0 || (N & 63) || (K & 63))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_nt: N and K must be positive multiples of 64 (got N=%d K=%d)", N, K);
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (mis(x) || mis(w) || mis(y) || (bias && mis(bias)) || (res && mis(res)) || (gate && mis(gate)) || (scratch && mis(scratch)))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_nt: pointers must be 16-byte aligned");
  if (stride < 0 || (stride > 0 && (Ho <= 0 || Wo <= 0 || Hi <= 0 || Wi <= 0 || M % ((int64_t)Ho * Wo) != 0 ||
                                    (int64_t)stride * (Ho - 1) >= Hi || (int64_t)stride * (Wo - 1) >= Wi)))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_nt: bad row map (stride=%d Ho=%d Wo=%d Hi=%d Wi=%d)", stride, Ho, Wo, Hi, Wi);
  if (scratch_bytes < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_nt: negative scratch size");
  if (M == 0) return DSKD_OK;
  GemmArgs a;
  a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = (const __bf16*)bias; a.res = (const __bf16*)res;
  a.gate = (const __bf16*)gate;
  a.y = (__bf16*)y; a.M = M; a.N = N; a.K = K; a.relu = relu;
  a.s = stride; a.HoWo = stride ? Ho * Wo : 1; a.Wo = stride ? Wo : 1; a.Hi = Hi; a.Wi = Wi;
  a.C = K; a.cshift = 0;
  const BigChoice c = choose_cfg(M, N, K, (res ? 1 : 0) + (gate ? 1 : 0), false, scratch ? scratch_bytes : 0);
  if (c.t >= 1e30) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_nt: tile configuration %d cannot take N=%d", c.cfg, N);
  return launch_choice<false>(a, c, scratch, (hipStream_t)stream);
}

extern "C" int dskd_gemm_nt(const void* x, const void* w, const void* bias, const void* res, void* y, int64_t M, int N, int K,
                            int relu, int stride, int Ho, int Wo, int Hi, int Wi, int dtype, void* stream) {
  return gemm_nt_impl(x, w, bias, res, nullptr, y, M, N, K, relu, stride, Ho, Wo, Hi, Wi, dtype, nullptr, 0, stream);
}

extern "C" int dskd_gemm_nt_dx(const void* g, const void* wt, const void* res, const void* gate, void* y, int64_t M, int N,
                               int K, int dtype, void* stream) {
  return gemm_nt_impl(g, wt, nullptr, res, gate, y, M, N, K, 0, 0, 0, 0, 0, 0, dtype, nullptr, 0, stream);
}

extern "C" int dskd_gemm_nt_ws(const void* x, const void* w, const void* bias, const void* res, const void* gate, void* y,
                               int64_t M, int N, int K, int relu, int stride, int Ho, int Wo, int Hi, int Wi, int dtype,
                               void* scratch, int64_t scratch_bytes, void* stream) {
  return gemm_nt_impl(x, w, bias, res, gate, y, M, N, K, relu, stride, Ho, Wo, Hi, Wi, dtype, scratch, scratch_bytes, stream);
}

static int conv3x3_impl(const void* x, const void* w, const void* bias, const void* res, const void* gate, void* y, int B,
                        int Hi, int Wi, int C, int N, int stride, int relu, int dtype, void* scratch, int64_t scratch_bytes,
                        void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3: bf16 only");
  if (!x || !w || !y || B < 0 || Hi <= 0 || Wi <= 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3: null pointer or bad size");
  int cshift = 0;
  while ((64 << cshift) < C) ++cshift;
  if (C <= 0 || (64 << cshift) != C || cshift > 4 || N <= 0 || (N & 63))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3: C must be 64 * 2^k (<= 1024) and N a multiple of 64 (got C=%d N=%d)", C, N);
  if (stride != 1 && stride != 2) return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3: stride 1 or 2 (got %d)", stride);
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (mis(x) || mis(w) || mis(y) || (bias && mis(bias)) || (res && mis(res)) || (gate && mis(gate)) || (scratch && mis(scratch)))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3: pointers must be 16-byte aligned");
  if (scratch_bytes < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3: negative scratch size");
  const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;        // kernel 3, padding 1
  if ((long long)Hi * Wi * C * 2 >= 0x7FFFFFFFll) return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3: image too large");
  if (B == 0) return DSKD_OK;
  GemmArgs a;
  a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = (const __bf16*)bias; a.res = (const __bf16*)res;
  a.gate = (const __bf16*)gate;
  a.y = (__bf16*)y; a.M = (long long)B * Ho * Wo; a.N = N; a.K = 9 * C; a.relu = relu;
  a.s = stride; a.HoWo = Ho * Wo; a.Wo = Wo; a.Hi = Hi; a.Wi = Wi; a.C = C; a.cshift = cshift;
  const BigChoice c = choose_cfg(a.M, N, a.K, (res ? 1 : 0) + (gate ? 1 : 0), true, scratch ? scratch_bytes : 0);
  if (c.t >= 1e30) return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3: tile configuration %d cannot take N=%d", c.cfg, N);
  return launch_choice<true>(a, c, scratch, (hipStream_t)stream);
}

extern "C" int dskd_conv3x3(const void* x, const void* w, const void* bias, const void* res, void* y, int B, int Hi, int Wi,
                            int C, int N, int stride, int relu, int dtype, void* stream) {
  return conv3x3_impl(x, w, bias, res, nullptr, y, B, Hi, Wi, C, N, stride, relu, dtype, nullptr, 0, stream);
}

extern "C" int dskd_conv3x3_dx(const void* g, const void* wt, const void* gate, void* y, int B, int Hi, int Wi, int C, int N,
                               int dtype, void* stream) {
  return conv3x3_impl(g, wt, nullptr, nullptr, gate, y, B, Hi, Wi, C, N, 1, 0, dtype, nullptr, 0, stream);
}

extern "C" int dskd_conv3x3_ws(const void* x, const void* w, const void* bias, const void* res, const void* gate, void* y, int B,
                               int Hi, int Wi, int C, int N, int stride, int relu, int dtype, void* scratch,
                               int64_t scratch_bytes, void* stream) {
  return conv3x3_impl(x, w, bias, res, gate, y, B, Hi, Wi, C, N, stride, relu, dtype, scratch, scratch_bytes, stream);
}

static int gemm_tn_plan(const void* g, const void* x, const void* c, int64_t M, int N, int K, int ldg, int ldx, int dtype,
                        TnArgs* a, long long* tiles_out, int force_tn = 0) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_tn: bf16 only");
  if (!g || !x || !c || M < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_tn: null pointer or negative row count");
  if (N <= 0 || K <= 0 || (N & 127) || (K & 127) || ldg < N || ldx < K || (ldg & 7) || (ldx & 7))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_tn: N and K must be multiples of 128, row strides multiples of 8 (got N=%d K=%d "
                "ldg=%d ldx=%d)", N, K, ldg, ldx);
  if ((reinterpret_cast<uintptr_t>(g) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(c) & 15))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_tn: pointers must be 16-byte aligned");
  a->g = (const __bf16*)g; a->x = (const __bf16*)x; a->c = (float*)c; a->M = M; a->N = N; a->K = K; a->ldg = ldg; a->ldx = ldx;
  a->db = nullptr;
  // 256 x 128 output tiles only for the large products of the encoder FFN (N K >= 256 K, M >= 64 K tokens: 3 % faster there,
  // 10-25 % slower on the convolution shapes: profiles/r04_gemm_tn_phases.txt)
  a->tn = (N % 256 == 0 && (long long)N * K >= 262144 && M >= 65536) ? 2 : 1;
  if (force_tn || g_tn_force) a->tn = ((g_tn_force ? g_tn_force : force_tn) == 2 && N % 256 == 0) ? 2 : 1;
  const long long tiles = (long long)(N / (128 * a->tn)) * (K >> 7);
  // splits: one workgroup per CU (256 in all: each flushes its 64 KB tile, 16 MB per launch), two per CU
  // where that still leaves the flush volume small and >= 1 024 tokens per workgroup; never fewer than 256 tokens each
  long long sp = 256 / tiles;
  if (sp < 1) sp = 1;
  if (tiles * sp * 2 * 65536 <= (long long)(DSKD_TN_ATOMIC_MB * 1.0e6) && M / (2 * sp) >= 1024) sp *= 2;
  if (g_tn_splits > 0) sp = g_tn_splits;
  const long long by_work = (M + 255) / 256;
  if (sp > by_work) sp = by_work;
  if (sp < 1) sp = 1;
  a->chunk = (((M + sp - 1) / sp) + DSKD_TN_TOK - 1) / DSKD_TN_TOK * DSKD_TN_TOK;
  sp = M > 0 ? (M + a->chunk - 1) / a->chunk : 1;       // every split has tokens
  a->splits = (int)sp;
  *tiles_out = tiles;
  return DSKD_OK;
}

template <bool ATOMIC, int TN, bool CONV = false>
static int gemm_tn_launch_t(const TnArgs& a, long long tiles, hipStream_t st) {
  constexpr int lds = DSKD_TN_NS * (TN + 1) * DSKD_TN_TOK * 256;
  static bool done[64] = {};
  if (!reserve_lds((const void*)gemm_tn_kernel<ATOMIC, TN, CONV>, lds, done))
    return fail(DSKD_ERR_LAUNCH, "dskd_gemm_tn: cannot reserve %d bytes of LDS", lds);
  hipLaunchKernelGGL((gemm_tn_kernel<ATOMIC, TN, CONV>), dim3((unsigned)(tiles * a.splits)), dim3((4 * TN + 4) * 64), lds, st, a);
  return check_launch("dskd_gemm_tn");
}
template <bool ATOMIC>
static int gemm_tn_launch(const TnArgs& a, long long tiles, hipStream_t st) {
  return a.tn == 2 ? gemm_tn_launch_t<ATOMIC, 2>(a, tiles, st) : gemm_tn_launch_t<ATOMIC, 1>(a, tiles, st);
}

extern "C" int dskd_gemm_tn(const void* g, const void* x, float* c, int64_t M, int N, int K, int ldg, int ldx, int dtype,
                            void* stream) {
  TnArgs a;
  long long tiles = 0;
  if (int rc = gemm_tn_plan(g, x, c, M, N, K, ldg, ldx, dtype, &a, &tiles)) return rc;
  if (M == 0) return DSKD_OK;
  return gemm_tn_launch<true>(a, tiles, (hipStream_t)stream);
}

extern "C" int64_t dskd_gemm_tn_scratch_bytes(int64_t M, int N, int K) {
  TnArgs a;
  long long tiles = 0;
  static const char dummy[16] __attribute__((aligned(16))) = {};
  if (gemm_tn_plan(dummy, dummy, dummy, M, N, K, N, K, DSKD_DTYPE_BF16, &a, &tiles)) return -1;
  return (int64_t)a.splits * ((int64_t)N * K + N) * (int64_t)sizeof(float);      // product planes + bias-gradient planes
}

static int gemm_tn_bf16_impl(const void* g, const void* x, void* out, void* db_out, void* scratch, int64_t scratch_bytes,
                             int64_t M, int N, int K, int ldg, int ldx, int dtype, void* stream) {
  TnArgs a;
  long long tiles = 0;
  if (!out || (reinterpret_cast<uintptr_t>(out) & 1)) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_tn_bf16: null output");
  if (int rc = gemm_tn_plan(g, x, scratch, M, N, K, ldg, ldx, dtype, &a, &tiles)) return rc;
  if (M == 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_tn_bf16: M must be positive");
  const long long n = (long long)N * K;
  const int64_t need = (int64_t)a.splits * (n + N) * (int64_t)sizeof(float);      // = dskd_gemm_tn_scratch_bytes, with or without db
  if (scratch_bytes < need)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_tn_bf16: scratch of %lld bytes, %lld needed", (long long)scratch_bytes,
                (long long)need);
  float* dbp = db_out ? reinterpret_cast<float*>(scratch) + (long long)a.splits * n : nullptr;
  a.db = dbp;
  if (int rc = gemm_tn_launch<false>(a, tiles, (hipStream_t)stream)) return rc;
  const long long n2 = db_out ? N : 0;
  if (a.splits >= 16 && n <= (1 << 20))
    hipLaunchKernelGGL(reduce_cvt_wide_kernel, dim3((unsigned)((n + 31) / 32 + (n2 + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)scratch, a.splits, n, (__bf16*)out, (const float*)dbp, n2, (__bf16*)db_out);
  else
    hipLaunchKernelGGL(reduce_cvt_kernel, dim3((unsigned)((n + 255) / 256 + (n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)scratch, a.splits, n, (__bf16*)out, (const float*)dbp, n2, (__bf16*)db_out);
  return check_launch("dskd_gemm_tn_bf16/reduce");
}

extern "C" int dskd_gemm_tn_bf16(const void* g, const void* x, void* out, void* scratch, int64_t scratch_bytes, int64_t M,
                                 int N, int K, int ldg, int ldx, int dtype, void* stream) {
  return gemm_tn_bf16_impl(g, x, out, nullptr, scratch, scratch_bytes, M, N, K, ldg, ldx, dtype, stream);
}

extern "C" int dskd_gemm_tn_bias_bf16(const void* g, const void* x, void* out, void* db_out, void* scratch, int64_t scratch_bytes,
                                      int64_t M, int N, int K, int ldg, int ldx, int dtype, void* stream) {
  if (!db_out || (reinterpret_cast<uintptr_t>(db_out) & 1)) return fail(DSKD_ERR_INVALID_ARG, "dskd_gemm_tn_bias_bf16: null db output");
  return gemm_tn_bf16_impl(g, x, out, db_out, scratch, scratch_bytes, M, N, K, ldg, ldx, dtype, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of a 3x3 convolution (padding 1, stride 1 | 2) as the SAME split-K kernel: dW[n][ky][kx][c] = sum over the
// output pixels of dY[pixel][n] * X[pixel shifted by the tap][c] -- gemm_tn over a virtual [pixels, 9 C] operand whose
// 128-column tiles each lie inside one tap, so only the producers' source addresses change (shifted row, zero page outside
// the image).  Replaces MIOpen's igemm_wrw + its f32 workspace helpers (SubTensorOp fill / cast: 1.3 ms per step for the 16
// convolutions of the trunk, and the memset nodes that keep its backward out of a hipGraph).
static int conv3x3_wgrad_plan(const void* g, const void* x, const void* c, int B, int Hi, int Wi, int C, int N, int stride,
                              int dtype, TnArgs* a, long long* tiles) {
  if (B < 1 || Hi < 1 || Wi < 1 || (stride != 1 && stride != 2) || C < 128 || (C & 127) || N < 128 || (N & 127))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3_wgrad: C and N must be multiples of 128, stride 1 or 2 (got C=%d N=%d "
                "stride=%d)", C, N, stride);
  const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  const long long M = (long long)B * Ho * Wo;
  if (M >= (1ll << 24) || (long long)B * Hi * Wi * C >= (1ll << 40))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3_wgrad: %lld output pixels (the index arithmetic is built for < 2^24)", M);
  // 256 x 128 output tiles where N allows (stages 3-4): 36-72 tiles x floor(256 / tiles) splits fill the chip once, and a
  // stage moves 48 KB for 4.2 MFLOP instead of 32 KB for 2.1 (tools/prof/conv3x3_wgrad_bench.py sweep: l3 53 -> 46 us, l4 76 -> 50)
  if (int rc = gemm_tn_plan(g, x, c, M, N, 9 * C, N, 9 * C, dtype, a, tiles, N % 256 == 0 ? 2 : 1)) return rc;
  a->ldx = C; a->Hi = Hi; a->Wi = Wi; a->Wo = Wo; a->HoWo = Ho * Wo; a->s = stride;
  a->inv_howo = 1.0f / (float)(Ho * Wo); a->inv_wo = 1.0f / (float)Wo;
  return DSKD_OK;
}

extern "C" int64_t dskd_conv3x3_wgrad_scratch_bytes(int B, int Hi, int Wi, int C, int N, int stride) {
  TnArgs a;
  long long tiles = 0;
  static const char dummy[16] __attribute__((aligned(16))) = {};
  if (conv3x3_wgrad_plan(dummy, dummy, dummy, B, Hi, Wi, C, N, stride, DSKD_DTYPE_BF16, &a, &tiles)) return -1;
  return (int64_t)a.splits * ((int64_t)N * 9 * C + N) * (int64_t)sizeof(float);
}

static int conv3x3_wgrad_impl(const void* g, const void* x, void* dw, void* db_out, void* scratch, int64_t scratch_bytes, int B,
                              int Hi, int Wi, int C, int N, int stride, int dtype, void* stream) {
  TnArgs a;
  long long tiles = 0;
  if (!dw || (reinterpret_cast<uintptr_t>(dw) & 1)) return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3_wgrad: null output");
  if (int rc = conv3x3_wgrad_plan(g, x, scratch, B, Hi, Wi, C, N, stride, dtype, &a, &tiles)) return rc;
  const long long n = (long long)N * 9 * C;
  const int64_t need = (int64_t)a.splits * (n + N) * (int64_t)sizeof(float);
  if (scratch_bytes < need)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3_wgrad: scratch of %lld bytes, %lld needed", (long long)scratch_bytes,
                (long long)need);
  float* dbp = db_out ? reinterpret_cast<float*>(scratch) + (long long)a.splits * n : nullptr;
  a.db = dbp;
  if (int rc = a.tn == 2 ? gemm_tn_launch_t<false, 2, true>(a, tiles, (hipStream_t)stream)
                         : gemm_tn_launch_t<false, 1, true>(a, tiles, (hipStream_t)stream)) return rc;
  const long long n2 = db_out ? N : 0;
  hipLaunchKernelGGL(reduce_cvt_kernel, dim3((unsigned)((n + 255) / 256 + (n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)scratch, a.splits, n, (__bf16*)dw, (const float*)dbp, n2, (__bf16*)db_out);
  return check_launch("dskd_conv3x3_wgrad/reduce");
}

extern "C" int dskd_conv3x3_wgrad(const void* g, const void* x, void* dw, void* scratch, int64_t scratch_bytes, int B, int Hi,
                                  int Wi, int C, int N, int stride, int dtype, void* stream) {
  return conv3x3_wgrad_impl(g, x, dw, nullptr, scratch, scratch_bytes, B, Hi, Wi, C, N, stride, dtype, stream);
}

extern "C" int dskd_conv3x3_wgrad_bias(const void* g, const void* x, void* dw, void* db_out, void* scratch, int64_t scratch_bytes,
                                       int B, int Hi, int Wi, int C, int N, int stride, int dtype, void* stream) {
  if (!db_out || (reinterpret_cast<uintptr_t>(db_out) & 1)) return fail(DSKD_ERR_INVALID_ARG, "dskd_conv3x3_wgrad_bias: null db output");
  return conv3x3_wgrad_impl(g, x, dw, db_out, scratch, scratch_bytes, B, Hi, Wi, C, N, stride, dtype, stream);
}

extern "C" int dskd_cvt_clear(float* src, void* dst, int64_t n, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_cvt_clear: bf16 only");
  if (!src || !dst || n < 0 || (n & 3) || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 7))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_cvt_clear: null / misaligned pointer or a count that is no multiple of 4");
  if (n == 0) return DSKD_OK;
  const long long n4 = n / 4;
  const unsigned blocks = (unsigned)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
  hipLaunchKernelGGL(cvt_clear_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n4);
  return check_launch("dskd_cvt_clear");
}
