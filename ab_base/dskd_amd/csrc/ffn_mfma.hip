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
// that a workgroup stages a tile's 32 KB with sixteen-byte LDS-DMA loads (global_load_lds) two tiles ahead into four
// buffers, one barrier per tile.  Rows of an A tile are permuted (pi below) so that a lane's 16 accumulator registers
// are 16 CONSECUTIVE hidden units / output features: H, g1, Y, dX are plain row-major tensors for the
// weight-gradient GEMMs that follow (H / g1 are turned through a small LDS scratch into whole 64-byte rows).
//
// MFMA work per launch 93 GFLOP (37 us at the 2.5 PFLOP/s dense peak); HBM 272 MB forward / 454 MB backward.
// Dropout is the mask of dskd_dropout_fwd (Philox4x32-10 on element index / 8, 16-bit fields), never stored: the
// backward reads it off H like dskd_relu_dropout_bwd.  Measured, floor analysis, what was tried: DESIGN.md 4.2c.
// -DDSKD_FFN_EXPERIMENT_NOSTAGE / _NOREAD / _NOEPI and -DDSKD_FFN_RING=n are TIMING-ONLY ablation builds (wrong
// results by construction; scratch/r02_ffn_profiles.sh); the second half of the file is lin256_kernel, GEMM-1 alone.
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
#ifndef DSKD_FFN_RING
#define DSKD_FFN_RING 8
#endif
constexpr int kRing = DSKD_FFN_RING;           // A fragments in flight per wave (32 VGPRs)

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

// A fragment reads are written as asm: hipcc sinks an ordinary LDS load down to its use (one register set, ds_read ->
// s_waitcnt 0 -> MFMA: every MFMA then pays the LDS latency), whatever the source order or sched_group_barrier says.
// The compiler does not count these reads, so each use is preceded by frag_wait<N>: "at most N younger reads in
// flight" (LDS returns in order; reads the compiler issues itself only make the wait longer, never shorter).
__device__ __forceinline__ bf16x8 frag_read(unsigned lds_addr, int byte_offset) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(byte_offset));
  return v;
}
__device__ __forceinline__ void frag_wait(bf16x8& v, int younger) {      // `younger`: a constant after unrolling
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(younger));
}
__device__ __forceinline__ unsigned lds_offset(const void* p) {
  return (unsigned)(unsigned long)((const __attribute__((address_space(3))) char*)p);
}

// Sums over the 32 lanes of a wave half (= the 32 tokens of the wave) of 16 registers at once, one DPP step per call
// (steps 0..4; after step 4 the totals sit in lanes 16..31 and 48..63).  Written as asm: left to the compiler the 80
// adds are SLP-packed into v_pk_add_f32, which cannot take a DPP operand, and every step becomes v_mov_b32_dpp + add.
// Sixteen independent registers per block keep dependent DPP reads 16 instructions apart (a DPP read needs two wait
// states behind the write of its source; s_nop 1 covers the instruction in front of the block).
#define DSKD_DPP16(CTRL)                                                                                              \
  asm("s_nop 1\n\t"                                                                                                   \
      "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\tv_add_f32_dpp %2, %2, %2 " CTRL "\n\t"  \
      "v_add_f32_dpp %3, %3, %3 " CTRL "\n\tv_add_f32_dpp %4, %4, %4 " CTRL "\n\tv_add_f32_dpp %5, %5, %5 " CTRL "\n\t"  \
      "v_add_f32_dpp %6, %6, %6 " CTRL "\n\tv_add_f32_dpp %7, %7, %7 " CTRL "\n\tv_add_f32_dpp %8, %8, %8 " CTRL "\n\t"  \
      "v_add_f32_dpp %9, %9, %9 " CTRL "\n\tv_add_f32_dpp %10, %10, %10 " CTRL "\n\tv_add_f32_dpp %11, %11, %11 " CTRL "\n\t" \
      "v_add_f32_dpp %12, %12, %12 " CTRL "\n\tv_add_f32_dpp %13, %13, %13 " CTRL "\n\tv_add_f32_dpp %14, %14, %14 " CTRL "\n\t" \
      "v_add_f32_dpp %15, %15, %15 " CTRL                                                                            \
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),   \
        "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]))
__device__ __forceinline__ void half_wave_sum16_step(float (&v)[16], int step) {
  switch (step) {
    case 0: DSKD_DPP16("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"); break;
    case 1: DSKD_DPP16("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"); break;
    case 2: DSKD_DPP16("row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"); break;
    case 3: DSKD_DPP16("row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"); break;
    default: DSKD_DPP16("row_bcast:15 row_mask:0xa bank_mask:0xf"); break;
  }
}
#undef DSKD_DPP16

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
  float* colsum;         // backward: [copies, 1024] f32, += column sums of g1 (= grad of b1); may be null
  int copies;
  const __bf16* acc_in;  // backward: [T, 256] added to dX in the epilogue (another gradient of the same x); may be null
};

enum { kFwdTrain = 0, kFwdEval = 1, kBwd = 2, kFwdTrainDrop = 3 };   // kFwdTrain: H stored, p = 0
enum { kFirst = 0, kMid = 1, kMidBeforeLast = 2, kLast = 3 };      // iteration kinds of the skewed tile loop
template <int V> struct IntTag { static constexpr int value = V; };

constexpr int kBufs = 4;           // LDS weight tiles: c - 1 (GEMM-2), c (GEMM-1), c + 1 (prefetch), c + 2 (landing)
constexpr int kWaves = 4;          // 128 tokens per workgroup, one wave per SIMD (the kernel needs ~300 registers)

__device__ __forceinline__ void vm_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void vm_wait_but2() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
__device__ __forceinline__ void vm_wait_but4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
__device__ __forceinline__ void lds_write16(unsigned lds_addr, const bf16x8& v) {
  asm volatile("ds_write_b128 %0, %1\n\ts_nop 2" ::"v"(lds_addr), "v"(v) : "memory");     // as for gstore16_masked
}
__device__ __forceinline__ bf16x8 gload16(const __bf16* p) {
  bf16x8 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// Store under a lane mask without compiler-visible control flow (a branch in the middle of the tile loop would cut
// the scheduling region that interleaves the MFMAs with everything else).  Called with every lane active; only
// s_mov, which leaves SCC alone (the compiler may hold a loop condition there across the asm).
__device__ __forceinline__ void gstore16_masked(__bf16* p, const bf16x8& v, unsigned long long lanes) {
  unsigned long long saved;
  // s_nop: a store of more than 64 bits reads its data registers over several cycles, and the compiler, which pads
  // nothing around asm, may overwrite them in the very next instruction (seen: the last lanes' fourth dword lost).
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\tglobal_store_dwordx4 %1, %2, off\n\ts_nop 2\n\t"
               "s_mov_b64 exec, %0"
               : "=&s"(saved) : "v"(p), "v"(v), "s"(lanes) : "memory");
}

// One workgroup = 4 waves x 32 tokens.  Iteration c of the tile loop runs GEMM-1 of hidden tile c and, behind it,
// GEMM-2 of tile c - 1 with the elementwise step of tile c in the vector slots between those MFMAs (one wave per SIMD:
// nothing else would fill them).  Weight tiles land in LDS two iterations ahead (LDS-DMA); every iteration starts
// with "my own DMA of the previous iteration has landed" + one barrier.  All memory instructions of the loop are
// either builtins with side effects or volatile asm, so they are issued in source order and the counted waits hold.
template <int MODE_>
__global__ __launch_bounds__(kWaves * 64) void ffn_fused_kernel(const FfnArgs a) {
  constexpr int MODE = MODE_ == kFwdTrainDrop ? kFwdTrain : MODE_;
  extern __shared__ __attribute__((aligned(16))) char smem[];       // 4 x 32 KB weight tiles | 4 KB b1 | 4 x 2 KB turn scratch | 4 x 4 KB column sums
  float* const s_b1 = reinterpret_cast<float*>(smem + kBufs * kTileBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const long long tok0 = (long long)blockIdx.x * (kWaves * 32) + wave * 32;
  const long long tok = tok0 + r;
  const bool live = tok < a.T;
  const long long tk = live ? tok : a.T - 1;
  const unsigned lds0 = lds_offset(smem) + lane * 16;

  auto stage = [&](int ht) {
    const char* src = reinterpret_cast<const char*>(a.wp) + (size_t)ht * kTileBytes;
    char* dst = smem + (ht % kBufs) * kTileBytes;
#pragma unroll
    for (int p = 0; p < 32 / kWaves; ++p) {
      const int blk = p * kWaves + wave;                             // 1 KB per wave instruction
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + blk * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*)(dst + blk * 1024), 16, 0, 0);
    }
  };

  stage(0);
  stage(1);
  if (MODE != kBwd)
    for (int i = threadIdx.x; i < kF; i += kWaves * 64) s_b1[i] = (float)a.b1[i];

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
  constexpr bool drop = MODE_ == kFwdTrainDrop;
  if (drop) offset = a.offset + (a.epoch ? *a.epoch : 0ull);
  const long long hrow = tk * kF + 16 * h;                           // + 32 ht: this lane's 16 hidden units of a tile

  // H / g1 tiles ([32 tokens, 32 hidden] per wave and iteration) cross HBM as whole 64-byte rows: in the MFMA layout
  // a lane owns 2 x 16 bytes of ONE token, so a store instruction would touch 64 different rows 16 bytes at a time
  // (request-bound: +34 us per launch measured).  A 2 KB per-wave LDS scratch turns the tile: in "memory order" lane
  // l of instruction i covers bytes 16 (l & 3) .. of token 16 i + (l >> 2), four lanes per row.  Slot rotation
  // (j + (token >> 2)) & 3 keeps both the b128 writes and the b128 reads conflict-free.
  const unsigned scr = lds_offset(smem) + kBufs * kTileBytes + kF * 4 + wave * 2048;
  const unsigned scr_mfma0 = scr + r * 64 + (((2 * h + 0) + (r >> 2)) & 3) * 16;       // chunk 2h of token r
  const unsigned scr_mfma1 = scr + r * 64 + (((2 * h + 1) + (r >> 2)) & 3) * 16;       // chunk 2h + 1
  const int mt = lane >> 2, mj = lane & 3;                           // memory order: token 16 i + mt, chunk mj
  const unsigned scr_mem0 = scr + mt * 64 + ((mj + (mt >> 2)) & 3) * 16;
  const unsigned scr_mem1 = scr + (16 + mt) * 64 + ((mj + ((16 + mt) >> 2)) & 3) * 16;
  const long long mtok0 = tok0 + mt, mtok1 = tok0 + 16 + mt;
  const bool mlive0 = mtok0 < a.T, mlive1 = mtok1 < a.T;
  const unsigned long long mmask0 = __builtin_amdgcn_ballot_w64(mlive0), mmask1 = __builtin_amdgcn_ballot_w64(mlive1);
  const long long mrow0 = (mlive0 ? mtok0 : a.T - 1) * kF + 8 * mj, mrow1 = (mlive1 ? mtok1 : a.T - 1) * kF + 8 * mj;
  const bool wave_full = tok0 + 32 <= a.T;                            // wave-uniform: every store below is issued
  // backward, grad of b1: per-wave row of 1024 column sums (each hidden unit is written once per tile loop)
  const unsigned csrow = lds_offset(smem) + kBufs * kTileBytes + kF * 4 + kWaves * 2048 + wave * (kF * 4) + 16 * h * 4;
  float gs_prev[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) gs_prev[i] = 0.f;
  const float lscale = live ? a.scale : 0.f;

  // backward: H tiles in flight in memory order (tile c is needed at iteration c, loaded at iteration c - 2)
  bf16x8 hraw[3][2];
  if (MODE == kBwd) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      hraw[t][0] = gload16(a.h_in + mrow0 + 32 * t);
      hraw[t][1] = gload16(a.h_in + mrow1 + 32 * t);
    }
  }
  vm_wait_all();
  if (MODE == kBwd) asm volatile("" : "+v"(hraw[0][0]), "+v"(hraw[0][1]), "+v"(hraw[1][0]), "+v"(hraw[1][1]));
  __syncthreads();

  bf16x8 fr[kRing];                                                  // A fragments in flight, slot = position % kRing
#pragma unroll
  for (int i = 0; i < kRing; ++i) fr[i] = frag_read(lds0, i * 1024);
  bf16x8 hp[2];                                                      // processed tile c - 1: B operand of GEMM-2
  hp[0] = hp[1] = bf16x8{};

  auto iteration = [&](auto kind_tag, auto phase_tag, const int c) {
    constexpr int KIND = decltype(kind_tag)::value;
    constexpr int PH = decltype(phase_tag)::value;                   // c % 3 (backward: which hin set holds tile c)
    constexpr bool G1 = KIND != kLast, G2 = KIND != kFirst;
    constexpr bool NEXT_G1 = KIND == kFirst || KIND == kMid;
    constexpr int P_END = (G1 ? 16 : 0) + (G2 ? 16 : 0);

    // ---- everything this wave put in flight one iteration ago has landed; then all waves agree
    // The youngest memory instructions of iteration c - 1 (two H / g1 stores; in the backward also the two H loads
    // issued before them) may stay in flight.  Edge iterations and waves with dead tokens issue fewer: wait for all.
    const bool steady = KIND == kMid && c >= 2 && wave_full;
    if (MODE == kFwdEval || !steady) vm_wait_all();
    else if (MODE == kFwdTrain) vm_wait_but2();
    else vm_wait_but4();
    __builtin_amdgcn_s_barrier();
#ifndef DSKD_FFN_EXPERIMENT_NOSTAGE
    if (c + 2 < kTiles) stage(c + 2);
#endif
    bf16x8 st[2], hin[2];
    if (G2 && MODE != kFwdEval) {                                    // tile c - 1 (written to the scratch last iteration)
      st[0] = frag_read(scr_mem0, 0);
      st[1] = frag_read(scr_mem1, 0);
    }
    if (MODE == kBwd && G1) {                                        // tile c landed two waits ago: memory -> MFMA order
      asm volatile("" : "+v"(hraw[PH][0]), "+v"(hraw[PH][1]));
      lds_write16(scr_mem0, hraw[PH][0]);
      lds_write16(scr_mem1, hraw[PH][1]);
      hin[0] = frag_read(scr_mfma0, 0);
      hin[1] = frag_read(scr_mfma1, 0);
    }
    if (MODE == kBwd && c + 2 < kTiles) {
      hraw[(PH + 2) % 3][0] = gload16(a.h_in + mrow0 + 32 * (c + 2));
      hraw[(PH + 2) % 3][1] = gload16(a.h_in + mrow1 + 32 * (c + 2));
    }

    const unsigned w1 = lds0 + (c % kBufs) * kTileBytes;                        // GEMM-1 fragments of tile c
    const unsigned w2 = lds0 + ((c + kBufs - 1) % kBufs) * kTileBytes + 16384;  // GEMM-2 fragments of tile c - 1
    const unsigned wn = NEXT_G1 ? lds0 + ((c + 1) % kBufs) * kTileBytes : w1 + 16384;   // head of the next iteration
    auto read_ahead = [&](int p) -> bf16x8 {                         // stream position p of this iteration (constant)
#ifdef DSKD_FFN_EXPERIMENT_NOREAD
      return fr[p % kRing];
#endif
      if (p < P_END) return (G1 && p < 16) ? frag_read(w1, p * 1024) : frag_read(w2, (p - (G1 ? 16 : 0)) * 1024);
      return frag_read(wn, (p - P_END) * 1024);
    };

    f32x16 acc;
    f32x4 bias[4];
    u32x4 rnd[2];
    if (G1) {
      if (MODE != kBwd) {
        const unsigned bl = lds_offset(s_b1) + (32 * c + 16 * h) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bias[g]) : "v"(bl), "n"(g * 16));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      unsigned c0[2], c1[2], c2[2], c3[2], k0 = (unsigned)a.seed, k1 = (unsigned)(a.seed >> 32);
      if (drop) {
        const unsigned long long idx = (unsigned long long)((hrow + 32 * c) >> 3);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          c0[q] = (unsigned)(idx + q); c1[q] = (unsigned)((idx + q) >> 32);
          c2[q] = (unsigned)offset; c3[q] = (unsigned)(offset >> 32);
        }
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        frag_wait(fr[s % kRing], kRing - 1);       // (one wait per TWO fragments measured 3-8 % slower: it halves the ring)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s % kRing], xf[s], acc, 0, 0, 0);
        fr[s % kRing] = read_ahead(s + kRing);
        if (MODE == kBwd && G2 && s % 3 == 0 && s < 15) half_wave_sum16_step(gs_prev, s / 3);   // column sums, tile c - 1
        if (drop && s < 10) {                                        // one Philox round per MFMA slot
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0[q];
            const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2[q];
            const unsigned n0 = __builtin_amdgcn_bitop3_b32((unsigned)(p1 >> 32), c1[q], k0, 0x96);   // three-way xor
            const unsigned n2 = __builtin_amdgcn_bitop3_b32((unsigned)(p0 >> 32), c3[q], k1, 0x96);
            c1[q] = (unsigned)p1; c3[q] = (unsigned)p0; c0[q] = n0; c2[q] = n2;
          }
          k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
      }
      if (drop) {
        rnd[0] = u32x4{c0[0], c1[0], c2[0], c3[0]};
        rnd[1] = u32x4{c0[1], c1[1], c2[1], c3[1]};
      }
      if (MODE != kBwd)                                               // the bias reads are older than the ring's 8
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]) : "n"(kRing < 15 ? kRing : 15));
      else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(hin[0]), "+v"(hin[1]) : "n"(kRing < 15 ? kRing : 15));
    }
    if (G2 && MODE != kFwdEval) {                                    // H (forward) / g1 (backward) of tile c - 1, whole rows
      // behind GEMM-1 the two reads are older than the ring's kRing; the last iteration has no GEMM-1 in between
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(st[0]), "+v"(st[1]) : "n"(!G1 ? 0 : kRing < 15 ? kRing : 15));
      gstore16_masked(a.h_out + mrow0 + 32 * (c - 1), st[0], mmask0);
      gstore16_masked(a.h_out + mrow1 + 32 * (c - 1), st[1], mmask1);
    }

    if (MODE == kBwd && G2) {
      if (!G1) {
#pragma unroll
        for (int st5 = 0; st5 < 5; ++st5) half_wave_sum16_step(gs_prev, st5);
      }
      {                                                               // lanes 16 and 48 hold the totals of h = 0 / 1
        unsigned long long saved;
        const f32x4 q0 = {gs_prev[0], gs_prev[1], gs_prev[2], gs_prev[3]}, q1 = {gs_prev[4], gs_prev[5], gs_prev[6], gs_prev[7]};
        const f32x4 q2 = {gs_prev[8], gs_prev[9], gs_prev[10], gs_prev[11]}, q3 = {gs_prev[12], gs_prev[13], gs_prev[14], gs_prev[15]};
        const unsigned dst = csrow + 32 * (c - 1) * 4;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %6\n\t"
                     "ds_write_b128 %1, %2\n\tds_write_b128 %1, %3 offset:16\n\t"
                     "ds_write_b128 %1, %4 offset:32\n\tds_write_b128 %1, %5 offset:48\n\ts_nop 2\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(dst), "v"(q0), "v"(q1), "v"(q2), "v"(q3), "s"(0x0001000000010000ull) : "memory");
      }
    }

    // elementwise step of tile c, one accumulator register at a time
    bf16x8 hc[2];
    auto element = [&](int i) {
      float v = acc[i];
      if (MODE == kBwd) {
        const float hv = (float)hin[i >> 3][i & 7];
        v = hv != 0.f ? v * lscale : 0.f;                             // lscale = 0 in lanes without a token
        gs_prev[i] = v;                                               // read (as tile c - 1) in the next iteration
      } else {
        v = fmaxf(v + bias[i >> 2][i & 3], 0.f);
        if (drop) {
          const unsigned w = rnd[i >> 3][(i & 7) >> 1];
          const unsigned field = (i & 1) ? (w >> 16) : (w & 0xFFFFu);
          v = field < a.thresh16 ? 0.f : v * a.scale;
        }
      }
      hc[i >> 3][i & 7] = (__bf16)v;
    };

    if (G2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int p = (G1 ? 16 : 0) + j;
        frag_wait(fr[p % kRing], (KIND == kLast && P_END - 1 - p < kRing - 1) ? P_END - 1 - p : kRing - 1);
        yacc[j >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[p % kRing], hp[j & 1], yacc[j >> 1], 0, 0, 0);
        if (KIND != kLast || p + kRing < P_END) fr[p % kRing] = read_ahead(p + kRing);
#ifndef DSKD_FFN_EXPERIMENT_NOEPI
        if (G1 && j >= 2 && j < 10) {                                 // GEMM-1's result is ready two MFMAs later
          element(2 * (j - 2));
          element(2 * (j - 2) + 1);
        }
#endif
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) element(i);
    }
#ifdef DSKD_FFN_EXPERIMENT_NOEPI
    if (G1 && G2) { hc[0] = xf[c & 15]; hc[1] = xf[(c + 1) & 15]; }
#endif
    if (G1) {
      hp[0] = hc[0];
      hp[1] = hc[1];
      if (MODE != kFwdEval) {                                         // MFMA -> memory order, read back next iteration
        lds_write16(scr_mfma0, hc[0]);
        lds_write16(scr_mfma1, hc[1]);
      }
    }
  };

  iteration(IntTag<kFirst>{}, IntTag<0>{}, 0);
  for (int c = 1; c + 2 <= kTiles - 2; c += 3) {                      // c = 1 .. 30
    iteration(IntTag<kMid>{}, IntTag<1>{}, c);
    iteration(IntTag<kMid>{}, IntTag<2>{}, c + 1);
    iteration(IntTag<kMid>{}, IntTag<0>{}, c + 2);
  }
  iteration(IntTag<kMidBeforeLast>{}, IntTag<(kTiles - 1) % 3>{}, kTiles - 1);
  iteration(IntTag<kLast>{}, IntTag<kTiles % 3>{}, kTiles);

  if (MODE == kBwd) {                                                 // grad of b1: 4 wave rows -> one atomic per column
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (a.colsum) {
      const float* rows = reinterpret_cast<const float*>(smem + kBufs * kTileBytes + kF * 4 + kWaves * 2048);
      float* dst = a.colsum + (size_t)(blockIdx.x % a.copies) * kF;
#pragma unroll
      for (int q = 0; q < kF / (kWaves * 64); ++q) {
        const int col = q * (kWaves * 64) + threadIdx.x;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) t += rows[w * kF + col];
        atomicAdd(dst + col, t);
      }
    }
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
        else if (a.acc_in) v += (float)a.acc_in[tok * kD + 16 * h + 32 * ot + i];
        o[i >> 3][i & 7] = (__bf16)v;
      }
      *reinterpret_cast<bf16x8*>(orow + 32 * ot) = o[0];
      *reinterpret_cast<bf16x8*>(orow + 32 * ot + 8) = o[1];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// y[T, N] = x[T, 256] W^T (+ bias) for a very tall x and N = 32 .. 512 outputs (the encoder's 256 -> 256 / 384 Linear
// layers and their dX GEMMs): GEMM-1 of the loop above alone.  hipBLASLt runs these at 43 us for [88 892, 256] x
// [256, 256] (2.1 TB/s of the 91 MB it must move); here the wave's X rows sit in registers for all output tiles, the
// packed weight tiles (16 KB each) stream through a three-deep LDS ring and an output tile leaves as two 16-byte
// stores per lane.  Memory-bound; ~150 registers and 49 KB of LDS, so three workgroups share a CU and cover each other's
// load phases (three buffers: tile t in use, t + 1 read ahead, t + 2 landing; 4 buffers / 2 workgroups measured 27.8 vs 24.8 us).
constexpr int kLinTileBytes = 16384;
#ifndef DSKD_LIN_BUFS
#define DSKD_LIN_BUFS 3
#endif
#ifndef DSKD_LIN_OCC
#define DSKD_LIN_OCC 3
#endif
constexpr int kLinBufs = DSKD_LIN_BUFS;

// fragment f (16 bytes) of out tile ot: k-step s, lane (r, h), element j = W[32 ot + pi(r)][128 h + 8 s + j]
// (transposed: W[128 h + 8 s + j][32 ot + pi(r)] -- the dX GEMM reads the same weight the other way round)
__global__ __launch_bounds__(256) void lin256_pack_kernel(const __bf16* __restrict__ W, __bf16* __restrict__ packed, int N,
                                                         int transposed) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= (N / 32) * 1024) return;
  const int ot = f >> 10, q = f & 1023, lane = q & 63, s = q >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int row = 32 * ot + pi_row(r), k0 = 128 * h + 8 * s;
  bf16x8 v;
  if (!transposed) {
    v = *reinterpret_cast<const bf16x8*>(W + (size_t)row * kD + k0);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = W[(size_t)(k0 + j) * N + row];
  }
  *reinterpret_cast<bf16x8*>(packed + (size_t)f * 8) = v;
}

// Many weights in ONE launch: table row e = {source, destination, N, transposed} (device int64 [n, 4]); blockIdx.y = row.
// The student's ~40 tall 256-input Linear weights change once per optimiser step: one launch right after the step's
// low-precision copies are made instead of one pack launch in front of every lin256 call (42 per step, ~4.6 us each,
// most of them on the forward / backward launch chains).
__global__ __launch_bounds__(256) void lin256_pack_many_kernel(const long long* __restrict__ table) {
  const long long* e = table + (size_t)blockIdx.y * 4;
  const __bf16* W = reinterpret_cast<const __bf16*>(e[0]);
  __bf16* packed = reinterpret_cast<__bf16*>(e[1]);
  const int N = (int)e[2], transposed = (int)e[3];
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= (N / 32) * 1024) return;
  const int ot = f >> 10, q = f & 1023, lane = q & 63, s = q >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int row = 32 * ot + pi_row(r), k0 = 128 * h + 8 * s;
  bf16x8 v;
  if (!transposed) {
    v = *reinterpret_cast<const bf16x8*>(W + (size_t)row * kD + k0);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = W[(size_t)(k0 + j) * N + row];
  }
  *reinterpret_cast<bf16x8*>(packed + (size_t)f * 8) = v;
}

struct LinArgs {
  const __bf16* x;       // [T, 256]
  const __bf16* wp;      // packed weight tiles [N / 32][16 KB]
  const __bf16* bias;    // [N] or null
  __bf16* y;             // [T, N]
  long long T;
  int N, relu;
};

#ifndef DSKD_LIN_WAVES
#define DSKD_LIN_WAVES 4
#endif
constexpr int kLinWaves = DSKD_LIN_WAVES;      // waves (of 32 tokens) per workgroup

__global__ __launch_bounds__(kLinWaves * 64, DSKD_LIN_OCC) void lin256_kernel(const LinArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];       // kLinBufs x 16 KB weight tiles | N floats of bias
  float* const s_b = reinterpret_cast<float*>(smem + kLinBufs * kLinTileBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const long long tok0 = (long long)blockIdx.x * (kLinWaves * 32) + wave * 32;
  const long long tok = tok0 + r;
  const bool live = tok < a.T;
  const long long tk = live ? tok : a.T - 1;
  const bool wave_full = tok0 + 32 <= a.T;
  const unsigned long long lmask = __builtin_amdgcn_ballot_w64(live);
  const unsigned lds0 = lds_offset(smem) + lane * 16;
  const int ntiles = a.N >> 5;

  auto stage = [&](int t) {
    const char* src = reinterpret_cast<const char*>(a.wp) + (size_t)t * kLinTileBytes;
    char* dst = smem + (t % kLinBufs) * kLinTileBytes;
#pragma unroll
    for (int p = 0; p < 16 / kLinWaves; ++p) {
      const int blk = p * kLinWaves + wave;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + blk * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*)(dst + blk * 1024), 16, 0, 0);
    }
  };
  stage(0);
  if (ntiles > 1) stage(1);
  for (int i = threadIdx.x; i < a.N; i += kLinWaves * 64) s_b[i] = a.bias ? (float)a.bias[i] : 0.f;
  bf16x8 xf[16];
  {
    const __bf16* xrow = a.x + tk * kD + 128 * h;
#pragma unroll
    for (int s = 0; s < 16; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xrow + 8 * s);
  }
  vm_wait_all();
  __syncthreads();

  bf16x8 fr[kRing];
#pragma unroll
  for (int i = 0; i < kRing; ++i) fr[i] = frag_read(lds0, i * 1024);
  __bf16* const yrow = a.y + tk * a.N + 16 * h;

  for (int t = 0; t < ntiles; ++t) {
    // my DMA of the previous iteration has landed (its two stores may stay in flight), then all waves agree
    if (t >= 1 && wave_full) vm_wait_but2(); else vm_wait_all();
    __builtin_amdgcn_s_barrier();
    if (t + 2 < ntiles) stage(t + 2);
    const unsigned w1 = lds0 + (t % kLinBufs) * kLinTileBytes;
    const unsigned wn = lds0 + ((t + 1) % kLinBufs) * kLinTileBytes;
    const bool more = t + 1 < ntiles;
    f32x4 bias[4];
    const unsigned bl = lds_offset(s_b) + (32 * t + 16 * h) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bias[g]) : "v"(bl), "n"(g * 16));
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      frag_wait(fr[s % kRing], kRing - 1);     // after the last tile the ring is not refilled: fewer reads are younger
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s % kRing], xf[s], acc, 0, 0, 0);
      if (s + kRing < 16) fr[s % kRing] = frag_read(w1, (s + kRing) * 1024);
      else if (more) fr[s % kRing] = frag_read(wn, (s + kRing - 16) * 1024);
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]) : "n"(kRing));
    bf16x8 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = acc[i] + bias[i >> 2][i & 3];
      if (a.relu) v = fmaxf(v, 0.f);
      o[i >> 3][i & 7] = (__bf16)v;
    }
    gstore16_masked(yrow + 32 * t, o[0], lmask);
    gstore16_masked(yrow + 32 * t + 8, o[1], lmask);
  }
}

constexpr size_t kFfnLds = kBufs * kTileBytes + kF * sizeof(float) + kWaves * 2048 + kWaves * kF * sizeof(float);   // 156 KB

// 156 KB of dynamic LDS needs the attribute on every instantiation, on every device the process launches on.
hipError_t ffn_attributes() {
  static bool done[4][64] = {};
  const void* fns[4] = {reinterpret_cast<const void*>(&ffn_fused_kernel<kFwdTrain>),
                        reinterpret_cast<const void*>(&ffn_fused_kernel<kFwdEval>),
                        reinterpret_cast<const void*>(&ffn_fused_kernel<kBwd>),
                        reinterpret_cast<const void*>(&ffn_fused_kernel<kFwdTrainDrop>)};
  for (int i = 0; i < 4; ++i)
    if (!reserve_lds(fns[i], (int)kFfnLds, done[i])) return hipErrorInvalidValue;
  return hipSuccess;
}

template <int MODE>
int launch_ffn(const FfnArgs& a, hipStream_t st) {
  const hipError_t attr = ffn_attributes();
  if (attr != hipSuccess) return fail(DSKD_ERR_LAUNCH, "dskd_ffn: LDS attribute: %s", hipGetErrorString(attr));
  const long long grid = (a.T + kWaves * 32 - 1) / (kWaves * 32);
  hipLaunchKernelGGL((ffn_fused_kernel<MODE>), dim3((unsigned)grid), dim3(kWaves * 64), kFfnLds, st, a);
  return check_launch("dskd_ffn");
}

bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int64_t dskd_ffn_packed_bytes(int d_model, int hidden) {
  if (d_model != kD || hidden != kF) {
    fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_packed_bytes: d_model 256 / hidden 1024 only (got %d / %d)", d_model, hidden);
    return -1;
  }
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
  (void)ffn_attributes();
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
  if (!h_out) return launch_ffn<kFwdEval>(a, (hipStream_t)stream);
  return a.thresh16 ? launch_ffn<kFwdTrainDrop>(a, (hipStream_t)stream) : launch_ffn<kFwdTrain>(a, (hipStream_t)stream);
}

extern "C" int dskd_ffn_bwd(const void* grad_y, const void* h, const void* packed_bwd, void* grad_h, void* grad_x,
                            const void* grad_x_add, float* grad_b1, int copies, int64_t tokens, int d_model, int hidden,
                            float p, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: bf16 only");
  if (d_model != kD || hidden != kF)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: d_model 256 / hidden 1024 only (got %d / %d)", d_model, hidden);
  if (!grad_y || !h || !packed_bwd || !grad_h || !grad_x || tokens < 0)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: null pointer or negative token count");
  if (misaligned(grad_y) || misaligned(h) || misaligned(packed_bwd) || misaligned(grad_h) || misaligned(grad_x))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: pointers must be 16-byte aligned");
  if (!(p >= 0.f) || p >= 1.f) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: p=%f", p);
  if (grad_b1 && copies < 1) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: copies < 1");
  if (tokens == 0) return DSKD_OK;
  FfnArgs a{};
  if (misaligned(grad_x_add)) return fail(DSKD_ERR_INVALID_ARG, "dskd_ffn_bwd: pointers must be 16-byte aligned");
  a.colsum = grad_b1; a.copies = copies; a.acc_in = (const __bf16*)grad_x_add;
  a.in = (const __bf16*)grad_y; a.wp = (const __bf16*)packed_bwd; a.h_in = (const __bf16*)h;
  a.h_out = (__bf16*)grad_h; a.out = (__bf16*)grad_x; a.T = tokens; a.scale = 1.0f / (1.0f - p);
  return launch_ffn<kBwd>(a, (hipStream_t)stream);
}

extern "C" int64_t dskd_lin256_packed_bytes(int N) {
  if (N < 32 || N > 512 || N % 32) {
    fail(DSKD_ERR_INVALID_ARG, "dskd_lin256: N must be a multiple of 32 in [32, 512] (got %d)", N);
    return -1;
  }
  return (int64_t)(N / 32) * kLinTileBytes;
}

extern "C" int dskd_lin256_pack(const void* w, void* packed, int N, int K, int transposed, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_lin256_pack: bf16 only");
  if (K != kD || dskd_lin256_packed_bytes(N) < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_lin256_pack: K must be 256, N a multiple of 32 <= 512 (got K=%d N=%d)", K, N);
  if (!w || !packed || misaligned(w) || misaligned(packed)) return fail(DSKD_ERR_INVALID_ARG, "dskd_lin256_pack: null or misaligned pointer");
  const int frags = (N / 32) * 1024;
  hipLaunchKernelGGL(lin256_pack_kernel, dim3((frags + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const __bf16*)w,
                     (__bf16*)packed, N, transposed);
  return check_launch("dskd_lin256_pack");
}

extern "C" int dskd_lin256_pack_many(const int64_t* table, int n, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_lin256_pack_many: bf16 only");
  if (n < 0 || (n > 0 && !table)) return fail(DSKD_ERR_INVALID_ARG, "dskd_lin256_pack_many: null table");
  if (n == 0) return DSKD_OK;
  if (n > 65535) return fail(DSKD_ERR_INVALID_ARG, "dskd_lin256_pack_many: more than 65535 entries");
  // 64 blocks of 256 fragments cover the largest image (N = 512); smaller ones leave their upper blocks idle
  hipLaunchKernelGGL(lin256_pack_many_kernel, dim3(64, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(table));
  return check_launch("dskd_lin256_pack_many");
}

extern "C" int dskd_lin256_fwd(const void* x, const void* packed, const void* bias, void* y, int64_t tokens, int N, int K,
                               int relu, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_lin256_fwd: bf16 only");
  if (K != kD || dskd_lin256_packed_bytes(N) < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_lin256_fwd: K must be 256, N a multiple of 32 <= 512 (got K=%d N=%d)", K, N);
  if (!x || !packed || !y || tokens < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_lin256_fwd: null pointer or negative token count");
  if (misaligned(x) || misaligned(packed) || misaligned(y)) return fail(DSKD_ERR_INVALID_ARG, "dskd_lin256_fwd: pointers must be 16-byte aligned");
  if (tokens == 0) return DSKD_OK;
  const size_t lds = kLinBufs * kLinTileBytes + (size_t)N * sizeof(float);
  static bool done[64] = {};
  if (!reserve_lds(reinterpret_cast<const void*>(&lin256_kernel), kLinBufs * kLinTileBytes + 512 * 4, done))
    return fail(DSKD_ERR_LAUNCH, "dskd_lin256_fwd: cannot reserve LDS");
  LinArgs a{};
  a.x = (const __bf16*)x; a.wp = (const __bf16*)packed; a.bias = (const __bf16*)bias; a.y = (__bf16*)y;
  a.T = tokens; a.N = N; a.relu = relu;
  const long long grid = (tokens + kLinWaves * 32 - 1) / (kLinWaves * 32);
  hipLaunchKernelGGL(lin256_kernel, dim3((unsigned)grid), dim3(kLinWaves * 64), lds, (hipStream_t)stream, a);
  return check_launch("dskd_lin256_fwd");
}
