// Internal interface between msda.hip (dispatch, gather kernels) and msda_pull.hip (the tiled
// "pull" backward of grad_value).  Not part of the ABI.
#pragma once
#include "common.h"

namespace dskd {

constexpr int kMsdaMaxLevels = 4;

struct MsdaLevels {
  int H[kMsdaMaxLevels];
  int W[kMsdaMaxLevels];
  int start[kMsdaMaxLevels];
};

// Workspace of dskd_msda_bwd_ws: a 64-byte header {count, overflow, ticket} followed by 16-byte
// fallback entries.  The header is zeroed at the start of every call (msda.hip) and is left zero by the apply kernel;
// the tail of the workspace holds the gather kernel's per-region statistics (msda.hip).
constexpr size_t kPullWsHeader = 64;
constexpr size_t kPullWsEntry = 16;

// Can the tiled pull backward take these levels (encoder shape: queries == pixels, 4 levels x 4
// points, level rows contiguous, level 0 the finest)?
bool pull_supported(const MsdaLevels& lg, int levels, int points, int Nv, int Nq, int dtype, int level_mask);

// grad_value rows of the levels in `level_mask` (bit l), written with plain stores (no zeroing
// needed for those rows), followed by the apply kernel for samples that left every tile's
// candidate range.  All launches go to `st`.  Returns DSKD_OK or an error code.
int launch_pull(const float* loc, const float* attn, const void* grad_out, float* grad_value,
                const MsdaLevels& lg, int level_mask, int B, int Nq, int dtype, void* workspace,
                size_t workspace_bytes, hipStream_t st);

// Matrix-core grad_value of the coarse levels (msda_mm.hip; encoder shape, bf16): rows of level 1 (mask bit 1) and / or levels
// 2+3 (mask bit 2) in ONE launch, added with atomics into rows the caller zeroed.  ``stats``: the gather kernel's per-(16 x 16 region, head)
// by-product {max |grad_out|, ...}, written earlier on the same stream; ``sgrid`` = {RX, RY, EX, EY} of that kernel's grid.
bool mm_supported(const MsdaLevels& lg, int levels, int points, int Nv, int Nq, int dtype, int lv0, int nlv);
int launch_bwd_mm(const float* loc, const float* attn, const void* grad_out, float* grad_value, const float* stats,
                  const int* sgrid, const MsdaLevels& lg, int mask, int B, int Nq, int points, hipStream_t st);

}  // namespace dskd
