// DSKD loss 1: between-class semantic distance-matrix distillation, forward value and the
// gradient with respect to the student query embeddings, in three small launches.
//
// Replaces (mmdet/models/dense_heads/gfl_deformable_detr_head_il.py):
//   :525-551  per-query Python loops that accumulate class prototypes
//             corr_student[label][:-1] += hs[idx]; corr_student[label][-1] += 1  (and teacher)
//   :1197-1222 correlation_mat: rows [:L]; divide rows whose TEACHER count is non-zero by the
//             teacher / student counts (yes, :1205 indexes the student rows by the teacher
//             counts -- reproduced); 2*L*L torch.dist launches; MSELoss(mean)/L.
// The reference issues ~2*L^2 tiny kernels forward and as many backward (L = 40 or 70);
// here the whole loss is latency-bound (<0.3 MB touched), so the design goal is launch count
// and determinism: prototypes are summed in ascending query order by one thread per channel
// (the same order as the reference's sequential +=, no atomics), and the loss partials are
// reduced in a fixed order.
//
// Gradient: dL/dD_s[i,j] = -2 (D_t - D_s)[i,j] * w / L^3; dD_s[i,j]/dc_s[i] = (c_s[i]-c_s[j])/D_s[i,j]
// with the 0 sub-gradient at D_s = 0 that torch.dist's backward uses; dc_s[i]/dP_s[i] = 1/num_s[i]
// where the row was normalised (1 otherwise); dP_s[label]/dhs[q] = 1 for the accumulated rows.
#include "common.h"

namespace dskd {
namespace {

// workspace layout (floats): cs[L*D] ct[L*D] gcs[L*D] scale[L] partial[L]
struct CorrWs {
  float* cs;
  float* ct;
  float* gcs;
  float* scale;
  float* partial;
};

__host__ __device__ inline CorrWs carve(void* ws, int L, int D) {
  CorrWs w;
  float* f = (float*)ws;
  w.cs = f;
  w.ct = f + (size_t)L * D;
  w.gcs = f + (size_t)2 * L * D;
  w.scale = f + (size_t)3 * L * D;
  w.partial = w.scale + L;
  return w;
}

// grid = L blocks (class row r), block = 256 threads striding over D channels
__global__ __launch_bounds__(256) void proto_kernel(
    const float* __restrict__ hs_s, const int64_t* __restrict__ labels_s,
    const unsigned char* __restrict__ prev_mask, const float* __restrict__ hs_t,
    const int64_t* __restrict__ keepid_t, const int64_t* __restrict__ labels_t, int N, int D,
    int C, int M, int L, CorrWs ws) {
#pragma clang fp contract(off)
  extern __shared__ int s_idx[];  // matching row indices, student then teacher
  __shared__ int s_ns, s_nt;
  const int r = blockIdx.x;
  const bool is_prev = r < C && prev_mask[r] != 0;
  // Ordered compaction by one wave (ascending index = the reference's loop order).
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int ns = 0;
    if (is_prev) {
      for (int base = 0; base < N; base += 64) {
        const int n = base + lane;
        const bool hit = n < N && labels_s[n] == (int64_t)r;
        const unsigned long long m = __ballot(hit);
        if (hit) s_idx[ns + __popcll(m & ((1ull << lane) - 1ull))] = n;
        ns += __popcll(m);
      }
    }
    int nt = 0;
    for (int base = 0; base < M; base += 64) {
      const int k = base + lane;
      // a keepid outside [0, N) would be an IndexError in the reference; here the detection is ignored (never read
      // out of bounds: a fault can take the whole node down)
      const bool hit = k < M && labels_t[k] == (int64_t)r && (unsigned long long)keepid_t[k] < (unsigned long long)N;
      const unsigned long long m = __ballot(hit);
      if (hit) s_idx[N + nt + __popcll(m & ((1ull << lane) - 1ull))] = (int)keepid_t[k];
      nt += __popcll(m);
    }
    if (lane == 0) { s_ns = ns; s_nt = nt; }
  }
  __syncthreads();
  const int ns = s_ns, nt = s_nt;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float as = 0.f, at = 0.f;
    for (int k = 0; k < ns; ++k) as += hs_s[(size_t)s_idx[k] * D + c];
    for (int k = 0; k < nt; ++k) at += hs_t[(size_t)s_idx[N + k] * D + c];
    if (nt != 0) {  // rows selected by the TEACHER count on both sides
      at = at / (float)nt;
      as = as / (float)ns;
    }
    ws.cs[(size_t)r * D + c] = as;
    ws.ct[(size_t)r * D + c] = at;
  }
  if (threadIdx.x == 0) ws.scale[r] = nt != 0 ? 1.f / (float)ns : 1.f;
}

// grid = L blocks (row i), block = 256 threads (4 waves)
__global__ __launch_bounds__(256) void pairdist_kernel(int L, int D, float coef_scale, CorrWs ws) {
  extern __shared__ float s_k[];  // [L] gradient coefficient per j, then [4] wave partials
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* csi = ws.cs + (size_t)i * D;
  const float* cti = ws.ct + (size_t)i * D;
  float lsum = 0.f;
  for (int j = wave; j < L; j += 4) {
    const float* csj = ws.cs + (size_t)j * D;
    const float* ctj = ws.ct + (size_t)j * D;
    float ss = 0.f, st = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float ds = csi[c] - csj[c];
      const float dt = cti[c] - ctj[c];
      ss = fmaf(ds, ds, ss);
      st = fmaf(dt, dt, st);
    }
    ss = wave_sum(ss);
    st = wave_sum(st);
    const float Ds = sqrtf(ss), Dt = sqrtf(st);
    const float e = Dt - Ds;
    lsum += e * e;
    // (i,j) and (j,i) both pull on c_s[i]: factor 2 on top of dMSE/dD_s = -2 e
    const float k = Ds > 0.f ? (-4.f * e * coef_scale) / Ds : 0.f;
    if (lane == 0) s_k[j] = k;
  }
  if (lane == 0) s_k[L + wave] = lsum;
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const float ci = csi[c];
    float g = 0.f;
    for (int j = 0; j < L; ++j) g = fmaf(s_k[j], ci - ws.cs[(size_t)j * D + c], g);
    ws.gcs[(size_t)i * D + c] = g;
  }
  if (threadIdx.x == 0) ws.partial[i] = (s_k[L] + s_k[L + 1]) + (s_k[L + 2] + s_k[L + 3]);
}

// grid over the N query rows; block 0 also reduces the loss partials in a fixed order
__global__ __launch_bounds__(256) void corr_scatter_kernel(
    const int64_t* __restrict__ labels_s, const unsigned char* __restrict__ prev_mask, int N,
    int D, int C, int L, float loss_scale, CorrWs ws, float* __restrict__ loss,
    float* __restrict__ grad_hs) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < L; ++i) s += ws.partial[i];
    loss[0] = s * loss_scale;
  }
  const int rows_per_block = blockDim.x / 64;
  const int n = blockIdx.x * rows_per_block + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  const int64_t lab = labels_s[n];
  const bool on = lab >= 0 && lab < L && lab < C && prev_mask[lab] != 0;
  const float sc = on ? ws.scale[lab] : 0.f;
  for (int c = lane; c < D; c += 64)
    grad_hs[(size_t)n * D + c] = on ? ws.gcs[(size_t)lab * D + c] * sc : 0.f;
}

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int64_t dskd_proto_corr_workspace(int L, int D) {
  if (L < 0 || D < 0) return 0;
  return (int64_t)sizeof(float) * ((int64_t)3 * L * D + 2 * L + 16);
}

extern "C" int dskd_proto_corr_fwd(const float* hs_s, const int64_t* labels_s,
                                   const uint8_t* prev_mask, const float* hs_t,
                                   const int64_t* keepid_t, const int64_t* labels_t, int N,
                                   int D, int C, int M, int L, float loss_weight,
                                   float* loss, float* grad_hs_s, void* workspace,
                                   void* stream) {
  if (N <= 0 || D <= 0 || C <= 0 || M < 0 || L <= 0 || L > C)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_proto_corr_fwd: bad sizes N=%d D=%d C=%d M=%d L=%d", N, D, C, M, L);
  if (!hs_s || !labels_s || !prev_mask || !hs_t || !loss || !grad_hs_s || !workspace ||
      (M > 0 && (!keepid_t || !labels_t)))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_proto_corr_fwd: null pointer");
  const size_t lds1 = sizeof(int) * (size_t)(N + M);
  if (lds1 > 150 * 1024) return fail(DSKD_ERR_INVALID_ARG, "dskd_proto_corr_fwd: N+M=%d too large", N + M);
  hipStream_t st = (hipStream_t)stream;
  const CorrWs ws = carve(workspace, L, D);
  hipLaunchKernelGGL(proto_kernel, dim3(L), dim3(256), lds1, st, hs_s, labels_s, prev_mask, hs_t,
                     keepid_t, labels_t, N, D, C, M, L, ws);
  if (int rc = check_launch("dskd_proto_corr_fwd/proto")) return rc;
  // loss = w * sum(e^2) / L^2 / L ; dL/dD_s = -2 e * w / L^3
  const float inv = loss_weight / ((float)L * (float)L * (float)L);
  hipLaunchKernelGGL(pairdist_kernel, dim3(L), dim3(256), sizeof(float) * (L + 4), st, L, D, inv, ws);
  if (int rc = check_launch("dskd_proto_corr_fwd/pairdist")) return rc;
  hipLaunchKernelGGL(corr_scatter_kernel, dim3((N + 3) / 4), dim3(256), 0, st, labels_s, prev_mask,
                     N, D, C, L, inv, ws, loss, grad_hs_s);
  return check_launch("dskd_proto_corr_fwd/scatter");
}
