// CRD memory-bank contrastive loss (DC-Distill): reference CL_utils/memory_new.py:249-397
// (ContrastMemory_v3.forward) + CL_utils/CRD_loss.py:221-244 (ContrastLoss_v2), fused so that nothing
// [B, P+K, 128]-sized is ever written:
//   crd_score   : one coalesced 512-B read per bank row, 4 dot products + 2 norms in registers,
//                 half-wave (32-lane) reductions -> exp scores (both directions) + cosine discrepancy
//   crd_select  : per-sample rank-by-counting in LDS (replaces two full torch.sort calls): P2 positives at the
//                 requested ranks of (student-sim - teacher-sim) descending, slot 0 forced to the exact
//                 positive; K2 negatives with the smallest discrepancy (ascending order kept)
//   crd_zsum/crd_setz : first-call normalisation constants Z (memory_new.py:368-375)
//   crd_loss_grad: NCE loss + analytic d loss / d (v1, v2) as a weighted gather-sum of the *pre-update*
//                 bank rows (the bank is momentum-updated right after, as in the reference)
//   crd_update  : momentum update + re-normalisation of the rows mem[y] (memory_new.py:382-395)
// All bank traffic is row-granular 512-B coalesced reads; the bank (n_data x 128 fp32 x 2) is L2/MALL
// resident for the reference's dataset sizes.
#include <mutex>
#include "ph_common.h"
#include "ph_dense.h"
#include "ph_kernels.h"

namespace {

constexpr int D = 128;   // feat_dim (options.py:83); asserted by the launcher

__device__ __forceinline__ float half_sum(float v) {   // reduce within each 32-lane half of the wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// grid (ceil(PK/64), B), block 256 = 8 half-waves, each half-wave walks 8 of the block's 64 columns
__global__ __launch_bounds__(256) void crd_score_kernel(const float* __restrict__ v1, const float* __restrict__ v2,
                                                        const int64_t* __restrict__ idx,
                                                        const int64_t* __restrict__ idx_b2,
                                                        const float* __restrict__ mem1,
                                                        const float* __restrict__ mem2, float* __restrict__ out1,
                                                        float* __restrict__ out2, float* __restrict__ diff, int PK,
                                                        float invT) {
  const int b = blockIdx.y;
  const int hw = threadIdx.x >> 5, l = threadIdx.x & 31;
  const f32x4 a1 = *reinterpret_cast<const f32x4*>(v1 + (size_t)b * D + l * 4);
  const f32x4 a2 = *reinterpret_cast<const f32x4*>(v2 + (size_t)b * D + l * 4);
  const float n1 = sqrtf(half_sum(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2] + a1[3] * a1[3]));
  const float n2 = sqrtf(half_sum(a2[0] * a2[0] + a2[1] * a2[1] + a2[2] * a2[2] + a2[3] * a2[3]));
  const int j0 = blockIdx.x * 64 + hw * 8;
#pragma unroll 4
  for (int jj = 0; jj < 8; ++jj) {
    const int j = j0 + jj;
    if (j >= PK) break;
    const int64_t row = idx[(size_t)b * PK + j];
    const int64_t row2 = idx_b2[(size_t)b * PK + j];   // MIA-2023 v10: the two banks have their own KNN positives
    const f32x4 m1 = *reinterpret_cast<const f32x4*>(mem1 + row * D + l * 4);
    const f32x4 m2 = *reinterpret_cast<const f32x4*>(mem2 + row2 * D + l * 4);
    float d12 = 0.f, d21 = 0.f, d11 = 0.f, d22 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      d12 += m1[k] * a2[k];   // memory_v1 . v2  -> out_v2   (:270-273)
      d21 += m2[k] * a1[k];   // memory_v2 . v1  -> out_v1   (:275-278)
      d11 += m1[k] * a1[k];   // "t_relation" = student bank x student query (:288-289, names swapped)
      d22 += m2[k] * a2[k];   // "s_relation" = teacher bank x teacher query (:291-292)
      q1 += m1[k] * m1[k];
      q2 += m2[k] * m2[k];
    }
    d12 = half_sum(d12); d21 = half_sum(d21); d11 = half_sum(d11); d22 = half_sum(d22);
    q1 = half_sum(q1); q2 = half_sum(q2);
    if (l == 0) {
      out2[(size_t)b * PK + j] = expf(d12 * invT);
      out1[(size_t)b * PK + j] = expf(d21 * invT);
      diff[(size_t)b * PK + j] = d11 / (sqrtf(q1) * n1) - d22 / (sqrtf(q2) * n2);
    }
  }
}

// one block per sample.  sel[b][0..P2) = positive columns, sel[b][P2..P2+K2) = negative columns (absolute
// column numbers in [0, P+K)); xs/xt = gathered raw exp scores.
__global__ __launch_bounds__(1024) void crd_select_kernel(const float* __restrict__ diff,
                                                         const float* __restrict__ out1,
                                                         const float* __restrict__ out2,
                                                         const int* __restrict__ ranks, int* __restrict__ sel,
                                                         float* __restrict__ xs, float* __restrict__ xt, int P, int K,
                                                         int P2, int K2, int select_neg, int select_pos) {
  extern __shared__ __attribute__((aligned(16))) float sd[];   // [P+K] discrepancies, then int rank_to_col[P]
  const int b = blockIdx.x, PK = P + K, S2 = P2 + K2;
  if (!select_pos && !select_neg) {      // every column in order (vanilla / v10 banks): a copy, no LDS - any list length
    for (int t = threadIdx.x; t < S2; t += blockDim.x) {
      const int col = t < P2 ? t : P + (t - P2);
      sel[(size_t)b * S2 + t] = col;
      xs[(size_t)b * S2 + t] = out1[(size_t)b * PK + col];
      xt[(size_t)b * S2 + t] = out2[(size_t)b * PK + col];
    }
    return;
  }
  int* r2c = reinterpret_cast<int*>(sd + PK);
  for (int i = threadIdx.x; i < PK; i += blockDim.x) sd[i] = diff[(size_t)b * PK + i];
  __syncthreads();
  // positives: rank in DEscending order of diff[0..P)  (memory_new.py:303)
  // (rank by counting; the list is read from LDS four values at a time - one ds_read_b128 per four comparisons)
  const int P4 = P & ~3;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    const float v = sd[i];
    int r = 0;
    for (int q = 0; q < P4; q += 4) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(sd + q);
      r += (w[0] > v) || (w[0] == v && q < i);
      r += (w[1] > v) || (w[1] == v && q + 1 < i);
      r += (w[2] > v) || (w[2] == v && q + 2 < i);
      r += (w[3] > v) || (w[3] == v && q + 3 < i);
    }
    for (int q = P4; q < P; ++q) { const float w = sd[q]; r += (w > v) || (w == v && q < i); }
    r2c[r] = i;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < P2; t += blockDim.x) {
    int col = r2c[ranks ? ranks[t] : t];   // "hard": ranks 0..P2-1 (:308); "mid"/"random": host-drawn ranks (:311-318)
    if (t == 0) col = 0;                    // slot 0 := exact positive (:325)
    if (!select_pos) col = t;               // vanilla / v10 banks: every positive column, in order
    sel[(size_t)b * S2 + t] = col;
    xs[(size_t)b * S2 + t] = out1[(size_t)b * PK + col];
    xt[(size_t)b * S2 + t] = out2[(size_t)b * PK + col];
  }
  // negatives: the K2 smallest of diff[P..P+K) in AScending order (:342-345)
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    int r;
    if (select_neg) {
      const float v = sd[P + i];
      r = 0;
      const int K4 = (P & 3) ? 0 : (K & ~3);   // 16-byte aligned only when P is a multiple of 4
      for (int q = 0; q < K4; q += 4) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sd + P + q);
        r += (w[0] < v) || (w[0] == v && q < i);
        r += (w[1] < v) || (w[1] == v && q + 1 < i);
        r += (w[2] < v) || (w[2] == v && q + 2 < i);
        r += (w[3] < v) || (w[3] == v && q + 3 < i);
      }
      for (int q = K4; q < K; ++q) { const float w = sd[P + q]; r += (w < v) || (w == v && q < i); }
    } else {
      r = i;
    }
    if (r < K2) {
      sel[(size_t)b * S2 + P2 + r] = P + i;
      xs[(size_t)b * S2 + P2 + r] = out1[(size_t)b * PK + P + i];
      xt[(size_t)b * S2 + P2 + r] = out2[(size_t)b * PK + P + i];
    }
  }
}

// sums[0] = sum(xs), sums[1] = sum(xt) over n elements (single block, deterministic)
__global__ __launch_bounds__(1024) void crd_zsum_kernel(const float* __restrict__ xs, const float* __restrict__ xt,
                                                        float* sums, int n) {
  __shared__ double sh[2][16];
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { s1 += xs[i]; s2 += xt[i]; }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, c = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a += sh[0][i]; c += sh[1][i]; }
    sums[0] = (float)a; sums[1] = (float)c;
  }
}
// params = [K, T, Z_v1, Z_v2, momentum, P] (memory_new.py:244); Z set only while negative (:368-375)
__global__ void crd_setz_kernel(float* params, const float* sums, float count, float n_data) {
  if (threadIdx.x == 0) {
    if (params[2] < 0.f) params[2] = sums[0] / count * n_data;
    if (params[3] < 0.f) params[3] = sums[1] / count * n_data;
  }
}

// one block per sample: NCE loss terms + coefficient-weighted gather-sum of bank rows
//   loss_s = -(1/Bn) [ sum_p log(x/(x+c)) / P2 + sum_n log(mPn/(x+c)) ],  x = xs/Z_v1, c = K2/n_data + eps
//   d loss_s / d dot_j = -(1/(Bn P2)) (c/(x+c))/T  (positives) ;  +(1/Bn) (x/(x+c))/T  (negatives)
//   dv1[b] = sum_j coef_s[j] mem2[idx[b][sel_j]] ; dv2[b] = sum_j coef_t[j] mem1[idx[b][sel_j]]
__global__ __launch_bounds__(1024) void crd_loss_grad_kernel(const float* __restrict__ xs,
                                                            const float* __restrict__ xt,
                                                            const int* __restrict__ sel,
                                                            const int64_t* __restrict__ idx,
                                                            const int64_t* __restrict__ idx_b2,
                                                            const float* __restrict__ posw_s,
                                                            const float* __restrict__ posw_t,
                                                            const float* __restrict__ mem1,
                                                            const float* __restrict__ mem2,
                                                            const float* __restrict__ params,
                                                            float* __restrict__ lossp, float* __restrict__ dv1,
                                                            float* __restrict__ dv2, int PK, int P2, int K2, int m_neg,
                                                            float n_data, float inv_bnorm, float* __restrict__ part) {
  // gridDim.y > 1 (long column lists, e.g. nce_k = 4096): the columns of a sample are dealt to gridDim.y workgroups whose
  // partial sums go to `part` and are added in a fixed order by crd_loss_grad_reduce_kernel
  const int b = blockIdx.x, S2 = P2 + K2;
  const int hw = threadIdx.x >> 5, l = threadIdx.x & 31;
  const float invT = 1.f / params[1], Z1 = params[2], Z2 = params[3];
  const float mPn = (float)m_neg / n_data, c = mPn + 1e-7f;      // m = the number of negatives of the loss (= K2 unless they are scanned)
  f32x4 g1 = {0.f, 0.f, 0.f, 0.f}, g2 = {0.f, 0.f, 0.f, 0.f};
  float ls = 0.f;
  // 32 half-waves per sample: the loop is a chain of dependent gathers (sel -> idx -> bank row), latency-bound; with 8
  // half-waves (256 threads) it took 54 us at B = 64, P2 + K2 = 532
  constexpr int NHW = 32;
  for (int j = blockIdx.y * NHW + hw; j < S2; j += NHW * gridDim.y) {
    const float x1 = xs[(size_t)b * S2 + j] / Z1, x2 = xt[(size_t)b * S2 + j] / Z2;
    float c1, c2;
    if (j < P2) {
      // weight of positive j: 1/P2 (MICCAI / v3) or similarity_j / sum_p similarity_p (MIA-2023 ContrastLoss_v2)
      const float w1 = posw_s ? posw_s[(size_t)b * P2 + j] : 1.f / (float)P2;
      const float w2 = posw_t ? posw_t[(size_t)b * P2 + j] : 1.f / (float)P2;
      ls += logf(x1 / (x1 + c)) * w1 + logf(x2 / (x2 + c)) * w2;
      c1 = -(c / (x1 + c)) * invT * inv_bnorm * w1;
      c2 = -(c / (x2 + c)) * invT * inv_bnorm * w2;
    } else {
      ls += logf(mPn / (x1 + c)) + logf(mPn / (x2 + c));
      c1 = (x1 / (x1 + c)) * invT * inv_bnorm;
      c2 = (x2 / (x2 + c)) * invT * inv_bnorm;
    }
    const int col = sel[(size_t)b * S2 + j];
    const int64_t row = idx[(size_t)b * PK + col], row2 = idx_b2[(size_t)b * PK + col];
    const f32x4 m2 = *reinterpret_cast<const f32x4*>(mem2 + row2 * D + l * 4);
    const f32x4 m1 = *reinterpret_cast<const f32x4*>(mem1 + row * D + l * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { g1[k] += c1 * m2[k]; g2[k] += c2 * m1[k]; }
  }
  __shared__ float sh[NHW][2][D];
  __shared__ float shl[NHW];
#pragma unroll
  for (int k = 0; k < 4; ++k) { sh[hw][0][l * 4 + k] = g1[k]; sh[hw][1][l * 4 + k] = g2[k]; }
  if (l == 0) shl[hw] = ls;
  __syncthreads();
  if (threadIdx.x < 2 * D) {
    const int which = threadIdx.x >> 7, d = threadIdx.x & 127;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < NHW; ++q) t += sh[q][which][d];
    if (gridDim.y == 1) (which ? dv2 : dv1)[(size_t)b * D + d] = t;
    else part[(((size_t)b * gridDim.y + blockIdx.y) * 2 + which) * D + d] = t;
  }
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int q = 0; q < NHW; ++q) t += shl[q];
    if (gridDim.y == 1) lossp[b] = -t * inv_bnorm;
    else part[(size_t)gridDim.x * gridDim.y * 2 * D + (size_t)b * gridDim.y + blockIdx.y] = t;
  }
}

__global__ __launch_bounds__(256) void crd_loss_grad_reduce_kernel(const float* __restrict__ part, float* __restrict__ lossp,
                                                                   float* __restrict__ dv1, float* __restrict__ dv2, int NS,
                                                                   float inv_bnorm) {
  const int b = blockIdx.x, which = threadIdx.x >> 7, d = threadIdx.x & 127;
  float t = 0.f;
  for (int y = 0; y < NS; ++y) t += part[(((size_t)b * NS + y) * 2 + which) * D + d];
  (which ? dv2 : dv1)[(size_t)b * D + d] = t;
  if (threadIdx.x == 0) {
    float l = 0.f;
    for (int y = 0; y < NS; ++y) l += part[(size_t)gridDim.x * NS * 2 * D + (size_t)b * NS + y];
    lossp[b] = -l * inv_bnorm;
  }
}

// mem[y[b]] = normalize(momentum * mem[y[b]] + (1 - momentum) * v[b]); one wave per (sample, bank)
__global__ __launch_bounds__(256) void crd_update_kernel(float* __restrict__ mem1, float* __restrict__ mem2,
                                                         const float* __restrict__ v1, const float* __restrict__ v2,
                                                         const int64_t* __restrict__ y, const float* params, int B) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= 2 * B) return;
  const int b = w >> 1;
  float* mem = (w & 1) ? mem2 : mem1;
  const float* v = (w & 1) ? v2 : v1;
  const float mom = params[4];
  const int64_t row = y[b];
  float a0 = mem[row * D + lane] * mom + v[(size_t)b * D + lane] * (1.f - mom);
  float a1 = mem[row * D + 64 + lane] * mom + v[(size_t)b * D + 64 + lane] * (1.f - mom);
  const float n = sqrtf(wave_sum(a0 * a0 + a1 * a1));
  mem[row * D + lane] = a0 / n;
  mem[row * D + 64 + lane] = a1 / n;
}

// MIA-2023 v10 positives (CL_utils/CRD_criterion_v10.py:72-79,110-116): for each query the num_pos bank rows of
// the query's class with the largest cosine similarity to the query's OWN bank row.  One block per (query, bank);
// every thread scans a strided slice keeping a private top-NP list, the lists are merged through LDS.  The
// reference copies the whole bank to the host and calls sklearn per step; here the bank is read once from L2/HBM.
constexpr int TOPK_MAX = 8;

// A similarity and its bank row as one unsigned key: larger key = earlier in torch.sort(descending) with ties broken by the
// lower row (value bits made monotone, -0 folded onto +0; low word = ~row).  0 = an empty list slot.
using u64 = unsigned long long;
__device__ __forceinline__ u64 knn_key(float v, int row) {
  unsigned u = __float_as_uint(v + 0.f);
  u ^= (u & 0x80000000u) ? 0xffffffffu : 0x80000000u;
  return ((u64)u << 32) | (unsigned)(~row);
}
__device__ __forceinline__ float knn_value(u64 key) {
  const unsigned u = (unsigned)(key >> 32);
  return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}
// insert into a descending list of TOPK_MAX keys held in registers (the caller has checked c > L[TOPK_MAX - 1])
__device__ __forceinline__ void knn_insert(u64 (&L)[TOPK_MAX], u64 c) {
#pragma unroll
  for (int k = 0; k < TOPK_MAX; ++k) {
    const bool g = c > L[k];
    const u64 hi = g ? c : L[k];
    c = g ? L[k] : c;
    L[k] = hi;
  }
}

// stages 1-4: class-masked cosine similarity of EVERY bank row with EVERY query (CRD_criterion_v10.py:72-79: the reference's
// cosine_similarity(bank[idx], bank) * class_mask) and the selection of the best rows without the similarity matrix ever
// existing in memory.  The product is a [n_data x 128] x [128 x B] GEMM per bank on the EXACT fp32 matrix instruction
// (v_mfma_f32_32x32x2_f32: bit for bit a k-ordered fmaf chain).  A wave owns tiles of 32 bank rows; bank rows are the M side and
// the queries the N side, so a lane's 16 accumulator registers are 16 bank rows of ONE query per 32-query block and the lane can
// select for that query in registers.  The rows arrive by LDS-DMA, a quarter tile (32 rows x 32 features) per group, two groups
// in flight behind the one being multiplied (three 4-KiB buffers per wave, hand-counted vmcnt); the 16-byte chunks are
// XOR-swizzled on the source side; the row norms are accumulated from the matrix operands themselves; the queries (the bank
// rows idx[b][0]) are gathered into LDS by every workgroup.  Eight waves of 147 registers, two per SIMD.
//   Keeping a sorted list per lane costs ~500 cycles per (row, query-block) step whenever ANY of the 64 lanes inserts, and with
// 2048 waves a list sees 32 rows - every step inserted (57 us for the pass, 4x its matrix time).  So the selection is seeded:
//   1. SAMPLE pass: every 16th tile, one per wave, one 32-query block per workgroup; a lane only keeps the MAXIMUM key of its 16
//      rows - one compare per element - and leaves it in gmax[bank][query][group].  The groups are disjoint row sets, so the
//      TOPK_MAX-th largest of a query's group maxima (stage 2, `thr`) is a lower bound of its TOPK_MAX-th best similarity over
//      the whole bank.
//   3. FULL pass: an element enters a lane's list only if its key reaches thr - about TOPK_MAX x 16 elements per query in
//      the whole bank, so the insertion path is skipped by almost every step: a step is one multiply-compare on the raw
//      accumulator (conservative by 2^-20, the exact key is formed inside the rare path).  Each wave leaves one list per query.
//   4. merge: the NP best keys of a query's lists (mostly empty: key plane 0 is read coalesced, plane k only behind a
//      non-empty plane k - 1).
// The result does not depend on the sample (any thr that is a true lower bound gives the same NP keys); a bank sorted by
// class only makes the full pass slower.  Measured at 65 536 rows x 64 queries (rocprofv3, tests/bench_topk_gpu.py): sample 9.5 +
// thr 5.1 + full 35 + merge 9.6 = 59 us.  The full pass runs at matrix time + selection time (3.9 + 2.3 us per tile and
// SIMD): the VALU work of one wave does not hide under the other wave's fp32 MFMAs (delaying one wave of every SIMD by half a
// matrix phase changed nothing), and the per-launch overhead of the four small launches is a third of the total.
// History: one workgroup per (query, bank, slice) re-read the bank per query (1.4 GB through L2, 278 us); a thread per bank
// row with the queries as scalar operands 120-132 us; similarity matrix [2][B][n_data] written by an MFMA kernel (59 us) and
// re-read by a selection kernel (41 us) + merge (9 us): 136 MB of traffic for 67 MB of bank, 157-162 us per call (round 3);
// one unseeded pass with lists 57 + 22 us; seeded, rows through registers and a half-tile LDS stage 16 + 5 + 38 + 10 us.
constexpr int SIM_RS = 129;     // dwords per staged query row (odd: the 32 lanes of a ds_read_b32 group hit 32 different banks)
constexpr int KNN_WAVES = 8, KNN_SAMPLE_WAVES = 4, KNN_NBUF = 3, KNN_MAX_GX = 128, KNN_SAMPLE_TILES = 128, KNN_MAX_B = 64;

// DPP helpers (no LDS round trip, unlike __shfl): x of the lane N to the right inside the 16-lane row / lane 15 or 31 broadcast
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned dpp_u32(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ u64 dpp_max_u64(u64 v) {
  const u64 o = ((u64)dpp_u32<CTRL, ROW_MASK>((unsigned)(v >> 32)) << 32) | dpp_u32<CTRL, ROW_MASK>((unsigned)v);   // (0 where no source)
  return o > v ? o : v;
}
// largest key of the wave, in every lane (uniform)
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
  v = dpp_max_u64<0x111>(v);           // row_shr:1
  v = dpp_max_u64<0x112>(v);           // row_shr:2
  v = dpp_max_u64<0x114>(v);           // row_shr:4
  v = dpp_max_u64<0x118>(v);           // row_shr:8   -> lane 15 of every row holds the row's maximum
  v = dpp_max_u64<0x142, 0xa>(v);      // row_bcast:15 into rows 1, 3
  v = dpp_max_u64<0x143, 0xc>(v);      // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave's maximum
  return ((u64)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// sum over the 16-lane row, valid in lane 15 of the row
__device__ __forceinline__ float row16_sum_hi(float x) {
  auto sh = [](float y, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  x += sh(x, std::integral_constant<int, 0x111>{});
  x += sh(x, std::integral_constant<int, 0x112>{});
  x += sh(x, std::integral_constant<int, 0x114>{});
  x += sh(x, std::integral_constant<int, 0x118>{});
  return x;
}

// one LDS-DMA wave-instruction: lane l copies 16 (4) bytes from its global address to LDS byte lds_addr + 16 (4) * l
__device__ __forceinline__ void knn_dma16(const void* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(g) : "memory");
}
__device__ __forceinline__ void knn_dma4(const void* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" : : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(g) : "memory");
}

template <int NQ, bool SAMPLE>
__global__ __launch_bounds__((SAMPLE ? KNN_SAMPLE_WAVES : KNN_WAVES) * 64) void crd_bank_knn_kernel(
    const float* __restrict__ mem1, const float* __restrict__ mem2, const int* __restrict__ labels, const int64_t* __restrict__ idx,
    int PK, const int64_t* __restrict__ batch_label, int B, int n_data, int stiles, int tstride, u64* __restrict__ gmax,
    const u64* __restrict__ thr, u64* __restrict__ cand) {
  constexpr int NW = SAMPLE ? KNN_SAMPLE_WAVES : KNN_WAVES;
  typedef __attribute__((address_space(3))) unsigned char lds_uchar;
  extern __shared__ __attribute__((aligned(16))) float sim_lds[];
  float* Qs = sim_lds;                                   // [NQ * 32][SIM_RS]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int RQ0 = NQ * 32 * SIM_RS, RQW = KNN_NBUF * 1024, AUX0 = RQ0 + NW * RQW, AUXW = 32 + 2 * 64;
  float* Rq = sim_lds + RQ0 + wv * RQW;                  // this wave's KNN_NBUF quarter tiles [32 rows][32 features], 16-B chunks swizzled
  float* rnl = sim_lds + AUX0 + wv * AUXW;               // its 32 row norms | 2 x 64 row labels (tile parity; a 64-lane DMA each)
  int* rli = reinterpret_cast<int*>(rnl) + 32;
  float* qpart = sim_lds + AUX0 + NW * AUXW;             // [NQ * 32][2] halves of |q|^2
  const unsigned lds0 = (unsigned)(size_t)(lds_uchar*)sim_lds;
  const unsigned rq_lds = lds0 + (RQ0 + wv * RQW) * 4, rli_lds = lds0 + (AUX0 + wv * AUXW + 32) * 4;
  const int bank = blockIdx.y;
  const int q0 = SAMPLE ? (int)blockIdx.z * 32 : 0;      // the sample pass: one 32-query block per workgroup (grid z)
  const float* mem = bank ? mem2 : mem1;
  const int ntiles = (n_data + 31) / 32;
  const int stride = SAMPLE ? ntiles : gridDim.x * NW;      // (a sample wave has one tile)
  // Quarter group (tile, q) = 4 DMA instructions of 8 rows x 128 B (+ the 32 row labels in front of quarter 0): lane l carries
  // chunk position l & 7 of row 8e + (l >> 3); the LDS image is linear, the XOR swizzle of the 16-byte chunks is applied to the
  // SOURCE (chunk p of row r holds features 4 (p ^ ((r >> 1) & 7)) ..): the 32 lanes of a matrix operand read hit 16 banks
  const int drow = lane >> 3, dchunk = ((lane & 7) ^ ((drow >> 1) & 3)) * 4;      // ((8e + drow) >> 1) & 7 = 4 (e & 1) + (drow >> 1)
  auto issue = [&](int tile, int q, int buf, int par) {
    tile = tile < ntiles ? tile : ntiles - 1;              // (past the end: harmless duplicates, the wait counts stay uniform)
    if (q == 0) {
      const int lrow = tile * 32 + (lane & 31);
      knn_dma4(labels + (lrow < n_data ? lrow : 0), rli_lds + par * 256);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = tile * 32 + e * 8 + drow;
      knn_dma16(mem + (size_t)(row < n_data ? row : 0) * D + q * 32 + (dchunk ^ ((e & 1) << 4)), rq_lds + buf * 4096 + e * 1024);
    }
  };
  const int slot = blockIdx.x * NW + wv;
  int t = SAMPLE ? (slot < stiles ? slot * tstride : ntiles) : slot;
  if (t < ntiles) {      // (in flight under the staging of the queries)
    issue(t, 0, 0, 0);
    issue(t, 1, 1, 0);
  }
  {   // the queries = the bank rows of the samples themselves, idx[b][0]: NQ * 32 rows of 32 16-byte pieces, all loads of a
      // thread in flight together; a half-wave holds one row, its |q|^2 comes from the two 16-lane DPP rows
    constexpr int NE = NQ * 32 * 32 / (NW * 64);
    f32x4 qv[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int pc = e * (NW * 64) + threadIdx.x, qi = q0 + (pc >> 5);
      qv[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (qi < B) qv[e] = *reinterpret_cast<const f32x4*>(mem + idx[(size_t)qi * PK] * D + (pc & 31) * 4);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int pc = e * (NW * 64) + threadIdx.x, ql_ = pc >> 5;
      float ps = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { Qs[ql_ * SIM_RS + (pc & 31) * 4 + k] = qv[e][k]; ps += qv[e][k] * qv[e][k]; }
      ps = row16_sum_hi(ps);
      if ((lane & 15) == 15) qpart[ql_ * 2 + ((lane >> 4) & 1)] = ps;
    }
  }
  __syncthreads();
  const int j = lane & 31, kk = lane >> 5;
  float qn[NQ], tv[NQ];
  int ql[NQ];
  u64 L[NQ][SAMPLE ? 1 : TOPK_MAX], tk[NQ];
#pragma unroll
  for (int nq = 0; nq < NQ; ++nq) {
    const int qi = q0 + nq * 32 + j;
    qn[nq] = qi < B ? sqrtf(qpart[(nq * 32 + j) * 2] + qpart[(nq * 32 + j) * 2 + 1]) : 0.f;
    ql[nq] = qi < B ? (int)batch_label[qi] : -2;
    tk[nq] = (!SAMPLE && qi < B) ? thr[bank * B + qi] : 0;
    tv[nq] = tk[nq] ? knn_value(tk[nq]) : -INFINITY;
#pragma unroll
    for (int k = 0; k < (SAMPLE ? 1 : TOPK_MAX); ++k) L[nq][k] = 0;
  }
  const int jrow = j * 32, gj = (j >> 1) & 7;
  int par = 0, bufc = 0;      // parity of the wave's tile count (label buffer), buffer of the quarter being computed
  for (; t < ntiles; t += stride, par ^= 1) {
    f32x16 acc[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nq][r] = 0.f;
    float nn = 0.f;       // |row j|^2 over the features of parity kk: the matrix operands themselves
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // quarter (t, q) has landed when only the group behind it is still in flight; then the group two ahead is issued into
      // the buffer the previous quarter was read from
      if (q == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      {
        const int b2 = bufc + 2 >= KNN_NBUF ? bufc + 2 - KNN_NBUF : bufc + 2;
        if (q < 2) issue(t, q + 2, b2, par);
        else issue(t + stride, q - 2, b2, par ^ 1);
      }
      const float* R = Rq + bufc * 1024 + jrow;
      // operands of 4 k-steps (8 features = two 16-byte chunks) per batch, the next batch's LDS reads in flight under this
      // batch's MFMAs
      float av[2][4], qv[2][NQ][4];
      auto ld = [&](int buf, int kb) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          av[buf][s4] = R[(((kb * 2 + (s4 >> 1)) ^ gj) << 2) + 2 * (s4 & 1) + kk];
#pragma unroll
          for (int nq = 0; nq < NQ; ++nq) qv[buf][nq][s4] = Qs[(nq * 32 + j) * SIM_RS + q * 32 + kb * 8 + 2 * s4 + kk];
        }
      };
      ld(0, 0);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if (kb + 1 < 4) ld((kb + 1) & 1, kb + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
          for (int nq = 0; nq < NQ; ++nq)
            acc[nq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kb & 1][s4], qv[kb & 1][nq][s4], acc[nq], 0, 0, 0);
          nn += av[kb & 1][s4] * av[kb & 1][s4];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      bufc = bufc + 1 >= KNN_NBUF ? 0 : bufc + 1;
    }
    nn += __shfl_xor(nn, 32, 64);
    if (kk == 0) rnl[j] = sqrtf(nn);
    __builtin_amdgcn_wave_barrier();
    const int* rl = rli + par * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * kk, row = t * 32 + m;
      const float rn = rnl[m];
      const int lab = rl[m];
#pragma unroll
      for (int nq = 0; nq < NQ; ++nq) {
        const float den = rn * qn[nq];
        const bool masked = lab != ql[nq];                    // other classes are masked to similarity 0 (class_mask *)
        const bool live = ql[nq] != -2 && row < n_data;
        if constexpr (SAMPLE) {
          float v = den > 0.f ? acc[nq][r] / den : 0.f;
          if (masked) v = 0.f;
          const u64 key = knn_key(v, row);
          if (live && key > L[nq][0]) L[nq][0] = key;
        } else {
          // v >= tv can only hold if acc >= tv * den up to the rounding of the division (masked / zero-norm rows: v = 0)
          const float bound = tv[nq] * den;
          const bool maybe = masked ? tv[nq] <= 0.f : acc[nq][r] >= bound - fabsf(bound) * 9.5367431640625e-7f - 1e-37f;
          if (live && maybe) {
            float v = den > 0.f ? acc[nq][r] / den : 0.f;
            if (masked) v = 0.f;
            const u64 key = knn_key(v, row);
            if (key >= tk[nq] && key > L[nq][TOPK_MAX - 1]) knn_insert(L[nq], key);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the groups issued past the end)
  if constexpr (SAMPLE) {
    // group maxima: two per (query, sample wave), the half-waves hold disjoint rows
    if (slot < stiles)
#pragma unroll
      for (int nq = 0; nq < NQ; ++nq) {
        const int qi = q0 + nq * 32 + j;
        if (qi < B) gmax[((size_t)bank * B + qi) * (2 * stiles) + 2 * slot + kk] = L[nq][0];
      }
  } else {
    // the two half-waves hold disjoint rows of the same queries: fold the upper half's lists into the lower's, one list per wave
    const int nlists = gridDim.x * KNN_WAVES;
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
      u64 other[TOPK_MAX];
#pragma unroll
      for (int k = 0; k < TOPK_MAX; ++k) other[k] = __shfl_xor(L[nq][k], 32, 64);
      if (kk == 0) {
#pragma unroll
        for (int k = 0; k < TOPK_MAX; ++k)
          if (other[k] > L[nq][TOPK_MAX - 1]) knn_insert(L[nq], other[k]);
        const int qi = nq * 32 + j;
        if (qi < B) {
          u64* dst = cand + ((size_t)bank * B + qi) * TOPK_MAX * nlists + slot;     // [k][list]
          dst[0] = L[nq][0];
#pragma unroll
          for (int k = 1; k < TOPK_MAX; ++k)
            if (L[nq][k - 1]) dst[(size_t)k * nlists] = L[nq][k];
        }
      }
    }
  }
}

// pop the largest key of the wave's sorted private lists (DPP max; keys are unique); the owner drops its head
__device__ __forceinline__ u64 knn_wave_pop(u64 (&L)[TOPK_MAX]) {
  const u64 v = wave_max_u64(L[0]);
  if (L[0] == v && v != 0) {
#pragma unroll
    for (int k = 0; k + 1 < TOPK_MAX; ++k) L[k] = L[k + 1];
    L[TOPK_MAX - 1] = 0;
  }
  return v;
}
// the block's TOPK_MAX largest keys, in order, left in wave 0 (returned one per call of the functor `out(pick, key)` on lane 0)
template <class F>
__device__ __forceinline__ void knn_block_best(u64 (&L)[TOPK_MAX], int npick, F out) {
  __shared__ u64 wk[4 * TOPK_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int pick = 0; pick < TOPK_MAX; ++pick) {
    const u64 v = knn_wave_pop(L);
    if (lane == 0) wk[wave * TOPK_MAX + pick] = v;
  }
  __syncthreads();
  if (wave == 0) {
    L[0] = lane < 4 * TOPK_MAX ? wk[lane] : 0;
#pragma unroll
    for (int k = 1; k < TOPK_MAX; ++k) L[k] = 0;
    for (int pick = 0; pick < npick; ++pick) {
      const u64 v = knn_wave_pop(L);
      if (lane == 0) out(pick, v);
    }
  }
}

// stage 2: thr[bank][query] = the TOPK_MAX-th largest of the query's group maxima (0 when there are fewer: everything passes)
__global__ __launch_bounds__(256) void crd_knn_thr_kernel(const u64* __restrict__ gmax, int ng, u64* __restrict__ thr) {
  const int b = blockIdx.x, bank = blockIdx.y;
  const u64* g = gmax + ((size_t)bank * gridDim.x + b) * ng;
  u64 L[TOPK_MAX];
#pragma unroll
  for (int k = 0; k < TOPK_MAX; ++k) L[k] = 0;
  for (int e = threadIdx.x; e < ng; e += 256) {
    const u64 key = g[e];
    if (key > L[TOPK_MAX - 1]) knn_insert(L, key);
  }
  knn_block_best(L, TOPK_MAX, [&](int pick, u64 v) {
    if (pick == TOPK_MAX - 1) thr[bank * gridDim.x + b] = v;
  });
}

// stage 4: the NP best of the keys a (query, bank) was left with, planes [k][list]
__global__ __launch_bounds__(256) void crd_knn_merge_kernel(const u64* __restrict__ cand, int nlists, int NP, int64_t* __restrict__ nb1,
                                                            int64_t* __restrict__ nb2, float* __restrict__ sim1,
                                                            float* __restrict__ sim2) {
  const int b = blockIdx.x, bank = blockIdx.y;
  const u64* c = cand + ((size_t)bank * gridDim.x + b) * TOPK_MAX * nlists;
  u64 L[TOPK_MAX];
#pragma unroll
  for (int k = 0; k < TOPK_MAX; ++k) L[k] = 0;
  for (int e = threadIdx.x; e < nlists; e += 256)
    for (int k = 0; k < TOPK_MAX; ++k) {
      const u64 key = c[(size_t)k * nlists + e];
      if (!key) break;
      if (key > L[TOPK_MAX - 1]) knn_insert(L, key);
    }
  knn_block_best(L, NP, [&](int pick, u64 v) {
    (bank ? nb2 : nb1)[(size_t)b * NP + pick] = v ? (int64_t)(~(unsigned)v) : (int64_t)0x7fffffff;
    (bank ? sim2 : sim1)[(size_t)b * NP + pick] = v ? knn_value(v) : -INFINITY;
  });
}


// ---- class-mean bank rows (MIA-2023 `pos_extra == "centers"`, nce_p == 2; reference
// "MIA 2023/stage2_unimodal_student/CL_utils/CRD_criterion_v10.py":84-89,121-126: torch.mean over the bank rows of
// every class, recomputed at every call).  The centres are written behind the bank, rows n_data .. n_data + C - 1 of an
// allocation of n_data + C rows, so that the score / loss / gradient kernels address them through the ordinary column
// index lists.  Coalesced 512-B row reads; partial sums per (class, 256-row chunk), combined in a fixed order in
// double: bitwise reproducible.
constexpr int CC_ROWS = 256;
__global__ __launch_bounds__(256) void class_center_partial_kernel(const float* __restrict__ mem, const int* __restrict__ members,
                                                                   const int* __restrict__ offsets, float* __restrict__ parts,
                                                                   int nchunks) {
  const int c = blockIdx.y, chunk = blockIdx.x;
  const int lo = offsets[c] + chunk * CC_ROWS, hi = min(offsets[c + 1], lo + CC_ROWS);
  const int f = threadIdx.x & (D - 1), rl = threadIdx.x >> 7;     // 2 row lanes x 128 features
  float s = 0.f;
  for (int r = lo + rl; r < hi; r += 2) s += mem[(size_t)members[r] * D + f];
  __shared__ float sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < D) parts[((size_t)c * nchunks + chunk) * D + f] = sh[threadIdx.x] + sh[threadIdx.x + D];
}
__global__ void class_center_finish_kernel(const float* __restrict__ parts, const int* __restrict__ offsets, float* __restrict__ mem,
                                           int nchunks, int n_data) {
  const int c = blockIdx.x, f = threadIdx.x;
  const int cnt = offsets[c + 1] - offsets[c], used = (cnt + CC_ROWS - 1) / CC_ROWS;
  double s = 0.0;
  for (int k = 0; k < used; ++k) s += (double)parts[((size_t)c * nchunks + k) * D + f];
  mem[((size_t)n_data + c) * D + f] = cnt > 0 ? (float)(s / (double)cnt) : 0.f;
}

// ---- the memory module / the NCE criterion as STANDALONE calls (ContrastMemory_v3.forward returns (out_v1, out_v2),
// memory_new.py:249-397; ContrastLoss_v2.forward, CRD_loss.py:221-252).  CRDLoss itself runs the fused kernels above.
// out1[b][j] = xs[b][j] / Z_v1, out2 = xt / Z_v2 (:378-379) and the selected PRE-update bank rows are gathered for the
// backward (the momentum update of :382-395 happens inside the same forward call, so a later backward cannot read them
// from the bank any more): rows2[b][j] = mem2[idx_b2[b][sel]] feeds d out_v1 / d v1, rows1 = mem1[idx[b][sel]] d out_v2 / d v2.
// grid (ceil(S2/8), B), block 256 = 8 half-waves, one (sample, column) per half-wave.
__global__ __launch_bounds__(256) void crd_outputs_kernel(const float* __restrict__ xs, const float* __restrict__ xt,
                                                          const int* __restrict__ sel, const int64_t* __restrict__ idx,
                                                          const int64_t* __restrict__ idx_b2,
                                                          const float* __restrict__ mem1, const float* __restrict__ mem2,
                                                          const float* __restrict__ params, float* __restrict__ out1,
                                                          float* __restrict__ out2, float* __restrict__ rows1,
                                                          float* __restrict__ rows2, int PK, int S2) {
  const int b = blockIdx.y, hw = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int j = blockIdx.x * 8 + hw;
  if (j >= S2) return;
  const size_t o = (size_t)b * S2 + j;
  const int col = sel[o];
  const int64_t row = idx[(size_t)b * PK + col], row2 = idx_b2[(size_t)b * PK + col];
  *reinterpret_cast<f32x4*>(rows1 + o * D + l * 4) = *reinterpret_cast<const f32x4*>(mem1 + row * D + l * 4);
  *reinterpret_cast<f32x4*>(rows2 + o * D + l * 4) = *reinterpret_cast<const f32x4*>(mem2 + row2 * D + l * 4);
  if (l == 0) {
    out1[o] = xs[o] / params[2];
    out2[o] = xt[o] / params[3];
  }
}

// dv1[b] = sum_j g1[b][j] out1[b][j] / T * rows2[b][j][:], dv2[b] = sum_j g2[b][j] out2[b][j] / T * rows1[b][j][:]
// (out = exp(row . v / T) / Z with Z a constant: memory_new.py:270-278,378-379).  One block per (sample, side), fixed
// summation order (8 half-waves stride the columns, then a fixed-order LDS combine).
__global__ __launch_bounds__(256) void crd_outputs_bwd_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                              const float* __restrict__ out1, const float* __restrict__ out2,
                                                              const float* __restrict__ rows1, const float* __restrict__ rows2,
                                                              float invT, float* __restrict__ dv1, float* __restrict__ dv2,
                                                              int S2) {
  const int b = blockIdx.x, side = blockIdx.y, hw = threadIdx.x >> 5, l = threadIdx.x & 31;
  const float* g = side ? g2 : g1;
  const float* out = side ? out2 : out1;
  const float* rows = side ? rows1 : rows2;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int j = hw; j < S2; j += 8) {
    const size_t o = (size_t)b * S2 + j;
    const float c = (g ? g[o] : 0.f) * out[o] * invT;
    const f32x4 r = *reinterpret_cast<const f32x4*>(rows + o * D + l * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += c * r[k];
  }
  __shared__ float sh[8][D];
#pragma unroll
  for (int k = 0; k < 4; ++k) sh[hw][l * 4 + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < D) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sh[q][threadIdx.x];
    (side ? dv2 : dv1)[(size_t)b * D + threadIdx.x] = t;
  }
}

// ContrastLoss_v2.forward(x [B][P+N], P) (CRD_loss.py:221-252): m = N, Pn = 1 / n_data, c = m Pn + eps;
//   rows[b] = -( sum_p log(x/(x+c)) / P + sum_n log(m Pn / (x+c)) )      (the sample_KD == "True" branch's per-sample loss)
//   the sample_KD == "False" branch's scalar is sum_b rows[b] / B (host: ph_sum with scale 1/B)
//   dx[b][j] = d rows[b] / d x[b][j] = -c / (x (x+c)) / P (positives), +1 / (x+c) (negatives)
__global__ __launch_bounds__(256) void contrast_loss_v2_kernel(const float* __restrict__ x, float* __restrict__ rows,
                                                               float* __restrict__ dx, int S, int P, float n_data) {
  const int b = blockIdx.x;
  const float mPn = (float)(S - P) / n_data, c = mPn + 1e-7f;
  double acc = 0.0;
  for (int j = threadIdx.x; j < S; j += blockDim.x) {
    const float v = x[(size_t)b * S + j];
    if (j < P) {
      acc += (double)logf(v / (v + c)) / (double)P;
      dx[(size_t)b * S + j] = -(c / (v * (v + c))) / (float)P;
    } else {
      acc += (double)logf(mPn / (v + c));
      dx[(size_t)b * S + j] = 1.f / (v + c);
    }
  }
  __shared__ double sh[4];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) rows[b] = -(float)(sh[0] + sh[1] + sh[2] + sh[3]);
}


// ---- bank-scan form of the negatives (SURVEY 8-e assumption (i): nce_k at or above the number of bank rows).  The gathered
// kernels read B x K bank rows per bank and call (K = 65 536, B = 64: 4.3 GB); every row of a 65 536-row bank is then drawn ~once
// per query, so the same sums are taken over ALL rows weighted by how often each was drawn: scores S = V x bank^T (one read of the
// 33.5 MB bank), mult[b][r] = the multiplicity of row r among query b's negatives,
//   loss_neg[b] = -sum_r mult log(m Pn / (x + c)),  x = exp(S / T) / Z;   d loss / d S = mult (x / (x + c)) / T.
// grid (bin chunks, B): a workgroup counts the negatives of query b that fall into its HIST_BINS bank rows in LDS (integer
// ds_add: order-independent) and writes the chunk's counts once.  (First version: one global atomicAdd per index into the
// [B][n_data] matrix, zeroed by a memset - 158 us per call at B = 64, K = n_data = 65 536, the largest item of the scan form.)
constexpr int HIST_BINS = 32768;      // 128 KB of LDS counters
__global__ __launch_bounds__(1024) void crd_neg_hist_kernel(const int64_t* __restrict__ idx, long row_stride, int K, int n_data,
                                                           int* __restrict__ mult) {
  extern __shared__ int hbins[];
  const int b = blockIdx.y, r0 = blockIdx.x * HIST_BINS;
  const int nb = n_data - r0 < HIST_BINS ? n_data - r0 : HIST_BINS;
  for (int i = threadIdx.x; i < nb; i += 1024) hbins[i] = 0;
  __syncthreads();
  const int64_t* row = idx + (size_t)b * row_stride;
  for (int k = threadIdx.x; k < K; k += 1024) {
    const int64_t r = row[k] - r0;
    if (r >= 0 && r < nb) atomicAdd(&hbins[(int)r], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 1024) mult[(size_t)b * n_data + r0 + i] = hbins[i];
}

constexpr int SCAN_CHUNK = 2048;      // bank rows per workgroup
// grid (chunks, B): S1 = v1 . bank2^T (pairs with Z_v1, feeds dv1), S2 = v2 . bank1^T; both overwritten by the coefficients
// d loss / d S (x inv_bnorm) unless zsum_only.  part[(b * chunks + chunk) * 4 + {0,1,2,3}] = loss terms of S1 | of S2 | sum mult e1 | sum mult e2
__global__ __launch_bounds__(256) void crd_scan_neg_kernel(float* __restrict__ S1, float* __restrict__ S2, const int* __restrict__ mult,
                                                          const float* __restrict__ params, float* __restrict__ part, int n_data,
                                                          int m_neg, float inv_bnorm, int zsum_only) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const float invT = 1.f / params[1], Z1 = params[2], Z2 = params[3];
  const float mPn = (float)m_neg / (float)n_data, c = mPn + 1e-7f;
  double l1 = 0.0, l2 = 0.0, z1 = 0.0, z2 = 0.0;
  const int r0 = chunk * SCAN_CHUNK, r1 = r0 + SCAN_CHUNK < n_data ? r0 + SCAN_CHUNK : n_data;
  for (int r = r0 + threadIdx.x; r < r1; r += 256) {
    const size_t o = (size_t)b * n_data + r;
    const int mu = mult[o];
    const float e1 = expf(S1[o] * invT), e2 = expf(S2[o] * invT);
    if (zsum_only) {
      z1 += (double)mu * (double)e1; z2 += (double)mu * (double)e2;
    } else {
      const float x1 = e1 / Z1, x2 = e2 / Z2, fm = (float)mu;
      if (mu) { l1 += (double)(fm * logf(mPn / (x1 + c))); l2 += (double)(fm * logf(mPn / (x2 + c))); }
      S1[o] = fm * (x1 / (x1 + c)) * invT * inv_bnorm;
      S2[o] = fm * (x2 / (x2 + c)) * invT * inv_bnorm;
    }
  }
  __shared__ double sh[4][4];
  l1 = wave_sum_d(l1); l2 = wave_sum_d(l2); z1 = wave_sum_d(z1); z2 = wave_sum_d(z2);
  if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sh[w][0] = l1; sh[w][1] = l2; sh[w][2] = z1; sh[w][3] = z2; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const double t = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
    part[((size_t)b * gridDim.x + chunk) * 4 + threadIdx.x] = (float)t;
  }
}
// fixed-order combine: loss_neg[b] = -(sum over chunks) * inv_bnorm, or (zsum_only) zsums[0..1] += the two exp sums over everything
__global__ __launch_bounds__(64) void crd_scan_neg_finish_kernel(const float* __restrict__ part, int chunks, int B, float inv_bnorm,
                                                                float* __restrict__ loss_neg, float* __restrict__ zsums, int zsum_only) {
  if (zsum_only) {
    if (threadIdx.x < 2 && blockIdx.x == 0) {
      double t = 0.0;
      for (int i = 0; i < B * chunks; ++i) t += (double)part[(size_t)i * 4 + 2 + threadIdx.x];
      zsums[threadIdx.x] += (float)t;
    }
    return;
  }
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double t = 0.0;
  for (int ch = 0; ch < chunks; ++ch) t += (double)part[((size_t)b * chunks + ch) * 4] + (double)part[((size_t)b * chunks + ch) * 4 + 1];
  loss_neg[b] = -(float)t * inv_bnorm;
}
}  // namespace

// workspace: list keys [2][B][TOPK_MAX][nlists] u64 | group maxima [2][B][2 * stiles] | thr [2][B]  (B <= 64 per pass)
static inline int knn_gx(int n_data) {
  const int g = cdiv(cdiv(n_data, 32), KNN_WAVES);
  return g < KNN_MAX_GX ? g : KNN_MAX_GX;      // one workgroup per CU over the two banks; a wave walks its tiles
}
static inline int knn_stiles(int n_data) {
  const int nt = cdiv(n_data, 32);
  return nt < KNN_SAMPLE_TILES ? nt : KNN_SAMPLE_TILES;
}
size_t ph_crd_bank_topk_workspace_bytes(int B, int n_data) {
  const size_t bc = B < KNN_MAX_B ? B : KNN_MAX_B;
  return 2 * bc * ((size_t)knn_gx(n_data) * KNN_WAVES * TOPK_MAX + 2 * knn_stiles(n_data) + 1) * sizeof(u64) + 256;
}

int ph_crd_bank_topk(const float* mem1, const float* mem2, const int* labels, const int64_t* idx, int PK,
                     const int64_t* batch_label, int B, int n_data, int num_pos, int feat_dim, int64_t* nb1,
                     int64_t* nb2, float* sim1, float* sim2, void* workspace, hipStream_t st) {
  if (feat_dim != D || num_pos < 1 || num_pos > TOPK_MAX || !workspace || B < 1 || n_data < 1) return PH_EINVAL;
  if (B > KNN_MAX_B) {
    // a lane selects for one query per 32-query block and two blocks fill its registers (accumulators + lists): larger batches
    // (the reference has no limit; a replica batch of 256 is the north-star size) run in chunks of 64 queries through the same
    // workspace, in stream order - the matrix work is the same, only the 67 MB bank is read once per chunk
    for (int c0 = 0; c0 < B; c0 += KNN_MAX_B) {
      const int bc = B - c0 < KNN_MAX_B ? B - c0 : KNN_MAX_B;
      const int rc = ph_crd_bank_topk(mem1, mem2, labels, idx + (size_t)c0 * PK, PK, batch_label + c0, bc, n_data, num_pos, feat_dim,
                                      nb1 + (size_t)c0 * num_pos, nb2 + (size_t)c0 * num_pos, sim1 + (size_t)c0 * num_pos,
                                      sim2 + (size_t)c0 * num_pos, workspace, st);
      if (rc) return rc;
    }
    return PH_OK;
  }
  const int gx = knn_gx(n_data), nlists = gx * KNN_WAVES, stiles = knn_stiles(n_data), tstride = cdiv(n_data, 32) / stiles;
  u64* cand = reinterpret_cast<u64*>(workspace);
  u64* gmax = cand + (size_t)2 * B * TOPK_MAX * nlists;
  u64* thr = gmax + (size_t)2 * B * 2 * stiles;
  void* tok = nullptr;
  if (ph_prof_on())   // algorithmic bytes: every row of both banks once + the row labels + the 2 x B x num_pos results
    ph_prof_begin(PH_CLS_CRD_TOPK, 2.0 * n_data * D * 4 + 4.0 * n_data + 2.0 * B * num_pos * 12, st, &tok);
  const int nq = cdiv(B, 32);      // <= 2 (chunked above)
#define PH_KNN_LAUNCH(N, SAMPLE, GX, GZ)                                                                                        \
  do {                                                                                                                      \
    constexpr int NW = SAMPLE ? KNN_SAMPLE_WAVES : KNN_WAVES;                                                               \
    const size_t lds = (size_t)(N * 32 * SIM_RS + NW * (KNN_NBUF * 1024 + 32 + 128) + N * 64) * sizeof(float);              \
    static bool done = false;                                                                                               \
    if (!done) {                                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(crd_bank_knn_kernel<N, SAMPLE>),                                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                          \
        return PH_ELAUNCH;                                                                                                  \
      done = true;                                                                                                          \
    }                                                                                                                       \
    hipLaunchKernelGGL((crd_bank_knn_kernel<N, SAMPLE>), dim3(GX, 2, GZ), dim3(NW * 64), lds, st, mem1, mem2, labels, idx, PK, \
                       batch_label, B, n_data, stiles, tstride, gmax, thr, cand);                                           \
  } while (0)
  PH_KNN_LAUNCH(1, true, cdiv(stiles, KNN_SAMPLE_WAVES), nq);      // one 32-query block per workgroup (grid z)
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(crd_knn_thr_kernel, dim3(B, 2), dim3(256), 0, st, gmax, 2 * stiles, thr);
  PH_LAUNCH_CHECK();
  if (nq == 1) PH_KNN_LAUNCH(1, false, gx, 1); else PH_KNN_LAUNCH(2, false, gx, 1);
#undef PH_KNN_LAUNCH
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(crd_knn_merge_kernel, dim3(B, 2), dim3(256), 0, st, cand, nlists, num_pos, nb1, nb2, sim1, sim2);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_crd_score(const float* v1, const float* v2, const int64_t* idx, const int64_t* idx_bank2, const float* mem1,
                 const float* mem2, float* out1, float* out2, float* diff, int B, int PK, int feat_dim, float T,
                 hipStream_t st) {
  if (feat_dim != D) return PH_EINVAL;
  void* tok = nullptr;
  if (ph_prof_on())   // algorithmic bytes: one 512-B row of each bank per (sample, column) + the three [B][P+K] outputs
    ph_prof_begin(PH_CLS_CRD_SCORE, 2.0 * B * PK * D * 4 + 3.0 * B * PK * 4 + 2.0 * B * D * 4, st, &tok);
  hipLaunchKernelGGL(crd_score_kernel, dim3(cdiv(PK, 64), B), dim3(256), 0, st, v1, v2, idx,
                     idx_bank2 ? idx_bank2 : idx, mem1, mem2, out1, out2, diff, PK, 1.f / T);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_crd_select(const float* diff, const float* out1, const float* out2, const int* ranks, int* sel, float* xs,
                  float* xt, int B, int P, int K, int P2, int K2, int select_neg, int select_pos, hipStream_t st) {
  if (P2 > P || K2 > K || (!select_pos && P2 != P) || (!select_neg && K2 != K)) return PH_EINVAL;
  const bool ranked = select_pos || select_neg;
  const size_t lds = ranked ? (size_t)(P + K) * 4 + (size_t)P * 4 : 0;
  if (lds > 160 * 1024) return PH_EINVAL;      // a ranked selection keeps the sample's discrepancy list in LDS
  // 1024 threads per sample: the rank counting is instruction-bound (P^2 + K^2 comparisons), 256 threads left one wave
  // per SIMD on B of the 256 CUs (76 us at B = 64, P + K = 1000)
  hipLaunchKernelGGL(crd_select_kernel, dim3(B), dim3(1024), lds, st, diff, out1, out2, ranks, sel, xs, xt, P, K, P2,
                     K2, select_neg, select_pos);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_crd_zsum(const float* xs, const float* xt, float* sums, int n, hipStream_t st) {
  hipLaunchKernelGGL(crd_zsum_kernel, dim3(1), dim3(1024), 0, st, xs, xt, sums, n);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_crd_setz(float* params, const float* sums, float count, float n_data, hipStream_t st) {
  hipLaunchKernelGGL(crd_setz_kernel, dim3(1), dim3(64), 0, st, params, sums, count, n_data);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
constexpr int LG_SPLIT_MAX = 8;
size_t ph_crd_loss_grad_workspace_bytes(int B) { return (size_t)B * LG_SPLIT_MAX * (2 * D + 1) * sizeof(float); }

static int crd_loss_grad_impl(const float* xs, const float* xt, const int* sel, const int64_t* idx, const int64_t* idx_bank2,
                              const float* posw_s, const float* posw_t, const float* mem1, const float* mem2,
                              const float* params, float* lossp, float* dv1, float* dv2, int B, int PK, int P2, int K2, int m_neg,
                              int feat_dim, float n_data, float inv_bnorm, void* workspace, hipStream_t st);
int ph_crd_loss_grad(const float* xs, const float* xt, const int* sel, const int64_t* idx, const int64_t* idx_bank2,
                     const float* posw_s, const float* posw_t, const float* mem1, const float* mem2,
                     const float* params, float* lossp, float* dv1, float* dv2, int B, int PK, int P2, int K2,
                     int feat_dim, float n_data, float inv_bnorm, void* workspace, hipStream_t st) {
  return crd_loss_grad_impl(xs, xt, sel, idx, idx_bank2, posw_s, posw_t, mem1, mem2, params, lossp, dv1, dv2, B, PK, P2, K2, K2,
                            feat_dim, n_data, inv_bnorm, workspace, st);
}
// the positive columns alone, with the NCE constant of m_neg negatives that ph_crd_scan_neg accounts for
int ph_crd_loss_grad_pos(const float* xs, const float* xt, const int* sel, const int64_t* idx, const int64_t* idx_bank2,
                         const float* posw_s, const float* posw_t, const float* mem1, const float* mem2,
                         const float* params, float* lossp, float* dv1, float* dv2, int B, int P, int m_neg,
                         int feat_dim, float n_data, float inv_bnorm, hipStream_t st) {
  if (m_neg < 1 || P < 1) return PH_EINVAL;
  return crd_loss_grad_impl(xs, xt, sel, idx, idx_bank2, posw_s, posw_t, mem1, mem2, params, lossp, dv1, dv2, B, P, P, 0, m_neg,
                            feat_dim, n_data, inv_bnorm, nullptr, st);
}
static int crd_loss_grad_impl(const float* xs, const float* xt, const int* sel, const int64_t* idx, const int64_t* idx_bank2,
                              const float* posw_s, const float* posw_t, const float* mem1, const float* mem2,
                              const float* params, float* lossp, float* dv1, float* dv2, int B, int PK, int P2, int K2, int m_neg,
                              int feat_dim, float n_data, float inv_bnorm, void* workspace, hipStream_t st) {
  if (feat_dim != D) return PH_EINVAL;
  int ns = workspace ? (P2 + K2) / 512 : 1;     // one workgroup per 512 columns of a sample, at most LG_SPLIT_MAX
  ns = ns < 1 ? 1 : (ns > LG_SPLIT_MAX ? LG_SPLIT_MAX : ns);
  void* tok = nullptr;
  if (ph_prof_on())   // algorithmic bytes: the selected rows of both banks + the two gradient rows per sample
    ph_prof_begin(PH_CLS_CRD_LOSSGRAD, 2.0 * B * (P2 + K2) * D * 4 + 2.0 * B * D * 4 + 2.0 * B * (P2 + K2) * 4, st, &tok);
  hipLaunchKernelGGL(crd_loss_grad_kernel, dim3(B, ns), dim3(1024), 0, st, xs, xt, sel, idx, idx_bank2 ? idx_bank2 : idx,
                     posw_s, posw_t, mem1, mem2, params, lossp, dv1, dv2, PK, P2, K2, m_neg, n_data, inv_bnorm,
                     reinterpret_cast<float*>(workspace));
  PH_LAUNCH_CHECK();
  if (ns > 1) {
    hipLaunchKernelGGL(crd_loss_grad_reduce_kernel, dim3(B), dim3(256), 0, st, reinterpret_cast<const float*>(workspace), lossp,
                       dv1, dv2, ns, inv_bnorm);
    PH_LAUNCH_CHECK();
  }
  ph_prof_end(tok, st);
  return PH_OK;
}
// ---- bank-scan form of the negatives (kernels above).  idx + col0: the K sampled negatives of query b are idx[b * row_stride + col0 ..]
int ph_crd_neg_hist(const int64_t* idx, long row_stride, int col0, int K, int B, int n_data, int* mult, hipStream_t st) {
  if (B < 1 || K < 1 || n_data < 1 || !idx || !mult) return PH_EINVAL;
  static std::once_flag attr_once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(attr_once, [] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(crd_neg_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  HIST_BINS * (int)sizeof(int));
  });
  if (attr_rc != hipSuccess) return PH_ELAUNCH;
  hipLaunchKernelGGL(crd_neg_hist_kernel, dim3(cdiv(n_data, HIST_BINS), B), dim3(1024), HIST_BINS * sizeof(int), st, idx + col0,
                     row_stride, K, n_data, mult);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
size_t ph_crd_scan_neg_workspace_bytes(int B, int n_data) { return (size_t)B * cdiv(n_data, SCAN_CHUNK) * 4 * sizeof(float); }
int ph_crd_scan_neg(float* S1, float* S2, const int* mult, const float* params, void* workspace, float* loss_neg, float* zsums,
                    int B, int n_data, int m_neg, float inv_bnorm, int zsum_only, hipStream_t st) {
  if (B < 1 || n_data < 1 || m_neg < 1 || !workspace || (zsum_only ? !zsums : !loss_neg)) return PH_EINVAL;
  const int chunks = cdiv(n_data, SCAN_CHUNK);
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(crd_scan_neg_kernel, dim3(chunks, B), dim3(256), 0, st, S1, S2, mult, params, part, n_data, m_neg, inv_bnorm,
                     zsum_only);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(crd_scan_neg_finish_kernel, dim3(zsum_only ? 1 : cdiv(B, 64)), dim3(64), 0, st, part, chunks, B, inv_bnorm,
                     loss_neg, zsums, zsum_only);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_crd_update(float* mem1, float* mem2, const float* v1, const float* v2, const int64_t* y, const float* params,
                  int B, int feat_dim, hipStream_t st) {
  if (feat_dim != D) return PH_EINVAL;
  hipLaunchKernelGGL(crd_update_kernel, dim3(cdiv(2 * B, 4)), dim3(256), 0, st, mem1, mem2, v1, v2, y, params, B);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

size_t ph_crd_class_centers_workspace_bytes(int num_classes, int max_class_rows) {
  return (size_t)num_classes * (size_t)cdiv(max_class_rows > 0 ? max_class_rows : 1, CC_ROWS) * D * sizeof(float);
}

int ph_crd_class_centers(float* mem_ext, const int* members, const int* offsets, int num_classes, int max_class_rows,
                         int n_data, int feat_dim, void* workspace, hipStream_t st) {
  if (feat_dim != D || !mem_ext || !members || !offsets || !workspace || num_classes < 1) return PH_EINVAL;
  const int nchunks = cdiv(max_class_rows > 0 ? max_class_rows : 1, CC_ROWS);
  hipLaunchKernelGGL(class_center_partial_kernel, dim3(nchunks, num_classes), dim3(256), 0, st, mem_ext, members, offsets,
                     reinterpret_cast<float*>(workspace), nchunks);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(class_center_finish_kernel, dim3(num_classes), dim3(D), 0, st, reinterpret_cast<const float*>(workspace),
                     offsets, mem_ext, nchunks, n_data);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_crd_outputs(const float* xs, const float* xt, const int* sel, const int64_t* idx, const int64_t* idx_bank2,
                   const float* mem1, const float* mem2, const float* params, float* out1, float* out2, float* rows1,
                   float* rows2, int B, int PK, int S2, int feat_dim, hipStream_t st) {
  if (feat_dim != D || B < 1 || S2 < 1) return PH_EINVAL;
  hipLaunchKernelGGL(crd_outputs_kernel, dim3(cdiv(S2, 8), B), dim3(256), 0, st, xs, xt, sel, idx,
                     idx_bank2 ? idx_bank2 : idx, mem1, mem2, params, out1, out2, rows1, rows2, PK, S2);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_crd_outputs_bwd(const float* g1, const float* g2, const float* out1, const float* out2, const float* rows1,
                       const float* rows2, float T, float* dv1, float* dv2, int B, int S2, int feat_dim, hipStream_t st) {
  if (feat_dim != D || B < 1 || S2 < 1) return PH_EINVAL;
  hipLaunchKernelGGL(crd_outputs_bwd_kernel, dim3(B, 2), dim3(256), 0, st, g1, g2, out1, out2, rows1, rows2, 1.f / T, dv1,
                     dv2, S2);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_contrast_loss_v2(const float* x, float* rows, float* dx, int B, int S, int P, float n_data, hipStream_t st) {
  if (B < 1 || P < 1 || P >= S) return PH_EINVAL;
  hipLaunchKernelGGL(contrast_loss_v2_kernel, dim3(B), dim3(256), 0, st, x, rows, dx, S, P, n_data);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
