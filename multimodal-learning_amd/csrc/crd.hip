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
                                                            float* __restrict__ dv2, int PK, int P2, int K2,
                                                            float n_data, float inv_bnorm, float* __restrict__ part) {
  // gridDim.y > 1 (long column lists, e.g. nce_k = 4096): the columns of a sample are dealt to gridDim.y workgroups whose
  // partial sums go to `part` and are added in a fixed order by crd_loss_grad_reduce_kernel
  const int b = blockIdx.x, S2 = P2 + K2;
  const int hw = threadIdx.x >> 5, l = threadIdx.x & 31;
  const float invT = 1.f / params[1], Z1 = params[2], Z2 = params[3];
  const float mPn = (float)K2 / n_data, c = mPn + 1e-7f;
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
constexpr int TOPK_MAX = 8, TOPK_SPLIT = 8;     // (32 slices: 4096 selection workgroups of 8 arg-max rounds each, 83 us at B = 64)

// block-wide pick of the best (value desc, index asc) live candidate among n entries of (sv, si); every thread gets it
__device__ __forceinline__ void block_argbest(const float* sv, const int* si, int n, float* rv, int* ri, int* rs, float& best,
                                              int& besti, int& bestslot) {
  const int tid = threadIdx.x;
  float v = -INFINITY; int i = 0x7fffffff, sl = -1;
  for (int e = tid; e < n; e += 256) {
    const float ve = sv[e]; const int ie = si[e];
    if (ie != 0x7fffffff && (ve > v || (ve == v && ie < i))) { v = ve; i = ie; sl = e; }
  }
  rv[tid] = v; ri[tid] = i; rs[tid] = sl;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      const float v2 = rv[tid + o]; const int i2 = ri[tid + o];
      if (i2 != 0x7fffffff && (v2 > rv[tid] || (v2 == rv[tid] && i2 < ri[tid]))) { rv[tid] = v2; ri[tid] = i2; rs[tid] = rs[tid + o]; }
    }
    __syncthreads();
  }
  best = rv[0]; besti = ri[0]; bestslot = rs[0];
  __syncthreads();
}

// stage 0: the B query rows of each bank (the bank row of the sample itself, idx[b][0]) copied next to their norms
__global__ __launch_bounds__(64) void crd_topk_queries_kernel(const float* __restrict__ mem1, const float* __restrict__ mem2,
                                                              const int64_t* __restrict__ idx, int PK, float* __restrict__ qbuf,
                                                              float* __restrict__ qnorm) {
  const int b = blockIdx.x, bank = blockIdx.y, t = threadIdx.x;
  const float* mem = bank ? mem2 : mem1;
  const int64_t qrow = idx[(size_t)b * PK];
  const float q0 = mem[qrow * D + t], q1 = mem[qrow * D + t + 64];
  float* q = qbuf + ((size_t)bank * gridDim.x + b) * D;
  q[t] = q0; q[t + 64] = q1;
  const float s = wave_sum(q0 * q0 + q1 * q1);
  if (t == 0) qnorm[bank * gridDim.x + b] = sqrtf(s);
}

// stage 1: class-masked cosine similarity of EVERY bank row with EVERY query, S[bank][b][j] (CRD_criterion_v10.py:72-79: the
// reference's cosine_similarity(bank[idx], bank) * class_mask) - a [B x 128] x [128 x n_data] product per bank on the EXACT
// fp32 matrix instruction (v_mfma_f32_32x32x2_f32: bit for bit a k-ordered fmaf chain, so every similarity is the
// sequential dot product the scalar kernels computed).  A wave owns tiles of 32 bank rows: the rows arrive by coalesced
// 16-byte loads (next tile prefetched into registers during the matrix loop) and sit in LDS with an odd row stride; the
// queries sit in LDS once per workgroup; queries are the M side, bank rows the N side, so a lane's results are 32 consecutive
// bank rows of one query: 128-byte stores.  History: one workgroup per (query, bank, slice) re-read the bank per query
// (1.4 GB through L2, 278 us); a thread per bank row with the queries as scalar operands read the bank once but ran 120-132 us
// (scalar-load latency per 128-feature query and one dependent fmac chain per pair; four interleaved chains and coalesced
// row loads moved it by 10 %).
constexpr int SIM_RS = 129;     // dwords per staged row (odd: the 32 lanes of a ds_read_b32 group hit 32 different banks)

template <int NQ>
__global__ __launch_bounds__(256) void crd_bank_sim_kernel(const float* __restrict__ mem1, const float* __restrict__ mem2,
                                                           const int* __restrict__ labels, const float* __restrict__ qbuf,
                                                           const float* __restrict__ qnorm,
                                                           const int64_t* __restrict__ batch_label, int B, int n_data,
                                                           float* __restrict__ S) {
  extern __shared__ __attribute__((aligned(16))) float sim_lds[];
  float* Qs = sim_lds;                                   // [NQ * 32][SIM_RS]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* Rs = sim_lds + NQ * 32 * SIM_RS + wv * 32 * SIM_RS;   // this wave's [32][SIM_RS]
  const int bank = blockIdx.y;
  const float* mem = bank ? mem2 : mem1;
  const float* q = qbuf + (size_t)bank * B * D;
  float* Sb = S + (size_t)bank * B * n_data;
  for (int e = threadIdx.x; e < NQ * 32 * D; e += 256) {
    const int qi = e / D, d = e - qi * D;
    Qs[qi * SIM_RS + d] = qi < B ? q[(size_t)qi * D + d] : 0.f;
  }
  // per lane: norm and label of the 16 * NQ queries its accumulator registers belong to
  float qn[NQ][16];
  int ql[NQ][16];
#pragma unroll
  for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = nq * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      qn[nq][r] = qi < B ? qnorm[bank * B + qi] : 0.f;
      ql[nq][r] = qi < B ? (int)batch_label[qi] : -2;
    }
  __syncthreads();
  const int ntiles = (n_data + 31) / 32;
  const int stride = gridDim.x * 4;
  f32x4 pre[16];
  auto load_tile = [&](int t) {       // 32 rows x 512 B: instruction e covers rows 2e, 2e + 1
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = t * 32 + e * 2 + (lane >> 5);
      pre[e] = *reinterpret_cast<const f32x4*>(mem + (size_t)(row < n_data ? row : 0) * D + (lane & 31) * 4);
    }
  };
  auto stage_tile = [&]() {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float* dst = Rs + (e * 2 + (lane >> 5)) * SIM_RS + (lane & 31) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[k] = pre[e][k];
    }
  };
  int t = blockIdx.x * 4 + wv;
  if (t < ntiles) load_tile(t);
  for (; t < ntiles; t += stride) {
    stage_tile();
    if (t + stride < ntiles) load_tile(t + stride);
    __builtin_amdgcn_wave_barrier();
    const int j = lane & 31, kk = lane >> 5, row = t * 32 + j;
    // norm of the lane's bank row, features in index order (bitwise the scalar kernels' value)
    float nn = 0.f;
#pragma unroll 16
    for (int d = 0; d < D; ++d) { const float v = Rs[j * SIM_RS + d]; nn += v * v; }
    const float rn = sqrtf(nn);
    const int lab = row < n_data ? labels[row] : -1;
    f32x16 acc[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nq][r] = 0.f;
#pragma unroll 8
    for (int k0 = 0; k0 < D; k0 += 2) {
      const float bv = Rs[j * SIM_RS + k0 + kk];
#pragma unroll
      for (int nq = 0; nq < NQ; ++nq)
        acc[nq] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[(nq * 32 + j) * SIM_RS + k0 + kk], bv, acc[nq], 0, 0, 0);
    }
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = nq * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        const float den = rn * qn[nq][r];
        float v = den > 0.f ? acc[nq][r] / den : 0.f;
        if (lab != ql[nq][r]) v = 0.f;                       // other classes are masked to similarity 0 (class_mask *)
        if (qi < B && row < n_data) Sb[(size_t)qi * n_data + row] = v;
      }
    __builtin_amdgcn_wave_barrier();
  }
}

// stage 2: block (query b, bank, slice z) scans its slice of the similarity row; every thread keeps a private top list over
// its columns, the block picks its TOPK_MAX best by parallel arg-max rounds and leaves them in the workspace
__global__ __launch_bounds__(256) void crd_bank_topk_kernel(const float* __restrict__ S, int B, int n_data,
                                                            float* __restrict__ cand_v, int* __restrict__ cand_i) {
  const int b = blockIdx.x, bank = blockIdx.y, z = blockIdx.z;
  const float* row = S + ((size_t)bank * B + b) * n_data;
  float bv[TOPK_MAX]; int bi[TOPK_MAX];
#pragma unroll
  for (int k = 0; k < TOPK_MAX; ++k) { bv[k] = -INFINITY; bi[k] = 0x7fffffff; }
  const int chunk = (n_data + TOPK_SPLIT - 1) / TOPK_SPLIT, lo = z * chunk, hi = min(n_data, lo + chunk);
  for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) {
    const float v = row[j];
    // insert (v, j) into the descending private list (ties: lower index first)
    if (v > bv[TOPK_MAX - 1] || (v == bv[TOPK_MAX - 1] && j < bi[TOPK_MAX - 1])) {
      int pos = TOPK_MAX - 1;
#pragma unroll
      for (int k = TOPK_MAX - 1; k > 0; --k) {
        const bool up = (v > bv[k - 1]) || (v == bv[k - 1] && j < bi[k - 1]);
        if (up) { bv[k] = bv[k - 1]; bi[k] = bi[k - 1]; pos = k - 1; }
      }
      bv[pos] = v; bi[pos] = j;
    }
  }
  // The block's TOPK_MAX best out of its 256 sorted private lists: every wave pops the best list head TOPK_MAX times
  // (butterfly arg-max over the lanes: value descending, index ascending - a total order, so the result does not depend on
  // how the elements are dealt to threads), the four waves' winners meet in LDS and wave 0 repeats the selection on those
  // 32.  (The first version kept all 2048 candidates in LDS and ran TOPK_MAX block-wide arg-max rounds of nine barriers
  // each: 62 us at 65 536 rows x 64 queries.)
  auto better = [](float v, int i, float v2, int i2) { return v2 > v || (v2 == v && i2 < i); };
  auto wave_pop_best = [&](float& ov, int& oi) {
    float v = bv[0]; int i = bi[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(v, o, 64); const int i2 = __shfl_xor(i, o, 64);
      if (better(v, i, v2, i2)) { v = v2; i = i2; }
    }
    ov = v; oi = i;
    if (bi[0] == i && i != 0x7fffffff) {      // the owner drops its head (indices are unique within the block)
#pragma unroll
      for (int k = 0; k + 1 < TOPK_MAX; ++k) { bv[k] = bv[k + 1]; bi[k] = bi[k + 1]; }
      bv[TOPK_MAX - 1] = -INFINITY; bi[TOPK_MAX - 1] = 0x7fffffff;
    }
  };
  __shared__ float wv_[4 * TOPK_MAX];
  __shared__ int wi_[4 * TOPK_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int pick = 0; pick < TOPK_MAX; ++pick) {
    float v; int i;
    wave_pop_best(v, i);
    if (lane == 0) { wv_[wave * TOPK_MAX + pick] = v; wi_[wave * TOPK_MAX + pick] = i; }
  }
  __syncthreads();
  if (wave == 0) {
    bv[0] = lane < 4 * TOPK_MAX ? wv_[lane] : -INFINITY;
    bi[0] = lane < 4 * TOPK_MAX ? wi_[lane] : 0x7fffffff;
#pragma unroll
    for (int k = 1; k < TOPK_MAX; ++k) { bv[k] = -INFINITY; bi[k] = 0x7fffffff; }
    const size_t base = (((size_t)b * 2 + bank) * TOPK_SPLIT + z) * TOPK_MAX;
#pragma unroll
    for (int pick = 0; pick < TOPK_MAX; ++pick) {
      float v; int i;
      wave_pop_best(v, i);
      if (lane == 0) { cand_v[base + pick] = v; cand_i[base + pick] = i; }
    }
  }
}

// stage 3: the NP best of the TOPK_SPLIT * TOPK_MAX candidates of a (query, bank)
__global__ __launch_bounds__(256) void crd_bank_topk_merge_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i,
                                                                  int NP, int64_t* __restrict__ nb1, int64_t* __restrict__ nb2,
                                                                  float* __restrict__ sim1, float* __restrict__ sim2) {
  const int b = blockIdx.x, bank = blockIdx.y;
  constexpr int NC = TOPK_SPLIT * TOPK_MAX;
  __shared__ float sv[NC];
  __shared__ int si[NC];
  __shared__ float rv[256];
  __shared__ int ri[256], rs[256];
  const size_t base = ((size_t)b * 2 + bank) * NC;
  for (int e = threadIdx.x; e < NC; e += 256) { sv[e] = cand_v[base + e]; si[e] = cand_i[base + e]; }
  __syncthreads();
  for (int pick = 0; pick < NP; ++pick) {
    float best; int besti, slot;
    block_argbest(sv, si, NC, rv, ri, rs, best, besti, slot);
    if (threadIdx.x == 0) {
      if (slot >= 0) si[slot] = 0x7fffffff;
      (bank ? nb2 : nb1)[(size_t)b * NP + pick] = besti;
      (bank ? sim2 : sim1)[(size_t)b * NP + pick] = best;
    }
    __syncthreads();
  }
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

}  // namespace

// workspace: similarity rows S [2][B][n_data] f32 | candidates | query rows [2][B][128] + norms [2][B]
static inline size_t topk_cand_bytes(int B) { return (size_t)B * 2 * TOPK_SPLIT * TOPK_MAX * (sizeof(float) + sizeof(int)); }
size_t ph_crd_bank_topk_workspace_bytes(int B, int n_data) {
  return (size_t)2 * B * n_data * sizeof(float) + topk_cand_bytes(B) + (size_t)2 * B * (D + 1) * sizeof(float) + 256;
}

int ph_crd_bank_topk(const float* mem1, const float* mem2, const int* labels, const int64_t* idx, int PK,
                     const int64_t* batch_label, int B, int n_data, int num_pos, int feat_dim, int64_t* nb1,
                     int64_t* nb2, float* sim1, float* sim2, void* workspace, hipStream_t st) {
  if (feat_dim != D || num_pos < 1 || num_pos > TOPK_MAX || !workspace || B < 1 || n_data < 1) return PH_EINVAL;
  if (B > 128) {
    // the similarity kernel holds up to 128 normalised queries in LDS: larger batches (the reference has no limit; a replica
    // batch of 256 is the north-star size) run in chunks of 128 queries through the same workspace, in stream order
    for (int c0 = 0; c0 < B; c0 += 128) {
      const int bc = B - c0 < 128 ? B - c0 : 128;
      const int rc = ph_crd_bank_topk(mem1, mem2, labels, idx + (size_t)c0 * PK, PK, batch_label + c0, bc, n_data, num_pos, feat_dim,
                                      nb1 + (size_t)c0 * num_pos, nb2 + (size_t)c0 * num_pos, sim1 + (size_t)c0 * num_pos,
                                      sim2 + (size_t)c0 * num_pos, workspace, st);
      if (rc) return rc;
    }
    return PH_OK;
  }
  float* S = reinterpret_cast<float*>(workspace);
  float* cv = S + (size_t)2 * B * n_data;
  int* ci = reinterpret_cast<int*>(cv + (size_t)B * 2 * TOPK_SPLIT * TOPK_MAX);
  float* qbuf = reinterpret_cast<float*>(ci + (size_t)B * 2 * TOPK_SPLIT * TOPK_MAX);
  float* qnorm = qbuf + (size_t)2 * B * D;
  void* tok = nullptr;
  if (ph_prof_on())   // algorithmic bytes: every row of both banks once + the row labels + the 2 x B x num_pos results
    ph_prof_begin(PH_CLS_CRD_TOPK, 2.0 * n_data * D * 4 + 4.0 * n_data + 2.0 * B * num_pos * 12, st, &tok);
  hipLaunchKernelGGL(crd_topk_queries_kernel, dim3(B, 2), dim3(64), 0, st, mem1, mem2, idx, PK, qbuf, qnorm);
  PH_LAUNCH_CHECK();
  {
    const int nq = cdiv(B, 32);      // <= 4 (chunked above)
    const int ntiles = cdiv(n_data, 32);
    int gx = cdiv(ntiles, 4);
    const int cus = ph_num_cus();
    if (gx > cus / 2) gx = cus / 2;     // one workgroup per CU over the two banks; a wave walks its tiles
    const size_t lds = (size_t)(nq * 32 + 4 * 32) * SIM_RS * sizeof(float);
#define PH_SIM_LAUNCH(N)                                                                                                    \
  do {                                                                                                                      \
    static bool done = false;                                                                                               \
    if (!done) {                                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(crd_bank_sim_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds) != hipSuccess) return PH_ELAUNCH;                                                   \
      done = true;                                                                                                          \
    }                                                                                                                       \
    hipLaunchKernelGGL(crd_bank_sim_kernel<N>, dim3(gx, 2), dim3(256), lds, st, mem1, mem2, labels, qbuf, qnorm, batch_label, B, \
                       n_data, S);                                                                                          \
  } while (0)
    if (nq == 1) PH_SIM_LAUNCH(1); else if (nq == 2) PH_SIM_LAUNCH(2); else if (nq == 3) PH_SIM_LAUNCH(3); else PH_SIM_LAUNCH(4);
#undef PH_SIM_LAUNCH
  }
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(crd_bank_topk_kernel, dim3(B, 2, TOPK_SPLIT), dim3(256), 0, st, S, B, n_data, cv, ci);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(crd_bank_topk_merge_kernel, dim3(B, 2), dim3(256), 0, st, cv, ci, num_pos, nb1, nb2, sim1, sim2);
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
  if (P2 > P || K2 > K || (!select_pos && P2 != P)) return PH_EINVAL;
  const size_t lds = (size_t)(P + K) * 4 + (size_t)P * 4;
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

int ph_crd_loss_grad(const float* xs, const float* xt, const int* sel, const int64_t* idx, const int64_t* idx_bank2,
                     const float* posw_s, const float* posw_t, const float* mem1, const float* mem2,
                     const float* params, float* lossp, float* dv1, float* dv2, int B, int PK, int P2, int K2,
                     int feat_dim, float n_data, float inv_bnorm, void* workspace, hipStream_t st) {
  if (feat_dim != D) return PH_EINVAL;
  int ns = workspace ? (P2 + K2) / 512 : 1;     // one workgroup per 512 columns of a sample, at most LG_SPLIT_MAX
  ns = ns < 1 ? 1 : (ns > LG_SPLIT_MAX ? LG_SPLIT_MAX : ns);
  void* tok = nullptr;
  if (ph_prof_on())   // algorithmic bytes: the selected rows of both banks + the two gradient rows per sample
    ph_prof_begin(PH_CLS_CRD_LOSSGRAD, 2.0 * B * (P2 + K2) * D * 4 + 2.0 * B * D * 4 + 2.0 * B * (P2 + K2) * 4, st, &tok);
  hipLaunchKernelGGL(crd_loss_grad_kernel, dim3(B, ns), dim3(1024), 0, st, xs, xt, sel, idx, idx_bank2 ? idx_bank2 : idx,
                     posw_s, posw_t, mem1, mem2, params, lossp, dv1, dv2, PK, P2, K2, n_data, inv_bnorm,
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
