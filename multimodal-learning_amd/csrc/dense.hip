// Small dense fp32 operators of the heads / SNN / fusion / losses (a2-a7, a12 of SURVEY.md section 8).
// These carry < 0.02 % of the step's FLOPs; they are plain LDS-tiled fp32 VALU kernels written for
// correctness and low launch count, not MFMA.
#include "ph_common.h"
#include "ph_dense.h"

namespace {

// nn.BatchNorm1d in eval mode is y = x * gamma / sqrt(running_var + eps) + const: dx = g * (y > 0 if relu) * gamma * rsqrt(var + eps)
__global__ void bn1d_eval_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ gamma,
                                     const float* __restrict__ var, float* __restrict__ dx, size_t n, int C, float eps, int relu) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const float s = gamma[c] * rsqrtf(var[c] + eps);
  dx[i] = (relu && !(y[i] > 0.f)) ? 0.f : g[i] * s;
}


// ------------------------------------------------------------------ generic strided SGEMM
// C[m][n] (ldc) = act( sum_k A(m,k) * B(k,n) + bias[n] ) (+ C if accumulate)
//   A(m,k) = A[m*sam + k*sak],  B(k,n) = B[k*sbk + n*sbn]
#ifndef PH_BN1D_NC
#define PH_BN1D_NC 16      // channels per workgroup of the BatchNorm1d kernels (256 / PH_BN1D_NC row lanes)
#endif
constexpr int GT = 64, GK = 32;   // all global loads of a K step are issued before any is consumed
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                    const float* __restrict__ bias, float* __restrict__ Cm, int M,
                                                    int N, int K, long sam, long sak, long sbk, long sbn, long ldc,
                                                    int act, int accumulate) {
  __shared__ float As[GK][GT + 1];
  __shared__ float Bs[GK][GT + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  // loader mapping: pick the unit-stride direction of each operand as the fast thread index
  const bool a_kfast = (sak == 1), b_nfast = (sbn == 1);
  for (int k0 = 0; k0 < K; k0 += GK) {
    constexpr int PER = GT * GK / 256;
    float ra[PER], rb[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {          // phase 1: every load in flight together
      const int e = tid + q * 256;
      int m, k;
      if (a_kfast) { k = e % GK; m = e / GK; } else { m = e % GT; k = e / GT; }
      const int gm = m0 + m, gk = k0 + k;
      ra[q] = (gm < M && gk < K) ? A[gm * sam + gk * sak] : 0.f;
      int n, kk;
      if (b_nfast) { n = e % GT; kk = e / GT; } else { kk = e % GK; n = e / GK; }
      const int gn = n0 + n, gk2 = k0 + kk;
      rb[q] = (gn < N && gk2 < K) ? Bm[gk2 * sbk + gn * sbn] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {          // phase 2: LDS
      const int e = tid + q * 256;
      int m, k;
      if (a_kfast) { k = e % GK; m = e / GK; } else { m = e % GT; k = e / GT; }
      As[k][m] = ra[q];
      int n, kk;
      if (b_nfast) { n = e % GT; kk = e / GT; } else { kk = e % GK; n = e / GK; }
      Bs[kk][n] = rb[q];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + ty * 4 + i;
    if (gm >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + tx * 4 + j;
      if (gn >= N) continue;
      float v = acc[i][j] + (bias ? bias[gn] : 0.f);
      if (act == PH_ACT_RELU) v = v > 0.f ? v : 0.f;
      else if (act == PH_ACT_ELU) v = v > 0.f ? v : expm1f(v);
      else if (act == PH_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
      if (accumulate) v += Cm[gm * ldc + gn];
      Cm[gm * ldc + gn] = v;
    }
  }
}

// Small problems (the heads, the SNN, the embeds: at most 128 x 128 outputs, 32 + 12 launches per step on one or two
// 64 x 64 tiles): 16 x 16 output tiles so that 16x more workgroups share the K loop, K staged 128 deep with the next
// step's global loads issued before the current step's FMAs, K-contiguous LDS rows read as float4.
constexpr int ST = 16, SK = 128, SLD = SK + 4;
__global__ __launch_bounds__(256) void sgemm_small_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                          const float* __restrict__ bias, float* __restrict__ Cm, int M,
                                                          int N, int K, long sam, long sak, long sbk, long sbn, long ldc,
                                                          int act, int accumulate) {
  __shared__ __attribute__((aligned(16))) float As[ST][SLD];
  __shared__ __attribute__((aligned(16))) float Bs[ST][SLD];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * ST, n0 = blockIdx.x * ST;
  constexpr int PER = ST * SK / 256;
  const bool a_kfast = (sak == 1), b_nfast = (sbn == 1);
  float ra[PER], rb[PER];
  auto load_regs = [&](int k0) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = tid + q * 256;
      int m, k;
      if (a_kfast) { k = e % SK; m = e / SK; } else { m = e % ST; k = e / ST; }
      const int gm = m0 + m, gk = k0 + k;
      ra[q] = (gm < M && gk < K) ? A[gm * sam + gk * sak] : 0.f;
      int n, kk;
      if (b_nfast) { n = e % ST; kk = e / ST; } else { kk = e % SK; n = e / SK; }
      const int gn = n0 + n, gk2 = k0 + kk;
      rb[q] = (gn < N && gk2 < K) ? Bm[gk2 * sbk + gn * sbn] : 0.f;
    }
  };
  float acc = 0.f;
  load_regs(0);
  for (int k0 = 0; k0 < K; k0 += SK) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = tid + q * 256;
      int m, k;
      if (a_kfast) { k = e % SK; m = e / SK; } else { m = e % ST; k = e / ST; }
      As[m][k] = ra[q];
      int n, kk;
      if (b_nfast) { n = e % ST; kk = e / ST; } else { kk = e % SK; n = e / SK; }
      Bs[n][kk] = rb[q];
    }
    __syncthreads();
    if (k0 + SK < K) load_regs(k0 + SK);   // in flight during the FMAs below
#pragma unroll 8
    for (int k = 0; k < SK; k += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&As[ty][k]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[tx][k]);
      acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc); acc = fmaf(a[2], b[2], acc); acc = fmaf(a[3], b[3], acc);
    }
    __syncthreads();
  }
  const int gm = m0 + ty, gn = n0 + tx;
  if (gm < M && gn < N) {
    float v = acc + (bias ? bias[gn] : 0.f);
    if (act == PH_ACT_RELU) v = v > 0.f ? v : 0.f;
    else if (act == PH_ACT_ELU) v = v > 0.f ? v : expm1f(v);
    else if (act == PH_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
    if (accumulate) v += Cm[gm * ldc + gn];
    Cm[gm * ldc + gn] = v;
  }
}

// split-K variant for skinny problems (M,N small, K huge: the 16641-wide fusion encoder, bilinear gates)
__global__ __launch_bounds__(256) void sgemm_splitk_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                           float* __restrict__ part, int M, int N, int K, long sam,
                                                           long sak, long sbk, long sbn, int kchunk) {
  __shared__ float As[GK][GT + 1];
  __shared__ float Bs[GK][GT + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int kb = blockIdx.z * kchunk, ke = min(K, kb + kchunk);
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const bool a_kfast = (sak == 1), b_nfast = (sbn == 1);
  // (round 6) the loads of K step s + 1 are in flight while step s is multiplied: a workgroup has only ~5 steps (K / nsplit / GK),
  // and as load -> LDS -> multiply one after the other every step exposed a full memory round trip (16-18 us for the fusion
  // head's three 8.5 MB weight matrices on the fused teacher's tail)
  constexpr int PER = GT * GK / 256;
  float ra[PER], rb[PER];
  auto load_step = [&](int k0) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {          // every load in flight together
      const int e = tid + q * 256;
      int m, k;
      if (a_kfast) { k = e % GK; m = e / GK; } else { m = e % GT; k = e / GT; }
      const int gm = m0 + m, gk = k0 + k;
      ra[q] = (gm < M && gk < ke) ? A[gm * sam + gk * sak] : 0.f;
      int n, kk;
      if (b_nfast) { n = e % GT; kk = e / GT; } else { kk = e % GK; n = e / GK; }
      const int gn = n0 + n, gk2 = k0 + kk;
      rb[q] = (gn < N && gk2 < ke) ? Bm[gk2 * sbk + gn * sbn] : 0.f;
    }
  };
  if (kb < ke) load_step(kb);
  for (int k0 = kb; k0 < ke; k0 += GK) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {          // LDS
      const int e = tid + q * 256;
      int m, k;
      if (a_kfast) { k = e % GK; m = e / GK; } else { m = e % GT; k = e / GT; }
      As[k][m] = ra[q];
      int n, kk;
      if (b_nfast) { n = e % GT; kk = e / GT; } else { kk = e % GK; n = e / GK; }
      Bs[kk][n] = rb[q];
    }
    __syncthreads();
    if (k0 + GK < ke) load_step(k0 + GK);      // next step's operands travel while this one is multiplied
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* P = part + (size_t)blockIdx.z * M * N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + ty * 4 + i;
    if (gm >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + tx * 4 + j;
      if (gn < N) P[(size_t)gm * N + gn] = acc[i][j];
    }
  }
}

// four lanes per output element (each sums every fourth slab in a fixed order, then a fixed-order combine): with one thread per
// element the 64 x 128 outputs of the heads made 32 workgroups walk 128 slabs each (9.7 us)
__global__ void splitk_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                     float* __restrict__ Cm, int M, int N, long ldc, int nsplit, int act) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 2, q = t & 3;
  const bool live = i < M * N;
  float p = 0.f;
  if (live)
    for (int s = q; s < nsplit; s += 4) p += part[(size_t)s * M * N + i];
  // lanes 4 i .. 4 i + 3 hold the four interleaved partial sums p0..p3 of the one-thread form: (p0 + p1) + (p2 + p3), bitwise as before
  const float p1 = __shfl_xor(p, 1, 64);
  const float s01 = (q & 1) ? p1 + p : p + p1;      // lanes 0,1: p0 + p1; lanes 2,3: p2 + p3 (operand order fixed)
  const float s23 = __shfl_xor(s01, 2, 64);
  if (!live || q != 0) return;
  const int m = i / N, n = i % N;
  float v = (bias ? bias[n] : 0.f) + (s01 + s23);
  if (act == PH_ACT_RELU) v = v > 0.f ? v : 0.f;
  else if (act == PH_ACT_ELU) v = v > 0.f ? v : expm1f(v);
  else if (act == PH_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
  Cm[m * ldc + n] = v;
}

// ------------------------------------------------------------------ BatchNorm1d (train mode), thread per channel
// block = 64 channels x 4 row lanes (the batch rows are split over the lanes; fp64 partial sums combined in a fixed
// order).  One thread per channel walking the batch serially took 34 us for a [64 x 128] input.
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y,
                                                       float* mean, float* invstd, float* running_mean,
                                                       float* running_var, int64_t* nbt, int B, int C, float eps,
                                                       float momentum, int relu) {
  // (round 6) 16 channels x 16 row lanes per workgroup (was 64 x 4): C / 16 workgroups instead of C / 64 = 2 for the heads'
  // 128-wide BatchNorms, a quarter of the rows per thread - these launches sit on the latency-bound head chains
  constexpr int NC = PH_BN1D_NC, NR = 256 / PH_BN1D_NC;
  const int cl = threadIdx.x % NC, rl = threadIdx.x / NC;
  const int c = blockIdx.x * NC + cl;
  __shared__ double sh1[NR][NC], sh2[NR][NC];
  __shared__ float shsc[NC], shsh[NC];
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int b = rl; b < B; b += NR) { const double v = x[(size_t)b * C + c]; s1 += v; s2 += v * v; }
  sh1[rl][cl] = s1; sh2[rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) { s1 += sh1[r][cl]; s2 += sh2[r][cl]; }      // fixed order
    const double m = s1 / B;
    double var = s2 / B - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m; invstd[c] = is;
    const float sc = gamma[c] * is;
    shsc[cl] = sc; shsh[cl] = beta[c] - (float)m * sc;
    if (running_mean) {
      const double unb = B > 1 ? var * B / (B - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      if (c == 0 && nbt) *nbt += 1;
    }
  }
  __syncthreads();
  if (c < C) {
    const float sc = shsc[cl], shf = shsh[cl];
    for (int b = rl; b < B; b += NR) {
      float v = x[(size_t)b * C + c] * sc + shf;
      if (relu) v = v > 0.f ? v : 0.f;
      y[(size_t)b * C + c] = v;
    }
  }
}

// eval mode: normalise with the running statistics
__global__ void bn1d_eval_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ rm,
                                 const float* __restrict__ rv, float* __restrict__ y, int B, int C, float eps, int relu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int c = i % C;
  float v = (x[i] - rm[c]) * rsqrtf(rv[c] + eps) * gamma[c] + beta[c];
  if (relu) v = v > 0.f ? v : 0.f;
  y[i] = v;
}

// g: grad wrt the (relu'd) output y; dz = g * (y > 0) when relu
// block = 64 channels x 4 row lanes as in the forward (one thread per channel walking the batch twice took 50 us for
// [64 x 128]); fp64 partial sums of the lanes combined in a fixed order
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                       const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       float* __restrict__ dx, float* dgamma, float* dbeta, int B, int C,
                                                       int relu) {
  constexpr int NC = PH_BN1D_NC, NR = 256 / PH_BN1D_NC;      // (16 channels x 16 row lanes, as in the forward)
  const int cl = threadIdx.x % NC, rl = threadIdx.x / NC;
  const int c = blockIdx.x * NC + cl;
  __shared__ double sh1[NR][NC], sh2[NR][NC];
  const float mu = c < C ? mean[c] : 0.f, is = c < C ? invstd[c] : 0.f;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int b = rl; b < B; b += NR) {
      float dz = g[(size_t)b * C + c];
      if (relu && !(y[(size_t)b * C + c] > 0.f)) dz = 0.f;
      s1 += dz;
      s2 += (double)dz * ((x[(size_t)b * C + c] - mu) * is);
    }
  sh1[rl][cl] = s1; sh2[rl][cl] = s2;
  __syncthreads();
  if (c >= C) return;
  s1 = 0.0; s2 = 0.0;
#pragma unroll
  for (int r = 0; r < NR; ++r) { s1 += sh1[r][cl]; s2 += sh2[r][cl]; }      // fixed order, every lane the same value
  if (rl == 0) {
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
  }
  const float c1 = (float)(s1 / B), c2 = (float)(s2 / B), sc = gamma[c] * is;
  for (int b = rl; b < B; b += NR) {
    float dz = g[(size_t)b * C + c];
    if (relu && !(y[(size_t)b * C + c] > 0.f)) dz = 0.f;
    const float xh = (x[(size_t)b * C + c] - mu) * is;
    dx[(size_t)b * C + c] = sc * (dz - c1 - xh * c2);
  }
}

// ------------------------------------------------------------------ row ops on [B][C<=64]
__global__ void log_softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, x[b * C + c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += expf(x[b * C + c] - mx);
  const float l = mx + logf(s);
  for (int c = 0; c < C; ++c) y[b * C + c] = x[b * C + c] - l;
}
__global__ void log_softmax_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                       float* __restrict__ dx, int B, int C) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += g[b * C + c];
  for (int c = 0; c < C; ++c) dx[b * C + c] = g[b * C + c] - expf(y[b * C + c]) * s;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sh[i];
  __syncthreads();
  return t;
}

// loss = -(1/Bnorm) sum_b pred[b][grade[b]]
__global__ void nll_fwd_kernel(const float* __restrict__ pred, const int64_t* __restrict__ grade, float* loss, int B,
                               int C, float inv_bnorm) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) s -= pred[b * C + (int)grade[b]];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *loss = s * inv_bnorm;
}
__global__ void nll_bwd_kernel(const float* __restrict__ gs, const int64_t* __restrict__ grade, float* dpred, int B,
                               int C, float inv_bnorm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  dpred[i] = (c == (int)grade[b]) ? -(*gs) * inv_bnorm : 0.f;
}

// DistillKL (KD_loss.py:13-17): T^2/Bnorm * sum p_t (log p_t - log_softmax(y_s/T))
__global__ void kl_fwd_kernel(const float* __restrict__ ys, const float* __restrict__ yt, float* loss, int B, int C,
                              float T, float inv_bnorm) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float ms = -INFINITY, mt = -INFINITY;
    for (int c = 0; c < C; ++c) { ms = fmaxf(ms, ys[b * C + c] / T); mt = fmaxf(mt, yt[b * C + c] / T); }
    float ss = 0.f, st = 0.f;
    for (int c = 0; c < C; ++c) { ss += expf(ys[b * C + c] / T - ms); st += expf(yt[b * C + c] / T - mt); }
    const float ls = ms + logf(ss), lt = mt + logf(st);
    for (int c = 0; c < C; ++c) {
      const float lpt = yt[b * C + c] / T - lt, lps = ys[b * C + c] / T - ls;
      const float pt = expf(lpt);
      if (pt > 0.f) s += pt * (lpt - lps);
    }
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *loss = s * T * T * inv_bnorm;
}
// d loss / d ys = gs * T/Bnorm * (softmax(ys/T) - softmax(yt/T))
__global__ void kl_bwd_kernel(const float* __restrict__ gs, const float* __restrict__ ys,
                              const float* __restrict__ yt, float* __restrict__ dys, int B, int C, float T,
                              float inv_bnorm) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float ms = -INFINITY, mt = -INFINITY;
  for (int c = 0; c < C; ++c) { ms = fmaxf(ms, ys[b * C + c] / T); mt = fmaxf(mt, yt[b * C + c] / T); }
  float ss = 0.f, st = 0.f;
  for (int c = 0; c < C; ++c) { ss += expf(ys[b * C + c] / T - ms); st += expf(yt[b * C + c] / T - mt); }
  const float k = (*gs) * T * inv_bnorm;
  for (int c = 0; c < C; ++c)
    dys[b * C + c] = k * (expf(ys[b * C + c] / T - ms) / ss - expf(yt[b * C + c] / T - mt) / st);
}

// The logit-level part of the stage-2 loss block in ONE launch (loss_head.py): log-softmax of the student logits, the
// two DistillKL terms and the NLL, and d(each loss)/d(logits) for a unit upstream gradient - the same arithmetic, row
// by row and in the same order, as log_softmax / kl_fwd / kl_bwd / nll_fwd / nll_bwd / log_softmax_bwd above (which it
// replaces inside DistillStep: 8 dependent launches of ~5 us).  losses[0..2] = KL(t1), KL(t2), NLL; dl = [3][B][C].
__global__ __launch_bounds__(256) void logit_losses_kernel(const float* __restrict__ ys, const float* __restrict__ yt1,
                                                           const float* __restrict__ yt2,
                                                           const int64_t* __restrict__ grade, float* __restrict__ pred,
                                                           float* __restrict__ losses, float* __restrict__ dl, int B, int C,
                                                           float T, float inv_bnorm) {
  __shared__ float sh[16];
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float kk = T * inv_bnorm;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float* y = ys + (size_t)b * C;
    // log-softmax of the raw logits (log_softmax_kernel)
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, y[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(y[c] - mx);
    const float l = mx + logf(se);
    const int gb = (int)grade[b];
    for (int c = 0; c < C; ++c) pred[(size_t)b * C + c] = y[c] - l;
    s3 -= y[gb] - l;                                            // nll_fwd
    // d NLL / d logits = log_softmax_bwd(nll_bwd): g - exp(pred) * sum(g), g = -inv_bnorm at the label
    const float gsum = -inv_bnorm;
    for (int c = 0; c < C; ++c)
      dl[((size_t)2 * B + b) * C + c] = (c == gb ? -inv_bnorm : 0.f) - expf(y[c] - l) * gsum;
    // temperature softmax of the student (kl_*_kernel)
    float ms = -INFINITY;
    for (int c = 0; c < C; ++c) ms = fmaxf(ms, y[c] / T);
    float ss = 0.f;
    for (int c = 0; c < C; ++c) ss += expf(y[c] / T - ms);
    const float ls = ms + logf(ss);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float* t = (k ? yt2 : yt1) + (size_t)b * C;
      float mt = -INFINITY;
      for (int c = 0; c < C; ++c) mt = fmaxf(mt, t[c] / T);
      float st = 0.f;
      for (int c = 0; c < C; ++c) st += expf(t[c] / T - mt);
      const float lt = mt + logf(st);
      float acc = 0.f;
      for (int c = 0; c < C; ++c) {
        const float lpt = t[c] / T - lt, lps = y[c] / T - ls;
        const float pt = expf(lpt);
        if (pt > 0.f) acc += pt * (lpt - lps);
        dl[((size_t)k * B + b) * C + c] = kk * (expf(y[c] / T - ms) / ss - expf(t[c] / T - mt) / st);
      }
      if (k) s2 += acc; else s1 += acc;
    }
  }
  s1 = block_sum(s1, sh);
  __syncthreads();
  s2 = block_sum(s2, sh);
  __syncthreads();
  s3 = block_sum(s3, sh);
  if (threadIdx.x == 0) {
    losses[0] = s1 * T * T * inv_bnorm;
    losses[1] = s2 * T * T * inv_bnorm;
    losses[2] = s3 * inv_bnorm;
  }
}

// MIA-2023 DistillKL ("MIA 2023/stage2_unimodal_student/KD_loss.py":14-20): per-sample KL rows
__global__ void kl_rows_fwd_kernel(const float* __restrict__ ys, const float* __restrict__ yt,
                                   float* __restrict__ sample_loss, int B, int C, float T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float ms = -INFINITY, mt = -INFINITY;
  for (int c = 0; c < C; ++c) { ms = fmaxf(ms, ys[b * C + c] / T); mt = fmaxf(mt, yt[b * C + c] / T); }
  float ss = 0.f, st = 0.f;
  for (int c = 0; c < C; ++c) { ss += expf(ys[b * C + c] / T - ms); st += expf(yt[b * C + c] / T - mt); }
  const float ls = ms + logf(ss), lt = mt + logf(st);
  float s = 0.f;
  for (int c = 0; c < C; ++c) {
    const float lpt = yt[b * C + c] / T - lt, lps = ys[b * C + c] / T - ls;
    const float pt = expf(lpt);
    if (pt > 0.f) s += pt * (lpt - lps);
  }
  sample_loss[b] = s * T * T;
}
// d sample_loss_b / d ys[b] = T (softmax(ys/T) - softmax(yt/T)); g = upstream gradient per sample
__global__ void kl_rows_bwd_kernel(const float* __restrict__ g, const float* __restrict__ ys,
                                   const float* __restrict__ yt, float* __restrict__ dys, int B, int C, float T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float ms = -INFINITY, mt = -INFINITY;
  for (int c = 0; c < C; ++c) { ms = fmaxf(ms, ys[b * C + c] / T); mt = fmaxf(mt, yt[b * C + c] / T); }
  float ss = 0.f, st = 0.f;
  for (int c = 0; c < C; ++c) { ss += expf(ys[b * C + c] / T - ms); st += expf(yt[b * C + c] / T - mt); }
  const float k = g[b] * T;
  for (int c = 0; c < C; ++c)
    dys[b * C + c] = k * (expf(ys[b * C + c] / T - ms) / ss - expf(yt[b * C + c] / T - mt) / st);
}
// assign_sample_weights (".../train_test_path_multi_distill.py":131-158) on logits: conf = log p_gt - log max_{c != gt} p_c,
// discrepancy = min(max(conf_t - conf_s, 0), max_discrep)
__global__ void conf_discrepancy_kernel(const float* __restrict__ logit_s, const float* __restrict__ logit_t,
                                        const int64_t* __restrict__ gt, float* __restrict__ out, int B, int C,
                                        float max_discrep) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int g = (int)gt[b];
  float conf[2];
  for (int w = 0; w < 2; ++w) {
    const float* y = (w ? logit_t : logit_s) + b * C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, y[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(y[c] - mx);
    float top2 = 0.f;                                  // max over c != gt of p_c   (p * (1 - onehot) has 0 at gt)
    for (int c = 0; c < C; ++c) if (c != g) top2 = fmaxf(top2, expf(y[c] - mx) / s);
    conf[w] = logf(expf(y[g] - mx) / s) - logf(top2);
  }
  out[b] = fminf(fmaxf(conf[1] - conf[0], 0.f), max_discrep);
}

// ------------------------------------------------------------------ L2 normalise rows (Normalize, CRD_loss.py:276-279)
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ nrm, int B,
                                  int D) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = x[(size_t)row * D + d]; s += v * v; }
  s = wave_sum(s);
  const float n = sqrtf(s);
  for (int d = lane; d < D; d += 64) y[(size_t)row * D + d] = x[(size_t)row * D + d] / n;
  if (lane == 0) nrm[row] = n;
}
// y = x / (||x||_2 + eps) per row with the norm treated as a constant (orthogonal_loss.py:24-28: `.detach()` on the
// norm), inv[row] = 1 / (||x|| + eps) kept for the backward (a plain row scaling)
__global__ void row_invnorm_scale_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv,
                                         int B, int D, float eps) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = x[(size_t)row * D + d]; s += v * v; }
  s = wave_sum(s);
  const float r = 1.f / (sqrtf(s) + eps);
  for (int d = lane; d < D; d += 64) y[(size_t)row * D + d] = x[(size_t)row * D + d] * r;
  if (lane == 0) inv[row] = r;
}
__global__ void row_scale_kernel(const float* __restrict__ x, const float* __restrict__ r, float* __restrict__ y, int B,
                                 int D) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  const float f = r[row];
  for (int d = lane; d < D; d += 64) y[(size_t)row * D + d] = x[(size_t)row * D + d] * f;
}
__global__ void l2norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                  const float* __restrict__ nrm, float* __restrict__ dx, int B, int D) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += g[(size_t)row * D + d] * y[(size_t)row * D + d];
  s = wave_sum(s);
  const float inv = 1.f / nrm[row];
  for (int d = lane; d < D; d += 64)
    dx[(size_t)row * D + d] = (g[(size_t)row * D + d] - y[(size_t)row * D + d] * s) * inv;
}

// ------------------------------------------------------------------ elementwise helpers
__global__ void eltwise_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                               size_t n, int op) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v;
  switch (op) {
    case PH_EW_RELU: v = a[i] > 0.f ? a[i] : 0.f; break;
    case PH_EW_GATE: v = (1.f / (1.f + expf(-a[i]))) * b[i]; break;   // sigmoid(z) * h
    case PH_EW_RELU_BWD: v = b[i] > 0.f ? a[i] : 0.f; break;          // g * (y > 0)
    case PH_EW_ADD: v = a[i] + b[i]; break;
    case PH_EW_ELU_BWD: v = b[i] > 0.f ? a[i] : a[i] * (b[i] + 1.f); break;   // g * ELU'(x), from y = ELU(x)
    case PH_EW_MUL: v = a[i] * b[i]; break;
    default: v = a[i];
  }
  o[i] = v;
}

// o12[b][i*(D2+1)+j] = o1e[b][i] * o2e[b][j] with an implicit trailing 1 on both (fusion.py:56-58) when
// append_one, plain outer product otherwise (the nn.Bilinear operand, fusion.py:43,50)
__global__ void outer_kernel(const float* __restrict__ o1, const float* __restrict__ o2, float* __restrict__ o12,
                             int B, int D1, int D2, int append_one) {
  const int E1 = D1 + append_one, E2 = D2 + append_one;
  const size_t n = (size_t)B * E1 * E2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = (int)(i % E2);
  const int ii = (int)((i / E2) % E1);
  const int b = (int)(i / ((size_t)E1 * E2));
  const float a = ii < D1 ? o1[(size_t)b * D1 + ii] : 1.f;
  const float c = j < D2 ? o2[(size_t)b * D2 + j] : 1.f;
  o12[i] = a * c;
}

// counter-based RNG: one 32-bit draw per element from (seed, offset + index); splitmix64 finaliser
__device__ __forceinline__ float u01(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
// nn.Dropout / nn.AlphaDropout in training mode (fusion.py:22-32, networks_new.py:193-211)
__global__ void dropout_kernel(float* __restrict__ x, size_t n, float p, uint64_t seed, uint64_t offset, int alpha) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool keep = u01(seed, offset + i) >= p;
  if (!alpha) {
    x[i] = keep ? x[i] / (1.f - p) : 0.f;
  } else {
    const float ap = -1.7580993408473766f;
    const float a = rsqrtf((1.f - p) * (1.f + p * ap * ap));
    const float b = -a * ap * p;
    x[i] = a * (keep ? x[i] : ap) + b;
  }
}

// graph-replayable variant: the per-step part of the RNG counter lives in device memory
// backward of the (alpha-)dropout applied by dropout_dev_kernel with the same (seed, site_offset, step counter):
// g <- g * d(out)/d(in)   (plain: keep / (1-p); alpha: a * keep)
// (src may be dst: in place)
__global__ void dropout_bwd_dev_kernel(const float* src, float* g, size_t n, float p, uint64_t seed, uint64_t site_offset,
                                       const uint64_t* __restrict__ step_ctr, int alpha) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t ctr = (*step_ctr << 34) ^ (site_offset + i);
  const bool keep = u01(seed, ctr) >= p;
  const float v = src[i];
  if (!alpha) {
    g[i] = keep ? v / (1.f - p) : 0.f;
  } else {
    const float ap = -1.7580993408473766f;
    const float a = rsqrtf((1.f - p) * (1.f + p * ap * ap));
    g[i] = keep ? a * v : 0.f;
  }
}

// backward of y = sigmoid(z) * h  (fusion.py:44-45,51-52)
__global__ void gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ h,
                                float* __restrict__ dz, float* __restrict__ dh, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = 1.f / (1.f + expf(-z[i]));
  dz[i] = g[i] * h[i] * s * (1.f - s);
  dh[i] = g[i] * s;
}

// backward of outer_kernel: do1[b][i] = sum_j g[b][i*E2+j] * o2e[b][j], do2[b][j] = sum_i g[b][i*E2+j] * o1e[b][i]
// (o?e = o? with the implicit trailing 1 when append_one).  One block per batch row.
__global__ __launch_bounds__(256) void outer_bwd_kernel(const float* __restrict__ g, const float* __restrict__ o1,
                                                        const float* __restrict__ o2, float* __restrict__ do1,
                                                        float* __restrict__ do2, int D1, int D2, int append_one) {
  const int b = blockIdx.x, E1 = D1 + append_one, E2 = D2 + append_one;
  const float* gb = g + (size_t)b * E1 * E2;
  for (int i = threadIdx.x; i < D1; i += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < E2; ++j) s += gb[(size_t)i * E2 + j] * (j < D2 ? o2[(size_t)b * D2 + j] : 1.f);
    do1[(size_t)b * D1 + i] = s;
  }
  for (int j = threadIdx.x; j < D2; j += blockDim.x) {
    float s = 0.f;
    for (int i = 0; i < E1; ++i) s += gb[(size_t)i * E2 + j] * (i < D1 ? o1[(size_t)b * D1 + i] : 1.f);
    do2[(size_t)b * D2 + j] = s;
  }
}

// (src may be x: in place)
__global__ void dropout_dev_kernel(const float* src, float* x, size_t n, float p, uint64_t seed, uint64_t site_offset,
                                   const uint64_t* __restrict__ step_ctr, int alpha) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t ctr = (*step_ctr << 34) ^ (site_offset + i);
  const bool keep = u01(seed, ctr) >= p;
  const float v = src[i];
  if (!alpha) {
    x[i] = keep ? v / (1.f - p) : 0.f;
  } else {
    const float ap = -1.7580993408473766f;
    const float a = rsqrtf((1.f - p) * (1.f + p * ap * ap));
    const float b = -a * ap * p;
    x[i] = a * (keep ? v : ap) + b;
  }
}
__global__ void counter_inc_kernel(uint64_t* c) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1;
}

__global__ void sum_kernel(const float* __restrict__ x, float* out, int n, float scale) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *out = s * scale;
}

inline unsigned nblk(size_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }

}  // namespace

int ph_sgemm(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, long sam, long sak,
             long sbk, long sbn, long ldc, int act, int accumulate, hipStream_t st) {
  if ((long)M * N <= 128L * 128L) {
    hipLaunchKernelGGL(sgemm_small_kernel, dim3(cdiv(N, ST), cdiv(M, ST)), dim3(256), 0, st, A, B, bias, C, M, N, K, sam,
                       sak, sbk, sbn, ldc, act, accumulate);
    PH_LAUNCH_CHECK();
    return PH_OK;
  }
  dim3 grid(cdiv(N, GT), cdiv(M, GT));
  hipLaunchKernelGGL(sgemm_kernel, grid, dim3(256), 0, st, A, B, bias, C, M, N, K, sam, sak, sbk, sbn, ldc, act,
                     accumulate);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_sgemm_splitk(const float* A, const float* B, const float* bias, float* C, float* part, int nsplit, int M, int N,
                    int K, long sam, long sak, long sbk, long sbn, long ldc, int act, hipStream_t st) {
  int kchunk = cdiv(cdiv(K, nsplit), GK) * GK;
  nsplit = cdiv(K, kchunk);
  dim3 grid(cdiv(N, GT), cdiv(M, GT), nsplit);
  hipLaunchKernelGGL(sgemm_splitk_kernel, grid, dim3(256), 0, st, A, B, part, M, N, K, sam, sak, sbk, sbn, kchunk);
  hipLaunchKernelGGL(splitk_finish_kernel, dim3(nblk((size_t)M * N * 4)), dim3(256), 0, st, part, bias, C, M, N, ldc,
                     nsplit, act);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_bn1d_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* invstd,
                float* running_mean, float* running_var, int64_t* nbt, int B, int C, float eps, float momentum,
                int relu, hipStream_t st) {
  hipLaunchKernelGGL(bn1d_fwd_kernel, dim3(cdiv(C, PH_BN1D_NC)), dim3(256), 0, st, x, gamma, beta, y, mean, invstd,
                     running_mean, running_var, nbt, B, C, eps, momentum, relu);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_bn1d_eval_bwd(const float* g, const float* y, const float* gamma, const float* running_var, float* dx, int B, int C,
                     float eps, int relu, hipStream_t st) {
  if (!g || !y || !gamma || !running_var || !dx || B < 1 || C < 1) return PH_EINVAL;
  const size_t n = (size_t)B * C;
  hipLaunchKernelGGL(bn1d_eval_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, y, gamma, running_var, dx, n,
                     C, eps, relu);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_bn1d_eval(const float* x, const float* gamma, const float* beta, const float* running_mean,
                 const float* running_var, float* y, int B, int C, float eps, int relu, hipStream_t st) {
  hipLaunchKernelGGL(bn1d_eval_kernel, dim3(nblk((size_t)B * C)), dim3(256), 0, st, x, gamma, beta, running_mean,
                     running_var, y, B, C, eps, relu);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_bn1d_bwd(const float* g, const float* y, const float* x, const float* mean, const float* invstd,
                const float* gamma, float* dx, float* dgamma, float* dbeta, int B, int C, int relu, hipStream_t st) {
  hipLaunchKernelGGL(bn1d_bwd_kernel, dim3(cdiv(C, PH_BN1D_NC)), dim3(256), 0, st, g, y, x, mean, invstd, gamma, dx, dgamma,
                     dbeta, B, C, relu);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_log_softmax(const float* x, float* y, int B, int C, hipStream_t st) {
  hipLaunchKernelGGL(log_softmax_kernel, dim3(nblk(B, 64)), dim3(64), 0, st, x, y, B, C);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_log_softmax_bwd(const float* g, const float* y, float* dx, int B, int C, hipStream_t st) {
  hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(nblk(B, 64)), dim3(64), 0, st, g, y, dx, B, C);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_nll_fwd(const float* pred, const int64_t* grade, float* loss, int B, int C, float inv_bnorm, hipStream_t st) {
  hipLaunchKernelGGL(nll_fwd_kernel, dim3(1), dim3(256), 0, st, pred, grade, loss, B, C, inv_bnorm);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_nll_bwd(const float* gs, const int64_t* grade, float* dpred, int B, int C, float inv_bnorm, hipStream_t st) {
  hipLaunchKernelGGL(nll_bwd_kernel, dim3(nblk((size_t)B * C)), dim3(256), 0, st, gs, grade, dpred, B, C, inv_bnorm);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_logit_losses(const float* ys, const float* yt1, const float* yt2, const int64_t* grade, float* pred, float* losses,
                    float* dl, int B, int C, float T, float inv_bnorm, hipStream_t st) {
  if (C > 64) return PH_EINVAL;
  hipLaunchKernelGGL(logit_losses_kernel, dim3(1), dim3(256), 0, st, ys, yt1, yt2, grade, pred, losses, dl, B, C, T, inv_bnorm);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_kl_fwd(const float* ys, const float* yt, float* loss, int B, int C, float T, float inv_bnorm, hipStream_t st) {
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(1), dim3(256), 0, st, ys, yt, loss, B, C, T, inv_bnorm);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_kl_bwd(const float* gs, const float* ys, const float* yt, float* dys, int B, int C, float T, float inv_bnorm,
              hipStream_t st) {
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(nblk(B, 64)), dim3(64), 0, st, gs, ys, yt, dys, B, C, T, inv_bnorm);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_kl_rows_fwd(const float* ys, const float* yt, float* sample_loss, int B, int C, float T, hipStream_t st) {
  hipLaunchKernelGGL(kl_rows_fwd_kernel, dim3(nblk(B, 64)), dim3(64), 0, st, ys, yt, sample_loss, B, C, T);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_kl_rows_bwd(const float* g, const float* ys, const float* yt, float* dys, int B, int C, float T, hipStream_t st) {
  hipLaunchKernelGGL(kl_rows_bwd_kernel, dim3(nblk(B, 64)), dim3(64), 0, st, g, ys, yt, dys, B, C, T);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_conf_discrepancy(const float* logit_s, const float* logit_t, const int64_t* gt, float* out, int B, int C,
                        float max_discrep, hipStream_t st) {
  hipLaunchKernelGGL(conf_discrepancy_kernel, dim3(nblk(B, 64)), dim3(64), 0, st, logit_s, logit_t, gt, out, B, C,
                     max_discrep);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_l2norm_fwd(const float* x, float* y, float* nrm, int B, int D, hipStream_t st) {
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, x, y, nrm, B, D);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_row_invnorm_scale(const float* x, float* y, float* inv, int B, int D, float eps, hipStream_t st) {
  hipLaunchKernelGGL(row_invnorm_scale_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, x, y, inv, B, D, eps);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_row_scale(const float* x, const float* r, float* y, int B, int D, hipStream_t st) {
  hipLaunchKernelGGL(row_scale_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, x, r, y, B, D);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_l2norm_bwd(const float* g, const float* y, const float* nrm, float* dx, int B, int D, hipStream_t st) {
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, g, y, nrm, dx, B, D);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_eltwise(const float* a, const float* b, float* o, size_t n, int op, hipStream_t st) {
  hipLaunchKernelGGL(eltwise_kernel, dim3(nblk(n)), dim3(256), 0, st, a, b, o, n, op);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_outer(const float* o1, const float* o2, float* o12, int B, int D1, int D2, int append_one, hipStream_t st) {
  const size_t n = (size_t)B * (D1 + append_one) * (D2 + append_one);
  hipLaunchKernelGGL(outer_kernel, dim3(nblk(n)), dim3(256), 0, st, o1, o2, o12, B, D1, D2, append_one);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_dropout(float* x, size_t n, float p, uint64_t seed, uint64_t offset, int alpha, hipStream_t st) {
  if (p <= 0.f) return PH_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(nblk(n)), dim3(256), 0, st, x, n, p, seed, offset, alpha);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_dropout_dev(float* x, size_t n, float p, uint64_t seed, uint64_t site_offset, const uint64_t* step_ctr, int alpha,
                   hipStream_t st) {
  if (p <= 0.f) return PH_OK;
  hipLaunchKernelGGL(dropout_dev_kernel, dim3(nblk(n)), dim3(256), 0, st, x, x, n, p, seed, site_offset, step_ctr, alpha);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
// out of place (one launch instead of a copy + an in-place launch on the latency-bound head chains); p <= 0 copies
int ph_dropout_dev_to(const float* src, float* dst, size_t n, float p, uint64_t seed, uint64_t site_offset, const uint64_t* step_ctr,
                      int alpha, hipStream_t st) {
  if (!src || !dst) return PH_EINVAL;
  if (p <= 0.f) return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess ? PH_OK : PH_ELAUNCH;
  hipLaunchKernelGGL(dropout_dev_kernel, dim3(nblk(n)), dim3(256), 0, st, src, dst, n, p, seed, site_offset, step_ctr, alpha);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_dropout_bwd_dev_to(const float* src, float* dst, size_t n, float p, uint64_t seed, uint64_t site_offset,
                          const uint64_t* step_ctr, int alpha, hipStream_t st) {
  if (!src || !dst) return PH_EINVAL;
  if (p <= 0.f) return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess ? PH_OK : PH_ELAUNCH;
  hipLaunchKernelGGL(dropout_bwd_dev_kernel, dim3(nblk(n)), dim3(256), 0, st, src, dst, n, p, seed, site_offset, step_ctr, alpha);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_dropout_bwd_dev(float* g, size_t n, float p, uint64_t seed, uint64_t site_offset, const uint64_t* step_ctr,
                       int alpha, hipStream_t st) {
  if (p <= 0.f) return PH_OK;
  hipLaunchKernelGGL(dropout_bwd_dev_kernel, dim3(nblk(n)), dim3(256), 0, st, g, g, n, p, seed, site_offset, step_ctr, alpha);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_gate_bwd(const float* g, const float* z, const float* h, float* dz, float* dh, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(nblk(n)), dim3(256), 0, st, g, z, h, dz, dh, n);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_outer_bwd(const float* g, const float* o1, const float* o2, float* do1, float* do2, int B, int D1, int D2,
                 int append_one, hipStream_t st) {
  hipLaunchKernelGGL(outer_bwd_kernel, dim3(B), dim3(256), 0, st, g, o1, o2, do1, do2, D1, D2, append_one);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_counter_inc(uint64_t* ctr, hipStream_t st) {
  hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(64), 0, st, ctr);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_sum(const float* x, float* out, int n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, x, out, n, scale);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
