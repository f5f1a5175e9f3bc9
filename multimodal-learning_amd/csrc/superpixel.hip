// Superpixel attention masks of the MIA-2023 stage-1 trainer (reference
// "MIA 2023/stage1_multi_modal_teacher/train_test_MT_SP_Masking.py":77-98): given the gradient of the loss with respect to
// the image and to the omic vector,
//   * aggregate the image gradient per superpixel (the reference builds a one-hot [B, N, H*W] tensor, moves it to the
//     HOST and runs a bmm there), divide by the superpixel area, keep the Path_K superpixels with the largest mean and
//     return their union as a [B, H, W] mask;
//   * mark the omic entries that are >= the Omic_K-th largest gradient of their row.
// One workgroup per image: the sums are accumulated as 64-bit fixed-point integers by LDS atomics (order-independent,
// so the result is bitwise reproducible), scaled by a power of two chosen from the image's largest |gradient|.
#include "ph_common.h"
#include "ph_kernels.h"

namespace {

constexpr int SP_MAXN = 2048;

__global__ __launch_bounds__(1024) void superpixel_mask_kernel(const float* __restrict__ grad, const int64_t* __restrict__ sp,
                                                               float* __restrict__ mask, float* __restrict__ mean_out,
                                                               int C, int HW, int N, int K) {
  __shared__ long long sums[SP_MAXN];
  __shared__ int area[SP_MAXN];
  __shared__ float mean[SP_MAXN];
  __shared__ unsigned char sel[SP_MAXN];
  __shared__ float redv[1024];
  __shared__ int redi[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* g = grad + (size_t)b * C * HW;
  const int64_t* lab = sp + (size_t)b * HW;
  for (int n = tid; n < N; n += 1024) { sums[n] = 0; area[n] = 0; sel[n] = 0; }
  float mx = 0.f;
  for (int i = tid; i < C * HW; i += 1024) mx = fmaxf(mx, fabsf(g[i]));
  redv[tid] = mx;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) redv[tid] = fmaxf(redv[tid], redv[tid + o]);
    __syncthreads();
  }
  mx = redv[0];
  __syncthreads();
  // power-of-two scale: |g| * scale * (C * HW) < 2^62
  int ex = 0;
  if (mx > 0.f) {
    int e1, e2;
    frexpf(mx, &e1);
    frexpf((float)C * (float)HW, &e2);
    ex = 61 - e1 - e2;
  }
  const float hi = ldexpf(1.f, ex > 126 ? 126 : ex), lo = ldexpf(1.f, ex > 126 ? ex - 126 : 0);   // scale = hi * lo
  for (int p = tid; p < HW; p += 1024) {
    const int n = (int)lab[p];
    if (n < 0 || n >= N) continue;            // a label outside [0, N) belongs to no superpixel
    long long q = 0;
    for (int c = 0; c < C; ++c) q += (long long)rintf(g[(size_t)c * HW + p] * hi * lo);
    atomicAdd(reinterpret_cast<unsigned long long*>(&sums[n]), (unsigned long long)q);
    atomicAdd(&area[n], 1);
  }
  __syncthreads();
  for (int n = tid; n < N; n += 1024) {
    const float v = (float)((double)sums[n] / ((double)hi * (double)lo));
    mean[n] = v / ((float)area[n] + 1e-9f);
    if (mean_out) mean_out[(size_t)b * N + n] = mean[n];
  }
  __syncthreads();
  // K rounds of arg-max (largest mean first; the lowest index among equal means)
  for (int k = 0; k < K; ++k) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int n = tid; n < N; n += 1024)
      if (!sel[n] && (mean[n] > bv || (mean[n] == bv && n < bi))) { bv = mean[n]; bi = n; }
    redv[tid] = bv; redi[tid] = bi;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if (tid < o) {
        const float v2 = redv[tid + o]; const int i2 = redi[tid + o];
        if (v2 > redv[tid] || (v2 == redv[tid] && i2 < redi[tid])) { redv[tid] = v2; redi[tid] = i2; }
      }
      __syncthreads();
    }
    if (tid == 0 && redi[0] != 0x7fffffff) sel[redi[0]] = 1;
    __syncthreads();
  }
  for (int p = tid; p < HW; p += 1024) {
    const int n = (int)lab[p];
    mask[(size_t)b * HW + p] = (n >= 0 && n < N && sel[n]) ? 1.f : 0.f;
  }
}

// mask[b][i] = 1 iff x[b][i] >= the K-th largest entry of row b  (fewer than K entries are strictly greater)
__global__ void topk_threshold_mask_kernel(const float* __restrict__ x, float* __restrict__ mask, int B, int D, int K) {
  const int row = blockIdx.x;
  extern __shared__ float xs[];
  for (int i = threadIdx.x; i < D; i += blockDim.x) xs[i] = x[(size_t)row * D + i];
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const float v = xs[i];
    int gt = 0;
    for (int j = 0; j < D; ++j) gt += xs[j] > v;
    mask[(size_t)row * D + i] = gt < K ? 1.f : 0.f;
  }
}

// out[b][c][p] = x[b][c][p] * (1 - mask[b][p])   (train_test_MT_SP_Masking.py:201-202: the masked views of the inputs)
__global__ void apply_mask_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out,
                                  size_t n, int C, size_t P) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t b = i / (C * P), p = i % P;
  out[i] = x[i] * (1.f - mask[b * P + p]);
}

}  // namespace

#include "pathomic_hip.h"

extern "C" {

int ph_superpixel_mask(const float* grad_nchw, const int64_t* sp_mask, float* mask, float* mean_out, int B, int C, int H,
                       int W, int N, int K, hipStream_t st) {
  if (!grad_nchw || !sp_mask || !mask || B < 1 || C < 1 || N < 1 || N > SP_MAXN || K < 1 || K > N) return PH_EINVAL;
  hipLaunchKernelGGL(superpixel_mask_kernel, dim3(B), dim3(1024), 0, st, grad_nchw, sp_mask, mask, mean_out, C, H * W, N, K);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_topk_threshold_mask(const float* x, float* mask, int B, int D, int K, hipStream_t st) {
  if (!x || !mask || B < 1 || D < 1 || K < 1 || K > D || D > 16384) return PH_EINVAL;
  hipLaunchKernelGGL(topk_threshold_mask_kernel, dim3(B), dim3(256), (size_t)D * sizeof(float), st, x, mask, B, D, K);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_apply_mask(const float* x, const float* mask, float* out, int B, int C, size_t P, hipStream_t st) {
  if (!x || !mask || !out || B < 1 || C < 1 || P < 1) return PH_EINVAL;
  const size_t n = (size_t)B * C * P;
  hipLaunchKernelGGL(apply_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, mask, out, n, C, P);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

}  // extern "C"
