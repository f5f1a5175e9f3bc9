// GK-Refine gradient-agreement weights (reference train_test_path_multi_distill.py:41-70) and the fused
// multi-tensor Adam + EMA update (networks_new.py:85 torch.optim.Adam with L2-in-grad weight decay;
// train_test_path_multi_distill.py:34-38 update_ema_variables).  HBM-bound streaming kernels.
#include "ph_common.h"
#include "ph_dense.h"
#include "ph_kernels.h"

namespace {

// G[NG][n] row-major -> gram[NG*NG] = G G^T (single block, fixed summation order: deterministic)
template <int NG>
__global__ __launch_bounds__(1024) void gram_kernel(const float* __restrict__ G, float* __restrict__ gram, int n) {
  float acc[NG * (NG + 1) / 2];
#pragma unroll
  for (int i = 0; i < NG * (NG + 1) / 2; ++i) acc[i] = 0.f;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    float v[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) v[i] = G[(size_t)i * n + e];
    int q = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i)
#pragma unroll
      for (int j = i; j < NG; ++j) acc[q++] += v[i] * v[j];
  }
  __shared__ float sh[16][NG * (NG + 1) / 2];
#pragma unroll
  for (int q = 0; q < NG * (NG + 1) / 2; ++q) {
    const float s = wave_sum(acc[q]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][q] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int q = 0;
    for (int i = 0; i < NG; ++i)
      for (int j = i; j < NG; ++j) {
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w][q];
        gram[i * NG + j] = s;
        gram[j * NG + i] = s;
        ++q;
      }
  }
}

// scale_i = sum_j gram_ij * mult / (sqrt(gram_ii) sqrt(gram_jj));  total = sum_{i<nl} scale_i * losses_i
// (AEKD_loss :59-68: mult = len(loss_t_list); scale[:-1] . losses)
__global__ void gk_scale_kernel(const float* __restrict__ gram, const float* const* __restrict__ losses, int ng,
                                int nl, float mult, float* __restrict__ scale, float* total) {
  if (threadIdx.x != 0) return;
  float t = 0.f;
  for (int i = 0; i < ng; ++i) {
    float s = 0.f;
    for (int j = 0; j < ng; ++j) s += gram[i * ng + j] * mult / (sqrtf(gram[i * ng + i]) * sqrtf(gram[j * ng + j]));
    scale[i] = s;
    if (i < nl) t += s * (*losses[i]);
  }
  if (total) *total = t;
}

// GK-Refine tail of loss_head.py in one launch: scale_i (as gk_scale_kernel) from the Gram matrix of the five gradients
// in the order [div1, div2, CE, kd1, kd2]; w = scale * coef + add (the per-term factors alpha / beta and lambda_nll);
// total = w . losses; scaled = losses * logc (the values the trainer logs); scale_ext = scale in the reference's order
// [div1, div2, kd1, kd2, CE].
__global__ void gk_finish_kernel(const float* __restrict__ gram, const float* __restrict__ losses,
                                 const float* __restrict__ coef, const float* __restrict__ add,
                                 const float* __restrict__ logc, float mult, float* __restrict__ scale_int,
                                 float* __restrict__ w, float* __restrict__ total, float* __restrict__ scaled,
                                 float* __restrict__ scale_ext) {
  if (threadIdx.x != 0) return;
  const int ng = 5;
  float t = 0.f, sc[5];
  for (int i = 0; i < ng; ++i) {
    float s = 0.f;
    for (int j = 0; j < ng; ++j) s += gram[i * ng + j] * mult / (sqrtf(gram[i * ng + i]) * sqrtf(gram[j * ng + j]));
    sc[i] = s;
    scale_int[i] = s;
    const float wi = add[i] + s * coef[i];
    w[i] = wi;
    t += wi * losses[i];
    scaled[i] = losses[i] * logc[i];
  }
  *total = t;
  scale_ext[0] = sc[0]; scale_ext[1] = sc[1]; scale_ext[2] = sc[3]; scale_ext[3] = sc[4]; scale_ext[4] = sc[2];
}

// The same for the MIA-2022 trainer's momentum GK-Refine ("MIA 2022/train_test_path_multi_distill_v2.py":89-132, 436-477): the
// Gram arrives in the loss head's internal order [div1, div2, CE, kd1, kd2]; the weights are the row sums of the cosine matrix in
// the trainer's order [div1, div2, kd1, kd2, CE] (optionally binarised with thresh), EMA-ed into the persistent state; the loss
// is lam * CE + mult * sum_i state_i c_i L_i with c = (alpha, alpha, beta e, beta e) and e the epoch weight of the CRD terms
// (a device scalar: a captured graph reads the current value).
__global__ void gk_finish_momentum_kernel(const float* __restrict__ gram, const float* __restrict__ losses, float alpha,
                                          float beta, const float* __restrict__ e_dev, float lam, float mult, int use_thresh,
                                          float thresh, float momentum, float* __restrict__ mo_scale, int* __restrict__ mo_init,
                                          float* __restrict__ w, float* __restrict__ total, float* __restrict__ scaled,
                                          float* __restrict__ scale_ext) {
  if (threadIdx.x != 0) return;
  const int ng = 5;
  const int ext2int[5] = {0, 1, 3, 4, 2};
  const float e = e_dev ? e_dev[0] : 1.f;
  const float c[5] = {alpha, alpha, 1.f, beta * e, beta * e};      // internal order; CE unscaled
  const int first = mo_init ? (*mo_init == 0) : 1;
  float t = 0.f;
  for (int ie = 0; ie < ng; ++ie) {
    const int i = ext2int[ie];
    float s = 0.f;
    for (int je = 0; je < ng; ++je) {
      const int j = ext2int[je];
      float r = gram[i * ng + j] / (sqrtf(gram[i * ng + i]) * sqrtf(gram[j * ng + j]));
      if (use_thresh) r = r > thresh ? 1.f : 0.f;
      s += r;
    }
    const float st = first ? s : momentum * mo_scale[ie] + (1.f - momentum) * s;
    mo_scale[ie] = st;
    scale_ext[ie] = st;
    const float wi = i == 2 ? lam : mult * st * c[i];
    w[i] = wi;
    t += wi * losses[i];
    scaled[i] = losses[i] * c[i];
  }
  if (mo_init) *mo_init = 1;
  *total = t;
}

// Adam (torch.optim.Adam semantics) + optional EMA of the parameters, 4 elements per thread.
//   g' = g + wd*p ; m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2 ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
//   ema = alpha*ema + (1-alpha)*p_new
// (1 - beta1), (1 - beta2), (1 - alpha) arrive as the host's double-precision differences rounded to float once, which
// is what torch's `addcmul_(g, g, value=1 - beta2)` / `add_(p, alpha=1 - alpha)` pass: 1.f - 0.999f is 1.3e-5 off.
__global__ void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, float* __restrict__ ema, size_t n, float lr, float b1, float b2,
                                float omb1, float omb2, float eps, float wd, float bc1, float bc2_sqrt, float ema_alpha,
                                float om_alpha) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float step = lr / bc1;
  if (i + 4 <= n) {
    f32x4 P = *reinterpret_cast<f32x4*>(p + i), Gr = *reinterpret_cast<const f32x4*>(g + i);
    f32x4 M = *reinterpret_cast<f32x4*>(m + i), V = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = Gr[k] + wd * P[k];
      M[k] = b1 * M[k] + omb1 * gg;
      V[k] = b2 * V[k] + omb2 * gg * gg;
      P[k] -= step * M[k] / (sqrtf(V[k]) / bc2_sqrt + eps);
    }
    *reinterpret_cast<f32x4*>(p + i) = P;
    *reinterpret_cast<f32x4*>(m + i) = M;
    *reinterpret_cast<f32x4*>(v + i) = V;
    if (ema) {
      f32x4 E = *reinterpret_cast<f32x4*>(ema + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) E[k] = ema_alpha * E[k] + om_alpha * P[k];
      *reinterpret_cast<f32x4*>(ema + i) = E;
    }
  } else {
    for (size_t e = i; e < n; ++e) {
      const float gg = g[e] + wd * p[e];
      m[e] = b1 * m[e] + omb1 * gg;
      v[e] = b2 * v[e] + omb2 * gg * gg;
      p[e] -= step * m[e] / (sqrtf(v[e]) / bc2_sqrt + eps);
      if (ema) ema[e] = ema_alpha * ema[e] + om_alpha * p[e];
    }
  }
}

// same update, per-step scalars read from device memory (hyper = [lr, bc1, sqrt(bc2), ema_alpha]) so that the
// launch can live inside a captured HIP graph and be replayed with new values
__global__ void adam_ema_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                    float* __restrict__ v, float* __restrict__ ema, size_t n, float b1, float b2,
                                    float omb1, float omb2, float eps, float wd, const float* __restrict__ hyper) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float step = hyper[0] / hyper[1], bc2_sqrt = hyper[2], ema_alpha = hyper[3], om_alpha = hyper[4];
  // beta1 < 0 in the launch arguments: the betas live in device memory, hyper[8..11] = {beta1, 1 - beta1, beta2, 1 - beta2} (a
  // schedule that cycles beta1 - OneCycleLR - must reach a launch replayed from a captured graph)
  if (b1 < 0.f) { b1 = hyper[8]; omb1 = hyper[9]; b2 = hyper[10]; omb2 = hyper[11]; }
  if (i + 4 <= n) {
    f32x4 P = *reinterpret_cast<f32x4*>(p + i), Gr = *reinterpret_cast<const f32x4*>(g + i);
    f32x4 M = *reinterpret_cast<f32x4*>(m + i), V = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = Gr[k] + wd * P[k];
      M[k] = b1 * M[k] + omb1 * gg;
      V[k] = b2 * V[k] + omb2 * gg * gg;
      P[k] -= step * M[k] / (sqrtf(V[k]) / bc2_sqrt + eps);
    }
    *reinterpret_cast<f32x4*>(p + i) = P;
    *reinterpret_cast<f32x4*>(m + i) = M;
    *reinterpret_cast<f32x4*>(v + i) = V;
    if (ema) {
      f32x4 E = *reinterpret_cast<f32x4*>(ema + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) E[k] = ema_alpha * E[k] + om_alpha * P[k];
      *reinterpret_cast<f32x4*>(ema + i) = E;
    }
  } else {
    for (size_t e = i; e < n; ++e) {
      const float gg = g[e] + wd * p[e];
      m[e] = b1 * m[e] + omb1 * gg;
      v[e] = b2 * v[e] + omb2 * gg * gg;
      p[e] -= step * m[e] / (sqrtf(v[e]) / bc2_sqrt + eps);
      if (ema) ema[e] = ema_alpha * ema[e] + om_alpha * p[e];
    }
  }
}

// torch.optim.Adagrad (lr_decay = 0: networks_new.py:86-87 passes lr, weight_decay, initial_accumulator_value = 0.1) with the
// EMA copy fused like the Adam kernel: g' = g + wd p; sum += g'^2; p -= lr g' / (sqrt(sum) + eps).  hyper as above ([0] = lr,
// [3] / [4] = the EMA rate and its complement).
__global__ void adagrad_ema_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sum,
                                       float* __restrict__ ema, size_t n, float eps, float wd, const float* __restrict__ hyper) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float lr = hyper[0], ema_alpha = hyper[3], om_alpha = hyper[4];
  const size_t e1 = i + 4 <= n ? i + 4 : n;
  for (size_t e = i; e < e1; ++e) {
    const float gg = g[e] + wd * p[e];
    const float sacc = sum[e] + gg * gg;
    sum[e] = sacc;
    const float pn = p[e] - lr * gg / (sqrtf(sacc) + eps);
    p[e] = pn;
    if (ema) ema[e] = ema_alpha * ema[e] + om_alpha * pn;
  }
}

__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, size_t n, float alpha) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ema[i] = alpha * ema[i] + (1.f - alpha) * p[i];
}

// MIA-2022 / MIA-2023 form: cosine Gram without multiplier, optional binarisation (> thresh -> 1 else 0), optional
// EMA of the weights across iterations (mo = m*mo + (1-m)*scale; first call: mo = scale when *mo_init == 0)
__global__ void gk_scale2_kernel(const float* __restrict__ gram, int ng, int use_thresh, float thresh, float momentum,
                                 float* __restrict__ mo_scale, int* __restrict__ mo_init) {
  if (threadIdx.x != 0) return;
  const int first = mo_init ? (*mo_init == 0) : 1;
  for (int i = 0; i < ng; ++i) {
    float s = 0.f;
    for (int j = 0; j < ng; ++j) {
      float r = gram[i * ng + j] / (sqrtf(gram[i * ng + i]) * sqrtf(gram[j * ng + j]));
      if (use_thresh) r = r > thresh ? 1.f : 0.f;
      s += r;
    }
    mo_scale[i] = first ? s : momentum * mo_scale[i] + (1.f - momentum) * s;
  }
  if (mo_init) *mo_init = 1;
}

// MIA-2023 GK_refine_thresh (".../train_test_path_multi_distill.py":81-128): PER-SAMPLE cosine matrix of the ng
// gradients (sklearn cosine_similarity: zero-norm rows give 0), column sums of max(cos, 0) or of (cos > thresh).
// One wave per sample; G is [ng][B][D] with D = 128.
template <int NG>
__global__ __launch_bounds__(256) void gk_rows_kernel(const float* __restrict__ G, int B, int use_thresh, float thresh,
                                                      float* __restrict__ all_scale) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float v[NG][2];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    v[i][0] = G[((size_t)i * B + b) * 128 + lane];
    v[i][1] = G[((size_t)i * B + b) * 128 + 64 + lane];
  }
  float gram[NG][NG];
#pragma unroll
  for (int i = 0; i < NG; ++i)
#pragma unroll
    for (int j = i; j < NG; ++j) {
      const float s = wave_sum(v[i][0] * v[j][0] + v[i][1] * v[j][1]);
      gram[i][j] = s; gram[j][i] = s;
    }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const float den = sqrtf(gram[i][i]) * sqrtf(gram[j][j]);
        const float c = den > 0.f ? gram[i][j] / den : 0.f;
        s += use_thresh ? (c > thresh ? 1.f : 0.f) : (c > 0.f ? c : 0.f);
      }
      all_scale[(size_t)b * NG + j] = s;
    }
  }
}


// ---- L1 weight regulariser of define_reg (reference networks_new.py:93-108 -> utils.py:60-198: sum of |W| over a list
// of parameter tensors).  The parameters live in one flat fp32 buffer (train_step.FlatParams, zero padding between
// tensors adds nothing), so a term is a handful of contiguous segments: pass 1 writes one partial per block, pass 2
// adds them in a fixed order in double (bitwise reproducible); the backward adds coef * sign(W) into the flat gradient.
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ w, size_t n, float* __restrict__ parts) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += fabsf(w[i]);
  s = wave_sum(s);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void l1_finish_kernel(const float* __restrict__ parts, int nparts, float* out, int accumulate) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += (double)parts[i];
  s = wave_sum_d(s);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
    out[0] = accumulate ? out[0] + v : v;
  }
}
__global__ void l1_sign_axpy_kernel(const float* __restrict__ w, float* __restrict__ g, size_t n, const float* coef_dev, float coef) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float c = coef_dev ? coef * coef_dev[0] : coef;
  const float x = w[i];
  g[i] += x > 0.f ? c : (x < 0.f ? -c : 0.f);     // d|x|/dx = sgn(x), 0 at 0 (torch.abs backward)
}

}  // namespace

int ph_gk_rows(const float* G, int ng, int B, int D, int use_thresh, float thresh, float* all_scale, hipStream_t st) {
  if (D != 128) return PH_EINVAL;
  dim3 grid(cdiv(B, 4));
  switch (ng) {
    case 3: hipLaunchKernelGGL(gk_rows_kernel<3>, grid, dim3(256), 0, st, G, B, use_thresh, thresh, all_scale); break;
    case 5: hipLaunchKernelGGL(gk_rows_kernel<5>, grid, dim3(256), 0, st, G, B, use_thresh, thresh, all_scale); break;
    default: return PH_EINVAL;
  }
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_gk_scale_momentum(const float* gram, int ng, int use_thresh, float thresh, float momentum, float* mo_scale,
                         int* mo_init, hipStream_t st) {
  hipLaunchKernelGGL(gk_scale2_kernel, dim3(1), dim3(64), 0, st, gram, ng, use_thresh, thresh, momentum, mo_scale,
                     mo_init);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_gram(const float* G, float* gram, int ng, int n, hipStream_t st) {
  switch (ng) {
    case 2: hipLaunchKernelGGL(gram_kernel<2>, dim3(1), dim3(1024), 0, st, G, gram, n); break;
    case 3: hipLaunchKernelGGL(gram_kernel<3>, dim3(1), dim3(1024), 0, st, G, gram, n); break;
    case 4: hipLaunchKernelGGL(gram_kernel<4>, dim3(1), dim3(1024), 0, st, G, gram, n); break;
    case 5: hipLaunchKernelGGL(gram_kernel<5>, dim3(1), dim3(1024), 0, st, G, gram, n); break;
    default: return PH_EINVAL;
  }
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_gk_scale(const float* gram, const float* const* losses, int ng, int nl, float mult, float* scale, float* total,
                hipStream_t st) {
  hipLaunchKernelGGL(gk_scale_kernel, dim3(1), dim3(64), 0, st, gram, losses, ng, nl, mult, scale, total);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_gk_finish(const float* gram, const float* losses, const float* coef, const float* add, const float* logc, float mult,
                 float* scale_int, float* w, float* total, float* scaled, float* scale_ext, hipStream_t st) {
  hipLaunchKernelGGL(gk_finish_kernel, dim3(1), dim3(64), 0, st, gram, losses, coef, add, logc, mult, scale_int, w, total,
                     scaled, scale_ext);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_gk_finish_momentum(const float* gram, const float* losses, float alpha, float beta, const float* e_dev, float lam,
                          float mult, int use_thresh, float thresh, float momentum, float* mo_scale, int* mo_init, float* w,
                          float* total, float* scaled, float* scale_ext, hipStream_t st) {
  if (!gram || !losses || !mo_scale || !w || !total || !scaled || !scale_ext) return PH_EINVAL;
  hipLaunchKernelGGL(gk_finish_momentum_kernel, dim3(1), dim3(64), 0, st, gram, losses, alpha, beta, e_dev, lam, mult,
                     use_thresh, thresh, momentum, mo_scale, mo_init, w, total, scaled, scale_ext);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_adam_ema_step(float* p, const float* g, float* m, float* v, float* ema, size_t n, double lr, double beta1,
                     double beta2, double eps, double weight_decay, int step, double ema_alpha, hipStream_t st) {
  // bias corrections in double, exactly like torch.optim.Adam's Python-side scalars
  const float bc1 = (float)(1.0 - pow(beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)step));
  const size_t nt = (n + 3) / 4;
  hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, p, g, m, v, ema, n, (float)lr,
                     (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, bc1,
                     bc2s, (float)ema_alpha, (float)(1.0 - ema_alpha));
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_adam_ema_step_dev(float* p, const float* g, float* m, float* v, float* ema, size_t n, double beta1, double beta2,
                         double eps, double weight_decay, const float* hyper, hipStream_t st) {
  const size_t nt = (n + 3) / 4;
  void* tok = nullptr;
  if (ph_prof_on()) ph_prof_begin(PH_CLS_ADAM_EMA, (double)n * (ema ? 36.0 : 28.0), st, &tok);   // p, m, v r/w + g (+ ema r/w)
  hipLaunchKernelGGL(adam_ema_dev_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, p, g, m, v, ema, n,
                     (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay,
                     hyper);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_adagrad_ema_step_dev(float* p, const float* g, float* sum, float* ema, size_t n, double eps, double weight_decay,
                            const float* hyper, hipStream_t st) {
  if (!p || !g || !sum || !hyper) return PH_EINVAL;
  if (n == 0) return PH_OK;
  const size_t nt = (n + 3) / 4;
  hipLaunchKernelGGL(adagrad_ema_dev_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, p, g, sum, ema, n, (float)eps,
                     (float)weight_decay, hyper);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_ema_update(float* ema, const float* p, size_t n, float alpha, hipStream_t st) {
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ema, p, n, alpha);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_l1_sum(const float* w, size_t n, float* partials, float* out, int accumulate, hipStream_t st) {
  if (!w || !partials || !out) return PH_EINVAL;
  size_t nb = (n + 2047) / 2048;           // ~8 elements per thread; at most PH_L1_PARTIALS partial sums
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(l1_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, w, n, partials);
  hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(256), 0, st, partials, (int)nb, out, accumulate);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_l1_sign_axpy(const float* w, float* g, size_t n, const float* coef_dev, float coef, hipStream_t st) {
  if (!w || !g) return PH_EINVAL;
  if (n == 0) return PH_OK;
  hipLaunchKernelGGL(l1_sign_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, g, n, coef_dev, coef);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
