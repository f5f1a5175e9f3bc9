// Relational distillation baselines of the reference's distiller zoo: PKT ("MIA 2022/distiller_zoo/
// PKT.py":17-46) and RKD ("MIA 2022/distiller_zoo/RKD.py":15-58), selected by `--distill pkt|rkd` in
// train_test_path_multi_distill_v2.py:339-342.  Both are functions of the pairwise geometry of one batch of feature rows;
// each entry computes the loss AND its gradient with respect to the student rows in closed form (the teacher rows are
// constants), in a fixed summation order (no atomics).  Small problems (B <= 128 rows of D <= 512): latency-bound.
#include "ph_common.h"
#include "ph_kernels.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// fixed-tree block reduction; every thread gets the total.  red: >= blockDim.x floats
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int tid = threadIdx.x, n = blockDim.x;
  red[tid] = v;
  __syncthreads();
  for (int o = n >> 1; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float t = red[0];
  __syncthreads();
  return t;
}

// ------------------------------------------------------------------------------------------------ PKT
// rows: y = x / (||x|| + eps) for the student and the teacher rows
__global__ void pkt_prep_kernel(const float* __restrict__ xs, const float* __restrict__ xt, float* __restrict__ s,
                                float* __restrict__ t, float* __restrict__ ns, int B, int D, float eps) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float a = 0.f, b = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float u = xs[(size_t)row * D + d], v = xt[(size_t)row * D + d];
    a += u * u; b += v * v;
  }
  a = sqrtf(wave_sum(a)); b = sqrtf(wave_sum(b));
  const float ia = 1.f / (a + eps), ib = 1.f / (b + eps);
  for (int d = lane; d < D; d += 64) {
    s[(size_t)row * D + d] = xs[(size_t)row * D + d] * ia;
    t[(size_t)row * D + d] = xt[(size_t)row * D + d] * ib;
  }
  if (lane == 0) ns[row] = a;
}

// one wave per row i of the similarity matrices S (student) and T (teacher):
//   P = M / rowsum(M), M = (S + 1) / 2 (Q from T alike); loss_i = sum_j Q log((Q + eps) / (P + eps));
//   H_ij = dL/dS_ij = ((g_ij - sum_k g_ik P_ik) / r_i) / 2 with g = -Q / (P + eps) / B^2
__global__ void pkt_rows_kernel(const float* __restrict__ S, const float* __restrict__ T, float* __restrict__ H,
                                float* __restrict__ loss_rows, int B, float eps) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float r = 0.f, rq = 0.f;
  for (int j = lane; j < B; j += 64) {
    r += (S[(size_t)row * B + j] + 1.f) * 0.5f;
    rq += (T[(size_t)row * B + j] + 1.f) * 0.5f;
  }
  r = wave_sum(r); rq = wave_sum(rq);
  const float inv_bb = 1.f / ((float)B * (float)B);
  float li = 0.f, c = 0.f;
  for (int j = lane; j < B; j += 64) {
    const float p = (S[(size_t)row * B + j] + 1.f) * 0.5f / r, q = (T[(size_t)row * B + j] + 1.f) * 0.5f / rq;
    li += q * logf((q + eps) / (p + eps));
    c += -q / (p + eps) * inv_bb * p;
  }
  li = wave_sum(li); c = wave_sum(c);
  for (int j = lane; j < B; j += 64) {
    const float p = (S[(size_t)row * B + j] + 1.f) * 0.5f / r, q = (T[(size_t)row * B + j] + 1.f) * 0.5f / rq;
    const float g = -q / (p + eps) * inv_bb;
    H[(size_t)row * B + j] = 0.5f * (g - c) / r;
  }
  if (lane == 0) loss_rows[row] = li;
}

// one workgroup: loss = sum(loss_rows) / B^2;  dx_i = ds_i / (n_i + eps) - x_i (x_i . ds_i) / (n_i (n_i + eps)^2)
__global__ __launch_bounds__(1024) void pkt_finish_kernel(const float* __restrict__ xs, const float* __restrict__ ds,
                                                          const float* __restrict__ ns, const float* __restrict__ loss_rows,
                                                          float* __restrict__ dx, float* __restrict__ loss, int B, int D,
                                                          float eps) {
  __shared__ float red[1024];
  float l = 0.f;
  for (int i = threadIdx.x; i < B; i += 1024) l += loss_rows[i];
  l = block_sum(l, red);
  if (threadIdx.x == 0) loss[0] = l / ((float)B * (float)B);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int row = wave; row < B; row += 16) {
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += xs[(size_t)row * D + d] * ds[(size_t)row * D + d];
    dot = wave_sum(dot);
    const float n = ns[row], ne = n + eps;
    const float k = n > 0.f ? dot / (n * ne * ne) : 0.f;
    for (int d = lane; d < D; d += 64) dx[(size_t)row * D + d] = ds[(size_t)row * D + d] / ne - xs[(size_t)row * D + d] * k;
  }
}

// ------------------------------------------------------------------------------------------------ RKD
// pairwise distances d_ij = sqrt(max(|x_i|^2 + |x_j|^2 - 2 x_i.x_j, eps)), 0 on the diagonal (RKD.py:47-58), from the
// Gram matrix; `act` marks the entries whose clamp is inactive (they carry gradient).  One workgroup.
__global__ __launch_bounds__(1024) void rkd_dist_kernel(const float* __restrict__ gs, const float* __restrict__ gt,
                                                        float* __restrict__ W, float* __restrict__ out, int B, float w_d) {
  // gs / gt: Gram matrices x x^T of the student / the teacher.  W_ij = dL_d / dd_ij / d_ij * (clamp inactive), so that
  // dx_i = 2 sum_j W_ij (x_i - x_j);  out[0] = w_d * loss_d
  __shared__ float red[1024];
  const int tid = threadIdx.x, n = B * B;
  const float eps = 1e-12f;
  float sd = 0.f, st = 0.f;
  for (int e = tid; e < n; e += 1024) {
    const int i = e / B, j = e % B;
    if (i == j) continue;
    sd += sqrtf(fmaxf(gs[i * B + i] + gs[j * B + j] - 2.f * gs[e], eps));
    st += sqrtf(fmaxf(gt[i * B + i] + gt[j * B + j] - 2.f * gt[e], eps));
  }
  const float N = (float)B * (float)(B - 1);
  const float md = block_sum(sd, red) / N, mt = block_sum(st, red) / N;
  float l = 0.f, ud = 0.f;
  for (int e = tid; e < n; e += 1024) {
    const int i = e / B, j = e % B;
    if (i == j) continue;
    const float d = sqrtf(fmaxf(gs[i * B + i] + gs[j * B + j] - 2.f * gs[e], eps)) / md;
    const float t = sqrtf(fmaxf(gt[i * B + i] + gt[j * B + j] - 2.f * gt[e], eps)) / mt;
    const float z = d - t, az = fabsf(z);
    l += az < 1.f ? 0.5f * z * z : az - 0.5f;
    ud += fminf(fmaxf(z, -1.f), 1.f) * d;
  }
  const float inv_n2 = 1.f / (float)n;
  l = block_sum(l, red) * inv_n2;
  const float corr = block_sum(ud, red) * inv_n2 / N;       // (1/N) sum_kl u_kl dhat_kl
  for (int e = tid; e < n; e += 1024) {
    const int i = e / B, j = e % B;
    float w = 0.f;
    if (i != j) {
      const float res = gs[i * B + i] + gs[j * B + j] - 2.f * gs[e];
      if (res > eps) {
        const float dr = sqrtf(res);
        const float t = sqrtf(fmaxf(gt[i * B + i] + gt[j * B + j] - 2.f * gt[e], eps)) / mt;
        const float u = fminf(fmaxf(dr / md - t, -1.f), 1.f) * inv_n2;
        w = w_d * (u - corr) / md / dr;
      }
    }
    W[e] = w;
  }
  if (tid == 0) out[0] = w_d * l;
}

// angle term, one workgroup per anchor i (RKD.py:33-43): e_ij = normalize(x_j - x_i), A_i = E_i E_i^T for the teacher
// and the student, smooth-L1 over all B^3 entries.  LDS: E [B][D+1] and A [B][B+1].  Writes the anchor's loss partial
// and dv[i][j][:] = dL/d(x_j - x_i); rkd_gather_kernel adds the anchors up.
__global__ __launch_bounds__(1024) void rkd_angle_kernel(const float* __restrict__ xs, const float* __restrict__ xt,
                                                         float* __restrict__ dv, float* __restrict__ part, int B, int D,
                                                         float w_a) {
  extern __shared__ float sm[];
  const int LDE = D + 1, LDA = B + 1;
  float* E = sm;                 // [B][LDE]
  float* A = sm + B * LDE;       // [B][LDA]  teacher angles, then U = clip(z) / B^3
  float* nv = A + B * LDA;       // [B]       ||x_j - x_i||
  __shared__ float red[1024];
  const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  auto load_dirs = [&](const float* x) {
    for (int j = wave; j < B; j += 16) {
      float s = 0.f;
      for (int d = lane; d < D; d += 64) {
        const float v = x[(size_t)j * D + d] - x[(size_t)i * D + d];
        E[j * LDE + d] = v;
        s += v * v;
      }
      s = sqrtf(wave_sum(s));
      const float inv = 1.f / fmaxf(s, 1e-12f);
      for (int d = lane; d < D; d += 64) E[j * LDE + d] *= inv;
      if (lane == 0) nv[j] = s;
    }
  };
  load_dirs(xt);
  __syncthreads();
  for (int e = tid; e < B * B; e += 1024) {
    const int j = e / B, k = e % B;
    float a = 0.f;
    for (int d = 0; d < D; ++d) a += E[j * LDE + d] * E[k * LDE + d];
    A[j * LDA + k] = a;
  }
  __syncthreads();
  load_dirs(xs);
  __syncthreads();
  const float inv_n3 = 1.f / ((float)B * (float)B * (float)B);
  float l = 0.f;
  for (int e = tid; e < B * B; e += 1024) {
    const int j = e / B, k = e % B;
    float a = 0.f;
    for (int d = 0; d < D; ++d) a += E[j * LDE + d] * E[k * LDE + d];
    const float z = a - A[j * LDA + k], az = fabsf(z);
    l += az < 1.f ? 0.5f * z * z : az - 0.5f;
    A[j * LDA + k] = fminf(fmaxf(z, -1.f), 1.f) * inv_n3;
  }
  l = block_sum(l, red);     // (also orders the writes of U before the reads below)
  if (tid == 0) part[i] = w_a * l * inv_n3;
  // G_j = 2 sum_k U_jk e_ik ;  dv_ij = (G_j - e_ij (e_ij . G_j)) / ||v_ij||   (zero for j == i: v_ii == 0 identically)
  for (int j = wave; j < B; j += 16) {
    float g[8];                                   // D <= 512: 8 elements per lane
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int d = lane + 64 * q;
      float a = 0.f;
      if (d < D) {
        for (int k = 0; k < B; ++k) a += A[j * LDA + k] * E[k * LDE + d];
        a *= 2.f * w_a;
        dot += a * E[j * LDE + d];
      }
      g[q] = a;
    }
    dot = wave_sum(dot);
    const float n = nv[j];
    const float inv = (j != i && n > 1e-12f) ? 1.f / n : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int d = lane + 64 * q;
      if (d < D) dv[((size_t)i * B + j) * D + d] = (g[q] - E[j * LDE + d] * dot) * inv;
    }
  }
}

// dx_j = sum_i dv[i][j] - sum_k dv[j][k]  (v_ij = x_j - x_i)  + the distance term 2 sum_k W_jk (x_j - x_k);
// loss = out[0] (distance part, already there) + sum_i part[i]
__global__ void rkd_gather_kernel(const float* __restrict__ dv, const float* __restrict__ W, const float* __restrict__ xs,
                                  const float* __restrict__ part, float* __restrict__ dx, float* __restrict__ out, int B,
                                  int D) {
  const int j = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float a = 0.f;
    for (int i = 0; i < B; ++i) a += dv[((size_t)i * B + j) * D + d] - dv[((size_t)j * B + i) * D + d];
    const float xj = xs[(size_t)j * D + d];
    float b = 0.f;
    for (int k = 0; k < B; ++k) b += W[j * B + k] * (xj - xs[(size_t)k * D + d]);
    dx[(size_t)j * D + d] = a + 2.f * b;
  }
  if (j == 0 && threadIdx.x == 0) {
    float l = out[0];
    for (int i = 0; i < B; ++i) l += part[i];
    out[0] = l;
  }
}

// ------------------------------------------------------------------------------------------------ Cox
// Negative partial log-likelihood of the survival task (MICCAI-2022/utils.py:361-376, after cox-nnet):
//   S_i = sum_j [t_j >= t_i] exp(theta_j);  loss = -mean_i c_i (theta_i - log S_i)
//   d loss / d theta_k = -(c_k - exp(theta_k) sum_i c_i [t_k >= t_i] / S_i) / B
// The reference fills the B x B risk-set matrix with a Python double loop on the host; one workgroup here (B <= 4096).
__global__ __launch_bounds__(1024) void cox_kernel(const float* __restrict__ theta, const float* __restrict__ t,
                                                   const float* __restrict__ c, float* __restrict__ loss,
                                                   float* __restrict__ dtheta, int B) {
  extern __shared__ float sm[];
  float* e = sm; float* tt = sm + B; float* w = tt + B;       // w_i = c_i / S_i
  __shared__ float red[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < B; i += 1024) { e[i] = expf(theta[i]); tt[i] = t[i]; }
  __syncthreads();
  float l = 0.f;
  for (int i = tid; i < B; i += 1024) {
    float s = 0.f;
    for (int j = 0; j < B; ++j) s += tt[j] >= tt[i] ? e[j] : 0.f;
    w[i] = c[i] / s;
    l += (theta[i] - logf(s)) * c[i];
  }
  l = block_sum(l, red);
  if (tid == 0) loss[0] = -l / (float)B;
  if (dtheta) {
    for (int k = tid; k < B; k += 1024) {
      float a = 0.f;
      for (int i = 0; i < B; ++i) a += tt[k] >= tt[i] ? w[i] : 0.f;
      dtheta[k] = -(c[k] - e[k] * a) / (float)B;
    }
  }
}

}  // namespace

#include "pathomic_hip.h"

extern "C" {

size_t ph_pkt_workspace_bytes(int B, int D) { return ((size_t)3 * B * D + (size_t)3 * B * B + 2 * B + 16) * sizeof(float); }

int ph_pkt_loss_grad(const float* f_s, const float* f_t, float* loss, float* dx, int B, int D, void* ws_, hipStream_t st) {
  if (!f_s || !f_t || !loss || !dx || !ws_ || B < 2 || D < 1) return PH_EINVAL;
  const float eps = 1e-7f;
  float* ws = reinterpret_cast<float*>(ws_);
  float* s = ws; float* t = s + (size_t)B * D; float* ds = t + (size_t)B * D;
  float* S = ds + (size_t)B * D; float* T = S + (size_t)B * B; float* H = T + (size_t)B * B;
  float* ns = H + (size_t)B * B; float* lr = ns + B;
  const int rows_per_block = 4;
  hipLaunchKernelGGL(pkt_prep_kernel, dim3((B + rows_per_block - 1) / rows_per_block), dim3(64 * rows_per_block), 0, st,
                     f_s, f_t, s, t, ns, B, D, eps);
  PH_LAUNCH_CHECK();
  int rc;
  // S = s s^T, T = t t^T
  if ((rc = ph_sgemm(s, s, nullptr, S, B, B, D, D, 1, 1, D, B, 0, 0, st))) return rc;
  if ((rc = ph_sgemm(t, t, nullptr, T, B, B, D, D, 1, 1, D, B, 0, 0, st))) return rc;
  hipLaunchKernelGGL(pkt_rows_kernel, dim3((B + rows_per_block - 1) / rows_per_block), dim3(64 * rows_per_block), 0, st,
                     S, T, H, lr, B, eps);
  PH_LAUNCH_CHECK();
  // ds = H s + H^T s
  if ((rc = ph_sgemm(H, s, nullptr, ds, B, D, B, B, 1, D, 1, D, 0, 0, st))) return rc;
  if ((rc = ph_sgemm(H, s, nullptr, ds, B, D, B, 1, B, D, 1, D, 0, 1, st))) return rc;
  hipLaunchKernelGGL(pkt_finish_kernel, dim3(1), dim3(1024), 0, st, f_s, ds, ns, lr, dx, loss, B, D, eps);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

size_t ph_rkd_workspace_bytes(int B, int D) {
  return ((size_t)B * B * D + (size_t)3 * B * B + B + 16) * sizeof(float);
}

int ph_rkd_loss_grad(const float* f_s, const float* f_t, float* loss, float* dx, int B, int D, float w_d, float w_a,
                     void* ws_, hipStream_t st) {
  if (!f_s || !f_t || !loss || !dx || !ws_ || B < 2 || B > 128 || D < 1 || D > 512) return PH_EINVAL;
  const size_t lds = ((size_t)B * (D + 1) + (size_t)B * (B + 1) + B) * sizeof(float);
  if (lds > 150 * 1024) return PH_EINVAL;
  float* ws = reinterpret_cast<float*>(ws_);
  float* dv = ws; float* gs = dv + (size_t)B * B * D; float* gt = gs + (size_t)B * B; float* W = gt + (size_t)B * B;
  float* part = W + (size_t)B * B;
  int rc;
  if ((rc = ph_sgemm(f_s, f_s, nullptr, gs, B, B, D, D, 1, 1, D, B, 0, 0, st))) return rc;
  if ((rc = ph_sgemm(f_t, f_t, nullptr, gt, B, B, D, D, 1, 1, D, B, 0, 0, st))) return rc;
  hipLaunchKernelGGL(rkd_dist_kernel, dim3(1), dim3(1024), 0, st, gs, gt, W, loss, B, w_d);
  PH_LAUNCH_CHECK();
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(rkd_angle_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            150 * 1024) != hipSuccess)
      return PH_ELAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(rkd_angle_kernel, dim3(B), dim3(1024), lds, st, f_s, f_t, dv, part, B, D, w_a);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(rkd_gather_kernel, dim3(B), dim3(128), 0, st, dv, W, f_s, part, dx, loss, B, D);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_cox_loss_grad(const float* theta, const float* survtime, const float* censor, float* loss, float* dtheta, int B,
                     hipStream_t st) {
  if (!theta || !survtime || !censor || !loss || B < 1 || B > 4096) return PH_EINVAL;
  hipLaunchKernelGGL(cox_kernel, dim3(1), dim3(1024), (size_t)3 * B * sizeof(float), st, theta, survtime, censor, loss, dtheta, B);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

}  // extern "C"
