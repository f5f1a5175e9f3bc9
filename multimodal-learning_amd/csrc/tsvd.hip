// t-SVD low-rank constraint of the MIA-2022 stage-1 trainer ("MIA 2022/train_test_tSVD.py").
//
//   * ph_sqdiff_sum / ph_scaled_diff: the Frobenius penalty mu/2 * ||adj - aux||^2 and its gradient (:418-431).
//   * ph_tsvd_update_aux: the auxiliary-variable update called at :382-391.  Its source (my_utils/TSVD_update_aux.py)
//     is NOT in the reference repository, so this implements the call contract with the standard proximal operator
//     of the tensor nuclear norm - DFT along the view axis, singular-value soft-thresholding of every frequency
//     slice, inverse DFT - and is tested against oracle/variants.py:update_aux (parity unpinned).
//
// Soft-thresholding without an SVD: for a slice X with X^H X = V diag(lambda) V^H, the thresholded slice is
// X * P with P = V diag(max(1 - tau / sqrt(lambda), 0)) V^H.  The Hermitian B x B matrix X^H X is embedded as the
// real symmetric 2B x 2B matrix [[Re, -Im], [Im, Re]] (every eigenvalue twice; a matrix function of the embedding
// is the embedding of the matrix function) and diagonalised in LDS by parallel cyclic Jacobi rotations (round-robin
// pairing: n/2 disjoint rotations per step).  One workgroup per frequency slice k = 0 .. V/2 (the others are
// complex conjugates).  B <= 64.
#include "ph_common.h"
#include "ph_kernels.h"

namespace {

constexpr int TS_MAXB = 64, TS_MAXN = 2 * TS_MAXB, TS_LD = TS_MAXN + 1;

__global__ __launch_bounds__(1024) void sqdiff_sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, size_t n, float scale) {
  __shared__ float sh[1024];
  float s = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 1024) {
    const float d = a[i] - b[i];
    s += d * d;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {   // fixed tree: bitwise reproducible
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = scale * sh[0];
}

__global__ void scaled_diff_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                   const float* __restrict__ gscalar, float alpha, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = gscalar[0] * alpha * (a[i] - b[i]);
}

// frequency slice k of the DFT along the view axis: X_k[i][j] = sum_v adj[v][i][j] * exp(-2 pi i k v / V)
__device__ __forceinline__ void dft_elem(const float* __restrict__ adj, int V, size_t bb, size_t ij, int k, float& re,
                                         float& im) {
  re = 0.f; im = 0.f;
  for (int v = 0; v < V; ++v) {
    float s, c;
    sincospif(2.0f * (float)((k * v) % V) / (float)V, &s, &c);
    const float x = adj[(size_t)v * bb + ij];
    re += x * c;
    im -= x * s;
  }
}

__global__ __launch_bounds__(256) void tsvd_slice_kernel(const float* __restrict__ adj, float* __restrict__ yre,
                                                         float* __restrict__ yim, float* __restrict__ tnn_k, int V,
                                                         int B, float tau) {
  extern __shared__ float sm[];
  const int n = 2 * B, tid = threadIdx.x, k = blockIdx.x;
  float* S = sm;                         // [n][TS_LD]  symmetric matrix, later P
  float* W = sm + TS_MAXN * TS_LD;       // [n][TS_LD]  first the staged slice, then the eigenvectors
  __shared__ float rc[TS_MAXB], rs[TS_MAXB], fl[TS_MAXN];
  __shared__ int rp[TS_MAXB], rq[TS_MAXB];
  __shared__ float red[256];
  const size_t bb = (size_t)B * B;
  float* Xre = W;                        // [B][TS_LD]
  float* Xim = W + TS_MAXB * TS_LD;      // [B][TS_LD]
  auto stage_slice = [&]() {
    for (int e = tid; e < B * B; e += 256) {
      float re, im;
      dft_elem(adj, V, bb, (size_t)e, k, re, im);
      Xre[(e / B) * TS_LD + (e % B)] = re;
      Xim[(e / B) * TS_LD + (e % B)] = im;
    }
  };
  stage_slice();
  __syncthreads();
  // M = X^H X (Hermitian) -> real symmetric embedding S = [[Re M, -Im M], [Im M, Re M]]
  for (int e = tid; e < B * B; e += 256) {
    const int a = e / B, b = e % B;
    float mr = 0.f, mi = 0.f;
    for (int i = 0; i < B; ++i) {
      const float ar = Xre[i * TS_LD + a], ai = Xim[i * TS_LD + a], br = Xre[i * TS_LD + b], bi = Xim[i * TS_LD + b];
      mr += ar * br + ai * bi;      // conj(x_ia) * x_ib
      mi += ar * bi - ai * br;
    }
    S[a * TS_LD + b] = mr; S[(B + a) * TS_LD + B + b] = mr;
    S[(B + a) * TS_LD + b] = mi; S[a * TS_LD + B + b] = -mi;
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += 256) W[(e / n) * TS_LD + (e % n)] = (e / n == e % n) ? 1.f : 0.f;
  __syncthreads();
  // ---- parallel cyclic Jacobi
  const int half = n / 2;
  for (int sweep = 0; sweep < 12; ++sweep) {
    float offmax = 0.f;
    for (int step = 0; step < n - 1; ++step) {
      if (tid < half) {
        // round-robin pairing: r[0] = n-1 fixed, r[j] = (step + j - 1) mod (n-1); pair kk = (r[kk], r[n-1-kk])
        const int j0 = tid, j1 = n - 1 - tid;
        int p = j0 == 0 ? n - 1 : (step + j0 - 1) % (n - 1);
        int q = (step + j1 - 1) % (n - 1);
        if (p > q) { const int t = p; p = q; q = t; }
        const float apq = S[p * TS_LD + q], app = S[p * TS_LD + p], aqq = S[q * TS_LD + q];
        float c = 1.f, s = 0.f;
        if (fabsf(apq) > 1e-30f) {
          const float th = (aqq - app) / (2.f * apq);
          const float t = (th >= 0.f ? 1.f : -1.f) / (fabsf(th) + sqrtf(1.f + th * th));
          c = 1.f / sqrtf(1.f + t * t);
          s = t * c;
        }
        rp[tid] = p; rq[tid] = q; rc[tid] = c; rs[tid] = s;
        offmax = fmaxf(offmax, fabsf(apq));
      }
      __syncthreads();
      for (int e = tid; e < half * n; e += 256) {   // rows: S <- J^T S
        const int kk = e / n, j = e % n;
        const int p = rp[kk], q = rq[kk];
        const float c = rc[kk], s = rs[kk];
        const float sp = S[p * TS_LD + j], sq = S[q * TS_LD + j];
        S[p * TS_LD + j] = c * sp - s * sq;
        S[q * TS_LD + j] = s * sp + c * sq;
      }
      __syncthreads();
      for (int e = tid; e < half * n; e += 256) {   // columns: S <- S J, W <- W J
        const int kk = e / n, i = e % n;
        const int p = rp[kk], q = rq[kk];
        const float c = rc[kk], s = rs[kk];
        const float sp = S[i * TS_LD + p], sq = S[i * TS_LD + q];
        S[i * TS_LD + p] = c * sp - s * sq;
        S[i * TS_LD + q] = s * sp + c * sq;
        const float wp = W[i * TS_LD + p], wq = W[i * TS_LD + q];
        W[i * TS_LD + p] = c * wp - s * wq;
        W[i * TS_LD + q] = s * wp + c * wq;
      }
      __syncthreads();
    }
    // converged when the largest rotated off-diagonal element of the sweep is negligible against the diagonal
    red[tid] = offmax;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
      __syncthreads();
    }
    const float om = red[0];
    __syncthreads();
    float dm = 0.f;
    for (int i = tid; i < n; i += 256) dm = fmaxf(dm, fabsf(S[i * TS_LD + i]));
    red[tid] = dm;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
      __syncthreads();
    }
    const bool done = om <= 1e-7f * red[0];
    __syncthreads();
    if (done) break;
  }
  // ---- spectral function: f = max(1 - tau / sigma, 0), sigma = sqrt(lambda); nuclear norm of the thresholded slice
  float part = 0.f;
  for (int i = tid; i < n; i += 256) {
    const float lam = S[i * TS_LD + i];
    const float sig = lam > 0.f ? sqrtf(lam) : 0.f;
    fl[i] = sig > tau ? 1.f - tau / sig : 0.f;
    part += fmaxf(sig - tau, 0.f);
  }
  red[tid] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) tnn_k[k] = 0.5f * red[0];   // every eigenvalue of the embedding appears twice
  __syncthreads();
  // P = V f(Lambda) V^H: Re P = Pemb[0:B][0:B], Im P = Pemb[B:2B][0:B]  (written over S: only W and fl are read)
  float* Pre = S;
  float* Pim = S + TS_MAXB * TS_LD;
  for (int e = tid; e < B * B; e += 256) {
    const int a = e / B, b = e % B;
    float pr = 0.f, pi = 0.f;
    for (int m = 0; m < n; ++m) {
      const float wb = W[b * TS_LD + m] * fl[m];
      pr += W[a * TS_LD + m] * wb;
      pi += W[(B + a) * TS_LD + m] * wb;
    }
    Pre[a * TS_LD + b] = pr;
    Pim[a * TS_LD + b] = pi;
  }
  __syncthreads();
  // thresholded slice Y = X P  (the slice is staged again over the eigenvectors, which are no longer needed)
  stage_slice();
  __syncthreads();
  for (int e = tid; e < B * B; e += 256) {
    const int i = e / B, b = e % B;
    float yr = 0.f, yi = 0.f;
    for (int a = 0; a < B; ++a) {
      const float xr = Xre[i * TS_LD + a], xi = Xim[i * TS_LD + a], pr = Pre[a * TS_LD + b], pi = Pim[a * TS_LD + b];
      yr += xr * pr - xi * pi;
      yi += xr * pi + xi * pr;
    }
    yre[(size_t)k * bb + e] = yr;
    yim[(size_t)k * bb + e] = yi;
  }
}

// inverse DFT along the view axis from the slices 0 .. V/2 (slice V-k = conj(slice k)), and the tensor nuclear norm
__global__ void tsvd_idft_kernel(const float* __restrict__ yre, const float* __restrict__ yim,
                                 const float* __restrict__ tnn_k, float* __restrict__ aux, float* __restrict__ tnn,
                                 int V, int B) {
  const size_t bb = (size_t)B * B;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && tnn) {
    float t = tnn_k[0] + tnn_k[V / 2];
    for (int k = 1; k < V / 2; ++k) t += 2.f * tnn_k[k];
    tnn[0] = t / (float)V;
  }
  if (i >= bb * V) return;
  const int v = (int)(i / bb);
  const size_t ij = i % bb;
  float acc = yre[ij] + ((v & 1) ? -1.f : 1.f) * yre[(size_t)(V / 2) * bb + ij];
  for (int k = 1; k < V / 2; ++k) {
    float s, c;
    sincospif(2.0f * (float)((k * v) % V) / (float)V, &s, &c);
    acc += 2.f * (yre[(size_t)k * bb + ij] * c - yim[(size_t)k * bb + ij] * s);
  }
  aux[i] = acc / (float)V;
}

}  // namespace

#include "pathomic_hip.h"

size_t ph_tsvd_workspace_bytes(int V, int B) { return ((size_t)(V / 2 + 1) * 2 * B * B + 16) * sizeof(float); }

int ph_sqdiff_sum(const float* a, const float* b, float* out, size_t n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(sqdiff_sum_kernel, dim3(1), dim3(1024), 0, st, a, b, out, n, scale);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_scaled_diff(const float* a, const float* b, const float* gscalar, float alpha, float* out, size_t n,
                   hipStream_t st) {
  hipLaunchKernelGGL(scaled_diff_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, gscalar, alpha, out, n);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_tsvd_update_aux(const float* adj, float* aux, float* tnn, int V, int B, float tau, void* ws_, hipStream_t st) {
  if (!adj || !aux || !ws_ || V < 2 || V > 8 || (V & 1) || B < 1 || B > TS_MAXB) return PH_EINVAL;
  float* ws = reinterpret_cast<float*>(ws_);
  const size_t bb = (size_t)B * B;
  float* yre = ws;
  float* yim = ws + (size_t)(V / 2 + 1) * bb;
  float* tk = yim + (size_t)(V / 2 + 1) * bb;
  const int lds = 2 * TS_MAXN * TS_LD * (int)sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(tsvd_slice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds) != hipSuccess)
      return PH_ELAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(tsvd_slice_kernel, dim3(V / 2 + 1), dim3(256), lds, st, adj, yre, yim, tk, V, B, tau);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(tsvd_idft_kernel, dim3((unsigned)((bb * V + 255) / 256)), dim3(256), 0, st, yre, yim, tk, aux, tnn, V, B);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
