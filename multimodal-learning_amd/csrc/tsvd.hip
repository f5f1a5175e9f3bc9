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
// complex conjugates).  B <= 64; larger batches (up to 128) use the one-sided Jacobi kernel below.
#include "ph_common.h"
#include "ph_kernels.h"

constexpr int TSVD_MAX_V = 8;   // views admitted by ph_tsvd_update_aux (couples the workspace tail layout, see tsvd_slice_big_kernel)
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

// mixed-feature views of n_views = 6 / 8 (train_test_tSVD.py:305-307,334-363): out = wa * a / max(a) + wb * b / max(b)
__global__ __launch_bounds__(1024) void maxnorm_mix_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ out, size_t n, float wa, float wb) {
  __shared__ float ra[1024], rb[1024];
  float ma = -INFINITY, mb = -INFINITY;
  for (size_t i = threadIdx.x; i < n; i += 1024) { ma = fmaxf(ma, a[i]); mb = fmaxf(mb, b[i]); }
  ra[threadIdx.x] = ma; rb[threadIdx.x] = mb;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      ra[threadIdx.x] = fmaxf(ra[threadIdx.x], ra[threadIdx.x + o]);
      rb[threadIdx.x] = fmaxf(rb[threadIdx.x], rb[threadIdx.x + o]);
    }
    __syncthreads();
  }
  ma = ra[0]; mb = rb[0];
  for (size_t i = threadIdx.x; i < n; i += 1024) out[i] = wa * (a[i] / ma) + wb * (b[i] / mb);
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
                                                         int B, float tau, const float* __restrict__ tau_dev) {
  if (tau_dev) tau = tau_dev[0];      // (a captured graph reads the current threshold: mu grows every update)
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

// ---- B up to 128: one-sided (Hestenes) Jacobi on the complex slice itself.  The 2B x 2B embedding above needs 2 x 263 KB
// at B = 128; the slice is 128 KB.  Column pairs (p, q) of A (initially X) are rotated until all columns are mutually
// orthogonal: A = X V = U Sigma, so sigma_j = ||a_j|| and the thresholded slice is
//   Y = U max(Sigma - tau, 0) V^H = A diag(f_j / sigma_j^2) A^H X,   f_j = max(1 - tau / sigma_j, 0),
// which needs neither V nor U explicitly.  X, T = A^H X and Y go through the workspace.
// One workgroup (1024 threads = 64 teams of 16 lanes, one team per column pair) per frequency slice - the launch is
// V / 2 + 1 workgroups, so the time of a call is the time of ONE compute unit walking 127 rotation steps per sweep: the
// step body is what matters.  A lives in LDS column-major as interleaved (re, im) pairs; rows >= B are zero and stay
// zero, so the body has no bounds checks.  Lane l of a team owns the row pairs 2 (l + 16 m), m < 4: each ds_read_b128 /
// ds_write_b128 of a team covers 256 contiguous bytes.  The arithmetic is packed fp32 on (re, im) pairs (v_pk_fma_f32),
// the four dot products are reduced over the team's 16 lanes with DPP butterflies (quad_perm / row_half_mirror /
// row_mirror: both partners of an exchange add the same two numbers, so all 16 lanes hold bitwise the same sums and
// apply the same rotation).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
// Column stride exactly 1 KiB: the lane groups of a ds_read_b128 mix lanes of two teams (MI355X_MICROARCH.md, LDS), and with
// every column base = 0 (mod 256 B) the pieces of the two columns still fall on disjoint banks.  Everything the kernel keeps in
// LDS sits in the DYNAMIC region (statics in front of it would leave A off its 16-byte alignment: replayed accesses).
constexpr int TB_MAXB = 128, TB_CS = 2 * TB_MAXB;
constexpr int TB_LDS_FLOATS = TB_MAXB * TB_CS + TB_MAXB + 1024 + 4;
#ifndef TB_TOL2
#define TB_TOL2 1e-12f
#endif
#ifndef TB_MAX_SWEEPS
#define TB_MAX_SWEEPS 30
#endif

template <int CTRL>
__device__ __forceinline__ float dpp_xchg(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, the same bits in every lane
__device__ __forceinline__ float row16_sum(float x) {
  x += dpp_xchg<0xB1>(x);      // quad_perm [1,0,3,2]
  x += dpp_xchg<0x4E>(x);      // quad_perm [2,3,0,1]
  x += dpp_xchg<0x141>(x);     // row_half_mirror
  x += dpp_xchg<0x140>(x);     // row_mirror
  return x;
}
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

__global__ __launch_bounds__(1024) void tsvd_slice_big_kernel(const float* __restrict__ adj, float* __restrict__ yre,
                                                              float* __restrict__ yim, float* __restrict__ tre,
                                                              float* __restrict__ tim, float* __restrict__ tnn_k, int V,
                                                              int B, float tau, const float* __restrict__ tau_dev) {
  if (tau_dev) tau = tau_dev[0];
  extern __shared__ float sm[];
  float* A = sm;                         // [n][TB_CS]: column j at A + j * TB_CS, row r at floats 2r (re), 2r + 1 (im)
  float* dj = sm + TB_MAXB * TB_CS;      // [TB_MAXB]
  float* red = dj + TB_MAXB;             // [1024]
  int& rotated = *reinterpret_cast<int*>(red + 1024);
  const int tid = threadIdx.x, k = blockIdx.x;
  const int n = B + (B & 1);             // an odd B gets a zero column that never rotates
  const size_t bb = (size_t)B * B;
  float* Xre = yre + (size_t)k * bb;     // the slice, row-major; overwritten by Y at the end
  float* Xim = yim + (size_t)k * bb;
  float* Tre = tre + (size_t)k * bb;
  float* Tim = tim + (size_t)k * bb;
  for (int e = tid; e < TB_MAXB * TB_CS / 4; e += 1024) reinterpret_cast<v4f*>(A)[e] = v4f{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  float part = 0.f;
  for (int e = tid; e < B * B; e += 1024) {
    const int i = e / B, b = e % B;
    float re, im;
    dft_elem(adj, V, bb, (size_t)e, k, re, im);
    Xre[e] = re;
    Xim[e] = im;
    *reinterpret_cast<v2f*>(A + b * TB_CS + 2 * i) = v2f{re, im};
    part += re * re + im * im;
  }
  red[tid] = part;
  if (tid == 0) rotated = 0;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float nrm2 = red[0];
  const float col_floor = 1e-14f * nrm2;  // columns below the rounding noise of the slice are left alone (and dropped)
  __syncthreads();
  const int team = tid >> 4, l = tid & 15, half = n / 2;
  const int nm = (B + 31) >> 5;          // row chunks of 32 that hold data
  int sweeps_done = 0;
  // One rotation on the register copies of two columns (a = column p, b = column q); true if it rotated.
  auto rotate = [&](v4f (&a)[4], v4f (&b)[4]) -> bool {
    v2f saa = {0.f, 0.f}, sbb = {0.f, 0.f}, sab = {0.f, 0.f}, sax = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m < nm) {
        const v2f a0 = a[m].xy, a1 = a[m].zw, b0 = b[m].xy, b1 = b[m].zw;
        saa = pk_fma(a0, a0, saa); saa = pk_fma(a1, a1, saa);
        sbb = pk_fma(b0, b0, sbb); sbb = pk_fma(b1, b1, sbb);
        sab = pk_fma(a0, b0, sab); sab = pk_fma(a1, b1, sab);        // (ar br, ai bi)
        sax = pk_fma(a0, b0.yx, sax); sax = pk_fma(a1, b1.yx, sax);  // (ar bi, ai br)
      }
    }
    const float al = row16_sum(saa.x + saa.y), be = row16_sum(sbb.x + sbb.y);
    const float gr = row16_sum(sab.x + sab.y);      // conj(a_p) . a_q
    const float gi = row16_sum(sax.x - sax.y);
    const float g2 = gr * gr + gi * gi;
    if (!(fminf(al, be) > col_floor && g2 > TB_TOL2 * al * be)) return false;
    const float rg = __frsqrt_rn(g2);
    const float ze = (be - al) * (0.5f * rg);
    const float t = copysignf(1.f, ze) / (fabsf(ze) + sqrtf(1.f + ze * ze));
    const float c = __frsqrt_rn(1.f + t * t), sn = c * t;
    const float ser = sn * gr * rg, sei = sn * gi * rg;
    // a_p' = c a_p - s e^{-i phi} a_q ;  a_q' = s e^{i phi} a_p + c a_q ;  e^{i phi} = (gr + i gi) / |g|
    const v2f c2 = {c, c}, s2 = {ser, ser}, ns2 = {-ser, -ser}, k1 = {-sei, sei};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m < nm) {
        const v2f a0 = a[m].xy, a1 = a[m].zw, b0 = b[m].xy, b1 = b[m].zw;
        const v2f p0 = pk_fma(c2, a0, pk_fma(ns2, b0, k1 * b0.yx));
        const v2f p1 = pk_fma(c2, a1, pk_fma(ns2, b1, k1 * b1.yx));
        const v2f q0 = pk_fma(c2, b0, pk_fma(s2, a0, k1 * a0.yx));
        const v2f q1 = pk_fma(c2, b1, pk_fma(s2, a1, k1 * a1.yx));
        a[m] = v4f{p0.x, p0.y, p1.x, p1.y};
        b[m] = v4f{q0.x, q0.y, q1.x, q1.y};
      }
    }
    return true;
  };
  for (int sweep = 0; sweep < TB_MAX_SWEEPS; ++sweep) {
    ++sweeps_done;
#ifndef TB_NO_BIPARTITE
    if (n == TB_MAXB) {
      // n = 128 (BASELINE configs[3]): a recursive bipartite ordering.  Phase with sets of S = 128, 64, ... 2 columns: every
      // set pairs its lower half with its upper half in S / 2 steps (team i of the set keeps lower column i and meets
      // upper column (i + s) mod S / 2) - 64 + 32 + ... + 1 = 127 steps and every pair once, like the round robin below,
      // but a team's FIRST column stays in its registers for the whole phase: per step one column is read and (if it
      // rotated) written instead of two, i.e. half the LDS traffic of a step whose stores (~80 B / clk) weigh as much as
      // its arithmetic.
      for (int S = n; S >= 2; S >>= 1) {
        const int h = S >> 1, g = team / h, i = team - g * h;
        v4f* cp = reinterpret_cast<v4f*>(A + (g * S + i) * TB_CS) + l;
        v4f a[4], b[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = cp[16 * m];
        bool moved = false;
        for (int st = 0; st < h; ++st) {
          v4f* cq = reinterpret_cast<v4f*>(A + (g * S + h + ((i + st) & (h - 1))) * TB_CS) + l;
#pragma unroll
          for (int m = 0; m < 4; ++m) b[m] = cq[16 * m];
          if (rotate(a, b)) {
            moved = true;
#pragma unroll
            for (int m = 0; m < 4; ++m) cq[16 * m] = b[m];
          }
          __syncthreads();      // column q is read by another team in the next step
        }
        if (moved) {
#pragma unroll
          for (int m = 0; m < 4; ++m) cp[16 * m] = a[m];
          if (l == 0) rotated = 1;
        }
        __syncthreads();
      }
    } else
#endif
    for (int step = 0; step < n - 1; ++step) {
      if (team < half) {
        const int j1 = n - 1 - team;
        const int p = team == 0 ? n - 1 : (step + team - 1) % (n - 1);
        const int q = (step + j1 - 1) % (n - 1);
        v4f* cp = reinterpret_cast<v4f*>(A + p * TB_CS) + l;
        v4f* cq = reinterpret_cast<v4f*>(A + q * TB_CS) + l;
        v4f a[4], b[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m < nm) { a[m] = cp[16 * m]; b[m] = cq[16 * m]; }
        if (rotate(a, b)) {
#pragma unroll
          for (int m = 0; m < 4; ++m)
            if (m < nm) { cp[16 * m] = a[m]; cq[16 * m] = b[m]; }
          if (l == 0) rotated = 1;
        }
      }
      __syncthreads();
    }
    const int any = rotated;
    __syncthreads();
    if (tid == 0) rotated = 0;
    __syncthreads();
    if (!any) break;
  }
  // sigma_j = ||a_j||;  d_j = f_j / sigma_j^2;  nuclear norm of the thresholded slice
  part = 0.f;
  if (tid < n) {
    float a2 = 0.f;
    for (int r = 0; r < 2 * B; ++r) a2 += A[tid * TB_CS + r] * A[tid * TB_CS + r];
    const float sig = sqrtf(a2);
    const bool keep = a2 > 100.f * col_floor && sig > tau;
    dj[tid] = keep ? (1.f - tau / sig) / a2 : 0.f;
    part = keep ? sig - tau : 0.f;
  }
  red[tid] = part;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    tnn_k[k] = red[0];
    // diagnostic: sweeps this slice took (tests/bench_tsvd_gpu.py).  Slots 8.. of the 16-float tail: ph_tsvd_update_aux
    // admits V <= 8, i.e. k <= V / 2 <= 4 < 8, so the counters never meet the per-slice TNN values in slots 0 .. V / 2
    static_assert(TSVD_MAX_V / 2 + 1 <= 8 && 8 + TSVD_MAX_V / 2 + 1 <= 16, "sweep counters share the 16-float tail with the TNN values");
    tnn_k[8 + k] = (float)sweeps_done;
  }
  // T = D A^H X
  for (int e = tid; e < B * B; e += 1024) {
    const int j = e / B, b = e % B;
    float tr = 0.f, ti = 0.f;
    for (int i = 0; i < B; ++i) {
      const v2f av = *reinterpret_cast<const v2f*>(A + j * TB_CS + 2 * i);
      const float xr = Xre[(size_t)i * B + b], xi = Xim[(size_t)i * B + b];
      tr += av.x * xr + av.y * xi;
      ti += av.x * xi - av.y * xr;
    }
    Tre[e] = tr * dj[j];
    Tim[e] = ti * dj[j];
  }
  __threadfence_block();
  __syncthreads();
  // Y = A T
  for (int e = tid; e < B * B; e += 1024) {
    const int i = e / B, b = e % B;
    float yr = 0.f, yi = 0.f;
    for (int j = 0; j < B; ++j) {
      const v2f av = *reinterpret_cast<const v2f*>(A + j * TB_CS + 2 * i);
      const float tr = Tre[(size_t)j * B + b], ti = Tim[(size_t)j * B + b];
      yr += av.x * tr - av.y * ti;
      yi += av.x * ti + av.y * tr;
    }
    Xre[e] = yr;
    Xim[e] = yi;
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

size_t ph_tsvd_workspace_bytes(int V, int B) { return ((size_t)(V / 2 + 1) * 4 * B * B + 16) * sizeof(float); }

int ph_sqdiff_sum(const float* a, const float* b, float* out, size_t n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(sqdiff_sum_kernel, dim3(1), dim3(1024), 0, st, a, b, out, n, scale);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_maxnorm_mix(const float* a, const float* b, float* out, size_t n, float wa, float wb, hipStream_t st) {
  if (!a || !b || !out || n == 0) return PH_EINVAL;
  hipLaunchKernelGGL(maxnorm_mix_kernel, dim3(1), dim3(1024), 0, st, a, b, out, n, wa, wb);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_scaled_diff(const float* a, const float* b, const float* gscalar, float alpha, float* out, size_t n,
                   hipStream_t st) {
  hipLaunchKernelGGL(scaled_diff_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, gscalar, alpha, out, n);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

static int tsvd_update_aux_impl(const float* adj, float* aux, float* tnn, int V, int B, float tau, const float* tau_dev, void* ws_,
                                hipStream_t st);
int ph_tsvd_update_aux(const float* adj, float* aux, float* tnn, int V, int B, float tau, void* ws_, hipStream_t st) {
  return tsvd_update_aux_impl(adj, aux, tnn, V, B, tau, nullptr, ws_, st);
}
int ph_tsvd_update_aux_dev(const float* adj, float* aux, float* tnn, int V, int B, const float* tau_dev, void* ws_, hipStream_t st) {
  if (!tau_dev) return PH_EINVAL;
  return tsvd_update_aux_impl(adj, aux, tnn, V, B, 0.f, tau_dev, ws_, st);
}
static int tsvd_update_aux_impl(const float* adj, float* aux, float* tnn, int V, int B, float tau, const float* tau_dev, void* ws_,
                                hipStream_t st) {
  if (!adj || !aux || !ws_ || V < 2 || V > TSVD_MAX_V || (V & 1) || B < 1 || B > TB_MAXB) return PH_EINVAL;
  float* ws = reinterpret_cast<float*>(ws_);
  const size_t bb = (size_t)B * B;
  float* yre = ws;
  float* yim = ws + (size_t)(V / 2 + 1) * bb;
  float* tre = yim + (size_t)(V / 2 + 1) * bb;
  float* tim = tre + (size_t)(V / 2 + 1) * bb;
  float* tk = tim + (size_t)(V / 2 + 1) * bb;
  if (B > TS_MAXB) {
    const int ldsb = TB_LDS_FLOATS * (int)sizeof(float);
    static bool attr_big = false;
    if (!attr_big) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(tsvd_slice_big_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, ldsb) != hipSuccess)
        return PH_ELAUNCH;
      attr_big = true;
    }
    hipLaunchKernelGGL(tsvd_slice_big_kernel, dim3(V / 2 + 1), dim3(1024), ldsb, st, adj, yre, yim, tre, tim, tk, V, B, tau, tau_dev);
    PH_LAUNCH_CHECK();
    hipLaunchKernelGGL(tsvd_idft_kernel, dim3((unsigned)((bb * V + 255) / 256)), dim3(256), 0, st, yre, yim, tk, aux, tnn, V, B);
    PH_LAUNCH_CHECK();
    return PH_OK;
  }
  const int lds = 2 * TS_MAXN * TS_LD * (int)sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(tsvd_slice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds) != hipSuccess)
      return PH_ELAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(tsvd_slice_kernel, dim3(V / 2 + 1), dim3(256), lds, st, adj, yre, yim, tk, V, B, tau, tau_dev);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(tsvd_idft_kernel, dim3((unsigned)((bb * V + 255) / 256)), dim3(256), 0, st, yre, yim, tk, aux, tnn, V, B);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
