// Train-mode BatchNorm, activation, pooling and their backward passes (HBM-bound, 16-B vectorised).
// Reference semantics: nn.BatchNorm2d in training mode (resnets.py:148 et al.), ReLU, MaxPool2d(3,2,1)
// (resnets.py:150), AdaptiveAvgPool2d (resnets.py:158,234-236).  Activations are NHWC of type T
// (bf16 in perf mode, float in parity mode); statistics are always fp32 with fp64 combination.
#include "ph_common.h"
#include <cstdlib>
#include "ph_kernels.h"

namespace {

template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ x, T* __restrict__ x4, int B, int H, int W) {
  const size_t npix = (size_t)B * H * W;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const size_t hw = (size_t)H * W;
  const size_t b = i / hw, r = i - b * hw;
  const float c0 = x[(b * 3 + 0) * hw + r], c1 = x[(b * 3 + 1) * hw + r], c2 = x[(b * 3 + 2) * hw + r];
  if constexpr (is_hp<T>::value) {
    // NHWC4 half-pair planes [2][npix][4] fp16: plane 0 = hi, plane 1 = lo (the stem kernels stage them as they lie)
    f16x4 h, l;
    f16 a, b;
    hp_split(c0, a, b); h[0] = a; l[0] = b;
    hp_split(c1, a, b); h[1] = a; l[1] = b;
    hp_split(c2, a, b); h[2] = a; l[2] = b;
    h[3] = (f16)0.f; l[3] = (f16)0.f;
    f16* base = reinterpret_cast<f16*>(x4);
    *reinterpret_cast<f16x4*>(base + i * 4) = h;
    *reinterpret_cast<f16x4*>(base + (npix + i) * 4) = l;
  } else if constexpr (is_f32<T>::value) {
    f32x4 v = {c0, c1, c2, 0.f};
    *reinterpret_cast<f32x4*>(x4 + i * 4) = v;
  } else {
    bf16x4 v = {(bf16)c0, (bf16)c1, (bf16)c2, (bf16)0.f};
    *reinterpret_cast<bf16x4*>(x4 + i * 4) = v;
  }
}

// one block per channel: parts[nparts][2][C] -> mean / invstd / scale / shift (+ running stats)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ parts, int nparts, int C,
                                                          double count, float eps, float momentum,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* mean, float* invstd,
                                                          float* scale, float* shift, float* running_mean,
                                                          float* running_var, int64_t* nbt) {
  const int c = blockIdx.x, tid = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int p = tid; p < nparts; p += 256) {
    s1 += (double)parts[((size_t)p * 2 + 0) * C + c];
    s2 += (double)parts[((size_t)p * 2 + 1) * C + c];
  }
  __shared__ double sh[2][4];
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if ((tid & 63) == 0) { sh[0][tid >> 6] = s1; sh[1][tid >> 6] = s2; }
  __syncthreads();
  if (tid == 0) {
    s1 = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    s2 = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
    if (running_mean) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      if (c == 0 && nbt) *nbt += 1;
    }
  }
}

// eval mode (model.eval(), reference test() train_test_path_multi_distill.py:409-411): scale/shift of every BN unit
// of a network from its RUNNING statistics, one launch for all units
__global__ void bn_eval_params_kernel(PhBnEvalTable t, float eps) {
  const int u = blockIdx.x;
  for (int c = threadIdx.x; c < t.C[u]; c += blockDim.x) {
    const float is = rsqrtf(t.running_var[u][c] + eps);
    const float sc = t.gamma[u][c] * is;
    t.mean[u][c] = t.running_mean[u][c];
    t.invstd[u][c] = is;
    t.scale[u][c] = sc;
    t.shift[u][c] = t.beta[u][c] - t.running_mean[u][c] * sc;
  }
}

// out = relu?( y*scale + shift + [res | y_r*scale_r + shift_r | relu(y_r*scale_r + shift_r)] ), 8 channels per thread;
// relu bit0: ReLU on the sum, bit1: ReLU (and rounding to T) on the shortcut term
// TY: type of the convolution outputs (y, y_r) and of everything an ELEMENTWISE pass reads (res); T: type of the activation
// as the next convolution reads it (out).  They differ in PH_PREC_FP16X3 only: there `out` is the half-pair MFMA operand
// image and `out32` (optional) an fp32 copy of the same values for the elementwise readers (the next block's shortcut term,
// the backward's ReLU masks, the average pool) - the identity path of the network then carries fp32 like the reference's.
template <typename T, typename TY>
__global__ void bn_apply_kernel(const TY* __restrict__ y, const float* __restrict__ scale,
                                const float* __restrict__ shift, const TY* __restrict__ res,
                                const TY* __restrict__ y_r, const float* __restrict__ scale_r,
                                const float* __restrict__ shift_r, T* __restrict__ out, TY* __restrict__ out32, size_t n8,
                                int C8, int relu, int res_as_t) {
  // grid-stride loop: the stride (gridDim.x * 256 elements) is a multiple of C8, so a thread keeps its channel group and its
  // per-channel constants stay in registers (one element per thread spent 4-8 sixteen-byte parameter loads per 16-byte data
  // load); the next element's operands are requested before the current one is computed (with the constants hoisted but no
  // prefetch the first data load waited behind the parameter loads: two memory round trips per thread, 25 % slower steps)
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 >= n8) return;
  const size_t gstride = (size_t)gridDim.x * blockDim.x;
  const int c = (int)(i0 % C8) * 8;
  const bool use_res = res != nullptr, use_yr = !use_res && y_r != nullptr;
  auto fetch = [&](size_t i, float (&v)[8], float (&r)[8]) {
    load8(y + i * 8, v);
    if (use_res) {
      // res_as_t (PH_PREC_FP16X3, forward-only networks): the shortcut term is read from the half-pair operand image of the
      // block input (hi + lo 2^-11: 22-23 significant bits) - those networks then keep no fp32 copy of their block outputs
      if (!std::is_same<T, TY>::value && res_as_t) load8(reinterpret_cast<const T*>(res) + i * 8, r);
      else load8(res + i * 8, r);
    } else if (use_yr) {
      load8(y_r + i * 8, r);
    }
  };
  float v[8], r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = 0.f;
  fetch(i0, v, r);
  // (explicit 16-byte loads: written as eight indexed reads the hoisted constants became 32 single-dword loads per thread)
  float sc8[8], sh8[8], scr8[8], shr8[8];
  load8(scale + c, sc8);
  load8(shift + c, sh8);
#pragma unroll
  for (int k = 0; k < 8; ++k) { scr8[k] = 0.f; shr8[k] = 0.f; }
  if (use_yr) { load8(scale_r + c, scr8); load8(shift_r + c, shr8); }
  for (size_t i = i0;;) {
    const size_t inext = i + gstride;
    const bool more = inext < n8;
    float vn[8], rn[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { vn[k] = 0.f; rn[k] = 0.f; }
    if (more) fetch(inext, vn, rn);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], sc8[k], sh8[k]);      // (explicit: the in-LDS form of the tap-conv kernels must round alike)
    if (use_res) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += r[k];
    } else if (use_yr) {
      if (relu & 2) {
        // the shortcut is relu(bn(y_r)) (the stem's pooled RAW output feeding layer1.0, forward-only networks): rounded to the
        // activation type exactly as the separate pass that used to materialise it did, so the sum is bitwise the same
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float t = __builtin_fmaf(r[k], scr8[k], shr8[k]);
          if constexpr (is_hp<T>::value) v[k] += (t > 0.f ? t : 0.f);
          else v[k] += (float)(T)(t > 0.f ? t : 0.f);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += r[k] * scr8[k] + shr8[k];
      }
    }
    if (relu & 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
    }
    store8(out + i * 8, v);
    if constexpr (!std::is_same<T, TY>::value) { if (out32) store8(out32 + i * 8, v); }
    if (!more) break;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = vn[k]; r[k] = rn[k]; }
    i = inext;
  }
}

// stem: relu(bn(y0)) -> maxpool 3x3/2 pad 1; argmax position code (kh*3+kw, first max wins) saved as u8
template <typename T, typename TY, bool IDX>
__global__ void bn_relu_maxpool_kernel(const TY* __restrict__ y, const float* __restrict__ scale,
                                       const float* __restrict__ shift, T* __restrict__ out,
                                       uint8_t* __restrict__ idx, TY* __restrict__ raw, TY* __restrict__ out32, int B, int H,
                                       int W, int C8) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;   // floor((H + 2 - 3)/2) + 1
  const size_t n = (size_t)B * OH * OW * C8;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int cg = (int)(i % C8);
  size_t pix = i / C8;
  const int ow = (int)(pix % OW); pix /= OW;
  const int oh = (int)(pix % OH);
  const int b = (int)(pix / OH);
  const int c = cg * 8;
  float sc[8], sh[8], best[8], braw[8];
  int bi[8];
  load8(scale + c, sc);
  load8(shift + c, sh);
#pragma unroll
  for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = 0; braw[k] = 0.f; }
  // all nine window loads are requested unconditionally (coordinates clamped into the image, out-of-image taps skipped when
  // they are consumed): behind a branch per tap every load was its own memory round trip
  float win[9][8];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = oh * 2 - 1 + kh;
    const int ihc = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ow * 2 - 1 + kw;
      const int iwc = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
      load8(y + ((((size_t)b * H + ihc) * W + iwc) * C8 + cg) * 8, win[kh * 3 + kw]);
    }
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = oh * 2 - 1 + kh;
    if (ih < 0 || ih >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ow * 2 - 1 + kw;
      if (iw < 0 || iw >= W) continue;
      const float (&v)[8] = win[kh * 3 + kw];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        // (perf mode pools the unrounded activation and rounds the maximum once on store: rounding is monotone, so
        // the pooled value equals pooling the stored bf16 activation; only an exact-vs-rounded tie can pick another
        // - equally large after rounding - tap.  Rounding every tap cost a third of this VALU-bound kernel.)
        const float a = fmaxf(v[k] * sc[k] + sh[k], 0.f);
        if constexpr (IDX) { if (a > best[k]) { best[k] = a; bi[k] = kh * 3 + kw; braw[k] = v[k]; } }
        else best[k] = fmaxf(best[k], a);
      }
    }
  }
  store8(out + i * 8, best);
  if constexpr (!std::is_same<T, TY>::value) { if (out32) store8(out32 + i * 8, best); }
  if constexpr (IDX) {
    uint64_t packed = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) packed |= (uint64_t)(bi[k] & 0xff) << (8 * k);
    *reinterpret_cast<uint64_t*>(idx + i * 8) = packed;
    // the conv output AT the arg-max (exact: a stored value, not re-rounded): the backward's BatchNorm sums over pooled
    // pixels read it instead of gathering 2-byte values out of the full-resolution tensor
    if (raw) store8(raw + i * 8, braw);
  }
}

// global average pool [B][HW][C] -> [B][C] fp32; block = (b, 64-channel chunk)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, float* __restrict__ out, int HW, int C) {
  const int b = blockIdx.y, c0 = blockIdx.x * 64;
  const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;   // 8 channel groups x 32 pixel lanes
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.f;
  for (int p = pl; p < HW; p += 32) {
    float v[8];
    load8(x + ((size_t)b * HW + p) * C + c0 + cg * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] += v[k];
  }
  __shared__ float sh[32][64];
#pragma unroll
  for (int k = 0; k < 8; ++k) sh[pl][cg * 8 + k] = s[k];
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
    for (int q = 0; q < 32; ++q) t += sh[q][threadIdx.x];
    out[(size_t)b * C + c0 + threadIdx.x] = t / (float)HW;
  }
}

template <typename T>
__global__ void avgpool_bwd_kernel(const float* __restrict__ g, T* __restrict__ dx, size_t n8, int HW, int C8,
                                   int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int cg = (int)(i % C8);
  const size_t b = i / ((size_t)C8 * HW);
  const float inv = 1.f / (float)HW;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = g[b * C8 * 8 + cg * 8 + k] * inv;
  if (accumulate) {
    float o[8];
    load8(dx + i * 8, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += o[k];
  }
  store8(dx + i * 8, v);
}

// ---------------------------------------------------------------- BN backward
// dz source of a plain block BN: dz = g * (a > 0).  (The stem - g scattered through the max-pool argmax - has its own
// geometry-aware kernel below.)
template <typename T, typename TY = T>
struct DzPlain {
  const TY* g; const TY* a; const TY* y;      // (a: a block output as the elementwise passes read it - the fp32 copy in PH_PREC_FP16X3)
  // optional: the ReLU mask of a BatchNorm whose OWN output went through the ReLU (bn1 of a BasicBlock: a1 =
  // relu(y * scale + shift)) is a function of y, which the kernels read anyway - the activation tensor `a` is then not
  // read at all (2 of the 6 / 8 bytes per element of the reduce / apply pass)
  const float* mscale; const float* mshift;
  // the same in two phases for the prefetching apply pass: raw loads now, the mask when the element is consumed
  __device__ __forceinline__ void fetch(size_t i8, float (&dz)[8], float (&yy)[8], float (&m)[8]) const {
    load8(g + i8 * 8, dz);
    load8(y + i8 * 8, yy);
    if (a) load8(a + i8 * 8, m);
  }
  __device__ __forceinline__ void mask(const float (&ms8)[8], const float (&mh8)[8], float (&dz)[8], const float (&yy)[8],
                                       const float (&m)[8]) const {
    if (a) {
#pragma unroll
      for (int k = 0; k < 8; ++k) dz[k] = m[k] > 0.f ? dz[k] : 0.f;
    } else if (mscale) {
#pragma unroll
      for (int k = 0; k < 8; ++k) dz[k] = __builtin_fmaf(yy[k], ms8[k], mh8[k]) > 0.f ? dz[k] : 0.f;      // (as the forward's a1)
    }
  }
  __device__ __forceinline__ void get(size_t i8, int c, float (&dz)[8], float (&yy)[8]) const {
    load8(g + i8 * 8, dz);
    load8(y + i8 * 8, yy);
    if (a) {
      float m[8];
      load8(a + i8 * 8, m);
#pragma unroll
      for (int k = 0; k < 8; ++k) dz[k] = m[k] > 0.f ? dz[k] : 0.f;
    } else if (mscale) {
#pragma unroll
      for (int k = 0; k < 8; ++k) dz[k] = __builtin_fmaf(yy[k], mscale[c + k], mshift[c + k]) > 0.f ? dz[k] : 0.f;
    }
  }
};

constexpr int BWD_BLOCKS_MAX = 1024;

// amax (optional, PH_PREC_FP16X3): amax[block] = max |dz| over the block's elements - the BatchNorm-backward output dz' is
// stored as fp16 pairs and needs a per-tensor power-of-two scale, which bn_bwd_finalize_kernel derives from this bound
template <typename T, typename Src>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(Src src, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            float* __restrict__ parts, size_t npix, int C,
                                                            float* __restrict__ amax) {
  const int C8 = C >> 3;
  const int cg = threadIdx.x % C8, pl = threadIdx.x / C8, npl = 256 / C8;
  const int c = cg * 8;
  float mu[8], is[8], s1[8], s2[8], am = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  load8(mean + c, mu); load8(invstd + c, is);
  const size_t per = (npix + gridDim.x - 1) / gridDim.x;
  const size_t p0 = (size_t)blockIdx.x * per, p1 = min(npix, p0 + per);
  // the mask's per-channel constants in registers, the next pixel's operands requested before this one is accumulated (as the
  // apply passes: one dependent memory round trip per iteration otherwise)
  float ms8[8], mh8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { ms8[k] = 0.f; mh8[k] = 0.f; }
  if (!src.a && src.mscale) { load8(src.mscale + c, ms8); load8(src.mshift + c, mh8); }
  size_t p = p0 + pl;
  if (p < p1) {
    float dz[8], yy[8], mm[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) mm[k] = 0.f;
    src.fetch(p * C8 + cg, dz, yy, mm);
    for (;;) {
      const size_t pn = p + npl;
      const bool more = pn < p1;
      float dzn[8], yyn[8], mn[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { dzn[k] = 0.f; yyn[k] = 0.f; mn[k] = 0.f; }
      if (more) src.fetch(pn * C8 + cg, dzn, yyn, mn);
      src.mask(ms8, mh8, dz, yy, mm);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s1[k] += dz[k];
        s2[k] += dz[k] * (yy[k] - mu[k]) * is[k];
        am = fmaxf(am, fabsf(dz[k]));
      }
      if (!more) break;
#pragma unroll
      for (int k = 0; k < 8; ++k) { dz[k] = dzn[k]; yy[k] = yyn[k]; mm[k] = mn[k]; }
      p = pn;
    }
  }
  __shared__ float sh[2][256][9];
  if (amax) {      // (uniform)
    __shared__ float sha[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0) sha[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) amax[blockIdx.x] = fmaxf(fmaxf(sha[0], sha[1]), fmaxf(sha[2], sha[3]));
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { sh[0][threadIdx.x][k] = s1[k]; sh[1][threadIdx.x][k] = s2[k]; }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += 256) {
    const int which = o / C, ch = o % C;
    float t = 0.f;
    for (int q = 0; q < npl; ++q) t += sh[which][q * C8 + (ch >> 3)][ch & 7];
    parts[((size_t)blockIdx.x * 2 + which) * C + ch] = t;
  }
}

// dzs (optional, PH_PREC_FP16X3; with amax[namax], gamma, invstd): block 0 also writes the power-of-two scale of the dz' tensor
// the apply pass is about to store as fp16 pairs: dzs[0] = 2^s (multiplier of the apply pass), dzs[1] = 2^-s (the consumers'
// un-scale).  |dz'| = |gamma invstd (dz - c1 - xhat c2)| <= G A (2 + |xhat|) with A = max |dz|, G = max |gamma invstd| (|c1|,
// |c2| <= A: means of dz and of dz xhat with E|xhat| <= 1); s puts G A at 2^9, so fp16 (max 65504) holds |xhat| up to 125 and
// elements down to 2^-23 of the bound keep 11 + 11 significant bits.
// Rows written by a convolution's fused sums (PhTapConv::bst_y): nrow = 3 per part, the second sum taken from row `row2` (1: this
// BatchNorm, 2: the downsample branch reduced beside it) as sum dz (y - mean) and scaled here by s2_scale[c] = invstd[c] (in
// double) to sum dz xhat.  The separate reduce pass writes nrow = 2 rows with xhat already applied (s2_scale = null).
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ parts, int nparts, int C,
                                                              double count, float* dgamma, float* dbeta, float* c1,
                                                              float* c2, const float* __restrict__ amax, int namax,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ invstd, float* dzs, int nrow, int row2,
                                                              const float* __restrict__ s2_scale) {
  const int c = blockIdx.x, tid = threadIdx.x;
  if (dzs && c == 0) {
    float a = 0.f, g = 0.f;
    for (int i = tid; i < namax; i += 256) a = fmaxf(a, amax[i]);
    for (int i = tid; i < C; i += 256) g = fmaxf(g, fabsf(gamma[i] * invstd[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a = fmaxf(a, __shfl_xor(a, o, 64)); g = fmaxf(g, __shfl_xor(g, o, 64)); }
    __shared__ float sa[2][4];
    if ((tid & 63) == 0) { sa[0][tid >> 6] = a; sa[1][tid >> 6] = g; }
    __syncthreads();
    if (tid == 0) {
      a = fmaxf(fmaxf(sa[0][0], sa[0][1]), fmaxf(sa[0][2], sa[0][3]));
      g = fmaxf(fmaxf(sa[1][0], sa[1][1]), fmaxf(sa[1][2], sa[1][3]));
      const float bound = a * g;
      int e = 0;
      if (bound > 0.f && bound < INFINITY) e = 9 - ilogbf(bound);      // bound * 2^e in [2^9, 2^10)
      e = e > 100 ? 100 : (e < -100 ? -100 : e);
      dzs[0] = ldexpf(1.f, e);
      dzs[1] = ldexpf(1.f, -e);
    }
  }
  double s1 = 0.0, s2 = 0.0;
  for (int p = tid; p < nparts; p += 256) {
    s1 += (double)parts[((size_t)p * nrow + 0) * C + c];
    s2 += (double)parts[((size_t)p * nrow + row2) * C + c];
  }
  __shared__ double sh[2][4];
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if ((tid & 63) == 0) { sh[0][tid >> 6] = s1; sh[1][tid >> 6] = s2; }
  __syncthreads();
  if (tid == 0) {
    s1 = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    s2 = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    if (s2_scale) s2 *= (double)s2_scale[c];
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    c1[c] = (float)(s1 / count);
    c2[c] = (float)(s2 / count);
  }
}

template <typename T, typename Src>
__global__ void bn_bwd_apply_kernel(Src src, const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ c1,
                                    const float* __restrict__ c2, T* __restrict__ dy, size_t n8, int C8,
                                    const float* __restrict__ dzs) {
  // grid-stride loop with the per-channel constants in registers and the next element's operands prefetched (see bn_apply_kernel)
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 >= n8) return;
  const size_t gstride = (size_t)gridDim.x * blockDim.x;
  const int c = (int)(i0 % C8) * 8;
  float dz[8], yy[8], mm[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) mm[k] = 0.f;
  src.fetch(i0, dz, yy, mm);
  const float sc = dzs ? dzs[0] : 1.f;      // power of two (bn_bwd_finalize_kernel): the product is exact
  float is8[8], mu8[8], gi8[8], c18[8], c28[8], ms8[8], mh8[8];
  load8(invstd + c, is8); load8(mean + c, mu8); load8(gamma + c, gi8); load8(c1 + c, c18); load8(c2 + c, c28);
#pragma unroll
  for (int k = 0; k < 8; ++k) { gi8[k] = gi8[k] * is8[k]; ms8[k] = 0.f; mh8[k] = 0.f; }
  if (!src.a && src.mscale) { load8(src.mscale + c, ms8); load8(src.mshift + c, mh8); }
  for (size_t i = i0;;) {
    const size_t inext = i + gstride;
    const bool more = inext < n8;
    float dzn[8], yyn[8], mn[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { dzn[k] = 0.f; yyn[k] = 0.f; mn[k] = 0.f; }
    if (more) src.fetch(inext, dzn, yyn, mn);
    src.mask(ms8, mh8, dz, yy, mm);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (yy[k] - mu8[k]) * is8[k];
      dz[k] = gi8[k] * (dz[k] - c18[k] - xh * c28[k]) * sc;
    }
    store8(dy + i * 8, dz);
    if (!more) break;
#pragma unroll
    for (int k = 0; k < 8; ++k) { dz[k] = dzn[k]; yy[k] = yyn[k]; mm[k] = mn[k]; }
    i = inext;
  }
}

// ---- stem (C = 64): max-pool backward + ReLU mask + BN backward with the image geometry on the grid - a block walks
// STEM_ROWS image rows, thread = (8-channel group, pixel lane), so no per-element 64-bit div/mod and the per-channel
// constants live in registers (the generic kernels above spent most of their time there: 23-32 % of the HBM roofline)
constexpr int STEM_ROWS = 16;

template <typename T>
__device__ __forceinline__ void stem_dz(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, int OH, int OW,
                                        size_t b, int h, int w, int cg, float (&dz)[8]) {
  // the (up to) four pooling windows that contain input pixel (h, w).  All four argmax words and gradient vectors
  // are requested unconditionally (indices clamped, contribution masked) so that the eight loads are in flight
  // together: with a branch per window the kernel paid four dependent memory round trips per pixel (535 us).
  const int ph0 = h >> 1, ph1 = (h + 1) >> 1, pw0 = w >> 1, pw1 = (w + 1) >> 1;
  uint64_t packed[4];
  float g8[4][8];
  unsigned code[4];
  bool ok[4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ph = a ? ph1 : ph0, pw = c ? pw1 : pw0;
      ok[a * 2 + c] = !(a && ph1 == ph0) && !(c && pw1 == pw0) && ph < OH && pw < OW;
      const int phc = ph < OH ? ph : OH - 1, pwc = pw < OW ? pw : OW - 1;
      code[a * 2 + c] = (unsigned)((h - (2 * ph - 1)) * 3 + (w - (2 * pw - 1)));
      const size_t o8 = ((b * OH + phc) * OW + pwc) * 8 + cg;
      packed[a * 2 + c] = *reinterpret_cast<const uint64_t*>(idx + o8 * 8);
      load8(dpool + o8 * 8, g8[a * 2 + c]);
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) dz[k] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      dz[k] += (ok[q] && ((unsigned)(packed[q] >> (8 * k)) & 0xffu) == code[q]) ? g8[q][k] : 0.f;
}

template <typename T, typename TY, bool APPLY>
__global__ __launch_bounds__(256) void stem_bwd_kernel(const TY* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                       const TY* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ c1, const float* __restrict__ c2,
                                                       float* __restrict__ parts, T* __restrict__ dy, int B, int H, int W,
                                                       const float* __restrict__ dzs) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const int cg = threadIdx.x & 7, wl = threadIdx.x >> 3, c = cg * 8;
  float sc[8], sf[8], mu[8], is[8], ga[8], k1[8], k2[8], s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = scale[c + k]; sf[k] = shift[c + k]; mu[k] = mean[c + k]; is[k] = invstd[c + k];
    s1[k] = 0.f; s2[k] = 0.f;
    if (APPLY) { ga[k] = gamma[c + k] * is[k]; k1[k] = c1[c + k]; k2[k] = c2[c + k]; }
  }
  const float dsc = (APPLY && dzs) ? dzs[0] : 1.f;      // PH_PREC_FP16X3: power-of-two scale of the stored dz (exact)
  const int row0 = blockIdx.x * STEM_ROWS, row1 = min(B * H, row0 + STEM_ROWS);
  for (int row = row0; row < row1; ++row) {
    const size_t b = row / H;
    const int h = row - (int)b * H;
    for (int w = wl; w < W; w += 32) {
      const size_t i8 = ((size_t)row * W + w) * 8 + cg;
      float dz[8], yy[8];
      load8(y + i8 * 8, yy);
      stem_dz(dpool, idx, OH, OW, b, h, w, cg, dz);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        dz[k] = (yy[k] * sc[k] + sf[k]) > 0.f ? dz[k] : 0.f;
        const float xh = (yy[k] - mu[k]) * is[k];
        if (APPLY) dz[k] = ga[k] * (dz[k] - k1[k] - xh * k2[k]) * dsc;
        else { s1[k] += dz[k]; s2[k] += dz[k] * xh; }
      }
      if (APPLY) store8(dy + i8 * 8, dz);
    }
  }
  if (!APPLY) {
    __shared__ float sh[2][256][9];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sh[0][threadIdx.x][k] = s1[k]; sh[1][threadIdx.x][k] = s2[k]; }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
      float t = 0.f;
      for (int q = 0; q < 32; ++q) t += sh[which][q * 8 + (ch >> 3)][ch & 7];
      parts[((size_t)blockIdx.x * 2 + which) * 64 + ch] = t;
    }
  }
}

// BatchNorm-backward sums of the stem taken over POOLED pixels: every pooling window sends its gradient to exactly one
// input pixel (its argmax), so  sum_pixels dz = sum_windows [relu] dpool  and  sum_pixels dz * xhat = sum_windows [relu]
// dpool * xhat(argmax pixel)  - one 16-byte gradient load, one 8-byte code load and eight 2-byte gathers of the conv
// output per window-vector instead of four window gathers per input pixel.  Same partial-row layout as the kernel above
// (a block walks STEM_ROWS / 2 pooled rows).
template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_reduce_pooled_kernel(const T* __restrict__ dpool,
                                                                     const uint8_t* __restrict__ idx,
                                                                     const T* __restrict__ y,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     float* __restrict__ parts, int B, int H, int W,
                                                                     float* __restrict__ amax) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const int cg = threadIdx.x & 7, wl = threadIdx.x >> 3, c = cg * 8;
  float sc[8], sf[8], mu[8], is[8], s1[8], s2[8], am = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = scale[c + k]; sf[k] = shift[c + k]; mu[k] = mean[c + k]; is[k] = invstd[c + k];
    s1[k] = 0.f; s2[k] = 0.f;
  }
  constexpr int PROWS = STEM_ROWS / 2;
  const int row0 = blockIdx.x * PROWS, row1 = min(B * OH, row0 + PROWS);
  for (int prow = row0; prow < row1; ++prow) {
    const size_t b = prow / OH;
    const int ph = prow - (int)b * OH;
    for (int pw = wl; pw < OW; pw += 32) {
      const size_t o8 = ((size_t)prow * OW + pw) * 8 + cg;
      float g8[8];
      load8(dpool + o8 * 8, g8);
      const uint64_t packed = *reinterpret_cast<const uint64_t*>(idx + o8 * 8);
      const T* ybase = y + (((size_t)b * H + (2 * ph - 1)) * W + (2 * pw - 1)) * 64 + c;   // window origin (may lie outside:
#pragma unroll                                                                            //  the argmax never does)
      for (int k = 0; k < 8; ++k) {
        const unsigned code = (unsigned)(packed >> (8 * k)) & 0xffu;
        const unsigned kh = (code * 171u) >> 9, kw = code - 3u * kh;
        const float z = ldf(ybase + ((size_t)kh * W + kw) * 64 + k);
        const float dz = (z * sc[k] + sf[k]) > 0.f ? g8[k] : 0.f;
        s1[k] += dz;
        s2[k] += dz * ((z - mu[k]) * is[k]);
        am = fmaxf(am, fabsf(dz));
      }
    }
  }
  __shared__ float sh[2][256][9];
  __shared__ float sha[4];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sh[0][threadIdx.x][k] = s1[k]; sh[1][threadIdx.x][k] = s2[k]; }
  if (amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0) sha[threadIdx.x >> 6] = am;
  }
  __syncthreads();
  // (x 4: up to four overlapping 3 x 3 / stride-2 windows scatter their gradients onto ONE stem pixel, so max |dz| of the apply
  // pass can be four times the largest window gradient - the bound the dz scale is derived from must cover it)
  if (amax && threadIdx.x == 0) amax[blockIdx.x] = 4.f * fmaxf(fmaxf(sha[0], sha[1]), fmaxf(sha[2], sha[3]));
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
    float t = 0.f;
    for (int q = 0; q < 32; ++q) t += sh[which][q * 8 + (ch >> 3)][ch & 7];
    parts[((size_t)blockIdx.x * 2 + which) * 64 + ch] = t;
  }
}

// The same sums from the conv output AT the arg-max, which the forward's pooling pass saved per window (`raw`): no arg codes,
// no gather - two streamed tensors of the pooled size.  Same thread -> window mapping, same arithmetic and summation
// order as stem_bwd_reduce_pooled_kernel: bitwise the same partial rows (186 -> ~60 us at B = 64, 512 x 512).
template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_reduce_raw_kernel(const T* __restrict__ dpool, const T* __restrict__ raw,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  float* __restrict__ parts, int B, int H, int W,
                                                                  float* __restrict__ amax) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const int cg = threadIdx.x & 7, wl = threadIdx.x >> 3, c = cg * 8;
  float sc[8], sf[8], mu[8], is[8], s1[8], s2[8], am = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = scale[c + k]; sf[k] = shift[c + k]; mu[k] = mean[c + k]; is[k] = invstd[c + k];
    s1[k] = 0.f; s2[k] = 0.f;
  }
  constexpr int PROWS = STEM_ROWS / 2;
  const int row0 = blockIdx.x * PROWS, row1 = min(B * OH, row0 + PROWS);
  for (int prow = row0; prow < row1; ++prow) {
    for (int pw = wl; pw < OW; pw += 32) {
      const size_t o8 = ((size_t)prow * OW + pw) * 8 + cg;
      float g8[8], z8[8];
      load8(dpool + o8 * 8, g8);
      load8(raw + o8 * 8, z8);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float z = z8[k];
        const float dz = (z * sc[k] + sf[k]) > 0.f ? g8[k] : 0.f;
        s1[k] += dz;
        s2[k] += dz * ((z - mu[k]) * is[k]);
        am = fmaxf(am, fabsf(dz));
      }
    }
  }
  __shared__ float sh[2][256][9];
  __shared__ float sha[4];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sh[0][threadIdx.x][k] = s1[k]; sh[1][threadIdx.x][k] = s2[k]; }
  if (amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0) sha[threadIdx.x >> 6] = am;
  }
  __syncthreads();
  // (x 4: up to four overlapping 3 x 3 / stride-2 windows scatter their gradients onto ONE stem pixel, so max |dz| of the apply
  // pass can be four times the largest window gradient - the bound the dz scale is derived from must cover it)
  if (amax && threadIdx.x == 0) amax[blockIdx.x] = 4.f * fmaxf(fmaxf(sha[0], sha[1]), fmaxf(sha[2], sha[3]));
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
    float t = 0.f;
    for (int q = 0; q < 32; ++q) t += sh[which][q * 8 + (ch >> 3)][ch & 7];
    parts[((size_t)blockIdx.x * 2 + which) * 64 + ch] = t;
  }
}

inline unsigned nblk(size_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }
// blocks of the grid-stride elementwise passes: ~PH_EW_ITEMS eight-channel vectors per thread once the tensor is large enough to
// fill the chip anyway (>= 2048 blocks = one wave of 8 blocks per CU); A/B: PH_EW_ITEMS=1 restores one element per thread
inline unsigned ew_grid(size_t n8, int C8) {
  if (C8 <= 0 || 256 % C8) return (unsigned)((n8 + 255) / 256);      // (a thread keeps its channel group only if C / 8 divides the stride)
  static const int items = [] { const char* e = getenv("PH_EW_ITEMS"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : v; }();
  const size_t b = (n8 + 255) / 256;
  if (b <= 2048) return (unsigned)b;
  const size_t g = (b + items - 1) / items;
  return (unsigned)(g < 2048 ? 2048 : g);
}

}  // namespace

// precision mode -> (activation type T, conv-output type TY): PH_DISPATCH(prec, CALL) expands CALL(T, TY)
#define PH_DISPATCH(prec, CALL)                                  \
  do {                                                           \
    if ((prec) == PH_PREC_BF16) { CALL(bf16, bf16); }            \
    else if ((prec) == PH_PREC_FP16X3) { CALL(hp16, float); }    \
    else { CALL(float, float); }                                 \
  } while (0)

int ph_pack_input_launch(const float* x, void* x4, int B, int H, int W, int prec, hipStream_t st) {
  const size_t n = (size_t)B * H * W;
#define PH_CALL(T, TY) hipLaunchKernelGGL(pack_input_kernel<T>, dim3(nblk(n)), dim3(256), 0, st, x, (T*)x4, B, H, W)
  PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_bn_finalize_launch(const float* parts, int nparts, int C, double count, float eps, float momentum,
                           const float* gamma, const float* beta, float* mean, float* invstd, float* scale,
                           float* shift, float* running_mean, float* running_var, int64_t* nbt, hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, st, parts, nparts, C, count, eps, momentum, gamma, beta,
                     mean, invstd, scale, shift, running_mean, running_var, nbt);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_bn_eval_params_launch(const PhBnEvalTable* t, float eps, hipStream_t st) {
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3(t->n), dim3(256), 0, st, *t, eps);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_bn_apply_launch(const void* y, const float* scale, const float* shift, const void* res, const void* y_r,
                       const float* scale_r, const float* shift_r, void* out, void* out32, size_t npix, int C, int relu, int prec,
                       hipStream_t st) {
  return ph_bn_apply_launch2(y, scale, shift, res, y_r, scale_r, shift_r, out, out32, npix, C, relu, prec, 0, st);
}

int ph_bn_apply_launch2(const void* y, const float* scale, const float* shift, const void* res, const void* y_r,
                        const float* scale_r, const float* shift_r, void* out, void* out32, size_t npix, int C, int relu, int prec,
                        int res_as_t, hipStream_t st) {
  const size_t n8 = npix * (C / 8);
  void* tok = nullptr;
  if (ph_prof_on())
    ph_prof_begin(PH_CLS_BN_APPLY, (double)npix * C * (prec == PH_PREC_BF16 ? 2.0 : 4.0) * ((res || y_r) ? 3.0 : 2.0), st, &tok);
#define PH_CALL(T, TY)                                                                                              \
  hipLaunchKernelGGL((bn_apply_kernel<T, TY>), dim3(ew_grid(n8, C / 8)), dim3(256), 0, st, (const TY*)y, scale, shift, (const TY*)res, \
                     (const TY*)y_r, scale_r, shift_r, (T*)out, (TY*)out32, n8, C / 8, relu, res_as_t)
  PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_bn_relu_maxpool_launch(const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, void* raw,
                              void* out32, int B, int H, int W, int C, int prec, hipStream_t st) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const size_t n = (size_t)B * OH * OW * (C / 8);
#define PH_CALL(T, TY)                                                                                                          \
  do {                                                                                                                          \
    if (idx) hipLaunchKernelGGL((bn_relu_maxpool_kernel<T, TY, true>), dim3(nblk(n)), dim3(256), 0, st, (const TY*)y, scale, shift, \
                                (T*)out, idx, (TY*)raw, (TY*)out32, B, H, W, C / 8);                                            \
    else hipLaunchKernelGGL((bn_relu_maxpool_kernel<T, TY, false>), dim3(nblk(n)), dim3(256), 0, st, (const TY*)y, scale, shift,   \
                            (T*)out, idx, (TY*)raw, (TY*)out32, B, H, W, C / 8);                                                \
  } while (0)
  PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_avgpool_launch(const void* x, float* out, int B, int HW, int C, int prec, hipStream_t st) {
  dim3 grid(C / 64, B);
#define PH_CALL(T, TY) hipLaunchKernelGGL(avgpool_kernel<TY>, grid, dim3(256), 0, st, (const TY*)x, out, HW, C)
  PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
  PH_LAUNCH_CHECK();
  return PH_OK;
}

// the same over an activation in the form the convolutions read it (PH_PREC_FP16X3: the half-pair image; other modes: as above)
int ph_avgpool_launch_t(const void* x, float* out, int B, int HW, int C, int prec, hipStream_t st) {
  dim3 grid(C / 64, B);
#define PH_CALL(T, TY) hipLaunchKernelGGL(avgpool_kernel<T>, grid, dim3(256), 0, st, (const T*)x, out, HW, C)
  PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
  PH_LAUNCH_CHECK();
  return PH_OK;
}

// dx: a GRADIENT buffer (conv-output type)
int ph_avgpool_bwd_launch(const float* g, void* dx, int B, int HW, int C, int accumulate, int prec, hipStream_t st) {
  const size_t n8 = (size_t)B * HW * (C / 8);
#define PH_CALL(T, TY) hipLaunchKernelGGL(avgpool_bwd_kernel<TY>, dim3(nblk(n8)), dim3(256), 0, st, g, (TY*)dx, n8, HW, C / 8, accumulate)
  PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
  PH_LAUNCH_CHECK();
  return PH_OK;
}

// number of partial rows (= reduce blocks): about 8 eight-channel vectors per thread, so that the small late layers
// (16 k pixels x 512 channels) still fill the chip - with one block per 256 pixels layer 4 ran on 64 workgroups
int ph_bn_bwd_parts(size_t npix, int C) {
  const size_t b = (npix * (size_t)(C / 8) + 2047) / 2048;
  return (int)(b < (size_t)BWD_BLOCKS_MAX ? (b ? b : 1) : BWD_BLOCKS_MAX);
}

int ph_bn_bwd_reduce_launch(const void* g, const void* a, const void* y, const float* mean, const float* invstd,
                            float* parts, size_t npix, int C, int prec, const float* mscale, const float* mshift,
                            float* amax, hipStream_t st) {
  const int nb = ph_bn_bwd_parts(npix, C);
#define PH_CALL(T, TY)                                                                                                       \
  do {                                                                                                                       \
    DzPlain<T, TY> s{(const TY*)g, (const TY*)a, (const TY*)y, mscale, mshift};                                              \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, DzPlain<T, TY>>), dim3(nb), dim3(256), 0, st, s, mean, invstd, parts, npix, C, amax); \
  } while (0)
  PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_bn_bwd_finalize_launch(const float* parts, int nparts, int C, double count, float* dgamma, float* dbeta,
                              float* c1, float* c2, const float* amax, int namax, const float* gamma, const float* invstd,
                              float* dzs, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, parts, nparts, C, count, dgamma, dbeta, c1, c2,
                     amax, namax, gamma, invstd, dzs, 2, 1, (const float*)nullptr);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

// the rows a convolution's fused sums left (PhTapConv::bst_y): [nparts][3][C]; row2 = 1 (the BatchNorm of bst_y) or 2 (of bst_y2)
int ph_bn_bwd_finalize_fused_launch(const float* parts, int nparts, int C, double count, float* dgamma, float* dbeta,
                                    float* c1, float* c2, const float* invstd, int row2, hipStream_t st) {
  if (row2 != 1 && row2 != 2) return PH_EINVAL;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, parts, nparts, C, count, dgamma, dbeta, c1, c2,
                     (const float*)nullptr, 0, (const float*)nullptr, invstd, (float*)nullptr, 3, row2, invstd);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_bn_bwd_apply_launch(const void* g, const void* a, const void* y, const float* mean, const float* invstd,
                           const float* gamma, const float* c1, const float* c2, void* dy, size_t npix, int C,
                           int prec, const float* mscale, const float* mshift, const float* dzs, hipStream_t st) {
  const size_t n8 = npix * (C / 8);
#define PH_CALL(T, TY)                                                                                                       \
  do {                                                                                                                       \
    DzPlain<T, TY> s{(const TY*)g, (const TY*)a, (const TY*)y, mscale, mshift};                                              \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, DzPlain<T, TY>>), dim3(ew_grid(n8, C / 8)), dim3(256), 0, st, s, mean, invstd, gamma, c1, c2, \
                       (T*)dy, n8, C / 8, dzs);                                                                              \
  } while (0)
  PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_stem_bwd_parts(int B, int H) { return (B * H + STEM_ROWS - 1) / STEM_ROWS; }

int ph_stem_bwd_reduce_launch(const void* dpool, const uint8_t* idx, const void* y0, const void* raw, const float* mean,
                              const float* invstd, const float* scale, const float* shift, float* parts, int B, int H,
                              int W, int C, int prec, float* amax, hipStream_t st) {
  if (C != 64) return PH_EINVAL;
  if (amax && (H & 1)) return PH_EINVAL;      // (the per-pixel form below does not produce the bound)
  const int nb = ph_stem_bwd_parts(B, H);
  if (raw && H % 2 == 0) {   // the forward saved the conv output at every window's arg-max
#define PH_CALL(T, TY)                                                                                                      \
  hipLaunchKernelGGL((stem_bwd_reduce_raw_kernel<TY>), dim3(nb), dim3(256), 0, st, (const TY*)dpool, (const TY*)raw, scale, shift, \
                     mean, invstd, parts, B, H, W, amax)
    PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
    PH_LAUNCH_CHECK();
    return PH_OK;
  }
#ifndef PH_STEM_BWD_PER_PIXEL   // (A/B switch: the per-input-pixel form of the reduction, 246 us against 190 us)
  if (H % 2 == 0) {   // (the block -> pooled-row mapping reuses the partial-row count of the per-pixel kernel)
#define PH_CALL(T, TY)                                                                                                      \
  hipLaunchKernelGGL((stem_bwd_reduce_pooled_kernel<TY>), dim3(nb), dim3(256), 0, st, (const TY*)dpool, idx, (const TY*)y0, scale, \
                     shift, mean, invstd, parts, B, H, W, amax)
    PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
    PH_LAUNCH_CHECK();
    return PH_OK;
  }
#endif
#define PH_CALL(T, TY)                                                                                                         \
  hipLaunchKernelGGL((stem_bwd_kernel<T, TY, false>), dim3(nb), dim3(256), 0, st, (const TY*)dpool, idx, (const TY*)y0, scale, shift, \
                     mean, invstd, nullptr, nullptr, nullptr, parts, (T*)nullptr, B, H, W, nullptr)
  PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_stem_bwd_apply_launch(const void* dpool, const uint8_t* idx, const void* y0, const float* mean,
                             const float* invstd, const float* scale, const float* shift, const float* gamma,
                             const float* c1, const float* c2, void* dy0, int B, int H, int W, int C, int prec,
                             const float* dzs, hipStream_t st) {
  if (C != 64) return PH_EINVAL;
  const int nb = ph_stem_bwd_parts(B, H);
#define PH_CALL(T, TY)                                                                                                        \
  hipLaunchKernelGGL((stem_bwd_kernel<T, TY, true>), dim3(nb), dim3(256), 0, st, (const TY*)dpool, idx, (const TY*)y0, scale, shift, \
                     mean, invstd, gamma, c1, c2, nullptr, (T*)dy0, B, H, W, dzs)
  PH_DISPATCH(prec, PH_CALL);
#undef PH_CALL
  PH_LAUNCH_CHECK();
  return PH_OK;
}
