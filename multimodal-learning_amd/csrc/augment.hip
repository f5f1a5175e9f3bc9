// On-device input pipeline of the training loader (reference MICCAI-2022/data_loaders_MT.py:168-175,
// 51-53: TransformTwice(Compose([RandomHorizontalFlip, RandomVerticalFlip, RandomCrop(input_size_path),
// ColorJitter(0.1, 0.1, 0.05, 0.01), ToTensor, Normalize(0.5, 0.5)])) run by PIL in four DataLoader workers).
// Source images stay uint8 [B][SH][SW][3] in HBM; one launch draws the per-(image, view) parameters, one computes the
// grey mean the contrast step needs, one writes both fp32 NCHW views.  Byte work, HBM-bound: 0.8 MB read + 3 MB written
// per 512 x 512 view.
//
// The colour arithmetic is Pillow's, as torchvision's PIL backend drives it, and is PINNED against Pillow itself
// (tests/golden/colorjitter_pil.npz, produced by the real ImageEnhance / convert("HSV") calls; oracle/augment.py states
// the same arithmetic in numpy): ImageEnhance = Image.blend(degenerate, image, factor) on uint8 with truncation;
// brightness blends with black, contrast with the rounded mean of the ITU-R 601-2 luma, saturation with the per-pixel
// luma; the four steps run in a drawn order; hue is a wrapping add on the uint8 H channel of Pillow's HSV image.
#include "ph_common.h"
#include "ph_kernels.h"

namespace {

constexpr int AP = 16;   // floats per (image, view) parameter row:
// 0 flipH 1 flipV 2 top 3 left 4 brightness 5 contrast 6 saturation 7 hue 8..11 order (0 b, 1 c, 2 s, 3 h) 12 grey mean
// 13 bit mask of DISABLED steps (bit k = step k is skipped: torchvision's ColorJitter drops a step whose range is zero -
//    note that an enabled hue step with factor 0 still runs Pillow's lossy uint8 HSV round trip)
// 14..15: the 64-bit integer grey sum (must be zero on entry to ph_augment_apply)

__device__ __forceinline__ uint64_t amix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(uint64_t h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }

__global__ void augment_params_kernel(float* __restrict__ params, int n, uint64_t seed, const uint64_t* __restrict__ step,
                                      int SH, int SW, int S, float jb, float jc, float js, float jh) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (image, view)
  if (i >= n) return;
  const uint64_t st = step ? step[0] : 0;
  const uint64_t base = amix(seed ^ amix(st * 0x9E3779B97F4A7C15ull + (uint64_t)i + 1));
  float* p = params + (size_t)i * AP;
  p[0] = (amix(base ^ 1) >> 63) ? 1.f : 0.f;
  p[1] = (amix(base ^ 2) >> 63) ? 1.f : 0.f;
  p[2] = (float)(amix(base ^ 3) % (uint64_t)(SH - S + 1));
  p[3] = (float)(amix(base ^ 4) % (uint64_t)(SW - S + 1));
  p[4] = 1.f - jb + 2.f * jb * u01(amix(base ^ 5));
  p[5] = 1.f - jc + 2.f * jc * u01(amix(base ^ 6));
  p[6] = 1.f - js + 2.f * js * u01(amix(base ^ 7));
  p[7] = -jh + 2.f * jh * u01(amix(base ^ 8));
  int idx = (int)(amix(base ^ 9) % 24), pool[4] = {0, 1, 2, 3};
  for (int k = 0, f = 6; k < 4; ++k) {   // factorial-base decode of one of the 24 orders
    const int q = idx / f; idx %= f;
    p[8 + k] = (float)pool[q];
    for (int t = q; t < 3 - k; ++t) pool[t] = pool[t + 1];
    if (k < 3) f /= (3 - k);
  }
  p[12] = 0.f; p[14] = p[15] = 0.f;
  p[13] = (float)((jb == 0.f ? 1 : 0) | (jc == 0.f ? 2 : 0) | (js == 0.f ? 4 : 0) | (jh == 0.f ? 8 : 0));
}

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// PIL ImagingBlend on one uint8 channel: degenerate d, image v, factor f
__device__ __forceinline__ int blend8(int d, int v, float f) {
  const float t = __fadd_rn((float)d, __fmul_rn(f, (float)(v - d)));
  if (f >= 0.f && f <= 1.f) return (int)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

// torchvision functional_pil.adjust_hue: img.convert("HSV"), wrapping uint8 add of uint8(hue_factor * 255) on H,
// convert("RGB") - Pillow's Convert.c (rgb2hsv_row / hsv2rgb) restated with its float / double mix; no contraction to
// FMA (the C library evaluates every product and sum separately).  Pinned bit for bit against Pillow itself
// (tests/golden/make_golden_colorjitter.py; oracle/augment.py is checked against Pillow over all 2^24 colours).
__device__ __forceinline__ int clip8(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }
__device__ __forceinline__ void hue_shift(int& r, int& g, int& b, float hf) {
#pragma clang fp contract(off)
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
    float h;
    if (r == maxc) h = __fsub_rn(bc, gc);
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = clip8((int)((double)h * 255.0));
    us = clip8((int)((double)s * 255.0));
  }
  uh = (uh + ((int)((double)hf * 255.0) & 255)) & 255;      // np.uint8(hue_factor * 255), wrapping add
  if (us == 0) { r = g = b = uv; return; }
  const double h6 = (double)(float)uh * 6.0 / 255.0;
  const double fi = floor(h6), f = h6 - fi, fs = (double)(float)us / 255.0, v = (double)(float)uv;
  const double om = 1.0 - f, a1 = fs * f, a2 = fs * om;
  const int p = clip8((int)floor(v * (1.0 - fs) + 0.5)), q = clip8((int)floor(v * (1.0 - a1) + 0.5)),
            t = clip8((int)floor(v * (1.0 - a2) + 0.5));      // C round() of non-negative values
  switch ((int)fi % 6) {
    case 0: r = uv; g = t; b = p; break;
    case 1: r = q; g = uv; b = p; break;
    case 2: r = p; g = uv; b = t; break;
    case 3: r = p; g = q; b = uv; break;
    case 4: r = t; g = p; b = uv; break;
    default: r = uv; g = p; b = q; break;
  }
}

// the colour steps in their drawn order; stops in front of the contrast step when `until_contrast`
__device__ __forceinline__ void jitter(int& r, int& g, int& b, const float* p, int mean, bool until_contrast) {
  const int off = (int)p[13];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int op = (int)p[8 + k];
    if ((off >> op) & 1) continue;
    if (op == 0) { r = blend8(0, r, p[4]); g = blend8(0, g, p[4]); b = blend8(0, b, p[4]); }
    else if (op == 1) {
      if (until_contrast) return;
      const int m = mean;
      r = blend8(m, r, p[5]); g = blend8(m, g, p[5]); b = blend8(m, b, p[5]);
    } else if (op == 2) { const int l = luma(r, g, b); r = blend8(l, r, p[6]); g = blend8(l, g, p[6]); b = blend8(l, b, p[6]); }
    else hue_shift(r, g, b, p[7]);
  }
}

__device__ __forceinline__ void fetch(const uint8_t* __restrict__ src, const float* p, size_t b, int SH, int SW, int y, int x,
                                      int& r, int& g, int& bl) {
  int sy = (int)p[2] + y, sx = (int)p[3] + x;              // crop window of the flipped image ...
  if (p[1] != 0.f) sy = SH - 1 - sy;                       // ... = mirrored coordinates of the source
  if (p[0] != 0.f) sx = SW - 1 - sx;
  sy = min(max(sy, 0), SH - 1); sx = min(max(sx, 0), SW - 1);   // caller-supplied windows cannot leave the tile
  const uint8_t* q = src + ((b * SH + sy) * SW + sx) * 3;
  r = q[0]; g = q[1]; bl = q[2];
}

// grey sum (PIL: int(ImageStat.Stat(image.convert("L")).mean[0] + 0.5)) of the image as it enters the contrast step:
// MSPLIT workgroups per (image, view) add their exact integer partial sums to the 64-bit slot of the parameter row
constexpr int MSPLIT = 16;
__global__ __launch_bounds__(256) void augment_mean_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ rows,
                                                           float* __restrict__ params, int SH, int SW, int S) {
  __shared__ unsigned long long red[256];
  const int iv = blockIdx.x, b = iv >> 1;
  float* p = params + (size_t)iv * AP;
  unsigned long long s = 0;
  const int n = S * S, per = (n + MSPLIT - 1) / MSPLIT, lo = blockIdx.y * per, hi = min(n, lo + per);
  for (int e = lo + threadIdx.x; e < hi; e += 256) {
    int r, g, bl;
    fetch(src, p, rows ? (size_t)rows[b] : (size_t)b, SH, SW, e / S, e % S, r, g, bl);
    jitter(r, g, bl, p, 0, true);
    s += (unsigned long long)luma(r, g, bl);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(p + 14), red[0]);
}

__device__ __forceinline__ int grey_mean(const float* p, int S) {
  const unsigned long long sum = *reinterpret_cast<const unsigned long long*>(p + 14);
  return (int)((double)sum / ((double)S * (double)S) + 0.5);
}

// both views: out_v[b][c][y][x] = (jittered / 255 - 0.5) / 0.5
__global__ void augment_apply_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ rows,
                                     const float* __restrict__ params, float* __restrict__ out0, float* __restrict__ out1, int B,
                                     int SH, int SW, int S) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)S * S;
  if (i >= (size_t)B * 2 * per) return;
  const int iv = (int)(i / per), b = iv >> 1, v = iv & 1;
  const int e = (int)(i % per);
  const float* p = params + (size_t)iv * AP;
  int r, g, bl;
  fetch(src, p, rows ? (size_t)rows[b] : (size_t)b, SH, SW, e / S, e % S, r, g, bl);
  jitter(r, g, bl, p, grey_mean(p, S), false);
  float* o = (v ? out1 : out0) + (size_t)b * 3 * per + e;
  o[0] = ((float)r / 255.f - 0.5f) / 0.5f;
  o[per] = ((float)g / 255.f - 0.5f) / 0.5f;
  o[2 * per] = ((float)bl / 255.f - 0.5f) / 0.5f;
}

__global__ void augment_publish_mean_kernel(float* __restrict__ params, int n, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) params[(size_t)i * AP + 12] = (float)grey_mean(params + (size_t)i * AP, S);
}

}  // namespace

#include "pathomic_hip.h"

extern "C" {

int ph_augment_params(float* params, int B, uint64_t seed, const uint64_t* step, int SH, int SW, int S, float brightness,
                      float contrast, float saturation, float hue, hipStream_t st) {
  if (!params || B < 1 || S < 1 || S > SH || S > SW) return PH_EINVAL;
  hipLaunchKernelGGL(augment_params_kernel, dim3((2 * B + 255) / 256), dim3(256), 0, st, params, 2 * B, seed, step, SH, SW, S,
                     brightness, contrast, saturation, hue);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_augment_apply(const uint8_t* src, const int64_t* rows, float* params, float* out0, float* out1, int B, int SH, int SW,
                     int S, hipStream_t st) {
  if (!src || !params || !out0 || !out1 || B < 1 || S < 1 || S > SH || S > SW) return PH_EINVAL;
  hipLaunchKernelGGL(augment_mean_kernel, dim3(2 * B, MSPLIT), dim3(256), 0, st, src, rows, params, SH, SW, S);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(augment_publish_mean_kernel, dim3((2 * B + 255) / 256), dim3(256), 0, st, params, 2 * B, S);
  PH_LAUNCH_CHECK();
  const size_t n = (size_t)B * 2 * S * S;
  hipLaunchKernelGGL(augment_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, rows, params, out0, out1, B, SH, SW, S);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

}  // extern "C"
