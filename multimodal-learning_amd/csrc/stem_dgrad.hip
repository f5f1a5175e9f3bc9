// Input gradient of the stem convolution (7x7, stride 2, pad 3, 3 -> 64 channels): the one dgrad the distillation hot
// path never needs (the image takes no gradient there), used by the MIA-2023 stage-1 superpixel attention masks, which
// rank superpixels by d loss / d image ("MIA 2023/stage1_multi_modal_teacher/train_test_MT_SP_Masking.py":45-75).
//   dx[b][c][iy][ix] = sum_o sum_{ky,kx} dy[b][(iy+3-ky)/2][(ix+3-kx)/2][o] * w[o][c][ky][kx]   ((iy+3-ky), (ix+3-kx) even)
// 3 output channels: no MFMA shape fits, so this is an LDS-tiled fp32 VALU kernel.  Workgroup = 16 x 16 input pixels,
// 4 waves = the 4 (row, column) parity classes - inside a wave every lane runs the same (ky, kx) taps, so the weight
// reads are LDS broadcasts and there is no divergence; dy rows are padded to 68 floats against bank conflicts.
#include "ph_common.h"
#include "ph_kernels.h"

namespace {

constexpr int SD_T = 16;                 // input-pixel tile edge
constexpr int SD_OT = SD_T / 2 + 4;      // dy rows / columns a tile can touch: (iy0-3)/2 .. (iy0+18)/2
constexpr int SD_LD = 68;                // floats per staged dy pixel (64 + pad)

template <typename T>
__global__ __launch_bounds__(256) void stem_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                         float* __restrict__ dx, int H, int W, int OH, int OW) {
  extern __shared__ float sm[];
  float* wl = sm;                                  // [ky][kx][c][o]  49 * 3 * 64
  float* dl = sm + 49 * 3 * 64;                    // [SD_OT][SD_OT][SD_LD]
  const int tid = threadIdx.x, b = blockIdx.z;
  const int iy0 = blockIdx.y * SD_T, ix0 = blockIdx.x * SD_T;
  const int oyb = iy0 / 2 - 2, oxb = ix0 / 2 - 2;  // first staged dy row / column (iy0, ix0 are multiples of 16)
  for (int e = tid; e < 49 * 3 * 64; e += 256) {
    const int o = e & 63, c = (e >> 6) % 3, k = e / 192;     // k = ky * 7 + kx
    wl[e] = w[((size_t)o * 3 + c) * 49 + k];
  }
  for (int e = tid; e < SD_OT * SD_OT * 8; e += 256) {
    const int ch8 = e & 7, p = e >> 3, r = p / SD_OT, q = p % SD_OT;
    const int oy = oyb + r, ox = oxb + q;
    float v[8];
    if (oy >= 0 && oy < OH && ox >= 0 && ox < OW) {
      load8(dy + (((size_t)b * OH + oy) * OW + ox) * 64 + ch8 * 8, v);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) dl[p * SD_LD + ch8 * 8 + k] = v[k];
  }
  __syncthreads();
  const int cls = tid >> 6, pa = cls >> 1, pb = cls & 1, l = tid & 63;
  const int iy = iy0 + 2 * (l >> 3) + pa, ix = ix0 + 2 * (l & 7) + pb;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  // ky = (pa + 1) & 1, +2, ...:  (iy + 3 - ky) even
  for (int ky = (pa + 1) & 1; ky < 7; ky += 2) {
    const int r = (iy + 3 - ky) / 2 - oyb;
    for (int kx = (pb + 1) & 1; kx < 7; kx += 2) {
      const int q = (ix + 3 - kx) / 2 - oxb;
      const float4* d4 = reinterpret_cast<const float4*>(dl + (r * SD_OT + q) * SD_LD);
      const float4* w4 = reinterpret_cast<const float4*>(wl + (ky * 7 + kx) * 192);
#pragma unroll 4
      for (int o = 0; o < 16; ++o) {
        const float4 d = d4[o], u0 = w4[o], u1 = w4[16 + o], u2 = w4[32 + o];
        a0 += d.x * u0.x + d.y * u0.y + d.z * u0.z + d.w * u0.w;
        a1 += d.x * u1.x + d.y * u1.y + d.z * u1.z + d.w * u1.w;
        a2 += d.x * u2.x + d.y * u2.y + d.z * u2.z + d.w * u2.w;
      }
    }
  }
  if (iy < H && ix < W) {
    const size_t hw = (size_t)H * W, o = (size_t)b * 3 * hw + (size_t)iy * W + ix;
    dx[o] = a0; dx[o + hw] = a1; dx[o + 2 * hw] = a2;
  }
}

}  // namespace

int ph_stem_dgrad_launch(const void* dy, const float* w_oihw, float* dx_nchw, int B, int H, int W, int prec, hipStream_t st) {
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  const size_t lds = (size_t)(49 * 3 * 64 + SD_OT * SD_OT * SD_LD) * sizeof(float);
  dim3 grid((W + SD_T - 1) / SD_T, (H + SD_T - 1) / SD_T, B);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_dgrad_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(stem_dgrad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return PH_ELAUNCH;
    attr_done = true;
  }
  if (prec == PH_PREC_BF16)
    hipLaunchKernelGGL(stem_dgrad_kernel<bf16>, grid, dim3(256), lds, st, (const bf16*)dy, w_oihw, dx_nchw, H, W, OH, OW);
  else
    hipLaunchKernelGGL(stem_dgrad_kernel<float>, grid, dim3(256), lds, st, (const float*)dy, w_oihw, dx_nchw, H, W, OH, OW);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
