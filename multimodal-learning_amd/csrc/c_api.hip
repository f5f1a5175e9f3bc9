// Fine-grained convolution entry points of the C-ABI (used by the parity tests to pin each MFMA kernel
// against the oracle's F.conv2d) + ABI version.
#include "ph_common.h"
#include "ph_kernels.h"
#include "ph_dense.h"

namespace {
constexpr size_t AL = 256;
inline size_t up(size_t x) { return (x + AL - 1) / AL * AL; }

__global__ void parts_sum_kernel(const float* __restrict__ parts, int nparts, int C, float* s1, float* s2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int p = 0; p < nparts; ++p) { a += parts[((size_t)p * 2) * C + c]; b += parts[((size_t)p * 2 + 1) * C + c]; }
  if (s1) s1[c] = (float)a;
  if (s2) s2[c] = (float)b;
}

__global__ void parts3_sum_kernel(const float* __restrict__ parts, int nparts, int C, float* __restrict__ sums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a[3] = {0.0, 0.0, 0.0};
  for (int p = 0; p < nparts; ++p)
    for (int k = 0; k < 3; ++k) a[k] += parts[((size_t)p * 3 + k) * C + c];
  for (int k = 0; k < 3; ++k) sums[(size_t)k * C + c] = (float)a[k];
}

// fp32 <-> half-pair tensors (ph_common.h), 8 channels per thread
__global__ void hp_pack_kernel(const float* __restrict__ src, hp16* __restrict__ dst, size_t n8, float scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float v[8];
  load8(src + i * 8, v);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] *= scale;
  store8(dst + i * 8, v);
}
__global__ void hp_unpack_kernel(const hp16* __restrict__ src, float* __restrict__ dst, size_t n8) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float v[8];
  load8(src + i * 8, v);
  store8(dst + i * 8, v);
}

int chunks_for(int B, int OH, int OW, int S, int Cout, int Cin, int* tpc) {
  const int th = ph_wgrad_tile_h(S);
  const int ntiles = B * cdiv(OH, th) * cdiv(OW, 16);
  int want = cdiv(512, (Cout / 64) * (Cin / 64));
  if (want > ntiles) want = ntiles;
  if (want < 1) want = 1;
  *tpc = cdiv(ntiles, want);
  return cdiv(ntiles, *tpc);
}
}  // namespace

int ph_abi_version(void) { return 1; }

size_t ph_conv2d_workspace_bytes(int B, int Cin, int IH, int IW, int Cout, int KS, int stride, int pad) {
  const int OH = (IH + 2 * pad - KS) / stride + 1, OW = (IW + 2 * pad - KS) / stride + 1;
  const size_t wbytes = up((size_t)PH_NPLANES * KS * KS * Cin * Cout * sizeof(bf16));
  PhTapConv t{}; t.B = B; t.Cout = Cout; t.OHt = OH; t.OWt = OW;
  // (the largest row count of any arithmetic: the split modes' small tiles, or two rows per persistent workgroup of conv_tap6.hip)
  const int nparts = ph_tapconv_stat_parts(&t, stride, PH_PREC_BF16X6), nparts_hp = ph_tapconv_stat_parts(&t, stride, PH_PREC_FP16X3);
  const size_t parts = up((size_t)(nparts > nparts_hp ? nparts : nparts_hp) * 2 * Cout * sizeof(float));
  int tpc; const int nc = chunks_for(B, OH, OW, stride, Cout, Cin, &tpc);
  const size_t slab = up((size_t)nc * KS * KS * Cin * Cout * sizeof(float));
  return wbytes + (parts > slab ? parts : slab) + 256;
}

int ph_conv2d_fwd(const void* x, const float* w, void* y, float* ch_sum, float* ch_sumsq, int B, int Cin, int IH,
                  int IW, int Cout, int KS, int stride, int pad, int prec, void* ws_, hipStream_t st) {
  if ((KS != 1 && KS != 3) || Cin % 64 || Cout % 64) return PH_EINVAL;
  unsigned char* ws = reinterpret_cast<unsigned char*>(ws_);
  const size_t plane = (size_t)KS * KS * Cin * Cout;
  bf16* hi = reinterpret_cast<bf16*>(ws);
  int rc = prec == PH_PREC_FP16X3 ? ph_pack_w_hp_launch(w, hi, Cout, Cin, KS, 0, st) : ph_pack_w_fwd_launch(w, hi, Cout, Cin, KS, st);
  if (rc) return rc;
  // perf mode, what conv_tap7.hip takes: the fragment-major copy in plane 1 (a split plane this mode does not read)
  if (prec == PH_PREC_BF16 && KS == 3 && Cin == Cout && Cin >= 128 && (rc = ph_frag7_repack_launch(hi, Cout, Cin, 9, st))) return rc;
  const int OH = (IH + 2 * pad - KS) / stride + 1, OW = (IW + 2 * pad - KS) / stride + 1;
  PhTapConv t{};
  t.in = x; t.w = hi; t.wplane = plane; t.out = y;
  t.stats = reinterpret_cast<float*>(ws + up(PH_NPLANES * plane * sizeof(bf16)));
  t.B = B; t.IH = IH; t.IW = IW; t.Cin = Cin; t.Cout = Cout; t.OHt = OH; t.OWt = OW; t.OH = OH; t.OW = OW;
  t.os = 1; t.iy0 = -pad; t.ix0 = -pad; t.ntaps = KS * KS;
  for (int k = 0; k < t.ntaps; ++k) { t.dy[k] = k / KS; t.dx[k] = k % KS; t.wtap[k] = k; }
  const bool tap6b = prec == PH_PREC_BF16 && KS == 3 && stride == 2 && pad == 1 && ph_tap6b_switch(-1) && ph_tapconv6b_eligible(&t);
  if (tap6b && (rc = ph_frag7_repack_launch(hi, Cout, Cin, 9, st))) return rc;      // the fragment-major copy in plane 1
  if (!tap6b && KS == 3 && stride == 2 && pad == 1 && ph_tapconv2_setup_s2_fwd(&t, Cin, Cout, IH, IW, prec)) stride = 1;
  if (KS == 1 && stride == 2) {   // strided view (see resnet_plan.hip conv_fwd)
    t.in_pix_stride = 2L * Cin; t.in_row_stride = 2L * IW * Cin; t.in_img_stride = (long)IH * IW * Cin;
    t.IH = OH; t.IW = OW; stride = 1;
  }
  if ((rc = ph_tapconv_launch(&t, stride, prec, st))) return rc;
  if (ch_sum || ch_sumsq) {
    hipLaunchKernelGGL(parts_sum_kernel, dim3(cdiv(Cout, 64)), dim3(64), 0, st, t.stats,
                       ph_tapconv_stat_parts(&t, stride, prec), Cout, ch_sum, ch_sumsq);
    PH_LAUNCH_CHECK();
  }
  return PH_OK;
}

int ph_conv2d_dgrad_res(const void* dy, const float* w, void* dx, const void* res_g, const void* res_a, int B, int Cin,
                        int IH, int IW, int Cout, int KS, int stride, int pad, int prec, void* ws_, hipStream_t st) {
  if ((KS != 1 && KS != 3) || Cin % 64 || Cout % 64 || (stride != 1 && stride != 2)) return PH_EINVAL;
  const size_t plane = (size_t)KS * KS * Cin * Cout;
  bf16* hi = reinterpret_cast<bf16*>(ws_);
  int rc = (prec == PH_PREC_FP16X3 || prec == PH_PREC_FP16X1) ? ph_pack_w_hp_launch(w, hi, Cout, Cin, KS, 1, st) : ph_pack_w_dgrad_launch(w, hi, Cout, Cin, KS, st);
  if (rc) return rc;
  if (prec == PH_PREC_BF16 && KS == 3 && Cin == Cout && Cin >= 128 && (rc = ph_frag7_repack_launch(hi, Cin, Cout, 9, st))) return rc;
  const int OH = (IH + 2 * pad - KS) / stride + 1, OW = (IW + 2 * pad - KS) / stride + 1;
  PhTapConv t{};
  t.in = dy; t.w = hi; t.wplane = plane; t.out = dx; t.res_g = res_g; t.res_a = res_a;
  t.B = B; t.IH = OH; t.IW = OW; t.Cin = Cout; t.Cout = Cin; t.OH = IH; t.OW = IW;
  if (stride == 1) {
    t.OHt = IH; t.OWt = IW; t.os = 1; t.iy0 = -(KS - 1 - pad); t.ix0 = t.iy0; t.ntaps = KS * KS;
    for (int k = 0; k < t.ntaps; ++k) {
      t.dy[k] = k / KS; t.dx[k] = k % KS; t.wtap[k] = (KS - 1 - k / KS) * KS + (KS - 1 - k % KS);
    }
    return ph_tapconv_launch(&t, 1, prec, st);
  }
  // stride 2, as resnet_plan.hip:conv_dgrad: one launch per output parity class; a class no tap reaches keeps what
  // dx already holds (the in-place residual res_g == dx of the downsample path)
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      int nk = 0, khs[3], dhs[3], nw = 0, kws[3], dws[3];
      for (int kh = 0; kh < KS; ++kh)
        if (((a + pad - kh) & 1) == 0) { khs[nk] = kh; dhs[nk] = (a + pad - kh) / 2; ++nk; }
      for (int kw = 0; kw < KS; ++kw)
        if (((b + pad - kw) & 1) == 0) { kws[nw] = kw; dws[nw] = (b + pad - kw) / 2; ++nw; }
      t.OHt = (IH - a + 1) / 2; t.OWt = (IW - b + 1) / 2;
      if (t.OHt <= 0 || t.OWt <= 0) continue;
      t.os = 2; t.oa_h = a; t.oa_w = b; t.iy0 = 0; t.ix0 = 0; t.ntaps = nk * nw;
      if (t.ntaps == 0) continue;
      int q = 0;
      for (int i = 0; i < nk; ++i)
        for (int j = 0; j < nw; ++j) {
          if (dhs[i] < 0 || dws[j] < 0 || dhs[i] > 2 || dws[j] > 2) return PH_EINVAL;
          t.dy[q] = dhs[i]; t.dx[q] = dws[j]; t.wtap[q] = khs[i] * KS + kws[j]; ++q;
        }
      if ((rc = ph_tapconv_launch(&t, 1, prec, st))) return rc;
    }
  return PH_OK;
}

// test access to the in-LDS BatchNorm + ReLU of a 3x3 stride-1 perf-mode forward launch (PhTapConv::in_scale): y = conv(relu(x *
// in_scale[c] + in_shift[c]) rounded to bf16, w), bitwise what ph_bn_apply_launch + ph_conv2d_fwd give
int ph_conv2d_fwd_fused_in(const void* x, const float* in_scale, const float* in_shift, const float* w, void* y, float* ch_sum,
                           float* ch_sumsq, int B, int Cin, int IH, int IW, int Cout, void* ws_, hipStream_t st) {
  if (Cin % 64 || Cout % 64 || !in_scale || !in_shift) return PH_EINVAL;
  unsigned char* ws = reinterpret_cast<unsigned char*>(ws_);
  const size_t plane = (size_t)9 * Cin * Cout;
  bf16* hi = reinterpret_cast<bf16*>(ws);
  int rc = ph_pack_w_fwd_launch(w, hi, Cout, Cin, 3, st);
  if (rc) return rc;
  PhTapConv t{};
  t.in = x; t.w = hi; t.wplane = plane; t.out = y; t.in_scale = in_scale; t.in_shift = in_shift;
  t.stats = reinterpret_cast<float*>(ws + up(PH_NPLANES * plane * sizeof(bf16)));
  t.B = B; t.IH = IH; t.IW = IW; t.Cin = Cin; t.Cout = Cout; t.OHt = IH; t.OWt = IW; t.OH = IH; t.OW = IW;
  t.os = 1; t.iy0 = -1; t.ix0 = -1; t.ntaps = 9;
  for (int k = 0; k < 9; ++k) { t.dy[k] = k / 3; t.dx[k] = k % 3; t.wtap[k] = k; }
  if ((rc = ph_tapconv_launch(&t, 1, PH_PREC_BF16, st))) return rc;
  if (ch_sum || ch_sumsq) {
    hipLaunchKernelGGL(parts_sum_kernel, dim3(cdiv(Cout, 64)), dim3(64), 0, st, t.stats, ph_tapconv_stat_parts(&t, 1, PH_PREC_BF16), Cout,
                       ch_sum, ch_sumsq);
    PH_LAUNCH_CHECK();
  }
  return PH_OK;
}

// test access to the fused BatchNorm-backward sums of a 3x3 stride-1 perf-mode dgrad launch (PhTapConv::bst_y, conv_tap4.hip):
// dx = dgrad(dy, w) (+ res_g * (res_a > 0 | 1)) and sums[3][Cin] = sum dz | sum dz (bst_y - bst_mean) | sum dz (bst_y2 - bst_mean2)
// with dz = dx * (bst_a ? bst_a > 0 : bst_y * bst_scale + bst_shift > 0), combined in double from the per-workgroup rows
int ph_conv2d_dgrad_bnstat(const void* dy, const float* w, void* dx, const void* res_g, const void* res_a, const void* bst_y,
                           const void* bst_a, const void* bst_y2, const float* bst_scale, const float* bst_shift,
                           const float* bst_mean, const float* bst_mean2, float* sums, int B, int Cin, int IH, int IW, int Cout,
                           void* ws_, hipStream_t st) {
  if (Cin % 64 || Cout % 64 || !bst_y || !bst_mean || !sums) return PH_EINVAL;
  unsigned char* ws = reinterpret_cast<unsigned char*>(ws_);
  const size_t plane = (size_t)9 * Cin * Cout;
  bf16* hi = reinterpret_cast<bf16*>(ws);
  int rc = ph_pack_w_dgrad_launch(w, hi, Cout, Cin, 3, st);
  if (rc) return rc;
  if (Cin == Cout && Cin >= 128 && (rc = ph_frag7_repack_launch(hi, Cin, Cout, 9, st))) return rc;
  PhTapConv t{};
  t.in = dy; t.w = hi; t.wplane = plane; t.out = dx; t.res_g = res_g; t.res_a = res_a;
  t.B = B; t.IH = IH; t.IW = IW; t.Cin = Cout; t.Cout = Cin; t.OH = IH; t.OW = IW;
  t.OHt = IH; t.OWt = IW; t.os = 1; t.iy0 = -1; t.ix0 = -1; t.ntaps = 9;
  for (int k = 0; k < 9; ++k) { t.dy[k] = k / 3; t.dx[k] = k % 3; t.wtap[k] = 8 - k; }
  t.bst_y = bst_y; t.bst_a = bst_a; t.bst_y2 = bst_y2; t.bst_scale = bst_scale; t.bst_shift = bst_shift;
  t.bst_mean = bst_mean; t.bst_mean2 = bst_mean2;
  t.stats = reinterpret_cast<float*>(ws + up(PH_NPLANES * plane * sizeof(bf16)));
  if ((rc = ph_tapconv_launch(&t, 1, PH_PREC_BF16, st))) return rc;
  const int nparts = ph_tapconv2_stat_parts(&t);
  hipLaunchKernelGGL(parts3_sum_kernel, dim3(cdiv(Cin, 64)), dim3(64), 0, st, t.stats, nparts, Cin, sums);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_conv2d_dgrad(const void* dy, const float* w, void* dx, int B, int Cin, int IH, int IW, int Cout, int KS,
                    int stride, int pad, int prec, void* ws_, hipStream_t st) {
  if ((KS != 1 && KS != 3) || Cin % 64 || Cout % 64) return PH_EINVAL;
  const size_t plane = (size_t)KS * KS * Cin * Cout;
  bf16* hi = reinterpret_cast<bf16*>(ws_);
  int rc = (prec == PH_PREC_FP16X3 || prec == PH_PREC_FP16X1) ? ph_pack_w_hp_launch(w, hi, Cout, Cin, KS, 1, st) : ph_pack_w_dgrad_launch(w, hi, Cout, Cin, KS, st);
  if (rc) return rc;
  if (prec == PH_PREC_BF16 && KS == 3 && Cin == Cout && Cin >= 128 && (rc = ph_frag7_repack_launch(hi, Cin, Cout, 9, st))) return rc;
  const int OH = (IH + 2 * pad - KS) / stride + 1, OW = (IW + 2 * pad - KS) / stride + 1;
  PhTapConv t{};
  t.in = dy; t.w = hi; t.wplane = plane; t.out = dx;
  t.B = B; t.IH = OH; t.IW = OW; t.Cin = Cout; t.Cout = Cin; t.OH = IH; t.OW = IW;
  if (stride == 1) {
    t.OHt = IH; t.OWt = IW; t.os = 1; t.iy0 = -(KS - 1 - pad); t.ix0 = t.iy0; t.ntaps = KS * KS;
    for (int k = 0; k < t.ntaps; ++k) {
      t.dy[k] = k / KS; t.dx[k] = k % KS; t.wtap[k] = (KS - 1 - k / KS) * KS + (KS - 1 - k % KS);
    }
    return ph_tapconv_launch(&t, 1, prec, st);
  }
  const size_t es = prec == PH_PREC_BF16 ? 2 : 4;
  if (hipMemsetAsync(dx, 0, (size_t)B * IH * IW * Cin * es, st) != hipSuccess) return PH_ELAUNCH;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      int nk = 0, khs[3], dhs[3], nw = 0, kws[3], dws[3];
      for (int kh = 0; kh < KS; ++kh)
        if (((a + pad - kh) & 1) == 0 && a + pad - kh >= 0) { khs[nk] = kh; dhs[nk] = (a + pad - kh) / 2; ++nk; }
      for (int kw = 0; kw < KS; ++kw)
        if (((b + pad - kw) & 1) == 0 && b + pad - kw >= 0) { kws[nw] = kw; dws[nw] = (b + pad - kw) / 2; ++nw; }
      t.OHt = (IH - a + 1) / 2; t.OWt = (IW - b + 1) / 2;
      t.os = 2; t.oa_h = a; t.oa_w = b; t.iy0 = 0; t.ix0 = 0; t.ntaps = nk * nw;
      if (t.ntaps == 0 || t.OHt <= 0 || t.OWt <= 0) continue;
      int q = 0;
      for (int i = 0; i < nk; ++i)
        for (int j = 0; j < nw; ++j) { t.dy[q] = dhs[i]; t.dx[q] = dws[j]; t.wtap[q] = khs[i] * KS + kws[j]; ++q; }
      if ((rc = ph_tapconv_launch(&t, 1, prec, st))) return rc;
    }
  return PH_OK;
}

int ph_conv2d_wgrad(const void* x, const void* dy, float* dw, int B, int Cin, int IH, int IW, int Cout, int KS,
                    int stride, int pad, int prec, void* ws_, hipStream_t st) {
  if ((KS != 1 && KS != 3) || Cin % 64 || Cout % 64) return PH_EINVAL;
  unsigned char* ws = reinterpret_cast<unsigned char*>(ws_);
  const size_t plane = (size_t)KS * KS * Cin * Cout;
  const int OH = (IH + 2 * pad - KS) / stride + 1, OW = (IW + 2 * pad - KS) / stride + 1;
  PhWgrad g{};
  // layout: [256 B zero page][slab]
  if (hipMemsetAsync(ws, 0, 256, st) != hipSuccess) return PH_ELAUNCH;
  g.zeros = ws;
  g.x = x; g.dy = dy; g.slab = reinterpret_cast<float*>(ws + 256);
  g.B = B; g.IH = IH; g.IW = IW; g.Cin = Cin; g.OH = OH; g.OW = OW; g.Cout = Cout; g.S = stride; g.pad = pad; g.KS = KS;
  if (KS == 1 && stride == 2) {
    g.x_pix_stride = 2L * Cin; g.x_row_stride = 2L * IW * Cin; g.x_img_stride = (long)IH * IW * Cin;
    g.IH = OH; g.IW = OW; g.S = 1;
  }
  g.nchunks = chunks_for(B, OH, OW, g.S, Cout, Cin, &g.tiles_per_chunk);
  int rc = ph_wgrad_launch(&g, prec, st);
  if (rc) return rc;
  return ph_wgrad_reduce_launch(g.slab, dw, g.nchunks, KS, Cout, Cin, nullptr, st);
}

int ph_stem_dgrad(const void* dy_nhwc, const float* w_oihw, float* dx_nchw, int B, int H, int W, int prec, hipStream_t st) {
  if (!dy_nhwc || !w_oihw || !dx_nchw || B < 1 || H < 2 || W < 2 || (prec != PH_PREC_BF16 && !PH_IS_SPLIT_PREC(prec)))
    return PH_EINVAL;
  return ph_stem_dgrad_launch(dy_nhwc, w_oihw, dx_nchw, B, H, W, prec, st);
}

// PH_PREC_FP16X3 storage of a tensor whose innermost extent is a multiple of 64 (NHWC activations, C % 64 == 0): fp32 -> the
// half-pair layout [..][C / 64][2][64] fp16 (x * scale ~= hi + lo * 2^-11; `scale` a power of two, 1 for activations) and
// back.  n = number of elements (multiple of 64), both pointers 256-B aligned, same byte size either way.
int ph_hp_pack(const float* src, void* dst, size_t n, float scale, hipStream_t st) {
  if (!src || !dst || (n & 63) || ((uintptr_t)dst & 255)) return PH_EINVAL;
  hipLaunchKernelGGL(hp_pack_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, src, reinterpret_cast<hp16*>(dst), n / 8, scale);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_hp_unpack(const void* src, float* dst, size_t n, hipStream_t st) {
  if (!src || !dst || (n & 63) || ((uintptr_t)src & 255)) return PH_EINVAL;
  hipLaunchKernelGGL(hp_unpack_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const hp16*>(src), dst, n / 8);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
