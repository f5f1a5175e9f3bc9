// Tap-convolution implicit GEMM on MFMA (gfx950): ONE kernel family serves
//   * forward 3x3 / 1x1 convolutions, stride 1 or 2   (reference resnets.py:26-34,50-74)
//   * data-gradient (dgrad) of the same convolutions: stride-1 dgrad = conv with transposed weights;
//     stride-2 dgrad = 4 output-parity classes, each a stride-1 tap-conv with 1/2/4 taps.
//
// GEMM view: C[pixel][cout] = sum_{tap,cin} A[pixel+tap][cin] * W[tap][cout][cin]
//   A: NHWC activations.  A spatial halo tile ((TH-1)*S+3) x ((TW-1)*S+3) x 64 channels is staged in LDS
//      ONCE per 64-channel slice and re-read by all taps ("LDS-staged im2col": nothing im2col-shaped
//      ever exists in HBM).  16-B chunks are XOR-swizzled by pixel index against bank conflicts.
//   W: packed [tap][Cout][Cin] bf16 (cin contiguous = the MFMA K direction), staged per tap group.
//   MFMA: v_mfma_f32_32x32x16_bf16; rows = pixels, cols = cout, so per-channel BatchNorm partial sums
//      are plain in-register sums over the accumulator registers (cout lives on the lane).
// Precision: T = bf16 -> perf mode (1 MFMA / k-step).  T = float -> parity mode "bf16x6": fp32 activations are
//   split into 3 bf16 planes while staging, weights come as 3 planes, 6 MFMAs / k-step reproduce the fp32
//   products to ~2^-24 (fp32 accumulate as in the reference).
#include <algorithm>
#include "ph_common.h"
#include "ph_kernels.h"
#include "tap_common.h"

namespace {
#ifdef PH_TAP_TRACE
__device__ unsigned long long ph_tap_trace[PH_TRACE_WGS * 12];
#endif

template <typename T, int S, int TH, int BNT, int WM, int WN, int FM, int FN, int TG>
struct TapCfg {
  static constexpr bool SPLIT = is_f32<T>::value;
  static constexpr bool HPM = is_hp<T>::value;   // PH_PREC_FP16X3: half-pair input, fp16 MFMA over 3 (A block, W block) pairs per slice, fp32 output
  static constexpr int TW = 16;
  static constexpr int HPH = (TH - 1) * S + 3;
  static constexpr int HPW = (TW - 1) * S + 3;
  static constexpr int HP = HPH * HPW;
  static constexpr int A_BYTES = (HP + 1) / 2 * 256;   // lds_off() addresses rows in 256-B pairs
  static constexpr int B_BYTES = TG * BNT * 128;
  static constexpr int NP = SPLIT ? PH_NPLANES : 1;
  // perf-mode 8-wave tiles double-buffer the weight stages in LDS when A + 2 B fits the 160 KB: the refill of
  // stage s+1 is written while stage s computes and one barrier per stage remains.  (An LDS-DMA ring for the same
  // purpose measured slower, r01: 597 vs 672 TFLOP/s - 6 x 1-KiB DMA pieces per wave and stage cost more issue time
  // than the VGPR round trip.)
  static constexpr bool DB = !SPLIT && (WM * WN == 8) && (A_BYTES + 2 * B_BYTES <= 160 * 1024);
  static constexpr int LDS_BYTES = DB ? (A_BYTES + 2 * B_BYTES) : (A_BYTES + B_BYTES) * NP;
  static constexpr int NTH = WM * WN * 64;
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  static_assert(WM * FM * 32 == TH * TW, "M tiling");
  static_assert(WN * FN * 32 == BNT, "N tiling");
};

template <typename T, int S, int TH, int BNT, int WM, int WN, int FM, int FN, int TG>
__global__ __launch_bounds__(WM * WN * 64) void tapconv_kernel(PhTapConv p) {
  using C = TapCfg<T, S, TH, BNT, WM, WN, FM, FN, TG>;
  constexpr bool SPLIT = C::SPLIT, HPM = C::HPM;
  constexpr int TW = C::TW, HPW = C::HPW, HP = C::HP, NP = C::NP, NTH = C::NTH;
  typedef typename std::conditional<HPM, f16, T>::type TI;       // element type of the input as the staging code addresses it
  typedef typename std::conditional<HPM, float, T>::type TO;     // output / residual-gradient type
  constexpr int EW = HPM ? 2 : 1;                                // fp16 elements per input element
  constexpr int HCH = (HP * 8 + NTH - 1) / NTH;          // halo 16-B chunks per thread
  constexpr int WPT = BNT * 8 / NTH;                     // weight 16-B chunks per thread and tap
  constexpr int WCH = TG * WPT;                          // ... per stage (full tap group)
  static_assert((BNT * 8) % NTH == 0, "a tap's weight block must split evenly over the workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ldsA = smem;                      // NP planes of A_BYTES
  unsigned char* ldsB = smem + C::A_BYTES * NP;    // NP planes of B_BYTES

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // output-parity classes of a stride-2 dgrad in one launch (PhTapConv::ncls): class = blockIdx.z / B, every per-class value a
  // wave-uniform scalar read from the descriptor
  int cls = 0, b = blockIdx.z;
  int L_ntaps = p.ntaps, L_oa_h = p.oa_h, L_oa_w = p.oa_w, L_OHt = p.OHt, L_OWt = p.OWt;
  if (p.ncls) {
    cls = (b >= p.B) + (b >= 2 * p.B) + (b >= 3 * p.B);
    b -= cls * p.B;
    L_ntaps = p.c_ntaps[cls]; L_oa_h = p.c_oa_h[cls]; L_oa_w = p.c_oa_w[cls]; L_OHt = p.c_OHt[cls]; L_OWt = p.c_OWt[cls];
  }
  const int tiles_w = (L_OWt + TW - 1) / TW;
  const int tile = blockIdx.x;
  if (p.ncls && tile >= tiles_w * ((L_OHt + TH - 1) / TH)) return;      // (a smaller class in a grid sized for the largest)
  const int r0 = (tile / tiles_w) * TH, c0 = (tile % tiles_w) * TW;
  const int n0 = blockIdx.y * BNT;
  const long pix_st = (p.in_pix_stride ? p.in_pix_stride : p.Cin) * EW;
  const long row_st = (p.in_row_stride ? p.in_row_stride : (long)p.IW * p.Cin) * EW;
  const long img_st = (p.in_img_stride ? p.in_img_stride : (long)p.IH * p.IW * p.Cin) * EW;
  const TI* in = reinterpret_cast<const TI*>(p.in) + (size_t)b * img_st;
  const bf16* wbase = reinterpret_cast<const bf16*>(p.w);       // (2-byte elements: bf16, or fp16 in the half-pair mode)
  const int wK = HPM ? 3 * p.Cin : p.Cin;                       // K extent of a packed weight row
  // slice index -> element offset of its A block in a pixel's channel record / of its W block in a weight row.  Half-pair
  // mode: slices 3c, 3c+1, 3c+2 of 64-channel group c pair the A blocks (hi, hi, lo) with the W blocks (hi 2^11, lo, hi).
  // (hi-only form, PH_PREC_FP16X1: one slice per group, A block hi x W block hi = the third block of the group's three)
  const bool hi1 = HPM && p.hp_hi_only;
  auto slice_a = [&](int sl) { return HPM ? (hi1 ? sl * 128 : ((sl / 3) * 128 + ((sl % 3 == 2) ? 64 : 0))) : (sl << 6); };
  auto slice_w = [&](int sl) { return hi1 ? (3 * sl + 2) << 6 : sl << 6; };
  PH_TRACE(0);
  PH_TRACE_HWID();
  const int iy_base = r0 * S + p.iy0, ix_base = c0 * S + p.ix0;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

  // per-lane constant pieces of the fragment addresses
  int prow[FM];   // (r*S)*HPW + c*S of this lane's pixel, per M fragment (fragment f covers tile rows 2f, 2f+1)
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int fr, c;
    frag_row_to_pixel(lane & 31, fr, c);
    prow[i] = (((wm * FM + i) * 2 + fr) * S) * HPW + c * S;
  }
  int nrow[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) nrow[j] = (wn * FN + j) * 32 + (lane & 31);
  const int khalf = lane >> 5;

  const int nslices = (p.Cin >> 6) * ((HPM && !p.hp_hi_only) ? 3 : 1);
  const int ngroups = (L_ntaps + TG - 1) / TG;
  const int nstages = nslices * ngroups;
  // tap table in a VGPR (lane t holds tap t): the per-tap weight slab and halo offset are fetched with
  // v_readlane instead of kernarg loads, so no memory round trip (and no s_waitcnt) sits on the stage path
  int tap_tab = 0;
  if (lane < L_ntaps)
    tap_tab = p.ncls ? ((p.c_wtap[cls][lane & 3] << 16) | (p.c_dy[cls][lane & 3] * HPW + p.c_dx[cls][lane & 3]))
                     : ((p.wtap[lane] << 16) | (p.dy[lane] * HPW + p.dx[lane]));
  auto tap_slab = [&](int t) { return __builtin_amdgcn_readlane(tap_tab, t) >> 16; };
  auto tap_off = [&](int t) { return __builtin_amdgcn_readlane(tap_tab, t) & 0xffff; };
  // per-lane constant parts of the weight staging addresses: chunk e of a tap is row (tid + e*NTH) >> 3, 16-B piece & 7
  int w_goff[WPT], w_loff[WPT];
#pragma unroll
  for (int e = 0; e < WPT; ++e) {
    const int i = tid + e * NTH;
    w_goff[e] = (i >> 3) * wK + (i & 7) * 8;
    w_loff[e] = lds_off(i >> 3, i & 7);
  }

  // ---- staging helpers ------------------------------------------------------------------------------
  auto halo_src = [&](int i, int k0, bool& ok) -> const TI* {
    const int pix = i >> 3, ch = i & 7;
    const int hr = pix / HPW, hc = pix - hr * HPW;
    const int iy = iy_base + hr, ix = ix_base + hc;
    ok = (i < HP * 8) && (iy >= 0) && (iy < p.IH) && (ix >= 0) && (ix < p.IW);
    return in + (size_t)iy * row_st + (size_t)ix * pix_st + k0 + ch * 8;
  };
  auto stage_halo_sync = [&](int k0) {   // parity mode: load, split into 3 planes, store
    for (int i = tid; i < HP * 8; i += NTH) {
      bool ok;
      const TI* src = halo_src(i, k0, ok);
      const int off = lds_off(i >> 3, i & 7);
      if constexpr (SPLIT) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
        if (ok) load8(src, v);
        bf16x8 p0, p1, p2;
#pragma unroll
        for (int q = 0; q < 8; ++q) { bf16 a, b2, c2; split3_bf16(v[q], a, b2, c2); p0[q] = a; p1[q] = b2; p2[q] = c2; }
        *reinterpret_cast<bf16x8*>(ldsA + off) = p0;
        *reinterpret_cast<bf16x8*>(ldsA + C::A_BYTES + off) = p1;
        *reinterpret_cast<bf16x8*>(ldsA + 2 * C::A_BYTES + off) = p2;
      } else {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) v = *reinterpret_cast<const u32x4*>(src);
        *reinterpret_cast<u32x4*>(ldsA + off) = v;
      }
    }
  };
  auto stage_w_sync = [&](int k0, int tg0, int gcount) {
    for (int t = 0; t < gcount; ++t) {
      const bf16* base = wbase + ((size_t)tap_slab(tg0 + t) * p.Cout + n0) * wK + k0;
#pragma unroll
      for (int e = 0; e < WPT; ++e)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          *reinterpret_cast<u32x4*>(ldsB + pl * C::B_BYTES + t * BNT * 128 + w_loff[e]) =
              *reinterpret_cast<const u32x4*>(base + (size_t)pl * p.wplane + w_goff[e]);
    }
  };
  // perf mode: register prefetch (issue the global loads before the MFMA block, write LDS after it)
  u32x4 hreg[SPLIT ? 1 : HCH], wreg[SPLIT ? 1 : WCH];
  auto load_halo_regs = [&](int k0) {
#pragma unroll
    for (int e = 0; e < HCH; ++e) {
      const int i = tid + e * NTH;
      bool ok;
      const TI* src = halo_src(i, k0, ok);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x4*>(src);
      hreg[e] = v;
    }
  };
  auto store_halo_regs = [&]() {
#pragma unroll
    for (int e = 0; e < HCH; ++e) {
      const int i = tid + e * NTH;
      if (i < HP * 8) *reinterpret_cast<u32x4*>(ldsA + lds_off(i >> 3, i & 7)) = hreg[e];
    }
  };
  auto load_w_regs = [&](int k0, int tg0, int gcount) {
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t < gcount) {   // uniform
        const bf16* base = wbase + ((size_t)tap_slab(tg0 + t) * p.Cout + n0) * wK + k0;
#pragma unroll
        for (int e = 0; e < WPT; ++e) wreg[t * WPT + e] = *reinterpret_cast<const u32x4*>(base + w_goff[e]);
      }
    }
  };
  auto store_w_regs = [&](int gcount, unsigned char* dstB) {
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t < gcount) {
#pragma unroll
        for (int e = 0; e < WPT; ++e) *reinterpret_cast<u32x4*>(dstB + t * BNT * 128 + w_loff[e]) = wreg[t * WPT + e];
      }
    }
  };
  // ---- MFMA over one staged tap group -----------------------------------------------------------------
  auto compute = [&](int tg0, int gcount, const unsigned char* ldsBcur) {
    for (int t = 0; t < gcount; ++t) {
      const int toff = tap_off(tg0 + t);
      int hp[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) hp[i] = prow[i] + toff;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int chunk = ks * 2 + khalf;
        bf16x8 a[NP][FM], bq[NP][FN];
        auto load_plane = [&](int pl) {
#pragma unroll
          for (int i = 0; i < FM; ++i)
            a[pl][i] = *reinterpret_cast<const bf16x8*>(ldsA + pl * C::A_BYTES + lds_off(hp[i], chunk));
#pragma unroll
          for (int j = 0; j < FN; ++j)
            bq[pl][j] = *reinterpret_cast<const bf16x8*>(ldsBcur + pl * C::B_BYTES + t * BNT * 128 + lds_off(nrow[j], chunk));
        };
        // (the third plane's fragments are only read where its products are issued: bf16x3 skips a third of the LDS reads)
#pragma unroll
        for (int pl = 0; pl < (SPLIT ? 2 : 1); ++pl) load_plane(pl);
        if constexpr (SPLIT) {
#define PH_MM(PI, PJ)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < FM; ++i) _Pragma("unroll") for (int j = 0; j < FN; ++j)          \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PI][i], bq[PJ][j], acc[i][j], 0, 0, 0);
          if (p.prod6) { load_plane(2); PH_SPLIT_PAIRS_LO(PH_MM) }
          PH_SPLIT_PAIRS_HI(PH_MM)
#undef PH_MM
        } else if constexpr (HPM) {
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0][i]), __builtin_bit_cast(f16x8, bq[0][j]),
                                                                 acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], bq[0][j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  if constexpr (SPLIT) {
    // parity mode: simple synchronous staging (throughput is irrelevant here, LDS holds 3 planes)
    for (int sl = 0; sl < nslices; ++sl) {
      const int k0 = sl << 6;      // (split-plane modes: A and W blocks coincide)
      __syncthreads();
      stage_halo_sync(k0);
      for (int tg0 = 0; tg0 < L_ntaps; tg0 += TG) {
        const int gcount = (L_ntaps - tg0) < TG ? (L_ntaps - tg0) : TG;
        if (tg0 > 0) __syncthreads();
        stage_w_sync(k0, tg0, gcount);
        __syncthreads();
        compute(tg0, gcount, ldsB);
      }
    }
  } else if constexpr (C::DB) {
    // perf mode, weights double-buffered in LDS.  Register prefetch runs two stages ahead: at the top of stage s
    // the weights of stage s+1 (loaded during stage s-1) are written to the other B buffer and the loads of stage
    // s+2 are issued; the MFMAs of stage s follow; ONE barrier ends the stage.  Only a slice change (new halo into
    // the single A buffer) still needs the write phase between two barriers.
    auto advance = [&](int& sl_, int& tg_) {
      tg_ += TG;
      if (tg_ >= L_ntaps) { tg_ = 0; ++sl_; }
    };
    auto gc = [&](int tg_) { return (L_ntaps - tg_) < TG ? (L_ntaps - tg_) : TG; };
    int sl = 0, tg0 = 0, sl1 = 0, tg1 = 0;
    advance(sl1, tg1);
    load_halo_regs(slice_a(0));
    load_w_regs(slice_w(0), 0, gc(0));
    store_halo_regs();
    store_w_regs(gc(0), ldsB);
    if (nstages > 1) load_w_regs(slice_w(sl1), tg1, gc(tg1));
    __syncthreads();
    PH_TRACE(1);
    unsigned long long cyc_a = 0, cyc_b = 0, cyc_c = 0, cyc_d = 0, cyc_e = 0;
    const unsigned long long ql0_ = PH_CLK();
    (void)ql0_;
    for (int st = 0; st < nstages; ++st) {
      int sl2 = sl1, tg2 = tg1;
      advance(sl2, tg2);
      const bool has1 = st + 1 < nstages, has2 = st + 2 < nstages;
      const bool new_slice = has1 && sl1 != sl;
      unsigned char* Bcur = ldsB + (st & 1) * C::B_BYTES;
      unsigned char* Bnext = ldsB + ((st + 1) & 1) * C::B_BYTES;
      const unsigned long long q0_ = PH_CLK();
      if (has1) store_w_regs(gc(tg1), Bnext);
      const unsigned long long q1_ = PH_CLK();
      if (has2) load_w_regs(slice_w(sl2), tg2, gc(tg2));
      if (new_slice) load_halo_regs(slice_a(sl1));
      const unsigned long long q2_ = PH_CLK();
      compute(tg0, gc(tg0), Bcur);
      const unsigned long long q3_ = PH_CLK();
      __syncthreads();
      const unsigned long long q4_ = PH_CLK();
      if (new_slice) {
        store_halo_regs();
        __syncthreads();
      }
      const unsigned long long q5_ = PH_CLK();
      cyc_a += q1_ - q0_; cyc_b += q2_ - q1_; cyc_c += q3_ - q2_; cyc_d += q4_ - q3_; cyc_e += q5_ - q4_;
      sl = sl1; tg0 = tg1; sl1 = sl2; tg1 = tg2;
    }
    PH_TRACE_ACC(6, cyc_c); PH_TRACE_ACC(8, cyc_d); PH_TRACE_ACC(9, cyc_e);
    PH_TRACE_ACC(10, PH_CLK() - ql0_); PH_TRACE_ACC(11, (unsigned long long)nstages | (cyc_a << 8) | (cyc_b << 36));
  } else {
    // perf mode: stage s = (slice, tap group).  While the MFMAs of stage s run, the global loads of stage
    // s+1 (weights, and the next slice's halo when the slice changes) are in flight into registers; they
    // are written to LDS after the barrier that ends stage s.
    load_halo_regs(slice_a(0));
    load_w_regs(slice_w(0), 0, L_ntaps < TG ? L_ntaps : TG);
    store_halo_regs();
    store_w_regs(L_ntaps < TG ? L_ntaps : TG, ldsB);
    __syncthreads();
    PH_TRACE(1);
    unsigned long long cyc_compute = 0, cyc_bar1 = 0, cyc_write = 0;
    const unsigned long long kl0_ = PH_CLK();
    (void)kl0_;
    int sl = 0, tg0 = 0;
    for (int st = 0; st < nstages; ++st) {
      const int gcount = (L_ntaps - tg0) < TG ? (L_ntaps - tg0) : TG;
      int nsl = sl, ntg0 = tg0 + TG;
      if (ntg0 >= L_ntaps) { ntg0 = 0; nsl = sl + 1; }
      const bool has_next = st + 1 < nstages;
      const int ngcount = (L_ntaps - ntg0) < TG ? (L_ntaps - ntg0) : TG;
      if (has_next) {
        load_w_regs(slice_w(nsl), ntg0, ngcount);
        if (nsl != sl) load_halo_regs(slice_a(nsl));
      }
      const unsigned long long k0_ = PH_CLK();
      compute(tg0, gcount, ldsB);
      const unsigned long long k1_ = PH_CLK();
      __syncthreads();
      const unsigned long long k2_ = PH_CLK();
      if (has_next) {
        store_w_regs(ngcount, ldsB);
        if (nsl != sl) store_halo_regs();
      }
      __syncthreads();
      const unsigned long long k3_ = PH_CLK();
      cyc_compute += k1_ - k0_; cyc_bar1 += k2_ - k1_; cyc_write += k3_ - k2_;
      sl = nsl; tg0 = ntg0;
    }
    PH_TRACE_ACC(6, cyc_compute); PH_TRACE_ACC(8, cyc_bar1); PH_TRACE_ACC(9, cyc_write);
    PH_TRACE_ACC(10, PH_CLK() - kl0_); PH_TRACE_ACC(11, (unsigned long long)nstages);
  }

  PH_TRACE(2);
  // ---------------- epilogue: mask, BN partial statistics, (residual), store
  // accumulator register q of fragment (i,j) holds MFMA row (q&3) + 8*(q>>2) + 4*khalf, i.e. (see
  // frag_row_to_pixel) tile row 2*(wm*FM+i) + ((popc(q>>2) + khalf) & 1), column q; channel n0 + nrow[j].
  const bool full = (r0 + TH <= L_OHt) && (c0 + TW <= L_OWt);
  TO* out = reinterpret_cast<TO*>(p.out) + (size_t)b * p.OH * p.OW * p.Cout;
  const TO* resg = p.res_g ? reinterpret_cast<const TO*>(p.res_g) + (size_t)b * p.OH * p.OW * p.Cout : nullptr;
  // (res_a: a block output as the elementwise passes read it - the fp32 copy in the half-pair mode)
  const TO* resa = p.res_a ? reinterpret_cast<const TO*>(p.res_a) + (size_t)b * p.OH * p.OW * p.Cout : nullptr;
  // half-pair mode: the accumulators hold 2^11 x the sum (the W blocks' scaling) times the dz tensor's power-of-two scale
  float osc = 1.f;
  if constexpr (HPM) osc = (p.hp_hi_only ? 1.f : PH_HP_LO_INV) * (p.in_unscale ? p.in_unscale[1] : 1.f);
  float s1[FN], s2[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  constexpr int BM = TH * TW;
  constexpr int CROW = BNT * 2;                       // bytes of one pixel row of the bf16 C image in LDS
  static_assert(!HPM || BM * BNT * 4 <= C::LDS_BYTES, "half-pair mode: the fp32 C image reuses the main loop's LDS");
  unsigned char* ldsC = smem;                          // perf mode: [BM][BNT] bf16 (the main loop's LDS is free)
  float* red = reinterpret_cast<float*>(smem + ((SPLIT || HPM) ? 0 : BM * CROW));   // [WM][2][BNT]
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int rl = (wm * FM + i) * 2 + ((__popc(q >> 2) + khalf) & 1);
      const int r = r0 + rl, c = c0 + q;
      const bool valid = full || (r < L_OHt && c < L_OWt);
      const size_t o = ((size_t)(r * p.os + L_oa_h) * p.OW + (c * p.os + L_oa_w)) * p.Cout + n0;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float v = valid ? acc[i][j][q] : 0.f;
        if constexpr (HPM) v *= osc;
        s1[j] += v;
        s2[j] = __builtin_fmaf(v, v, s2[j]);      // (explicit: the compiler's contraction choice must not move the statistics)
        if constexpr (HPM) {
          // fp32 C image in LDS (the main loop's LDS is free), stored below as 16-byte chunks like the bf16 image of the perf
          // mode: one 4-byte store per value was 64 store instructions per wave and tile (+ 128 residual loads in a dgrad)
          *reinterpret_cast<float*>(ldsC + (rl * TW + q) * (BNT * 4) + nrow[j] * 4) = v;
        } else if constexpr (SPLIT) {
          if (valid) {
            if (resg) {
              float g = ldf(resg + o + nrow[j]);
              if (resa) g = (ldf(resa + o + nrow[j]) > 0.f) ? g : 0.f;
              v += g;
            }
            stf(out + o + nrow[j], v);
          }
        } else {
          *reinterpret_cast<bf16*>(ldsC + (rl * TW + q) * CROW + nrow[j] * 2) = (bf16)v;
        }
      }
    }
  }
  if constexpr (!SPLIT && !HPM) {
    // coalesced store: the C tile is re-read from LDS as 16-B chunks, BNT/8 consecutive lanes per pixel
    __syncthreads();
    PH_TRACE(3);
    constexpr int CPR = BNT / 8;                       // chunks per pixel row
    for (int id = tid; id < BM * CPR; id += NTH) {
      const int m = id / CPR, ch = id - m * CPR;
      const int r = r0 + (m >> 4), c = c0 + (m & 15);
      if (!(full || (r < L_OHt && c < L_OWt))) continue;
      const size_t o = ((size_t)(r * p.os + L_oa_h) * p.OW + (c * p.os + L_oa_w)) * p.Cout + n0 + ch * 8;
      bf16x8 v = *reinterpret_cast<const bf16x8*>(ldsC + m * CROW + ch * 16);
      if (resg) {
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(resg + o);
        if (resa) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(resa + o);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = (bf16)((float)v[k] + ((float)a[k] > 0.f ? (float)g[k] : 0.f));
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = (bf16)((float)v[k] + (float)g[k]);
        }
      }
      *reinterpret_cast<bf16x8*>(out + o) = v;
    }
  }
  if constexpr (HPM) {
    __syncthreads();
    constexpr int CPRF = BNT / 4;                      // 16-byte chunks (4 floats) per pixel row
    for (int id = tid; id < BM * CPRF; id += NTH) {
      const int m = id / CPRF, ch = id - m * CPRF;
      const int r = r0 + (m >> 4), c = c0 + (m & 15);
      if (!(full || (r < L_OHt && c < L_OWt))) continue;
      const size_t o = ((size_t)(r * p.os + L_oa_h) * p.OW + (c * p.os + L_oa_w)) * p.Cout + n0 + ch * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(ldsC + m * (BNT * 4) + ch * 16);
      if (resg) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(resg + o);
        if (resa) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(resa + o);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] += a[k] > 0.f ? g[k] : 0.f;
        } else {
          v += g;
        }
      }
      *reinterpret_cast<f32x4*>(out + o) = v;
    }
  }
  PH_TRACE(4);
  if (p.stats) {
    if constexpr (SPLIT || HPM) __syncthreads();   // all MFMA reads of LDS done; smem is reused as float[WM][2][BNT]
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float a1 = s1[j] + __shfl_xor(s1[j], 32, 64);
      float a2 = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (khalf == 0 && wn * FN * 32 + j * 32 + (lane & 31) == nrow[j]) {
        // WN > 1: waves with the same wm cover disjoint channel ranges, so each (wm, channel) has one writer
        red[(wm * 2 + 0) * BNT + nrow[j]] = a1;
        red[(wm * 2 + 1) * BNT + nrow[j]] = a2;
      }
    }
    __syncthreads();
    if (tid < 2 * BNT) {
      const int which = tid / BNT, n = tid % BNT;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(w * 2 + which) * BNT + n];
      const size_t part = (size_t)b * gridDim.x + tile;
      p.stats[(part * 2 + which) * p.Cout + n0 + n] = v;
    }
  }
  PH_TRACE(5);
}

template <typename T, int S, int TH, int BNT, int WM, int WN, int FM, int FN, int TG>
int launch_cfg(const PhTapConv& p, hipStream_t st) {
  using C = TapCfg<T, S, TH, BNT, WM, WN, FM, FN, TG>;
  auto kern = tapconv_kernel<T, S, TH, BNT, WM, WN, FM, FN, TG>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess)
      return PH_ELAUNCH;
    attr_done = true;
  }
  if (p.ncls && (S != 1 || p.ncls < 2 || p.ncls > 4 || p.stats)) return PH_EINVAL;
  dim3 grid(cdiv(p.OHt, TH) * cdiv(p.OWt, 16), p.Cout / BNT, p.B * (p.ncls ? p.ncls : 1));
  void* tok = nullptr;
  if (ph_prof_on()) {   // algorithmic FLOPs: 2 * positions * Cout * ntaps * Cin (merged classes: summed)
    double work = 2.0 * p.B * p.OHt * p.OWt * (double)p.Cout * p.ntaps * p.Cin;
    if (p.ncls) {
      work = 0;
      for (int k = 0; k < p.ncls; ++k) work += 2.0 * p.B * p.c_OHt[k] * p.c_OWt[k] * (double)p.Cout * p.c_ntaps[k] * p.Cin;
    }
    ph_prof_begin2(S == 2 ? PH_CLS_TAPCONV_S2 : (BNT == 64 ? PH_CLS_TAPCONV_N64 : PH_CLS_TAPCONV_N128),
                   work, ph_tapconv_bytes(p, S, sizeof(T)), st, &tok);
  }
  hipLaunchKernelGGL(kern, grid, dim3(C::NTH), C::LDS_BYTES, st, p);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

}  // namespace
// algorithmic HBM bytes of a tap-conv launch: the input pixels its taps touch (at most the whole input view; all planes
// of a masked stride-2 forward), the output positions it writes (+ the residual it reads; all classes of a masked
// stride-2 dgrad), the weight slabs of its taps - each counted once
double ph_tapconv_bytes(const PhTapConv& p, int S, int es) {
  const double in_pix = (double)p.B * ((double)p.OHt * S + 2) * ((double)p.OWt * S + 2);
  const double in_full = (double)p.B * p.IH * p.IW;
  const double in = (in_pix < in_full ? in_pix : in_full) * p.Cin * es * (p.m_groups ? p.m_groups : 1);
  double pos = (double)p.OHt * p.OWt, taps = p.ntaps;
  if (p.ncls) {      // merged output-parity classes: every class's positions and weight slabs
    pos = 0; taps = 0;
    for (int k = 0; k < p.ncls; ++k) { pos += (double)p.c_OHt[k] * p.c_OWt[k]; taps += p.c_ntaps[k]; }
  }
  const double out = (double)p.B * pos * p.Cout * es * (p.res_g ? (p.res_a ? 3.0 : 2.0) : 1.0);
  return in + out + taps * p.Cin * p.Cout * 2.0;
}
namespace {

template <typename T>
int launch_T(const PhTapConv& p, int S, hipStream_t st) {
  constexpr bool SPLIT = is_f32<T>::value;      // (T = hp16 runs the perf-mode configurations)
  constexpr int TG = SPLIT ? 1 : 3;   // parity mode stages one tap at a time (3 planes must fit 160 KB LDS)
  if (S == 1) {
    if (p.Cout % 128 == 0) {
      // perf mode: 256 x 128 tile, 8 waves (weights staged once per 256 pixels: LDS traffic per FLOP halves);
      // parity mode keeps the 128 x 128 tile (3 planes must fit the 160 KB LDS)
      if constexpr (SPLIT) return launch_cfg<T, 1, 8, 128, 2, 2, 2, 2, TG>(p, st);
      else return launch_cfg<T, 1, 16, 128, 4, 2, 2, 2, TG>(p, st);
    }
    return launch_cfg<T, 1, 16, 64, 4, 1, 2, 2, TG>(p, st);
  } else {
    if (p.Cout % 128) return PH_EINVAL;   // stride-2 forward convs of ResNet-18 all have Cout >= 128
    if constexpr (SPLIT) return launch_cfg<T, 2, 2, 128, 1, 4, 1, 1, 1>(p, st);
    else return launch_cfg<T, 2, 8, 128, 2, 4, 2, 1, 3>(p, st);   // 128 x 128 tile, 8 waves, 121 KB LDS
  }
}

}  // namespace

// number of statistic partial rows a launch writes: B * tiles
int ph_tapconv_stat_parts(const PhTapConv* p, int S, int prec) {
  if (prec == PH_PREC_BF16 && S == 2 && ph_tap6b_switch(-1)) {
    // (ntaps == 0: a sizing query with B / OHt / OWt / Cout only - the larger of the candidates' counts)
    if (p->ntaps == 0 && p->Cout % 128 == 0) return std::max(p->B * cdiv(p->OHt, 8) * cdiv(p->OWt, 16), ph_tapconv6b_stat_parts(p));
    if (p->ntaps != 0 && ph_tapconv6b_eligible(p)) return ph_tapconv6b_stat_parts(p);
  }
  if (ph_tapconv2_tile_h(p, S, prec)) return ph_tapconv2_stat_parts(p);
  // (half-pair mode on the third-generation kernel: one partial row per persistent workgroup, like the second generation)
  if (prec == PH_PREC_FP16X3 && S == 1 && ph_tap3_switch(-1) && ph_tapconv3_eligible(p)) return ph_tapconv2_stat_parts(p);
  if ((prec == PH_PREC_FP16X3 || prec == PH_PREC_FP16X1) && S == 1 && ph_tap5_switch(-1) && ph_tapconv5_eligible(p)) return ph_tapconv5_stat_parts(p);
  if (prec == PH_PREC_FP16X3 && S == 2 && ph_tap6_switch(-1)) {
    // (ntaps == 0: a sizing query with B / OHt / OWt / Cout only - resnet_plan.hip - take the larger of the two kernels' counts)
    const int first_gen = p->B * cdiv(p->OHt, 8) * cdiv(p->OWt, 16);
    if (p->ntaps == 0) return std::max(first_gen, p->Cout % 128 == 0 ? ph_tapconv6_stat_parts(p) : 0);
    if (ph_tapconv6_eligible(p)) return ph_tapconv6_stat_parts(p);
  }
  const bool perf_cfg = prec == PH_PREC_BF16 || prec == PH_PREC_FP16X3;
  const int TH = (S == 1) ? ((p->Cout % 128 == 0 && !perf_cfg) ? 8 : 16) : (perf_cfg ? 8 : 2);
  return p->B * cdiv(p->OHt, TH) * cdiv(p->OWt, 16);
}

int ph_tapconv_launch(const PhTapConv* p, int S, int prec, hipStream_t st) {
  if (p->Cin % 64 || p->Cout % 64 || p->ntaps < 1 || p->ntaps > 9 || (S != 1 && S != 2)) return PH_EINVAL;
  // perf mode, 3x3 / stride 2 forward over the un-masked descriptor (the callers skip ph_tapconv2_setup_s2_fwd for it): conv_tap6b.hip
  if (S == 2 && prec == PH_PREC_BF16 && ph_tap6b_switch(-1) && ph_tapconv6b_eligible(p)) return ph_tapconv6b_launch(p, st);
  if (ph_tapconv2_tile_h(p, S, prec)) return ph_tapconv2_launch(p, st);
  if (p->in_scale || p->m_groups) return PH_EINVAL;   // in-LDS BatchNorm + ReLU / masked tap grids: second-generation kernels only
  if (prec == PH_PREC_BF16) return launch_T<bf16>(*p, S, st);
  if (prec == PH_PREC_FP16X3 || prec == PH_PREC_FP16X1) {
    PhTapConv q = *p;
    q.hp_hi_only = prec == PH_PREC_FP16X1;
    // dense 3x3 stride-1, Cout % 128 == 0: the third-generation kernel's half-pair form (conv_tap3.hip)
    if (S == 1 && ph_tap3_switch(-1) && ph_tapconv3_eligible(&q)) return ph_tapconv3_launch_hp(&q, st);
    // dense 3x3 stride-1, Cin = Cout = 64 (layer 1): conv_tap5.hip
    if (S == 1 && ph_tap5_switch(-1) && ph_tapconv5_eligible(&q)) return ph_tapconv5_launch(&q, st);
    // 3x3 / stride 2 forward (layers 2-4 conv1): conv_tap6.hip
    if (S == 2 && prec == PH_PREC_FP16X3 && ph_tap6_switch(-1) && ph_tapconv6_eligible(&q)) return ph_tapconv6_launch(&q, st);
    return launch_T<hp16>(q, S, st);
  }
  if (PH_IS_SPLIT_PREC(prec)) {
    PhTapConv q = *p;
    q.prod6 = prec == PH_PREC_BF16X6;
    return launch_T<float>(q, S, st);
  }
  return PH_EINVAL;
}

#ifdef PH_TAP_TRACE
extern "C" int ph_debug_tap_trace(unsigned long long* host_out, int nwg) {
  if (nwg > PH_TRACE_WGS) nwg = PH_TRACE_WGS;
  if (hipDeviceSynchronize() != hipSuccess) return PH_ELAUNCH;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ph_tap_trace), (size_t)nwg * 12 * sizeof(unsigned long long)) == hipSuccess
             ? PH_OK : PH_ELAUNCH;
}
#endif
