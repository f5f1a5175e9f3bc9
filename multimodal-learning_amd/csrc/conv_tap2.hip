// Tap-convolution implicit GEMM, second generation (perf mode bf16, stride-1 3x3): ONE wave per SIMD, 128 x 64 wave
// tiles, LDS-DMA operand streams, persistent workgroups.
//
// Why (measured with the phase tracer and its ablation builds, layer-4 3x3 conv, cycles per 3-tap stage and CU):
// the first-generation kernel (conv_tap.hip: 8 waves, 64 x 64 wave tiles) needs 3072 MFMA cycles per SIMD but also
// 2640 cycles of ds_read_b128 for the fragments (one fragment read per MFMA) plus the ds_write_b128 of the weight
// refill - the LDS, not the matrix pipe, paces it (6240 cycles per stage).  A 128 x 64 wave tile re-uses every B
// fragment for 4 and every A fragment for 2 MFMAs: 0.75 fragment reads per MFMA instead of 1.  Its 128 accumulator
// registers only fit with the whole 512-entry register file, so the workgroup is 4 waves (one per SIMD) and all
// latency hiding happens inside the wave:
//   * fragment reads run one k-step ahead in a second register set (also across the tap barrier);
//   * weights (one tap = 16 KiB per step, ring of 4) and the next slice's halo (second A buffer) arrive by LDS-DMA
//     (global_load_lds_dwordx4), issued three taps / up to eight taps ahead - no VGPR staging, no ds_write.  The DMA
//     is emitted as inline asm on purpose: the compiler serialises every ds_read behind an outstanding LDS-DMA it
//     knows of (s_waitcnt vmcnt(0), "may alias"), and register-staged prefetch across a loop back-edge gets the
//     same conservative vmcnt(0).  Completion is tracked by hand: vmcnt retires in order, so "all but the youngest
//     4 (+2 halo) DMAs" at the end of a tap means the tap after next is complete; the s_barrier publishes it;
//   * the workgroup is persistent: the operand streams never drain at a tile edge, the epilogue of tile k runs
//     while tile k+1's first operands are already in LDS.
// The LDS side of a DMA is linear (M0 + lane*16), so the XOR swizzle of the LDS image is applied to each lane's
// SOURCE (row, chunk); out-of-image halo pixels read a zero page.
//
// Same GEMM view, LDS image and epilogue semantics as conv_tap.hip (PhTapConv; forward and stride-1 dgrad with the
// fused residual mask; per-tile BatchNorm partial sums).  Workgroup tile: (8*WM) x 16 pixels x (64*WN) channels.
#include "ph_common.h"
#include <cstdlib>
#include <type_traits>
#include "ph_kernels.h"
#include "tap_common.h"

namespace {
#ifdef PH_TAP_TRACE
__device__ unsigned long long ph_tap_trace[PH_TRACE_WGS * 12];
#endif

__device__ const u32x4 ph_zero16[4] = {};   // source of out-of-image halo pixels
// ... when the input's BatchNorm + ReLU is applied in LDS (PhTapConv::in_scale): padding must be zero AFTER that map, for
// any scale / shift.  Quiet NaNs do it: fma(NaN, s, b) = NaN and the ReLU's v_max_f32(NaN, 0) returns the number, 0.
__device__ const u32x4 ph_nan16[4] = {{0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}, {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u},
                                      {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}, {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}};

// relu(x * s + h) on the 8 bf16 values of one 16-byte chunk (channel 2q in the low half of dword q), result rounded to
// bf16 like the stand-alone bn_apply pass stores it
__device__ __forceinline__ u32x4 bn_relu_chunk(u32x4 v, const f32x4& sA, const f32x4& sB, const f32x4& hA, const f32x4& hB) {
  u32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = __builtin_bit_cast(float, v[q] << 16), x1 = __builtin_bit_cast(float, v[q] & 0xffff0000u);
    const float s0 = q < 2 ? sA[2 * q] : sB[2 * q - 4], s1 = q < 2 ? sA[2 * q + 1] : sB[2 * q - 3];
    const float h0 = q < 2 ? hA[2 * q] : hB[2 * q - 4], h1 = q < 2 ? hA[2 * q + 1] : hB[2 * q - 3];
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 r;
    r[0] = (bf16)fmaxf(x0 * s0 + h0, 0.f);
    r[1] = (bf16)fmaxf(x1 * s1 + h1, 0.f);
    o[q] = __builtin_bit_cast(unsigned, r);
  }
  return o;
}

typedef __attribute__((address_space(3))) unsigned char lds_uchar;

// one LDS-DMA wave-instruction: lane l copies 16 B from its global address g to LDS byte lds_addr + 16*l
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_addr) {
#ifdef PH_ABL_NODMA   // timing ablation only (results are garbage)
  asm volatile("" : : "s"(lds_addr), "v"(g) : "memory");
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_addr), "v"(g) : "memory");
#endif
}
// 4-byte form: used only to pull cache lines towards the L2 (the LDS side is a scratch row nobody reads)
__device__ __forceinline__ void lds_dma4(const void* g, unsigned lds_addr) {
#ifndef PH_ABL_NODMA
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" : : "s"(lds_addr), "v"(g) : "memory");
#endif
}
#define PH_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define PH_BARRIER() asm volatile("s_barrier" ::: "memory")

// Configurations: <WM, WN, FM, RES>.  Wave tile (32*FM) pixels x 64 channels, workgroup tile 16 x 16 pixels x (64*WN)
// channels.  RES = false: weights stream through a ring of 4 taps (any Cin).  RES = true (Cin = Cout = 64, ResNet
// layer 1): all 9 taps stay resident in LDS for the lifetime of the workgroup - no weight traffic and no barrier
// inside a tile.
template <int WM, int WN, int FM_, bool RES_>
struct Tap2Cfg {
  static constexpr int FM = FM_, FN = 2, NTAPS = 9, RING = 4;
  static constexpr bool RES = RES_;
  static constexpr int TH = WM * FM * 2, TW = 16;
  static constexpr int BNT = WN * 64;
  static constexpr int HPH = TH + 2, HPW = TW + 2, HP = HPH * HPW;
  static constexpr int A_BYTES = (((HP + 1) / 2 * 256) + 1023) / 1024 * 1024;   // whole 1-KiB DMA pieces
  static constexpr int NHD = A_BYTES / 1024;       // halo DMA wave-instructions per slice
  static constexpr int NHE = (NHD + 3) / 4;        // ... per wave
  static constexpr int TAPB = BNT * 128;           // bytes of one tap's weight block
  static constexpr int NBE = TAPB / 1024 / 4;      // weight DMA wave-instructions per wave and tap
  static constexpr int HALO_TAPS = 6;              // taps 0..5 of a slice issue the next slice's halo ...
  static constexpr int HPT = (NHE + HALO_TAPS - 1) / HALO_TAPS;   // ... HPT pieces per wave each
  static constexpr int LDS_MAIN = 2 * A_BYTES + (RES ? NTAPS : RING) * TAPB;
  static constexpr int SS_OFF = LDS_MAIN + 4096;       // [2][512] floats: scale / shift of the input's BatchNorm (in_scale)
  static constexpr int LDS_BYTES = LDS_MAIN + 4096 + 4096;   // + the workgroup's BatchNorm partial row (Cout <= 512) + SS
  static constexpr int NTH = WM * WN * 64;
  static constexpr int NHALF = FM / 2;                       // the C tile is staged in NHALF passes of 2 fragments
  static constexpr int C_BYTES = WM * 4 * TW * BNT * 2;      // one pass of the C tile (bf16)
  static constexpr int RED_BYTES = WM * 2 * BNT * 4;
  static_assert(TH == 16, "16 x 16 pixel tiles");
  static_assert(NTH == 256, "4 waves: one per SIMD");
  static_assert(TAPB % 4096 == 0, "weight tap block splits into 1-KiB pieces over 4 waves");
  static_assert(HPT == 2, "the vmcnt bookkeeping below assumes 2 halo pieces per wave and tap");
  static_assert(C_BYTES + RED_BYTES <= A_BYTES, "the epilogue stages through one A buffer");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int WM, int WN, int FM_, bool RES, bool MASKED = false>
__global__ __launch_bounds__(256) void tapconv2_kernel(PhTapConv p) {
  using C = Tap2Cfg<WM, WN, FM_, RES>;
  static_assert(!MASKED || !RES, "masked taps: streamed-weights configuration only");
  constexpr int FM = C::FM, FN = C::FN, TH = C::TH, TW = C::TW, BNT = C::BNT;
  constexpr int HPW = C::HPW, HP = C::HP, NTH = C::NTH, NTAPS = C::NTAPS;
  typedef __bf16 T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(lds_uchar*)smem;   // LDS byte address of the allocation
  // LDS: A[0] | A[1] | B ring (RING taps)
  constexpr int B_BASE = 2 * C::A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int khalf = lane >> 5;
  const int tiles_w = (p.OWt + TW - 1) / TW;
  const int tiles_sp = tiles_w * ((p.OHt + TH - 1) / TH);
  const int nblk = p.Cout / BNT;
  const int total = tiles_sp * nblk * p.B;
  const long pix_st = p.in_pix_stride ? p.in_pix_stride : p.Cin;
  const long row_st = p.in_row_stride ? p.in_row_stride : (long)p.IW * p.Cin;
  const long img_st = p.in_img_stride ? p.in_img_stride : (long)p.IH * p.IW * p.Cin;
  const bf16* wbase = reinterpret_cast<const bf16*>(p.w);
  PH_TRACE(0);

  // ---- tile list of this (persistent) workgroup.  Linear tile id -> (spatial tile fastest, Cout block, image).
  // The workgroups of one XCD (blockIdx.x % 8: hardware round-robin) walk one contiguous eighth of the list, so
  // neighbouring halos and the weight block of a Cout block are shared inside one L2.
  struct TileCtx { int tile, r0, c0, n0, b, iy_base, ix_base; const T* in; };
  // q = a / d for 0 <= a < 2^22 with a precomputed float reciprocal and one correction step (the decode below runs
  // once per tile on the critical path; hardware integer division costs ~40 instructions per quotient)
  const float rcp_sp = 1.0f / (float)tiles_sp, rcp_nb = 1.0f / (float)nblk, rcp_tw = 1.0f / (float)tiles_w;
  auto fdiv = [](int a, int d, float rcp) {
    int q = (int)((float)a * rcp);
    int r = a - q * d;
    if (r >= d) ++q;
    if (r < 0) --q;
    return q;
  };
  auto decode = [&](int t) -> TileCtx {
    TileCtx c;
    const int rest = fdiv(t, tiles_sp, rcp_sp);
    c.tile = t - rest * tiles_sp;
    c.b = fdiv(rest, nblk, rcp_nb);
    c.n0 = (rest - c.b * nblk) * BNT;
    const int trow = fdiv(c.tile, tiles_w, rcp_tw);
    c.r0 = trow * TH;
    c.c0 = (c.tile - trow * tiles_w) * TW;
    c.iy_base = c.r0 + p.iy0;
    c.ix_base = c.c0 + p.ix0;
    c.in = reinterpret_cast<const T*>(p.in) + (size_t)c.b * img_st;
    return c;
  };
  const int G = gridDim.x;
  const bool xcd_map = (G & 7) == 0 && G < total;
  const int per_xcd = (total + 7) >> 3;
  auto tile_id = [&](int k) -> int {   // k-th tile of this workgroup, -1 when the list is exhausted
    if (!xcd_map) {
      const int t = (int)blockIdx.x + k * G;
      return t < total ? t : -1;
    }
    const int local = ((int)blockIdx.x >> 3) + k * (G >> 3);
    const int t = ((int)blockIdx.x & 7) * per_xcd + local;
    return (local < per_xcd && t < total) ? t : -1;
  };

  const int nsl_c = p.Cin >> 6;      // 64-channel slices of the real channel count
  const int nsl_sh = nsl_c == 1 ? 0 : (nsl_c == 2 ? 1 : (nsl_c == 4 ? 2 : 3));
  const int nslices = MASKED ? nsl_c * 4 : nsl_c;      // masked: the four pixel-parity planes of a stride-2 forward
  // tap table in a VGPR (lane t holds tap t; masked: lane 9 g + t holds grid tap t of plane g), read with v_readlane: no
  // memory access on the tap path
  int tap_tab = 0;
  if constexpr (MASKED) {
    if (lane < 36) tap_tab = p.m_slab[lane / 9][lane % 9] << 16;
  } else {
    if (lane < NTAPS) tap_tab = p.wtap[lane] << 16;
  }
  auto tap_slab = [&](int g, int t) { return __builtin_amdgcn_readlane(tap_tab, MASKED ? g * 9 + t : t) >> 16; };
  // slice index -> (plane, channel offset, element offset of the plane's input view)
  auto slice_grp = [&](int sl) { return MASKED ? (sl >> nsl_sh) : 0; };
  auto slice_k0 = [&](int sl) { return MASKED ? ((sl & (nsl_c - 1)) << 6) : (sl << 6); };
  auto slice_in_off = [&](int sl) -> long { return MASKED ? p.m_in_off[sl >> nsl_sh] + slice_k0(sl) : (long)(sl << 6); };

  // ---- per-lane DMA sources.  Piece d of an LDS image covers row pairs 4d..4d+3; lane l writes slot l & 15 of row
  // pair rp = 4d + (l >> 4), which the swizzle assigns to (row 2*rp + (u >> 3), chunk u & 7), u = (l & 15) ^ (rp & PH_SWZ_MASK)
  int wb_off[C::NBE];   // weights: byte offset of this lane's chunk inside a [BNT][Cin] tap block
#pragma unroll
  for (int e = 0; e < C::NBE; ++e) {
    const int rp = (wave * C::NBE + e) * 4 + (lane >> 4), u = (lane & 15) ^ (rp & PH_SWZ_MASK);
    wb_off[e] = ((2 * rp + (u >> 3)) * p.Cin + (u & 7) * 8) * 2;
  }
  // halo piece h = wave + 4e: byte offset of this lane's chunk relative to the tile's halo origin pixel, and the halo
  // (row, column) packed for the per-tile in-image test (-1: padding of the last piece)
  int h_off[C::NHE], h_rc[C::NHE];
#pragma unroll
  for (int e = 0; e < C::NHE; ++e) {
    const int rp = (wave + 4 * e) * 4 + (lane >> 4), u = (lane & 15) ^ (rp & PH_SWZ_MASK);
    const int pix = 2 * rp + (u >> 3), hr = pix / HPW, hc = pix - hr * HPW;
    h_rc[e] = pix < HP ? ((hr << 8) | hc) : -1;
    h_off[e] = (int)(((long)hr * row_st + (long)hc * pix_st + (u & 7) * 8) * 2);
  }
  // bit e of the result: piece e of this lane lies inside the image for a tile whose halo origin is (iy_base, ix_base).
  // The valid halo rows / columns of a tile are two scalar bit masks; a piece costs two shifts and two ands.
  auto halo_mask = [&](int iy_base, int ix_base) {
    auto range_bits = [](int lo, int hi) -> unsigned {   // bits lo..hi-1 set (0 <= lo, hi <= 31)
      return hi > lo ? ((hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u)) : 0u;
    };
    const int r_lo = iy_base < 0 ? -iy_base : 0, r_hi = (p.IH - iy_base) < C::HPH ? (p.IH - iy_base) : C::HPH;
    const int c_lo = ix_base < 0 ? -ix_base : 0, c_hi = (p.IW - ix_base) < HPW ? (p.IW - ix_base) : HPW;
    const unsigned rowok = range_bits(r_lo, r_hi < 0 ? 0 : r_hi), colok = range_bits(c_lo, c_hi < 0 ? 0 : c_hi);
    int m = 0;
#pragma unroll
    for (int e = 0; e < C::NHE; ++e) {
      const unsigned ok = (rowok >> ((h_rc[e] >> 8) & 31)) & (colok >> (h_rc[e] & 31)) & (h_rc[e] >= 0 ? 1u : 0u);
      m |= (int)(ok & 1u) << e;
    }
    return m;
  };
  const bool fuse_in = p.in_scale != nullptr;
  const unsigned char* zero_src = reinterpret_cast<const unsigned char*>(fuse_in ? ph_nan16 : ph_zero16);
  // BatchNorm + ReLU of the INPUT applied in LDS (PhTapConv::in_scale): every wave transforms the halo pieces it issued
  // itself (its own counted vmcnt wait covers them; the next barrier publishes the result).  The chunk of lane l in
  // piece wave + 4e holds channel group cgrp of the slice for every e (the swizzle term (4 * piece) & 15 drops out).
  float* ss = reinterpret_cast<float*>(smem + C::SS_OFF);
  const int cgrp = (((lane & 15) ^ ((wave * 4 + (lane >> 4)) & 15)) & 7) * 8;
  auto xform_halo = [&](int abuf, int k0) {
    const f32x4 sA = *reinterpret_cast<const f32x4*>(ss + k0 + cgrp), sB = *reinterpret_cast<const f32x4*>(ss + k0 + cgrp + 4);
    const f32x4 hA = *reinterpret_cast<const f32x4*>(ss + 512 + k0 + cgrp), hB = *reinterpret_cast<const f32x4*>(ss + 512 + k0 + cgrp + 4);
#pragma unroll
    for (int e = 0; e < C::NHE; ++e)
      if (wave + 4 * e < C::NHD) {
        u32x4* a = reinterpret_cast<u32x4*>(smem + abuf * C::A_BYTES + (wave + 4 * e) * 1024 + lane * 16);
        *a = bn_relu_chunk(*a, sA, sB, hA, hB);
      }
  };

  // ---- per-lane fragment addressing
  int prow[FM];   // halo pixel index of this lane's row in M fragment i (fragment f covers tile rows 2f, 2f+1)
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int fr, c;
    frag_row_to_pixel(lane & 31, fr, c);
    prow[i] = ((wm * FM + i) * 2 + fr) * HPW + c;
  }
  int nrow[FN], bx[FN];   // channel row of this lane in N fragment j and its offset in a tap block at k-step 0
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    nrow[j] = (wn * FN + j) * 32 + (lane & 31);
    bx[j] = lds_off(nrow[j], khalf);
  }

  f32x16 acc[FM][FN];
#ifdef PH_ABL_MFMA16   // timing ablation only (results are garbage): same FLOPs through v_mfma_f32_16x16x32_bf16
  f32x4 abl[FM][FN][2];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) { abl[i][j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; abl[i][j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#endif
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  };

  // ---- epilogue of one tile: mask, BN partial statistics, (residual), store.  Accumulator register q of fragment
  // (i,j) holds MFMA row (q&3) + 8*(q>>2) + 4*khalf, i.e. (frag_row_to_pixel) tile row 2*(wm*FM+i) + ((popc(q>>2) +
  // khalf) & 1), column q; channel n0 + nrow[j].  Lanes l, l^1 hold neighbouring channels of the same pixels, so they
  // swap one value of each column pair (2m, 2m+1) by DPP and each lane stores one [even channel, odd channel] 4-byte
  // word itself.  (Staging the tile through LDS for 16-byte stores measured the same time per tile on layers 2-4 and
  // 5 % slower on layer 1, at 50 more registers and four more barriers; 2-byte LDS staging writes were 50 cycles/value.)
  unsigned long long ep_w = 0;   // trace build: cycles in the store part of the epilogues
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  // BatchNorm partial sums: kept in registers ACROSS the tiles of this workgroup (per column parity; combined at the
  // flush) and folded into a per-workgroup LDS row [Cout/BNT][2][BNT] only when the Cout block changes or the tile
  // list ends - ONE partial row per workgroup instead of one per tile (bn_finalize combined 4096 rows for layer 1),
  // and no per-tile reduction / barriers in the epilogue.  Fixed order of additions: bitwise reproducible.
  float* stat_acc = reinterpret_cast<float*>(smem + C::LDS_MAIN);
  for (int i = tid; i < 2 * p.Cout; i += NTH) stat_acc[i] = 0.f;   // (published by the prologue barrier)
  if (p.in_scale) {   // (read before any LDS-DMA is in flight; published by the barrier below)
    float* ss_ = reinterpret_cast<float*>(smem + C::SS_OFF);
    for (int i = tid; i < p.Cin; i += NTH) { ss_[i] = p.in_scale[i]; ss_[512 + i] = p.in_shift[i]; }
    __syncthreads();
  }
  f32x2 s1[FN], s2[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) { s1[j] = f32x2{0.f, 0.f}; s2[j] = f32x2{0.f, 0.f}; }
  auto flush_stats = [&](int n0, unsigned char* scratch) {   // all threads; `scratch`: an A buffer nobody reads
    float* red = reinterpret_cast<float*>(scratch);          // [WM][2][BNT]
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const float t1 = s1[j][0] + s1[j][1], t2 = s2[j][0] + s2[j][1];
      const float a1 = t1 + __shfl_xor(t1, 32, 64);
      const float a2 = t2 + __shfl_xor(t2, 32, 64);
      if (khalf == 0) {   // waves with the same wm cover disjoint channel ranges: one writer per (wm, channel)
        red[(wm * 2 + 0) * BNT + nrow[j]] = a1;
        red[(wm * 2 + 1) * BNT + nrow[j]] = a2;
      }
      s1[j] = f32x2{0.f, 0.f};
      s2[j] = f32x2{0.f, 0.f};
    }
    __syncthreads();
    if (tid < 2 * BNT) {
      const int which = tid / BNT, n = tid % BNT;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(w * 2 + which) * BNT + n];
      stat_acc[((n0 / BNT) * 2 + which) * BNT + n] += v;
    }
    __syncthreads();
  };
  // FULL (tile inside the output) and RM (0: no residual, 1: + res_g, 2: + res_g masked by res_a > 0) are compile-time:
  // as a run-time test inside the unrolled loops the residual cost a taken branch per stored word.
  auto epilogue = [&](const TileCtx& tc, auto fullc, auto rmc) {
    constexpr bool FULL = decltype(fullc)::value;   // the tile lies completely inside the output: no masking
    constexpr int RM = decltype(rmc)::value;
    const int r0 = tc.r0, c0 = tc.c0, n0 = tc.n0;
    const int oa_h = p.oa_h, oa_w = p.oa_w;
    T* out = reinterpret_cast<T*>(p.out) + (size_t)tc.b * p.OH * p.OW * p.Cout;
    const T* resg = reinterpret_cast<const T*>(p.res_g) + (size_t)tc.b * p.OH * p.OW * p.Cout;
    const T* resa = reinterpret_cast<const T*>(p.res_a) + (size_t)tc.b * p.OH * p.OW * p.Cout;
    // byte selector of v_perm_b32 {neighbour's packed pair, own packed pair}: even lanes build [own lo | neighbour lo],
    // odd lanes [neighbour hi | own hi]
    const unsigned psel = (lane & 1) ? 0x03020706u : 0x05040100u;
    // direct form: after the lane-pair exchange every lane owns one [even channel, odd channel] word of one pixel and
    // stores it itself (a wave-instruction writes four 64-B runs); no LDS staging, no barrier before the statistics.
    // The residual words of a fragment row (dgrad) are all requested before the first is used.
    {
      const unsigned long long e0_ = PH_CLK();
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        size_t o[8];
        bool mine[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int fr = (__popc(m >> 1) + khalf) & 1;     // columns 2m, 2m+1 share q >> 2 = m >> 1
          const int r = r0 + (wm * FM + i) * 2 + fr, c = c0 + 2 * m + (lane & 1);
          mine[m] = FULL || (r < p.OHt && c < p.OWt);
          o[m] = ((size_t)(r * p.os + oa_h) * p.OW + (c * p.os + oa_w)) * p.Cout + n0 + (nrow[0] & ~1);
        }
        bf16x2 rg[8][FN], ra[8][FN];
        if constexpr (RM > 0) {
#pragma unroll
          for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              rg[m][j] = bf16x2{(bf16)0.f, (bf16)0.f};
              ra[m][j] = bf16x2{(bf16)1.f, (bf16)1.f};
              if (mine[m]) {
                rg[m][j] = *reinterpret_cast<const bf16x2*>(resg + o[m] + j * 32);
                if constexpr (RM > 1) ra[m][j] = *reinterpret_cast<const bf16x2*>(resa + o[m] + j * 32);
              }
            }
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            f32x2 v = {acc[i][j][2 * m], acc[i][j][2 * m + 1]};
            if constexpr (!FULL) {
              const int fr = (__popc(m >> 1) + khalf) & 1;
              const int r = r0 + (wm * FM + i) * 2 + fr;
              v[0] = (r < p.OHt && c0 + 2 * m < p.OWt) ? v[0] : 0.f;
              v[1] = (r < p.OHt && c0 + 2 * m + 1 < p.OWt) ? v[1] : 0.f;
            }
            s1[j] += v;
            s2[j] = __builtin_elementwise_fma(v, v, s2[j]);      // (explicit fused multiply-add: see conv_tap.hip)
            bf16x2 own;
            own[0] = (bf16)v[0];
            own[1] = (bf16)v[1];
            const unsigned x = __builtin_bit_cast(unsigned, own);
            const unsigned y = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   // lane ^ 1
            bf16x2 w = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(y, x, psel));
            if constexpr (RM > 0) {
              w[0] = (bf16)((float)w[0] + ((RM < 2 || (float)ra[m][j][0] > 0.f) ? (float)rg[m][j][0] : 0.f));
              w[1] = (bf16)((float)w[1] + ((RM < 2 || (float)ra[m][j][1] > 0.f) ? (float)rg[m][j][1] : 0.f));
            }
#ifdef PH_ABL_NOSTORE   // timing ablation only
            asm volatile("" ::"v"(w));
#else
            if (mine[m]) *reinterpret_cast<bf16x2*>(out + o[m] + j * 32) = w;
#endif
          }
        }
      }
      ep_w += PH_CLK() - e0_;
    }
  };

  const int rmode = p.res_g ? (p.res_a ? 2 : 1) : 0;
  auto epilogue_any = [&](const TileCtx& tc) {
    const bool full = (tc.r0 + TH <= p.OHt) && (tc.c0 + TW <= p.OWt);
    auto with_full = [&](auto fullc) {
      if (rmode == 0) epilogue(tc, fullc, std::integral_constant<int, 0>{});
      else if (rmode == 1) epilogue(tc, fullc, std::integral_constant<int, 1>{});
      else epilogue(tc, fullc, std::integral_constant<int, 2>{});
    };
    if (full) with_full(std::true_type{});
    else with_full(std::false_type{});
  };

  // ---- the tap stream.  Tile (outer loop) -> 64-channel slice -> 9 taps (fully unrolled: tap offsets, the piece of
  // the next halo a tap issues and the ring arithmetic are then compile-time).  Every tap issues exactly NBE weight
  // pieces (tap +3 of the stream) and, in taps 0..5, HPT pieces of the next slice's halo into the other A buffer; past
  // the end of the stream the same pieces are issued from the current tile again (harmless refills of buffers nobody
  // reads) so that the vmcnt bookkeeping has no special cases.
  TileCtx tcur = decode(tile_id(0));
  int tn = tile_id(1);
  bool nvalid = tn >= 0;
  TileCtx tnext = tcur;
  if (nvalid) tnext = decode(tn);
  int hm_cur = halo_mask(tcur.iy_base, tcur.ix_base);
  int hm_next = nvalid ? halo_mask(tnext.iy_base, tnext.ix_base) : hm_cur;
  auto halo_base = [&](const TileCtx& tc, long off) {   // address of element `off` (channel offset of the slice, + the
    // group's view offset when masked) of the halo origin pixel (may lie outside the image)
    return reinterpret_cast<const unsigned char*>(tc.in) + ((long)tc.iy_base * row_st + (long)tc.ix_base * pix_st + off) * 2;
  };
  auto w_base = [&](int n0, int k0, int g, int tap) {
    return reinterpret_cast<const unsigned char*>(wbase + ((size_t)tap_slab(g, tap) * p.Cout + n0) * p.Cin + k0);
  };

  // masked grid: the weight stream runs over the LIVE taps only - live tap l of a tile is (plane, slice, i-th live tap
  // of the plane), planes in order, L = 1, 2, 2, 4 live taps per slice.  The iterator below is three live taps ahead
  // of the MFMAs (scalar state; it enters the next tile while the MFMAs still finish this one).
  int wq_p = 0, wq_s = 0, wq_i = 0, wq_n0 = tcur.n0;
  auto wq_addr = [&]() {
    const int tapid = (int)((0x4310004100430004ull >> (16 * wq_p + 4 * wq_i)) & 15);   // grid taps 4 | 3 4 | 1 4 | 0 1 3 4
    return w_base(wq_n0, wq_s << 6, wq_p, tapid);
  };
  auto wq_next = [&]() {
    if (++wq_i == ((0x4221 >> (4 * wq_p)) & 15)) {
      wq_i = 0;
      if (++wq_s == nsl_c) {
        wq_s = 0;
        if (++wq_p == 4) { wq_p = 0; wq_n0 = nvalid ? tnext.n0 : tcur.n0; }
      }
    }
  };

  // prologue: first halo and the first RING-1 taps of weights
  {
    const unsigned char* hb = halo_base(tcur, slice_in_off(0));
#pragma unroll
    for (int e = 0; e < C::NHE; ++e)
      if (wave + 4 * e < C::NHD)
        lds_dma16(((hm_cur >> e) & 1) ? hb + h_off[e] : zero_src, lds0 + (wave + 4 * e) * 1024);
#pragma unroll
    for (int j = 0; j < (RES ? NTAPS : C::RING - 1); ++j) {
      const unsigned char* wb = MASKED ? wq_addr() : w_base(tcur.n0, 0, 0, j);
      if constexpr (MASKED) wq_next();
#pragma unroll
      for (int e = 0; e < C::NBE; ++e)
        lds_dma16(wb + wb_off[e], lds0 + B_BASE + j * C::TAPB + (wave * C::NBE + e) * 1024);
    }
  }
  zero_acc();
  PH_WAIT_VMCNT(0);
  if (fuse_in) xform_halo(0, 0);
  PH_BARRIER();
  PH_TRACE(1);

  // fragment registers: two sets (k-steps alternate), and the LDS addresses of the tap in flight.  k-step ks of a
  // fragment row sits at (its address at ks = 0) ^ (ks << 5): one v_xor per read.
  bf16x8 fa[2][FM], fb[2][FN];
  int aaddr[FM], bxs[FN];
  auto tap_addr = [&](int toff, int abuf, int slot) {
#pragma unroll
    for (int i = 0; i < FM; ++i) aaddr[i] = abuf * C::A_BYTES + lds_off(prow[i] + toff, khalf);
#pragma unroll
    for (int j = 0; j < FN; ++j) bxs[j] = B_BASE + slot * C::TAPB + bx[j];
  };
  // ---- one k-step = 8 MFMAs (4 x 2 fragments), each of the first six followed by one fragment read of the NEXT
  // k-step into the other register set, in the order the next k-step consumes them; F0..F7 are filler statements (one
  // DMA piece, or a few VALU of address arithmetic) that ride in the shadow of the MFMA issued just before them.  The
  // MFMAs are inline asm with the accumulators pinned to AGPRs ("+a"): left to itself the register allocator keeps
  // the loop-carried accumulators in VGPRs and copies all 128 into and out of AGPRs around every tap.
  // sched_barrier(0) after every slot keeps the compiler from regrouping (it cannot classify the asm as an MFMA);
  // the s_waitcnt lgkmcnt(N) in front of each MFMA is still inserted by the compiler, which sees the ds_reads.
  // (Macros, not lambdas: clang rejects asm operands that name captured variables inside a generic lambda.)
#ifdef PH_ABL_NOMFMA   // timing ablation only
#define PH_MM(CB, I, J) asm volatile("" : "+a"(acc[I][J]) : "v"(fa[CB][I]), "v"(fb[CB][J]))
#elif defined(PH_ABL_MFMA16)
#define PH_MM(CB, I, J)                                                                                         \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1"               \
               : "+a"(abl[I][J][0]), "+a"(abl[I][J][1]) : "v"(fa[CB][I]), "v"(fb[CB][J]))
#else
#define PH_MM(CB, I, J)                                                                                  \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(fa[CB][I]), "v"(fb[CB][J]))
#endif
#ifdef PH_ABL_NOLDS   // timing ablation only
#define PH_LDA(AA, I, KS) (fa[0][0])
#else
#define PH_LDA(AA, I, KS) (*reinterpret_cast<const bf16x8*>(smem + ((AA)[I] ^ ((KS) << 5))))
#endif
#define PH_SB() __builtin_amdgcn_sched_barrier(0)
#define PH_KSTEP(CB, NB, KS, F0, F1, F2, F3, F4, F5, F6, F7)             \
  PH_MM(CB, 0, 0); fa[NB][0] = PH_LDA(aaddr, 0, KS); F0; PH_SB();         \
  PH_MM(CB, 0, 1); fb[NB][0] = PH_LDA(bxs, 0, KS); F1; PH_SB();           \
  PH_MM(CB, 1, 0); fb[NB][1] = PH_LDA(bxs, 1, KS); F2; PH_SB();           \
  PH_MM(CB, 1, 1); fa[NB][1] = PH_LDA(aaddr, 1, KS); F3; PH_SB();         \
  PH_MM(CB, 2, 0); fa[NB][2] = PH_LDA(aaddr, 2, KS); F4; PH_SB();         \
  PH_MM(CB, 2, 1); fa[NB][3] = PH_LDA(aaddr, 3, KS); F5; PH_SB();         \
  PH_MM(CB, 3, 0); F6; PH_SB();                                           \
  PH_MM(CB, 3, 1); F7; PH_SB()
#define PH_KSTEP2(CB, NB, KS, F0, F1, F2, F3)                            \
  PH_MM(CB, 0, 0); fa[NB][0] = PH_LDA(aaddr, 0, KS); F0; PH_SB();         \
  PH_MM(CB, 0, 1); fb[NB][0] = PH_LDA(bxs, 0, KS); F1; PH_SB();           \
  PH_MM(CB, 1, 0); fb[NB][1] = PH_LDA(bxs, 1, KS); F2; PH_SB();           \
  PH_MM(CB, 1, 1); fa[NB][1] = PH_LDA(aaddr, 1, KS); F3; PH_SB()
#define PH_NOP_ ((void)0)
  tap_addr(0, 0, 0);
#pragma unroll
  for (int i = 0; i < FM; ++i) fa[0][i] = PH_LDA(aaddr, i, 0);
#pragma unroll
  for (int j = 0; j < FN; ++j) fb[0][j] = PH_LDA(bxs, j, 0);

  unsigned long long cyc_c = 0, cyc_b = 0, cyc_e = 0, cyc_s = 0, cyc_x = 0;
  const unsigned long long ql0_ = PH_CLK();
  (void)ql0_;
  int acur = 0, gt = 0;
  for (int k = 0;; ++k) {   // tiles of this workgroup
    // One 64-channel slice = 9 grid taps.  `mk` / `mkn`: the taps that are LIVE in this slice / in the slice that
    // follows it in the stream (all nine, except in the masked grid of a stride-2 forward).  Literals at every call
    // site: after inlining and unrolling every liveness test folds - no run-time branch, no second arm merging the 128
    // accumulators.  In the masked grid a dead tap issues nothing but its share of the next slice's halo (and, in
    // front of a live tap, that tap's first fragment reads): no MFMAs, no weight DMAs, no wait, no barrier, no ring
    // slot.  `gt` counts LIVE taps there: the weight ring, its three-taps-ahead prefetch (wq_*) and the end-of-tap
    // wait + barrier are those of the dense stream over the live taps alone, so a dead tap costs a few issue cycles.
    // The halo pieces go out in grid taps 0..2 (four per wave each) and are complete at the end-of-tap wait of grid
    // tap 4, which is live in every plane and is the last live tap of its slice.
    // (Not a generic lambda: clang rejects asm operands that name captured variables inside one.)
    auto slice_body = [&](const int sl, const unsigned mk, const unsigned mkn) __attribute__((always_inline)) {
      const unsigned long long q0_ = PH_CLK();   // (per-slice stamps only: a stamp costs ~50 cycles with its s_waitcnt)
      const bool last_sl = sl + 1 == nslices;
      // source of the next slice's halo (next slice of this tile / slice 0 of the next tile / past the end: this tile)
      const bool h_next_tile = last_sl && nvalid;
      const int nsl = last_sl ? 0 : sl + 1;                       // index of the next slice inside its tile
      const unsigned char* hb = h_next_tile ? halo_base(tnext, slice_in_off(0)) : halo_base(tcur, slice_in_off(nsl));
      const int hm = h_next_tile ? hm_next : hm_cur;
      // weights of the taps that wrap into the next slice / tile
      const int wn0 = (last_sl && nvalid) ? tnext.n0 : tcur.n0;
      const int wk0 = slice_k0(nsl);
      const int gc = slice_grp(sl), wg = slice_grp(nsl);
      const unsigned long long q0b_ = PH_CLK();
      cyc_s += q0b_ - q0_;
      auto live = [&](int u) { return u < NTAPS ? ((mk >> u) & 1u) != 0 : ((mkn >> (u - NTAPS)) & 1u) != 0; };
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        const bool lv = live(t), lv1 = live(t + 1);
        // weight pieces of stream tap gt+3 -> ring slot (gt+3) & 3 (released by the barrier that ended tap gt-1)
        const unsigned char* wb;
        if constexpr (MASKED) {
          wb = nullptr;
          if (lv) { wb = wq_addr(); wq_next(); }
        } else {
          wb = (t + 3 < NTAPS) ? w_base(tcur.n0, slice_k0(sl), gc, t + 3) : w_base(wn0, wk0, wg, t + 3 - NTAPS);
        }
        constexpr int HT = MASKED ? 3 : C::HALO_TAPS, HPP = MASKED ? 4 : 2;   // halo: grid taps 0..HT-1, HPP pieces each
        const unsigned wdst = lds0 + B_BASE + ((gt + 3) & 3) * C::TAPB + wave * C::NBE * 1024;
        const unsigned hdst = lds0 + (acur ^ 1) * C::A_BYTES + wave * 1024;
#define PH_DMA_B(E) lds_dma16(wb + wb_off[E], wdst + (E) * 1024)
#define PH_DMA_H(E)                                                                                         \
  do {                                                                                                      \
    if ((E) < C::NHE && wave + 4 * (E) < C::NHD)                                                            \
      lds_dma16(((hm >> (E)) & 1) ? hb + h_off[(E) < C::NHE ? (E) : 0] : zero_src, hdst + (E) * 4096);      \
  } while (0)
        const int toff_n = ((t + 1) % NTAPS) / 3 * HPW + ((t + 1) % NTAPS) % 3;
        const int abuf_n = t + 1 == NTAPS ? (acur ^ 1) : acur;
        if (!lv) {
          // ---- dead tap of a masked grid
          if constexpr (FM == 4) {
            if (t < HT) { PH_DMA_H(HPP * t); PH_DMA_H(HPP * t + 1); PH_DMA_H(HPP * t + 2); PH_DMA_H(HPP * t + 3); }
            if (lv1) {   // the first fragments of the next tap, read where a live tap reads them
              tap_addr(toff_n, abuf_n, gt & 3);
#pragma unroll
              for (int i = 0; i < FM; ++i) fa[0][i] = PH_LDA(aaddr, i, 0);
#pragma unroll
              for (int j = 0; j < FN; ++j) fb[0][j] = PH_LDA(bxs, j, 0);
            }
            PH_SB();
          }
        } else if constexpr (FM == 4) {
          // ---- 32 MFMAs; the tap's DMA pieces ride between them
          PH_KSTEP(0, 1, 1, PH_NOP_, PH_DMA_B(0), PH_NOP_, PH_NOP_, PH_DMA_B(1), PH_NOP_, PH_NOP_, PH_DMA_B(2));
          PH_KSTEP(1, 0, 2, PH_NOP_, PH_NOP_, PH_DMA_B(3), PH_NOP_, PH_NOP_,
                   if (t < HT) PH_DMA_H(HPP * t), PH_NOP_, if (t < HT) PH_DMA_H(HPP * t + 1));
          PH_KSTEP(0, 1, 3, PH_NOP_, if (MASKED && t < HT) PH_DMA_H(HPP * t + 2), PH_NOP_, PH_NOP_,
                   if (MASKED && t < HT) PH_DMA_H(HPP * t + 3), PH_NOP_, PH_NOP_, PH_NOP_);
          tap_addr(toff_n, abuf_n, (gt + 1) & 3);
          PH_KSTEP(1, 0, 0, PH_NOP_, PH_NOP_, PH_NOP_, PH_NOP_, PH_NOP_, PH_NOP_, PH_NOP_, PH_NOP_);
        } else {
          // ---- 16 MFMAs (resident weights: ring slot = tap)
          PH_KSTEP2(0, 1, 1, PH_NOP_, PH_NOP_, PH_NOP_, if (t < C::HALO_TAPS) PH_DMA_H(2 * t));
          PH_KSTEP2(1, 0, 2, PH_NOP_, PH_NOP_, PH_NOP_, if (t < C::HALO_TAPS) PH_DMA_H(2 * t + 1));
          PH_KSTEP2(0, 1, 3, PH_NOP_, PH_NOP_, PH_NOP_, PH_NOP_);
          if (t == NTAPS - 2) {   // the next tile's halo must be visible before the last tap reads its first fragments
            PH_WAIT_VMCNT(0);
            PH_BARRIER();
          }
          tap_addr(toff_n, abuf_n, (t + 1) % NTAPS);
          PH_KSTEP2(1, 0, 0, PH_NOP_, PH_NOP_, PH_NOP_, PH_NOP_);
        }
        if constexpr (!RES) {
          // ---- tap end: the weight pieces of stream tap gt+2 (and a halo that is due) have landed once at most the
          // pieces issued during this tap are still in flight; the barrier publishes them and releases ring slot gt & 3
          if constexpr (MASKED) {
            static_assert(!MASKED || (HT * HPP >= C::NHE), "halo piece schedule of the masked grid");
            if (lv) { if (t < HT) PH_WAIT_VMCNT(8); else PH_WAIT_VMCNT(4); }
            if (t == 4) { if (fuse_in) xform_halo(acur ^ 1, wk0); }   // every halo piece (taps 0..2) is older than tap 4
            if (lv) PH_BARRIER();
          } else {
            if (t < C::HALO_TAPS) PH_WAIT_VMCNT(6); else PH_WAIT_VMCNT(4);
            // all halo pieces of the next slice (issued in taps 0..5) have landed once tap 6 has passed its wait: apply
            // the input's BatchNorm + ReLU to this wave's pieces; the barriers of taps 6 and 7 publish them before the
            // last k-step of tap 8 reads the next slice's first fragments
            if (t == C::HALO_TAPS) { if (fuse_in) xform_halo(acur ^ 1, wk0); }
            PH_BARRIER();
          }
        }
        if (!MASKED || lv) ++gt;
      }
      acur ^= 1;
      cyc_c += PH_CLK() - q0b_;
    };
    if constexpr (!MASKED) {
      for (int sl = 0; sl < nslices; ++sl) slice_body(sl, 511u, 511u);
    } else {
      // stride-2 forward: the four pixel-parity planes in a fixed order, plane (a, b) with its (1 + a)(1 + b) live grid
      // taps (ph_tapconv2_setup_s2_fwd): rows gy = 1 (a = 0) or 0, 1 (a = 1), columns likewise.  The last slice of a
      // plane is followed by the next plane's mask (the last plane's by plane 0 of the next tile).
      constexpr unsigned M0 = 0x010u, M1 = 0x018u, M2 = 0x012u, M3 = 0x01Bu;
      static_assert(((M0 & M1 & M2 & M3) & 0x10u) && !((M0 | M1 | M2 | M3) & ~0x1Bu), "tap 4 live and last, tap 2 dead");
      const int nl = nsl_c - 1;
      for (int s = 0; s < nl; ++s) slice_body(s, M0, M0);
      slice_body(nl, M0, M1);
      for (int s = 0; s < nl; ++s) slice_body(nsl_c + s, M1, M1);
      slice_body(nsl_c + nl, M1, M2);
      for (int s = 0; s < nl; ++s) slice_body(2 * nsl_c + s, M2, M2);
      slice_body(2 * nsl_c + nl, M2, M3);
      for (int s = 0; s < nl; ++s) slice_body(3 * nsl_c + s, M3, M3);
      slice_body(3 * nsl_c + nl, M3, M0);
    }
    const unsigned long long q2b_ = PH_CLK();
    if constexpr (RES) PH_BARRIER();   // no per-tap barrier in this configuration: all waves are done with the A buffer
    const unsigned long long q3_ = PH_CLK();
    cyc_b += q3_ - q2b_;
    epilogue_any(tcur);
    if (p.stats && (!nvalid || tnext.n0 != tcur.n0)) flush_stats(tcur.n0, smem + (acur ^ 1) * C::A_BYTES);
    zero_acc();
    cyc_e += PH_CLK() - q3_;
    if (!nvalid) break;
    const unsigned long long x0_ = PH_CLK();
    tcur = tnext;
    hm_cur = hm_next;
    tn = tile_id(k + 2);
    nvalid = tn >= 0;
    if (nvalid) {
      tnext = decode(tn);
      hm_next = halo_mask(tnext.iy_base, tnext.ix_base);
    }
    cyc_x += PH_CLK() - x0_;
  }
  PH_WAIT_VMCNT(0);   // the refills issued past the end of the stream must not outlive the workgroup's LDS
  if (p.stats) {      // this workgroup's partial row [2][Cout] (zeros for Cout blocks it never visited)
    for (int i = tid; i < 2 * p.Cout; i += NTH) {
      const int which = i / p.Cout, ch = i - which * p.Cout;
      p.stats[((size_t)blockIdx.x * 2 + which) * p.Cout + ch] = stat_acc[((ch / BNT) * 2 + which) * BNT + ch % BNT];
    }
  }
  PH_TRACE(5);
  PH_TRACE_ACC(6, cyc_c); PH_TRACE_ACC(8, cyc_b); PH_TRACE_ACC(3, cyc_e); PH_TRACE_ACC(9, ep_w); PH_TRACE_ACC(4, cyc_s); PH_TRACE_ACC(7, cyc_x); (void)cyc_s; (void)cyc_x; (void)ep_w;
  PH_TRACE_ACC(10, PH_CLK() - ql0_); PH_TRACE_ACC(11, (unsigned long long)gt);
}


// ---------------------------------------------------------------------------------------------------------------
// Layer-1 configuration (Cin = Cout = 64), two wave groups per workgroup.
//
// With 64 output channels a wave tile is 64 x 64 (64 accumulator registers), the main loop of a 16 x 16 tile is only
// 144 MFMAs (4608 cycles) and the epilogue - ~14 VALU per output pair plus the 4-byte stores - is nearly as long: the
// one-wave-per-SIMD kernel above spent about half of every tile outside the matrix pipe (measured 700 TFLOP/s).
// Here the workgroup has 8 waves = two groups of four (one wave of each group per SIMD).  Each group owns ONE halo
// buffer and walks its own tiles; all 9 taps of weights (72 KiB) are resident and shared.  The groups run half a tile
// out of phase: while group A streams the MFMAs of its tile, group B stores the previous tile, zeroes its
// accumulators and lets the LDS-DMA of its next halo land - the s_barrier at the end of every half-phase is the only
// synchronisation (it publishes B's halo to B's waves and releases A's halo buffer for A's next DMA).  VALU / store
// work of one wave and MFMAs of the other wave on the same SIMD issue concurrently, so the matrix pipe sees a
// more continuous stream.  Measured (phase tracer, B = 64, 128 x 128 x 64): matrix phase 6000-7000 cycles (floor 4608: the
// LDS delivers one ds_read_b128 per MFMA and is saturated), store phase 5300-6700; the older group wins the SIMD's
// issue arbitration and is ~1000 cycles faster in both phases (s_setprio only moves that advantage to the other
// group, the sum stays).  81-85 us per launch against 102-110 us for the one-wave-per-SIMD configuration on the same
// box; the kernel moves 304 MB, so ~60 us is its HBM floor.  Registers: 256 VGPRs per wave (two waves per SIMD).
struct L1Cfg {
  static constexpr int FM = 2, FN = 2, WM = 4, NTAPS = 9, TH = 16, TW = 16, BNT = 64;
  static constexpr int HPH = TH + 2, HPW = TW + 2, HP = HPH * HPW;
  static constexpr int A_BYTES = (((HP + 1) / 2 * 256) + 1023) / 1024 * 1024;
  static constexpr int NHD = A_BYTES / 1024, NHE = (NHD + 3) / 4;
  static constexpr int TAPB = BNT * 128;
  static constexpr int B_BASE = 2 * A_BYTES;
  static constexpr int LDS_MAIN = B_BASE + NTAPS * TAPB;
  static constexpr int SS_OFF = LDS_MAIN + 8 * 2 * BNT * 4;      // [2][64] floats: scale / shift of the input's BatchNorm
  static constexpr int LDS_BYTES = SS_OFF + 2 * BNT * 4;         // + the final BatchNorm reduce [8 waves][2][64] + SS
  static constexpr int NTH = 512;
  static constexpr int NWP = NTAPS * TAPB / 1024 / 8;            // weight DMA pieces per wave in the prologue
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(NTAPS * TAPB % (8 * 1024) == 0, "weights split into 1-KiB pieces over 8 waves");
};

__global__ __launch_bounds__(512) void tapconv2_l1_kernel(PhTapConv p) {
  using C = L1Cfg;
  constexpr int FM = C::FM, FN = C::FN, TH = C::TH, TW = C::TW, BNT = C::BNT, HPW = C::HPW, HP = C::HP;
  constexpr int NTAPS = C::NTAPS, B_BASE = C::B_BASE;
  typedef __bf16 T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(lds_uchar*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wm = wave & 3;   // group, wave inside the group (= its 4-row band of the tile)
  const int khalf = lane >> 5;
  const int tiles_w = (p.OWt + TW - 1) / TW;
  const int tiles_sp = tiles_w * ((p.OHt + TH - 1) / TH);
  const int total = tiles_sp * p.B;
  const long pix_st = p.in_pix_stride ? p.in_pix_stride : p.Cin;
  const long row_st = p.in_row_stride ? p.in_row_stride : (long)p.IW * p.Cin;
  const long img_st = p.in_img_stride ? p.in_img_stride : (long)p.IH * p.IW * p.Cin;
  const bf16* wbase = reinterpret_cast<const bf16*>(p.w);
  PH_TRACE(0);

  struct TileCtx { int r0, c0, b, iy_base, ix_base; const T* in; };
  const float rcp_sp = 1.0f / (float)tiles_sp, rcp_tw = 1.0f / (float)tiles_w;
  auto fdiv = [](int a, int d, float rcp) {
    int q = (int)((float)a * rcp);
    int r = a - q * d;
    if (r >= d) ++q;
    if (r < 0) --q;
    return q;
  };
  auto decode = [&](int t) -> TileCtx {
    TileCtx c;
    c.b = fdiv(t, tiles_sp, rcp_sp);
    const int tile = t - c.b * tiles_sp;
    const int trow = fdiv(tile, tiles_w, rcp_tw);
    c.r0 = trow * TH;
    c.c0 = (tile - trow * tiles_w) * TW;
    c.iy_base = c.r0 + p.iy0;
    c.ix_base = c.c0 + p.ix0;
    c.in = reinterpret_cast<const T*>(p.in) + (size_t)c.b * img_st;
    return c;
  };
  // the workgroup's tile list (same XCD-contiguous order as above); group g takes entries g, g + 2, g + 4, ...
  const int G = gridDim.x;
  const bool xcd_map = (G & 7) == 0 && G < total;
  const int per_xcd = (total + 7) >> 3;
  auto tile_id = [&](int k) -> int {
    if (!xcd_map) {
      const int t = (int)blockIdx.x + k * G;
      return t < total ? t : -1;
    }
    const int local = ((int)blockIdx.x >> 3) + k * (G >> 3);
    const int t = ((int)blockIdx.x & 7) * per_xcd + local;
    return (local < per_xcd && t < total) ? t : -1;
  };
  int K = 0;
  while (tile_id(K) >= 0) ++K;
  const int n_mine = (K - grp + 1) >> 1;       // tiles of this group

  int tap_tab = 0;
  if (lane < NTAPS) tap_tab = p.wtap[lane] << 16;

  // halo DMA sources of this lane: piece h = wm + 4e of the group's halo image (swizzle applied to the source).  Row
  // pair rp = 4h + (lane >> 4) advances by 16 per e, so the swizzled slot u is the same for all e and the pixel index
  // advances by 32 = one halo row + 14 columns: (row, column, byte offset) are carried incrementally instead of kept
  // in 22 registers per lane (two waves per SIMD: the register file is the scarce resource here).
  const int rp0 = wm * 4 + (lane >> 4), u0 = (lane & 15) ^ (rp0 & PH_SWZ_MASK);
  const int pix0 = 2 * rp0 + (u0 >> 3);
  int hr_base = pix0 / HPW, hc_base = pix0 - hr_base * HPW;
  int hoff_base = (int)(((long)hr_base * row_st + (long)hc_base * pix_st + (u0 & 7) * 8) * 2);
  const int hd_row = (int)((row_st + 14 * pix_st) * 2), hd_wrap = (int)((row_st - 18 * pix_st) * 2);
  const bool fuse_in = p.in_scale != nullptr;
  const unsigned char* zero_src = reinterpret_cast<const unsigned char*>(fuse_in ? ph_nan16 : ph_zero16);
  const unsigned a_lds = lds0 + grp * C::A_BYTES;
  // BatchNorm + ReLU of the input applied in LDS (PhTapConv::in_scale), see tapconv2_kernel: each wave transforms the
  // pieces it issued, after its vmcnt(0) at the end of the store phase and before the barrier that opens the matrix phase
  auto xform_halo = [&]() {
    const float* ss = reinterpret_cast<const float*>(smem + C::SS_OFF);
    const int cg8 = (u0 & 7) * 8;
    const f32x4 sA = *reinterpret_cast<const f32x4*>(ss + cg8), sB = *reinterpret_cast<const f32x4*>(ss + cg8 + 4);
    const f32x4 hA = *reinterpret_cast<const f32x4*>(ss + BNT + cg8), hB = *reinterpret_cast<const f32x4*>(ss + BNT + cg8 + 4);
#pragma unroll
    for (int e = 0; e < C::NHE; ++e)
      if (wm + 4 * e < C::NHD) {
        u32x4* a = reinterpret_cast<u32x4*>(smem + grp * C::A_BYTES + (wm + 4 * e) * 1024 + lane * 16);
        *a = bn_relu_chunk(*a, sA, sB, hA, hB);
      }
  };
  // The next halo is issued piece by piece BETWEEN the blocks of the epilogue (halo_piece): issued back to back, the
  // 11 wave-instructions of 1 KiB each wait ~2400 cycles for the address/data path to accept them.
  unsigned h_rowok = 0, h_colok = 0;
  const unsigned char* h_hb = nullptr;
  int h_hr = 0, h_hc = 0, h_o = 0;
  auto halo_begin = [&](const TileCtx& tc) {
    auto range_bits = [](int lo, int hi) -> unsigned {   // bits lo..hi-1 set (0 <= lo, hi <= 31)
      return hi > lo ? ((hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u)) : 0u;
    };
    const int iy = tc.iy_base, ix = tc.ix_base;
    const int r_lo = iy < 0 ? -iy : 0, r_hi = (p.IH - iy) < C::HPH ? (p.IH - iy) : C::HPH;
    const int c_lo = ix < 0 ? -ix : 0, c_hi = (p.IW - ix) < HPW ? (p.IW - ix) : HPW;
    h_rowok = range_bits(r_lo, r_hi < 0 ? 0 : r_hi);
    h_colok = range_bits(c_lo, c_hi < 0 ? 0 : c_hi);
    h_hb = reinterpret_cast<const unsigned char*>(tc.in) + ((long)iy * row_st + (long)ix * pix_st) * 2;
    h_hr = hr_base; h_hc = hc_base; h_o = hoff_base;
    asm volatile("" : "+v"(h_hr), "+v"(h_hc), "+v"(h_o));   // (opaque: no precomputed per-piece tables)
  };
  auto halo_piece = [&](int e) {   // e = 0 .. NHE-1 in order
    if (wm + 4 * e < C::NHD) {
      const bool ok = ((h_rowok >> (h_hr & 31)) & (h_colok >> h_hc) & 1u) != 0 && h_hr < C::HPH;
      lds_dma16(ok ? h_hb + h_o : zero_src, a_lds + (wm + 4 * e) * 1024);
    }
    h_hc += 14; h_hr += 1; h_o += hd_row;
    if (h_hc >= HPW) { h_hc -= HPW; h_hr += 1; h_o += hd_wrap; }
  };
  auto issue_halo = [&](const TileCtx& tc) {   // all pieces at once (prologue)
    halo_begin(tc);
#pragma unroll
    for (int e = 0; e < C::NHE; ++e) halo_piece(e);
  };

  // With ONE halo buffer per group the DMA of the next halo cannot start before the group's matrix phase has ended,
  // so its memory latency would sit in the store phase (measured: the store phase then waits ~2 us for HBM).  During
  // the matrix phase every wave therefore touches the next halo's cache lines (one 4-byte LDS-DMA per 128-byte pixel
  // row into a scratch row): the real DMA one phase later is served by the L2.
  auto prefetch_halo = [&](const TileCtx& tc) {
    int ln = lane;
    asm volatile("" : "+v"(ln));   // (opaque: the per-lane halo coordinates are recomputed, not kept live / spilled -
                                   // a spill reload here waits, vmcnt(0), for the touch that was just issued)
#pragma unroll
    for (int z = 0; z < 2; ++z) {
      const int q = wm + 4 * z;             // 6 x 64 lanes cover the 324 halo pixels
      if (q < (HP + 63) / 64) {
        const int pidx = q * 64 + ln;
        int hr = (pidx * 3641) >> 16;       // pidx / 18 for pidx < 384
        const int hc = pidx - hr * HPW;
        hr = hr < C::HPH ? hr : C::HPH - 1;
        int iy = tc.iy_base + hr, ix = tc.ix_base + hc;
        iy = iy < 0 ? 0 : (iy >= p.IH ? p.IH - 1 : iy);
        ix = ix < 0 ? 0 : (ix >= p.IW ? p.IW - 1 : ix);
        lds_dma4(reinterpret_cast<const unsigned char*>(tc.in) + ((long)iy * row_st + (long)ix * pix_st) * 2,
                 lds0 + C::LDS_MAIN + wave * 256);
      }
    }
  };

  // fragment addressing
  int prow[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int fr, c;
    frag_row_to_pixel(lane & 31, fr, c);
    prow[i] = ((wm * FM + i) * 2 + fr) * HPW + c;
  }
  int nrow[FN], bx[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    nrow[j] = j * 32 + (lane & 31);
    bx[j] = lds_off(nrow[j], khalf);
  }
  // Accumulators: D = W_frag x X_frag (the MFMA operands swapped), i.e. TRANSPOSED tiles: lane l holds pixel l % 32 of
  // the fragment and register q holds channel (q & 3) + 8 * (q >> 2) + 4 * khalf of the 32-channel block.  A lane
  // then owns whole groups of 4 consecutive channels of ONE pixel: the output is packed with v_cvt_pk_bf16_f32 straight
  // from neighbouring registers, one v_permlane32_swap per packed pair gives every lane 8 consecutive channels, and the
  // store is 16 bytes per lane - 4 instructions per 4 outputs where the pixel-major layout needed 9 per 2 (accumulator
  // reads, DPP exchange, byte permute, 4-byte stores).  The BatchNorm sums become per-lane, per-channel registers
  // (64 of them), reduced across the 32 pixel lanes once at the end of the kernel.  The accumulators live in VGPRs
  // (two waves per SIMD: 256 registers per wave, no AGPR round trips in the epilogue) and the first k-step of a
  // tile multiplies into a zero constant instead of zeroed registers.
  f32x16 acc[FM][FN];
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 s1[FN][8], s2[FN][8];   // [j][q / 2]: channels j*32 + chan(q), chan(q + 1) - over ALL tiles of this wave
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int q = 0; q < 8; ++q) { s1[j][q] = f32x2{0.f, 0.f}; s2[j][q] = f32x2{0.f, 0.f}; }
  int pfr, pcc;   // this lane's pixel inside a fragment's 2 x 16 patch
  frag_row_to_pixel(lane & 31, pfr, pcc);

  // FULL (tile inside the output) and RM (0: no residual, 1: + res_g, 2: + res_g masked by res_a > 0) are compile-time:
  // as run-time tests inside the unrolled loops they cost a branch per stored word.
  auto epilogue = [&](const TileCtx& tc, auto fullc, auto rmc, const bool halo) {
    constexpr bool FULL = decltype(fullc)::value;
    constexpr int RM = decltype(rmc)::value;
    const size_t img = (size_t)tc.b * p.OH * p.OW * p.Cout;
    T* out = reinterpret_cast<T*>(p.out) + img;
    const T* resg = reinterpret_cast<const T*>(p.res_g) + img;
    const T* resa = reinterpret_cast<const T*>(p.res_a) + img;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int r = tc.r0 + (wm * FM + i) * 2 + pfr, c = tc.c0 + pcc;
      const bool mine = FULL || (r < p.OHt && c < p.OWt);
      const unsigned o = (unsigned)(((r * p.os + p.oa_h) * p.OW + (c * p.os + p.oa_w)) * p.Cout) + 8 * khalf;
      unsigned rg[FN][2][4], ra[FN][2][4];   // residual words: 8 channels = 4 packed pairs per (j, gp)
      if constexpr (RM > 0) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            u32x4 tg = {0u, 0u, 0u, 0u}, ta = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            if (mine) {
              tg = *reinterpret_cast<const u32x4*>(resg + o + j * 32 + gp * 16);
              if constexpr (RM > 1) ta = *reinterpret_cast<const u32x4*>(resa + o + j * 32 + gp * 16);
            }
            rg[j][gp][0] = tg[0]; rg[j][gp][1] = tg[1]; rg[j][gp][2] = tg[2]; rg[j][gp][3] = tg[3];
            ra[j][gp][0] = ta[0]; ra[j][gp][1] = ta[1]; ra[j][gp][2] = ta[2]; ra[j][gp][3] = ta[3];
          }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        unsigned P[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            f32x2 v = {acc[i][j][4 * g + 2 * h2], acc[i][j][4 * g + 2 * h2 + 1]};
            if constexpr (!FULL) {
              v[0] = mine ? v[0] : 0.f;
              v[1] = mine ? v[1] : 0.f;
            }
            s1[j][2 * g + h2] += v;
            s2[j][2 * g + h2] = __builtin_elementwise_fma(v, v, s2[j][2 * g + h2]);
            bf16x2 b;
            b[0] = (bf16)v[0];
            b[1] = (bf16)v[1];
            P[g][h2] = __builtin_bit_cast(unsigned, b);
          }
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          {   // 8 blocks per tile carry the 11 pieces of the next halo: 2, 1, 2, 1, 2, 1, 1, 1
            constexpr int NB_ = FM * FN * 2;
            const int blk = (i * FN + j) * 2 + gp;
            const int e0 = blk < 6 ? blk + (blk + 1) / 2 : blk + 3, ne = (blk < 6 && !(blk & 1)) ? 2 : 1;
            static_assert(NB_ == 8 && C::NHE == 11, "halo piece schedule");
            if (halo) {
              halo_piece(e0);
              if (ne == 2) halo_piece(e0 + 1);
            }
          }
          // lanes < 32 give their upper group to, and take the lower group from, lane + 32 (same pixel, other k-half)
          const auto a0 = __builtin_amdgcn_permlane32_swap(P[2 * gp][0], P[2 * gp + 1][0], false, false);
          const auto a1 = __builtin_amdgcn_permlane32_swap(P[2 * gp][1], P[2 * gp + 1][1], false, false);
          unsigned ww[4] = {(unsigned)a0[0], (unsigned)a1[0], (unsigned)a0[1], (unsigned)a1[1]};   // 8 consecutive channels
          if constexpr (RM > 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              // bf16 pairs handled as bits: low half = even channel, high half = odd channel
              const float w0 = __builtin_bit_cast(float, ww[d] << 16), w1 = __builtin_bit_cast(float, ww[d] & 0xffff0000u);
              const float g0 = __builtin_bit_cast(float, rg[j][gp][d] << 16), g1 = __builtin_bit_cast(float, rg[j][gp][d] & 0xffff0000u);
              const float a0f = __builtin_bit_cast(float, ra[j][gp][d] << 16), a1f = __builtin_bit_cast(float, ra[j][gp][d] & 0xffff0000u);
              bf16x2 wb;
              wb[0] = (bf16)(w0 + ((RM < 2 || a0f > 0.f) ? g0 : 0.f));
              wb[1] = (bf16)(w1 + ((RM < 2 || a1f > 0.f) ? g1 : 0.f));
              ww[d] = __builtin_bit_cast(unsigned, wb);
            }
          }
          const u32x4 w = {ww[0], ww[1], ww[2], ww[3]};
#ifdef PH_ABL_NOSTORE
          asm volatile("" ::"v"(w));
#else
          // (write-through `sc1` stores, tried so that the output lines would not occupy the XCD's L2 beside the halos:
          // the layer-1 launches ran 43 % slower on the same box - profiles/EXPERIMENTS.md)
          if (mine) *reinterpret_cast<u32x4*>(out + o + j * 32 + gp * 16) = w;
#endif
        }
      }
    }
  };
  const int rmode = p.res_g ? (p.res_a ? 2 : 1) : 0;
  auto epilogue_any = [&](const TileCtx& tc, const bool halo) {
    const bool full = (tc.r0 + TH <= p.OHt) && (tc.c0 + TW <= p.OWt);
    auto with_full = [&](auto fullc) {
      if (rmode == 0) epilogue(tc, fullc, std::integral_constant<int, 0>{}, halo);
      else if (rmode == 1) epilogue(tc, fullc, std::integral_constant<int, 1>{}, halo);
      else epilogue(tc, fullc, std::integral_constant<int, 2>{}, halo);
    };
    if (full) with_full(std::true_type{});
    else with_full(std::false_type{});
  };

  if (fuse_in) {   // (before any LDS-DMA is in flight)
    float* ss = reinterpret_cast<float*>(smem + C::SS_OFF);
    if (tid < 2 * BNT) ss[tid] = tid < BNT ? p.in_scale[tid] : p.in_shift[tid - BNT];
    __syncthreads();
  }
  // ---- prologue: all 9 taps of weights (8 waves x NWP pieces) and each group's first halo
  {
#pragma unroll
    for (int j = 0; j < C::NWP; ++j) {
      const int q = wave * C::NWP + j, tap = q >> 3;          // 8 pieces of 1 KiB per tap
      const int rp = (q & 7) * 4 + (lane >> 4), u = (lane & 15) ^ (rp & PH_SWZ_MASK);
      const int slab = __builtin_amdgcn_readlane(tap_tab, tap) >> 16;
      const unsigned char* wb = reinterpret_cast<const unsigned char*>(wbase + (size_t)slab * p.Cout * p.Cin);
      lds_dma16(wb + ((2 * rp + (u >> 3)) * p.Cin + (u & 7) * 8) * 2, lds0 + B_BASE + q * 1024);
    }
  }
  // tile contexts run one ahead of the store phase: each tile is decoded once
  TileCtx tc_cur{}, tc_next{};
  if (n_mine > 0) {
    tc_cur = decode(tile_id(grp));
    issue_halo(tc_cur);
  }
  if (n_mine > 1) tc_next = decode(tile_id(2 + grp));
  PH_WAIT_VMCNT(0);
  if (fuse_in && n_mine > 0) xform_halo();
  PH_BARRIER();
  PH_TRACE(1);

  bf16x8 fa[2][FM], fb[2][FN];
  int aaddr[FM], bxs[FN];
  auto tap_addr = [&](int toff, int slot) {
#pragma unroll
    for (int i = 0; i < FM; ++i) aaddr[i] = grp * C::A_BYTES + lds_off(prow[i] + toff, khalf);
#pragma unroll
    for (int j = 0; j < FN; ++j) bxs[j] = B_BASE + slot * C::TAPB + bx[j];
  };
#ifdef PH_TRACE_FINE   // stamps inside the store phase (each costs an s_waitcnt lgkmcnt(0), which also waits for LDS-DMA)
#define PH_CLKF() PH_CLK()
#else
#define PH_CLKF() 0ull
#endif
  // transposed product, accumulators in VGPRs; PH_MMT0 starts a tile (C = 0)
#ifdef PH_ABL_NOMFMA
#define PH_MMT(CB, I, J) asm volatile("" : "+v"(acc[I][J]) : "v"(fb[CB][J]), "v"(fa[CB][I]))
#define PH_MMT0(CB, I, J) asm volatile("" : "=v"(acc[I][J]) : "v"(fb[CB][J]), "v"(fa[CB][I]))
#else
#define PH_MMT(CB, I, J) \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[I][J]) : "v"(fb[CB][J]), "v"(fa[CB][I]))
#define PH_MMT0(CB, I, J) \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[I][J]) : "v"(fb[CB][J]), "v"(fa[CB][I]))
#endif
#define PH_KSTEP2T(MM, CB, NB, KS)                              \
  MM(CB, 0, 0); fa[NB][0] = PH_LDA(aaddr, 0, KS); PH_SB();      \
  MM(CB, 0, 1); fb[NB][0] = PH_LDA(bxs, 0, KS); PH_SB();        \
  MM(CB, 1, 0); fb[NB][1] = PH_LDA(bxs, 1, KS); PH_SB();        \
  MM(CB, 1, 1); fa[NB][1] = PH_LDA(aaddr, 1, KS); PH_SB()
#define PH_KSTEP2T_END(CB) \
  PH_MMT(CB, 0, 0); PH_MMT(CB, 0, 1); PH_MMT(CB, 1, 0); PH_MMT(CB, 1, 1); PH_SB()

  unsigned long long cyc_m = 0, cyc_e = 0, cyc_b = 0, cyc_e1 = 0, cyc_e2 = 0, cyc_e3 = 0;
  const unsigned long long ql0_ = PH_CLK();
  (void)ql0_;
  // Group 1 enters one barrier late: its matrix phases coincide with group 0's store phases.  Every wave executes
  // 2 * ceil(K / 2) + 1 barriers in total (the shorter list pads at the end).  The loop body is straight-line on
  // purpose: with the two phases in the arms of a branch the accumulators were merged in VGPRs at the loop header and
  // copied to and from the AGPRs (128 moves) in every half-phase.
  if (grp == 1) PH_BARRIER();
  for (int i = 0; i < n_mine; ++i) {
    const unsigned long long q0_ = PH_CLK();
    {
      // ---- matrix phase: 9 taps x 4 k-steps x 4 MFMAs from the group's halo buffer and the resident weights.
      // (The empty asm makes the per-lane bases opaque: otherwise the 9 taps' LDS addresses - loop invariants -
      // are all hoisted in front of the loop and spilled; they cost a few VALU in MFMA shadows here.)
#ifndef PH_L1_NOPREFETCH
      if (i + 1 < n_mine) prefetch_halo(tc_next);
#endif
#pragma unroll
      for (int ii = 0; ii < FM; ++ii) asm volatile("" : "+v"(prow[ii]));
#pragma unroll
      for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(bx[j]));
      tap_addr(0, 0);
#pragma unroll
      for (int ii = 0; ii < FM; ++ii) fa[0][ii] = PH_LDA(aaddr, ii, 0);
#pragma unroll
      for (int j = 0; j < FN; ++j) fb[0][j] = PH_LDA(bxs, j, 0);
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        if (t == 0) { PH_KSTEP2T(PH_MMT0, 0, 1, 1); } else { PH_KSTEP2T(PH_MMT, 0, 1, 1); }
        PH_KSTEP2T(PH_MMT, 1, 0, 2);
        PH_KSTEP2T(PH_MMT, 0, 1, 3);
        if (t + 1 < NTAPS) {
          tap_addr(((t + 1) / 3) * HPW + (t + 1) % 3, t + 1);
          PH_KSTEP2T(PH_MMT, 1, 0, 0);
        } else {
          PH_KSTEP2T_END(1);
        }
      }
    }
    const unsigned long long q1_ = PH_CLK();
    PH_BARRIER();
    const unsigned long long q2_ = PH_CLK();
    {
      // ---- store phase: the halo buffer is free (barrier) -> next halo in flight behind the epilogue
      const bool halo = i + 1 < n_mine;
      if (halo) halo_begin(tc_next);
      const unsigned long long e1_ = PH_CLKF();
      epilogue_any(tc_cur, halo);
      tc_cur = tc_next;
      if (i + 2 < n_mine) tc_next = decode(tile_id(2 * (i + 2) + grp));
      const unsigned long long e2_ = PH_CLKF();
      PH_WAIT_VMCNT(0);
      if (fuse_in && halo) xform_halo();
      const unsigned long long e3_ = PH_CLK();
      cyc_e += e3_ - q2_;
      cyc_e1 += e1_ - q2_; cyc_e2 += e2_ - e1_; cyc_e3 += e3_ - e2_;
    }
    const unsigned long long q3_ = PH_CLK();
    PH_BARRIER();
    cyc_m += q1_ - q0_;
    cyc_b += (q2_ - q1_) + (PH_CLK() - q3_);
  }
  for (int r = 2 * (((K + 1) >> 1) - n_mine) + (grp == 0 ? 1 : 0); r > 0; --r) PH_BARRIER();

  if (p.stats) {   // one partial row [2][Cout] per workgroup
    // per-lane channel sums -> sum over the 32 pixel lanes of each k-half (xor butterfly: fixed order), then the 8
    // waves' rows are combined in a fixed order
    float* red = reinterpret_cast<float*>(smem + C::LDS_MAIN);   // [8][2][BNT]
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int cmp = 0; cmp < 2; ++cmp) {
          float a1 = s1[j][q][cmp], a2 = s2[j][q][cmp];
#pragma unroll
          for (int d = 1; d < 32; d <<= 1) {
            a1 += __shfl_xor(a1, d, 64);
            a2 += __shfl_xor(a2, d, 64);
          }
          if ((lane & 31) == 0) {
            const int ch = j * 32 + 8 * (q >> 1) + 4 * khalf + 2 * (q & 1) + cmp;
            red[(wave * 2 + 0) * BNT + ch] = a1;
            red[(wave * 2 + 1) * BNT + ch] = a2;
          }
        }
    __syncthreads();
    if (tid < 2 * BNT) {
      const int which = tid / BNT, n = tid % BNT;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[(w * 2 + which) * BNT + n];
      p.stats[((size_t)blockIdx.x * 2 + which) * p.Cout + n] = v;
    }
  }
  PH_TRACE(5);
  PH_TRACE_ACC_T(6, cyc_m, 0); PH_TRACE_ACC_T(3, cyc_e, 0); PH_TRACE_ACC_T(8, cyc_b, 0);
#ifdef PH_TRACE_FINE
  PH_TRACE_ACC_T(4, cyc_e1, 256); PH_TRACE_ACC_T(9, cyc_e2, 256); PH_TRACE_ACC_T(7, cyc_e3, 256);   // group 1's store phase: issue / epilogue / wait
#else
  PH_TRACE_ACC_T(4, cyc_m, 256); PH_TRACE_ACC_T(9, cyc_e, 256); PH_TRACE_ACC_T(7, cyc_b, 256);   // group 1
#endif
  (void)cyc_e1; (void)cyc_e2; (void)cyc_e3;
  PH_TRACE_ACC(10, PH_CLK() - ql0_); PH_TRACE_ACC(11, (unsigned long long)K);
  (void)cyc_m; (void)cyc_e; (void)cyc_b;
}

int launch_l1(const PhTapConv& p, hipStream_t st) {
  using C = L1Cfg;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(tapconv2_l1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess)
      return PH_ELAUNCH;
    attr_done = true;
  }
  const int total = cdiv(p.OHt, C::TH) * cdiv(p.OWt, C::TW) * p.B;
  const int resident = ph_num_cus();
  dim3 grid(total < resident ? total : resident);
  void* tok = nullptr;
  if (ph_prof_on())
    ph_prof_begin2(p.in_scale ? PH_CLS_TAPCONV2_RES_FUSEDIN : PH_CLS_TAPCONV2_RES, 2.0 * p.B * p.OHt * p.OWt * (double)p.Cout * p.ntaps * p.Cin, ph_tapconv_bytes(p, 1, 2), st, &tok);
  hipLaunchKernelGGL(tapconv2_l1_kernel, grid, dim3(C::NTH), C::LDS_BYTES, st, p);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

template <int WM, int WN, int FM, bool RES, bool MASKED = false>
int launch2(const PhTapConv& p, hipStream_t st) {
  using C = Tap2Cfg<WM, WN, FM, RES>;
  auto kern = tapconv2_kernel<WM, WN, FM, RES, MASKED>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess)
      return PH_ELAUNCH;
    attr_done = true;
  }
  const int total = cdiv(p.OHt, C::TH) * cdiv(p.OWt, C::TW) * (p.Cout / C::BNT) * p.B;
  const int resident = ph_num_cus();   // one workgroup per CU (LDS)
  dim3 grid(total < resident ? total : resident);
  void* tok = nullptr;
  if (ph_prof_on()) {   // algorithmic FLOPs: 2 * positions * Cout * ntaps * Cin (masked: the LIVE taps of all groups)
    double taps = p.ntaps;
    if (MASKED) {
      taps = 9;      // 1 + 2 + 2 + 4 live taps over the four planes, each on Cin channels
    }
    ph_prof_begin2(MASKED ? PH_CLS_TAPCONV2_MASKED : (RES ? PH_CLS_TAPCONV2_RES : (p.in_scale ? PH_CLS_TAPCONV2_FUSEDIN : PH_CLS_TAPCONV2)),
                   2.0 * p.B * p.OHt * p.OWt * (double)p.Cout * taps * p.Cin, ph_tapconv_bytes(p, 1, 2), st, &tok);
  }
  hipLaunchKernelGGL(kern, grid, dim3(C::NTH), C::LDS_BYTES, st, p);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

}  // namespace

// A/B and test switch between the second- and third-generation dense kernels: PH_TAP3=0 in the environment, or
// ph_debug_set_tap3() at run time (not part of the public C-ABI).  set < 0: query.
int ph_tap3_switch(int set) {
  static int on = [] { const char* e = getenv("PH_TAP3"); return (e && e[0] == '0') ? 0 : 1; }();
  if (set >= 0) on = set ? 1 : 0;
  return on;
}
extern "C" int ph_debug_set_tap3(int on) { return ph_tap3_switch(on ? 1 : 0); }
// ... and between tapconv2_l1_kernel and the fourth-generation layer-1 kernel (conv_tap4.hip): PH_TAP4=0 / ph_debug_set_tap4()
int ph_tap4_switch(int set) {
  static int on = [] { const char* e = getenv("PH_TAP4"); return (e && e[0] == '0') ? 0 : 1; }();
  if (set >= 0) on = set ? 1 : 0;
  return on;
}
extern "C" int ph_debug_set_tap4(int on) { return ph_tap4_switch(on ? 1 : 0); }

// ---- stride-2 3x3 convolutions as MASKED stride-1 tap grids (PhTapConv::m_*).  Both fill the tap-grid part of a
// descriptor whose tensors / batch / channel fields the caller has set (forward: in = x [B][IH][IW][Cin], Cin / Cout of
// the convolution; dgrad: in = dy [B][OH][OW][Cout_fwd], t->Cin = Cout_fwd, t->Cout = Cin_fwd, OH / OW of dx) and
// return false when the shape is not eligible (odd sizes, channel counts the <2,2,4> configuration does not tile,
// parity mode): the caller then runs the first-generation kernels.
//
// Forward: output (r, c) reads input (2r + kh - 1, 2c + kw - 1).  On the pixel-parity plane (a, b) = pixels (2i + a,
// 2j + b) - a strided view of x - that is plane pixel (r + gy - 1, c + gx - 1) with grid tap gy = 1 for kh = 1 (a = 0)
// and gy = 0 / 1 for kh = 0 / 2 (a = 1): plane (a, b) has (1 + a)(1 + b) live taps, 9 in all, every input pixel is
// staged once, and the K loop walks 4 * Cin / 64 slices.
bool ph_tapconv2_setup_s2_fwd(PhTapConv* t, int Cin, int Cout, int IH, int IW, int prec) {
  if (prec != PH_PREC_BF16 || Cout % 128 || Cout > 512 || Cin % 64 || Cin > 512 || ((Cin >> 6) & ((Cin >> 6) - 1)) ||
      (IH & 1) || (IW & 1))
    return false;
  t->in_pix_stride = 2L * Cin; t->in_row_stride = 2L * IW * Cin; t->in_img_stride = (long)IH * IW * Cin;
  t->IH = IH / 2; t->IW = IW / 2;
  t->iy0 = -1; t->ix0 = -1; t->ntaps = 9;
  for (int k = 0; k < 9; ++k) { t->dy[k] = k / 3; t->dx[k] = k % 3; t->wtap[k] = 0; }
  t->m_groups = 4;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int g = a * 2 + b;
      t->m_in_off[g] = ((long)a * IW + b) * Cin;
      t->m_mask[g] = 0;
      for (int gy = 0; gy < 3; ++gy)
        for (int gx = 0; gx < 3; ++gx) {
          const int kh = a ? (gy == 0 ? 0 : (gy == 1 ? 2 : -1)) : (gy == 1 ? 1 : -1);
          const int kw = b ? (gx == 0 ? 0 : (gx == 1 ? 2 : -1)) : (gx == 1 ? 1 : -1);
          const bool live = kh >= 0 && kw >= 0;
          t->m_slab[g][gy * 3 + gx] = live ? kh * 3 + kw : 0;
          if (live) t->m_mask[g] |= 1 << (gy * 3 + gx);
        }
    }
  // the kernel hard-codes these live-tap sets (its four slice loops): keep them in step
  if (t->m_mask[0] != 0x010 || t->m_mask[1] != 0x018 || t->m_mask[2] != 0x012 || t->m_mask[3] != 0x01B) return false;
  return true;
}

// 0: not eligible (the first-generation kernel runs); otherwise the tile height of the configuration chosen
int ph_tapconv2_tile_h(const PhTapConv* p, int S, int prec) {
  if (S != 1 || prec != PH_PREC_BF16 || p->ntaps != 9 || p->Cout > 512) return 0;
  if (p->m_groups && (p->Cout % 128 || p->m_groups != 4 || ((p->Cin >> 6) & ((p->Cin >> 6) - 1)) || p->Cin > 512)) return 0;
  if (p->Cout % 128 && !(p->Cout == 64 && p->Cin == 64)) return 0;
  for (int k = 0; k < 9; ++k)   // the kernel hard-codes the 3x3 tap geometry (only the weight slab order is a table)
    if (p->dy[k] != k / 3 || p->dx[k] != k % 3) return 0;
  return 16;
}

// number of BatchNorm partial rows a launch writes = its (persistent) workgroups
int ph_tapconv2_stat_parts(const PhTapConv* p) {
  const int bnt = (p->Cout % 128 == 0) ? 128 : 64;
  const int total = cdiv(p->OHt, 16) * cdiv(p->OWt, 16) * (p->Cout / bnt) * p->B;
  const int resident = ph_num_cus();
  return total < resident ? total : resident;
}

int ph_tapconv2_launch(const PhTapConv* p, hipStream_t st) {
  if (p->in_scale && (!p->in_shift || p->Cin > 512)) return PH_EINVAL;
  if (p->m_groups) return (p->Cout % 128 == 0 && !p->in_scale) ? launch2<2, 2, 4, false, true>(*p, st) : PH_EINVAL;
  if (p->Cout % 128 == 0) {
    if (ph_tap3_switch(-1) && ph_tapconv3_eligible(p)) {
      // every form but the in-LDS input BatchNorm, Cin = Cout: conv_tap7.hip, same outputs
      if (ph_tap7_switch(-1) && ph_tapconv7_eligible(p)) return ph_tapconv7_launch(p, st);
      return ph_tapconv3_launch(p, st);
    }
    if (p->bst_y) return PH_EINVAL;      // (the fused BatchNorm-backward sums exist in conv_tap3.hip / conv_tap4.hip only)
    return launch2<2, 2, 4, false>(*p, st);
  }
  if (ph_tap4_switch(-1) && ph_tapconv4_eligible(p)) return ph_tapconv4_launch(p, st);
  if (p->bst_y) return PH_EINVAL;      // (the fused BatchNorm-backward sums exist in conv_tap4.hip only)
#ifdef PH_L1_ONE_GROUP   // A/B build: the one-wave-per-SIMD resident-weights configuration
  return launch2<4, 1, 2, true>(*p, st);
#else
  return launch_l1(*p, st);   // Cin = Cout = 64: resident weights, two wave groups
#endif
}

#ifdef PH_TAP_TRACE
extern "C" int ph_debug_tap2_trace(unsigned long long* host_out, int nwg) {
  if (nwg > PH_TRACE_WGS) nwg = PH_TRACE_WGS;
  if (hipDeviceSynchronize() != hipSuccess) return PH_ELAUNCH;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ph_tap_trace), (size_t)nwg * 12 * sizeof(unsigned long long)) == hipSuccess
             ? PH_OK : PH_ELAUNCH;
}
#endif
