// Stem convolution 7x7 / stride 2 / pad 3, Cin = 3 (reference resnets.py:146-147) on MFMA.
//
// The input is repacked NHWC4 (channel 3 = 0) so that one kernel row kh of one output pixel is
// 8 pixels x 4 channels = 32 contiguous bf16 = exactly two 16-B MFMA A-fragments: K = 7 rows x 32
// (147 real taps padded to 224).  The halo tile (21 x 38 input pixels for an 8x16 output tile) and ALL
// weights (28 KB) live in LDS.  Forward: rows = pixels, cols = 64 cout (BN partial sums in the epilogue).
// wgrad: M = cout, N = (kw,ch) 32 per kh, K = pixels, both operands through ds_read_b64_tr_b16.
// There is no dgrad (the image needs no gradient on the hot path; dx is produced only on request by a
// separate small kernel for the parity tests).
#include <cstdlib>
#include "ph_common.h"
#include <type_traits>
#include "ph_kernels.h"

#ifndef PH_STEM_HP_ABL      // timing ablations of the half-pair stem (make trace TRACE_TAG=_x EXTRA=-DPH_STEM_HP_ABL=n): 1 no output
#define PH_STEM_HP_ABL 0    // stores, 2 no MFMAs
#endif

namespace {

__device__ const u32x4 stem_zero8[1] = {};   // source of out-of-image halo pixels (perf-mode register prefetch)

constexpr int TH = 8, TW = 16, HPH = (TH - 1) * 2 + 7, HPW = (TW - 1) * 2 + 8;   // 21 x 38
constexpr int HP = HPH * HPW;
constexpr int XB = HP * 8;            // bytes of one bf16 halo plane
constexpr int WB = 7 * 64 * 64;       // bytes of one weight plane [7][64][32] bf16
constexpr int STEM_TPW = 8;           // output tiles per forward workgroup (weights staged once per workgroup)

__device__ __forceinline__ int wsw(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }

// scalar fp32 add / fused multiply-add the SLP vectoriser cannot pair: hipcc -O3 turned the BatchNorm sums of the epilogue
// into v_pk_add_f32 / v_pk_mul_f32, which cost ~16 cycles each beside an MFMA stream (the staging phase: 2200 cycles for
// ~200 instructions, interval tracer) - MI355X_MICROARCH.md, "price of one filler beside MFMAs"
__device__ __forceinline__ float fadd_s(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float ffma_s(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}


// (T = hp16: x4 = the two fp16 planes [2][B IH IW][4] of ph_pack_input_launch, staged as they lie; `nb` = batch size)
template <typename T>
__device__ __forceinline__ void stage_halo(const T* x4, int b, int IH, int IW, int iy_base, int ix_base,
                                           unsigned char* hi, int tid, int nb = 0) {
  for (int i = tid; i < HP; i += 256) {
    const int hr = i / HPW, hc = i - hr * HPW;
    const int iy = iy_base + hr, ix = ix_base + hc;
    const bool ok = iy >= 0 && iy < IH && ix >= 0 && ix < IW;
    const size_t g = (((size_t)b * IH + iy) * IW + ix) * 4;
    if constexpr (is_hp<T>::value) {
      const f16* xp = reinterpret_cast<const f16*>(x4);
      u32x2 vh = {0u, 0u}, vl = {0u, 0u};
      if (ok) {
        vh = *reinterpret_cast<const u32x2*>(xp + g);
        vl = *reinterpret_cast<const u32x2*>(xp + (size_t)nb * IH * IW * 4 + g);
      }
      *reinterpret_cast<u32x2*>(hi + i * 8) = vh;
      *reinterpret_cast<u32x2*>(hi + XB + i * 8) = vl;
    } else if constexpr (!is_f32<T>::value) {
      u32x2 v = {0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x2*>(x4 + g);
      *reinterpret_cast<u32x2*>(hi + i * 8) = v;
    } else {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(x4 + g);
      bf16x4 p0, p1, p2;
#pragma unroll
      for (int q = 0; q < 4; ++q) { bf16 a, b2, c2; split3_bf16(v[q], a, b2, c2); p0[q] = a; p1[q] = b2; p2[q] = c2; }
      *reinterpret_cast<bf16x4*>(hi + i * 8) = p0;
      *reinterpret_cast<bf16x4*>(hi + XB + i * 8) = p1;
      *reinterpret_cast<bf16x4*>(hi + 2 * XB + i * 8) = p2;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(PhStem p) {
  // SPLIT: the plane-staged path (three bf16 split planes of fp32 operands; T = hp16, PH_PREC_FP16X3: the two fp16 planes
  // of the half-pair input against the three weight planes (hi 2^11, lo, hi), fp32 output = accumulators * 2^-11)
  constexpr bool HPM = is_hp<T>::value;
  constexpr bool SPLIT = is_f32<T>::value;      // synchronous plane staging (fp32 operands split in the kernel)
  constexpr int NP = (SPLIT || HPM) ? PH_NPLANES : 1;      // weight planes
  constexpr int NPA = SPLIT ? PH_NPLANES : (HPM ? 2 : 1);   // activation planes of ONE halo buffer
  typedef typename std::conditional<HPM, float, T>::type TO;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // perf mode and half-pair mode: two halo buffers of NPA planes (tile t+1 is written while nobody reads it: one barrier per
  // tile; the half-pair planes arrive as they lie in HBM, nothing is converted); split-plane modes: one buffer of 3 planes
  constexpr int XBUFS = SPLIT ? NP : 2 * NPA;
  unsigned char* ldsX = smem;               // XBUFS x XB
  unsigned char* ldsW = smem + XB * XBUFS;  // NP planes of WB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_w = (p.OW + TW - 1) / TW, tiles_h = (p.OH + TH - 1) / TH;
  const int tiles_img = tiles_w * tiles_h, ntiles = tiles_img * p.B;
  const T* x4 = reinterpret_cast<const T*>(p.x4);

  // the 28 KB (x planes) weight image is staged ONCE per workgroup and reused for STEM_TPW consecutive tiles
  if constexpr (!HPM) {
  for (int i = tid; i < 7 * 64 * 4; i += 256) {   // 16-B chunks of the weight plane(s)
    const int ch = i & 3, srow = i >> 2;          // source row = kh*64 + cout
    // perf mode (round 6): LDS row kh*64 + j*32 + l holds cout 2 l + j, so that lane l of the MFMA owns the channel PAIR
    // (2 l, 2 l + 1) in its two accumulator sets: a pixel's [even, odd] bf16 word is one v_cvt_pk of the lane's own values - the
    // lane-pair DPP exchange + byte permute per word (32 of a tile's ~590 instructions) are gone; values, statistics and their
    // summation order are unchanged (an MFMA column's result does not depend on which column it is)
    const int co = srow & 63;
    const int row = (!SPLIT) ? ((srow & ~63) | ((co & 1) << 5) | (co >> 1)) : srow;
    const int off = row * 64 + (wsw(row, ch) << 4);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
      *reinterpret_cast<u32x4*>(ldsW + pl * WB + off) =
          reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16*>(p.w) + (size_t)pl * p.wplane)[i];
  }
  }
  const int m = wave * 32 + (lane & 31), khalf = lane >> 5;
  // Half-pair mode (round 6): ALL weight fragments of the lane - 14 (kh, s) steps x 3 planes x 2 cout halves = 84 fragments of
  // 16 bytes = 336 registers - live in REGISTERS for the whole workgroup (one wave per SIMD: 512 registers; the bf16 pooled
  // kernel does the same with one plane).  With the three weight planes in LDS a (kh, s) step read 6 weight + 2 input fragments
  // for its 6 MFMAs - 170 B / clk per CU against the array's 256 - and the kernel ran at a quarter of the fp16 MFMA rate.
  // Planes 0 / 1 (w hi 2^11, w lo) are pinned to the accumulation registers (an MFMA takes its B operand from either file),
  // plane 2 and everything else stays in the vector registers: 224 + 32 (accumulators) AGPRs, ~190 VGPRs.
  bf16x8 wr[HPM ? 14 : 1][HPM ? 3 : 1][2];
  if constexpr (HPM) {
#pragma unroll
    for (int st = 0; st < 14; ++st)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = (st >> 1) * 64 + j * 32 + (lane & 31), chunk = (st & 1) * 2 + khalf;
          wr[st][pl][j] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.w) + (size_t)pl * p.wplane + row * 32 + chunk * 8);
        }
#pragma unroll
    for (int st = 0; st < 14; ++st)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("" : "+a"(wr[st][0][j]));
        asm volatile("" : "+a"(wr[st][1][j]));
        asm volatile("" : "+v"(wr[st][2][j]));
      }
  }
  const int pbase = ((m >> 4) * 2) * HPW + (m & 15) * 2;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  // blocks of STEM_TPW tiles: one per workgroup, or (PhStem::vblocks, half-pair mode) several per workgroup - the 336 weight
  // registers are loaded once per workgroup, every block still leaves its own statistics row
  const int nvb = p.vblocks > 0 ? p.vblocks : (int)gridDim.x;
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
  if (vb != (int)blockIdx.x) { __syncthreads(); s1[0] = s1[1] = s2[0] = s2[1] = 0.f; }      // (the row reduction below reads the halo buffers' LDS)
  const int t_begin = vb * STEM_TPW, t_end = min(ntiles, t_begin + STEM_TPW);
  // perf mode: the halo of tile t+1 is loaded into registers before the MFMAs of tile t (unconditional loads - an
  // out-of-image pixel reads a zero page - and, for interior tiles, unconditional stores in the epilogue, so the
  // compiler can count the s_waitcnt instead of draining the queue) and written to the other LDS buffer after them.
  constexpr int HCH = (HP + 255) / 256;
  u32x2 hreg[SPLIT ? 1 : HCH], hregl[HPM ? HCH : 1];
  // (round 6) per-thread constants of the halo prefetch: element e of the thread is halo pixel (hrow, hcol), `hoff` elements
  // past the tile's first halo pixel; a tile only contributes a wave-uniform base and the two range checks.  Recomputing row /
  // column / 64-bit address per element and tile was ~180 of a tile's ~590 instructions (device assembly, bf16 kernel).
  int hrow_[SPLIT ? 1 : HCH], hcol_[SPLIT ? 1 : HCH];
  long hoff_[SPLIT ? 1 : HCH];
  if constexpr (!SPLIT) {
#pragma unroll
    for (int e = 0; e < HCH; ++e) {
      const int i = tid + e * 256;
      const int hr = i / HPW, hc = i - hr * HPW;
      hrow_[e] = i < HP ? hr : (1 << 28);      // (an element past the halo: a row that is never inside)
      hcol_[e] = hc;
      hoff_[e] = ((long)hr * p.IW + hc) * 4;
    }
  }
  auto load_halo_regs = [&](int t) {
    if constexpr (!SPLIT) {
      const int b = t / tiles_img, tile = t - b * tiles_img;
      const int iy_base = (tile / tiles_w) * TH * 2 - 3, ix_base = (tile % tiles_w) * TW * 2 - 3;
      // (T = hp16: two fp16 planes [2][B IH IW][4]; 4 fp16 = 8 bytes per pixel and plane, like the bf16 image)
      const f16* xp = reinterpret_cast<const f16*>(p.x4);
      const f16* base = xp + (((long)b * p.IH + iy_base) * p.IW + ix_base) * 4;
      const long lo_plane = (long)p.B * p.IH * p.IW * 4;
#pragma unroll
      for (int e = 0; e < HCH; ++e) {
        const bool ok = (unsigned)(iy_base + hrow_[e]) < (unsigned)p.IH && (unsigned)(ix_base + hcol_[e]) < (unsigned)p.IW;
        const f16* src = ok ? base + hoff_[e] : reinterpret_cast<const f16*>(stem_zero8);
        hreg[e] = ld_global<u32x2>(src);
        if constexpr (HPM) hregl[e] = ld_global<u32x2>(ok ? src + lo_plane : src);
      }
    }
  };
  auto store_halo_regs = [&](unsigned char* dst) {
    if constexpr (!SPLIT) {
#pragma unroll
      for (int e = 0; e < HCH; ++e) {
        const int i = tid + e * 256;
        if (i < HP) {
          *reinterpret_cast<u32x2*>(dst + i * 8) = hreg[e];
          if constexpr (HPM) *reinterpret_cast<u32x2*>(dst + XB + i * 8) = hregl[e];
        }
      }
    }
  };
  if constexpr (!SPLIT) {
    if (t_begin < t_end) {
      load_halo_regs(t_begin);
      store_halo_regs(ldsX);
    }
    __syncthreads();   // (also publishes the weight image)
  }
  for (int tt = t_begin; tt < t_end; ++tt) {
    const int b = tt / tiles_img, tile = tt - b * tiles_img;
    const int r0 = (tile / tiles_w) * TH, c0 = (tile % tiles_w) * TW;
    const unsigned char* ldsXc = ldsX;
    if constexpr (SPLIT) {
      __syncthreads();   // previous tile's fragment reads of ldsX are done (and, first time, nothing)
      stage_halo<T>(x4, b, p.IH, p.IW, r0 * 2 - 3, c0 * 2 - 3, ldsX, tid, p.B);
      __syncthreads();
    } else {
      ldsXc = ldsX + ((tt - t_begin) & 1) * NPA * XB;
      if (tt + 1 < t_end) load_halo_regs(tt + 1);
    }

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    if constexpr (HPM) {
      // half-pair mode runs ONE workgroup per CU (143 KB of LDS: two halo buffers of two planes, three weight planes, the fp32
      // output image), so nothing hides a fragment read issued in front of its MFMA - the compiler-scheduled loop below waited
      // for lgkmcnt(0) 41 times per tile (25 % of the fp16 MFMA rate).  Here the 8 fragments of step st + 1 (of the 14 (kh, s)
      // steps) are read before the 6 MFMAs of step st; MFMAs as asm behind sched_barriers, accumulators pinned to AGPRs.
      asm volatile("" : "+a"(acc[0]));
      asm volatile("" : "+a"(acc[1]));
      bf16x8 fa[2][2];
      auto ldstep = [&](const int st, const int buf) {
        const int kh = st >> 1, s_ = st & 1;
        const int aoff = (pbase + kh * HPW + 4 * s_ + 2 * khalf) * 8;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) fa[buf][pl] = *reinterpret_cast<const bf16x8*>(ldsXc + pl * XB + aoff);
      };
#if PH_STEM_HP_ABL & 2      // timing ablation: no MFMAs (the operands stay live)
#define PH_STEM_MMA(J, A, B) asm volatile("" : "+a"(acc[J]) : "v"(A), "v"(B)); __builtin_amdgcn_sched_barrier(0)
#define PH_STEM_MMA_A(J, A, B) asm volatile("" : "+a"(acc[J]) : "v"(A), "a"(B)); __builtin_amdgcn_sched_barrier(0)
#else
#define PH_STEM_MMA(J, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[J]) : "v"(A), "v"(B)); __builtin_amdgcn_sched_barrier(0)
#define PH_STEM_MMA_A(J, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[J]) : "v"(A), "a"(B)); __builtin_amdgcn_sched_barrier(0)
#endif
      ldstep(0, 0);
#pragma unroll
      for (int st = 0; st < 14; ++st) {
        const int cur = st & 1;
        if (st + 1 < 14) ldstep(st + 1, cur ^ 1);
        // (the product order of the loop below: (x lo, w hi), (x hi, w lo), (x hi, w hi 2^11))
        PH_STEM_MMA(0, fa[cur][1], wr[st][2][0]); PH_STEM_MMA(1, fa[cur][1], wr[st][2][1]);
        PH_STEM_MMA_A(0, fa[cur][0], wr[st][1][0]); PH_STEM_MMA_A(1, fa[cur][0], wr[st][1][1]);
        PH_STEM_MMA_A(0, fa[cur][0], wr[st][0][0]); PH_STEM_MMA_A(1, fa[cur][0], wr[st][0][1]);
      }
#undef PH_STEM_MMA_A
#undef PH_STEM_MMA
    } else {
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int chunk = s * 2 + khalf;
        const int aoff = (pbase + kh * HPW + 4 * s + 2 * khalf) * 8;
        bf16x8 a[NP], bq[NP][2];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          if (!HPM || pl < 2) a[pl] = *reinterpret_cast<const bf16x8*>(ldsXc + pl * XB + aoff);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = kh * 64 + j * 32 + (lane & 31);
            bq[pl][j] = *reinterpret_cast<const bf16x8*>(ldsW + pl * WB + row * 64 + (wsw(row, chunk) << 4));
          }
        }
        if constexpr (HPM) {
#define PH_MMH(PI, PJ)                                                                              \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[PI]), __builtin_bit_cast(f16x8, bq[PJ][j]), acc[j], 0, 0, 0);
          PH_MMH(1, 2) PH_MMH(0, 1) PH_MMH(0, 0)
#undef PH_MMH
        } else if constexpr (SPLIT) {
#define PH_MM(PI, PJ)                                                                               \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PI], bq[PJ][j], acc[j], 0, 0, 0);
          if (p.prod6) { PH_SPLIT_PAIRS_LO(PH_MM) }
          PH_SPLIT_PAIRS_HI(PH_MM)
#undef PH_MM
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[0][j], acc[j], 0, 0, 0);
        }
      }
    }
    }
    // ---- per-tile epilogue: mask, accumulate statistics in registers, store.  bf16 outputs leave as 4-byte words:
    // lanes l, l^1 hold neighbouring channels of the same pixels, so they swap one value of each column pair by DPP and
    // each stores one [even channel, odd channel] word (half the store instructions of 2-byte stores, which bound the
    // kernel: 537 MB of output left the CU in 128-B wave-instructions)
    TO* out = reinterpret_cast<TO*>(p.out) + (size_t)b * p.OH * p.OW * 64;
    if constexpr (!SPLIT && !HPM) {
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
      auto store_tile = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
        for (int q2 = 0; q2 < 8; ++q2) {
          const int mm = wave * 32 + ((2 * q2) & 3) + 8 * ((2 * q2) >> 2) + 4 * khalf;   // pixel of column 2*q2 (even)
          const int r = r0 + (mm >> 4), c = c0 + (mm & 15);
          const bool v0ok = FULL || (r < p.OH && c < p.OW), v1ok = FULL || (r < p.OH && c + 1 < p.OW);
          // the lane's channel pair (2 l, 2 l + 1) of pixels (r, c) and (r, c + 1): 32 lanes write one pixel's 128 bytes
          bf16* dst = reinterpret_cast<bf16*>(out) + ((size_t)r * p.OW + c) * 64 + 2 * (lane & 31);
          float v[2][2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            v[j][0] = v0ok ? acc[j][2 * q2] : 0.f; v[j][1] = v1ok ? acc[j][2 * q2 + 1] : 0.f;
            // (scalar ops in asm: left to the compiler the sums of the two accumulator sets became v_pk_add_f32 / v_pk_fma_f32,
            // which issue far slower than two plain ops on a SIMD that other waves keep busy with MFMAs; same values, same order)
            s1[j] = fadd_s(s1[j], fadd_s(v[j][0], v[j][1]));
            s2[j] = ffma_s(v[j][1], v[j][1], ffma_s(v[j][0], v[j][0], s2[j]));
          }
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            bf16x2 w;
            w[0] = (bf16)v[0][px];
            w[1] = (bf16)v[1][px];
            if (FULL || (px ? v1ok : v0ok)) *reinterpret_cast<unsigned*>(dst + px * 64) = __builtin_bit_cast(unsigned, w);
          }
        }
      };
      // the next tile's halo (requested before the MFMAs) goes to the buffer nobody reads BEFORE this tile's output
      // stores are issued: the wait in front of the LDS write then covers the loads only, not 16 stores in flight
      if (tt + 1 < t_end) store_halo_regs(ldsX + (((tt - t_begin) & 1) ^ 1) * NPA * XB);
      if (r0 + TH <= p.OH && c0 + TW <= p.OW) store_tile(std::true_type{});
      else store_tile(std::false_type{});
      // LDS-only barrier (round 6): it publishes the next tile's halo; __syncthreads() would also wait for this tile's 16 KB of
      // output stores (vmcnt(0)) in every wave before the next tile's MFMAs may start - the stores drain behind them instead
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      // fp32 outputs, one 4-byte store per value (a wave-instruction writes two 128-byte runs).  Half-pair mode: the next tile's
      // halo planes (requested before the MFMAs) go to the other buffer first, one barrier per tile as in perf mode
      if constexpr (HPM) { if (tt + 1 < t_end) store_halo_regs(ldsX + (((tt - t_begin) & 1) ^ 1) * NPA * XB); }
      // half-pair mode: the fp32 tile goes through an LDS image [128 pixels][64 channels] behind the operand buffers and leaves as
      // 16-byte chunks (one 4-byte store per value: 32 store instructions per wave and tile for 1.07 GB of output)
      float* cimg = reinterpret_cast<float*>(smem + XB * XBUFS + (HPM ? 0 : WB * NP));      // (half-pair mode keeps no weight image)
      const bool full_tile = r0 + TH <= p.OH && c0 + TW <= p.OW;
      auto stats_and_image = [&](auto fullc) {      // (interior tiles: no per-value range selects - half-pair mode, round 6)
        constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int mm = wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * khalf;
        const int r = r0 + (mm >> 4), c = c0 + (mm & 15);
        const bool valid = FULL || (r < p.OH && c < p.OW);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float v = valid ? (HPM ? acc[j][q] * PH_HP_LO_INV : acc[j][q]) : 0.f;
          s1[j] += v; s2[j] = __builtin_fmaf(v, v, s2[j]);
          if constexpr (HPM) cimg[mm * 64 + j * 32 + (lane & 31)] = v;
          else if (valid) stf(out + ((size_t)r * p.OW + c) * 64 + j * 32 + (lane & 31), v);
        }
      }
      };
      if (HPM && full_tile) stats_and_image(std::true_type{});
      else stats_and_image(std::false_type{});
      if constexpr (HPM) {
        // (LDS-only barriers: the 32 KB of output stores of a tile drain behind the next tile's MFMAs; the image is free again
        // once every wave has READ its chunks - lgkmcnt - not once the stores have completed)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (full_tile) {
          // all eight 16-byte chunks of the thread are read before the first store (as a loop each iteration waited for its own read)
          f32x4 cv[TH * TW * 16 / 256];
#pragma unroll
          for (int e = 0; e < TH * TW * 16 / 256; ++e) {
            const int id = tid + e * 256;
            cv[e] = *reinterpret_cast<const f32x4*>(cimg + (id >> 4) * 64 + (id & 15) * 4);
          }
#pragma unroll
          for (int e = 0; e < TH * TW * 16 / 256; ++e) {
            const int id = tid + e * 256, mm = id >> 4, ch = id & 15;
#if PH_STEM_HP_ABL & 1
            asm volatile("" ::"v"(cv[e]));
#else
            *reinterpret_cast<f32x4*>(out + ((size_t)(r0 + (mm >> 4)) * p.OW + c0 + (mm & 15)) * 64 + ch * 4) = cv[e];
#endif
          }
        } else
        for (int id = tid; id < TH * TW * 16; id += 256) {
          const int mm = id >> 4, ch = id & 15;
          const int r = r0 + (mm >> 4), c = c0 + (mm & 15);
#if PH_STEM_HP_ABL & 1      // timing ablation: no output stores
          if (r < p.OH && c < p.OW) { f32x4 v_ = *reinterpret_cast<const f32x4*>(cimg + mm * 64 + ch * 4); asm volatile("" ::"v"(v_)); }
#else
          if (r < p.OH && c < p.OW)
            *reinterpret_cast<f32x4*>(out + ((size_t)r * p.OW + c) * 64 + ch * 4) = *reinterpret_cast<const f32x4*>(cimg + mm * 64 + ch * 4);
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
  }
  if (p.stats) {   // one partial row per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][2][64]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float a1 = s1[j] + __shfl_xor(s1[j], 32, 64);
      const float a2 = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (khalf == 0) {
        const int chn = (!SPLIT && !HPM) ? 2 * (lane & 31) + j : j * 32 + (lane & 31);      // (perf mode: channel pairs per lane)
        red[(wave * 2 + 0) * 64 + chn] = a1;
        red[(wave * 2 + 1) * 64 + chn] = a2;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, n = tid & 63;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[(w * 2 + which) * 64 + n];
      p.stats[((size_t)vb * 2 + which) * 64 + n] = v;
    }
  }
  }      // blocks of this workgroup
}


// ---------------------------------------------------------------------------------------------------------------------
// conv 7x7/2 + max-pool 3x3/2 of the raw output in one kernel (forward-only networks, perf mode; ph_kernels.h PhStemPool).
// A workgroup owns a STRIP: image b, 14 conv columns (7 pooled columns) - its 16-column tiles start one column to the left
// of the owned ones, so every pooling window's three columns lie inside the tile (2 of 16 columns are computed by two
// strips: 12.5 % more MFMA work on a kernel that was bound by writing 537 MB) - and walks down the image tile by tile;
// the last conv row of a tile stays in LDS for the first pooling row of the next tile, so no row is recomputed (a strip
// cut into `nsplit` pieces recomputes one tile per cut for that row).  Per tile: the same MFMA stream as stem_fwd_kernel,
// then the bf16-rounded outputs go to an LDS staging tile [9 rows][16][64] instead of HBM, one barrier, and 224 threads
// pool 4 x 7 x 64 outputs from it (16-byte LDS reads, sign-aware max) while the next tile's halo loads are in flight.
// Staging rows 1..7 are double-buffered and the last row triple-buffered: one barrier per tile protects everything.
// The staging tile holds the bf16 bit patterns themselves and the pooling loop compares them with v_pk_max_f16: bf16 and
// fp16 are both sign-magnitude formats, so for finite values the fp16 ordering of two bit patterns IS the bf16 ordering
// (patterns that fp16 reads as inf / NaN are bf16 magnitudes >= 2^121, fp16 denormals - preserved by the default float
// mode - bf16 magnitudes < 2^-119).  Channels with gamma < 0 are staged with their sign bit flipped, which turns the
// minimum they need into a maximum; the flip is undone on the pooled value.  Out-of-image pixels are staged as 0xFC00,
// fp16's -inf, the identity of the maximum.  One v_pk_max_f16 per two channels and tap, no conversions: the first version
// (fp32 compares, a sign multiply per tap, per-element masks everywhere) issued ~1000 vector instructions per tile and
// wave against 28 MFMAs and was bound by exactly that (ablation builds: every phase cost its full time, nothing overlapped).
constexpr int PCOLS = 14;
#ifndef PH_STEM_ABL      // ablation builds (`make trace TRACE_TAG=_sN EXTRA=-DPH_STEM_ABL=N`, tests/bench_stem_pool_gpu.py):
#define PH_STEM_ABL 0    // 1 no halo loads after the first, 2 no pooling phase, 4 no MFMAs, 8 no staging writes
#endif
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
// (inline asm: through __builtin_elementwise_max the compiler first canonicalises every operand it has not produced itself - a
// v_pk_max_f16 x, x, x per loaded word, 34 of the 70 packed maxima of a tile; the staged values are finite bf16 patterns or the
// -inf identity, for which the plain instruction is exact)
#ifndef PH_POOL_VARIANT
#define PH_POOL_VARIANT 1      // 1 = inline-asm packed maximum; 2 = negated weight rows (OFF: not bitwise - see below)
#endif
__device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b) {
#if PH_POOL_VARIANT & 1
  unsigned r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)));
#endif
}
__device__ __forceinline__ unsigned pkmax3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
constexpr unsigned POOL_IDENT = 0xFC00FC00u;
constexpr int PROWB = TW * 64;   // bf16 elements of one staged conv row

__global__ __launch_bounds__(256, 2) void stem_fwd_pool_kernel(PhStemPool p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ldsX = smem;                 // 2 x XB
  unsigned char* mid = smem + 2 * XB;         // [2][7][16][64] bf16
  unsigned char* last = mid + 2 * 7 * PROWB * 2;   // [3][16][64] bf16
  // (the wave number read through readfirstlane lives in an SGPR: branches and row addresses derived from it are scalar)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncs = (p.OW + PCOLS - 1) / PCOLS;
  int blk = blockIdx.x;
  const int split = blk % p.nsplit; blk /= p.nsplit;
  const int cs = blk % ncs, b = blk / ncs;
  const int tiles_h = (p.OH + TH - 1) / TH, per = (tiles_h + p.nsplit - 1) / p.nsplit;
  const int tr_own0 = split * per, tr_end = min(tiles_h, tr_own0 + per);
  const int tr_begin = tr_own0 > 0 ? tr_own0 - 1 : 0;       // (a cut strip recomputes the tile above for its last row)
  const int x0 = cs * PCOLS - 1;
  const bf16* x4 = reinterpret_cast<const bf16*>(p.x4);

  const int m = wave * 32 + (lane & 31), khalf = lane >> 5;
  // ALL weight fragments of the lane live in registers for the whole strip (7 x 2 x 2 fragments of 16 bytes = 112 VGPRs):
  // with the weights in LDS a tile needed 28 + 14 ds_read_b128 per wave for its 28 MFMAs and the LDS array, not the
  // matrix pipe, set the tile time (ablation with everything but the operand reads removed: 111 of 291 us)
  bf16x8 wfrag[7][2][2];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        // (round 6: accumulator set j of lane l = channel 2 l + j - the lane owns a channel PAIR, see stem_fwd_kernel)
        wfrag[kh][s2][j] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.w) +
                                                            ((size_t)(kh * 64 + 2 * (lane & 31) + j) * 32 + (s2 * 2 + khalf) * 8));
  // (PH_POOL_VARIANT & 2, measured and OFF, round 6) Channels with gamma < 0 need the window's MINIMUM: with their weight rows
  // NEGATED once per strip the accumulators would hold -y and the staged word would need no exclusive-or (16 of a tile's vector
  // instructions).  But the matrix pipe's fp32 accumulation is NOT sign-symmetric: A (-B) + C differs from -(A B + C) in the last
  // bit for rare operands - at B = 64 / 512 x 512 a handful of the 67 M pooled values moved by one bf16 step and the eval-mode
  // comparison with the separate passes (bitwise) failed.  Kept as a documented dead end.
  float wsgn[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bool neg = (PH_POOL_VARIANT & 2) && p.gamma[2 * (lane & 31) + j] < 0.f;
    wsgn[j] = neg ? -1.f : 1.f;
    const unsigned fx = neg ? 0x80008000u : 0u;
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 w_ = __builtin_bit_cast(u32x4, wfrag[kh][s2][j]);
        w_[0] ^= fx; w_[1] ^= fx; w_[2] ^= fx; w_[3] ^= fx;
        wfrag[kh][s2][j] = __builtin_bit_cast(bf16x8, w_);
      }
  }
  const int pbase = ((m >> 4) * 2) * HPW + (m & 15) * 2;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  // ---- halo: the thread's (up to) four pixels of the 21 x 38 halo keep their offsets for the whole strip; only the row
  // validity changes from tile to tile
  constexpr int HCH = (HP + 255) / 256;
  u32x2 hreg[HCH];
  int hrow[HCH];            // halo row of element e, or a large value when the element does not exist / its column is outside
  long hoff[HCH];           // element offset from the tile's first halo row
  {
    const int ix_base = x0 * 2 - 3;
#pragma unroll
    for (int e = 0; e < HCH; ++e) {
      const int i = tid + e * 256;
      const int hr = i / HPW, hc = i - hr * HPW;
      const int ix = ix_base + hc;
      hrow[e] = (i < HP && ix >= 0 && ix < p.IW) ? hr : (1 << 28);
      hoff[e] = ((long)hr * p.IW + ix) * 4;
    }
  }
  const bf16* ximg = x4 + (size_t)b * p.IH * p.IW * 4;
  auto load_halo_regs = [&](int tr) {
    const int iy_base = tr * TH * 2 - 3;
    const bf16* rowbase = ximg + (long)iy_base * p.IW * 4;
#pragma unroll
    for (int e = 0; e < HCH; ++e) {
      const unsigned iy = (unsigned)(iy_base + hrow[e]);      // (invalid elements: a huge row)
      const bf16* src = iy < (unsigned)p.IH ? rowbase + hoff[e] : reinterpret_cast<const bf16*>(stem_zero8);
      hreg[e] = ld_global<u32x2>(src);
    }
  };
  auto store_halo_regs = [&](unsigned char* dst) {
#pragma unroll
    for (int e = 0; e < HCH; ++e) {
      const int i = tid + e * 256;
      if (i < HP) *reinterpret_cast<u32x2*>(dst + i * 8) = hreg[e];
    }
  };
  // ---- pooling role of this thread: (pooled row prl of 4, pooled column pcl of 7, 8-channel group cg)
  const int cg = tid & 7, pcl = (tid >> 3) % 7, prl = (tid >> 3) / 7;
  unsigned pflip[4];      // sign-bit flips of the group's channel pairs (gamma < 0)
#pragma unroll
  for (int k = 0; k < 4; ++k)
    pflip[k] = (p.gamma[cg * 8 + 2 * k] < 0.f ? 0x8000u : 0u) | (p.gamma[cg * 8 + 2 * k + 1] < 0.f ? 0x80000000u : 0u);
  const int poff = (2 * pcl * 64 + cg * 8) * 2;               // byte offset of the window's first column in a staged row
  // ---- epilogue role: the lane's channel pair (lane & 30, + 1) of each 32-channel half j; its staging byte offset
  // ---- epilogue role: the lane's channel pair (2 l, 2 l + 1); staging byte offset of its word in pixel column 4 khalf
  const int eoff = ((4 * khalf) * 64 + 2 * (lane & 31)) * 2;
  const unsigned eflipw = (PH_POOL_VARIANT & 2) ? 0u : ((p.gamma[2 * (lane & 31)] < 0.f ? 0x8000u : 0u) |
                                                        (p.gamma[2 * (lane & 31) + 1] < 0.f ? 0x80000000u : 0u));
  // local columns 0 and 15 of a tile belong to the neighbouring strips: registers (khalf 0: q = 0, 8), (khalf 1: q = 7, 15)
  const float own_lo = khalf == 0 ? 0.f : 1.f, own_hi = khalf == 1 ? 0.f : 1.f;
  // a strip that starts at the top of the image reads a "row above" that does not exist
  for (int i = tid; i < 3 * PROWB / 2; i += 256) reinterpret_cast<unsigned*>(last)[i] = POOL_IDENT;

  if (tr_begin < tr_end) {
    load_halo_regs(tr_begin);
    store_halo_regs(ldsX);
  }
  __syncthreads();
  const bool strip_inside = x0 >= 0 && x0 + TW <= p.OW;
  for (int tr = tr_begin; tr < tr_end; ++tr) {
    const int tpar = tr - tr_begin;
    const unsigned char* ldsXc = ldsX + (tpar & 1) * XB;
    if (!(PH_STEM_ABL & 1) && tr + 1 < tr_end) load_halo_regs(tr + 1);
    f32x16 acc[2];
    // all fourteen input fragments of the tile are requested before the first MFMA (the accumulators and the pooling window
    // are dead here: 56 registers are free): as `read, wait, two MFMAs` per step every step exposed an LDS round trip
    bf16x8 afr[14];
#pragma unroll
    for (int st = 0; st < 14; ++st)
      afr[st] = *reinterpret_cast<const bf16x8*>(ldsXc + (pbase + (st >> 1) * HPW + 4 * (st & 1) + 2 * khalf) * 8);
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 a = afr[kh * 2 + s];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (kh == 0 && s == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wfrag[0][0][j], zero, 0, 0, 0);   // (C = inline constant 0)
          } else if (PH_STEM_ABL & 4) { asm volatile("" ::"v"(a), "v"(wfrag[kh][s][j])); acc[j][0] += 1.f; }
          else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wfrag[kh][s][j], acc[j], 0, 0, 0);
        }
      }
    }
    // ---- epilogue 1: statistics of the OWNED pixels (fp32 accumulators), bf16 outputs to the staging tile
    const bool own_tile = tr >= tr_own0;
    const int r0 = tr * TH;
    unsigned char* midb = mid + (tpar & 1) * 7 * PROWB * 2;
    unsigned char* lastb = last + (tpar % 3) * PROWB * 2;
    const unsigned char* prevb = last + ((tpar + 2) % 3) * PROWB * 2;
    // the wave's two conv rows of the tile: 2 * wave and 2 * wave + 1 (the tile's row 7 is the carried one)
    unsigned char* erow0 = midb + (2 * wave) * PROWB * 2 + eoff;
    unsigned char* erow1 = (wave < 3 ? midb + (2 * wave + 1) * PROWB * 2 : lastb) + eoff;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    // (out-of-image pixels are staged like any other: the pooling pass skips them by coordinate)
    // (`OWN` - does the tile count for the statistics - is a template argument, not a branch around each of the 16 updates:
    // the inline-asm statistics ops cannot be hoisted, so hipcc had emitted one s_cbranch per register pair, and ~30 scalar
    // branches per tile are what made this phase cost ~12 cycles per vector instruction)
    auto stage = [&](auto insidec, auto ownc) {
      constexpr bool INSIDE = decltype(insidec)::value;
      constexpr bool OWN = decltype(ownc)::value;
      if constexpr (!INSIDE) asm volatile("; edge tile: masked statistics" ::: "memory");   // keeps the two paths apart
#pragma unroll
      for (int q2 = 0; q2 < 8; ++q2) {
        constexpr int CO[4] = {0, 2, 8, 10};
        const int clc = CO[q2 & 3];                      // + 4 * khalf (in eoff): local column of register 2 * q2
        unsigned char* dst = (q2 < 4 ? erow0 : erow1) + clc * 128;
        float k0 = 1.f, k1 = 1.f;
        if constexpr (!INSIDE) {
          const int r = r0 + 2 * wave + (q2 >> 2), c = x0 + clc + 4 * khalf;
          k0 = (r < p.OH && c >= 0 && c < p.OW) ? 1.f : 0.f;
          k1 = (r < p.OH && c + 1 >= 0 && c + 1 < p.OW) ? 1.f : 0.f;
        }
        if (q2 == 0 || q2 == 4) k0 *= own_lo;            // register 0 / 8: local column 0 when khalf == 0
        if (q2 == 3 || q2 == 7) k1 *= own_hi;            // register 7 / 15: local column 15 when khalf == 1
        const bool edge0 = !INSIDE || q2 == 0 || q2 == 4, edge1 = !INSIDE || q2 == 3 || q2 == 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float v0 = acc[j][2 * q2], v1 = acc[j][2 * q2 + 1];
          if constexpr (OWN && !(PH_STEM_ABL & 32)) {
            const float w0 = edge0 ? v0 * k0 : v0, w1 = edge1 ? v1 * k1 : v1;
            s1[j] = fadd_s(fadd_s(s1[j], w0), w1);
            s2[j] = ffma_s(w1, w1, ffma_s(w0, w0, s2[j]));
          }
        }
        if (PH_STEM_ABL & 64) { asm volatile("" ::"v"(acc[0][2 * q2]), "v"(acc[1][2 * q2])); continue; }
        // the lane's channel pair of the two pixels (local columns clc, clc + 1): one conversion and one LDS word each
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          bf16x2 w;
          w[0] = (bf16)acc[0][2 * q2 + px];
          w[1] = (bf16)acc[1][2 * q2 + px];
          const unsigned word = __builtin_bit_cast(unsigned, w) ^ eflipw;      // [even channel, odd channel], gamma < 0: sign flipped
          if (PH_STEM_ABL & 8) asm volatile("" ::"v"(word));
          else *reinterpret_cast<unsigned*>(dst + px * 128) = word;
        }
      }
    };
    if (own_tile) {
      if (strip_inside && r0 + TH <= p.OH) stage(std::true_type{}, std::true_type{});
      else stage(std::false_type{}, std::true_type{});
    } else {
      stage(std::false_type{}, std::false_type{});
    }
    if (tr + 1 < tr_end) store_halo_regs(ldsX + ((tpar & 1) ^ 1) * XB);
    // LDS-only barrier (__syncthreads() would also drain the vector-memory counter: the pooled stores of the tile before)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- epilogue 2: 4 x 7 pooled pixels x 64 channels from the staging tile (+ the previous tile's last row)
    if (!(PH_STEM_ABL & 2) && own_tile && tid < 224) {
      const int ph = tr * 4 + prl, pw = cs * 7 + pcl;
      if (ph < p.PH && pw < p.PW) {
        u32x4 best = {POOL_IDENT, POOL_IDENT, POOL_IDENT, POOL_IDENT};
        // a window leaves the image only at column -1 (pw == 0), at column OW (last pw of an odd OW) and at row OH (last
        // ph of an odd OH); row -1 (ph == 0) reads the identity rows `last` was initialised with
        const bool cut_l = pw == 0, cut_r = 2 * pw + 1 >= p.OW, cut_b = 2 * ph + 1 >= p.OH;
        auto pool = [&](auto edgec) {
          constexpr bool EDGE = decltype(edgec)::value;
          if constexpr (EDGE) asm volatile("; edge window" ::: "memory");
          u32x4 v[3][3];       // all nine 16-byte reads are issued before the first compare (the accumulators are dead here)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int R = 2 * prl + kh;              // staging row: 0 = last row of the tile above, 1..8 = this tile
            const unsigned char* rowp = (R == 0 ? prevb : (R <= 7 ? midb + (R - 1) * PROWB * 2 : lastb)) + poff;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) v[kh][kw] = *reinterpret_cast<const u32x4*>(rowp + kw * 128);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (EDGE) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) {
                const bool skip = (kw == 0 && cut_l) || (kw == 2 && cut_r) || (kh == 2 && cut_b);
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) v[kh][kw][kq] = skip ? POOL_IDENT : v[kh][kw][kq];
              }
          }
          // the nine taps as a tree of three-input packed maxima (v_pk_maximum3_f16, gfx950): 4 instructions per word instead of 9
#pragma unroll
          for (int kq = 0; kq < 4; ++kq)
            best[kq] = pkmax3(pkmax3(v[0][0][kq], v[0][1][kq], v[0][2][kq]), pkmax3(v[1][0][kq], v[1][1][kq], v[1][2][kq]),
                              pkmax3(v[2][0][kq], v[2][1][kq], v[2][2][kq]));
        };
        if (cut_l || cut_r || cut_b) pool(std::true_type{});
        else pool(std::false_type{});
#pragma unroll
        for (int k = 0; k < 4; ++k) best[k] ^= pflip[k];
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(p.pooled) + ((((size_t)b * p.PH + ph) * p.PW + pw) * 64 + cg * 8)) = best;
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][2][64] (the halo buffers are dead)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s1[j] *= wsgn[j];      // (the negated channels accumulated -y: exact)
      const float a1 = s1[j] + __shfl_xor(s1[j], 32, 64);
      const float a2 = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (khalf == 0) {
        red[(wave * 2 + 0) * 64 + 2 * (lane & 31) + j] = a1;
        red[(wave * 2 + 1) * 64 + 2 * (lane & 31) + j] = a2;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, n = tid & 63;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[(w * 2 + which) * 64 + n];
      p.stats[((size_t)blockIdx.x * 2 + which) * 64 + n] = v;
    }
  }
}

__device__ __forceinline__ int sw_piece(int pix, int piece) { return piece ^ (((pix >> 1) & 1) << 1); }

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* base, int off0, int off1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
  union { s16x4 s[2]; bf16x8 v; } u;
  u.s[0] = a; u.s[1] = b;
  return u.v;
}

template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(PhStemWgrad p) {
  // T = hp16 (PH_PREC_FP16X3): two fp16 planes of dz' (a [hi 64 | lo 64] pixel record) and of the image; the leading product
  // and the two cross products accumulate separately and are combined (cross * 2^-11) when the slab is written
  constexpr bool HPM = is_hp<T>::value;
  constexpr bool SPLIT = is_f32<T>::value;
  constexpr int NP = SPLIT ? PH_NPLANES : (HPM ? 2 : 1);
  constexpr int DB = TH * TW * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ldsD = smem;               // NP planes of DB
  unsigned char* ldsX = smem + DB * NP;     // NP planes of XB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x;
  const int tiles_w = (p.OW + TW - 1) / TW, tiles_h = (p.OH + TH - 1) / TH;
  const int tiles_img = tiles_w * tiles_h, ntiles = tiles_img * p.B;
  const int t_begin = chunk * p.tiles_per_chunk, t_end = min(ntiles, t_begin + p.tiles_per_chunk);
  const T* X = reinterpret_cast<const T*>(p.x4);
  const T* DY = reinterpret_cast<const T*>(p.dy);
  // wave w owns kernel rows kh = w and w+4 (w+4 == 7 does not exist)
  const int nkh = (wave + 4 < 7) ? 2 : 1;
  f32x16 acc[2][2], accx[HPM ? 2 : 1][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc[a][j][q] = 0.f; if (HPM) accx[HPM ? a : 0][j][q] = 0.f; }
  const int q4 = (lane & 15) >> 2, p4 = lane & 3, colhalf = (lane >> 4) & 1, khalf = lane >> 5;

  // perf mode: the NEXT tile's operands are requested into registers before this tile's MFMAs and written to LDS after them (a
  // tile is only 32 MFMAs per wave against 22 KB of operands: staged synchronously - loads, wait, LDS writes, barrier, MFMAs - the
  // kernel ran at 2.8 TB/s of its 640 MB)
  constexpr bool PREF = !SPLIT;      // (half-pair mode: both fp16 planes of either operand)
  constexpr int NPL = HPM ? 2 : 1;
  constexpr int DCH = TH * TW * 8 / 256, XCH = (HP + 255) / 256;
  u32x4 dreg[PREF ? NPL : 1][PREF ? DCH : 1];
  u32x2 xreg[PREF ? NPL : 1][PREF ? XCH : 1];
  auto prefetch = [&](int tt) {
    if constexpr (PREF) {
      const int b = tt / tiles_img, ti = tt - b * tiles_img;
      const int r0 = (ti / tiles_w) * TH, c0 = (ti % tiles_w) * TW;
      const unsigned char* zero = reinterpret_cast<const unsigned char*>(stem_zero8);
#pragma unroll
      for (int e = 0; e < DCH; ++e) {
        const int i = tid + e * 256, pix = i >> 3, ch = i & 7;
        const int r = r0 + (pix >> 4), c = c0 + (pix & 15);
        const bool ok = r < p.OH && c < p.OW;
        const size_t px = ((size_t)b * p.OH + r) * p.OW + c;
        if constexpr (HPM) {      // a [hi 64 | lo 64] fp16 pixel record
          const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(reinterpret_cast<const f16*>(p.dy) + px * 128 + ch * 8) : zero;
          dreg[0][e] = ld_global<u32x4>(src);
          dreg[HPM ? 1 : 0][e] = ld_global<u32x4>(ok ? src + 128 : zero);
        } else {
          const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(DY + px * 64 + ch * 8) : zero;
          dreg[0][e] = ld_global<u32x4>(src);
        }
      }
      const int iy_base = r0 * 2 - 3, ix_base = c0 * 2 - 3;
#pragma unroll
      for (int e = 0; e < XCH; ++e) {
        const int i = tid + e * 256;
        const int hr = i / HPW, hc = i - hr * HPW;
        const int iy = iy_base + hr, ix = ix_base + hc;
        const bool ok = i < HP && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        const size_t g = (((size_t)b * p.IH + iy) * p.IW + ix) * 4;
        if constexpr (HPM) {      // the two fp16 planes [2][B IH IW][4] of ph_pack_input_launch
          const f16* xp = reinterpret_cast<const f16*>(p.x4);
          xreg[0][e] = ld_global<u32x2>(ok ? reinterpret_cast<const unsigned char*>(xp + g) : zero);
          xreg[HPM ? 1 : 0][e] = ld_global<u32x2>(ok ? reinterpret_cast<const unsigned char*>(xp + (size_t)p.B * p.IH * p.IW * 4 + g) : zero);
        } else {
          xreg[0][e] = ld_global<u32x2>(ok ? reinterpret_cast<const unsigned char*>(X + g) : zero);
        }
      }
    }
  };
  auto commit = [&]() {
    if constexpr (PREF) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int e = 0; e < DCH; ++e) {
          const int i = tid + e * 256, pix = i >> 3, ch = i & 7;
          *reinterpret_cast<u32x4*>(ldsD + pl * DB + pix * 128 + sw_piece(pix, ch >> 1) * 32 + (ch & 1) * 16) = dreg[pl][e];
        }
#pragma unroll
        for (int e = 0; e < XCH; ++e) {
          const int i = tid + e * 256;
          if (i < HP) *reinterpret_cast<u32x2*>(ldsX + pl * XB + i * 8) = xreg[pl][e];
        }
      }
    }
  };
  if (PREF && t_begin < t_end) prefetch(t_begin);
  for (int tt = t_begin; tt < t_end; ++tt) {
    const int b = tt / tiles_img, ti = tt - b * tiles_img;
    const int r0 = (ti / tiles_w) * TH, c0 = (ti % tiles_w) * TW;
    __syncthreads();
    if constexpr (PREF) {
      commit();
      __syncthreads();
      if (tt + 1 < t_end) prefetch(tt + 1);
    } else {
    for (int i = tid; i < TH * TW * 8; i += 256) {
      const int pix = i >> 3, ch = i & 7;
      const int r = r0 + (pix >> 4), c = c0 + (pix & 15);
      const bool ok = r < p.OH && c < p.OW;
      const int off = pix * 128 + sw_piece(pix, ch >> 1) * 32 + (ch & 1) * 16;
      const T* src = DY + (((size_t)b * p.OH + r) * p.OW + c) * 64 + ch * 8;
      if constexpr (HPM) {
        const f16* sh = reinterpret_cast<const f16*>(p.dy) + (((size_t)b * p.OH + r) * p.OW + c) * 128 + ch * 8;
        u32x4 vh = {0u, 0u, 0u, 0u}, vl = {0u, 0u, 0u, 0u};
        if (ok) { vh = *reinterpret_cast<const u32x4*>(sh); vl = *reinterpret_cast<const u32x4*>(sh + 64); }
        *reinterpret_cast<u32x4*>(ldsD + off) = vh;
        *reinterpret_cast<u32x4*>(ldsD + DB + off) = vl;
      } else if constexpr (!SPLIT) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) v = *reinterpret_cast<const u32x4*>(src);
        *reinterpret_cast<u32x4*>(ldsD + off) = v;
      } else {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
        if (ok) load8(src, v);
        bf16x8 p0, p1, p2;
#pragma unroll
        for (int q = 0; q < 8; ++q) { bf16 a, b2, c2; split3_bf16(v[q], a, b2, c2); p0[q] = a; p1[q] = b2; p2[q] = c2; }
        *reinterpret_cast<bf16x8*>(ldsD + off) = p0;
        *reinterpret_cast<bf16x8*>(ldsD + DB + off) = p1;
        *reinterpret_cast<bf16x8*>(ldsD + 2 * DB + off) = p2;
      }
    }
    stage_halo<T>(X, b, p.IH, p.IW, r0 * 2 - 3, c0 * 2 - 3, ldsX, tid, p.B);
    __syncthreads();
    }
#pragma unroll 2
    for (int kk = 0; kk < TH * TW / 16; ++kk) {
      const int t0 = kk * 16 + 8 * khalf + q4, t1 = t0 + 4;
      bf16x8 a[NP][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int piece = j * 2 + colhalf;
        const int o0 = t0 * 128 + sw_piece(t0, piece) * 32 + p4 * 8;
        const int o1 = t1 * 128 + sw_piece(t1, piece) * 32 + p4 * 8;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) a[pl][j] = tr_pair(ldsD + pl * DB, o0, o1);
      }
      const int kw = 4 * colhalf + p4;
      const int hb0 = ((t0 >> 4) * 2) * HPW + (t0 & 15) * 2 + kw;
      const int hb1 = ((t1 >> 4) * 2) * HPW + (t1 & 15) * 2 + kw;
#pragma unroll
      for (int ai = 0; ai < 2; ++ai) {
        if (ai < nkh) {
          const int kh = wave + 4 * ai;
          const int o0 = (hb0 + kh * HPW) * 8, o1 = (hb1 + kh * HPW) * 8;
          bf16x8 bq[NP];
#pragma unroll
          for (int pl = 0; pl < NP; ++pl) bq[pl] = tr_pair(ldsX + pl * XB, o0, o1);
          if constexpr (HPM) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f16x8 ah = __builtin_bit_cast(f16x8, a[0][j]), al = __builtin_bit_cast(f16x8, a[1][j]);
              const f16x8 bh = __builtin_bit_cast(f16x8, bq[0]), bl = __builtin_bit_cast(f16x8, bq[1]);
              if (!p.hp_hi_only) {      // (PH_PREC_FP16X1: the leading product alone)
                accx[HPM ? ai : 0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx[HPM ? ai : 0][j], 0, 0, 0);
                accx[HPM ? ai : 0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx[HPM ? ai : 0][j], 0, 0, 0);
              }
              acc[ai][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[ai][j], 0, 0, 0);
            }
          } else if constexpr (SPLIT) {
#define PH_MM(PI, PJ)                                                                                       \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
      acc[ai][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PI][j], bq[PJ], acc[ai][j], 0, 0, 0);
            if (p.prod6) { PH_SPLIT_PAIRS_LO(PH_MM) }
            PH_SPLIT_PAIRS_HI(PH_MM)
#undef PH_MM
          } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[ai][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][j], bq[0], acc[ai][j], 0, 0, 0);
          }
        }
      }
    }
  }
  float* slab = p.slab + (size_t)chunk * 7 * 64 * 32;
#pragma unroll
  for (int ai = 0; ai < 2; ++ai) {
    if (ai < nkh) {
      const int kh = wave + 4 * ai;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = j * 32 + (q & 3) + 8 * (q >> 2) + 4 * khalf;
          slab[((size_t)kh * 64 + row) * 32 + (lane & 31)] =
              HPM ? acc[ai][j][q] + accx[HPM ? ai : 0][j][q] * PH_HP_LO_INV : acc[ai][j][q];
        }
    }
  }
}

// dw [64][3][7][7] <- sum_chunks slab[chunk][kh][co][kw*4+ch].  64 outputs x 4 chunk-lanes per block, 4 independent
// partial sums per thread (loads in flight), fixed summation order (bitwise reproducible).  (One thread per output
// walking all 512 chunks serially took 156 us for 29 MB.)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                int nchunks, const float* __restrict__ unscale) {
  const int o = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  const size_t n = (size_t)7 * 64 * 32;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < 64 * 3 * 49) {
    const int kw = i % 7, kh = (i / 7) % 7, ch = (i / 49) % 3, co = i / 147;
    const size_t e = ((size_t)kh * 64 + co) * 32 + kw * 4 + ch;
    int c = cl;
    for (; c + 12 < nchunks; c += 16) {
      s0 += slab[(size_t)c * n + e];
      s1 += slab[(size_t)(c + 4) * n + e];
      s2 += slab[(size_t)(c + 8) * n + e];
      s3 += slab[(size_t)(c + 12) * n + e];
    }
    for (; c < nchunks; c += 4) s0 += slab[(size_t)c * n + e];
  }
  __shared__ float sh[4][64];
  sh[cl][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (cl == 0 && i < 64 * 3 * 49) dw[i] = ((sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o])) * (unscale ? unscale[1] : 1.f);
}

template <typename K>
int set_lds(K kern, int bytes, bool& done) {
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) !=
        hipSuccess)
      return PH_ELAUNCH;
    done = true;
  }
  return PH_OK;
}

}  // namespace

int ph_stem_stat_parts(int B, int OH, int OW) { return cdiv(B * cdiv(OH, TH) * cdiv(OW, TW), STEM_TPW); }

int ph_stem_fwd_launch(const PhStem* p, int prec, hipStream_t st) {
  dim3 grid(ph_stem_stat_parts(p->B, p->OH, p->OW));
  void* tok = nullptr;
  if (ph_prof_on())   // bytes: the packed image (4 channels) once + the 64-channel output
    ph_prof_begin2(PH_CLS_STEM_FWD, 2.0 * p->B * p->OH * p->OW * 64.0 * 147.0,
                   ((double)p->B * p->IH * p->IW * 4 + (double)p->B * p->OH * p->OW * 64) * (prec == PH_PREC_BF16 ? 2.0 : 4.0), st, &tok);
  struct EndGuard { void* t; hipStream_t s; ~EndGuard() { ph_prof_end(t, s); } } guard{tok, st};
  if (prec == PH_PREC_BF16) {
    static bool done = false;
    const int lds = 2 * XB + WB;
    if (set_lds(stem_fwd_kernel<bf16>, lds, done)) return PH_ELAUNCH;
    // 3 x CUs workgroups (what the registers admit: 3 waves per SIMD) walk the blocks - the 28 KB weight image is staged once per
    // workgroup: 173 -> 165 us per launch, two alternations on one box; 2 x: 185, 4 x: 209.  PH_STEM_VB16=n overrides, 0 = one block each
    static const int vb16 = [] { const char* e = getenv("PH_STEM_VB16"); return e ? atoi(e) : 3; }();
    PhStem q = *p;
    if (vb16 > 0 && (int)grid.x > vb16 * ph_num_cus()) { q.vblocks = (int)grid.x; grid.x = vb16 * ph_num_cus(); }
    hipLaunchKernelGGL(stem_fwd_kernel<bf16>, grid, dim3(256), lds, st, q);
  } else if (PH_IS_SPLIT_PREC(prec)) {
    static bool done = false;
    const int lds = PH_NPLANES * (XB + WB);
    if (set_lds(stem_fwd_kernel<float>, lds, done)) return PH_ELAUNCH;
    PhStem q = *p;
    q.prod6 = prec == PH_PREC_BF16X6;
    hipLaunchKernelGGL(stem_fwd_kernel<float>, grid, dim3(256), lds, st, q);
  } else if (prec == PH_PREC_FP16X3) {
    static bool done = false;
    const int lds = 4 * XB + TH * TW * 64 * 4;      // two halo buffers of two planes + the fp32 output image (the weights live in registers)
    if (set_lds(stem_fwd_kernel<hp16>, lds, done)) return PH_ELAUNCH;
    // one wave per SIMD (the weights hold 336 of the 512 registers): two workgroups per CU can never be resident, so at most
    // 2 x CUs workgroups walk the blocks (PH_STEM_VB=0: one block per workgroup, A/B switch)
    static const int vb_mul = [] { const char* e = getenv("PH_STEM_VB"); return e ? atoi(e) : 1; }();
    PhStem q = *p;
    const int cap = vb_mul * ph_num_cus();
    if (vb_mul > 0 && (int)grid.x > cap) { q.vblocks = (int)grid.x; grid.x = cap; }
    hipLaunchKernelGGL(stem_fwd_kernel<hp16>, grid, dim3(256), lds, st, q);
  } else {
    return PH_EINVAL;
  }
  PH_LAUNCH_CHECK();
  return PH_OK;
}


namespace {
int stem_pool_nsplit(int B, int OH, int OW) {
  const int ncs = cdiv(OW, PCOLS), th = cdiv(OH, TH);
  int ns = 1;
  while (B * ncs * ns < 2048 && th / (ns * 2) >= 4) ns *= 2;   // >= 8 workgroups per CU; a cut costs one recomputed tile
  return ns;
}
}  // namespace

int ph_stem_pool_stat_parts(int B, int OH, int OW) { return B * cdiv(OW, PCOLS) * stem_pool_nsplit(B, OH, OW); }

int ph_stem_fwd_pool_launch(const PhStemPool* p_, hipStream_t st) {
  PhStemPool p = *p_;
  if (!p.x4 || !p.w || !p.pooled || !p.gamma) return PH_EINVAL;
  p.nsplit = stem_pool_nsplit(p.B, p.OH, p.OW);
  void* tok = nullptr;
  if (ph_prof_on())   // bytes: the packed image once + the pooled 64-channel output
    ph_prof_begin2(PH_CLS_STEM_FWD, 2.0 * p.B * p.OH * p.OW * 64.0 * 147.0,
                   ((double)p.B * p.IH * p.IW * 4 + (double)p.B * p.PH * p.PW * 64) * 2.0, st, &tok);
  struct EndGuard { void* t; hipStream_t s; ~EndGuard() { ph_prof_end(t, s); } } guard{tok, st};
  static bool done = false;
  const int lds = 2 * XB + (2 * 7 + 3) * PROWB * 2;
  if (set_lds(stem_fwd_pool_kernel, lds, done)) return PH_ELAUNCH;
  hipLaunchKernelGGL(stem_fwd_pool_kernel, dim3(ph_stem_pool_stat_parts(p.B, p.OH, p.OW)), dim3(256), lds, st, p);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_stem_wgrad_launch(const PhStemWgrad* p, int prec, hipStream_t st) {
  dim3 grid(p->nchunks);
  void* tok = nullptr;
  if (ph_prof_on())
    ph_prof_begin2(PH_CLS_STEM_WGRAD, 2.0 * p->B * p->OH * p->OW * 64.0 * 147.0,
                   ((double)p->B * p->IH * p->IW * 4 + (double)p->B * p->OH * p->OW * 64) * (prec == PH_PREC_BF16 ? 2.0 : 4.0), st, &tok);
  struct EndGuard { void* t; hipStream_t s; ~EndGuard() { ph_prof_end(t, s); } } guard{tok, st};
  const int base = TH * TW * 128 + XB;
  if (prec == PH_PREC_BF16) {
    static bool done = false;
    if (set_lds(stem_wgrad_kernel<bf16>, base, done)) return PH_ELAUNCH;
    hipLaunchKernelGGL(stem_wgrad_kernel<bf16>, grid, dim3(256), base, st, *p);
  } else if (PH_IS_SPLIT_PREC(prec)) {
    static bool done = false;
    if (set_lds(stem_wgrad_kernel<float>, PH_NPLANES * base, done)) return PH_ELAUNCH;
    PhStemWgrad q = *p;
    q.prod6 = prec == PH_PREC_BF16X6;
    hipLaunchKernelGGL(stem_wgrad_kernel<float>, grid, dim3(256), PH_NPLANES * base, st, q);
  } else if (prec == PH_PREC_FP16X3 || prec == PH_PREC_FP16X1) {
    static bool done = false;
    if (set_lds(stem_wgrad_kernel<hp16>, 2 * base, done)) return PH_ELAUNCH;
    PhStemWgrad q = *p;
    q.hp_hi_only = prec == PH_PREC_FP16X1;
    hipLaunchKernelGGL(stem_wgrad_kernel<hp16>, grid, dim3(256), 2 * base, st, q);
  } else {
    return PH_EINVAL;
  }
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_stem_wgrad_reduce_launch(const float* slab, float* dw, int nchunks, const float* unscale, hipStream_t st) {
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((64 * 147 + 63) / 64), dim3(256), 0, st, slab, dw, nchunks, unscale);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
