// Weight-gradient (wgrad) of the 3x3 / 1x1 convolutions on MFMA, plus weight packing kernels.
//
//   dW[tap][cout][cin] = sum_{b,r,c} dY[b,r,c,cout] * X[b, r*S + kh - pad, c*S + kw - pad, cin]
// GEMM view: M = cout, N = cin, K = pixels (huge).  Both operands are K-major in HBM (NHWC: pixel is the
// slow index), so both MFMA fragments are fetched from LDS with the CDNA4 transposing read
// ds_read_b64_tr_b16 - the tiles are staged exactly as they lie in HBM (coalesced 16-B rows), no
// transposed copy is ever made.  One workgroup owns a 64x64 (cout x cin) block for ALL taps and walks a
// chunk of spatial tiles, keeping 9 x 32x32 accumulators per wave in registers (144 VGPRs); the X halo
// tile is staged once per spatial tile and re-read by all 9 taps.  Partial sums go to a slab
// [chunk][tap][cout][cin] and are combined in fixed order by a reduce kernel (bitwise reproducible,
// no float atomics).
#include "ph_common.h"
#include "ph_kernels.h"

namespace {

// one LDS-DMA wave-instruction: lane l copies 16 B from its global address g to LDS byte lds_addr + 16*l
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_addr), "v"(g) : "memory");
}

// 32-B piece swizzle of a [pixel][64 ch] bf16 image (128-B rows): makes the 4-row tr-reads conflict-free
__device__ __forceinline__ int sw_piece(int pix, int piece) { return piece ^ (((pix >> 1) & 1) << 1); }

template <typename T, int S, int TH, int KS>
struct WgCfg {
  static constexpr bool SPLIT = is_f32<T>::value;
  static constexpr bool HPM = is_hp<T>::value;   // PH_PREC_FP16X3: both operands are half-pair tensors, three passes over the chunk
  static constexpr int TW = 16, BM = TH * TW;
  static constexpr int HPH = (TH - 1) * S + KS, HPW = (TW - 1) * S + KS, HP = HPH * HPW;
  static constexpr int HPP = (HP + 7) / 8 * 8;            // halo pixels padded to whole 1-KiB LDS-DMA pieces
  static constexpr int D_BYTES = BM * 128, X_BYTES = HPP * 128;
  static constexpr int NP = SPLIT ? PH_NPLANES : 1;
  // perf mode: two LDS buffers (tile t+1 lands by LDS-DMA while tile t feeds the MFMAs)
  static constexpr int LDS_BYTES = SPLIT ? (D_BYTES + X_BYTES) * NP : 2 * (D_BYTES + X_BYTES);
  static constexpr int NT = KS * KS;
};

// stage 8 channels of one pixel into plane 0 (bf16) or into the 3 split planes (fp32 source)
template <typename T>
__device__ __forceinline__ void stage_row(const T* src, bool ok, unsigned char* base, int plane_bytes, int off) {
  if constexpr (!is_f32<T>::value) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ok) v = *reinterpret_cast<const u32x4*>(src);
    *reinterpret_cast<u32x4*>(base + off) = v;
  } else {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = 0.f;
    if (ok) load8(src, v);
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int q = 0; q < 8; ++q) { bf16 a, b, c; split3_bf16(v[q], a, b, c); p0[q] = a; p1[q] = b; p2[q] = c; }
    *reinterpret_cast<bf16x8*>(base + off) = p0;
    *reinterpret_cast<bf16x8*>(base + plane_bytes + off) = p1;
    *reinterpret_cast<bf16x8*>(base + 2 * plane_bytes + off) = p2;
  }
}

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* base, int off0, int off1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
  union { s16x4 s[2]; bf16x8 v; } u;
  u.s[0] = a; u.s[1] = b;
  return u.v;
}

template <typename T, int S, int TH, int KS>
__global__ __launch_bounds__(256) void wgrad_kernel(PhWgrad p) {
  using C = WgCfg<T, S, TH, KS>;
  constexpr bool SPLIT = C::SPLIT, HPM = C::HPM;
  constexpr int TW = C::TW, BM = C::BM, HPW = C::HPW, HP = C::HP, NT = C::NT, NP = C::NP;
  typedef typename std::conditional<HPM, f16, T>::type TI;       // element type as the staging code addresses it
  constexpr int EW = HPM ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ldsD = smem;                      // NP planes of D_BYTES
  unsigned char* ldsX = smem + C::D_BYTES * NP;    // NP planes of X_BYTES

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (scalar) by construction
  const int cf = wave >> 1, kf = wave & 1;     // 32-wide cout / cin fragment of this wave
  const int cin_blocks = p.Cin >> 6;
  const int co0 = (blockIdx.x / cin_blocks) << 6, ci0 = (blockIdx.x % cin_blocks) << 6;
  const int chunk = blockIdx.y;
  const int tiles_w = (p.OW + TW - 1) / TW, tiles_h = (p.OH + TH - 1) / TH;
  const int tiles_img = tiles_w * tiles_h, ntiles = tiles_img * p.B;
  const int t_begin = chunk * p.tiles_per_chunk;
  const int t_end = min(ntiles, t_begin + p.tiles_per_chunk);
  const TI* X = reinterpret_cast<const TI*>(p.x);
  const TI* DY = reinterpret_cast<const TI*>(p.dy);
  const long xpix = (p.x_pix_stride ? p.x_pix_stride : p.Cin) * EW;
  const long xrow = (p.x_row_stride ? p.x_row_stride : (long)p.IW * p.Cin) * EW;
  const long ximg = (p.x_img_stride ? p.x_img_stride : (long)p.IH * p.IW * p.Cin) * EW;
  const long dpix = (long)p.Cout * EW;      // elements per dy pixel record

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

  // tr-read lane roles: within each 16-lane group, lane 4q+p supplies row q, columns 4p..4p+3
  const int q4 = (lane & 15) >> 2, p4 = lane & 3, colhalf = (lane >> 4) & 1, khalf = lane >> 5;
  const int pieceA = cf * 2 + colhalf, pieceB = kf * 2 + colhalf;

  // ---- MFMA over one staged tile (K = the tile's pixels, 16 per MFMA)
  auto compute = [&](const unsigned char* dD, const unsigned char* dX) {
#pragma unroll 2
    for (int kk = 0; kk < BM / 16; ++kk) {
      // the two 4-pixel row groups this lane's tr-reads address
      const int t0 = kk * 16 + 8 * khalf + q4, t1 = t0 + 4;
      const int offA0 = t0 * 128 + sw_piece(t0, pieceA) * 32 + p4 * 8;
      const int offA1 = t1 * 128 + sw_piece(t1, pieceA) * 32 + p4 * 8;
      bf16x8 a[NP];      // (third plane: only in the six-product mode)
#ifdef PH_ABL_WG_NOLDS   // timing ablation only (garbage results): no fragment reads
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) { bf16x8 z; asm volatile("" : "=v"(z)); a[pl] = z; }
#else
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        if (pl < 2 || p.prod6) a[pl] = tr_pair(dD + pl * C::D_BYTES, offA0, offA1);
#endif
      const int hb0 = ((t0 >> 4) * S) * HPW + (t0 & 15) * S;
      const int hb1 = ((t1 >> 4) * S) * HPW + (t1 & 15) * S;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int toff = (t / KS) * HPW + (t % KS);
        const int h0 = hb0 + toff, h1 = hb1 + toff;
        const int offB0 = h0 * 128 + sw_piece(h0, pieceB) * 32 + p4 * 8;
        const int offB1 = h1 * 128 + sw_piece(h1, pieceB) * 32 + p4 * 8;
        bf16x8 bq[NP];
#ifdef PH_ABL_WG_NOLDS
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) { bf16x8 z; asm volatile("" : "=v"(z) : "v"(offB0 + offB1)); bq[pl] = z; }
#else
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          if (pl < 2 || p.prod6) bq[pl] = tr_pair(dX + pl * C::X_BYTES, offB0, offB1);
#endif
        if constexpr (SPLIT) {
#define PH_MM(PI, PJ) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PI], bq[PJ], acc[t], 0, 0, 0);
          if (p.prod6) { PH_SPLIT_PAIRS_LO(PH_MM) }
          PH_SPLIT_PAIRS_HI(PH_MM)
#undef PH_MM
        } else if constexpr (HPM) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, bq[0]), acc[t], 0, 0, 0);
        } else {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[0], acc[t], 0, 0, 0);
        }
      }
    }
  };

  // ---- the same MFMA stream with the fragment reads issued THREE taps ahead of the MFMA that consumes them (perf mode, 3 x 3
  // stride 1: the 13 large launches of a step).  compute() above leaves the order to the compiler, which reads a tap's fragments
  // right in front of its MFMA and relies on a second wave of the SIMD to cover the latency - in the step the kernel runs one
  // workgroup per CU beside the BatchNorm-backward passes and 60 % of its time was exposed (skip-ablation: -0.86 ms per step).
  // Every fragment address is one of five per-lane base registers + an immediate: halo pixel h = 18 kk + u + toff (u = 8 khalf +
  // q4, + 4 for the second row group), swizzle bit = bit 1 of h = (kk & 1) ^ bit1(toff) ^ bit1(u) ^ (bit0(u) & bit0(toff)).
  constexpr bool PIPE = !SPLIT && S == 1 && KS == 3 && (BM / 16 == TH);      // (half-pair mode: every pass of its item stream)
  constexpr int PBUF = C::D_BYTES + C::X_BYTES;
  const int u_lane = 8 * khalf + q4;
  const int baseA = u_lane * 128 + ((pieceA ^ (((u_lane >> 1) & 1) << 1)) << 5) + p4 * 8;
  const int baseB = u_lane * 128 + (pieceB << 5) + p4 * 8;
  const int bE0 = baseB ^ (((u_lane >> 1) & 1) << 6), bE1 = bE0 ^ 64;                          // taps with an even toff
  const int bO0 = baseB ^ ((((u_lane >> 1) ^ u_lane) & 1) << 6), bO1 = bO0 ^ 64;               // taps with an odd toff
  // (the accumulators live in AGPRs from here on: without this the loop-carried values sat in vector registers and were copied
  // to the accumulation registers and back around every tile - 144 + 144 moves and 190 vector registers)
  if constexpr (PIPE) {
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+a"(acc[t]));
  }
  auto compute_p = [&](const int boff) __attribute__((always_inline)) {      // boff: byte offset of the LDS buffer pair (ONE body:
    if constexpr (PIPE) {                                                      // two copies made a 144-register phi of the accumulators)
      const unsigned char* dD = smem + boff;
      const unsigned char* dX = dD + C::D_BYTES;
      constexpr int NI = (BM / 16) * 9, LA = 3;
      bf16x8 af[2], bfr[4];
      auto ldA = [&](const int kk) { return tr_pair(dD, baseA + kk * 2048, baseA + 512 + kk * 2048); };
      auto ldB = [&](const int i) {
        const int kk = i / 9, t = i % 9;
        const int toff = (t / 3) * HPW + (t % 3);
        const int flip = ((kk & 1) ^ ((toff >> 1) & 1)) & 1;
        const int base = (toff & 1) ? (flip ? bO1 : bO0) : (flip ? bE1 : bE0);
        const int imm = (kk * HPW + toff) * 128;
        return tr_pair(dX, base + imm, base + imm + 512);
      };
      af[0] = ldA(0);
#pragma unroll
      for (int i = 0; i < LA; ++i) bfr[i] = ldB(i);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int kk = i / 9, t = i % 9;
        if (i + LA < NI) bfr[(i + LA) & 3] = ldB(i + LA);
        if (t == 4 && kk + 1 < BM / 16) af[(kk + 1) & 1] = ldA(kk + 1);
        if constexpr (HPM) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(af[kk & 1]), "v"(bfr[i & 3]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(af[kk & 1]), "v"(bfr[i & 3]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if constexpr (SPLIT) {
    for (int tt = t_begin; tt < t_end; ++tt) {
      const int b = tt / tiles_img, ti = tt - b * tiles_img;
      const int r0 = (ti / tiles_w) * TH, c0 = (ti % tiles_w) * TW;
      __syncthreads();
      // ---- stage dY tile [BM pix][64 cout] and X halo [HP pix][64 cin]
      for (int i = tid; i < BM * 8; i += 256) {
        const int pix = i >> 3, ch = i & 7;
        const int r = r0 + (pix >> 4), c = c0 + (pix & 15);
        const bool ok = r < p.OH && c < p.OW;
        const int off = pix * 128 + sw_piece(pix, ch >> 1) * 32 + (ch & 1) * 16;
        stage_row<T>(DY + (((size_t)b * p.OH + r) * p.OW + c) * p.Cout + co0 + ch * 8, ok, ldsD, C::D_BYTES, off);
      }
      const int iy_base = r0 * S - p.pad, ix_base = c0 * S - p.pad;
      for (int i = tid; i < HP * 8; i += 256) {
        const int pix = i >> 3, ch = i & 7;
        const int hr = pix / HPW, hc = pix - hr * HPW;
        const int iy = iy_base + hr, ix = ix_base + hc;
        const bool ok = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        const int off = pix * 128 + sw_piece(pix, ch >> 1) * 32 + (ch & 1) * 16;
        stage_row<T>(X + (size_t)b * ximg + (size_t)iy * xrow + (size_t)ix * xpix + ci0 + ch * 8, ok, ldsX, C::X_BYTES, off);
      }
      __syncthreads();
      compute(ldsD, ldsX);
    }
  } else {
    // perf mode: LDS-DMA (global_load_lds_dwordx4) straight from HBM/L2 into the swizzled LDS image.  The LDS
    // destination of one wave-instruction is linear (base + lane*16 = 8 pixel rows), so the XOR swizzle is
    // applied to the per-lane SOURCE chunk instead; out-of-image pixels read a zero page.  Tile t+1 is issued
    // into the other buffer before the MFMAs of tile t and waited for (s_waitcnt vmcnt(0)) only after them, in
    // front of the one barrier per tile.  The DMA is inline asm: with the builtin the compiler orders every
    // ds_read behind an outstanding LDS-DMA ("may alias") and put that vmcnt(0) in FRONT of the MFMA block, i.e.
    // the two buffers never overlapped anything.
    typedef __attribute__((address_space(3))) unsigned char lds_uchar;
    const unsigned lds0 = (unsigned)(size_t)(lds_uchar*)smem;
    constexpr int BUF = C::D_BYTES + C::X_BYTES;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(p.zeros);
    // Half-pair mode: dz' = dh + dl 2^-11 (times the tensor's power-of-two scale, undone by the reduce kernel), x = xh +
    // xl 2^-11, so dW = sum dh xh + 2^-11 sum (dh xl + dl xh): the chunk's tiles are walked three times - (dh, xl), (dl, xh),
    // then the accumulators are multiplied by 2^-11, then (dh, xh) - as ONE stream of 3 * ntiles items through the same
    // two LDS buffers; a 64-channel block of a pixel record is [hi 64 | lo 64] fp16 (ph_common.h).
    const int nt = t_end > t_begin ? t_end - t_begin : 0;
    const bool hi1 = HPM && p.hp_hi_only;      // PH_PREC_FP16X1: ONE pass, (dz hi, x hi)
    const int nitems = (HPM && !hi1) ? 3 * nt : nt;
    const int dblk = HPM ? (co0 >> 6) * 128 : co0, xblk = HPM ? (ci0 >> 6) * 128 : ci0;
    auto issue = [&](int item, int buf) {
      int pass = 0, tt = t_begin + item;
      if constexpr (HPM) { if (!hi1) { pass = item / nt; tt = t_begin + (item - pass * nt); } else pass = 2; }
      const int dpl = (HPM && pass == 1) ? 64 : 0, xpl = (HPM && pass == 0) ? 64 : 0;     // lo plane of dz' / of x
      const int b = tt / tiles_img, ti = tt - b * tiles_img;
      const int r0 = (ti / tiles_w) * TH, c0 = (ti % tiles_w) * TW;
      unsigned char* dD = smem + buf * BUF;
      unsigned char* dX = dD + C::D_BYTES;
      for (int i0 = wave * 64; i0 < BM * 8; i0 += 256) {
        const int i = i0 + lane, pix = i >> 3, pos = i & 7;
        const int ch = sw_piece(pix, pos >> 1) * 2 + (pos & 1);          // source chunk of this LDS slot
        const int r = r0 + (pix >> 4), c = c0 + (pix & 15);
        const bool ok = r < p.OH && c < p.OW;
        const void* src = ok ? (const void*)(DY + (((size_t)b * p.OH + r) * p.OW + c) * dpix + dblk + dpl + ch * 8)
                             : (const void*)zero;
        lds_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(dD - smem) + i0 * 16));
      }
      const int iy_base = r0 * S - p.pad, ix_base = c0 * S - p.pad;
      for (int i0 = wave * 64; i0 < C::HPP * 8; i0 += 256) {
        const int i = i0 + lane, pix = i >> 3, pos = i & 7;
        const int ch = sw_piece(pix, pos >> 1) * 2 + (pos & 1);
        const int hr = pix / HPW, hc = pix - hr * HPW;
        const int iy = iy_base + hr, ix = ix_base + hc;
        const bool ok = pix < HP && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        const void* src = ok ? (const void*)(X + (size_t)b * ximg + (size_t)iy * xrow + (size_t)ix * xpix + xblk + xpl + ch * 8)
                             : (const void*)zero;
        lds_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(dX - smem) + i0 * 16));
      }
    };
    if (nitems > 0) issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int it = 0; it < nitems; ++it) {
#ifndef PH_ABL_WG_NODMA   // timing ablation only (garbage results): no operand traffic
      if (it + 1 < nitems) issue(it + 1, buf ^ 1);
#endif
      if constexpr (HPM) {
        if (!hi1 && it == 2 * nt) {      // the cross terms are complete: weight them before the leading products accumulate
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              if constexpr (PIPE) {      // one register at a time through ONE vector register (all 144 at once cost the second wave per SIMD)
                float tmp_;
                asm volatile("v_accvgpr_read_b32 %1, %0\n\ts_nop 4\n\tv_mul_f32 %1, %2, %1\n\ts_nop 4\n\tv_accvgpr_write_b32 %0, %1\n\ts_nop 4"
                             : "+a"(acc[t][q]), "=&v"(tmp_) : "v"(PH_HP_LO_INV));
              } else {
                acc[t][q] *= PH_HP_LO_INV;
              }
            }
        }
      }
      if constexpr (PIPE) compute_p(buf * PBUF);
      else compute(smem + buf * BUF, smem + buf * BUF + C::D_BYTES);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next item have landed
      __syncthreads();                                    // ... and everybody's; buffer buf is free again
      buf ^= 1;
    }
  }
  // ---- partial slab: acc[t][q] -> row (cout) = (q&3)+8*(q>>2)+4*khalf, col (cin) = lane&31
  float* slab = p.slab + (size_t)chunk * NT * p.Cout * p.Cin;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = co0 + cf * 32 + (q & 3) + 8 * (q >> 2) + 4 * khalf;
      const int col = ci0 + kf * 32 + (lane & 31);
#ifdef PH_ABL_WG_NOSLAB   // timing ablation only
      asm volatile("" ::"v"(acc[t][q]), "v"(row + col));
#else
      if constexpr (PIPE) {      // straight from the accumulation register (copied through vector registers the 144 values cost the
        float* dst = slab + ((size_t)t * p.Cout + row) * p.Cin + col;      // kernel its second wave per SIMD)
        asm volatile("global_store_dword %0, %1, off" : : "v"(dst), "a"(acc[t][q]) : "memory");
      } else {
        slab[((size_t)t * p.Cout + row) * p.Cin + col] = acc[t][q];
      }
#endif
      // (accumulators pinned to AGPRs: one tap's 16 values at a time through the vector registers - hoisted together the 144
      // reads cost the second workgroup of the CU its registers)
      if (PIPE && q == 15) __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename T, int S, int TH, int KS>
int launch_wg(const PhWgrad& p, hipStream_t st) {
  using C = WgCfg<T, S, TH, KS>;
  auto kern = wgrad_kernel<T, S, TH, KS>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess)
      return PH_ELAUNCH;
    attr_done = true;
  }
  dim3 grid((p.Cout / 64) * (p.Cin / 64), p.nchunks);
  void* tok = nullptr;
  if (ph_prof_on())
    ph_prof_begin2(PH_CLS_WGRAD, 2.0 * p.B * p.OH * p.OW * (double)p.Cout * p.KS * p.KS * p.Cin,
                   ((double)p.B * p.IH * p.IW * p.Cin + (double)p.B * p.OH * p.OW * p.Cout) * sizeof(T) +
                       (double)p.nchunks * p.KS * p.KS * p.Cout * p.Cin * 4.0, st, &tok);   // x + dy once, the fp32 slab
  hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, st, p);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

template <typename T>
int launch_wg_T(const PhWgrad& p, hipStream_t st) {
  if (p.S == 1 && p.KS == 3) return launch_wg<T, 1, 8, 3>(p, st);
  if (p.S == 2 && p.KS == 3) return launch_wg<T, 2, 4, 3>(p, st);
  if (p.S == 2 && p.KS == 1) return launch_wg<T, 2, 4, 1>(p, st);
  if (p.S == 1 && p.KS == 1) return launch_wg<T, 1, 8, 1>(p, st);
  return PH_EINVAL;
}

// slab[nchunks][NT][Cout][Cin] -> OIHW.  64 outputs x 4 chunk-lanes per block, 4 independent partial sums per
// thread (loads in flight), fixed summation order (bitwise reproducible)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           int nchunks, int NT, int Cout, int Cin,
                                                           const float* __restrict__ unscale) {
  const size_t n = (size_t)NT * Cout * Cin;
  const int e = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int c = cl;
    for (; c + 12 < nchunks; c += 16) {
      s0 += slab[(size_t)c * n + i];
      s1 += slab[(size_t)(c + 4) * n + i];
      s2 += slab[(size_t)(c + 8) * n + i];
      s3 += slab[(size_t)(c + 12) * n + i];
    }
    for (; c < nchunks; c += 4) s0 += slab[(size_t)c * n + i];
  }
  __shared__ float sh[4][64];
  sh[cl][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (cl == 0 && i < n) {
    float s = (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
    if (unscale) s *= unscale[1];      // (power of two: exact)
    const int ci = i % Cin;
    const int co = (i / Cin) % Cout;
    const int t = i / ((size_t)Cin * Cout);
    dw[((size_t)co * Cin + ci) * NT + t] = s;
  }
}

__global__ void pack_w_kernel(const float* __restrict__ w, bf16* __restrict__ planes, int O, int I, int NT,
                              int dgrad) {
  // w OIHW [O][I][NT].  fwd layout [tap][O][I]; dgrad layout [tap][I][O]
  const size_t n = (size_t)NT * O * I;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int t, o, ii;
  if (!dgrad) { ii = i % I; o = (i / I) % O; t = i / ((size_t)I * O); }
  else        { o = i % O; ii = (i / O) % I; t = i / ((size_t)I * O); }
  const float v = w[((size_t)o * I + ii) * NT + t];
  bf16 a, b, c;
  split3_bf16(v, a, b, c);
  planes[i] = a; planes[n + i] = b; planes[2 * n + i] = c;
}

__global__ void pack_w_stem_kernel(const float* __restrict__ w, bf16* __restrict__ planes) {
  // w [64][3][7][7] -> [kh 7][cout 64][kw 8 x ch 4], zero padded
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 7 * 64 * 32) return;
  const int k = i & 31, co = (i >> 5) & 63, kh = i >> 11;
  const int kw = k >> 2, ch = k & 3;
  float v = 0.f;
  if (kw < 7 && ch < 3) v = w[((co * 3 + ch) * 7 + kh) * 7 + kw];
  bf16 a, b, c;
  split3_bf16(v, a, b, c);
  planes[i] = a; planes[7 * 64 * 32 + i] = b; planes[2 * 7 * 64 * 32 + i] = c;
}

// all conv weights of one network in ONE launch: thread i walks the concatenated fwd-layout index space
__global__ void pack_all_kernel(PhPackAll t, bf16* __restrict__ packed, int nplanes) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.total) return;
  int u = 0;
#pragma unroll 1
  while (u + 1 < t.n && i >= t.start[u + 1]) ++u;
  const size_t e = i - t.start[u];
  const int O = t.O[u], I = t.I[u], NT = t.NT[u];
  const size_t n = (size_t)NT * O * I;
  const int ii = e % I, o = (e / I) % O, tp = e / ((size_t)I * O);
  const float v = t.w[u][((size_t)o * I + ii) * NT + tp];
  bf16 a, b, c;
  split3_bf16(v, a, b, c);
  const size_t fdst = t.dst_fwd[u] + e;                                     // [tap][O][I]
  const size_t ddst = t.dst_dg[u] + ((size_t)tp * I + ii) * O + o;          // [tap][I][O]
  packed[fdst] = a; packed[ddst] = a;
  if (nplanes == 3) {
    packed[fdst + n] = b; packed[fdst + 2 * n] = c;
    packed[ddst + n] = b; packed[ddst + 2 * n] = c;
  }
}

// The same, tiled through LDS: one workgroup per (conv, 32 output channels, 32 input channels), all taps.  The
// element-per-thread kernel above writes the dgrad layout [tap][I][O] as 2-byte words scattered with a stride of O
// elements (110 us for the 11 M weights of a ResNet-18, twice per step); here the OIHW rows are read as contiguous
// runs of 32*NT floats and both layouts leave as 64-byte runs.
struct PackTiles { int tstart[21]; };
// fragment-major order of a 64 x 64 (row, k) block for the register-window kernels (conv_tap5/6/7.hip): [k-step 2][N tile 4][lane 64][8]:
// row r = 4 li + n, k = 32 ks + 8 lg + j, lane = 16 lg + li
__device__ __forceinline__ size_t frag64_index(int r, int k) {
  return (size_t)(((k >> 5) * 4 + (r & 3)) * 512 + ((((k >> 3) & 3) << 4) + (r >> 2)) * 8 + (k & 7));
}
template <int NP>
__global__ __launch_bounds__(256) void pack_all_tiled_kernel(PhPackAll t, PackTiles pt, bf16* __restrict__ packed) {
  __shared__ bf16 sh[NP][32][32 * 9 + 2];
  int u = 0;
#pragma unroll 1
  while (u + 1 < t.n && (int)blockIdx.x >= pt.tstart[u + 1]) ++u;
  const int O = t.O[u], I = t.I[u], NT = t.NT[u];
  const int tile = (int)blockIdx.x - pt.tstart[u];
  const int itiles = I >> 5;
  const int o0 = (tile / itiles) << 5, i0 = (tile % itiles) << 5;
  const size_t n = (size_t)NT * O * I;
  const int run = 32 * NT;   // contiguous floats per output-channel row of the tile
  for (int idx = threadIdx.x; idx < 32 * run; idx += 256) {
    const int o = idx / run, r = idx - o * run;
    const float v = t.w[u][((size_t)(o0 + o) * I + i0) * NT + r];
    if constexpr (NP == 1) {
      sh[0][o][r] = (bf16)v;
    } else {
      bf16 a, b, c;
      split3_bf16(v, a, b, c);
      sh[0][o][r] = a; sh[1][o][r] = b; sh[2][o][r] = c;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 32 * run; idx += 256) {
    const int x = idx & 31, y = (idx >> 5) & 31, tp = idx >> 10;
    // forward layout [tap][O][I]: x = input channel (fastest), y = output channel
    const size_t fdst = t.dst_fwd[u] + ((size_t)tp * O + o0 + y) * I + i0 + x;
    // dgrad layout [tap][I][O]: x = output channel (fastest), y = input channel
    const size_t ddst = t.dst_dg[u] + ((size_t)tp * I + i0 + y) * O + o0 + x;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      packed[fdst + pl * n] = sh[pl][y][x * NT + tp];
      packed[ddst + pl * n] = sh[pl][x][y * NT + tp];
    }
    // perf mode, dense 3x3 stride-1 units with Cin = Cout >= 128 (ResNet layers 2-4: what conv_tap7.hip takes): a fragment-major
    // copy of both orientations in plane 1 of the unit's region (unused in this mode): [tap][rows / 64][K / 64] x 4096 elements
    if constexpr (NP == 1) {
      if (NT == 9 && O == 2 * I) {      // the stride-2 units, forward orientation (conv_tap6b.hip)
        const int o = o0 + y, i = i0 + x;
        packed[t.dst_fwd[u] + n + (size_t)tp * O * I + (size_t)((o >> 6) * (I >> 6) + (i >> 6)) * 4096 + frag64_index(o & 63, i & 63)] = sh[0][y][x * NT + tp];
      }
      if (NT == 9 && O == I && I >= 128) {
        const int o = o0 + y, i = i0 + x;      // forward: row o, k i
        packed[t.dst_fwd[u] + n + (size_t)tp * O * I + (size_t)((o >> 6) * (I >> 6) + (i >> 6)) * 4096 + frag64_index(o & 63, i & 63)] = sh[0][y][x * NT + tp];
        const int r = i0 + y, k = o0 + x;      // dgrad: row = input channel, k = output channel
        packed[t.dst_dg[u] + n + (size_t)tp * I * O + (size_t)((r >> 6) * (O >> 6) + (k >> 6)) * 4096 + frag64_index(r & 63, k & 63)] = sh[0][x][y * NT + tp];
      }
    }
  }
}

// PH_PREC_FP16X3 layout: per (tap, output row) the K direction holds, for every 64-channel slice, the three fp16 blocks
// [hi * 2^11 | lo | hi] of the weights' half-pair split (w ~= hi + lo * 2^-11): the K loop of a convolution multiplies them
// with the activation blocks (hi, hi, lo) and accumulates 2^11 * (x * w) in ONE accumulator set.  |w| < 32 (hi * 2^11 is fp16).
__device__ __forceinline__ size_t hp_k(int k) { return (size_t)(k >> 6) * 192 + (k & 63); }
__device__ __forceinline__ void hp_w3(float v, f16& b0, f16& b1, f16& b2) {
  hp_split(v, b2, b1);
  b0 = (f16)((float)b2 * PH_HP_LO);
}
// conv_tap5.hip's layout of a 64 x 64 x 9 slab set: per tap [block 3][k-step 2][N tile 4][lane 64][8]: row r = 4 li + n, k = 32 ks + 8 lg + j,
// lane = 16 lg + li -> element index inside the tap (block 0); blocks are 4096 elements apart.  frag5: bit 0 = conv_tap5.hip's
// switch, bit 1 = conv_tap6.hip's (host: ph_tap5_switch / ph_tap6_switch at pack time)
__device__ __forceinline__ size_t ph5_frag_index(int r, int k) { return frag64_index(r, k); }
template <bool ONE>
__global__ __launch_bounds__(256) void pack_all_tiled_hp_kernel(PhPackAll t, PackTiles pt, f16* __restrict__ packed, int dgrad_only, int frag5) {
  __shared__ float sh[32][32 * 9 + 1];
  int u = 0;
#pragma unroll 1
  while (u + 1 < t.n && (int)blockIdx.x >= pt.tstart[u + 1]) ++u;
  const int O = t.O[u], I = t.I[u], NT = t.NT[u];
  const int tile = (int)blockIdx.x - pt.tstart[u];
  const int itiles = I >> 5;
  const int o0 = (tile / itiles) << 5, i0 = (tile % itiles) << 5;
  const int run = 32 * NT;
  for (int idx = threadIdx.x; idx < 32 * run; idx += 256) {
    const int o = idx / run, r = idx - o * run;
    sh[o][r] = t.w[u][((size_t)(o0 + o) * I + i0) * NT + r];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 32 * run; idx += 256) {
    const int x = idx & 31, y = (idx >> 5) & 31, tp = idx >> 10;
    f16 b0, b1, b2;
    if (!ONE || dgrad_only == 0) {   // forward layout [tap][O][3 I]: x = input channel (fastest), y = output channel
      hp_w3(sh[y][x * NT + tp], b0, b1, b2);
      if ((frag5 & 1) && O == 64 && I == 64 && NT == 9) {
        f16* d = packed + t.dst_fwd[u] + (size_t)tp * 12288 + ph5_frag_index(o0 + y, i0 + x);
        d[0] = b0; d[4096] = b1; d[8192] = b2;
      } else if ((frag5 & 2) && O == 2 * I && NT == 9) {
        // conv_tap6.hip (the stride-2 convolutions, forward orientation only): [tap][O / 64][I / 64][block 3] x 4096 elements
        const int o = o0 + y, i = i0 + x;
        f16* d = packed + t.dst_fwd[u] + (size_t)tp * O * 3 * I + (size_t)(((o >> 6) * (I >> 6) + (i >> 6)) * 3) * 4096 + ph5_frag_index(o & 63, i & 63);
        d[0] = b0; d[4096] = b1; d[8192] = b2;
      } else {
        f16* d = packed + t.dst_fwd[u] + ((size_t)tp * O + o0 + y) * (3 * (size_t)I) + hp_k(i0 + x);
        d[0] = b0; d[64] = b1; d[128] = b2;
      }
    }
    if (!ONE || dgrad_only == 1) {   // dgrad layout [tap][I][3 O]: x = output channel (fastest), y = input channel
      hp_w3(sh[x][y * NT + tp], b0, b1, b2);
      if ((frag5 & 1) && O == 64 && I == 64 && NT == 9) {
        f16* d = packed + t.dst_dg[u] + (size_t)tp * 12288 + ph5_frag_index(i0 + y, o0 + x);
        d[0] = b0; d[4096] = b1; d[8192] = b2;
      } else {
        f16* d = packed + t.dst_dg[u] + ((size_t)tp * I + i0 + y) * (3 * (size_t)O) + hp_k(o0 + x);
        d[0] = b0; d[64] = b1; d[128] = b2;
      }
    }
  }
}

// stem, PH_PREC_FP16X3: three fp16 planes [3][kh 7][cout 64][kw 8 x ch 4] = w hi * 2^11, w lo, w hi (the layout of the
// three bf16 split planes): the kernel multiplies them with the activation planes (hi, hi, lo)
__global__ void pack_w_stem_hp_kernel(const float* __restrict__ w, f16* __restrict__ packed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 7 * 64 * 32) return;
  const int k = i & 31, co = (i >> 5) & 63, kh = i >> 11;
  const int kw = k >> 2, ch = k & 3;
  float v = 0.f;
  if (kw < 7 && ch < 3) v = w[((co * 3 + ch) * 7 + kh) * 7 + kw];
  f16 b0, b1, b2;
  hp_w3(v, b0, b1, b2);
  packed[i] = b0; packed[7 * 64 * 32 + i] = b1; packed[2 * 7 * 64 * 32 + i] = b2;
}

}  // namespace

int ph_pack_w_stem_hp_launch(const float* w, void* packed, hipStream_t st) {
  hipLaunchKernelGGL(pack_w_stem_hp_kernel, dim3((7 * 64 * 32 + 255) / 256), dim3(256), 0, st, w, (f16*)packed);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

// one convolution in the PH_PREC_FP16X3 layout (forward or dgrad orientation) at the start of `packed` (test hooks)
int ph_pack_w_hp_launch(const float* w, void* packed, int O, int I, int KS, int dgrad, hipStream_t st) {
  if ((O & 31) || (I & 31) || (KS != 1 && KS != 3)) return PH_EINVAL;
  PhPackAll t{};
  t.n = 1; t.w[0] = w; t.O[0] = O; t.I[0] = I; t.NT[0] = KS * KS; t.dst_fwd[0] = 0; t.dst_dg[0] = 0;
  t.start[0] = 0; t.start[1] = t.total = (size_t)KS * KS * O * I;
  PackTiles pt;
  pt.tstart[0] = 0; pt.tstart[1] = (O >> 5) * (I >> 5);
  hipLaunchKernelGGL(pack_all_tiled_hp_kernel<true>, dim3(pt.tstart[1]), dim3(256), 0, st, t, pt, (f16*)packed, dgrad ? 1 : 0, ph_tap5_switch(-1) | (ph_tap6_switch(-1) << 1));
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_pack_all_launch(const PhPackAll* t, void* packed, int nplanes, hipStream_t st) {
  bool tiled = t->n <= 20;
  PackTiles pt;
  pt.tstart[0] = 0;
  for (int u = 0; u < t->n && tiled; ++u) {
    if ((t->O[u] & 31) || (t->I[u] & 31) || t->NT[u] > 9) tiled = false;
    pt.tstart[u + 1] = pt.tstart[u] + (t->O[u] >> 5) * (t->I[u] >> 5);
  }
  if (nplanes == -3) {
    if (!tiled) return PH_EINVAL;
    hipLaunchKernelGGL(pack_all_tiled_hp_kernel<false>, dim3(pt.tstart[t->n]), dim3(256), 0, st, *t, pt, (f16*)packed, 0, ph_tap5_switch(-1) | (ph_tap6_switch(-1) << 1));
  } else if (tiled && nplanes == 1)
    hipLaunchKernelGGL(pack_all_tiled_kernel<1>, dim3(pt.tstart[t->n]), dim3(256), 0, st, *t, pt, (bf16*)packed);
  else if (tiled && nplanes == 3)
    hipLaunchKernelGGL(pack_all_tiled_kernel<3>, dim3(pt.tstart[t->n]), dim3(256), 0, st, *t, pt, (bf16*)packed);
  else
    hipLaunchKernelGGL(pack_all_kernel, dim3((unsigned)((t->total + 255) / 256)), dim3(256), 0, st, *t, (bf16*)packed,
                       nplanes);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_wgrad_tile_h(int S) { return S == 1 ? 8 : 4; }

int ph_wgrad_launch(const PhWgrad* p, int prec, hipStream_t st) {
  if (p->Cin % 64 || p->Cout % 64 || p->nchunks < 1) return PH_EINVAL;
  if (prec == PH_PREC_BF16) return launch_wg_T<bf16>(*p, st);
  if (prec == PH_PREC_FP16X3 || prec == PH_PREC_FP16X1) {
    PhWgrad q = *p;
    q.hp_hi_only = prec == PH_PREC_FP16X1;
    return launch_wg_T<hp16>(q, st);
  }
  if (PH_IS_SPLIT_PREC(prec)) {
    PhWgrad q = *p;
    q.prod6 = prec == PH_PREC_BF16X6;
    return launch_wg_T<float>(q, st);
  }
  return PH_EINVAL;
}

int ph_wgrad_reduce_launch(const float* slab, float* dw, int nchunks, int KS, int Cout, int Cin, const float* unscale,
                           hipStream_t st) {
  const size_t n = (size_t)KS * KS * Cout * Cin;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, slab, dw, nchunks,
                     KS * KS, Cout, Cin, unscale);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

int ph_pack_w_fwd_launch(const float* w, void* planes, int O, int I, int KS, hipStream_t st) {
  const size_t n = (size_t)KS * KS * O * I;
  hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, (bf16*)planes, O, I,
                     KS * KS, 0);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_pack_w_dgrad_launch(const float* w, void* planes, int O, int I, int KS, hipStream_t st) {
  const size_t n = (size_t)KS * KS * O * I;
  hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, (bf16*)planes, O, I,
                     KS * KS, 1);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
int ph_pack_w_stem_launch(const float* w, void* planes, hipStream_t st) {
  hipLaunchKernelGGL(pack_w_stem_kernel, dim3((7 * 64 * 32 + 255) / 256), dim3(256), 0, st, w, (bf16*)planes);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
