// Tap-convolution implicit GEMM, fifth kernel: the half-pair arithmetic (PH_PREC_FP16X3 forward / backward, PH_PREC_FP16X1
// backward) of the dense 3x3 stride-1 convolutions with Cin = Cout = 64 - ResNet layer 1 (reference resnets.py:58-74,193-198),
// 12 forward + 4 dgrad launches per distillation step, until round 6 on the first-generation kernel (conv_tap.hip
// <hp16,1,16,64,..>: 234 us per launch = 0.40 of the fp16 MFMA peak over three products, profiles/r06_kernel_stats_fp16x3.txt).
//
// What bounds a 64-channel layer on 16x16x32 fragments is the LDS read port: a wave tile of 8 x 4 fragments (128 pixels x 64
// channels) costs 12 ds_read_b128 per 32 MFMAs (0.75 of the port with four waves, conv_tap3.hip), and with only 64 output
// channels the four waves of a workgroup cannot share a weight tile - each needs all of it.  This kernel therefore
//   * stacks the four waves in the PIXEL direction (workgroup tile 32 rows x 16 columns x 64 channels) and keeps BOTH halo
//     planes of the tile resident in LDS: x hi (34 x 18 pixels x 128 B = 77 KiB) and x lo (77 KiB) - the three products of a
//     tile are the slices (x hi, w hi 2^11), (x hi, w lo), (x lo, w hi), so x hi is staged ONCE for two slices (conv_tap3.hip's
//     half-pair form stages it twice), x lo of the tile arrives while the two hi slices compute and x hi of the NEXT tile while
//     the lo slice computes;
//   * takes the weight fragments straight from global memory into registers (they are the same 24 KiB per tap for every tile:
//     L2 / L1 resident), two taps ahead in a rotating window of three register sets - no weight image in LDS (there is no room
//     beside 154 KiB of halos), 8 instead of 12 LDS reads per k-step (0.50 of the port), and NO per-tap barrier: the only
//     workgroup barriers are the two per tile that hand a halo buffer over;
//   * counts its own vmcnt: the halo LDS-DMAs and the weight loads are inline assembly, every tap ends with one s_waitcnt that
//     names how many younger operations may stay in flight (loads return in order).
// Same GEMM view, descriptor (PhTapConv) and epilogue semantics as conv_tap3.hip's half-pair form: fp32 16-byte stores, fused
// dgrad residual / mask (fp32 operands), per-workgroup BatchNorm partial sums (one row per persistent workgroup).
#include "ph_common.h"
#include <mutex>
#include <type_traits>
#include "ph_kernels.h"
#include "tap_common.h"
#ifndef PH5_DBG
#define PH5_DBG 0      // ablation builds (timing only): 2 = no halo pieces inside the slices, 4 = no weight loads inside the slices
#endif

namespace {

__device__ const u32x4 ph5_zero16[4] = {};
#ifdef PH_TAP_TRACE
__device__ unsigned long long ph_tap_trace[PH_TRACE_WGS * 12];
#endif

typedef __attribute__((address_space(3))) unsigned char lds_uchar;

__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(g) : "memory");
}

struct Tap5Cfg {
  static constexpr int NW = 4, NM = 8, NN = 4, NTAPS = 9;
  static constexpr int TH = NW * NM, TW = 16, BNT = 64;
  static constexpr int HPH = TH + 2, HPW = TW + 2;
  static constexpr int ROW_BYTES = (HPW / 2) * 256;                             // one halo row: 9 pixel pairs of 2 x 128 B
  static constexpr int A_BYTES = (HPH * ROW_BYTES + 1023) / 1024 * 1024;        // whole 1-KiB DMA pieces
  static constexpr int NHD = A_BYTES / 1024, NHE = (NHD + NW - 1) / NW;         // pieces per plane / per wave
  static constexpr int RED_OFF = 2 * A_BYTES + 512, DUMMY_OFF = RED_OFF + NW * 2 * BNT * 4;      // cross-wave sum scratch, dummy piece
  static constexpr int LDS_BYTES = DUMMY_OFF + 1024;
  static constexpr int NTH = NW * 64;
  static constexpr int WK = 192;                                                // packed weight row: [hi 2^11 | lo | hi] x 64
  static_assert(NHE == 20, "DMA schedule below: 20 pieces per wave and plane");
  static_assert((NM + 2) * ROW_BYTES < 65536, "ds_read immediate offsets");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// halo image: pixel (hr, hc), 16-byte chunk c of its 64 fp16 channels -> LDS byte offset inside a plane buffer (conv_tap3.hip's
// image: swizzle by the pixel column only, a fragment's address is linear in the halo row)
__device__ __forceinline__ int a5_off(int hr, int hc, int c) {
  return (Tap5Cfg::HPW / 2 * hr + (hc >> 1)) * 256 + ((hc & 1) << 7) + ((c ^ (((hc >> 1) & 3) << 1)) << 4);
}

// DMA pieces a slice kind issues in tap t (k-step 1, groups 0..3) and the first piece index.  Kinds: 0 / 1 / 2 = the three slices
// of a tile of the 3-product form (x hi . w hi 2^11 | x hi . w lo | x lo . w hi), 3 / 4 = the single slice of a tile of the hi-only
// form on buffer 0 / 1.  Kind 1 fetches x lo of THIS tile (buffer 1, free since the previous tile's lo slice), kinds 2, 3, 4 fetch
// x hi of the NEXT tile into the buffer the barrier of the slice before released - 4, 4, 3, 3, 3, 3 pieces in taps 0..5: a piece of
// tap t has landed at the wait that ends tap t + 2, the hand-over barrier sits in tap 8.  Kind 0 issues none (its issue slots take
// the piece mask of the next tile); with the lo plane's pieces in kind 0 instead, that slice ran 1.4 us longer.
__host__ __device__ constexpr int ph5_ndma(int kind, int t) {
  return kind == 0 ? 0 : (t < 2 ? 4 : (t < 6 ? 3 : 0));
}
__host__ __device__ constexpr int ph5_dma0(int kind, int t) {
  return t < 2 ? 4 * t : 8 + 3 * (t - 2);
}
__host__ __device__ constexpr int ph5_next_kind(int kind) { return kind == 0 ? 1 : kind == 1 ? 2 : kind == 2 ? 0 : kind == 3 ? 4 : 3; }
__host__ __device__ constexpr int ph5_abuf(int kind) { return (kind == 2 || kind == 4) ? 1 : 0; }
__host__ __device__ constexpr int ph5_wblk(int kind) { return kind == 0 ? 0 : kind == 1 ? 1 : 2; }

__device__ __forceinline__ void ph5_wait_vmcnt(int n) {
  switch (n) {
#define PH5_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    PH5_W(8) PH5_W(9) PH5_W(10) PH5_W(11) PH5_W(12) PH5_W(13) PH5_W(14) PH5_W(15) PH5_W(16) PH5_W(17) PH5_W(18) PH5_W(19) PH5_W(20)
#undef PH5_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// HI1: the hi planes' product alone (PH_PREC_FP16X1 backward): one slice per tile, buffers alternate tile by tile
template <bool HI1>
__global__ __launch_bounds__(256) void tapconv5_kernel(PhTapConv p) {
  using C = Tap5Cfg;
  constexpr int NM = C::NM, NN = C::NN, TH = C::TH, TW = C::TW, HPW = C::HPW, NTAPS = C::NTAPS, NTH = C::NTH, BNT = C::BNT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(lds_uchar*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;      // MFMA 16x16x32: row / column index, k group (A, B) or pixel group (C)
  const int tiles_w = (p.OWt + TW - 1) / TW;
  const int tiles_sp = tiles_w * ((p.OHt + TH - 1) / TH);
  const int total = tiles_sp * p.B;
  // strides of the input view in 2-byte elements (a half-pair pixel record is 2 x Cin fp16)
  const long pix_st = (p.in_pix_stride ? p.in_pix_stride : p.Cin) * 2;
  const long row_st = (p.in_row_stride ? p.in_row_stride : (long)p.IW * p.Cin) * 2;
  const long img_st = (p.in_img_stride ? p.in_img_stride : (long)p.IH * p.IW * p.Cin) * 2;

  // ---- tile list: linear tile id -> (spatial tile fastest, image), XCD-contiguous (as conv_tap3.hip)
  struct TileCtx { int r0, c0, b, iy_base, ix_base; const unsigned char* in; };
  const float rcp_sp = 1.0f / (float)tiles_sp, rcp_tw = 1.0f / (float)tiles_w;
  auto fdiv = [](int a, int d, float rcp) {
    int q = (int)((float)a * rcp);
    int r = a - q * d;
    if (r >= d) ++q;
    if (r < 0) --q;
    return q;
  };
  auto decode = [&](int t) __attribute__((always_inline)) -> TileCtx {
    TileCtx c;
    c.b = fdiv(t, tiles_sp, rcp_sp);
    const int tile = t - c.b * tiles_sp;
    const int trow = fdiv(tile, tiles_w, rcp_tw);
    c.r0 = trow * TH;
    c.c0 = (tile - trow * tiles_w) * TW;
    c.b = __builtin_amdgcn_readfirstlane(c.b);
    c.r0 = __builtin_amdgcn_readfirstlane(c.r0);
    c.c0 = __builtin_amdgcn_readfirstlane(c.c0);
    c.iy_base = c.r0 + p.iy0;
    c.ix_base = c.c0 + p.ix0;
    // byte address of halo pixel (0, 0), channel 0 of the hi plane (may lie outside the tensor: out-of-image lanes never use it)
    c.in = reinterpret_cast<const unsigned char*>(p.in) + ((long)c.b * img_st + (long)c.iy_base * row_st + (long)c.ix_base * pix_st) * 2;
    return c;
  };
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

  // ---- weights: tap slab t = wtap[0] + t * (wtap[1] - wtap[0]) (forward 0..8, dgrad 8..0).  The 64 -> 64 slabs are packed
  // FRAGMENT-MAJOR for this kernel (conv_wgrad.hip: pack_all_tiled_hp_kernel, ph5_frag_index below): per tap
  // [block 3][k-step 2][N tile 4][lane 64][8 fp16] - the fragment of N tile n / k-step ks (row 4 li + n, k = 32 ks + 8 lg ..) of
  // all 64 lanes is ONE contiguous KiB, so a load instruction touches 8 full cache lines.  (In the row-major layout [row][192]
  // a fragment is 16 rows x 64 B = 16 half lines: the same bytes took the vector L1 twice as long - a slice with weight loads
  // 9.5 us against 5.7 without; 6.2 us with contiguous fragments, round 6.)
  const long slab_bytes = (long)p.Cout * C::WK * 2;
  const unsigned char* w0 = reinterpret_cast<const unsigned char*>(p.w) + (long)p.wtap[0] * slab_bytes + 4096;
  const long wtap_step = (long)(p.wtap[1] - p.wtap[0]) * slab_bytes;
  const int voffB = lane * 16;
  auto w_base = [&](int blk, int tap) -> const unsigned char* {      // per-lane; + 4 KiB: the 8 fragments are immediates -4096 .. 3072
    return w0 + (long)tap * wtap_step + blk * 8192 + voffB;
  };

  // ---- per-lane halo DMA sources: piece h = wave + 4 e of a plane covers row pairs 4 h .. 4 h + 3; lane l fills slot l & 15 of
  // row pair rp = 4 h + (l >> 4): halo row rp / 9, column 2 (rp % 9) + (slot >> 3), chunk (slot & 7) ^ T(column)
  int h_off[C::NHE];
#pragma unroll
  for (int e = 0; e < C::NHE; ++e) {
    const int rp = (wave + 4 * e) * 4 + (lane >> 4), s = lane & 15;
    const int hr = rp / (HPW / 2), q = rp - hr * (HPW / 2);
    const int hc = 2 * q + (s >> 3), ch = (s & 7) ^ ((q & 3) << 1);
    h_off[e] = (int)(((long)hr * row_st + (long)hc * pix_st + ch * 8) * 2);
  }
  const unsigned char* zero_src = reinterpret_cast<const unsigned char*>(ph5_zero16);
  // Which of its 20 pieces lie inside the image is a property of the TILE (halo origin iy_base / ix_base), the same for both
  // planes: bit e of a per-lane mask, computed once per tile - x hi of tile k + 1 (fetched during tile k's lo slice) and x lo of
  // tile k + 1 (fetched one tile later) share it.  piece_bit(e) = this lane's source pixel of piece e is inside the image.
  auto piece_bit = [&](const int e, const int iy_base, const int ix_base) __attribute__((always_inline)) -> unsigned {
    int ln = lane;
    asm volatile("" : "+v"(ln));                       // (recomputed per piece instead of living in registers)
    const int h = wave + 4 * e;
    const int rp = h * 4 + (ln >> 4), s = ln & 15;
    const int hr = (int)(__umul24((unsigned)rp, 7282u) >> 16);      // rp / 9 for rp < 1024
    const int hc = 2 * (rp - hr * (HPW / 2)) + (s >> 3);
    const bool ok = (h < C::NHD) && (hr < C::HPH) && ((unsigned)(iy_base + hr) < (unsigned)p.IH) && ((unsigned)(ix_base + hc) < (unsigned)p.IW);
    return ok ? (1u << e) : 0u;
  };
  // one piece: e (literal after unrolling) of plane `plane_el` (element offset of the plane inside a pixel record: 0 = hi, 64 = lo)
  // of the tile at `tin` into buffer abuf.  Every wave issues every e (pieces past the image go to a dummy KiB): the vmcnt
  // bookkeeping is the same in all waves.
  auto dma_piece = [&](const int e, const unsigned char* tin, const unsigned mask, const int plane_el, const int abuf) __attribute__((always_inline)) {
    const int h = wave + 4 * e;
    int off = h_off[e];
    asm volatile("" : "+v"(off));                      // (or the 64-bit extension of all 20 offsets is hoisted out of the tile loop: 40 registers)
    const unsigned char* src = ((mask >> e) & 1u) ? tin + plane_el * 2 + off : zero_src;
    const unsigned dst = lds0 + (h < C::NHD ? abuf * C::A_BYTES + h * 1024 : C::DUMMY_OFF);
    lds_dma16(src, dst);
  };

  // ---- per-lane fragment addressing.  A: base of (halo row 8 wave, column li + dx, chunk lg) for dx = 0, 1, 2 in both buffers
  // and its k-step-1 twin (chunk lg + 4 = address ^ 64); M tile m and the tap row dy are immediate offsets.
  int ab0[2][3], ab1[2][3];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      ab0[bf][dx] = bf * C::A_BYTES + a5_off(wave * NM, li + dx, lg);
      ab1[bf][dx] = ab0[bf][dx] ^ 64;
    }

  f32x4 acc[NM][NN];
  // BatchNorm partial sums in registers across the tiles of this workgroup: a lane owns channels 4 li + n
  float s1[NN], s2[NN];
#pragma unroll
  for (int n = 0; n < NN; ++n) { s1[n] = 0.f; s2[n] = 0.f; }

  // ---- epilogue of one tile.  Accumulator register q of tile (m, n) is pixel (row 8 wave + m, column 4 lg + q), channel 4 li + n:
  // the four N tiles give four consecutive channels = one 16-byte store per (m, q), 256 contiguous bytes per pixel.
  const float osc = (HI1 ? 1.f : PH_HP_LO_INV) * (p.in_unscale ? p.in_unscale[1] : 1.f);
  // Stores and residual loads go through buffer resources of the tile's IMAGE (conv_tap4.hip): a lane outside the output gets an
  // offset past the resource - the hardware drops its store / returns 0 for its load; 32-bit byte offsets, no 64-bit address per
  // piece, and every tile issues exactly 32 stores per wave whatever its shape.
  constexpr unsigned OOB = 0x7ffffff0u;
  constexpr int RSRC_FLAGS = 0x00020000;      // raw buffer, 32-bit offsets (gfx90a / gfx94x / gfx950 data format word)
  const int img_bytes = p.OH * p.OW * BNT * 4;
  auto epilogue = [&](const TileCtx& tc, auto fullc, auto rmc) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(fullc)::value;
    constexpr int RM = decltype(rmc)::value;
    const size_t img = (size_t)tc.b * p.OH * p.OW * BNT;
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(p.out) + img, 0, img_bytes, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(reinterpret_cast<const float*>(p.res_g)) + img, 0, RM > 0 ? img_bytes : 0, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(reinterpret_cast<const float*>(p.res_a)) + img, 0, RM > 1 ? img_bytes : 0, RSRC_FLAGS);
    // byte offset of this lane's piece (0, 0) inside the image; rows advance by rowstep, columns by 256 B
    const unsigned o00 = 4u * ((unsigned)((tc.r0 + wave * NM) * p.OW + tc.c0 + 4 * lg) * (unsigned)BNT + 4u * (unsigned)li);
    const unsigned rowstep = 4u * (unsigned)(p.OW * BNT);
    const int rlim = p.OHt - (tc.r0 + wave * NM), clim = p.OWt - (tc.c0 + 4 * lg);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const unsigned orow = o00 + (unsigned)m * rowstep;
      int off[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) off[q] = (int)((FULL || (m < rlim && q < clim)) ? orow + (unsigned)q * 256u : OOB);
      // the residual operands of the whole tile row first: 8 loads in flight per wave (one pixel at a time the round trips of a
      // tile were 32 in a row - 8 KiB in flight per compute unit, a quarter of what the HBM-bound masked-residual dgrad needs)
      f32x4 g[4], a[4];
      if constexpr (RM > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          g[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_g, off[q], 0, 0));
          if constexpr (RM > 1) a[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_a, off[q], 0, 0));
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool mine = FULL || (m < rlim && q < clim);
        f32x4 v;
#pragma unroll
        for (int n = 0; n < NN; ++n) {
          // (read through inline assembly: the operand stays an accumulation register up to this point - with plain reads the
          // allocator, six epilogue variants downstream of the last tap, copies all 128 values into vector registers right behind
          // the MFMAs that produce them)
          float x;
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[m][n][q]));
          v[n] = x * osc;
          if constexpr (!FULL) v[n] = mine ? v[n] : 0.f;
          s1[n] += v[n];
          s2[n] = __builtin_fmaf(v[n], v[n], s2[n]);
        }
        if constexpr (RM > 1) {
#pragma unroll
          for (int n = 0; n < NN; ++n) v[n] += a[q][n] > 0.f ? g[q][n] : 0.f;
        } else if constexpr (RM > 0) {
          v += g[q];
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, off[q], 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);      // one tile row at a time
    }
  };
  const int rmode = p.res_g ? (p.res_a ? 2 : 1) : 0;
  auto epilogue_any = [&](const TileCtx& tc) __attribute__((always_inline)) {
    const bool full = (tc.r0 + TH <= p.OHt) && (tc.c0 + TW <= p.OWt);
    auto with_full = [&](auto fullc) __attribute__((always_inline)) {
      if (rmode == 0) epilogue(tc, fullc, std::integral_constant<int, 0>{});
      else if (rmode == 1) epilogue(tc, fullc, std::integral_constant<int, 1>{});
      else epilogue(tc, fullc, std::integral_constant<int, 2>{});
    };
    if (full) with_full(std::true_type{});
    else with_full(std::false_type{});
  };

  // ---- the tap stream
  TileCtx tcur = decode(tile_id(0));
  int tn = tile_id(1);
  bool nvalid = tn >= 0;
  TileCtx tnext = tcur;
  if (nvalid) tnext = decode(tn);

  // Fragment registers: A ring of 4 (M tile m of a k-step is read three MFMA groups ahead), B three sets of 2 k-steps x 4 N tiles:
  // tap g multiplies set g % 3, set (g + 1) % 3 has landed when tap g ends, set (g + 2) % 3 is being loaded.  The B sets live in
  // ACCUMULATION registers (the loads write them, the MFMAs read them as their B operand): 96 + 128 of the 256, and the vector
  // file stays far enough from its limit that the allocator never copies a set whose data is still in flight
  // (profiles/scripts/check_tap5_asm.py checks the assembly for exactly that).
  u32x4 fa[4], fb[3][2][NN];
  // (FIRST: the first k-step of a tile writes its accumulators - C operand 0 - instead of accumulating: no zeroing pass, and the
  // old tile's accumulators are dead before the new ones are born, which the full accumulation-register file needs)
#define PH5_MM(M, N, AI, SET, KS, FIRST)                                                                                             \
  do {                                                                                                                               \
    if (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(acc[M][N]) : "v"(fa[AI]), "a"(fb[SET][KS][N]));         \
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[M][N]) : "v"(fa[AI]), "a"(fb[SET][KS][N]));               \
  } while (0)
#define PH5_LD(ADDR, IMM) (*reinterpret_cast<const u32x4*>(smem + (ADDR) + (IMM)))
#define PH5_SB() __builtin_amdgcn_sched_barrier(0)
#define PH5_LDA(AB, AOFF, MT) PH5_LD(AB, (AOFF) + (MT) * C::ROW_BYTES)
  // weight fragment (KS, N) of the tap at wave-uniform base WB -> register set SET (literal byte offset (4 KS + N) KiB - 4 KiB)
  // (WB is a per-lane 64-bit address - the lane's 16 bytes of fragment (0, 0) - made by ONE vector add per tap; the loads carry nothing
  // but an immediate.  The first version passed a scalar base: a base restored from a spill lane by v_readlane right in front of the
  // inline assembly needs 5 wait states before a VMEM instruction reads it, the hazard recognizer does not look inside inline
  // assembly, and the first run loaded through a stale high word.)
#define PH5_BLD(SET, KS, N, WB, OFF) \
  asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=a"(fb[SET][KS][N]) : "v"(WB) : "memory")
#define PH5_BLD_I(SET, I, WB)                                        \
  do {                                                               \
    if (PH5_DBG & 4) break;                                          \
    if ((I) == 0) PH5_BLD(SET, 0, 0, WB, -4096);                     \
    else if ((I) == 1) PH5_BLD(SET, 0, 1, WB, -3072);                \
    else if ((I) == 2) PH5_BLD(SET, 0, 2, WB, -2048);                \
    else if ((I) == 3) PH5_BLD(SET, 0, 3, WB, -1024);                \
    else if ((I) == 4) PH5_BLD(SET, 1, 0, WB, 0);                    \
    else if ((I) == 5) PH5_BLD(SET, 1, 1, WB, 1024);                 \
    else if ((I) == 6) PH5_BLD(SET, 1, 2, WB, 2048);                 \
    else PH5_BLD(SET, 1, 3, WB, 3072);                               \
  } while (0)
  // one group: 4 MFMAs of M tile M with A ring slot M & 3 on B (SET, KS); RA = this group's A read, X1 / X2 = other issue slots
#define PH5_GROUP(M, SET, KS, RA, X1, X2)                        \
  PH5_MM(M, 0, (M) & 3, SET, KS, first && (KS) == 0); RA; PH5_SB();  \
  PH5_MM(M, 1, (M) & 3, SET, KS, first && (KS) == 0); X1; PH5_SB();  \
  PH5_MM(M, 2, (M) & 3, SET, KS, first && (KS) == 0); PH5_SB();      \
  PH5_MM(M, 3, (M) & 3, SET, KS, first && (KS) == 0); X2; PH5_SB()
#define PH5_NOP ((void)0)

  constexpr int KIND0 = HI1 ? 3 : 0;
  unsigned mask_cur = 0, mask_next = 0;      // piece masks of tcur / of tnext
#ifndef PH5_TRACE_KIND
  PH_TRACE(0);
#endif
  {  // prologue: x hi of the first tile, the weights of the first two taps
#pragma unroll
    for (int e = 0; e < C::NHE; ++e) mask_cur |= piece_bit(e, tcur.iy_base, tcur.ix_base);
#pragma unroll
    for (int e = 0; e < C::NHE; ++e) dma_piece(e, tcur.in, mask_cur, 0, 0);
    const unsigned char *wb0 = w_base(ph5_wblk(KIND0), 0), *wb1 = w_base(ph5_wblk(KIND0), 1);
    PH5_BLD(0, 0, 0, wb0, -4096); PH5_BLD(0, 0, 1, wb0, -3072); PH5_BLD(0, 0, 2, wb0, -2048); PH5_BLD(0, 0, 3, wb0, -1024);
    PH5_BLD(0, 1, 0, wb0, 0); PH5_BLD(0, 1, 1, wb0, 1024); PH5_BLD(0, 1, 2, wb0, 2048); PH5_BLD(0, 1, 3, wb0, 3072);
    PH5_BLD(1, 0, 0, wb1, -4096); PH5_BLD(1, 0, 1, wb1, -3072); PH5_BLD(1, 0, 2, wb1, -2048); PH5_BLD(1, 0, 3, wb1, -1024);
    PH5_BLD(1, 1, 0, wb1, 0); PH5_BLD(1, 1, 1, wb1, 1024); PH5_BLD(1, 1, 2, wb1, 2048); PH5_BLD(1, 1, 3, wb1, 3072);
  }
  if constexpr (HI1) {
#pragma unroll
    for (int e = 0; e < C::NHE; ++e) mask_next |= piece_bit(e, tnext.iy_base, tnext.ix_base);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifndef PH5_TRACE_KIND
  PH_TRACE(1);
#endif
  fa[0] = PH5_LDA(ab0[0][0], 0, 0);
  fa[1] = PH5_LDA(ab0[0][0], 0, 1);
  fa[2] = PH5_LDA(ab0[0][0], 0, 2);

  // One slice = 9 taps x 2 k-steps x 8 groups.  KIND is a literal at every call site: after inlining + unrolling every register
  // array index, immediate offset and wait count below is a constant.
  int trace_k = 0;      // (debug builds: tile counter for the tracer)
  auto slice_body = [&](const int KIND) __attribute__((always_inline)) {
    const int NEXT = ph5_next_kind(KIND);
    const int BUFC = ph5_abuf(KIND), BUFN = ph5_abuf(NEXT);
    // what this slice's DMAs fetch: kind 0 / 1 the lo plane of the current tile (buffer 1), kinds 2.. the hi plane of the next tile
    const bool lo_cur = KIND == 1;
    // (scalars, not a reference chosen between the two contexts: a select of their addresses keeps every captured array in memory)
    const bool dnext = !lo_cur && nvalid;
    const unsigned char* d_in = dnext ? tnext.in : tcur.in;
    const unsigned d_mask = dnext ? mask_next : mask_cur;
    const int dplane = lo_cur ? 64 : 0;
    const int dbuf = lo_cur ? 1 : (KIND == 2 ? 0 : (KIND == 3 ? 1 : 0));
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      const int dy = t / 3, dx = t % 3;
      const int tnx = (t + 1) % NTAPS, dyn = tnx / 3, dxn = tnx % 3;
      const int aoff = dy * C::ROW_BYTES, aoffn = dyn * C::ROW_BYTES;
      const int bufn = (t + 1 == NTAPS) ? BUFN : BUFC;
      const int set = t % 3, setl = (t + 2) % 3;
      const bool first = t == 0 && (KIND == 0 || KIND >= 3);      // first tap of a tile
      // weights of stream tap g + 2
      const unsigned char* wb = (t + 2 < NTAPS) ? w_base(ph5_wblk(KIND), t + 2) : w_base(ph5_wblk(NEXT), t + 2 - NTAPS);
      const int nd = ph5_ndma(KIND, t), e0 = ph5_dma0(KIND, t);
      const int ndp = t > 0 ? ph5_ndma(KIND, t - 1) : 0;      // (no kind issues pieces in its last taps)
#if PH5_DBG & 2
#define PH5_DMA(I) ((void)0)
#else
#define PH5_DMA(I) do { if ((I) < nd) dma_piece(e0 + (I), d_in, d_mask, dplane, dbuf); } while (0)
#endif
      // ---- k-step 0 (chunks lg): A tile m + 3 is read by group m (tiles 3..7 of this k-step, then 0..2 of k-step 1); one weight
      // fragment of tap g + 2 per group
      PH5_GROUP(0, set, 0, fa[3] = PH5_LDA(ab0[BUFC][dx], aoff, 3), PH5_BLD_I(setl, 0, wb), PH5_NOP);
      PH5_GROUP(1, set, 0, fa[0] = PH5_LDA(ab0[BUFC][dx], aoff, 4), PH5_BLD_I(setl, 1, wb), PH5_NOP);
      PH5_GROUP(2, set, 0, fa[1] = PH5_LDA(ab0[BUFC][dx], aoff, 5), PH5_BLD_I(setl, 2, wb), PH5_NOP);
      PH5_GROUP(3, set, 0, fa[2] = PH5_LDA(ab0[BUFC][dx], aoff, 6), PH5_BLD_I(setl, 3, wb), PH5_NOP);
      PH5_GROUP(4, set, 0, fa[3] = PH5_LDA(ab0[BUFC][dx], aoff, 7), PH5_BLD_I(setl, 4, wb), PH5_NOP);
      PH5_GROUP(5, set, 0, fa[0] = PH5_LDA(ab1[BUFC][dx], aoff, 0), PH5_BLD_I(setl, 5, wb), PH5_NOP);
      PH5_GROUP(6, set, 0, fa[1] = PH5_LDA(ab1[BUFC][dx], aoff, 1), PH5_BLD_I(setl, 6, wb), PH5_NOP);
      PH5_GROUP(7, set, 0, fa[2] = PH5_LDA(ab1[BUFC][dx], aoff, 2), PH5_BLD_I(setl, 7, wb), PH5_NOP);
      // ---- k-step 1 (chunks lg + 4): A tiles 3..7, then tiles 0..2 of the NEXT tap's k-step 0; the halo pieces of this tap
      PH5_GROUP(0, set, 1, fa[3] = PH5_LDA(ab1[BUFC][dx], aoff, 3), PH5_NOP, PH5_DMA(0));
      PH5_GROUP(1, set, 1, fa[0] = PH5_LDA(ab1[BUFC][dx], aoff, 4), PH5_NOP, PH5_DMA(1));
      PH5_GROUP(2, set, 1, fa[1] = PH5_LDA(ab1[BUFC][dx], aoff, 5), PH5_NOP, PH5_DMA(2));
      PH5_GROUP(3, set, 1, fa[2] = PH5_LDA(ab1[BUFC][dx], aoff, 6), PH5_NOP, PH5_DMA(3));
      PH5_GROUP(4, set, 1, fa[3] = PH5_LDA(ab1[BUFC][dx], aoff, 7), PH5_NOP, PH5_NOP);
      // (kind 0 has no DMA: its issue slots take the piece mask of the NEXT tile, two pieces per tap + the last two)
      if (KIND == 0) {
        if (t == 0) mask_next = 0;
        mask_next |= piece_bit(2 * t, tnext.iy_base, tnext.ix_base) | piece_bit(2 * t + 1, tnext.iy_base, tnext.ix_base);
        if (t + 1 == NTAPS) mask_next |= piece_bit(18, tnext.iy_base, tnext.ix_base) | piece_bit(19, tnext.iy_base, tnext.ix_base);
      }
      // Hand-over (last tap of kinds 1..4): this wave has issued its last read of the current buffer, its own pieces of the buffer the
      // next slice reads landed at earlier waits (all issued by tap 5).  After the barrier the next slice's DMAs may overwrite the
      // current buffer and the reads below may touch the next one.
      if (t + 1 == NTAPS && KIND != 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
      }
      PH5_GROUP(5, set, 1, fa[0] = PH5_LDA(ab0[bufn][dxn], aoffn, 0), PH5_NOP, PH5_NOP);
      PH5_GROUP(6, set, 1, fa[1] = PH5_LDA(ab0[bufn][dxn], aoffn, 1), PH5_NOP, PH5_NOP);
      PH5_GROUP(7, set, 1, fa[2] = PH5_LDA(ab0[bufn][dxn], aoffn, 2), PH5_NOP, PH5_NOP);
      // ---- tap end: the weights of tap g + 1 (loaded during tap g - 1) must be in their registers.  Younger than them: the previous
      // tap's pieces, this tap's 8 weight fragments, this tap's pieces.
      // (first tap of a tile: the wait in front of the epilogue already covered the weights of tap g + 1, and nothing this tap could wait
      // for is older than the epilogue's 32 stores - a counted wait here would stall until they have drained)
      if (!first) ph5_wait_vmcnt(ndp + 8 + nd);
#if defined(PH_TAP_TRACE) && defined(PH5_TRACE_KIND)
      if (KIND == PH5_TRACE_KIND && trace_k == 1) PH_TRACE_ACC(t + 1, wall_clock64());      // per-tap stamps of one slice of the second tile
#endif
#undef PH5_DMA
    }
  };

  auto advance = [&](int k) __attribute__((always_inline)) {
    tcur = tnext;
    mask_cur = mask_next;
    tn = tile_id(k + 2);
    nvalid = tn >= 0;
    if (nvalid) tnext = decode(tn);
    if constexpr (HI1) {      // (no hi.lo slice to hide it in)
      mask_next = 0;
#pragma unroll
      for (int e = 0; e < C::NHE; ++e) mask_next |= piece_bit(e, tnext.iy_base, tnext.ix_base);
    }
  };
  for (int k = 0;;) {
    trace_k = k;
    if constexpr (!HI1) {
#ifdef PH5_TRACE_KIND      // (debug: per-tap stamps of one slice of the second tile, slot 0 = its start)
      if (k == 1 && PH5_TRACE_KIND == 0) PH_TRACE(0);
      slice_body(0);
      if (k == 1 && PH5_TRACE_KIND == 1) PH_TRACE(0);
      slice_body(1);
      if (k == 1 && PH5_TRACE_KIND == 2) PH_TRACE(0);
      slice_body(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      epilogue_any(tcur);
#else
      slice_body(0);
      if (k == 0) PH_TRACE(2); else if (k == 1) PH_TRACE(6);
      slice_body(1);
      if (k == 0) PH_TRACE(3); else if (k == 1) PH_TRACE(7);
      slice_body(2);
      if (k == 0) PH_TRACE(4); else if (k == 1) PH_TRACE(8);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // weights of the next tile's first two taps (see the first tap's end)
      epilogue_any(tcur);
      if (k == 0) PH_TRACE(5); else if (k == 1) PH_TRACE(9);
#endif
      if (!nvalid) break;
      advance(k);
      ++k;
    } else {
      slice_body(3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // weights of the next tile's first two taps (see the first tap's end)
      epilogue_any(tcur);
      if (!nvalid) break;
      advance(k);
      ++k;
      slice_body(4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // weights of the next tile's first two taps (see the first tap's end)
      epilogue_any(tcur);
      if (!nvalid) break;
      advance(k);
      ++k;
    }
  }
#ifndef PH5_TRACE_KIND
  PH_TRACE(10);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // refills issued past the end of the stream must not outlive the workgroup's LDS
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem + C::RED_OFF);      // [NW][2][BNT]
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      float a1 = s1[n], a2 = s2[n];
      a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64);
      a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
      if (lg == 0) {
        red[(wave * 2 + 0) * BNT + 4 * li + n] = a1;
        red[(wave * 2 + 1) * BNT + 4 * li + n] = a2;
      }
    }
    __syncthreads();
    if (tid < 2 * BNT) {
      const int which = tid / BNT, n = tid % BNT;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < C::NW; ++w) v += red[(w * 2 + which) * BNT + n];
      p.stats[((size_t)blockIdx.x * 2 + which) * BNT + n] = v;
    }
  }
}

template <bool HI1>
int launch5(const PhTapConv& p, hipStream_t st) {
  using C = Tap5Cfg;
  auto kern = tapconv5_kernel<HI1>;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [&] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  });
  if (attr_rc != hipSuccess) return PH_ELAUNCH;
  const int total = cdiv(p.OHt, C::TH) * cdiv(p.OWt, C::TW) * p.B;
  const int resident = ph_num_cus();
  dim3 grid(total < resident ? total : resident);
  void* tok = nullptr;
  if (ph_prof_on())
    ph_prof_begin2(PH_CLS_TAPCONV2_RES, 2.0 * p.B * p.OHt * p.OWt * (double)p.Cout * p.ntaps * p.Cin, ph_tapconv_bytes(p, 1, 4), st, &tok);
  hipLaunchKernelGGL(kern, grid, dim3(C::NTH), C::LDS_BYTES, st, p);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

}  // namespace

// PH_TAP5=0 in the environment / ph_debug_set_tap5(0) keeps the first-generation kernel (same-box A/B)
int ph_tap5_switch(int set) {
  static int on = [] { const char* e = getenv("PH_TAP5"); return (e && e[0] == '0') ? 0 : 1; }();
  if (set >= 0) on = set ? 1 : 0;
  return on;
}
extern "C" int ph_debug_set_tap5(int on) { return ph_tap5_switch(on ? 1 : 0); }
#ifdef PH_TAP_TRACE
extern "C" int ph_debug_tap5_trace(unsigned long long* host_out, int nwg) {
  if (nwg > PH_TRACE_WGS) nwg = PH_TRACE_WGS;
  if (hipDeviceSynchronize() != hipSuccess) return PH_ELAUNCH;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ph_tap_trace), (size_t)nwg * 12 * sizeof(unsigned long long)) == hipSuccess ? PH_OK : PH_ELAUNCH;
}
#endif

// eligible: dense 3x3 stride-1 over the whole map, Cin = Cout = 64, no fused extras
bool ph_tapconv5_eligible(const PhTapConv* p) {
  if (p->ntaps != 9 || p->Cin != 64 || p->Cout != 64) return false;
  const int step = p->wtap[1] - p->wtap[0];
  for (int k = 0; k < 9; ++k)
    if (p->wtap[k] != p->wtap[0] + k * step || p->dy[k] != k / 3 || p->dx[k] != k % 3) return false;
  return !p->m_groups && !p->ncls && !p->in_scale && !p->bst_y && p->os == 1 && p->oa_h == 0 && p->oa_w == 0 && p->OHt == p->OH &&
         p->OWt == p->OW && p->iy0 == -1 && p->ix0 == -1 && p->IH == p->OH && p->IW == p->OW && p->OH >= 1 && p->OW >= 1;
}

int ph_tapconv5_stat_parts(const PhTapConv* p) {
  const int total = cdiv(p->OHt, Tap5Cfg::TH) * cdiv(p->OWt, Tap5Cfg::TW) * p->B;
  const int resident = ph_num_cus();
  return total < resident ? total : resident;
}

// p->hp_hi_only selects the hi-only form (PH_PREC_FP16X1)
int ph_tapconv5_launch(const PhTapConv* p, hipStream_t st) {
  if (!ph_tapconv5_eligible(p)) return PH_EINVAL;
  return p->hp_hi_only ? launch5<true>(*p, st) : launch5<false>(*p, st);
}
