// Tap-convolution implicit GEMM, seventh kernel: the perf-mode (bf16) dense 3x3 stride-1 convolutions with Cin = Cout >= 128 -
// ResNet layers 2-4 (reference resnets.py:58-74), forward (+ BatchNorm partial sums) and stride-1 dgrad (+ fused residual / mask):
// the launches of conv_tap3.hip's plain form, the dominant kernel of the step (32 per distillation step).
//
// conv_tap3.hip streams weights through an LDS ring with one workgroup barrier per tap, and its end-of-tap vmcnt wait forces every
// halo piece to land within ~1.5 taps of its issue (768 cycles): loads return in order, and the next tap's weight pieces are
// queued behind them.  It sits at 0.48 of the bf16 MFMA peak, its half-pair form - three times the MFMA work per halo byte - at 0.61.
// This kernel is the same GEMM (same tile, same wave grid, same fragment mapping, same summation order: its outputs are bitwise
// conv_tap3.hip's) on conv_tap6.hip's machinery:
//   * weight fragments from global memory straight into registers, FOUR taps ahead, in a window of nine k-step sets (six in
//     accumulation registers, three in vector registers), from a fragment-major copy of the packed weights ([tap][Cout / 64][Cin / 64]
//     x 8 KiB, every load instruction one contiguous KiB: plane 1 of the unit's packed region, which perf mode does not otherwise use);
//   * the halo image of the NEXT 64-channel slice by LDS-DMA in the first four taps of the current one; a piece may stay in flight for
//     4.5 taps, the hand-over barrier in the slice's last tap is the only workgroup barrier: one per 9 taps instead of nine;
//   * hand-counted vmcnt (profiles/scripts/check_tap5_asm.py checks the assembly).
// Workgroup: 4 waves as 2 (pixel rows 0-7 / 8-15) x 2 (channels 0-63 / 64-127); tile 16 x 16 x 128; wave tile 8 x 4 fragments of
// v_mfma_f32_16x16x32_bf16; 8-byte stores through a buffer resource; BatchNorm partial sums: conv_tap3.hip's row per workgroup, bitwise.
#include "ph_common.h"
#include <mutex>
#include <type_traits>
#include "ph_kernels.h"
#include "tap_common.h"
#ifndef PH7_DBG
#define PH7_DBG 0      // ablation builds (timing only): 2 = no image pieces inside the slices, 4 = no weight loads inside the slices, 8 = no stores
#endif

namespace {

__device__ const u32x4 ph7_zero16[4] = {};

typedef __attribute__((address_space(3))) unsigned char lds_uchar;

__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(g) : "memory");
}

struct Tap7Cfg {
  static constexpr int NW = 4, WM = 2, WN = 2, NM = 8, NN = 4, NTAPS = 9, NBUF = 2;
  static constexpr int TH = WM * NM, TW = 16, BNT = WN * 64;
  static constexpr int HPH = TH + 2, HPW = TW + 2;
  static constexpr int ROW_BYTES = (HPW / 2) * 256;
  static constexpr int A_BYTES = (HPH * ROW_BYTES + 1023) / 1024 * 1024;        // whole 1-KiB DMA pieces
  static constexpr int NHD = A_BYTES / 1024, NHE = (NHD + NW - 1) / NW;         // pieces per image / per wave
  static constexpr int LDS_BYTES = NBUF * A_BYTES;
  static constexpr int NTH = NW * 64;
  static_assert(NHE == 11, "DMA schedule below: 11 pieces per wave and image");
  static_assert((NM + 2) * ROW_BYTES < 65536, "ds_read immediate offsets");
  // pieces issued in tap t (second k-step, groups 0..2) and the first piece index: the image of the NEXT slice in taps 0..3
  static constexpr int ND[9] = {3, 3, 3, 2, 0, 0, 0, 0, 0};
  static constexpr int E0[9] = {0, 3, 6, 9, 0, 0, 0, 0, 0};
  // vmcnt allowances.  Weight fragments: 4 loads per k-step, 8 k-steps ahead; at the end of k-step j the fragments of k-step j + 1
  // (issued at k-step j - 7) must have landed, everything issued after them may stay in flight: 7 x 4 loads + the pieces of the taps
  // since.  HWAIT: at the hand-over point of tap 8 the pieces of the next image (last issued in tap 3) must have landed: 10 k-steps
  // of weight loads have been issued since.
  static constexpr int WAIT0[9] = {28, 31, 34, 37, 36, 33, 30, 28, 28};
  static constexpr int WAIT1[9] = {31, 34, 37, 39, 36, 33, 30, 28, 28};
  static constexpr int HWAIT = 40;
};

__device__ __forceinline__ int a7_off(int hr, int hc, int c) {
  return (Tap7Cfg::HPW / 2 * hr + (hc >> 1)) * 256 + ((hc & 1) << 7) + ((c ^ (((hc >> 1) & 3) << 1)) << 4);
}

__device__ __forceinline__ void ph7_wait_vmcnt(int n) {
  switch (n) {
#define PH7_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    PH7_W(16) PH7_W(28) PH7_W(30) PH7_W(31) PH7_W(33) PH7_W(34) PH7_W(36) PH7_W(37) PH7_W(39) PH7_W(40)
#undef PH7_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// BST: fused BatchNorm-backward sums over the tensor this (dgrad) launch writes (PhTapConv::bst_y, conv_tap3.hip's semantics and row
// layout): 0 = none (rows [2][Cout]: sum y | sum y^2 of a forward launch), 1 = mask from the BatchNorm's own ReLU
// (bst_y * bst_scale + bst_shift > 0), 2 = mask (bst_a > 0), 3 = 2 + a second BatchNorm (bst_y2) over the same dz; rows [3][Cout]:
// sum dz | sum dz (y - mean) | sum dz (y2 - mean2), taken over the STORED (bf16) gradient.
template <int BST>
__global__ __launch_bounds__(256) void tapconv7_kernel(PhTapConv p) {
  using C = Tap7Cfg;
  constexpr int NM = C::NM, NN = C::NN, TH = C::TH, TW = C::TW, HPW = C::HPW, NTAPS = C::NTAPS, BNT = C::BNT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(lds_uchar*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int tiles_w = (p.OW + TW - 1) / TW;
  const int tiles_sp = tiles_w * ((p.OH + TH - 1) / TH);
  const int nblk = p.Cout / BNT;
  const int total = tiles_sp * nblk * p.B;
  const int G = p.Cin >> 6;                      // 64-channel slices of the input (even: the launcher checks)
  const long pixB = (long)p.Cin * 2, rowB = (long)p.IW * pixB, imgB = (long)p.IH * rowB;

  // ---- tile list (conv_tap3.hip's): linear tile id -> (spatial tile fastest, Cout block, image), XCD-contiguous
  struct TileCtx { int r0, c0, b, nb; const unsigned char* in; };      // in: pixel (r0 - 1, c0 - 1), slice 0
  const float rcp_sp = 1.0f / (float)tiles_sp, rcp_nb = 1.0f / (float)nblk, rcp_tw = 1.0f / (float)tiles_w;
  auto fdiv = [](int a, int d, float rcp) {
    int q = (int)((float)a * rcp);
    int r = a - q * d;
    if (r >= d) ++q;
    if (r < 0) --q;
    return q;
  };
  auto decode = [&](int t) __attribute__((always_inline)) -> TileCtx {
    TileCtx c;
    const int rest = fdiv(t, tiles_sp, rcp_sp);
    const int tile = t - rest * tiles_sp;
    c.b = fdiv(rest, nblk, rcp_nb);
    c.nb = rest - c.b * nblk;
    const int trow = fdiv(tile, tiles_w, rcp_tw);
    c.r0 = trow * TH;
    c.c0 = (tile - trow * tiles_w) * TW;
    c.b = __builtin_amdgcn_readfirstlane(c.b);
    c.nb = __builtin_amdgcn_readfirstlane(c.nb);
    c.r0 = __builtin_amdgcn_readfirstlane(c.r0);
    c.c0 = __builtin_amdgcn_readfirstlane(c.c0);
    c.in = reinterpret_cast<const unsigned char*>(p.in) + (long)c.b * imgB + (long)(c.r0 - 1) * rowB + (long)(c.c0 - 1) * pixB;
    return c;
  };
  const int GR = gridDim.x;
  const bool xcd_map = (GR & 7) == 0 && GR < total;
  const int per_xcd = (total + 7) >> 3;
  auto tile_id = [&](int k) -> int {
    if (!xcd_map) {
      const int t = (int)blockIdx.x + k * GR;
      return t < total ? t : -1;
    }
    const int local = ((int)blockIdx.x >> 3) + k * (GR >> 3);
    const int t = ((int)blockIdx.x & 7) * per_xcd + local;
    return (local < per_xcd && t < total) ? t : -1;
  };

  // ---- weights: the fragment-major copy (plane 1 of the unit's packed region): [slab wtap[t]][Cout / 64][Cin / 64] x 8 KiB; a wave
  // reads row block 2 nb + wn.  Base + 4 KiB: the 8 fragments of a (tap, slice) are the immediates -4096 .. 3072.
  const long slabB = (long)p.Cout * p.Cin * 2;
  const unsigned char* w0 = reinterpret_cast<const unsigned char*>(p.w) + (long)p.wplane * 2 + 4096;
  const int voffB = lane * 16;
  auto w_base = [&](int nb, int c, int t) -> const unsigned char* {      // per-lane address of the lane's 16 bytes of fragment (0, 0); t = tap (literal)
    return w0 + (long)p.wtap[t] * slabB + (long)((2 * nb + wn) * G + c) * 8192 + voffB;
  };

  // ---- per-lane DMA sources: piece h = wave + 4 e covers row pairs 4 h .. 4 h + 3 of an image; lane l fills slot l & 15 of row pair
  // rp = 4 h + (l >> 4): image row rp / 9, column 2 (rp % 9) + (slot >> 3), chunk (slot & 7) ^ T(column)
  int h_off[C::NHE];
#pragma unroll
  for (int e = 0; e < C::NHE; ++e) {
    const int rp = (wave + 4 * e) * 4 + (lane >> 4), s = lane & 15;
    const int hr = rp / (HPW / 2), q = rp - hr * (HPW / 2);
    const int hc = 2 * q + (s >> 3), ch = (s & 7) ^ ((q & 3) << 1);
    h_off[e] = (int)((long)hr * rowB + (long)hc * pixB + ch * 16);
  }
  const unsigned char* zero_src = reinterpret_cast<const unsigned char*>(ph7_zero16);
  // bit e of a tile's mask: this lane's source pixel of piece e lies inside the image
  auto piece_bit = [&](const int e, const int r0, const int c0) __attribute__((always_inline)) -> unsigned {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int h = wave + 4 * e;
    const int rp = h * 4 + (ln >> 4), s = ln & 15;
    const int hr = (int)(__umul24((unsigned)rp, 7282u) >> 16);      // rp / 9 for rp < 1024
    const int hc = 2 * (rp - hr * (HPW / 2)) + (s >> 3);
    const bool ok = (h < C::NHD) && (hr < C::HPH) && ((unsigned)(r0 - 1 + hr) < (unsigned)p.IH) && ((unsigned)(c0 - 1 + hc) < (unsigned)p.IW);
    return ok ? (1u << e) : 0u;
  };
  // one piece: e of slice c of the tile at `tin` into buffer abuf (a piece past the image - the last e of waves 1..3 - repeats the
  // wave's previous piece: same bytes to the same place, so that every wave issues every e)
  auto dma_piece = [&](const int e, const unsigned char* tin, const int c, const unsigned mask, const int abuf) __attribute__((always_inline)) {
    const bool past = wave + 4 * e >= C::NHD;
    const int ee = e > 0 ? e - 1 : 0;
    const int h = past ? wave + 4 * ee : wave + 4 * e;
    int off_e = h_off[e], off_p = h_off[ee];
    asm volatile("" : "+v"(off_e));      // (opaque: see conv_tap6.hip)
    asm volatile("" : "+v"(off_p));
    const int off = past ? off_p : off_e;
    const unsigned bit = (past ? mask >> ee : mask >> e) & 1u;
    const unsigned char* src = bit ? tin + c * 128 + off : zero_src;
    const unsigned dst = lds0 + abuf * C::A_BYTES + h * 1024;
    lds_dma16(src, dst);
  };

  // ---- per-lane fragment addressing: base of (image row 8 wm, column li + dx, chunk lg), dx = 0, 1, 2, both buffers, + k-step-1 twin
  int ab0[2][3], ab1[2][3];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      ab0[bf][dx] = bf * C::A_BYTES + a7_off(wm * NM, li + dx, lg);
      ab1[bf][dx] = ab0[bf][dx] ^ 64;
    }

  f32x4 acc[NM][NN];
  constexpr bool Y2 = BST == 3;
  constexpr int NSR = BST > 0 ? 3 : 2;      // rows per workgroup (BST = -1: a launch without sums - plain dgrad - skips the arithmetic too)
  float s1[NN], s2[NN], s3[Y2 ? NN : 1];
#pragma unroll
  for (int n = 0; n < NN; ++n) { s1[n] = 0.f; s2[n] = 0.f; s3[Y2 ? n : 0] = 0.f; }
  // BatchNorm partial row [blockIdx][2][Cout] (conv_tap3.hip's: one per persistent workgroup, the same fp32 additions in the same
  // order - the two pixel-row halves of a block combined first, then added to what earlier visits of the block left): zeroed here
  if (p.stats) {
    float* rows = p.stats + (size_t)blockIdx.x * NSR * p.Cout;
    for (int i = tid; i < NSR * p.Cout; i += C::NTH) rows[i] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // (called between two tiles: buffer 1 - the image of the tile's last slice - is free until the next tile's first pieces)
  auto flush_stats = [&](const int nb) __attribute__((always_inline)) {
    if (!p.stats) return;
    float* red = reinterpret_cast<float*>(smem + C::A_BYTES);      // [WM][NSR][BNT]
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      float x1 = s1[n], x2 = s2[n], x3 = s3[Y2 ? n : 0];
      x1 += __shfl_xor(x1, 16, 64); x2 += __shfl_xor(x2, 16, 64);
      x1 += __shfl_xor(x1, 32, 64); x2 += __shfl_xor(x2, 32, 64);
      if (Y2) { x3 += __shfl_xor(x3, 16, 64); x3 += __shfl_xor(x3, 32, 64); } else x3 = 0.f;
      if (lg == 0) {
        red[(wm * NSR + 0) * BNT + wn * 64 + 4 * li + n] = x1;
        red[(wm * NSR + 1) * BNT + wn * 64 + 4 * li + n] = x2;
        if (NSR == 3) red[(wm * NSR + 2) * BNT + wn * 64 + 4 * li + n] = x3;
      }
      s1[n] = 0.f; s2[n] = 0.f; s3[Y2 ? n : 0] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // an earlier flush of this row has landed before it is read back
    __syncthreads();
    for (int i = tid; i < NSR * BNT; i += C::NTH) {
      const int which = i / BNT, n = i % BNT;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < C::WM; ++w) v += red[(w * NSR + which) * BNT + n];
      volatile float* row = p.stats + ((size_t)blockIdx.x * NSR + which) * p.Cout + nb * BNT + n;
      *row = *row + v;
    }
    __syncthreads();
  };

  // ---- epilogue of one tile: accumulator register q of tile (m, n) is pixel (row 8 wm + m, column 4 lg + q), channel
  // n0 + 64 wn + 4 li + n: four bf16 = one 8-byte store per (m, q) through a buffer resource of the tile's image (a lane outside the
  // output gets an offset past the resource); residual / mask operands (dgrad) are loaded a tile row at a time.
  constexpr unsigned OOB = 0x7ffffff0u;
  constexpr int RSRC_FLAGS = 0x00020000;
  const int img_bytes = p.OH * p.OW * p.Cout * 2;
  auto epilogue = [&](const TileCtx& tc, auto fullc, auto rmc) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(fullc)::value;
    constexpr int RM = decltype(rmc)::value;
    const size_t img = (size_t)tc.b * p.OH * p.OW * p.Cout;
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16*>(p.out) + img, 0, img_bytes, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(reinterpret_cast<const bf16*>(p.res_g)) + img, 0, RM > 0 ? img_bytes : 0, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(reinterpret_cast<const bf16*>(p.res_a)) + img, 0, RM > 1 ? img_bytes : 0, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(reinterpret_cast<const bf16*>(p.bst_y)) + img, 0, BST > 0 ? img_bytes : 0, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(reinterpret_cast<const bf16*>(p.bst_a)) + img, 0, BST >= 2 ? img_bytes : 0, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_y2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(reinterpret_cast<const bf16*>(p.bst_y2)) + img, 0, Y2 ? img_bytes : 0, RSRC_FLAGS);
    // per-channel constants of this lane's four channels: mask scale / shift (BST 1), mean, second mean
    f32x4 cms = {0.f, 0.f, 0.f, 0.f}, cmh = cms, cmu = cms, cmu2 = cms;
    if constexpr (BST > 0) {
      const int chan = tc.nb * BNT + wn * 64 + 4 * li;
      if constexpr (BST == 1) { cms = *reinterpret_cast<const f32x4*>(p.bst_scale + chan); cmh = *reinterpret_cast<const f32x4*>(p.bst_shift + chan); }
      cmu = *reinterpret_cast<const f32x4*>(p.bst_mean + chan);
      if constexpr (Y2) cmu2 = *reinterpret_cast<const f32x4*>(p.bst_mean2 + chan);
    }
    const unsigned o00 = 2u * ((unsigned)((tc.r0 + wm * NM) * p.OW + tc.c0 + 4 * lg) * (unsigned)p.Cout + (unsigned)(tc.nb * BNT + wn * 64 + 4 * li));
    const unsigned rowstep = 2u * (unsigned)(p.OW * p.Cout), colstep = 2u * (unsigned)p.Cout;
    const int rlim = p.OH - (tc.r0 + wm * NM), clim = p.OW - (tc.c0 + 4 * lg);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const unsigned orow = o00 + (unsigned)m * rowstep;
      int off[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) off[q] = (int)((FULL || (m < rlim && q < clim)) ? orow + (unsigned)q * colstep : OOB);
      u32x2 rg[4], ra[4], ry[4], rb[4], ry2[4];
      if constexpr (RM > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          rg[q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r_g, off[q], 0, 0));
          if constexpr (RM > 1) ra[q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r_a, off[q], 0, 0));
        }
      }
      if constexpr (BST > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ry[q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r_y, off[q], 0, 0));
          if constexpr (BST >= 2) rb[q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r_b, off[q], 0, 0));
          if constexpr (Y2) ry2[q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r_y2, off[q], 0, 0));
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool mine = FULL || (m < rlim && q < clim);
        float v[4];
#pragma unroll
        for (int n = 0; n < NN; ++n) {
          float x;      // (read through inline assembly: see conv_tap5.hip)
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[m][n][q]));
          v[n] = x;
          if constexpr (!FULL) v[n] = mine ? v[n] : 0.f;
          if constexpr (BST == 0) {
            s1[n] += v[n];
            s2[n] = __builtin_fmaf(v[n], v[n], s2[n]);
          }
        }
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        u32x2 w;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x2_t b;
          b[0] = (bf16)v[2 * h];
          b[1] = (bf16)v[2 * h + 1];
          if constexpr (RM > 0) {      // (conv_tap3.hip's arithmetic: round, add the residual gradient where the mask passes, round)
            const float g0 = __builtin_bit_cast(float, rg[q][h] << 16), g1 = __builtin_bit_cast(float, rg[q][h] & 0xffff0000u);
            float a0 = 1.f, a1 = 1.f;
            if constexpr (RM > 1) { a0 = __builtin_bit_cast(float, ra[q][h] << 16); a1 = __builtin_bit_cast(float, ra[q][h] & 0xffff0000u); }
            b[0] = (bf16)((float)b[0] + ((RM < 2 || a0 > 0.f) ? g0 : 0.f));
            b[1] = (bf16)((float)b[1] + ((RM < 2 || a1 > 0.f) ? g1 : 0.f));
          }
          w[h] = __builtin_bit_cast(unsigned, b);
          if constexpr (BST > 0) {
            // the sums are taken over the STORED gradient (bf16), exactly what the separate reduction pass reads back
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int n = 2 * h + e;
              const float y = __builtin_bit_cast(float, e ? (ry[q][h] & 0xffff0000u) : (ry[q][h] << 16));
              bool on;
              if constexpr (BST == 1) on = __builtin_fmaf(y, cms[n], cmh[n]) > 0.f;
              else on = __builtin_bit_cast(float, e ? (rb[q][h] & 0xffff0000u) : (rb[q][h] << 16)) > 0.f;
              float dz = (float)b[e];
              dz = (on && mine) ? dz : 0.f;
              s1[n] += dz;
              s2[n] = __builtin_fmaf(dz, y - cmu[n], s2[n]);
              if constexpr (Y2) {
                const float y2 = __builtin_bit_cast(float, e ? (ry2[q][h] & 0xffff0000u) : (ry2[q][h] << 16));
                s3[Y2 ? n : 0] = __builtin_fmaf(dz, y2 - cmu2[n], s3[Y2 ? n : 0]);
              }
            }
          }
        }
        __builtin_amdgcn_raw_buffer_store_b64(w, r_out, off[q], 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);      // one tile row at a time
    }
  };
  const int rmode = p.res_g ? (p.res_a ? 2 : 1) : 0;
  auto epilogue_any = [&](const TileCtx& tc) __attribute__((always_inline)) {
    const bool full = (tc.r0 + TH <= p.OH) && (tc.c0 + TW <= p.OW);
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
  unsigned mask_cur = 0, mask_next = 0;

  // Fragment registers: A ring of 4; B nine k-step sets (4 N tiles each): k-step j of the stream (18 per slice) multiplies set
  // j % 9 while the fragments of k-step j + 8 are loaded into set (j + 8) % 9 (conv_tap6.hip).  Sets 0..5 in accumulation registers,
  // 6..8 in vector registers.
  u32x4 fa[4], fba[6][NN], fbv[3][NN];
#define PH7_MM(M, N, AI, Q, FIRST)                                                                                                                  \
  do {                                                                                                                                              \
    if ((Q) < 6) {                                                                                                                                  \
      if (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc[M][N]) : "v"(fa[AI]), "a"(fba[(Q) < 6 ? (Q) : 0][N]));          \
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[M][N]) : "v"(fa[AI]), "a"(fba[(Q) < 6 ? (Q) : 0][N]));                \
    } else {                                                                                                                                        \
      if (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc[M][N]) : "v"(fa[AI]), "v"(fbv[(Q) >= 6 ? (Q) - 6 : 0][N]));     \
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[M][N]) : "v"(fa[AI]), "v"(fbv[(Q) >= 6 ? (Q) - 6 : 0][N]));           \
    }                                                                                                                                               \
  } while (0)
#define PH7_LD(ADDR, IMM) (*reinterpret_cast<const u32x4*>(smem + (ADDR) + (IMM)))
#define PH7_SB() __builtin_amdgcn_sched_barrier(0)
#define PH7_LDA(AB, AOFF, MT) PH7_LD(AB, (AOFF) + (MT) * C::ROW_BYTES)
  // (per-lane 64-bit address, no scalar operand: a scalar base restored by v_readlane right in front of inline assembly needs wait
  // states the hazard recognizer does not insert - conv_tap5.hip copies it with s_mov_b64 in front of every load; here ONE vector add
  // per tap makes the address and the eight loads carry nothing but an immediate)
#define PH7_BLD(Q, N, WB, OFF)                                                                                             \
  do {                                                                                                                     \
    if ((Q) < 6) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=a"(fba[(Q) < 6 ? (Q) : 0][N]) : "v"(WB) : "memory");     \
    else asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(fbv[(Q) >= 6 ? (Q) - 6 : 0][N]) : "v"(WB) : "memory");       \
  } while (0)
#define PH7_BLD_KN(Q, KS, N, WB)                                     \
  do {                                                               \
    if (PH7_DBG & 4) break;                                          \
    if ((KS) == 0) {                                                 \
      if ((N) == 0) PH7_BLD(Q, 0, WB, -4096);                        \
      else if ((N) == 1) PH7_BLD(Q, 1, WB, -3072);                   \
      else if ((N) == 2) PH7_BLD(Q, 2, WB, -2048);                   \
      else PH7_BLD(Q, 3, WB, -1024);                                 \
    } else {                                                         \
      if ((N) == 0) PH7_BLD(Q, 0, WB, 0);                            \
      else if ((N) == 1) PH7_BLD(Q, 1, WB, 1024);                    \
      else if ((N) == 2) PH7_BLD(Q, 2, WB, 2048);                    \
      else PH7_BLD(Q, 3, WB, 3072);                                  \
    }                                                                \
  } while (0)
#define PH7_GROUP(M, Q, KS, RA, X1, X2)                            \
  PH7_MM(M, 0, (M) & 3, Q, first && (KS) == 0); RA; PH7_SB();      \
  PH7_MM(M, 1, (M) & 3, Q, first && (KS) == 0); X1; PH7_SB();      \
  PH7_MM(M, 2, (M) & 3, Q, first && (KS) == 0); PH7_SB();          \
  PH7_MM(M, 3, (M) & 3, Q, first && (KS) == 0); X2; PH7_SB()
#define PH7_NOP ((void)0)

  {  // prologue: the image of slice 0, the weights of taps 0..3 (k-steps 0..7)
#pragma unroll
    for (int e = 0; e < C::NHE; ++e) mask_cur |= piece_bit(e, tcur.r0, tcur.c0);
#pragma unroll
    for (int e = 0; e < C::NHE; ++e) dma_piece(e, tcur.in, 0, mask_cur, 0);
    const unsigned char *wb0 = w_base(tcur.nb, 0, 0), *wb1 = w_base(tcur.nb, 0, 1), *wb2 = w_base(tcur.nb, 0, 2), *wb3 = w_base(tcur.nb, 0, 3);
    PH7_BLD(0, 0, wb0, -4096); PH7_BLD(0, 1, wb0, -3072); PH7_BLD(0, 2, wb0, -2048); PH7_BLD(0, 3, wb0, -1024);
    PH7_BLD(1, 0, wb0, 0); PH7_BLD(1, 1, wb0, 1024); PH7_BLD(1, 2, wb0, 2048); PH7_BLD(1, 3, wb0, 3072);
    PH7_BLD(2, 0, wb1, -4096); PH7_BLD(2, 1, wb1, -3072); PH7_BLD(2, 2, wb1, -2048); PH7_BLD(2, 3, wb1, -1024);
    PH7_BLD(3, 0, wb1, 0); PH7_BLD(3, 1, wb1, 1024); PH7_BLD(3, 2, wb1, 2048); PH7_BLD(3, 3, wb1, 3072);
    PH7_BLD(4, 0, wb2, -4096); PH7_BLD(4, 1, wb2, -3072); PH7_BLD(4, 2, wb2, -2048); PH7_BLD(4, 3, wb2, -1024);
    PH7_BLD(5, 0, wb2, 0); PH7_BLD(5, 1, wb2, 1024); PH7_BLD(5, 2, wb2, 2048); PH7_BLD(5, 3, wb2, 3072);
    PH7_BLD(6, 0, wb3, -4096); PH7_BLD(6, 1, wb3, -3072); PH7_BLD(6, 2, wb3, -2048); PH7_BLD(6, 3, wb3, -1024);
    PH7_BLD(7, 0, wb3, 0); PH7_BLD(7, 1, wb3, 1024); PH7_BLD(7, 2, wb3, 2048); PH7_BLD(7, 3, wb3, 3072);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  fa[0] = PH7_LDA(ab0[0][0], 0, 0);
  fa[1] = PH7_LDA(ab0[0][0], 0, 1);
  fa[2] = PH7_LDA(ab0[0][0], 0, 2);

  // One slice = 9 taps x 2 k-steps x 8 MFMA groups.  ABUF (literal): the buffer of its image; FIRSTS (literal): it opens a tile
  // (first k-step writes the accumulators, first k-steps skip the waits: the one in front of the epilogue covered their fragments).
  // c: slice index; last: it closes the tile.
  auto slice_body = [&](const int ABUF, const bool FIRSTS, const int c, const bool last) __attribute__((always_inline)) {
    // the image and the weights "of the next slice": slice c + 1 of this tile, or slice 0 of the next tile
    const unsigned char* nin = last ? (nvalid ? tnext.in : tcur.in) : tcur.in;
    const int nc = last ? 0 : c + 1;
    const int nnb = (last && nvalid) ? tnext.nb : tcur.nb;
    auto tap = [&](const int t) __attribute__((always_inline)) {
      const int dy = t / 3, dx = t % 3;
      const int tnx = (t + 1) % NTAPS, dyn = tnx / 3, dxn = tnx % 3;
      const int aoff = dy * C::ROW_BYTES, aoffn = dyn * C::ROW_BYTES;
      const int BUFN = (t + 1 == NTAPS) ? (ABUF ^ 1) : ABUF;
      const int q0 = (2 * t) % 9, q1 = (2 * t + 1) % 9, ql0 = (2 * t + 8) % 9, ql1 = (2 * t + 9) % 9;
      const bool first = FIRSTS && t == 0;
      const unsigned char* wb = (t + 4 < NTAPS) ? w_base(tcur.nb, c, t + 4) : w_base(nnb, nc, t + 4 - NTAPS);
      const int nd = C::ND[t], e0 = C::E0[t];
#define PH7_DMA(I) do { if ((I) < nd && !(PH7_DBG & 2)) dma_piece(e0 + (I), nin, nc, (last && nvalid) ? mask_next : mask_cur, ABUF ^ 1); } while (0)
      PH7_GROUP(0, q0, 0, fa[3] = PH7_LDA(ab0[ABUF][dx], aoff, 3), PH7_BLD_KN(ql0, 0, 0, wb), PH7_NOP);
      PH7_GROUP(1, q0, 0, fa[0] = PH7_LDA(ab0[ABUF][dx], aoff, 4), PH7_BLD_KN(ql0, 0, 1, wb), PH7_NOP);
      PH7_GROUP(2, q0, 0, fa[1] = PH7_LDA(ab0[ABUF][dx], aoff, 5), PH7_BLD_KN(ql0, 0, 2, wb), PH7_NOP);
      PH7_GROUP(3, q0, 0, fa[2] = PH7_LDA(ab0[ABUF][dx], aoff, 6), PH7_BLD_KN(ql0, 0, 3, wb), PH7_NOP);
      PH7_GROUP(4, q0, 0, fa[3] = PH7_LDA(ab0[ABUF][dx], aoff, 7), PH7_NOP, PH7_NOP);
      PH7_GROUP(5, q0, 0, fa[0] = PH7_LDA(ab1[ABUF][dx], aoff, 0), PH7_NOP, PH7_NOP);
      PH7_GROUP(6, q0, 0, fa[1] = PH7_LDA(ab1[ABUF][dx], aoff, 1), PH7_NOP, PH7_NOP);
      PH7_GROUP(7, q0, 0, fa[2] = PH7_LDA(ab1[ABUF][dx], aoff, 2), PH7_NOP, PH7_NOP);
      // (a tile's k-steps 0..2 skip the wait: the one in front of the epilogue covered the fragments of its k-steps 0..3)
      if (!(FIRSTS && t <= 1)) ph7_wait_vmcnt(C::WAIT0[t]);
      PH7_GROUP(0, q1, 1, fa[3] = PH7_LDA(ab1[ABUF][dx], aoff, 3), PH7_BLD_KN(ql1, 1, 0, wb), PH7_DMA(0));
      PH7_GROUP(1, q1, 1, fa[0] = PH7_LDA(ab1[ABUF][dx], aoff, 4), PH7_BLD_KN(ql1, 1, 1, wb), PH7_DMA(1));
      PH7_GROUP(2, q1, 1, fa[1] = PH7_LDA(ab1[ABUF][dx], aoff, 5), PH7_BLD_KN(ql1, 1, 2, wb), PH7_DMA(2));
      PH7_GROUP(3, q1, 1, fa[2] = PH7_LDA(ab1[ABUF][dx], aoff, 6), PH7_BLD_KN(ql1, 1, 3, wb), PH7_NOP);
      PH7_GROUP(4, q1, 1, fa[3] = PH7_LDA(ab1[ABUF][dx], aoff, 7), PH7_NOP, PH7_NOP);
      // the taps without pieces take the piece mask of the NEXT tile (first slice of a tile): 3, 2, 2, 2, 2 pieces in taps 4..8
      if (FIRSTS && t >= 4) {
        if (t == 4) mask_next = piece_bit(0, tnext.r0, tnext.c0);
        mask_next |= piece_bit(2 * (t - 4) + 1, tnext.r0, tnext.c0) | piece_bit(2 * (t - 4) + 2, tnext.r0, tnext.c0);
      }
      // hand-over (last tap): this wave's pieces of the next image have landed (HWAIT), its last read of this buffer is issued
      if (t + 1 == NTAPS) {
        ph7_wait_vmcnt(C::HWAIT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
      }
      PH7_GROUP(5, q1, 1, fa[0] = PH7_LDA(ab0[BUFN][dxn], aoffn, 0), PH7_NOP, PH7_NOP);
      PH7_GROUP(6, q1, 1, fa[1] = PH7_LDA(ab0[BUFN][dxn], aoffn, 1), PH7_NOP, PH7_NOP);
      PH7_GROUP(7, q1, 1, fa[2] = PH7_LDA(ab0[BUFN][dxn], aoffn, 2), PH7_NOP, PH7_NOP);
      if (!(FIRSTS && t == 0)) ph7_wait_vmcnt(C::WAIT1[t]);
#undef PH7_DMA
    };
    tap(0); tap(1); tap(2); tap(3); tap(4); tap(5); tap(6); tap(7); tap(8);
  };

  for (int k = 0;;) {
    slice_body(0, true, 0, false);
    slice_body(1, false, 1, G == 2);
    for (int c = 2; c < G; c += 2) {
      slice_body(0, false, c, false);
      slice_body(1, false, c + 1, c + 2 == G);
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // all but the last 16 operations: the fragments of the next tile's k-steps 0..3 are older
    epilogue_any(tcur);
    if (!nvalid || tnext.nb != tcur.nb) flush_stats(tcur.nb);
    if (!nvalid) break;
    tcur = tnext;
    mask_cur = mask_next;
    tn = tile_id(k + 2);
    nvalid = tn >= 0;
    if (nvalid) tnext = decode(tn);
    ++k;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // pieces / fragments issued past the end of the stream must not outlive the workgroup
}

// row-major perf-mode weights [tap][R][K] (plane 0) -> the fragment-major copy (plane 1): one thread per 16-byte chunk
__global__ void frag7_repack_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int R, int K, int ntaps) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // chunk index: [tap][r][k / 8]
  const size_t n = (size_t)ntaps * R * (K >> 3);
  if (i >= n) return;
  const int k8 = (int)(i % (K >> 3));
  const int r = (int)((i / (K >> 3)) % R);
  const int tp = (int)(i / ((size_t)(K >> 3) * R));
  const int k = k8 * 8;
  const size_t d = (size_t)tp * R * K + (size_t)((r >> 6) * (K >> 6) + (k >> 6)) * 4096 +
                   (size_t)((((k & 63) >> 5) * 4 + (r & 3)) * 512 + (((((k & 63) >> 3) & 3) << 4) + ((r & 63) >> 2)) * 8);
  *reinterpret_cast<u32x4*>(dst + d) = *reinterpret_cast<const u32x4*>(src + ((size_t)tp * R + r) * K + k);
}

}  // namespace

// PH_TAP7=0 in the environment / ph_debug_set_tap7(0) keeps conv_tap3.hip's plain form (same-box A/B; the fragment-major copy of the
// weights is written either way)
int ph_tap7_switch(int set) {
  static int on = [] { const char* e = getenv("PH_TAP7"); return (e && e[0] == '0') ? 0 : 1; }();
  if (set >= 0) on = set ? 1 : 0;
  return on;
}
extern "C" int ph_debug_set_tap7(int on) { return ph_tap7_switch(on ? 1 : 0); }

// eligible: conv_tap3.hip's perf-mode configuration without the in-LDS input BatchNorm (dense 3x3 stride-1 over the whole map,
// Cin = Cout in 128 .. 512) on a dense NHWC tensor
bool ph_tapconv7_eligible(const PhTapConv* p) {
  return ph_tapconv3_eligible(p) && !p->in_scale && !p->m_groups && !p->ncls && p->Cin == p->Cout && p->os == 1 && p->oa_h == 0 &&
         p->oa_w == 0 && p->OHt == p->OH && p->OWt == p->OW && p->IH == p->OH && p->IW == p->OW && p->iy0 == -1 && p->ix0 == -1 &&
         !p->in_pix_stride && !p->in_row_stride && !p->in_img_stride && p->wplane == (size_t)9 * p->Cin * p->Cout &&
         (long)p->OH * p->OW * p->Cout * 2 < 0x7ffffff0L;
}

namespace {
template <int BST>
int launch7(const PhTapConv* p, hipStream_t st) {
  using C = Tap7Cfg;
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [&] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(tapconv7_kernel<BST>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  });
  if (attr_rc != hipSuccess) return PH_ELAUNCH;
  const int total = cdiv(p->OH, C::TH) * cdiv(p->OW, C::TW) * (p->Cout / C::BNT) * p->B;
  const int resident = ph_num_cus();
  dim3 grid(total < resident ? total : resident);
  void* tok = nullptr;
  if (ph_prof_on())
    ph_prof_begin2(PH_CLS_TAPCONV2, 2.0 * p->B * p->OH * p->OW * (double)p->Cout * 9 * p->Cin, ph_tapconv_bytes(*p, 1, 2), st, &tok);
  hipLaunchKernelGGL(tapconv7_kernel<BST>, grid, dim3(C::NTH), C::LDS_BYTES, st, *p);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
}  // namespace

int ph_tapconv7_launch(const PhTapConv* p, hipStream_t st) {
  if (!ph_tapconv7_eligible(p)) return PH_EINVAL;
  if (p->bst_y) {      // fused BatchNorm-backward sums (dgrad launches): conv_tap3.hip's argument rules
    if (!p->stats || !p->bst_mean || (p->bst_y2 && !p->bst_mean2) || (!p->bst_a && (!p->bst_scale || !p->bst_shift))) return PH_EINVAL;
    if (!p->bst_a) return p->bst_y2 ? PH_EINVAL : launch7<1>(p, st);
    return p->bst_y2 ? launch7<3>(p, st) : launch7<2>(p, st);
  }
  return p->stats ? launch7<0>(p, st) : launch7<-1>(p, st);
}

// the fragment-major copy of ONE convolution's packed perf-mode weights (test hooks, c_api.hip): plane 0 -> plane 1
int ph_frag7_repack_launch(void* packed, int R, int K, int ntaps, hipStream_t st) {
  if ((R & 63) || (K & 63)) return PH_EINVAL;
  bf16* base = reinterpret_cast<bf16*>(packed);
  const size_t n = (size_t)ntaps * R * (K >> 3);
  hipLaunchKernelGGL(frag7_repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, base, base + (size_t)ntaps * R * K, R, K, ntaps);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
