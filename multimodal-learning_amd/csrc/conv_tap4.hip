// Tap-convolution implicit GEMM, fourth generation: the 3x3 stride-1 convolutions with Cin = Cout = 64 (ResNet layer 1; forward,
// dgrad with the fused residual mask, optional BatchNorm + ReLU of the input applied in LDS) on v_mfma_f32_16x16x32_bf16.
//
// Why (profiles/EXPERIMENTS.md "Layer-1 kernel"): tapconv2_l1_kernel - 32x32x16 fragments, two wave groups per
// workgroup that alternate between a matrix phase and a store phase - sat at 93-100 us per 77 GFLOP launch (0.31 of the MFMA peak,
// 0.40 of HBM) for three rounds: both phases stretch to 6000-7000 cycles while they overlap (one ds_read_b128 per MFMA, two waves
// per SIMD contending for issue), 16.3 k cycles per tile pair against 9.2 k of MFMA.  This kernel keeps what worked (all nine
// taps of weights resident in LDS, persistent XCD-contiguous tile lists, LDS-DMA halos with the swizzle applied to the source)
// and changes the rest:
//   * ONE wave per SIMD, wave tile = 4 tile rows x 16 pixels x all 64 channels on 16x16x32 fragments: 16 MFMAs per k-step for
//     4 A + 4 B fragment reads (0.5 ds_read_b128 per MFMA of 16 cycles: half the LDS array's rate), every fragment address a
//     base register + an immediate (conv_tap3.hip's column-only halo swizzle);
//   * with the weights resident NOTHING inside a tile needs a workgroup barrier: the only shared state is the halo double
//     buffer, so a tile is 288 MFMAs per wave without a wait on the other waves, and one s_barrier per tile publishes the next
//     halo and releases the last one (the dense kernel pays a counted vmcnt wait + barrier per TAP for its weight ring);
//   * the weight rows of the four N tiles are interleaved by the DMA source mapping (tile n, row i = channel 4 i + n), so a lane
//     owns four consecutive channels of a pixel: an accumulator quad leaves as one 8-byte store, a wave-instruction writes four
//     whole 128-byte pixel records (conv_tap3.hip's epilogue);
//   * 64 accumulator registers per wave: the kernel needs < 256 registers, so BatchNorm / elementwise waves of the other streams
//     of the step can be resident beside it (the 512-register tap-conv kernels own their CUs).
// Optional fused BatchNorm-backward sums (PhTapConv::bst_y): a dgrad launch that produces the gradient a
// BatchNorm backward reduces over can take the per-channel sums sum dz and sum dz (y - mean) in its epilogue - the ReLU mask and
// y arrive as 8-byte loads in the accumulator layout, prefetched a tile row ahead - and the separate bn_bwd_reduce pass (a full
// read of the gradient, y and the mask tensor) disappears from the dgrad chain.
//
// Same GEMM view, descriptor (PhTapConv) and semantics as conv_tap2.hip / conv_tap3.hip; outputs BITWISE those of
// tapconv2_l1_kernel (same products, same fp32 accumulation order per output: slices of 32 channels, taps in order).
#include "ph_common.h"
#include <type_traits>
#include "ph_kernels.h"
#include "tap_common.h"

namespace {

__device__ const u32x4 ph4_zero16[4] = {};
__device__ const u32x4 ph4_nan16[4] = {{0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}, {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u},
                                       {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}, {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}};

typedef __attribute__((address_space(3))) unsigned char lds_uchar4;

__device__ __forceinline__ void lds_dma16_4(const void* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(g) : "memory");
}
#define PH4_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define PH4_BARRIER() asm volatile("s_barrier" ::: "memory")

// relu(x * s + h) on the 8 bf16 values of one 16-byte chunk (as conv_tap2.hip)
__device__ __forceinline__ u32x4 bn_relu_chunk4(u32x4 v, const f32x4& sA, const f32x4& sB, const f32x4& hA, const f32x4& hB) {
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

struct Tap4Cfg {
  static constexpr int NM = 4, NN = 4, NTAPS = 9;
  static constexpr int TH = 16, TW = 16, BNT = 64;
  static constexpr int HPH = TH + 2, HPW = TW + 2, HP = HPH * HPW;
  static constexpr int ROW_BYTES = (HPW / 2) * 256;                              // one halo row: 9 pixel pairs
  static constexpr int A_BYTES = (((HP + 1) / 2 * 256) + 1023) / 1024 * 1024;    // whole 1-KiB DMA pieces
  static constexpr int NHD = A_BYTES / 1024, NHE = (NHD + 3) / 4;
  static constexpr int TAPB = BNT * 128;                                         // one tap's weight block: 64 rows of 128 B
  static constexpr int W_BYTES = NTAPS * TAPB, NWP = W_BYTES / 1024 / 4;         // resident weights; DMA pieces per wave
  static constexpr int HALO_TAPS = 6, HPT = (NHE + HALO_TAPS - 1) / HALO_TAPS;
  static constexpr int B_BASE = 2 * A_BYTES;
  static constexpr int RED_OFF = B_BASE + W_BYTES;                               // [4 waves][3][64] floats: the final reduction of the sums
  static constexpr int SS_OFF = RED_OFF + 4 * 3 * BNT * 4;                       // [4][64] floats: scale / shift of the input's BatchNorm,
  static constexpr int LDS_BYTES = SS_OFF + 4 * BNT * 4;                         // or mask scale / shift / mean / mean2 of the fused sums
  static constexpr int NTH = 256;
  static_assert(HPT == 2, "2 halo pieces per wave in each of the first six taps");
  static_assert(W_BYTES % 4096 == 0, "weights split into 1-KiB pieces over 4 waves");
  static_assert(A_BYTES + (NM + 2) * ROW_BYTES < 65536, "ds_read immediate offsets");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// halo image: pixel (hr, hc), 16-byte chunk c of its 64 channels -> LDS byte offset inside an A buffer (conv_tap3.hip: a3_off)
__device__ __forceinline__ int a4_off(int hr, int hc, int c) {
  return (Tap4Cfg::HPW / 2 * hr + (hc >> 1)) * 256 + ((hc & 1) << 7) + ((c ^ (((hc >> 1) & 3) << 1)) << 4);
}

// BST: fused BatchNorm-backward sums over the tensor this launch writes (0 = none: forward launches take sum y / sum y^2 when
// p.stats is set).  1: dz = out * (bst_y * bst_scale + bst_shift > 0), the BatchNorm's own ReLU (bn1 of a BasicBlock, reduced
// over the output of conv2's dgrad); 2: dz = out * (bst_a > 0) (bn2, reduced over the block-input gradient that conv1's dgrad +
// residual writes).  Row [3][64] per workgroup: sum dz, sum dz (bst_y - bst_mean), sum dz (bst_y2 - bst_mean2) (0 without bst_y2).
template <bool FUSE_IN, int BST>
__global__ __launch_bounds__(256) void tapconv4_kernel(PhTapConv p) {
  static_assert(!(FUSE_IN && BST), "forward-only feature vs backward feature");
  using C = Tap4Cfg;
  constexpr int NM = C::NM, NN = C::NN, TH = C::TH, TW = C::TW, BNT = C::BNT, HPW = C::HPW, NTAPS = C::NTAPS;
  constexpr int NTH = C::NTH, B_BASE = C::B_BASE;
  typedef __bf16 T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(lds_uchar4*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = the wave's band of four tile rows
  const int li = lane & 15, lg = lane >> 4;                       // MFMA 16x16x32: row / column index, k group (A, B) or pixel group (C)
  const int tiles_w = (p.OWt + TW - 1) / TW;
  const int tiles_sp = tiles_w * ((p.OHt + TH - 1) / TH);
  const int total = tiles_sp * p.B;
  const long pix_st = p.in_pix_stride ? p.in_pix_stride : p.Cin;
  const long row_st = p.in_row_stride ? p.in_row_stride : (long)p.IW * p.Cin;
  const long img_st = p.in_img_stride ? p.in_img_stride : (long)p.IH * p.IW * p.Cin;
  const bf16* wbase = reinterpret_cast<const bf16*>(p.w);

  // ---- tile list: linear tile id -> (spatial tile fastest, image), XCD-contiguous (as conv_tap3.hip)
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
    c.b = __builtin_amdgcn_readfirstlane(c.b);
    c.r0 = __builtin_amdgcn_readfirstlane(c.r0);
    c.c0 = __builtin_amdgcn_readfirstlane(c.c0);
    c.iy_base = c.r0 + p.iy0;
    c.ix_base = c.c0 + p.ix0;
    c.in = reinterpret_cast<const T*>(p.in) + (size_t)c.b * img_st;
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

  // ---- per-lane DMA sources.  Halo piece h = wave + 4 e covers row pairs 4 h .. 4 h + 3 of the image; lane l fills slot l & 15 of
  // row pair rp = 4 h + (l >> 4): halo row rp / 9, column 2 (rp % 9) + (slot >> 3), chunk (slot & 7) ^ T(column)
  int h_off[C::NHE];
#pragma unroll
  for (int e = 0; e < C::NHE; ++e) {
    const int rp = (wave + 4 * e) * 4 + (lane >> 4), s = lane & 15;
    const int hr = rp / (HPW / 2), q = rp - hr * (HPW / 2);
    const int hc = 2 * q + (s >> 3), ch = (s & 7) ^ ((q & 3) << 1);
    h_off[e] = (int)(((long)hr * row_st + (long)hc * pix_st + ch * 8) * 2);
  }
  auto piece_rc = [&](int e, int& hr, int& hc) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int rp = (wave + 4 * e) * 4 + (ln >> 4), s = ln & 15;
    hr = (rp * 7282) >> 16;                            // rp / 9 for rp < 1024
    const int q = rp - hr * (HPW / 2);
    hc = 2 * q + (s >> 3);
  };
  auto halo_mask = [&](int iy_base, int ix_base) {
    auto range_bits = [](int lo, int hi) -> unsigned {
      return hi > lo ? ((hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u)) : 0u;
    };
    const int r_lo = iy_base < 0 ? -iy_base : 0, r_hi = (p.IH - iy_base) < C::HPH ? (p.IH - iy_base) : C::HPH;
    const int c_lo = ix_base < 0 ? -ix_base : 0, c_hi = (p.IW - ix_base) < HPW ? (p.IW - ix_base) : HPW;
    const unsigned rowok = range_bits(r_lo, r_hi < 0 ? 0 : r_hi), colok = range_bits(c_lo, c_hi < 0 ? 0 : c_hi);
    int m = 0;
#pragma unroll
    for (int e = 0; e < C::NHE; ++e) {
      int hr, hc;
      piece_rc(e, hr, hc);
      const unsigned ok = (rowok >> (hr & 31)) & (colok >> (hc & 31)) & (hr < C::HPH ? 1u : 0u);
      m |= (int)(ok & 1u) << e;
    }
    return m;
  };
  const unsigned char* zero_src = reinterpret_cast<const unsigned char*>(FUSE_IN ? ph4_nan16 : ph4_zero16);
  float* ss = reinterpret_cast<float*>(smem + C::SS_OFF);
  // BatchNorm + ReLU of the INPUT applied in LDS: every wave transforms the halo pieces it issued itself
  auto xform_halo = [&](int abuf) {
#pragma unroll
    for (int e = 0; e < C::NHE; ++e)
      if (wave + 4 * e < C::NHD) {
        const int rp = (wave + 4 * e) * 4 + (lane >> 4);
        const int q = rp % (HPW / 2);
        const int cg = (((lane & 7) ^ ((q & 3) << 1)) & 7) * 8;
        const f32x4 sA = *reinterpret_cast<const f32x4*>(ss + cg), sB = *reinterpret_cast<const f32x4*>(ss + cg + 4);
        const f32x4 hA = *reinterpret_cast<const f32x4*>(ss + BNT + cg), hB = *reinterpret_cast<const f32x4*>(ss + BNT + cg + 4);
        u32x4* a = reinterpret_cast<u32x4*>(smem + abuf * C::A_BYTES + (wave + 4 * e) * 1024 + lane * 16);
        *a = bn_relu_chunk4(*a, sA, sB, hA, hB);
      }
  };

  // ---- per-lane fragment addressing.  A: base of (halo row 4 wave, column li + dx, chunk lg) for dx = 0, 1, 2 and its k-step-1
  // twin (chunk lg + 4 = address ^ 64); tile row m, tap row dy and the A buffer are immediate offsets.  B: row 16 n + li of the
  // tap's 64 rows, chunk lg; the tap is an immediate (tap 8 = 7 taps past a second base: 8 * TAPB exceeds the 16-bit offset).
  int abase0[3], abase1[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    abase0[dx] = a4_off(wave * NM, li + dx, lg);
    abase1[dx] = abase0[dx] ^ 64;
  }
  int bx0[NN], bx1[NN];
#pragma unroll
  for (int n = 0; n < NN; ++n) {
    bx0[n] = B_BASE + lds_off(n * 16 + li, lg);
    bx1[n] = bx0[n] ^ 64;
  }

  // ---- sums over all tiles of this workgroup: a lane owns channels 4 li + n (n = 0..3)
  float s1[NN], s2[NN], s3[NN];
#pragma unroll
  for (int n = 0; n < NN; ++n) { s1[n] = 0.f; s2[n] = 0.f; s3[n] = 0.f; }

  // ---- prologue: per-channel constants, all nine taps of weights, the first halo
  if (FUSE_IN) {
    if (tid < 2 * BNT) ss[tid] = tid < BNT ? p.in_scale[tid] : p.in_shift[tid - BNT];
  }
  if (BST) {   // [mask scale | mask shift | mean | mean2]
    if (tid < BNT) {
      ss[tid] = (BST == 1) ? p.bst_scale[tid] : 0.f;
      ss[BNT + tid] = (BST == 1) ? p.bst_shift[tid] : 0.f;
      ss[2 * BNT + tid] = p.bst_mean[tid];
      ss[3 * BNT + tid] = p.bst_y2 ? p.bst_mean2[tid] : 0.f;
    }
  }
  if (FUSE_IN || BST) __syncthreads();
  {
    // weight piece q = wave * NWP + j of the 72: tap q / 8, LDS rows 8 (q % 8) .. + 7 of the tap block; LDS row R holds channel
    // 4 (R & 15) + (R >> 4) (N tile R >> 4, MFMA row R & 15), so that a lane of the MFMA owns four consecutive channels
#pragma unroll
    for (int j = 0; j < C::NWP; ++j) {
      const int q = wave * C::NWP + j, tap = q >> 3;
      const int rp = (q & 7) * 4 + (lane >> 4), u = (lane & 15) ^ (rp & PH_SWZ_MASK);
      const int R = 2 * rp + (u >> 3);
      const int ch = ((R & 15) << 2) | (R >> 4);
      const int slab = p.wtap[tap];
      const unsigned char* wb = reinterpret_cast<const unsigned char*>(wbase + (size_t)slab * p.Cout * p.Cin);
      lds_dma16_4(wb + (ch * p.Cin + (u & 7) * 8) * 2, lds0 + B_BASE + q * 1024);
    }
  }
  TileCtx tcur = decode(tile_id(0));
  int tn = tile_id(1);
  bool nvalid = tn >= 0;
  TileCtx tnext = tcur;
  if (nvalid) tnext = decode(tn);
  int hm_next = nvalid ? halo_mask(tnext.iy_base, tnext.ix_base) : 0;
  auto halo_base = [&](const TileCtx& tc) {
    return reinterpret_cast<const unsigned char*>(tc.in) + ((long)tc.iy_base * row_st + (long)tc.ix_base * pix_st) * 2;
  };
  {
    const int hm = halo_mask(tcur.iy_base, tcur.ix_base);
    const unsigned char* hb = halo_base(tcur);
#pragma unroll
    for (int e = 0; e < C::NHE; ++e)
      if (wave + 4 * e < C::NHD)
        lds_dma16_4(((hm >> e) & 1) ? hb + h_off[e] : zero_src, lds0 + (wave + 4 * e) * 1024);
  }
  PH4_WAIT_VMCNT(0);
  if (FUSE_IN) xform_halo(0);
  PH4_BARRIER();

  f32x4 acc[NM][NN];
  bf16x8 fa[2][NM], fb[2][NN];
#define PH4_MM(M, N, S) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[M][N]) : "v"(fa[S][M]), "v"(fb[S][N]))
#define PH4_MM0(M, N, S) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc[M][N]) : "v"(fa[S][M]), "v"(fb[S][N]))
#define PH4_LD(ADDR, IMM) (*reinterpret_cast<const bf16x8*>(smem + (ADDR) + (IMM)))
#define PH4_SB() __builtin_amdgcn_sched_barrier(0)
#define PH4_NOP ((void)0)
  // one M group of a k-step on fragment set S: RA / RB = the reads of the NEXT k-step's A tile M / B tile M into set S ^ 1
#define PH4_GROUP(MMAC, M, S, RA, RB, F0, F1)   \
  MMAC(M, 0, S); RA; PH4_SB();                  \
  MMAC(M, 1, S); RB; PH4_SB();                  \
  MMAC(M, 2, S); F0; PH4_SB();                  \
  MMAC(M, 3, S); F1; PH4_SB()

  // ---- epilogue of one tile.  Accumulator register r of tile (m, n) is pixel (row 4 wave + m, column 4 lg + r), channel
  // 4 li + n: the four N tiles give four consecutive channels = one 8-byte store per (m, r); a wave-instruction writes 128
  // contiguous bytes for each of four pixels.
  auto epilogue = [&](const TileCtx& tc, auto fullc, auto rmc) {
    constexpr bool FULL = decltype(fullc)::value;
    constexpr int RM = decltype(rmc)::value;
    const size_t img = (size_t)tc.b * p.OH * p.OW * p.Cout;
    T* out = reinterpret_cast<T*>(p.out) + img;
    const T* resg = reinterpret_cast<const T*>(p.res_g) + img;
    const T* resa = reinterpret_cast<const T*>(p.res_a) + img;
    const T* bsty = reinterpret_cast<const T*>(p.bst_y) + img;
    const T* bsta = reinterpret_cast<const T*>(p.bst_a) + img;
    const T* bsty2 = reinterpret_cast<const T*>(p.bst_y2) + img;
    const bool has_y2 = BST && p.bst_y2 != nullptr;
    const unsigned colstep = (unsigned)(p.os * p.Cout);
    const unsigned o00 = (unsigned)(((tc.r0 + wave * NM) * p.os + p.oa_h) * p.OW + (tc.c0 + 4 * lg) * p.os + p.oa_w) * (unsigned)p.Cout + 4u * (unsigned)li;
    const unsigned rowstep = (unsigned)(p.os * p.OW * p.Cout);
    f32x4 cms = {0.f, 0.f, 0.f, 0.f}, cmh = cms, cmu = cms, cmu2 = cms;
    if constexpr (BST != 0) {
      if constexpr (BST == 1) { cms = *reinterpret_cast<const f32x4*>(ss + 4 * li); cmh = *reinterpret_cast<const f32x4*>(ss + BNT + 4 * li); }
      cmu = *reinterpret_cast<const f32x4*>(ss + 2 * BNT + 4 * li);
      cmu2 = *reinterpret_cast<const f32x4*>(ss + 3 * BNT + 4 * li);
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int r = tc.r0 + wave * NM + m;
      const unsigned orow = o00 + (unsigned)m * rowstep;
      u32x2 rg[4], ra[4], ry[4], rb[4], ry2[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = tc.c0 + 4 * lg + q;
        const bool mine = FULL || (r < p.OHt && c < p.OWt);
        const unsigned o = orow + (unsigned)q * colstep;
        rg[q] = u32x2{0u, 0u};
        ra[q] = u32x2{0x3f803f80u, 0x3f803f80u};
        ry[q] = u32x2{0u, 0u}; rb[q] = u32x2{0u, 0u}; ry2[q] = u32x2{0u, 0u};
        if (mine) {
          if constexpr (RM > 0) rg[q] = *reinterpret_cast<const u32x2*>(resg + o);
          if constexpr (RM > 1) ra[q] = *reinterpret_cast<const u32x2*>(resa + o);
          if constexpr (BST != 0) {
            ry[q] = *reinterpret_cast<const u32x2*>(bsty + o);
            if constexpr (BST == 2) rb[q] = *reinterpret_cast<const u32x2*>(bsta + o);
            if (has_y2) ry2[q] = *reinterpret_cast<const u32x2*>(bsty2 + o);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = tc.c0 + 4 * lg + q;
        const bool mine = FULL || (r < p.OHt && c < p.OWt);
        float v[4];
#pragma unroll
        for (int n = 0; n < NN; ++n) {
          v[n] = acc[m][n][q];
          if constexpr (!FULL) v[n] = mine ? v[n] : 0.f;
          if constexpr (BST == 0) {
            s1[n] += v[n];
            s2[n] += v[n] * v[n];
          }
        }
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        u32x2 w;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x2 b;
          b[0] = (bf16)v[2 * h];
          b[1] = (bf16)v[2 * h + 1];
          if constexpr (RM > 0) {
            const float g0 = __builtin_bit_cast(float, rg[q][h] << 16), g1 = __builtin_bit_cast(float, rg[q][h] & 0xffff0000u);
            const float a0 = __builtin_bit_cast(float, ra[q][h] << 16), a1 = __builtin_bit_cast(float, ra[q][h] & 0xffff0000u);
            b[0] = (bf16)((float)b[0] + ((RM < 2 || a0 > 0.f) ? g0 : 0.f));
            b[1] = (bf16)((float)b[1] + ((RM < 2 || a1 > 0.f) ? g1 : 0.f));
          }
          w[h] = __builtin_bit_cast(unsigned, b);
          if constexpr (BST != 0) {
            // the sums are taken over the STORED gradient (bf16), exactly what the separate reduction pass reads back
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int n = 2 * h + e;
              const float y = __builtin_bit_cast(float, e ? (ry[q][h] & 0xffff0000u) : (ry[q][h] << 16));
              bool on;
              if constexpr (BST == 1) on = (y * cms[n] + cmh[n]) > 0.f;
              else on = __builtin_bit_cast(float, e ? (rb[q][h] & 0xffff0000u) : (rb[q][h] << 16)) > 0.f;
              float dz = (float)b[e];
              dz = (on && mine) ? dz : 0.f;
              s1[n] += dz;
              s2[n] += dz * (y - cmu[n]);
              if (has_y2) {
                const float y2 = __builtin_bit_cast(float, e ? (ry2[q][h] & 0xffff0000u) : (ry2[q][h] << 16));
                s3[n] += dz * (y2 - cmu2[n]);
              }
            }
          }
        }
        if (mine) *reinterpret_cast<u32x2*>(out + (orow + (unsigned)q * colstep)) = w;
      }
      __builtin_amdgcn_sched_barrier(0);      // one tile row at a time: hoisting every load of the tile in front spills
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

  // ---- the tile stream.  ABUF (the halo buffer of this tile) is a literal at both call sites, so that after inlining every
  // fragment address is a base register + an immediate.  Fragment set 0 / 1 = k-step 0 / 1 of a tap; each M group's four MFMAs
  // carry the read of the next k-step's A tile and B tile of the same index.  Taps 0..5 also issue, two pieces per wave, the
  // LDS-DMA of the NEXT tile's halo into the other buffer (nobody reads it before the barrier at the end of this tile).
  auto tile_body = [&](const int ABUF) __attribute__((always_inline)) {
    const int AOFF = ABUF * C::A_BYTES;
    const unsigned char* hb = halo_base(tnext);
    const unsigned hdst = lds0 + (ABUF ^ 1) * C::A_BYTES + wave * 1024;
#define PH4_DMA_H(E)                                                                                           \
  do {                                                                                                         \
    if (nvalid && (E) < C::NHE && wave + 4 * (E) < C::NHD)                                                     \
      lds_dma16_4(((hm_next >> (E)) & 1) ? hb + h_off[(E) < C::NHE ? (E) : 0] : zero_src, hdst + (E) * 4096); \
  } while (0)
    // first fragments of the tile: tap 0, k-step 0
#pragma unroll
    for (int m = 0; m < NM; ++m) fa[0][m] = PH4_LD(abase0[0], AOFF + m * C::ROW_BYTES);
#pragma unroll
    for (int n = 0; n < NN; ++n) fb[0][n] = PH4_LD(bx0[n], 0);
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      const int dy = t / 3, dx = t % 3;
      const int tn_ = t + 1, dyn = tn_ / 3, dxn = tn_ % 3;
      const int aoff = AOFF + dy * C::ROW_BYTES, aoffn = AOFF + dyn * C::ROW_BYTES;
      // (tap 8's block lies 8 * TAPB = 65536 bytes past the base: one more than the offset field holds)
      const int boff = (t == 8 ? 7 : t) * C::TAPB, bext = t == 8 ? C::TAPB : 0;
      const int boffn = (tn_ == 8 ? 7 : tn_) * C::TAPB, bextn = tn_ == 8 ? C::TAPB : 0;
      // ---- k-step 0 (chunks lg) on set 0; reads of k-step 1 (chunks lg + 4) into set 1
      if (t == 0) {
        PH4_GROUP(PH4_MM0, 0, 0, fa[1][0] = PH4_LD(abase1[dx], aoff + 0 * C::ROW_BYTES), fb[1][0] = PH4_LD(bx1[0] + bext, boff), PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM0, 1, 0, fa[1][1] = PH4_LD(abase1[dx], aoff + 1 * C::ROW_BYTES), fb[1][1] = PH4_LD(bx1[1] + bext, boff), PH4_NOP, PH4_DMA_H(0));
        PH4_GROUP(PH4_MM0, 2, 0, fa[1][2] = PH4_LD(abase1[dx], aoff + 2 * C::ROW_BYTES), fb[1][2] = PH4_LD(bx1[2] + bext, boff), PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM0, 3, 0, fa[1][3] = PH4_LD(abase1[dx], aoff + 3 * C::ROW_BYTES), fb[1][3] = PH4_LD(bx1[3] + bext, boff), PH4_NOP, PH4_NOP);
      } else {
        PH4_GROUP(PH4_MM, 0, 0, fa[1][0] = PH4_LD(abase1[dx], aoff + 0 * C::ROW_BYTES), fb[1][0] = PH4_LD(bx1[0] + bext, boff), PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM, 1, 0, fa[1][1] = PH4_LD(abase1[dx], aoff + 1 * C::ROW_BYTES), fb[1][1] = PH4_LD(bx1[1] + bext, boff), PH4_NOP,
                  if (t < C::HALO_TAPS) PH4_DMA_H(2 * t));
        PH4_GROUP(PH4_MM, 2, 0, fa[1][2] = PH4_LD(abase1[dx], aoff + 2 * C::ROW_BYTES), fb[1][2] = PH4_LD(bx1[2] + bext, boff), PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM, 3, 0, fa[1][3] = PH4_LD(abase1[dx], aoff + 3 * C::ROW_BYTES), fb[1][3] = PH4_LD(bx1[3] + bext, boff), PH4_NOP, PH4_NOP);
      }
      // ---- k-step 1 on set 1; reads of the next tap's k-step 0 into set 0 (the last tap has none: the next tile's halo is
      // published by the barrier that follows)
      if (t + 1 < NTAPS) {
        PH4_GROUP(PH4_MM, 0, 1, fa[0][0] = PH4_LD(abase0[dxn], aoffn + 0 * C::ROW_BYTES), fb[0][0] = PH4_LD(bx0[0] + bextn, boffn), PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM, 1, 1, fa[0][1] = PH4_LD(abase0[dxn], aoffn + 1 * C::ROW_BYTES), fb[0][1] = PH4_LD(bx0[1] + bextn, boffn), PH4_NOP,
                  if (t < C::HALO_TAPS) PH4_DMA_H(2 * t + 1));
        PH4_GROUP(PH4_MM, 2, 1, fa[0][2] = PH4_LD(abase0[dxn], aoffn + 2 * C::ROW_BYTES), fb[0][2] = PH4_LD(bx0[2] + bextn, boffn), PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM, 3, 1, fa[0][3] = PH4_LD(abase0[dxn], aoffn + 3 * C::ROW_BYTES), fb[0][3] = PH4_LD(bx0[3] + bextn, boffn), PH4_NOP, PH4_NOP);
      } else {
        PH4_GROUP(PH4_MM, 0, 1, PH4_NOP, PH4_NOP, PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM, 1, 1, PH4_NOP, PH4_NOP, PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM, 2, 1, PH4_NOP, PH4_NOP, PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM, 3, 1, PH4_NOP, PH4_NOP, PH4_NOP, PH4_NOP);
      }
    }
#undef PH4_DMA_H
  };

  for (int k = 0;; k += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 0) tile_body(0); else tile_body(1);
      epilogue_any(tcur);
      // the next halo (this wave's pieces) has landed, the stores are out: transform, publish, release this tile's buffer
      PH4_WAIT_VMCNT(0);
      if (FUSE_IN && nvalid) xform_halo(half ^ 1);
      PH4_BARRIER();
      if (!nvalid) goto done;
      tcur = tnext;
      tn = tile_id(k + half + 2);
      nvalid = tn >= 0;
      if (nvalid) {
        tnext = decode(tn);
        hm_next = halo_mask(tnext.iy_base, tnext.ix_base);
      }
    }
  }
done:
  if (p.stats) {   // one partial row per workgroup: [2][64] (forward) or [3][64] (fused BatchNorm-backward sums)
    constexpr int NS = BST ? 3 : 2;
    float* red = reinterpret_cast<float*>(smem + C::RED_OFF);   // [4 waves][NS][BNT]
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      float a1 = s1[n], a2 = s2[n], a3 = s3[n];
      a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64); a3 += __shfl_xor(a3, 16, 64);
      a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64); a3 += __shfl_xor(a3, 32, 64);
      if (lg == 0) {
        red[(wave * NS + 0) * BNT + 4 * li + n] = a1;
        red[(wave * NS + 1) * BNT + 4 * li + n] = a2;
        if (NS == 3) red[(wave * NS + 2) * BNT + 4 * li + n] = a3;
      }
    }
    __syncthreads();
    if (tid < NS * BNT) {
      const int which = tid / BNT, n = tid % BNT;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[(w * NS + which) * BNT + n];
      p.stats[((size_t)blockIdx.x * NS + which) * p.Cout + n] = v;
    }
  }
}

template <bool FUSE_IN, int BST>
int launch4(const PhTapConv& p, hipStream_t st) {
  using C = Tap4Cfg;
  auto kern = tapconv4_kernel<FUSE_IN, BST>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
      return PH_ELAUNCH;
    attr_done = true;
  }
  const int total = cdiv(p.OHt, C::TH) * cdiv(p.OWt, C::TW) * p.B;
  const int resident = ph_num_cus();
  dim3 grid(total < resident ? total : resident);
  void* tok = nullptr;
  if (ph_prof_on())
    ph_prof_begin2(p.in_scale ? PH_CLS_TAPCONV2_RES_FUSEDIN : PH_CLS_TAPCONV2_RES, 2.0 * p.B * p.OHt * p.OWt * (double)p.Cout * p.ntaps * p.Cin,
                   ph_tapconv_bytes(p, 1, 2), st, &tok);
  hipLaunchKernelGGL(kern, grid, dim3(C::NTH), C::LDS_BYTES, st, p);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

}  // namespace

// eligible: 3x3 stride-1 perf-mode configuration with Cin = Cout = 64 and the hard-coded 3x3 tap geometry (every ResNet-18 shape
// that reached tapconv2_l1_kernel)
bool ph_tapconv4_eligible(const PhTapConv* p) {
  if (p->ntaps != 9 || p->Cin != 64 || p->Cout != 64 || p->m_groups) return false;
  for (int k = 0; k < 9; ++k)
    if (p->dy[k] != k / 3 || p->dx[k] != k % 3 || p->wtap[k] < 0 || p->wtap[k] > 8) return false;
  if (p->bst_y && (p->in_scale || !p->bst_mean || !p->stats || (p->bst_y2 && !p->bst_mean2))) return false;
  if (p->bst_y && !p->bst_a && (!p->bst_scale || !p->bst_shift)) return false;
  return true;
}

int ph_tapconv4_launch(const PhTapConv* p, hipStream_t st) {
  if (!ph_tapconv4_eligible(p)) return PH_EINVAL;
  if (p->bst_y) return p->bst_a ? launch4<false, 2>(*p, st) : launch4<false, 1>(*p, st);
  return p->in_scale ? launch4<true, 0>(*p, st) : launch4<false, 0>(*p, st);
}
