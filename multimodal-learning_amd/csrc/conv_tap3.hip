// Tap-convolution implicit GEMM, third generation (perf mode bf16, dense 3x3 stride-1, Cout % 128 == 0): the pipeline of
// conv_tap2.hip's tapconv2_kernel<2,2,4,false> - one wave per SIMD, 128 x 64 wave tiles, LDS-DMA operand streams with
// hand-counted vmcnt, persistent XCD-contiguous tile lists - on v_mfma_f32_16x16x32_bf16 fragments.
//
// Why (profiles/EXPERIMENTS.md "MFMA shape"): the 32x32x16 kernel sat at 0.40 of the dense bf16 peak for
// three rounds.  (i) On this part a 16x16x32 stream holds a higher clock than a 32x32x16 stream of the same FLOPs
// (MI355X_MICROARCH.md, DVFS give-back item 7; the timing ablation PH_ABL_MFMA16 measured 5-6 % per launch).  (ii) With
// 16 x 16 tiles the output mapping can be chosen so that a lane owns FOUR CONSECUTIVE channels of one pixel: the weight rows
// of the four N tiles of a wave are interleaved (tile n, row i = channel 4 i + n; the LDS-DMA source mapping does it, the
// LDS image and the fragment reads do not change), so an accumulator quad packs into one 8-byte store and a wave-instruction
// writes 128 contiguous bytes per pixel - ~4 vector instructions per output instead of ~7 (lane-pair DPP exchange, byte
// permute, 4-byte stores in 64-byte runs): the un-overlapped epilogue was 18 % of a layer-2 tile.  (iii) The halo image is
// swizzled by the pixel COLUMN only (chunk bits 1-2 ^ (column / 2) & 3: conflict-free for the three tap alignments of a
// 16-lane ds_read_b128 group, brute-forced), which makes a fragment's LDS address LINEAR in the halo row: the eight M tiles
// of a wave (one tile row each) and the three tap rows are immediate offsets of three per-lane base registers that live for
// the whole kernel - no per-tap address arithmetic for the A side (the 32x32 kernel spent ~28 vector instructions per tap
// on it; with 16x16x32 an MFMA leaves only 8 of its 16 cycles for other issue).
//
// Same GEMM view, descriptor (PhTapConv) and semantics as conv_tap2.hip: forward and stride-1 dgrad with the fused residual
// mask, per-workgroup BatchNorm partial sums, optional BatchNorm + ReLU of the input applied in LDS (in_scale).
#include "ph_common.h"
#include <type_traits>
#include "ph_kernels.h"
#include "tap_common.h"

namespace {

__device__ const u32x4 ph3_zero16[4] = {};
__device__ const u32x4 ph3_nan16[4] = {{0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}, {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u},
                                       {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}, {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}};

typedef __attribute__((address_space(3))) unsigned char lds_uchar;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(g) : "memory");
}
// the same with a wave-uniform base (scalar register pair) and a 32-bit per-lane byte offset: no 64-bit vector add per piece
__device__ __forceinline__ void lds_dma16_s(const unsigned char* sbase, int voff, unsigned lds_addr) {
  // (both are wave-uniform by construction; readfirstlane states it for builds that do not prove it - it folds away at -O3)
  const unsigned long long sb = reinterpret_cast<unsigned long long>(sbase);
  const unsigned long long sbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sb >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)sb);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               : : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(sbu) : "memory");
}
#define PH3_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define PH3_BARRIER() asm volatile("s_barrier" ::: "memory")

// relu(x * s + h) on the 8 bf16 values of one 16-byte chunk (as conv_tap2.hip)
__device__ __forceinline__ u32x4 bn_relu_chunk3(u32x4 v, const f32x4& sA, const f32x4& sB, const f32x4& hA, const f32x4& hB) {
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

struct Tap3Cfg {
  static constexpr int WM = 2, WN = 2, NM = 8, NN = 4, NTAPS = 9, RING = 4;
  static constexpr int TH = 16, TW = 16, BNT = 128;
  static constexpr int HPH = TH + 2, HPW = TW + 2, HP = HPH * HPW;
  static constexpr int ROW_BYTES = (HPW / 2) * 256;                              // one halo row: 9 pixel pairs
  static constexpr int A_BYTES = (((HP + 1) / 2 * 256) + 1023) / 1024 * 1024;    // whole 1-KiB DMA pieces
  static constexpr int NHD = A_BYTES / 1024, NHE = (NHD + 3) / 4;
  static constexpr int TAPB = BNT * 128, NBE = TAPB / 1024 / 4;
  static constexpr int HALO_TAPS = 6, HPT = (NHE + HALO_TAPS - 1) / HALO_TAPS;
  static constexpr int B_BASE = 2 * A_BYTES;
  static constexpr int LDS_MAIN = B_BASE + RING * TAPB;
  static constexpr int SS_OFF = LDS_MAIN + 4096;
  static constexpr int LDS_BYTES = LDS_MAIN + 4096 + 4096;
  static constexpr int NTH = 256;
  static_assert(HPT == 2 && NBE == 4, "vmcnt bookkeeping: 4 weight + 2 halo pieces per wave and tap");
  static_assert(A_BYTES + (NM + 2) * ROW_BYTES < 65536, "ds_read immediate offsets");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// halo image: pixel (hr, hc), 16-byte chunk c of its 64 channels -> LDS byte offset inside an A buffer
__device__ __forceinline__ int a3_off(int hr, int hc, int c) {
  return (Tap3Cfg::HPW / 2 * hr + (hc >> 1)) * 256 + ((hc & 1) << 7) + ((c ^ (((hc >> 1) & 3) << 1)) << 4);
}

// HPM: the half-pair arithmetic (PH_PREC_FP16X3, ph_common.h): `in` is an fp16-pair tensor (per 64-channel slice a 128-B line
// of hi values and one of lo values: element strides x 2), the weights hold per slice the fp16 blocks [hi 2^11 | lo | hi], the K
// loop walks 3 Cin / 64 (A block, W block) pairs (x hi, w hi 2^11), (x hi, w lo), (x lo, w hi) on v_mfma_f32_16x16x32_f16 and the
// epilogue stores fp32: four consecutive channels of a lane = one 16-byte store, 256 contiguous bytes per pixel.
// BST: fused BatchNorm-backward sums over the tensor this (dgrad) launch writes, as in conv_tap4.hip (PhTapConv::bst_y): 0 = none,
// 1 = mask from the BatchNorm's own ReLU (bst_y * bst_scale + bst_shift > 0), 2 = mask (bst_a > 0), 3 = 2 + a second BatchNorm
// (bst_y2) over the same dz; rows [3][Cout] per workgroup.
template <bool FUSE_IN, bool HPM = false, int BST = 0>
__global__ __launch_bounds__(256) void tapconv3_kernel(PhTapConv p) {
  static_assert(!(FUSE_IN && HPM), "the in-LDS BatchNorm + ReLU is a perf-mode feature");
  static_assert(!(BST && (FUSE_IN || HPM)), "the fused BatchNorm-backward sums are a perf-mode dgrad feature");
  using C = Tap3Cfg;
  constexpr int NM = C::NM, NN = C::NN, TH = C::TH, TW = C::TW, BNT = C::BNT, HPW = C::HPW, HP = C::HP, NTAPS = C::NTAPS;
  constexpr int WN = C::WN, WM = C::WM, NTH = C::NTH, B_BASE = C::B_BASE;
  typedef __bf16 T;                    // (2-byte operand element: bf16, or fp16 in the half-pair mode)
  typedef typename std::conditional<HPM, float, __bf16>::type TO;      // output / residual element
  constexpr int EW = HPM ? 2 : 1;      // operand elements per input element
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(lds_uchar*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, lg = lane >> 4;      // MFMA 16x16x32: row / column index, k group (A, B) or pixel group (C)
  const int tiles_w = (p.OWt + TW - 1) / TW;
  const int tiles_sp = tiles_w * ((p.OHt + TH - 1) / TH);
  const int nblk = p.Cout / BNT;
  const int total = tiles_sp * nblk * p.B;
  const long pix_st = (p.in_pix_stride ? p.in_pix_stride : p.Cin) * EW;
  const long row_st = (p.in_row_stride ? p.in_row_stride : (long)p.IW * p.Cin) * EW;
  const long img_st = (p.in_img_stride ? p.in_img_stride : (long)p.IH * p.IW * p.Cin) * EW;
  const bf16* wbase = reinterpret_cast<const bf16*>(p.w);
  const int wK = HPM ? 3 * p.Cin : p.Cin;      // K extent of a packed weight row
  // slice -> element offset of its A block inside a pixel record (half-pair: slices 3c, 3c+1, 3c+2 read the blocks hi, hi, lo
  // of 64-channel group c); its W block is columns 64 sl .. of the weight row
  // (hi-only form, PH_PREC_FP16X1: one slice per 64-channel group: A block hi x W block hi, the third of the group's three)
  const bool hi1 = HPM && p.hp_hi_only;
  auto slice_a = [&](int sl) -> long { return HPM ? (hi1 ? (long)(sl * 128) : (long)((sl / 3) * 128 + ((sl % 3 == 2) ? 64 : 0))) : (long)(sl << 6); };
  auto slice_w = [&](int sl) -> int { return hi1 ? (3 * sl + 2) << 6 : sl << 6; };

  // ---- tile list (as tapconv2_kernel): linear tile id -> (spatial tile fastest, Cout block, image), XCD-contiguous
  struct TileCtx { int r0, c0, n0, b, iy_base, ix_base; const T* in; };
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
    const int tile = t - rest * tiles_sp;
    c.b = fdiv(rest, nblk, rcp_nb);
    c.n0 = (rest - c.b * nblk) * BNT;
    const int trow = fdiv(tile, tiles_w, rcp_tw);
    c.r0 = trow * TH;
    c.c0 = (tile - trow * tiles_w) * TW;
    // (wave-uniform values computed through float arithmetic: pin them to scalar registers - as vector registers they were
    // spilled to scratch, and a scratch reload at a slice start drains the whole LDS-DMA queue with its vmcnt(0))
    c.b = __builtin_amdgcn_readfirstlane(c.b);
    c.n0 = __builtin_amdgcn_readfirstlane(c.n0);
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
  const int nslices = (p.Cin >> 6) * ((HPM && !p.hp_hi_only) ? 3 : 1);      // even (the launcher checks): the A buffer index is compile-time inside a slice pair

  // weight slab of tap t = wtap[0] + t * (wtap[1] - wtap[0]) (the launcher checks: forward 0, 1, .., 8; dgrad 8, 7, .., 0)
  const long slab_bytes = (long)p.Cout * wK * 2;
  const long wtap0 = (long)p.wtap[0] * slab_bytes, wtap_step = (long)(p.wtap[1] - p.wtap[0]) * slab_bytes;

  // ---- per-lane DMA sources.  Weights: piece (wave * NBE + e) of a tap block covers LDS rows 8 q .. 8 q + 7; LDS row R of
  // a wave column's 64-row half holds channel 4 (R & 15) + (R >> 4) of that half (N tile R >> 4, MFMA row R & 15), so that a
  // lane of the MFMA owns four consecutive channels.  The B image keeps the row-pair swizzle of tap_common.h (lds_off).
  int wb_off[C::NBE];
#pragma unroll
  for (int e = 0; e < C::NBE; ++e) {
    const int rp = (wave * C::NBE + e) * 4 + (lane >> 4), u = (lane & 15) ^ (rp & PH_SWZ_MASK);
    const int R = 2 * rp + (u >> 3);
    const int ch = (R & 64) | ((R & 15) << 2) | ((R >> 4) & 3);
    wb_off[e] = (ch * wK + (u & 7) * 8) * 2;
  }
  // halo piece h = wave + 4 e covers row pairs 4 h .. 4 h + 3 of the image; lane l fills slot l & 15 of row pair rp = 4 h + (l >> 4):
  // halo row rp / 9, column 2 (rp % 9) + (slot >> 3), chunk (slot & 7) ^ T(column)
  int h_off[C::NHE];
  auto piece_rc = [&](int e, int& hr, int& hc) {      // halo (row, column) this lane fills in piece wave + 4 e (cheap: recomputed
    int ln = lane;                                     // per tile in halo_mask instead of living in 11 registers)
    asm volatile("" : "+v"(ln));
    const int rp = (wave + 4 * e) * 4 + (ln >> 4), s = ln & 15;
    hr = (rp * 7282) >> 16;                            // rp / 9 for rp < 1024
    const int q = rp - hr * (HPW / 2);
    hc = 2 * q + (s >> 3);
  };
#pragma unroll
  for (int e = 0; e < C::NHE; ++e) {
    const int rp = (wave + 4 * e) * 4 + (lane >> 4), s = lane & 15;
    const int hr = rp / (HPW / 2), q = rp - hr * (HPW / 2);
    const int hc = 2 * q + (s >> 3), ch = (s & 7) ^ ((q & 3) << 1);
    h_off[e] = (int)(((long)hr * row_st + (long)hc * pix_st + ch * 8) * 2);
  }
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
  const unsigned char* zero_src = reinterpret_cast<const unsigned char*>(FUSE_IN ? ph3_nan16 : ph3_zero16);
  // BatchNorm + ReLU of the INPUT applied in LDS: every wave transforms the halo pieces it issued itself
  float* ss = reinterpret_cast<float*>(smem + C::SS_OFF);
  auto xform_halo = [&](int abuf, int k0) {
#pragma unroll
    for (int e = 0; e < C::NHE; ++e)
      if (wave + 4 * e < C::NHD) {
        const int rp = (wave + 4 * e) * 4 + (lane >> 4);
        const int q = rp % (HPW / 2);
        const int cg = (((lane & 7) ^ ((q & 3) << 1)) & 7) * 8;
        const f32x4 sA = *reinterpret_cast<const f32x4*>(ss + k0 + cg), sB = *reinterpret_cast<const f32x4*>(ss + k0 + cg + 4);
        const f32x4 hA = *reinterpret_cast<const f32x4*>(ss + 512 + k0 + cg), hB = *reinterpret_cast<const f32x4*>(ss + 512 + k0 + cg + 4);
        u32x4* a = reinterpret_cast<u32x4*>(smem + abuf * C::A_BYTES + (wave + 4 * e) * 1024 + lane * 16);
        *a = bn_relu_chunk3(*a, sA, sB, hA, hB);
      }
  };

  // ---- per-lane fragment addressing.  A: base of (halo row 8 wm, column li + dx, chunk lg) for dx = 0, 1, 2 and its k-step-1
  // twin (chunk lg + 4 = address ^ 64); M tile m, tap row dy and the A buffer are immediate offsets.  B: row 16 n + li of the
  // wave column's 64-row half, chunk lg.
  int abase0[3], abase1[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    abase0[dx] = a3_off(wm * NM, li + dx, lg);
    abase1[dx] = abase0[dx] ^ 64;
  }
  int bx0[NN], bx1[NN];      // k-step 0 / k-step 1 (chunk lg + 4: address bit 6 flipped; B_BASE and the ring slots are multiples of 128)
#pragma unroll
  for (int n = 0; n < NN; ++n) {
    bx0[n] = B_BASE + lds_off(wn * 64 + n * 16 + li, lg);
    bx1[n] = bx0[n] ^ 64;
  }

  f32x4 acc[NM][NN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int n = 0; n < NN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- BatchNorm partial sums in registers across the tiles of this workgroup (as tapconv2_kernel): a lane owns channels
  // n0 + 64 wn + 4 li + n (n = 0..3)
  constexpr bool Y2 = BST == 3;
  constexpr int NS = BST ? 3 : 2;      // rows per channel block: sum y | sum y^2, or sum dz | sum dz (y - mean) | sum dz (y2 - mean2)
  float* stat_acc = reinterpret_cast<float*>(smem + C::LDS_MAIN);      // [Cout / BNT][NS][BNT]
  for (int i = tid; i < NS * p.Cout; i += NTH) stat_acc[i] = 0.f;
  if (FUSE_IN) {
    for (int i = tid; i < p.Cin; i += NTH) { ss[i] = p.in_scale[i]; ss[512 + i] = p.in_shift[i]; }
    __syncthreads();
  }
  // BST: per-channel constants [mask scale | mask shift | mean | mean2] x 512 behind the three sum rows
  float* bss = reinterpret_cast<float*>(smem + C::LDS_MAIN + 6144);
  if (BST) {
    for (int i = tid; i < p.Cout; i += NTH) {
      bss[i] = (BST == 1) ? p.bst_scale[i] : 0.f;
      bss[512 + i] = (BST == 1) ? p.bst_shift[i] : 0.f;
      bss[1024 + i] = p.bst_mean[i];
      bss[1536 + i] = Y2 ? p.bst_mean2[i] : 0.f;
    }
    __syncthreads();
  }
  float s1[NN], s2[NN], s3[Y2 ? NN : 1];
#pragma unroll
  for (int n = 0; n < NN; ++n) { s1[n] = 0.f; s2[n] = 0.f; s3[Y2 ? n : 0] = 0.f; }
  auto flush_stats = [&](int n0, unsigned char* scratch) {   // all threads; `scratch`: an A buffer nobody reads
    float* red = reinterpret_cast<float*>(scratch);          // [WM][NS][BNT]
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      float a1 = s1[n], a2 = s2[n], a3 = s3[Y2 ? n : 0];
      a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64);
      a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
      if (Y2) { a3 += __shfl_xor(a3, 16, 64); a3 += __shfl_xor(a3, 32, 64); } else a3 = 0.f;
      if (lg == 0) {
        red[(wm * NS + 0) * BNT + wn * 64 + 4 * li + n] = a1;
        red[(wm * NS + 1) * BNT + wn * 64 + 4 * li + n] = a2;
        if (NS == 3) red[(wm * NS + 2) * BNT + wn * 64 + 4 * li + n] = a3;
      }
      s1[n] = 0.f; s2[n] = 0.f; s3[Y2 ? n : 0] = 0.f;
    }
    __syncthreads();
    for (int i = tid; i < NS * BNT; i += NTH) {
      const int which = i / BNT, n = i % BNT;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(w * NS + which) * BNT + n];
      stat_acc[((n0 / BNT) * NS + which) * BNT + n] += v;
    }
    __syncthreads();
  };

  // ---- epilogue of one tile.  Accumulator register r of tile (m, n) is pixel (row wm * 8 + m, column 4 lg + r), channel
  // n0 + 64 wn + 4 li + n: the four N tiles give four consecutive channels = one 8-byte store per (m, r); a wave-instruction
  // writes 128 contiguous bytes for each of four pixels.
  auto epilogue = [&](const TileCtx& tc, auto fullc, auto rmc) {
    constexpr bool FULL = decltype(fullc)::value;
    constexpr int RM = decltype(rmc)::value;
    const size_t img = (size_t)tc.b * p.OH * p.OW * p.Cout;
    // per-image base pointers are wave-uniform; the per-lane part is a 32-bit element offset (one image < 2^31 elements)
    TO* out = reinterpret_cast<TO*>(p.out) + img;
    const TO* resg = reinterpret_cast<const TO*>(p.res_g) + img;
    const TO* resa = reinterpret_cast<const TO*>(p.res_a) + img;
    const unsigned chan = (unsigned)(tc.n0 + wn * 64 + 4 * li);
    const unsigned colstep = (unsigned)(p.os * p.Cout);
    // element offset of (tile row 0 of this wave, column 4 lg) - rows advance by rowstep, columns by colstep
    const unsigned o00 = (unsigned)(((tc.r0 + wm * NM) * p.os + p.oa_h) * p.OW + (tc.c0 + 4 * lg) * p.os + p.oa_w) * (unsigned)p.Cout + chan;
    const unsigned rowstep = (unsigned)(p.os * p.OW * p.Cout);
    const float osc = HPM ? (p.hp_hi_only ? 1.f : PH_HP_LO_INV) * (p.in_unscale ? p.in_unscale[1] : 1.f) : 1.f;
    const T* bsty = reinterpret_cast<const T*>(p.bst_y) + img;
    const T* bsta = reinterpret_cast<const T*>(p.bst_a) + img;
    const T* bsty2 = reinterpret_cast<const T*>(p.bst_y2) + img;
    constexpr bool has_y2 = Y2;
    f32x4 cms = {0.f, 0.f, 0.f, 0.f}, cmh = cms, cmu = cms, cmu2 = cms;
    if constexpr (BST != 0) {
      if constexpr (BST == 1) { cms = *reinterpret_cast<const f32x4*>(bss + chan); cmh = *reinterpret_cast<const f32x4*>(bss + 512 + chan); }
      cmu = *reinterpret_cast<const f32x4*>(bss + 1024 + chan);
      cmu2 = *reinterpret_cast<const f32x4*>(bss + 1536 + chan);
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int r = tc.r0 + wm * NM + m;
      const unsigned orow = o00 + (unsigned)m * rowstep;
      if constexpr (HPM) {
        // fp32 output: accumulators x 2^-11 (the W blocks' scaling) x the dz tensor's un-scale; residual / mask operands are fp32
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = tc.c0 + 4 * lg + q;
          const bool mine = FULL || (r < p.OHt && c < p.OWt);
          f32x4 v;
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            v[n] = acc[m][n][q] * osc;
            if constexpr (!FULL) v[n] = mine ? v[n] : 0.f;
            s1[n] += v[n];
            s2[n] = __builtin_fmaf(v[n], v[n], s2[n]);
          }
          if constexpr (RM > 0) {
            if (mine) {
              const f32x4 g = *reinterpret_cast<const f32x4*>(resg + (orow + (unsigned)q * colstep));
              if constexpr (RM > 1) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(resa + (orow + (unsigned)q * colstep));
#pragma unroll
                for (int n = 0; n < NN; ++n) v[n] += a[n] > 0.f ? g[n] : 0.f;
              } else {
                v += g;
              }
            }
          }
          if (mine) *reinterpret_cast<f32x4*>(out + (orow + (unsigned)q * colstep)) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      // (with the fused sums a row's loads are taken two pixels at a time: five 8-byte operands per pixel and lane)
      constexpr int QB = BST ? 2 : 4;
#pragma unroll
      for (int q0 = 0; q0 < 4; q0 += QB) {
      u32x2 rg[4], ra[4], ry[4], rb[4], ry2[4];
      if constexpr (RM > 0 || BST != 0) {
#pragma unroll
        for (int q = q0; q < q0 + QB; ++q) {
          const int c = tc.c0 + 4 * lg + q;
          rg[q] = u32x2{0u, 0u};
          ra[q] = u32x2{0x3f803f80u, 0x3f803f80u};
          ry[q] = u32x2{0u, 0u}; rb[q] = u32x2{0u, 0u}; ry2[q] = u32x2{0u, 0u};
          if (FULL || (r < p.OHt && c < p.OWt)) {
            const unsigned o = orow + (unsigned)q * colstep;
            if constexpr (RM > 0) rg[q] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(resg) + o);
            if constexpr (RM > 1) ra[q] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(resa) + o);
            if constexpr (BST != 0) {
              ry[q] = *reinterpret_cast<const u32x2*>(bsty + o);
              if constexpr (BST >= 2) rb[q] = *reinterpret_cast<const u32x2*>(bsta + o);
              if constexpr (has_y2) ry2[q] = *reinterpret_cast<const u32x2*>(bsty2 + o);
            }
          }
        }
      }
#pragma unroll
      for (int q = q0; q < q0 + QB; ++q) {
        const int c = tc.c0 + 4 * lg + q;
        const bool mine = FULL || (r < p.OHt && c < p.OWt);
        float v[4];
#pragma unroll
        for (int n = 0; n < NN; ++n) {
          v[n] = acc[m][n][q];
          if constexpr (!FULL) v[n] = mine ? v[n] : 0.f;
          if constexpr (BST == 0) {
            s1[n] += v[n];
            s2[n] = __builtin_fmaf(v[n], v[n], s2[n]);
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
              if constexpr (BST == 1) on = __builtin_fmaf(y, cms[n], cmh[n]) > 0.f;
              else on = __builtin_bit_cast(float, e ? (rb[q][h] & 0xffff0000u) : (rb[q][h] << 16)) > 0.f;
              float dz = (float)b[e];
              dz = (on && mine) ? dz : 0.f;
              s1[n] += dz;
              s2[n] = __builtin_fmaf(dz, y - cmu[n], s2[n]);
              if constexpr (has_y2) {
                const float y2 = __builtin_bit_cast(float, e ? (ry2[q][h] & 0xffff0000u) : (ry2[q][h] << 16));
                s3[has_y2 ? n : 0] = __builtin_fmaf(dz, y2 - cmu2[n], s3[has_y2 ? n : 0]);
              }
            }
          }
        }
        if (mine) *reinterpret_cast<u32x2*>(reinterpret_cast<T*>(out) + (orow + (unsigned)q * colstep)) = w;
      }
      if constexpr (BST != 0) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);      // one tile row at a time: hoisting all 64 residual loads in front spills
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

  // ---- the tap stream (tile -> 64-channel slice -> 9 taps, fully unrolled), bookkeeping as in tapconv2_kernel: every tap
  // issues NBE weight pieces (tap +3 of the stream) and, in taps 0..5, 2 pieces of the next slice's halo.
  TileCtx tcur = decode(tile_id(0));
  int tn = tile_id(1);
  bool nvalid = tn >= 0;
  TileCtx tnext = tcur;
  if (nvalid) tnext = decode(tn);
  int hm_cur = halo_mask(tcur.iy_base, tcur.ix_base);
  int hm_next = nvalid ? halo_mask(tnext.iy_base, tnext.ix_base) : hm_cur;
  auto halo_base = [&](const TileCtx& tc, long off) {
    return reinterpret_cast<const unsigned char*>(tc.in) + ((long)tc.iy_base * row_st + (long)tc.ix_base * pix_st + off) * 2;
  };
  auto w_base = [&](int n0, int k0, int tap) {
    return reinterpret_cast<const unsigned char*>(wbase) + wtap0 + (long)tap * wtap_step + ((long)n0 * wK + k0) * 2;
  };

  {  // prologue: first halo and the first RING-1 taps of weights
    const unsigned char* hb = halo_base(tcur, 0);
#pragma unroll
    for (int e = 0; e < C::NHE; ++e)
      if (wave + 4 * e < C::NHD)
        lds_dma16(((hm_cur >> e) & 1) ? hb + h_off[e] : zero_src, lds0 + (wave + 4 * e) * 1024);
#pragma unroll
    for (int j = 0; j < C::RING - 1; ++j) {
      const unsigned char* wb = w_base(tcur.n0, slice_w(0), j);
#pragma unroll
      for (int e = 0; e < C::NBE; ++e)
        lds_dma16_s(wb, wb_off[e], lds0 + B_BASE + j * C::TAPB + (wave * C::NBE + e) * 1024);
    }
  }
  zero_acc();
  PH3_WAIT_VMCNT(0);
  if (FUSE_IN) xform_halo(0, 0);
  PH3_BARRIER();

  // Fragment registers: A ring of 4 (M tile m of a k-step is read two MFMA groups ahead), B double buffered.  One k-step =
  // 8 M tiles x 4 N tiles = 32 MFMAs of 16 cycles; each M group's four MFMAs carry one A read (tile m + 2; the last two groups
  // read tiles 0 and 1 of the NEXT k-step), groups 0..3 also one B read of the next k-step.  The MFMAs are inline asm with
  // the accumulators pinned to AGPRs ("+a"), a sched_barrier after every slot keeps the order (see conv_tap2.hip).
  bf16x8 fa[4], fb[2][NN];
#define PH3_MM(M, N, AI, BS)                                                                                        \
  do {                                                                                                              \
    if constexpr (HPM) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[M][N]) : "v"(fa[AI]), "v"(fb[BS][N]));  \
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[M][N]) : "v"(fa[AI]), "v"(fb[BS][N]));  \
  } while (0)
#define PH3_LD(ADDR, IMM) (*reinterpret_cast<const bf16x8*>(smem + (ADDR) + (IMM)))
#define PH3_SB() __builtin_amdgcn_sched_barrier(0)
  // A read of M tile MT (0..7) of the k-step whose bases are (AB0, AB1)[KS] with immediate offset AOFF (buffer + tap row)
#define PH3_LDA(AB, AOFF, MT) PH3_LD(AB, (AOFF) + (MT) * C::ROW_BYTES)
  // one group: 4 MFMAs of M tile M with A ring slot M & 3 and B set BS; RA = statement issuing this group's A read, RB = its B read
#define PH3_GROUP(M, BS, RA, RB, F0, F1)                \
  PH3_MM(M, 0, (M) & 3, BS); RA; PH3_SB();              \
  PH3_MM(M, 1, (M) & 3, BS); RB; PH3_SB();              \
  PH3_MM(M, 2, (M) & 3, BS); F0; PH3_SB();              \
  PH3_MM(M, 3, (M) & 3, BS); F1; PH3_SB()
#define PH3_NOP ((void)0)

  unsigned bslot_cur = 0;      // LDS byte offset of the ring slot of the tap in flight (gt & 3) * TAPB
  int gt = 0;
  // first fragments of the stream: A tiles 0, 1, 2 and the four B tiles of tap 0 / k-step 0
  fa[0] = PH3_LDA(abase0[0], 0, 0);
  fa[1] = PH3_LDA(abase0[0], 0, 1);
  fa[2] = PH3_LDA(abase0[0], 0, 2);
#pragma unroll
  for (int n = 0; n < NN; ++n) fb[0][n] = PH3_LD(bx0[n], 0);

  for (int k = 0;; ++k) {   // tiles of this workgroup
    // one slice; ABUF = A buffer index - a literal at both call sites (slices come in pairs), so that after inlining every
    // fragment address below is a base register + an immediate.  (Not a generic lambda: clang rejects asm operands that name
    // captured variables inside one.)
    auto slice_body = [&](const int sl, const int ABUF) __attribute__((always_inline)) {
      const int AOFF_CUR = ABUF * C::A_BYTES, AOFF_NXT = (ABUF ^ 1) * C::A_BYTES;
      const bool last_sl = sl + 1 == nslices;
      const bool h_next_tile = last_sl && nvalid;
      const int nsl = last_sl ? 0 : sl + 1;
      const unsigned char* hb = h_next_tile ? halo_base(tnext, 0) : halo_base(tcur, slice_a(nsl));
      const int hm = h_next_tile ? hm_next : hm_cur;
      const int wn0 = (last_sl && nvalid) ? tnext.n0 : tcur.n0;
      const int wk0 = slice_w(nsl);
      const unsigned char* wcur = w_base(tcur.n0, slice_w(sl), 0);      // tap 0 of this slice's weight block / of the next one's
      const unsigned char* wnxt = w_base(wn0, wk0, 0);
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        const int dy = t / 3, dx = t % 3;
        const int tn_ = (t + 1) % NTAPS, dyn = tn_ / 3, dxn = tn_ % 3;
        // immediate offsets of this tap's / the next tap's A fragments (the next tap of tap 8 reads the other buffer)
        const int aoff = AOFF_CUR + dy * C::ROW_BYTES;
        const int aoffn = (t + 1 == NTAPS ? AOFF_NXT : AOFF_CUR) + dyn * C::ROW_BYTES;
        // weight pieces of stream tap gt+3 -> ring slot (gt+3) & 3
        const unsigned char* wb = (t + 3 < NTAPS) ? wcur + (t + 3) * wtap_step : wnxt + (t + 3 - NTAPS) * wtap_step;
        const unsigned wdst = lds0 + B_BASE + ((gt + 3) & 3) * C::TAPB + wave * C::NBE * 1024;
        const unsigned hdst = lds0 + (ABUF ^ 1) * C::A_BYTES + wave * 1024;
        const unsigned bslot_nxt = ((gt + 1) & 3) * C::TAPB;
#define PH3_DMA_B(E) lds_dma16_s(wb, wb_off[E], wdst + (E) * 1024)
#define PH3_DMA_H(E)                                                                                          \
  do {                                                                                                        \
    if ((E) < C::NHE && wave + 4 * (E) < C::NHD)                                                              \
      lds_dma16(((hm >> (E)) & 1) ? hb + h_off[(E) < C::NHE ? (E) : 0] : zero_src, hdst + (E) * 4096);        \
  } while (0)
        // ---- k-step 0 (chunks lg): A tile m + 3 is read by group m (tiles 3..7 of this k-step, then 0..2 of k-step 1); B of k-step 1
        PH3_GROUP(0, 0, fa[3] = PH3_LDA(abase0[dx], aoff, 3), fb[1][0] = PH3_LD(bx1[0], bslot_cur), PH3_NOP, PH3_DMA_B(0));
        PH3_GROUP(1, 0, fa[0] = PH3_LDA(abase0[dx], aoff, 4), fb[1][1] = PH3_LD(bx1[1], bslot_cur), PH3_NOP, PH3_NOP);
        PH3_GROUP(2, 0, fa[1] = PH3_LDA(abase0[dx], aoff, 5), fb[1][2] = PH3_LD(bx1[2], bslot_cur), PH3_NOP, PH3_DMA_B(1));
        PH3_GROUP(3, 0, fa[2] = PH3_LDA(abase0[dx], aoff, 6), fb[1][3] = PH3_LD(bx1[3], bslot_cur), PH3_NOP, PH3_NOP);
        PH3_GROUP(4, 0, fa[3] = PH3_LDA(abase0[dx], aoff, 7), PH3_NOP, PH3_NOP, PH3_DMA_B(2));
        PH3_GROUP(5, 0, fa[0] = PH3_LDA(abase1[dx], aoff, 0), PH3_NOP, PH3_NOP, PH3_NOP);
        PH3_GROUP(6, 0, fa[1] = PH3_LDA(abase1[dx], aoff, 1), PH3_NOP, PH3_NOP, PH3_DMA_B(3));
        PH3_GROUP(7, 0, fa[2] = PH3_LDA(abase1[dx], aoff, 2), PH3_NOP, PH3_NOP, PH3_NOP);
        // ---- k-step 1 (chunks lg + 4): A tiles 3..7, then tiles 0..2 of the NEXT tap's k-step 0; B of the next tap in groups 4..7
        // (the two halo pieces are the LAST operations a tap issues: in a tap where a wave has none left - piece index past the
        // image - the end-of-tap wait below lets two more operations of the PREVIOUS tap stay in flight, and those must be its
        // halo pieces, which nobody reads before tap 6's wait, never weights that the next tap's early fragment reads need)
        PH3_GROUP(0, 1, fa[3] = PH3_LDA(abase1[dx], aoff, 3), PH3_NOP, PH3_NOP, if (t < C::HALO_TAPS) PH3_DMA_H(2 * t));
        PH3_GROUP(1, 1, fa[0] = PH3_LDA(abase1[dx], aoff, 4), PH3_NOP, PH3_NOP, PH3_NOP);
        PH3_GROUP(2, 1, fa[1] = PH3_LDA(abase1[dx], aoff, 5), PH3_NOP, PH3_NOP, if (t < C::HALO_TAPS) PH3_DMA_H(2 * t + 1));
        PH3_GROUP(3, 1, fa[2] = PH3_LDA(abase1[dx], aoff, 6), PH3_NOP, PH3_NOP, PH3_NOP);
        // The reads below are the first of the NEXT tap: weights of ring slot (gt + 1) & 3 were published one tap ago; the next
        // slice's halo (read by tap 8) is complete at tap 6's wait and published by the barriers of taps 6 and 7.
        PH3_GROUP(4, 1, fa[3] = PH3_LDA(abase1[dx], aoff, 7), fb[0][0] = PH3_LD(bx0[0], bslot_nxt), PH3_NOP, PH3_NOP);
        PH3_GROUP(5, 1, fa[0] = PH3_LDA(abase0[dxn], aoffn, 0), fb[0][1] = PH3_LD(bx0[1], bslot_nxt), PH3_NOP, PH3_NOP);
        PH3_GROUP(6, 1, fa[1] = PH3_LDA(abase0[dxn], aoffn, 1), fb[0][2] = PH3_LD(bx0[2], bslot_nxt), PH3_NOP, PH3_NOP);
        PH3_GROUP(7, 1, fa[2] = PH3_LDA(abase0[dxn], aoffn, 2), fb[0][3] = PH3_LD(bx0[3], bslot_nxt), PH3_NOP, PH3_NOP);
        // ---- tap end.  The weight pieces of stream tap gt+2 (and a halo that is due) have landed once at most the pieces issued
        // during this tap are still in flight; the barrier publishes them and releases ring slot gt & 3.
        if (t < C::HALO_TAPS) PH3_WAIT_VMCNT(6); else PH3_WAIT_VMCNT(4);
        if (t == C::HALO_TAPS) { if (FUSE_IN) xform_halo(ABUF ^ 1, wk0); }
        PH3_BARRIER();
        bslot_cur = bslot_nxt;
        ++gt;
      }
    };
    for (int sl = 0; sl < nslices; sl += 2) {
      slice_body(sl, 0);
      slice_body(sl + 1, 1);
    }
    epilogue_any(tcur);
    if (p.stats && (!nvalid || tnext.n0 != tcur.n0)) flush_stats(tcur.n0, smem + C::A_BYTES);
    zero_acc();
    if (!nvalid) break;
    tcur = tnext;
    hm_cur = hm_next;
    tn = tile_id(k + 2);
    nvalid = tn >= 0;
    if (nvalid) {
      tnext = decode(tn);
      hm_next = halo_mask(tnext.iy_base, tnext.ix_base);
    }
  }
  PH3_WAIT_VMCNT(0);   // the refills issued past the end of the stream must not outlive the workgroup's LDS
  if (p.stats) {
    for (int i = tid; i < NS * p.Cout; i += NTH) {
      const int which = i / p.Cout, ch = i - which * p.Cout;
      p.stats[((size_t)blockIdx.x * NS + which) * p.Cout + ch] = stat_acc[((ch / BNT) * NS + which) * BNT + ch % BNT];
    }
  }
}

template <bool FUSE_IN, bool HPM = false, int BST = 0>
int launch3(const PhTapConv& p, hipStream_t st) {
  using C = Tap3Cfg;
  auto kern = tapconv3_kernel<FUSE_IN, HPM, BST>;
  // (the fused sums keep three rows of sums and four rows of per-channel constants behind the operand buffers: all 160 KiB)
  constexpr int LDS_BYTES = BST ? C::LDS_MAIN + 6144 + 8192 : C::LDS_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return PH_ELAUNCH;
    attr_done = true;
  }
  const int total = cdiv(p.OHt, C::TH) * cdiv(p.OWt, C::TW) * (p.Cout / C::BNT) * p.B;
  const int resident = ph_num_cus();
  dim3 grid(total < resident ? total : resident);
  void* tok = nullptr;
  if (ph_prof_on())
    ph_prof_begin2(p.in_scale ? PH_CLS_TAPCONV2_FUSEDIN : PH_CLS_TAPCONV2, 2.0 * p.B * p.OHt * p.OWt * (double)p.Cout * p.ntaps * p.Cin,
                   ph_tapconv_bytes(p, 1, HPM ? 4 : 2), st, &tok);
  hipLaunchKernelGGL(kern, grid, dim3(C::NTH), LDS_BYTES, st, p);
  ph_prof_end(tok, st);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

}  // namespace

// eligible: the dense 3x3 stride-1 perf-mode configuration with an even number of 64-channel slices (every ResNet-18 shape
// that reaches tapconv2_kernel<2,2,4,false>: Cin = 128 / 256 / 512)
bool ph_tapconv3_eligible(const PhTapConv* p) {
  const int step = p->wtap[1] - p->wtap[0];
  for (int k = 0; k < 9; ++k)      // the kernel addresses the tap's weight slab as wtap[0] + k * step, and hard-codes the 3x3 geometry
    if (p->ntaps != 9 || p->wtap[k] != p->wtap[0] + k * step || p->dy[k] != k / 3 || p->dx[k] != k % 3) return false;
  return p->Cout % 128 == 0 && p->Cout <= 512 && !p->m_groups && p->ntaps == 9 && ((p->Cin >> 6) & 1) == 0 && p->Cin >= 128 &&
         (!p->in_scale || p->Cin <= 512);
}

int ph_tapconv3_launch(const PhTapConv* p, hipStream_t st) {
  if (!ph_tapconv3_eligible(p)) return PH_EINVAL;
  if (p->bst_y) {      // fused BatchNorm-backward sums (dgrad launches)
    if (p->in_scale || !p->stats || !p->bst_mean || (p->bst_y2 && !p->bst_mean2) || (!p->bst_a && (!p->bst_scale || !p->bst_shift)))
      return PH_EINVAL;
    if (!p->bst_a) return p->bst_y2 ? PH_EINVAL : launch3<false, false, 1>(*p, st);
    return p->bst_y2 ? launch3<false, false, 3>(*p, st) : launch3<false, false, 2>(*p, st);
  }
  return p->in_scale ? launch3<true>(*p, st) : launch3<false>(*p, st);
}

// the same kernel in the half-pair arithmetic (PH_PREC_FP16X3): 3 Cin / 64 slices (even for Cin = 128 / 256 / 512), no in-LDS BatchNorm
int ph_tapconv3_launch_hp(const PhTapConv* p, hipStream_t st) {
  if (!ph_tapconv3_eligible(p) || p->in_scale) return PH_EINVAL;
  return launch3<false, true>(*p, st);
}
