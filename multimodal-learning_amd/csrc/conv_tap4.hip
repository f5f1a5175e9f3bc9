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
//   * the epilogue of tile k rides in the MFMA gaps of tile k + 1 (two accumulator sets, OVL below); the tile switch is scalar
//     arithmetic; the end-of-tile wait counts the halo pieces only, not the stores' acknowledgements.
// Optional fused BatchNorm-backward sums (PhTapConv::bst_y): a dgrad launch that produces the gradient a
// BatchNorm backward reduces over can take the per-channel sums sum dz and sum dz (y - mean) in its epilogue - the ReLU mask and
// y arrive as 8-byte loads in the accumulator layout, requested while the tile's last taps run - and the separate
// bn_bwd_reduce pass (a full read of the gradient, y and the mask tensor) disappears from the dgrad chain.
//
// Same GEMM view, descriptor (PhTapConv) and semantics as conv_tap2.hip / conv_tap3.hip; outputs BITWISE those of
// tapconv2_l1_kernel (same products, same fp32 accumulation order per output: slices of 32 channels, taps in order).
#include <mutex>
#include "ph_common.h"
#include <cstdlib>
#include <type_traits>
#include "ph_kernels.h"
#include "tap_common.h"

namespace {

__device__ const u32x4 ph4_zero16[4] = {};
__device__ const u32x4 ph4_nan16[4] = {{0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}, {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u},
                                       {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}, {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}};

typedef __attribute__((address_space(3))) unsigned char lds_uchar4;

#ifdef PH_TAP_TRACE      // debug build only (make trace): per-workgroup cycle accounts, read back by tests/trace_tapconv4_gpu.py
__device__ unsigned long long ph_tap4_trace[1024 * 8];
#define PH4_CLK() clock64()
#define PH4_TR(k, v) do { if (threadIdx.x == 0 && blockIdx.x < 1024) ph_tap4_trace[blockIdx.x * 8 + (k)] = (v); } while (0)
#else
#define PH4_CLK() 0ull
#define PH4_TR(k, v)
#endif

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
// residual writes); 3: as 2 plus the second BatchNorm (bst_y2: a downsample branch reducing the same dz).  Row [3][64] per
// workgroup: sum dz, sum dz (bst_y - bst_mean), sum dz (bst_y2 - bst_mean2) (0 without bst_y2).
//
// OVL: the epilogue of tile k runs INSIDE the tap stream of tile k + 1 (two accumulator sets): one 8-byte store and its ~17
// vector instructions per k-step, cut into 8 micro-steps that sit in the MFMA gaps.  One after the other, the stores of a tile
// and its 288 MFMAs per wave made the first version of this kernel 91 us, no faster than the kernel it replaces.  OVL launches
// read nothing in their epilogue (forward: no residual, no fused sums); launches that do (dgrad) run the epilogue after the
// tile (OVL = false), their operands requested two taps earlier.
//
// OVL with operands (ORM = 2: dgrad + masked residual; BST = 1: own-ReLU fused sums): the operands of a piece are 8-byte loads
// issued PD k-steps ahead of the micro-steps that consume them - pieces PD .. 15 from inside the stream that stores them (a
// rotating window of PD + 1 register slots), pieces 0 .. PD - 1 ("head") from taps 6-7 of the tile's OWN stream, so that they
// have landed before the tile-end wait and nothing is in flight at the barrier.  The loads are compiler-visible builtins: its
// wait in front of a consumer counts only the operations it knows (later loads and stores), so the inline-asm LDS-DMAs issued
// since make that wait stricter by at most the few DMAs of PD k-steps - bounded, unlike the all-sixteen-pieces-after-the-tile
// form, whose first consumer waited for everything.
template <bool FUSE_IN, int BST, bool OVL, int ORM = 0>
__global__ __launch_bounds__(256) void tapconv4_kernel(PhTapConv p) {
  static_assert(!(FUSE_IN && BST), "forward-only feature vs backward feature");
  static_assert(!(OVL && BST > 1), "the overlapped epilogue carries at most one operand pair");
  static_assert(ORM == 0 || (ORM == 2 && OVL && !BST && !FUSE_IN), "ORM: overlapped dgrad + masked residual");
  constexpr bool OPS = OVL && (ORM != 0 || BST == 1);      // overlapped epilogue WITH operands
  #ifndef PH4_PD
#define PH4_PD 4
#endif
  constexpr int PD = PH4_PD, NWIN = PD + 1, NSL = PD + NWIN;
  using C = Tap4Cfg;
  constexpr int NM = C::NM, NN = C::NN, TH = C::TH, TW = C::TW, BNT = C::BNT, HPW = C::HPW, NTAPS = C::NTAPS;
  constexpr int B_BASE = C::B_BASE;
  typedef __bf16 T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(lds_uchar4*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = the wave's band of four tile rows
  const int li = lane & 15, lg = lane >> 4;                       // MFMA 16x16x32: row / column index, k group (A, B) or row group (C)
  const int tiles_w = (p.OWt + TW - 1) / TW;
  const int tiles_sp = tiles_w * ((p.OHt + TH - 1) / TH);
  const int total = tiles_sp * p.B;
  const long pix_st = p.in_pix_stride ? p.in_pix_stride : p.Cin;
  const long row_st = p.in_row_stride ? p.in_row_stride : (long)p.IW * p.Cin;
  const long img_st = p.in_img_stride ? p.in_img_stride : (long)p.IH * p.IW * p.Cin;
  const bf16* wbase = reinterpret_cast<const bf16*>(p.w);

  // ---- tile list: linear tile id -> (spatial tile fastest, image), XCD-contiguous (as conv_tap3.hip)
  // A tile switch used to cost ~2400 cycles per tile behind the barrier (float-reciprocal divisions, a bit mask of the halo
  // pieces built from recomputed coordinates - cheap against the dense kernel's 23 k-cycle tiles, a quarter of this kernel's):
  // the decode is wave-uniform integer arithmetic (multiply-high by a precomputed reciprocal: scalar instructions, which issue
  // beside the MFMAs), and a halo piece's validity is tested from its packed coordinates when it is issued.
  struct TileCtx { int r0, c0, b, iy_base, ix_base; const T* in; };
  const unsigned m_sp = (unsigned)((0x100000000ull + (unsigned)tiles_sp - 1) / (unsigned)tiles_sp);      // t / d = (t * ceil(2^32 / d)) >> 32
  const unsigned m_tw = (unsigned)((0x100000000ull + (unsigned)tiles_w - 1) / (unsigned)tiles_w);       // (exact for t, d < 2^16)
  auto udiv = [](unsigned a, unsigned d, unsigned m) -> int { return d == 1 ? (int)a : (int)__umulhi(a, m); };
  auto decode = [&](int t) -> TileCtx {
    TileCtx c;
    c.b = udiv((unsigned)t, (unsigned)tiles_sp, m_sp);
    const int tile = t - c.b * tiles_sp;
    const int trow = udiv((unsigned)tile, (unsigned)tiles_w, m_tw);
    c.r0 = trow * TH;
    c.c0 = (tile - trow * tiles_w) * TW;
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
  // packed halo coordinates of the element this lane fills in piece wave + 4 e: row | column << 8 (row >= HPH: past the image)
  int h_rc[C::NHE];
#pragma unroll
  for (int e = 0; e < C::NHE; ++e) {
    const int rp = (wave + 4 * e) * 4 + (lane >> 4), s_ = lane & 15;
    const int hr = rp / (HPW / 2), q = rp - hr * (HPW / 2);
    h_rc[e] = hr | ((2 * q + (s_ >> 3)) << 8);
  }
  auto piece_ok = [&](int e, int iy_base, int ix_base) -> bool {
    const int hr = h_rc[e] & 0xff, hc = h_rc[e] >> 8;
    return hr < C::HPH && (unsigned)(iy_base + hr) < (unsigned)p.IH && (unsigned)(ix_base + hc) < (unsigned)p.IW;
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

  // ---- per-lane fragment addressing.  Pixels: base of (halo row 4 wave, column li + dx, chunk lg) for dx = 0, 1, 2 and its
  // k-step-1 twin (chunk lg + 4 = address ^ 64); tile row, tap row dy and the halo buffer are immediate offsets.  Weights: LDS row
  // 16 t + li of the tap's 64 rows, chunk lg; the tap is an immediate (tap 8 = 7 taps past a second base: 8 * TAPB exceeds the
  // 16-bit offset field).
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

  // ---- sums over all tiles of this workgroup.  Accumulator register r of (pixel tile row m, channel tile n) is pixel (row
  // 4 wave + m, column 4 lg + r), LDS weight row 16 n + li; the DMA source mapping puts channel 4 i + n in LDS row 16 n + i, so a
  // lane owns the four consecutive channels 4 li .. 4 li + 3: one 8-byte store per (m, r), and the 16 lanes of a quarter wave
  // write one whole 128-byte pixel record (a 16-byte-per-lane layout - lane = pixel - was tried: every wave instruction then
  // touches 16 cache lines in 16-byte pieces and costs ~450 cycles to issue, loads and stores alike).
  float s1[NN], s2[NN], s3[BST == 3 ? NN : 1];
#pragma unroll
  for (int n = 0; n < NN; ++n) { s1[n] = 0.f; s2[n] = 0.f; s3[BST == 3 ? n : 0] = 0.f; }

  // ---- prologue: per-channel constants, all nine taps of weights, the first halo
  if (FUSE_IN) {
    if (tid < 2 * BNT) ss[tid] = tid < BNT ? p.in_scale[tid] : p.in_shift[tid - BNT];
  }
  if (BST) {   // [mask scale | mask shift | mean | mean2]
    if (tid < BNT) {
      ss[tid] = (BST == 1) ? p.bst_scale[tid] : 0.f;
      ss[BNT + tid] = (BST == 1) ? p.bst_shift[tid] : 0.f;
      ss[2 * BNT + tid] = p.bst_mean[tid];
      ss[3 * BNT + tid] = (BST == 3) ? p.bst_mean2[tid] : 0.f;
    }
  }
  if (FUSE_IN || BST) __syncthreads();
  {
    // weight piece q = wave * NWP + j of the 72: tap q / 8, LDS rows 8 (q % 8) .. + 7 of the tap block
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
  auto halo_base = [&](const TileCtx& tc) {
    return reinterpret_cast<const unsigned char*>(tc.in) + ((long)tc.iy_base * row_st + (long)tc.ix_base * pix_st) * 2;
  };
  {
    const unsigned char* hb = halo_base(tcur);
#pragma unroll
    for (int e = 0; e < C::NHE; ++e)
      if (wave + 4 * e < C::NHD)
        lds_dma16_4(piece_ok(e, tcur.iy_base, tcur.ix_base) ? hb + h_off[e] : zero_src, lds0 + (wave + 4 * e) * 1024);
  }
  PH4_WAIT_VMCNT(0);
  if (FUSE_IN) { xform_halo(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  PH4_BARRIER();

  // two accumulator sets (OVL: the tile in the MFMAs and the tile being stored), pinned to AGPRs; acc[set][pixel row m][channel tile n]
  f32x4 acc[OVL ? 2 : 1][NM][NN];
  bf16x8 fa[2][NM], fb[2][NN];
#define PH4_MM(AS, M, N, S) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[AS][M][N]) : "v"(fa[S][M]), "v"(fb[S][N]))
#define PH4_MM0(AS, M, N, S) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc[AS][M][N]) : "v"(fa[S][M]), "v"(fb[S][N]))
#define PH4_LD(ADDR, IMM) (*reinterpret_cast<const bf16x8*>(smem + (ADDR) + (IMM)))
#define PH4_SB() __builtin_amdgcn_sched_barrier(0)
#define PH4_NOP ((void)0)
  // one pixel-row group of a k-step on fragment set S: four MFMAs (channel tiles 0..3); RA / RB = the reads of the NEXT k-step's
  // pixel fragment / weight fragment of the same index into set S ^ 1; F0 / F1 = filler slots
#define PH4_GROUP(MMAC, AS, M, S, RA, RB, F0, F1)   \
  MMAC(AS, M, 0, S); RA; PH4_SB();                  \
  MMAC(AS, M, 1, S); RB; PH4_SB();                  \
  MMAC(AS, M, 2, S); F0; PH4_SB();                  \
  MMAC(AS, M, 3, S); F1; PH4_SB()

  // ---- the epilogue in 8-byte pieces: piece (m, r) = pixel (row 4 wave + m, column 4 lg + r), channels 4 li .. + 3.
  // Stores and operand loads go through buffer resources of the tile's IMAGE (base + size in scalar registers): a lane outside
  // the output gets an offset past the resource and the hardware drops its store / returns 0 for its load - no exec masking, no
  // branch around the instruction, and every piece issues exactly one store whatever the tile's shape, which is what lets
  // the end-of-tile wait COUNT them.
  constexpr unsigned OOB = 0x7ffffff0u;
  constexpr int RSRC_FLAGS = 0x00020000;      // raw buffer, 32-bit offsets (gfx90a / gfx94x / gfx950 data format word)
  struct EpiCtx { __amdgpu_buffer_rsrc_t out, q0, q1; unsigned o00, rowstep, colstep; int rlim, clim; };
  const int img_bytes = p.OH * p.OW * p.Cout * 2;
  auto in_rsrc = [&](const void* base, const TileCtx& tc) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(base)) + (size_t)tc.b * p.OH * p.OW * p.Cout, 0,
                                             base ? img_bytes : 0, RSRC_FLAGS);
  };
  auto epi_ctx = [&](const TileCtx& tc) -> EpiCtx {
    EpiCtx e;
    const size_t img = (size_t)tc.b * p.OH * p.OW * p.Cout;
    e.out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(p.out) + img, 0, img_bytes, RSRC_FLAGS);
    if constexpr (OPS) {      // operand 0 = residual gradient / the BatchNorm's y; operand 1 = the residual's mask tensor
      e.q0 = in_rsrc(ORM ? p.res_g : p.bst_y, tc);
      e.q1 = in_rsrc(ORM ? p.res_a : p.bst_y, tc);
    } else { e.q0 = e.out; e.q1 = e.out; }
    // BYTE offset of this lane's piece (0, 0) inside the image; pixel rows advance by rowstep, columns by colstep
    e.o00 = 2u * ((unsigned)(((tc.r0 + wave * NM) * p.os + p.oa_h) * p.OW + (tc.c0 + 4 * lg) * p.os + p.oa_w) * (unsigned)p.Cout + 4u * (unsigned)li);
    e.rowstep = 2u * (unsigned)(p.os * p.OW * p.Cout);
    e.colstep = 2u * (unsigned)(p.os * p.Cout);
    e.rlim = p.OHt - (tc.r0 + wave * NM);      // piece row m lies inside the output iff m < rlim,
    e.clim = p.OWt - tc.c0 - 4 * lg;           // its column r iff r < clim
    return e;
  };
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  // (OVL) the 8 micro-steps of one piece, spread over the filler slots of one k-step; state between the steps:
  float ev[4], ey[4];
  u32x2 ew;
  bool emine = true;
  unsigned eoff = 0;
  // (OPS) operand registers: head slots 0 .. PD - 1 = pieces 0 .. PD - 1, then the rotating window
  u32x2 o_r0[OPS ? NSL : 1], o_r1[(OPS && ORM) ? NSL : 1];
  auto slot_of = [](const int pc) -> int { return pc < PD ? pc : PD + (pc - PD) % NWIN; };
  // (the first stream of a workgroup consumes head slots nobody loaded: its dummy previous tile contributes 0 x (y - mean) to
  // the fused sums, which must not be 0 x Inf)
#pragma unroll
  for (int i = 0; i < (OPS ? NSL : 1); ++i) { o_r0[i] = u32x2{0u, 0u}; o_r1[(OPS && ORM) ? i : 0] = u32x2{0u, 0u}; }
  f32x4 ocms = {0.f, 0.f, 0.f, 0.f}, ocmh = ocms, ocmu = ocms;      // (OPS, BST = 1) mask scale / shift, mean of this lane's 4 channels
  if constexpr (OPS && BST == 1) {
    ocms = *reinterpret_cast<const f32x4*>(ss + 4 * li);
    ocmh = *reinterpret_cast<const f32x4*>(ss + BNT + 4 * li);
    ocmu = *reinterpret_cast<const f32x4*>(ss + 2 * BNT + 4 * li);
  }
  auto ovl_load = [&](const EpiCtx& ec, const int pc) {
    if constexpr (OPS) {
      const int m = pc >> 2, r = pc & 3, sl = slot_of(pc);
      const bool mine = m < ec.rlim && r < ec.clim;
      const int off = (int)(mine ? ec.o00 + (unsigned)m * ec.rowstep + (unsigned)r * ec.colstep : OOB);
      o_r0[sl] = __builtin_amdgcn_raw_buffer_load_b64(ec.q0, off, 0, 0);
      if constexpr (ORM != 0) o_r1[sl] = __builtin_amdgcn_raw_buffer_load_b64(ec.q1, off, 0, 0);
    }
  };
  // step K (0..7) of piece (M, R) of the tile in accumulator set AS with context EC
#define PH4_EPI_STEP(K, AS, M, R, EC)                                                                                        \
  do {                                                                                                                       \
    if ((K) == 0) { ev[0] = acc[AS][M][0][R]; ev[1] = acc[AS][M][1][R]; }                                                    \
    else if ((K) == 1) { ev[2] = acc[AS][M][2][R]; ev[3] = acc[AS][M][3][R]; }                                               \
    else if ((K) == 2) {                                                                                                     \
      emine = (M) < (EC).rlim && (R) < (EC).clim;                                                                            \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) ev[j_] = emine ? ev[j_] : 0.f;                                        \
    }                                                                                                                        \
    else if ((K) == 3) {                                                                                                     \
      if constexpr (OPS) {      /* the stored bf16 gradient: the residual is added to it, the fused sums are taken over it */  \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) { bf16x2 b_; b_[0] = (bf16)ev[2 * j_]; b_[1] = (bf16)ev[2 * j_ + 1]; ew[j_] = __builtin_bit_cast(unsigned, b_); } \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) { ev[2 * j_] = __builtin_bit_cast(float, ew[j_] << 16); ev[2 * j_ + 1] = __builtin_bit_cast(float, ew[j_] & 0xffff0000u); } \
      } else { s1[0] += ev[0]; s1[1] += ev[1]; s1[2] += ev[2]; s1[3] += ev[3]; }                                             \
    }                                                                                                                        \
    else if ((K) == 4) {                                                                                                     \
      if constexpr (OPS && ORM != 0) {                                                                                       \
        const u32x2 g_ = o_r0[slot_of(4 * (M) + (R))], a_ = o_r1[(OPS && ORM) ? slot_of(4 * (M) + (R)) : 0];                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                                   \
          const float g0_ = __builtin_bit_cast(float, g_[j_] << 16), g1_ = __builtin_bit_cast(float, g_[j_] & 0xffff0000u);   \
          const float a0_ = __builtin_bit_cast(float, a_[j_] << 16), a1_ = __builtin_bit_cast(float, a_[j_] & 0xffff0000u);   \
          ev[2 * j_] = ev[2 * j_] + (a0_ > 0.f ? g0_ : 0.f);                                                                 \
          ev[2 * j_ + 1] = ev[2 * j_ + 1] + (a1_ > 0.f ? g1_ : 0.f);                                                         \
        }                                                                                                                    \
      } else if constexpr (OPS) {      /* BST = 1: dz = gradient under the BatchNorm's own ReLU */                           \
        const u32x2 y_ = o_r0[slot_of(4 * (M) + (R))];                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                   \
          ey[j_] = __builtin_bit_cast(float, (j_ & 1) ? (y_[j_ >> 1] & 0xffff0000u) : (y_[j_ >> 1] << 16));                  \
          ev[j_] = __builtin_fmaf(ey[j_], ocms[j_], ocmh[j_]) > 0.f ? ev[j_] : 0.f;                                          \
        }                                                                                                                    \
      } else { _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) s2[j_] = __builtin_fmaf(ev[j_], ev[j_], s2[j_]); }           \
    }                                                                                                                        \
    else if ((K) == 5) {                                                                                                     \
      if constexpr (OPS && ORM == 0) {                                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) { s1[j_] += ev[j_]; s2[j_] = __builtin_fmaf(ev[j_], ey[j_] - ocmu[j_], s2[j_]); } \
      } else {                                                                                                               \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) { bf16x2 b_; b_[0] = (bf16)ev[2 * j_]; b_[1] = (bf16)ev[2 * j_ + 1]; ew[j_] = __builtin_bit_cast(unsigned, b_); } \
      }                                                                                                                      \
    }                                                                                                                        \
    else if ((K) == 6) { eoff = emine ? (EC).o00 + (unsigned)(M) * (EC).rowstep + (unsigned)(R) * (EC).colstep : OOB; }      \
    else __builtin_amdgcn_raw_buffer_store_b64(ew, (EC).out, (int)eoff, 0, 0);                                               \
  } while (0)

  // The epilogue's operands (residual + mask, the fused sums' y / mask / second y; launches with OVL = false) are requested
  // while the tile's last taps still run - taps 6 and 7 carry eight pieces each: a load issued where it is consumed exposes a
  // full memory round trip per pixel row (17.6 k cycles per tile against 5.5 k of taps).  (One piece per k-step from tap 0 on
  // was slower still: a one-wave-per-SIMD kernel has nowhere near the memory parallelism of a streaming pass - 120 to 400
  // cycles of issue stall per load, growing with the loads in flight; the four-operand form is therefore off by default.)
  // 8-byte loads, coalesced like the stores,
  // unconditional (out-of-image lanes read 0 through the buffer resource: with "if (inside) load" on top of a default value the
  // compiler's hazard tracking put an s_waitcnt vmcnt(0) in front of every piece - each load waited for the one before it).
  constexpr bool EPI_LOADS = !OVL || OPS;      // (OPS: the last tile of a workgroup only)
  u32x2 e_rg[EPI_LOADS ? 16 : 1], e_ra[EPI_LOADS ? 16 : 1], e_ry[(EPI_LOADS && BST) ? 16 : 1], e_rb[(EPI_LOADS && BST >= 2) ? 16 : 1],
      e_ry2[(EPI_LOADS && BST == 3) ? 16 : 1];
  const int rmode = p.res_g ? (p.res_a ? 2 : 1) : 0;
  auto epi_load_piece = [&](const TileCtx& tc, const EpiCtx& ec, const int pc) {      // piece pc = (pixel row pc >> 2, column pc & 3)
    if constexpr (EPI_LOADS) {
      const int m = pc >> 2, r = pc & 3;
      const bool mine = m < ec.rlim && r < ec.clim;
      const int off = (int)(mine ? ec.o00 + (unsigned)m * ec.rowstep + (unsigned)r * ec.colstep : OOB);
      if (rmode >= 1) e_rg[pc] = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc(p.res_g, tc), off, 0, 0);
      if (rmode == 2) e_ra[pc] = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc(p.res_a, tc), off, 0, 0);
      if constexpr (BST != 0) e_ry[pc] = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc(p.bst_y, tc), off, 0, 0);
      if constexpr (BST >= 2) e_rb[pc] = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc(p.bst_a, tc), off, 0, 0);
      if constexpr (BST == 3) e_ry2[pc] = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc(p.bst_y2, tc), off, 0, 0);
    }
  };

  // ---- the sequential epilogue (OVL = false, and the last tile of an OVL workgroup): all sixteen pieces of a tile
  auto epilogue_seq = [&](const TileCtx& tc, const int AS, auto rmc) {
    constexpr int RM = decltype(rmc)::value;
    const EpiCtx ec = epi_ctx(tc);
    constexpr bool has_y2 = BST == 3;
    f32x4 cms = {0.f, 0.f, 0.f, 0.f}, cmh = cms, cmu = cms, cmu2 = cms;
    if constexpr (BST != 0) {
      if constexpr (BST == 1) { cms = *reinterpret_cast<const f32x4*>(ss + 4 * li); cmh = *reinterpret_cast<const f32x4*>(ss + BNT + 4 * li); }
      cmu = *reinterpret_cast<const f32x4*>(ss + 2 * BNT + 4 * li);
      if constexpr (has_y2) cmu2 = *reinterpret_cast<const f32x4*>(ss + 3 * BNT + 4 * li);
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pc = 4 * m + q;
        const bool mine = m < ec.rlim && q < ec.clim;
        const u32x2 rg = e_rg[EPI_LOADS ? pc : 0], ra = e_ra[EPI_LOADS ? pc : 0], ry = e_ry[(EPI_LOADS && BST) ? pc : 0],
                    rb = e_rb[(EPI_LOADS && BST >= 2) ? pc : 0], ry2 = e_ry2[(EPI_LOADS && BST == 3) ? pc : 0];
        float v[4];
#pragma unroll
        for (int n = 0; n < NN; ++n) {
          v[n] = mine ? acc[AS][m][n][q] : 0.f;
          if constexpr (BST == 0) {
            s1[n] += v[n];
            s2[n] = __builtin_fmaf(v[n], v[n], s2[n]);      // (explicit: every instantiation must round the sums alike)
          }
        }
        u32x2 w;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x2 b;
          b[0] = (bf16)v[2 * h];
          b[1] = (bf16)v[2 * h + 1];
          if constexpr (RM > 0) {
            const float g0 = __builtin_bit_cast(float, rg[h] << 16), g1 = __builtin_bit_cast(float, rg[h] & 0xffff0000u);
            const float a0 = __builtin_bit_cast(float, ra[h] << 16), a1 = __builtin_bit_cast(float, ra[h] & 0xffff0000u);
            b[0] = (bf16)((float)b[0] + ((RM < 2 || a0 > 0.f) ? g0 : 0.f));
            b[1] = (bf16)((float)b[1] + ((RM < 2 || a1 > 0.f) ? g1 : 0.f));
          }
          w[h] = __builtin_bit_cast(unsigned, b);
          if constexpr (BST != 0) {
            // the sums are taken over the STORED gradient (bf16), exactly what the separate reduction pass reads back
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int n = 2 * h + e;
              const float y = __builtin_bit_cast(float, e ? (ry[h] & 0xffff0000u) : (ry[h] << 16));
              bool on;
              if constexpr (BST == 1) on = __builtin_fmaf(y, cms[n], cmh[n]) > 0.f;
              else on = __builtin_bit_cast(float, e ? (rb[h] & 0xffff0000u) : (rb[h] << 16)) > 0.f;
              float dz = (float)b[e];
              dz = (on && mine) ? dz : 0.f;
              s1[n] += dz;
              s2[n] = __builtin_fmaf(dz, y - cmu[n], s2[n]);
              if constexpr (has_y2) {
                const float y2 = __builtin_bit_cast(float, e ? (ry2[h] & 0xffff0000u) : (ry2[h] << 16));
                s3[has_y2 ? n : 0] = __builtin_fmaf(dz, y2 - cmu2[n], s3[has_y2 ? n : 0]);
              }
            }
          }
        }
        __builtin_amdgcn_raw_buffer_store_b64(w, ec.out, (int)(mine ? ec.o00 + (unsigned)m * ec.rowstep + (unsigned)q * ec.colstep : OOB), 0, 0);
      }
    }
  };
  auto epilogue_any = [&](const TileCtx& tc, const int AS) {
    if (rmode == 0) epilogue_seq(tc, AS, std::integral_constant<int, 0>{});
    else if (rmode == 1) epilogue_seq(tc, AS, std::integral_constant<int, 1>{});
    else epilogue_seq(tc, AS, std::integral_constant<int, 2>{});
  };

  // ---- the tile stream.  ABUF (the halo buffer of this tile = its accumulator set under OVL) is a literal at both call sites,
  // so that after inlining every fragment address is a base register + an immediate.  Fragment set 0 / 1 = k-step 0 / 1 of a tap;
  // each pixel-row group's four MFMAs carry the read of the next k-step's pixel and weight fragment of the same index.  Taps
  // 0..5 also issue, two pieces per wave, the LDS-DMA of the NEXT tile's halo into the other buffer (nobody reads it before the
  // barrier at the end of this tile); under OVL every k-step of taps 0..7 carries one 8-byte piece of the PREVIOUS tile's
  // epilogue, otherwise taps 6 and 7 carry the loads of this tile's epilogue operands.
  // (OVL) the first tile's "previous tile" is a dummy whose pieces all lie outside the output: its stores are issued and
  // dropped, its accumulator garbage is masked out of the sums - no first-tile special case inside the stream
  EpiCtx eprev = epi_ctx(tcur);
  eprev.rlim = 0;
  EpiCtx ecur = eprev;
  auto tile_body = [&](const int ABUF) __attribute__((always_inline)) {
    const int AOFF = ABUF * C::A_BYTES;
    const int AS = OVL ? ABUF : 0, PS = OVL ? (ABUF ^ 1) : 0;
    const bool need_loads = !OVL && (BST != 0 || rmode != 0);
    const unsigned char* hb = halo_base(tnext);
    const unsigned hdst = lds0 + (ABUF ^ 1) * C::A_BYTES + wave * 1024;
#define PH4_DMA_H(E)                                                                                           \
  do {                                                                                                         \
    if (nvalid && (E) < C::NHE && wave + 4 * (E) < C::NHD)                                                     \
      lds_dma16_4(piece_ok((E) < C::NHE ? (E) : 0, tnext.iy_base, tnext.ix_base) ? hb + h_off[(E) < C::NHE ? (E) : 0] : zero_src, hdst + (E) * 4096); \
  } while (0)
    // slot J (0..7) of k-step KS of tap t: micro-step J of the previous tile's piece 2 t + KS = (m, r) = ((2 t + KS) >> 2, .. & 3),
    // or (launches that read in their epilogue) one piece's loads
#define PH4_EPI(KS, J)                                                                                                 \
  do {                                                                                                                 \
    if constexpr (OVL) {                                                                                               \
      if (t < 8) PH4_EPI_STEP(J, PS, (2 * t + (KS)) >> 2, (2 * t + (KS)) & 3, eprev);                                    \
      if constexpr (OPS) {                                                                                             \
        if (t < 8 && (J) == 1 && 2 * t + (KS) + PD < 16) ovl_load(eprev, (2 * t + (KS) + PD) & 15);                      \
        if ((t == 6 || t == 7) && ((J) == 3 || (J) == 6) && (2 * t + (KS) - 12) * 2 + ((J) == 6) < PD)                   \
          ovl_load(ecur, ((2 * t + (KS) - 12) * 2 + ((J) == 6)) & 15);                                                   \
      }                                                                                                                \
    }                                                                                                                  \
    else if (need_loads && (t == 6 || t == 7) && ((J) & 1) == 0) epi_load_piece(tcur, ecur, (t - 6) * 8 + (KS) * 4 + ((J) >> 1)); \
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
        PH4_GROUP(PH4_MM0, AS, 0, 0, fa[1][0] = PH4_LD(abase1[dx], aoff + 0 * C::ROW_BYTES), fb[1][0] = PH4_LD(bx1[0] + bext, boff), PH4_EPI(0, 0), PH4_EPI(0, 1));
        PH4_GROUP(PH4_MM0, AS, 1, 0, fa[1][1] = PH4_LD(abase1[dx], aoff + 1 * C::ROW_BYTES), fb[1][1] = PH4_LD(bx1[1] + bext, boff), PH4_EPI(0, 2), PH4_EPI(0, 3); PH4_DMA_H(0));
        PH4_GROUP(PH4_MM0, AS, 2, 0, fa[1][2] = PH4_LD(abase1[dx], aoff + 2 * C::ROW_BYTES), fb[1][2] = PH4_LD(bx1[2] + bext, boff), PH4_EPI(0, 4), PH4_EPI(0, 5));
        PH4_GROUP(PH4_MM0, AS, 3, 0, fa[1][3] = PH4_LD(abase1[dx], aoff + 3 * C::ROW_BYTES), fb[1][3] = PH4_LD(bx1[3] + bext, boff), PH4_EPI(0, 6), PH4_EPI(0, 7));
      } else {
        PH4_GROUP(PH4_MM, AS, 0, 0, fa[1][0] = PH4_LD(abase1[dx], aoff + 0 * C::ROW_BYTES), fb[1][0] = PH4_LD(bx1[0] + bext, boff), PH4_EPI(0, 0), PH4_EPI(0, 1));
        PH4_GROUP(PH4_MM, AS, 1, 0, fa[1][1] = PH4_LD(abase1[dx], aoff + 1 * C::ROW_BYTES), fb[1][1] = PH4_LD(bx1[1] + bext, boff), PH4_EPI(0, 2),
                  PH4_EPI(0, 3); if (t < C::HALO_TAPS) PH4_DMA_H(2 * t));
        PH4_GROUP(PH4_MM, AS, 2, 0, fa[1][2] = PH4_LD(abase1[dx], aoff + 2 * C::ROW_BYTES), fb[1][2] = PH4_LD(bx1[2] + bext, boff), PH4_EPI(0, 4), PH4_EPI(0, 5));
        PH4_GROUP(PH4_MM, AS, 3, 0, fa[1][3] = PH4_LD(abase1[dx], aoff + 3 * C::ROW_BYTES), fb[1][3] = PH4_LD(bx1[3] + bext, boff), PH4_EPI(0, 6), PH4_EPI(0, 7));
      }
      // ---- k-step 1 on set 1; reads of the next tap's k-step 0 into set 0 (the last tap has none: the next tile's halo is
      // published by the barrier that follows)
      if (t + 1 < NTAPS) {
        PH4_GROUP(PH4_MM, AS, 0, 1, fa[0][0] = PH4_LD(abase0[dxn], aoffn + 0 * C::ROW_BYTES), fb[0][0] = PH4_LD(bx0[0] + bextn, boffn), PH4_EPI(1, 0), PH4_EPI(1, 1));
        PH4_GROUP(PH4_MM, AS, 1, 1, fa[0][1] = PH4_LD(abase0[dxn], aoffn + 1 * C::ROW_BYTES), fb[0][1] = PH4_LD(bx0[1] + bextn, boffn), PH4_EPI(1, 2),
                  PH4_EPI(1, 3); if (t < C::HALO_TAPS) PH4_DMA_H(2 * t + 1));
        PH4_GROUP(PH4_MM, AS, 2, 1, fa[0][2] = PH4_LD(abase0[dxn], aoffn + 2 * C::ROW_BYTES), fb[0][2] = PH4_LD(bx0[2] + bextn, boffn), PH4_EPI(1, 4), PH4_EPI(1, 5));
        PH4_GROUP(PH4_MM, AS, 3, 1, fa[0][3] = PH4_LD(abase0[dxn], aoffn + 3 * C::ROW_BYTES), fb[0][3] = PH4_LD(bx0[3] + bextn, boffn), PH4_EPI(1, 6), PH4_EPI(1, 7));
      } else {
        PH4_GROUP(PH4_MM, AS, 0, 1, PH4_NOP, PH4_NOP, PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM, AS, 1, 1, PH4_NOP, PH4_NOP, PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM, AS, 2, 1, PH4_NOP, PH4_NOP, PH4_NOP, PH4_NOP);
        PH4_GROUP(PH4_MM, AS, 3, 1, PH4_NOP, PH4_NOP, PH4_NOP, PH4_NOP);
      }
    }
#undef PH4_DMA_H
#undef PH4_EPI
  };

  int last_as = 0;
  unsigned long long cy_main = 0, cy_epi = 0, cy_wait = 0, ntile = 0;
  const unsigned long long cy_t0 = PH4_CLK();
#ifdef PH_TAP_TRACE
  const unsigned long long wall_t0 = wall_clock64();
#endif
  for (int k = 0;; k += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // End of tile: the next halo (this wave's pieces, issued in taps 0..5) must have LANDED before the barrier publishes it.
      // vmcnt(0) also covers the stores issued meanwhile - under OVL the previous tile's pieces (taps 0..7: the youngest is a
      // tap old and acknowledged; a counted wait that leaves them in flight measured ~80 cycles per tile less and is not worth
      // an ordering assumption), otherwise none: the sequential epilogue runs AFTER the barrier, so that its stores drain beside
      // the next tile's first taps instead of being waited for.
      const unsigned long long q0_ = PH4_CLK();
      ecur = epi_ctx(tcur);
      if (half == 0) tile_body(0); else tile_body(1);
      const unsigned long long q1_ = PH4_CLK();
      if constexpr (OVL) { eprev = ecur; last_as = half; }
      const unsigned long long q2_ = PH4_CLK();
      PH4_WAIT_VMCNT(0);
      if (FUSE_IN && nvalid) {
        xform_halo(half ^ 1);
        // the transform's ds_writes must have completed before the barrier lets the other waves read them (an inline-asm
        // s_barrier is invisible to the compiler's wait insertion; the first fragment reads follow the barrier at once)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      PH4_BARRIER();
      const unsigned long long q3_ = PH4_CLK();
      if constexpr (!OVL) epilogue_any(tcur, 0);
      cy_main += q1_ - q0_; cy_wait += q3_ - q2_; cy_epi += (q2_ - q1_) + (PH4_CLK() - q3_); ++ntile;
      if (!nvalid) goto done;
      tcur = tnext;
      tn = tile_id(k + half + 2);
      nvalid = tn >= 0;
      if (nvalid) tnext = decode(tn);
    }
  }
done:
  if constexpr (OVL) {      // the last tile's epilogue has no tap stream to ride in
    if constexpr (OPS) {
      const EpiCtx ec = epi_ctx(tcur);
#pragma unroll
      for (int pc = 0; pc < 16; ++pc) epi_load_piece(tcur, ec, pc);
    }
    // (distinct markers at both ends of both arms: merged into one body by the optimizer, the two accumulator sets become one
    // phi of 64 values, which the register coalescer resolves by giving both sets the SAME registers and copying at every tile)
    if (last_as == 0) {
      asm volatile("; tail of accumulator set 0" ::: "memory");
      epilogue_seq(tcur, 0, std::integral_constant<int, ORM>{});
      asm volatile("; end of tail 0" ::: "memory");
    } else {
      asm volatile("; tail of accumulator set 1" ::: "memory");
      epilogue_seq(tcur, OVL ? 1 : 0, std::integral_constant<int, ORM>{});
      asm volatile("; end of tail 1" ::: "memory");
    }
  }
  PH4_TR(0, cy_main); PH4_TR(1, cy_epi); PH4_TR(2, cy_wait); PH4_TR(3, ntile); PH4_TR(4, PH4_CLK() - cy_t0);
#ifdef PH_TAP_TRACE
  PH4_TR(5, wall_clock64() - wall_t0);
#endif
  (void)cy_main; (void)cy_epi; (void)cy_wait; (void)ntile; (void)cy_t0;
  if (p.stats) {   // one partial row per workgroup: [2][64] (forward) or [3][64] (fused BatchNorm-backward sums)
    constexpr int NS = BST ? 3 : 2;
    float* red = reinterpret_cast<float*>(smem + C::RED_OFF);   // [4 waves][NS][BNT]
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      float a1 = s1[n], a2 = s2[n], a3 = s3[BST == 3 ? n : 0];
      a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64); a3 += __shfl_xor(a3, 16, 64);
      a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64); a3 += __shfl_xor(a3, 32, 64);
      if (lg == 0) {
        red[(wave * NS + 0) * BNT + 4 * li + n] = a1;
        red[(wave * NS + 1) * BNT + 4 * li + n] = a2;
        if (NS == 3) red[(wave * NS + 2) * BNT + 4 * li + n] = BST == 3 ? a3 : 0.f;
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

template <bool FUSE_IN, int BST, bool OVL, int ORM = 0>
int launch4(const PhTapConv& p, hipStream_t st) {
  using C = Tap4Cfg;
  auto kern = tapconv4_kernel<FUSE_IN, BST, OVL, ORM>;
  static std::once_flag attr_once;      // (one flag per template instantiation; thread-safe)
  static hipError_t attr_rc = hipSuccess;
  std::call_once(attr_once, [&] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  });
  if (attr_rc != hipSuccess) return PH_ELAUNCH;
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

// A/B and test switch: PH_TAP4_OVL=0 / ph_debug_set_tap4_ovl(0) = every launch with the epilogue after the tile
static int ph_tap4_ovl_switch(int set) {
  static int on = [] { const char* e = getenv("PH_TAP4_OVL"); return (e && e[0] == '0') ? 0 : 1; }();
  if (set >= 0) on = set ? 1 : 0;
  return on;
}
extern "C" int ph_debug_set_tap4_ovl(int on) { return ph_tap4_ovl_switch(on ? 1 : 0); }
// A/B switch: PH_TAP4_OVL_OPS=0 = launches whose epilogue reads operands keep it after the tile (the round-5 first form)
static bool ph_tap4_ovl_ops() {
  static const bool on = [] { const char* e = getenv("PH_TAP4_OVL_OPS"); return !(e && e[0] == '0'); }();
  return on && ph_tap4_ovl_switch(-1);
}

// eligible: 3x3 stride-1 perf-mode configuration with Cin = Cout = 64 and the hard-coded 3x3 tap geometry (every ResNet-18 shape
// that reached tapconv2_l1_kernel)
bool ph_tapconv4_eligible(const PhTapConv* p) {
  if (p->ntaps != 9 || p->Cin != 64 || p->Cout != 64 || p->m_groups) return false;
  // the kernel is hard-coded for the stride-1 / pad-1 geometry over the whole output: no output stride or offset
  if (p->os != 1 || p->oa_h || p->oa_w || p->OHt != p->OH || p->OWt != p->OW || p->iy0 != -1 || p->ix0 != -1) return false;
  if (p->IH != p->OH || p->IW != p->OW) return false;
  for (int k = 0; k < 9; ++k)
    if (p->dy[k] != k / 3 || p->dx[k] != k % 3 || p->wtap[k] < 0 || p->wtap[k] > 8) return false;
  if (p->bst_y && (p->in_scale || !p->bst_mean || !p->stats || (p->bst_y2 && !p->bst_mean2))) return false;
  if (p->bst_y && !p->bst_a && (!p->bst_scale || !p->bst_shift)) return false;
  return true;
}

int ph_tapconv4_launch(const PhTapConv* p, hipStream_t st) {
  if (!ph_tapconv4_eligible(p)) return PH_EINVAL;
  if (p->bst_y) {
    if (!p->bst_a) {
      if (p->bst_y2) return PH_EINVAL;
      return (ph_tap4_ovl_ops() && !p->res_g) ? launch4<false, 1, true>(*p, st) : launch4<false, 1, false>(*p, st);
    }
    return p->bst_y2 ? launch4<false, 3, false>(*p, st) : launch4<false, 2, false>(*p, st);
  }
  // launches whose epilogue reads nothing (forward; dgrad without a residual) store the previous tile inside the next tile's taps
  if (p->res_g && p->res_a && !p->in_scale && ph_tap4_ovl_ops()) return launch4<false, 0, true, 2>(*p, st);
  if (p->res_g || !ph_tap4_ovl_switch(-1)) return p->in_scale ? launch4<true, 0, false>(*p, st) : launch4<false, 0, false>(*p, st);
  return p->in_scale ? launch4<true, 0, true>(*p, st) : launch4<false, 0, true>(*p, st);
}

#ifdef PH_TAP_TRACE
extern "C" int ph_debug_tap4_trace(unsigned long long* host_out, int nwg) {
  if (nwg > 1024) nwg = 1024;
  if (hipDeviceSynchronize() != hipSuccess) return PH_ELAUNCH;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ph_tap4_trace), (size_t)nwg * 8 * sizeof(unsigned long long)) == hipSuccess
             ? PH_OK : PH_ELAUNCH;
}
#endif
