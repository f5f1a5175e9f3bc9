// Pieces shared by the tap-convolution kernels (conv_tap.hip, conv_tap2.hip): the swizzled LDS image addressing,
// the MFMA-row -> pixel permutation, and the debug-build phase tracer.
#pragma once
#include "ph_common.h"

// Debug build only (make trace): per-workgroup phase timestamps (100 MHz wall clock) of a tap-conv kernel; the
// including file defines the buffer `__device__ unsigned long long ph_tap_trace[PH_TRACE_WGS * 12]`,
// read back by tests/trace_tapconv_gpu.py.  Not part of the product library or of the public C-ABI.
#ifdef PH_TAP_TRACE
#define PH_TRACE_WGS 65536
#define PH_TRACE(k)                                                                                        \
  do {                                                                                                     \
    const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                   \
    if (threadIdx.x == 0 && wg_ < PH_TRACE_WGS) ph_tap_trace[(size_t)wg_ * 12 + (k)] = wall_clock64();      \
  } while (0)
#define PH_TRACE_HWID()                                                                                    \
  do {                                                                                                     \
    const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                   \
    if (threadIdx.x == 0 && wg_ < PH_TRACE_WGS)                                                            \
      ph_tap_trace[(size_t)wg_ * 12 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) |   \
                                          (unsigned)__builtin_amdgcn_s_getreg(63492);                      \
  } while (0)
#define PH_TRACE_ACC(k, v)                                                                                 \
  do {                                                                                                     \
    const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                   \
    if (threadIdx.x == 0 && wg_ < PH_TRACE_WGS) ph_tap_trace[(size_t)wg_ * 12 + (k)] = (v);                \
  } while (0)
#define PH_TRACE_ACC_T(k, v, thr)                                                                          \
  do {                                                                                                     \
    const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                   \
    if (threadIdx.x == (thr) && wg_ < PH_TRACE_WGS) ph_tap_trace[(size_t)wg_ * 12 + (k)] = (v);            \
  } while (0)
#define PH_CLK() clock64()
#else
#define PH_TRACE_ACC_T(k, v, thr)
#define PH_TRACE(k)
#define PH_TRACE_HWID()
#define PH_TRACE_ACC(k, v)
#define PH_CLK() 0ull
#endif


// LDS image addressing shared by the A (halo pixels) and B (weight rows) tiles: 128-B rows (64 bf16), two rows
// per 256-B bank row; the 16-B chunk index is XOR-swizzled with 3 bits of the row-pair index, the row's half of the
// bank row stays where it is.  A ds_read_b128 lane group (16 lanes, MI355X_MICROARCH.md: LDS) must touch 16 distinct
// 16-B slots of the 256-B bank row.  With the lane->pixel permutation below a group reads 16 CONSECUTIVE pixels:
// eight row pairs with eight different swizzle terms when the run starts on an even pixel, and when it starts on an
// odd one (the dx = 1 taps) its first and last pixel share a swizzle term but sit in opposite halves.  (Round 1 XORed
// FOUR bits of the row pair into the slot: equally conflict-free on even runs, but on odd runs the last pixel then
// lands on the first pixel's slot - 2-way on a third of the taps: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.18 -> 0.00
// for the dense kernel, 0.11 -> 0.00 for the layer-1 kernel; the kernel times did not move, the LDS array was not the limit.)  B rows are consecutive channels per lane, i.e. the
// hardware groups see rows {0-3, 12-15, 20-27} + const: row pairs 0,1,6,7,10,11,12,13 -> terms 0,1,6,7,2,3,4,5.
#define PH_SWZ_MASK 7   // (bits of the row-pair index in the swizzle: the LDS-DMA source mapping in conv_tap2.hip uses it too)
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return (row >> 1) * 256 + ((((row & 1) << 3) | (chunk ^ ((row >> 1) & PH_SWZ_MASK))) << 4);
}
// MFMA A-fragment row i (0..31) -> pixel (fr, c) inside a 2 x 16 patch such that the hardware's
// ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} each read 16 CONSECUTIVE pixels of one row
__device__ __forceinline__ void frag_row_to_pixel(int i, int& fr, int& c) {
  const int k = i >> 2;
  fr = __popc(k) & 1;
  c = ((k >> 1) << 2) | (i & 3);
}

