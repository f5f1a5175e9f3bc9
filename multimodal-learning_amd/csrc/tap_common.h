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
// per 256-B bank row, 16-B chunk slot XOR-swizzled with 4 bits of the row-pair index.  With the lane->pixel
// permutation below every 16-lane ds_read_b128 group touches 16 distinct slots (no bank conflicts); the
// previous 3-bit swizzle measured 38-54 % conflict cycles (profiles/r01_pmc_before.txt).
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return (row >> 1) * 256 + (((((row & 1) << 3) | chunk) ^ ((row >> 1) & 15)) << 4);
}
// MFMA A-fragment row i (0..31) -> pixel (fr, c) inside a 2 x 16 patch such that the hardware's
// ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} each read 16 CONSECUTIVE pixels of one row
__device__ __forceinline__ void frag_row_to_pixel(int i, int& fr, int& c) {
  const int k = i >> 2;
  fr = __popc(k) & 1;
  c = ((k >> 1) << 2) | (i & 3);
}

