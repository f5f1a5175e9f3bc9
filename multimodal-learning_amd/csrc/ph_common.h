// Common device/host helpers for the pathomic-distill HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define PH_OK 0
#define PH_EINVAL (-22)
#define PH_ELAUNCH (-5)

// precision modes of the C-ABI (activation storage type follows the mode)
#define PH_PREC_BF16 0    // perf mode: bf16 operands+activations, fp32 accumulate / statistics
#define PH_PREC_BF16X6 1  // parity mode: fp32 activations, operands split into 3 bf16 planes, 6 MFMA products
#define PH_PREC_BF16X3 2  // fp32 activations, the 3 leading products only (hi*hi, hi*mid, mid*hi: 16-bit operands, ~2^-16 per
                          // product): half the matrix work of the parity mode; same kernels, same packed weights

#define PH_LAUNCH_CHECK()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return PH_ELAUNCH;              \
  } while (0)

template <typename T> struct is_f32 { static constexpr bool value = std::is_same<T, float>::value; };

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return (float)*p; }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16* p, float v) { *p = (bf16)v; }

// 8 consecutive elements <-> 8 floats (16 B for bf16, 32 B for f32); pointers must be 16-B aligned
__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
  bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[i + 4] = b[i]; }
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = r;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[i + 4]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}

// 3-way bf16 split of an fp32 value (parity mode "bf16x6"): x = p0 + p1 + p2 exactly (3 x 8 significant
// bits = the fp32 mantissa).  Products p_i*q_j with i+j <= 2 (6 MFMAs) reproduce the fp32 product to
// ~2^-24 relative; accumulation is fp32 as in the reference.
__device__ __forceinline__ void split3_bf16(float x, bf16& p0, bf16& p1, bf16& p2) {
  p0 = (bf16)x;
  const float r1 = x - (float)p0;
  p1 = (bf16)r1;
  p2 = (bf16)(r1 - (float)p1);
}
constexpr int PH_NPLANES = 3;
// the (i, j) plane pairs, least significant first
#define PH_SPLIT_PAIRS_LO(X) X(2, 0) X(0, 2) X(1, 1)
#define PH_SPLIT_PAIRS_HI(X) X(1, 0) X(0, 1) X(0, 0)
#define PH_SPLIT_PAIRS(X) PH_SPLIT_PAIRS_LO(X) PH_SPLIT_PAIRS_HI(X)
#define PH_IS_SPLIT_PREC(p) ((p) == PH_PREC_BF16X6 || (p) == PH_PREC_BF16X3)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
