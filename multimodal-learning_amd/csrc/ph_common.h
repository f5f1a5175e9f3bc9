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

#define PH_PREC_FP16X3 3  // "half-pair" mode: every tensor a convolution READS (activations, BatchNorm-backward dz) is stored as two fp16
                          // planes x = hi + lo * 2^-11 (4 B per element, layout below); conv outputs / gradients stay fp32; three
                          // fp16 MFMA products per k-step (hi*hi, hi*lo, lo*hi: 22-bit operands, ~2^-22 per product)
#define PH_PREC_FP16X1 4  // BACKWARD arithmetic of a PH_PREC_FP16X3 plan only (ph_resnet_plan_set_backward_prec): dgrad / wgrad multiply the hi
                          // planes alone (one fp16 product, 11-bit operands, fp32 accumulation) - the tensors are the half-pair ones
#define PH_IS_F32_OUT_PREC(p) ((p) == PH_PREC_BF16X6 || (p) == PH_PREC_BF16X3 || (p) == PH_PREC_FP16X3)   // conv outputs are fp32

#define PH_LAUNCH_CHECK()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return PH_ELAUNCH;              \
  } while (0)

template <typename T> struct is_f32 { static constexpr bool value = std::is_same<T, float>::value; };

// ---- half-pair ("HP") storage of PH_PREC_FP16X3.  A tensor [..][C] (C % 64 == 0, base 256-B aligned) is stored as
// [..][C / 64][2][64] fp16: per 64-channel slice one 128-B line of hi values followed by one 128-B line of lo values,
//     x ~= hi + lo * 2^-11,   hi = fp16(x),   lo = fp16((x - hi) * 2^11)      (22 significant bits, |x| < 65504)
// 4 bytes per element, so `hp16` is a 4-byte placeholder type: pointer arithmetic on `hp16*` is the fp32 tensor's, and the
// address of element i's hi half follows from the pointer alone (the byte offset inside a 256-B slice record is halved).
// A convolution's LDS-DMA reads the hi (or lo) plane of a slice as whole 128-B lines: no conversion while staging.
struct hp16 { unsigned int raw; };
template <typename T> struct is_hp { static constexpr bool value = std::is_same<T, hp16>::value; };
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
constexpr float PH_HP_LO = 2048.0f, PH_HP_LO_INV = 1.0f / 2048.0f;
__device__ __forceinline__ const unsigned char* hp_hi_addr(const hp16* p) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  return reinterpret_cast<const unsigned char*>((a & ~(uintptr_t)255) + ((a & 255) >> 1));
}
__device__ __forceinline__ void hp_split(float x, f16& hi, f16& lo) {
  // (saturating: a value beyond the fp16 range degrades to +-65504 instead of inf -> NaN gradients; the dz tensors' power-of-two
  // scale keeps real data far below it)
  // (v_med3_f32 returns the minimum of the other two operands for a NaN input: a NaN must stay a NaN so that a diverged run is
  // visible - ADVICE r05)
  x = (x != x) ? x : __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  hi = (f16)x;
  lo = (f16)((x - (float)hi) * PH_HP_LO);
}

// A load through a pointer the compiler cannot prove to be global - `in_image ? tensor + offset : zero_page` selects between a
// kernel argument and a __device__ variable and degrades to FLAT instructions: those count in vmcnt AND lgkmcnt and return out of
// order, so every wait on them is a full drain (vmcnt(0) + lgkmcnt(0)) of the wave's memory queue - outstanding output stores
// and LDS traffic included (round 6: all stem kernels prefetched their halos this way).  ld_global states the address space.
template <typename V>
__device__ __forceinline__ V ld_global(const void* p) {
  typedef __attribute__((address_space(1))) const V GV;
  return *reinterpret_cast<GV*>(reinterpret_cast<unsigned long long>(p));
}
template <typename V>
__device__ __forceinline__ void st_global(void* p, const V& v) {
  typedef __attribute__((address_space(1))) V GV;
  *reinterpret_cast<GV*>(reinterpret_cast<unsigned long long>(p)) = v;
}

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return (float)*p; }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16* p, float v) { *p = (bf16)v; }
// (hp_hi_addr goes through an integer: the accesses state the global address space themselves - ld_global / st_global above)
__device__ __forceinline__ float ldf(const hp16* p) {
  const unsigned char* h = hp_hi_addr(p);
  return (float)ld_global<f16>(h) + (float)ld_global<f16>(h + 128) * PH_HP_LO_INV;
}
__device__ __forceinline__ void stf(hp16* p, float v) {
  unsigned char* h = const_cast<unsigned char*>(hp_hi_addr(p));
  f16 a, b;
  hp_split(v, a, b);
  st_global<f16>(h, a); st_global<f16>(h + 128, b);
}

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

// 8 consecutive channels (p 32-B aligned in the fp32 view): one 16-B hi chunk + one 16-B lo chunk, 128 B apart
__device__ __forceinline__ void load8(const hp16* p, float (&v)[8]) {
  const unsigned char* a = hp_hi_addr(p);
  const f16x8 h = ld_global<f16x8>(a);
  const f16x8 l = ld_global<f16x8>(a + 128);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i] + (float)l[i] * PH_HP_LO_INV;
}
__device__ __forceinline__ void store8(hp16* p, const float (&v)[8]) {
  unsigned char* a = const_cast<unsigned char*>(hp_hi_addr(p));
  f16x8 h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) { f16 x, y; hp_split(v[i], x, y); h[i] = x; l[i] = y; }
  st_global<f16x8>(a, h);
  st_global<f16x8>(a + 128, l);
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
