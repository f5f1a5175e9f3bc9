// On-device contrast-index sampler: the rule of the reference's dataset class
// (MICCAI-2022/data_loaders_MT.py:229-249; MIA-2023 neg_mode variants, "MIA 2023/stage2_unimodal_student/
// data_loaders_MT.py":205-238) that draws, for every query of a batch, its positive columns from the bank rows of the same
// class and its negative columns from the other classes (or from every other row).
//
// The reference draws with numpy's global Mersenne-Twister inside the DataLoader workers (np.random.choice); a device
// sampler cannot reproduce that stream, so parity here is distributional: every column is uniform over the same candidate
// list, "replace=False" draws are distinct, slot 0 is the query's own index, "replace=True" is used exactly when
// K exceeds the list (:243).  Sampling WITHOUT replacement = the first m outputs of a keyed pseudo-random permutation of
// [0, n): a 4-round Feistel network over the next even power of two with cycle walking - O(1) per drawn index, no sort,
// no rejection bitmap, bit-reproducible for a given (seed, step, query).  Sampling with replacement = one hash per slot.
#include "ph_common.h"
#include "ph_kernels.h"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// pseudo-random permutation of [0, n), evaluated at j; `key` selects the permutation
__device__ __forceinline__ unsigned perm_at(unsigned j, unsigned n, uint64_t key) {
  int bits = 32 - __clz(n > 1 ? n - 1 : 1);
  bits += bits & 1;                               // balanced halves
  const int hb = bits >> 1;
  const unsigned hm = (1u << hb) - 1u;
  unsigned x = j;
  do {
    unsigned l = x >> hb, r = x & hm;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
      const unsigned f = (unsigned)mix64(key + 0x9E3779B97F4A7C15ull * (uint64_t)(round + 1) + r) & hm;
      const unsigned t = l ^ f;
      l = r; r = t;
    }
    x = (l << hb) | r;
  } while (x >= n);                               // cycle walking: the walk of a point < n returns below n
  return x;
}

struct SamplerArgs {
  const int64_t* index; const int64_t* grade;
  const int* cls_pos; const int* cls_pos_off; const int* cls_neg; const int* cls_neg_off;
  int n_data, B, P, K, pos_mode, neg_mode;
  uint64_t seed; const uint64_t* step;
  int64_t* out;
};

__global__ __launch_bounds__(256) void contrast_sampler_kernel(SamplerArgs a) {
  const int b = blockIdx.x;
  const int64_t own = a.index[b];
  const int g = (int)a.grade[b];
  const int np = a.pos_mode == 2 ? a.P : 1;
  const int S = np + a.K;
  int64_t* row = a.out + (size_t)b * S;
  const uint64_t step = a.step ? *a.step : 0;
  const uint64_t base = mix64(a.seed ^ mix64(step * 0x9E3779B97F4A7C15ull + (uint64_t)b + 1));
  // ---- positives (:229-239)
  const int* pl = a.cls_pos + a.cls_pos_off[g];
  const unsigned npos = (unsigned)(a.cls_pos_off[g + 1] - a.cls_pos_off[g]);
  if (a.pos_mode == 0) {                                   // 'exact'
    if (threadIdx.x == 0) row[0] = own;
  } else if (a.pos_mode == 1) {                            // 'relax': one uniform same-class row
    if (threadIdx.x == 0) row[0] = pl[mix64(base ^ 0x51ull) % npos];
  } else {                                                 // 'multi_pos': P distinct same-class rows, slot 0 := the query
    for (int j = threadIdx.x; j < np; j += blockDim.x) row[j] = j == 0 ? own : pl[perm_at((unsigned)j, npos, base ^ 0xA5ull)];
  }
  // ---- negatives (:241-243; MIA-2023 neg_mode)
  const int* nl = a.cls_neg + a.cls_neg_off[g];
  const unsigned nneg = a.neg_mode == 1 ? (unsigned)(a.n_data - 1) : (unsigned)(a.cls_neg_off[g + 1] - a.cls_neg_off[g]);
  const bool replace = (unsigned)a.K > nneg;
  for (int j = threadIdx.x; j < a.K; j += blockDim.x) {
    const unsigned t = replace ? (unsigned)(mix64(base ^ (0xC3ull + 0x9E3779B97F4A7C15ull * (uint64_t)(j + 1))) % nneg)
                               : perm_at((unsigned)j, nneg, base ^ 0x3Cull);
    int64_t v;
    if (a.neg_mode == 1) v = (int64_t)t + ((int64_t)t >= own ? 1 : 0);   // every row but the query itself
    else v = nl[t];
    row[np + j] = v;
  }
}

// DataLoader(shuffle=True, drop_last=True) batch indices (train_cv_path_multi_MT.py / data_loaders_MT.py: the sampler of
// the training loader): batch number `*batch_no` of an endless run takes rows q .. q+B-1 of its epoch's permutation of
// [0, n); the permutation of epoch e is the keyed Feistel permutation perm_at(., n, key(seed, e)) - O(1) per index.
__global__ void shuffle_indices_kernel(int64_t* __restrict__ out, int n, int B, uint64_t seed, const uint64_t* __restrict__ batch_no) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= B) return;
  const uint64_t per_epoch = (uint64_t)(n / B);              // drop_last
  const uint64_t bn = batch_no ? batch_no[0] : 0;
  const uint64_t epoch = bn / per_epoch, q = (bn % per_epoch) * (uint64_t)B + (uint64_t)j;
  out[j] = (int64_t)perm_at((unsigned)q, (unsigned)n, mix64(seed ^ mix64(epoch + 0x5bd1e995ull)));
}

// ContrastMemory_v3.forward with idx == None (memory_new.py:265-267): AliasMethod(torch.ones(n_data)).draw(B * (K + P)) viewed
// [B, K + P], column 0 := y.  With uniform unigrams every alias-table entry has probability 1 (:418-440), torch.bernoulli(1) is 1
// and the draw IS the uniform integer draw kk (:450-458): one hash per slot, with replacement.  Distributional parity (the
// reference draws from torch's CUDA generator); bit-reproducible for a given (seed, step).
__global__ void alias_uniform_draw_kernel(const int64_t* __restrict__ y, int64_t* __restrict__ out, int n_data, int B, int S,
                                          uint64_t seed, const uint64_t* __restrict__ step) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * S) return;
  const int b = (int)(i / S), j = (int)(i - (size_t)b * S);
  const uint64_t stp = step ? *step : 0;
  const uint64_t h = mix64(mix64(seed ^ (stp * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull)) + i * 0xD1B54A32D192ED03ull);
  // (multiply-shift instead of %: unbiased to 2^-32 for n_data < 2^32)
  out[i] = j == 0 ? y[b] : (int64_t)(((h >> 32) * (uint64_t)(unsigned)n_data) >> 32);
}

}  // namespace

extern "C" int ph_alias_uniform_draw(const int64_t* y, int64_t* out, int n_data, int B, int S, uint64_t seed, const uint64_t* step,
                                     hipStream_t st) {
  if (!y || !out || n_data < 1 || B < 1 || S < 1) return PH_EINVAL;
  const size_t n = (size_t)B * S;
  hipLaunchKernelGGL(alias_uniform_draw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, out, n_data, B, S, seed, step);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

extern "C" int ph_shuffle_indices(int64_t* out, int n, int B, uint64_t seed, const uint64_t* batch_no, hipStream_t st) {
  if (!out || n < 1 || B < 1 || B > n) return PH_EINVAL;
  hipLaunchKernelGGL(shuffle_indices_kernel, dim3((B + 255) / 256), dim3(256), 0, st, out, n, B, seed, batch_no);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

extern "C" int ph_contrast_sampler(const int64_t* index, const int64_t* grade, const int* cls_pos, const int* cls_pos_off,
                                   const int* cls_neg, const int* cls_neg_off, int n_data, int B, int P, int K,
                                   int pos_mode, int neg_mode, uint64_t seed, const uint64_t* step, int64_t* out,
                                   hipStream_t st) {
  if (!index || !grade || !cls_pos || !cls_pos_off || !out || B <= 0 || K <= 0) return PH_EINVAL;
  if (neg_mode != 1 && (!cls_neg || !cls_neg_off)) return PH_EINVAL;
  if (pos_mode < 0 || pos_mode > 2 || neg_mode < 0 || neg_mode > 1) return PH_EINVAL;
  SamplerArgs a{index, grade, cls_pos, cls_pos_off, cls_neg ? cls_neg : cls_pos, cls_neg_off ? cls_neg_off : cls_pos_off,
                n_data, B, P, K, pos_mode, neg_mode, seed, step, out};
  hipLaunchKernelGGL(contrast_sampler_kernel, dim3(B), dim3(256), 0, st, a);
  PH_LAUNCH_CHECK();
  return PH_OK;
}
