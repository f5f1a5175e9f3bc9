// Internal launcher prototypes shared by the .hip translation units (not part of the public C-ABI;
// the public boundary is include/pathomic_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// in-library kernel timer (prof.hip); classes are indices of ph_prof_summary()
#define PH_CLS_TAPCONV_N64 0    // tapconv_kernel<.., S=1, BNT=64 ..>   (first generation: Cout = 64 dgrad parity classes)
#define PH_CLS_TAPCONV_N128 1   // tapconv_kernel<.., S=1, BNT=128 ..>  (first generation: 1x1 and dgrad parity classes, Cout >= 128)
#define PH_CLS_TAPCONV_S2 2     // tapconv_kernel<.., S=2 ..>
#define PH_CLS_WGRAD 3
#define PH_CLS_STEM_FWD 4
#define PH_CLS_STEM_WGRAD 5
#define PH_CLS_TAPCONV2 6       // tapconv2_kernel<2,2,4,false>: 3x3 stride-1 fwd + dgrad, Cout >= 128  (the dominant kernel)
#define PH_CLS_TAPCONV2_RES 7   // tapconv2_kernel<4,1,2,true>: 3x3 stride-1 fwd + dgrad, Cin = Cout = 64 (layer 1)
// HBM-bound classes: the `work` of these is algorithmic BYTES, not FLOPs
#define PH_CLS_CRD_SCORE 8      // crd_score_kernel: 2 banks x B x (P+K) rows of 512 B
#define PH_CLS_CRD_LOSSGRAD 9   // crd_loss_grad_kernel (+ its reduce): 2 banks x B x (P2+K2) rows of 512 B
#define PH_CLS_ADAM_EMA 10      // adam_ema(_dev)_kernel: 28 B per parameter + 8 B per EMA parameter
#define PH_CLS_BN_APPLY 11      // bn_apply_kernel: y (+ residual | + downsample y) read, activation written
#define PH_CLS_TAPCONV2_MASKED 12   // tapconv2_kernel<2,2,4,false,MASKED>: stride-2 fwd / merged-class dgrad as masked stride-1 grids
#define PH_CLS_TAPCONV2_FUSEDIN 13   // class 6 launches that also apply their input's BatchNorm + ReLU in LDS (PhTapConv::in_scale)
#define PH_CLS_TAPCONV2_RES_FUSEDIN 14   // class 7 (layer-1 kernel) launches that do
#define PH_CLS_CRD_TOPK 15      // crd_bank_topk (+ merge): class-masked full-bank cosine KNN, 2 banks x n_data rows of 512 B (each once)
#define PH_NCLS 16
#define PH_NUM_CLS 6
bool ph_prof_on();
int ph_num_cus();   // compute units of the current device (cached)
void ph_prof_begin(int cls, double work, hipStream_t st, void** token);
// + the launch's ALGORITHMIC HBM bytes (input + output activations + packed weights, each counted once)
void ph_prof_begin2(int cls, double work, double bytes, hipStream_t st, void** token);
void ph_prof_end(void* token, hipStream_t st);

struct PhTapConv {
  const void* in;        // [B][IH][IW][Cin]   activation type T of the precision mode
  const void* w;         // [nplanes][nslabs][Cout][Cin] bf16 (1 plane in perf mode, 3 in parity mode)
  size_t wplane;         // elements per plane
  void* out;             // [B][OH][OW][Cout]
  float* stats;          // [B*tiles][2][Cout] per-workgroup sum / sum-of-squares partials, or null
  const void* res_g;     // optional: out += res_g * (res_a > 0 | 1)   (dgrad residual fusion)
  const void* res_a;
  // optional (second-generation stride-1 kernels only): `in` is the RAW output of a convolution whose BatchNorm + ReLU
  // has not been applied - relu(in * in_scale[c] + in_shift[c]) is applied to every halo tile in LDS right after its
  // LDS-DMA has landed (no separate bn_apply pass, no second activation tensor in HBM).  Forward-only networks.
  const float* in_scale; const float* in_shift;
  int B, IH, IW, Cin, Cout;
  int OHt, OWt;          // extent of the (r,c) output-position space this launch covers
  int OH, OW;            // full output tensor dims; output pixel = (r*os+oa_h, c*os+oa_w)
  int os, oa_h, oa_w;
  int iy0, ix0;          // input pixel of tap t for position (r,c) = (r*S + dy[t] + iy0, c*S + dx[t] + ix0)
  int ntaps;
  int dy[9], dx[9];      // 0..2
  int wtap[9];           // weight slab index of each tap
  // element strides of the input view (0 = dense NHWC: Cin, IW*Cin, IH*IW*Cin).  A 1x1 stride-2 convolution is run
  // as a 1x1 stride-1 tap-conv over the view {pixel stride 2*Cin, row stride 2*IW*Cin} (no wasted halo pixels).
  long in_pix_stride, in_row_stride, in_img_stride;
  // optional MASKED 3x3 tap grid (second-generation kernel <2,2,4,false> only; m_groups = 4, 0 = off): a stride-2 3x3
  // FORWARD convolution as a stride-1 tap-conv over the four pixel-parity planes of its input - strided views
  // (in_pix/row_stride = 2 pixels / 2 rows) starting m_in_off[g] elements into `in`; the K loop walks 4 * Cin/64
  // slices; in plane g only the grid taps of m_mask[g] are live (dead taps issue their operand DMAs - the pipeline's
  // bookkeeping stays - but no MFMAs).  Filled by ph_tapconv2_setup_s2_fwd.
  int m_groups;
  int m_mask[4];         // bit t: grid tap t (row-major, dy = t / 3, dx = t % 3) is live (fixed sets, see the kernel)
  int m_slab[4][9];      // weight slab of every grid tap (any valid slab for dead taps)
  long m_in_off[4];
  int prod6;             // split-plane modes (first-generation kernel): six products or the three leading ones (set by the launcher)
  // PH_PREC_FP16X3: `in` is a half-pair tensor (ph_common.h), `w` holds per 64-channel slice of Cin the three fp16 blocks
  // [hi * 2^11 | lo | hi] of the weights' half-pair split (ph_pack_all_launch, nplanes = -3), the K loop walks 3 * Cin / 64 (A block, W block) pairs
  // (x hi, w hi 2^11), (x hi, w lo), (x lo, w hi) and the fp32 result is acc * 2^-11 * (in_unscale ? in_unscale[1] : 1):
  // in_unscale = the {2^s, 2^-s} record of a dz tensor (ph_bn_bwd_finalize_launch), null for activations
  const float* in_unscale;
  int hp_hi_only;        // half-pair kernels: 1 = the hi planes' product alone (A block hi x W block hi: one slice per 64 channels,
                         // result = acc * in_unscale[1]); set by the launcher for PH_PREC_FP16X1
  // optional (perf-mode stride-1 dgrad launches of conv_tap4.hip / conv_tap3.hip): the BatchNorm-backward sums of the tensor this
  // launch WRITES, taken in its epilogue instead of by a separate bn_bwd_reduce pass.  bst_y: the raw output y of that BatchNorm's
  // convolution (same shape as `out`); the ReLU mask is (bst_a > 0) when bst_a is given (a block output: bn2 / downsample) and
  // (bst_y * bst_scale + bst_shift > 0) otherwise (the BatchNorm's own ReLU: bn1); dz = out * mask.  `stats` then receives per
  // workgroup [3][Cout]: sum dz, sum dz (bst_y - bst_mean), sum dz (bst_y2 - bst_mean2) - bst_y2 (optional) = the downsample
  // branch's raw output, whose BatchNorm backward reduces the same dz.  ph_bn_bwd_finalize_launch combines the rows.
  const void* bst_y; const void* bst_a; const void* bst_y2;
  const float* bst_scale; const float* bst_shift; const float* bst_mean; const float* bst_mean2;
  // optional (first-generation kernel, stride-1 configurations): ONE launch for the output-parity classes of a stride-2 dgrad
  // (round 6).  ncls in 2..4 (0 = off): class k = blockIdx.z / B covers the positions (r * os + c_oa_h[k], c * os + c_oa_w[k]),
  // r < c_OHt[k], c < c_OWt[k], with c_ntaps[k] <= 4 taps (c_dy / c_dx / c_wtap).  The classes write disjoint pixels of `out`
  // and read the same `in`: as separate launches (1 / 2 / 2 / 4 taps each) layers 3.0 / 4.0 filled 256 / 128 of the 256 CUs
  // four times in a row.  OHt / OWt / ntaps / oa_* / dy / dx / wtap above are then the largest class's (grid, profiler).
  int ncls;
  int c_ntaps[4], c_oa_h[4], c_oa_w[4], c_OHt[4], c_OWt[4];
  int c_dy[4][4], c_dx[4][4], c_wtap[4][4];
};
int ph_tapconv_launch(const PhTapConv* p, int S, int prec, hipStream_t st);
double ph_tapconv_bytes(const PhTapConv& p, int S, int es);
int ph_tapconv_stat_parts(const PhTapConv* p, int S, int prec);
// second-generation stride-1 perf-mode kernel (conv_tap2.hip): tile height of the configuration it would run, 0 = not eligible
int ph_tapconv2_tile_h(const PhTapConv* p, int S, int prec);
int ph_tapconv2_launch(const PhTapConv* p, hipStream_t st);
int ph_tapconv2_stat_parts(const PhTapConv* p);   // one BatchNorm partial row per persistent workgroup
// third-generation dense 3x3 stride-1 kernel (conv_tap3.hip: 16x16x32 fragments, 8-byte stores); PH_TAP3=0 in the environment
// keeps the second-generation kernel (same-box A/B)
int ph_tap3_switch(int set);
bool ph_tapconv3_eligible(const PhTapConv* p);
int ph_tapconv3_launch(const PhTapConv* p, hipStream_t st);
int ph_tapconv3_launch_hp(const PhTapConv* p, hipStream_t st);      // PH_PREC_FP16X3 form of the same kernel
// fourth-generation kernel for Cin = Cout = 64 (conv_tap4.hip: ResNet layer 1; one wave per SIMD, 16x16x32 fragments, resident
// weights, one barrier per tile, optional fused BatchNorm-backward sums); PH_TAP4=0 keeps tapconv2_l1_kernel (same-box A/B)
int ph_tap4_switch(int set);
bool ph_tapconv4_eligible(const PhTapConv* p);
int ph_tapconv4_launch(const PhTapConv* p, hipStream_t st);
// half-pair kernel for Cin = Cout = 64 (conv_tap5.hip: both halo planes of a 32 x 16 tile resident in LDS, weight fragments from
// global memory into a rotating register window, two barriers per tile); p->hp_hi_only selects the hi-only form; PH_TAP5=0 keeps
// the first-generation kernel (same-box A/B)
int ph_tap5_switch(int set);
bool ph_tapconv5_eligible(const PhTapConv* p);
int ph_tapconv5_launch(const PhTapConv* p, hipStream_t st);
int ph_tapconv5_stat_parts(const PhTapConv* p);
// (its 64 x 64 x 9 weight slabs are packed fragment-major while the switch is on: pack_all_tiled_hp_kernel, conv_wgrad.hip - do not
// change the switch between a pack and the launches that read it)
// perf-mode kernel of the dense 3x3 stride-1 convolutions with Cin = Cout >= 128 (conv_tap7.hip: conv_tap3.hip's plain form on the
// register-window machinery, bitwise the same outputs); reads the fragment-major copy of the weights in plane 1 of the unit's packed
// region (p->w + p->wplane elements; pack_all_tiled_kernel<1> writes it, ph_frag7_repack_launch for a single convolution)
int ph_tap7_switch(int set);
bool ph_tapconv7_eligible(const PhTapConv* p);
int ph_tapconv7_launch(const PhTapConv* p, hipStream_t st);
int ph_frag7_repack_launch(void* packed, int R, int K, int ntaps, hipStream_t st);
// half-pair kernel of the 3x3 / stride-2 forward convolutions (conv_tap6.hip: parity-plane images through four LDS buffers on a
// compile-time DMA schedule, conv_tap5.hip's weight window); PH_TAP6=0 keeps the first-generation kernel; its weights are packed
// fragment-major while the switch is on (same caveat as PH_TAP5)
int ph_tap6_switch(int set);
bool ph_tapconv6_eligible(const PhTapConv* p);
int ph_tapconv6_launch(const PhTapConv* p, hipStream_t st);
int ph_tapconv6_stat_parts(const PhTapConv* p);
// ... and its perf-mode (bf16) form (conv_tap6b.hip); reads the fragment-major copy in plane 1 of the unit's packed region; PH_TAP6B=0
// keeps conv_tap2.hip's masked grid
int ph_tap6b_switch(int set);
bool ph_tapconv6b_eligible(const PhTapConv* p);
int ph_tapconv6b_launch(const PhTapConv* p, hipStream_t st);
int ph_tapconv6b_stat_parts(const PhTapConv* p);
// stride-2 3x3 convolutions as masked stride-1 tap grids (conv_tap2.hip); false = not eligible, descriptor untouched
bool ph_tapconv2_setup_s2_fwd(PhTapConv* t, int Cin, int Cout, int IH, int IW, int prec);

struct PhWgrad {
  const void* x;         // [B][IH][IW][Cin]
  const void* dy;        // [B][OH][OW][Cout]
  float* slab;           // [nchunks][KS*KS][Cout][Cin] fp32 partial sums
  const void* zeros;     // >= 16 B of device zeros (source of out-of-image pixels for the LDS-DMA path)
  int B, IH, IW, Cin, OH, OW, Cout;
  int S, pad, KS;
  int nchunks, tiles_per_chunk;
  long x_pix_stride, x_row_stride, x_img_stride;   // element strides of the x view (0 = dense NHWC)
  int prod6;             // split-plane modes: 1 = all six products (bf16x6), 0 = the three leading ones (bf16x3); set by the launcher
  int hp_hi_only;        // half-pair mode: 1 = one pass over the chunk on the hi planes (PH_PREC_FP16X1); set by the launcher
};
int ph_wgrad_launch(const PhWgrad* p, int prec, hipStream_t st);
int ph_wgrad_tile_h(int S);
// slab -> OIHW fp32 gradient; unscale (optional): the {2^s, 2^-s} record of the dz tensor, the sums are multiplied by unscale[1]
int ph_wgrad_reduce_launch(const float* slab, float* dw_oihw, int nchunks, int KS, int Cout, int Cin, const float* unscale,
                           hipStream_t st);

struct PhStem {
  const void* x4;        // [B][IH][IW][4]
  const void* w;         // [nplanes][7][64][32] bf16: k = kw*4 + ch, kw==7 and ch==3 are zero
  size_t wplane;
  void* out;             // [B][OH][OW][64]
  float* stats;          // [B*tiles][2][64]
  int B, IH, IW, OH, OW;
  int prod6;             // split-plane modes: six products or the three leading ones (set by the launcher)
  int vblocks;           // > 0: the grid is smaller than the `vblocks` blocks of STEM_TPW tiles - workgroup b walks blocks b, b + grid, ..
                         // (one statistics row per BLOCK either way: rows and summation order do not depend on the grid)
};
int ph_stem_fwd_launch(const PhStem* p, int prec, hipStream_t st);
int ph_stem_stat_parts(int B, int OH, int OW);
// forward-only networks, perf mode: the same convolution with the 3x3 / stride-2 / pad-1 max-pool applied to its RAW
// output in the epilogue - the [B][OH][OW][64] conv output is never written.  pooled [B][PH][PW][64] bf16 receives, per
// channel, the window's MAXIMUM of the bf16-rounded conv output where gamma >= 0 and its MINIMUM where gamma < 0:
// relu(scale * y + shift) is monotone in y with the sign of scale = gamma * invstd = the sign of gamma, so BatchNorm +
// ReLU applied to this tensor afterwards (bn_apply, in place) gives bitwise the separate pooling pass's result.
// stats: BatchNorm partial sums over ALL conv pixels (each counted once), ph_stem_pool_stat_parts() rows of [2][64].
struct PhStemPool {
  const void* x4; const void* w; size_t wplane;
  void* pooled; float* stats; const float* gamma;
  int B, IH, IW, OH, OW, PH, PW, nsplit;
};
int ph_stem_fwd_pool_launch(const PhStemPool* p, hipStream_t st);
int ph_stem_pool_stat_parts(int B, int OH, int OW);

struct PhStemWgrad {
  const void* x4; const void* dy; float* slab;   // slab [nchunks][7][64][32]
  int B, IH, IW, OH, OW;
  int nchunks, tiles_per_chunk;
  int prod6;
  int hp_hi_only;        // half-pair mode: the hi planes' product alone (PH_PREC_FP16X1); set by the launcher
};
int ph_stem_wgrad_launch(const PhStemWgrad* p, int prec, hipStream_t st);
// input gradient of the 7x7/2 stem conv: dy [B][H/2][W/2][64] (type of the mode) -> dx [B][3][H][W] f32 (stem_dgrad.hip)
int ph_stem_dgrad_launch(const void* dy, const float* w_oihw, float* dx_nchw, int B, int H, int W, int prec, hipStream_t st);
int ph_stem_wgrad_reduce_launch(const float* slab, float* dw_oihw, int nchunks, const float* unscale, hipStream_t st);

// weight packing (OIHW fp32 -> MFMA-friendly bf16, 3 split planes of `plane` elements each)
int ph_pack_w_fwd_launch(const float* w, void* planes, int O, int I, int KS, hipStream_t st);   // [tap][O][I]
int ph_pack_w_dgrad_launch(const float* w, void* planes, int O, int I, int KS, hipStream_t st); // [tap][I][O]
int ph_pack_w_stem_launch(const float* w, void* planes, hipStream_t st);                        // [7][64][32]

// all 3x3 / 1x1 conv weights of one network in one launch (stem handled by ph_pack_w_stem_launch)
struct PhPackAll {
  const float* w[20];
  size_t dst_fwd[20], dst_dg[20], start[21];
  int O[20], I[20], NT[20];
  int n;
  size_t total;
};
// nplanes: 1 (bf16), 3 (three bf16 split planes), -3 (PH_PREC_FP16X3: fp16 [hi 2^11 | lo | hi] per 64-channel K slice, (hi, lo) = the half-pair split)
int ph_pack_all_launch(const PhPackAll* t, void* packed, int nplanes, hipStream_t st);
int ph_pack_w_hp_launch(const float* w, void* packed, int O, int I, int KS, int dgrad, hipStream_t st);   // one conv, same layout
int ph_pack_w_stem_hp_launch(const float* w, void* packed, hipStream_t st);

// ---- BatchNorm / elementwise (bn_act.hip).  `prec` selects the activation type (bf16 | float).
int ph_pack_input_launch(const float* x_nchw, void* x4, int B, int H, int W, int prec, hipStream_t st);
// partial slab [nparts][2][C] -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale
// (+ running stats update when running_mean != null)
int ph_bn_finalize_launch(const float* parts, int nparts, int C, double count, float eps, float momentum,
                          const float* gamma, const float* beta, float* mean, float* invstd, float* scale,
                          float* shift, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          hipStream_t st);
struct PhBnEvalTable {
  const float* gamma[20]; const float* beta[20]; const float* running_mean[20]; const float* running_var[20];
  float* mean[20]; float* invstd[20]; float* scale[20]; float* shift[20];
  int C[20];
  int n;
};
int ph_bn_eval_params_launch(const PhBnEvalTable* t, float eps, hipStream_t st);
// out = relu?( y*scale + shift + [res | y_r*scale_r + shift_r] ).  PH_PREC_FP16X3: `out` is the half-pair operand image,
// `out32` (optional) the fp32 copy the elementwise passes read; `res` is such an fp32 copy.  Other modes: out32 = null.
int ph_bn_apply_launch(const void* y, const float* scale, const float* shift, const void* res, const void* y_r,
                       const float* scale_r, const float* shift_r, void* out, void* out32, size_t npix, int C, int relu, int prec,
                       hipStream_t st);
// res_as_t: `res` is an activation as the convolutions read it (PH_PREC_FP16X3: the half-pair image instead of the fp32 copy)
int ph_bn_apply_launch2(const void* y, const float* scale, const float* shift, const void* res, const void* y_r,
                        const float* scale_r, const float* shift_r, void* out, void* out32, size_t npix, int C, int relu, int prec,
                        int res_as_t, hipStream_t st);
int ph_avgpool_launch_t(const void* x, float* out, int B, int HW, int C, int prec, hipStream_t st);
// raw (optional, with idx): the conv output at every window's arg-max, [B][OH/2][OW/2][C] of the mode's type
int ph_bn_relu_maxpool_launch(const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, void* raw,
                              void* out32, int B, int H, int W, int C, int prec, hipStream_t st);
// x: an activation as the elementwise passes read it (the fp32 copy in PH_PREC_FP16X3)
int ph_avgpool_launch(const void* x, float* out, int B, int HW, int C, int prec, hipStream_t st);
// d_x (+)= g / HW  broadcast
int ph_avgpool_bwd_launch(const float* g, void* dx, int B, int HW, int C, int accumulate, int prec, hipStream_t st);
// BN backward: dz = g * (a > 0 ? 1 : 0) (a may be null).  reduce -> parts [nparts][2][C] (sum dz, sum dz*xhat)
int ph_bn_bwd_parts(size_t npix, int C);   // <= 1024
int ph_stem_bwd_parts(int B, int H);         // partial rows written by ph_stem_bwd_reduce_launch
// mscale / mshift (optional, with a == null): mask = (y * mscale + mshift > 0), the ReLU of this BN's own output
// amax (optional): [ph_bn_bwd_parts()] per-block max |dz|, the input of the dz scale below
int ph_bn_bwd_reduce_launch(const void* g, const void* a, const void* y, const float* mean, const float* invstd,
                            float* parts, size_t npix, int C, int prec, const float* mscale, const float* mshift,
                            float* amax, hipStream_t st);
// parts -> dgamma, dbeta, c1 = mean(dz), c2 = mean(dz*xhat).  dzs (optional, PH_PREC_FP16X3, with amax[namax], gamma, invstd):
// float[2] = {2^s, 2^-s}, the power-of-two scale under which the apply pass stores its output as fp16 pairs
int ph_bn_bwd_finalize_launch(const float* parts, int nparts, int C, double count, float* dgamma, float* dbeta,
                              float* c1, float* c2, const float* amax, int namax, const float* gamma, const float* invstd,
                              float* dzs, hipStream_t st);
// the same from the [nparts][3][C] rows of a convolution's fused sums (PhTapConv::bst_y): sum dz | sum dz (y - mean) | sum dz (y2 - mean2)
int ph_bn_bwd_finalize_fused_launch(const float* parts, int nparts, int C, double count, float* dgamma, float* dbeta,
                                    float* c1, float* c2, const float* invstd, int row2, hipStream_t st);
int ph_bn_bwd_apply_launch(const void* g, const void* a, const void* y, const float* mean, const float* invstd,
                           const float* gamma, const float* c1, const float* c2, void* dy, size_t npix, int C,
                           int prec, const float* mscale, const float* mshift, const float* dzs, hipStream_t st);
// stem: da0 = scatter of d_pool through the saved argmax, dz = da0 * (bn(y0) > 0)
int ph_stem_bwd_reduce_launch(const void* dpool, const uint8_t* idx, const void* y0, const void* raw_at_argmax, const float* mean,
                              const float* invstd, const float* scale, const float* shift, float* parts, int B, int H,
                              int W, int C, int prec, float* amax, hipStream_t st);
int ph_stem_bwd_apply_launch(const void* dpool, const uint8_t* idx, const void* y0, const float* mean,
                             const float* invstd, const float* scale, const float* shift, const float* gamma,
                             const float* c1, const float* c2, void* dy0, int B, int H, int W, int C, int prec,
                             const float* dzs, hipStream_t st);
