/* pathomic_hip.h - C-ABI of libpathomic_hip.so (MI355X / gfx950).
 *
 * The reference (CityU-AIM-Group/MultiModal-learning) has no FFI layer: its hot path is pure Python on
 * torch built-ins.  This header is therefore the boundary a maintainer binds
 * (ctypes stub in INTEGRATION.md) to replace the torch ops under the reference's Python module API;
 * each entry cites the reference site (path relative to /root/reference/MICCAI-2022) it replaces.
 *
 * Conventions: plain pointers to DEVICE memory + sizes, no torch types; every call enqueues work on
 * `stream` and returns immediately; return 0 on success, negative errno-style code otherwise
 * (-22 invalid argument, -5 launch failure, -16 busy).  No device memory is allocated inside: callers pass workspaces.
 * Thread-compatible (one stream per call).  Process-wide state the library owns: per device ONE side stream and a pool of
 * 16 x 64 events for the two-stream trunk backward (created by the first ph_resnet_backward* call outside a stream capture,
 * never destroyed; a 17th backward in flight at once gets -16), the records of the in-library kernel timer (ph_prof_*), and
 * per-kernel "attributes set" flags.  A plan (PhResnetPlan) additionally remembers, per workspace, the pre-packed input of
 * the last forward that ran on it.
 */
#ifndef PATHOMIC_HIP_H_
#define PATHOMIC_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ph_stream_t; /* a hipStream_t */

#define PH_PREC_BF16 0   /* perf mode: bf16 operands + activations, fp32 accumulate/statistics */
#define PH_PREC_BF16X6 1 /* parity mode: fp32 activations, 3-plane split-bf16 (6-product, fp32-equivalent) MFMA */
#define PH_PREC_BF16X3 2 /* fp32 activations, the 3 leading split-bf16 products (16-bit operands): half the matrix work */
#define PH_PREC_FP16X3 3 /* half-pair mode: tensors a convolution reads are fp16 pairs x = hi + lo * 2^-11 (4 B / element, 22
                            significant bits), conv outputs / gradients fp32, 3 fp16 MFMA products (hi*hi, hi*lo, lo*hi) */

#define PH_PREC_FP16X1 4 /* only as the backward arithmetic of a PH_PREC_FP16X3 plan (set_backward_prec below): dgrad / wgrad on the hi planes alone */

#define PH_ACT_NONE 0
#define PH_ACT_RELU 1
#define PH_ACT_ELU 2
#define PH_ACT_SIGMOID 3

#define PH_EW_RELU 0
#define PH_EW_GATE 1
#define PH_EW_RELU_BWD 2
#define PH_EW_ADD 3
#define PH_EW_ELU_BWD 4  /* a * ELU'(x) given b = ELU(x) */
#define PH_EW_MUL 5

int ph_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * ResNet-18 trunk: conv7x7/2-BN-ReLU-maxpool, 8 BasicBlocks, global avg-pool of layer3 and layer4.
 * Replaces ResNet._forward_impl up to the pooled features (resnets.py:217-236, BasicBlock :58-74) and
 * its autograd backward.  Train-mode BatchNorm everywhere (train_test_path_multi_distill.py:231-232).
 *
 * params: 20 units x 6 pointers [conv weight OIHW f32, bn weight, bn bias, running_mean, running_var,
 *         num_batches_tracked (int64)] in state_dict order: conv1/bn1, then per block conv1/bn1,
 *         conv2/bn2, downsample.0/downsample.1 (layer2-4 block 0 only).
 * grads : 20 units x 3 pointers [d conv weight (OIHW f32), d bn weight, d bn bias] (overwritten).
 * ---------------------------------------------------------------------------------------------- */
typedef struct PhResnetPlan PhResnetPlan;
PhResnetPlan* ph_resnet_plan_create(int B, int H, int W, int prec);
void ph_resnet_plan_destroy(PhResnetPlan* plan);
size_t ph_resnet_workspace_bytes(const PhResnetPlan* plan);
size_t ph_resnet_packed_bytes(const PhResnetPlan* plan);
int ph_resnet_num_units(const PhResnetPlan* plan);
int ph_resnet_unit_shape(const PhResnetPlan* plan, int unit, int* out4 /* Cout, Cin, KS, stride */);
/* OIHW fp32 -> MFMA operand layouts (bf16 hi/lo planes, fwd [tap][O][I] and dgrad [tap][I][O]); call after
 * every optimiser step */
int ph_resnet_pack_weights(const PhResnetPlan* plan, const void* const* params, void* packed, ph_stream_t stream);
/* x_nchw [B,3,H,W] f32 -> f3 [B,256], f4 [B,512] f32 (either may be NULL).  flags bit0: train mode, update the running
 * statistics; bit1: eval mode (normalise with the running statistics; no backward); bit2: forward only - no
 * ph_resnet_backward will read this workspace (the no_grad EMA / teacher forwards of train_test_path_multi_distill.py:
 * 253-256): bn1 + ReLU of every BasicBlock (resnets.py:61-63) is then applied by conv2 while it stages its input and the
 * a1 tensor is not materialised (perf mode); bit3: keep the separate passes nevertheless (A/B and test switch); bit4 /
 * bit5: A/B and test switches (first-generation stride-2 kernel / separate stem conv and pooling passes); bit6: `x_nchw`
 * is the NHWC4 tensor ph_pack_input made of the image (the student and the teacher read the same x_path,
 * train_test_path_multi_distill.py:249,256: packed once) - it must stay valid until the matching backward has run */
/* Arithmetic of the backward's dgrad / wgrad launches where it differs from the plan's (PH_PREC_BF16X3 on a PH_PREC_BF16X6
 * plan: parity-mode forward, three-product backward; PH_PREC_FP16X1 on a PH_PREC_FP16X3 plan: three-product forward, the hi
 * planes' product alone in the backward; -1 = follow the plan) */
int ph_resnet_plan_set_backward_prec(const PhResnetPlan* plan, int prec);
/* Scheduling of ph_resnet_backward / _part: 1 (default) = the weight-gradient launches run on a second, process-wide
 * stream of the library (created once per device, never destroyed) beside the BatchNorm-backward / dgrad chain and are
 * joined before the call returns (inside a stream capture: a parallel branch of the graph); 0 = everything on the
 * caller's stream.  Same kernels; BatchNorm gradients bitwise the same,
 * weight gradients summed from half as many partial slabs (autograd of resnets.py:58-74 has no order between a layer's
 * weight and input gradients either). */
int ph_resnet_plan_set_backward_overlap(const PhResnetPlan* plan, int on);
int ph_pack_input(const float* x_nchw, void* x4 /* B*H*W*4 elements of the mode's activation type */, int B, int H, int W,
                  int prec, ph_stream_t stream);
int ph_resnet_forward(const PhResnetPlan* plan, const void* const* params, const void* packed, const float* x_nchw,
                      void* workspace, float* f3, float* f4, int flags, ph_stream_t stream);
int ph_resnet_backward(const PhResnetPlan* plan, const void* const* params, const void* packed, void* workspace,
                       const float* g_f3 /* may be NULL */, const float* g_f4, void* const* grads,
                       ph_stream_t stream);
/* The same backward in two calls for a data-parallel caller: part 0 = layers 4 and 3 (afterwards the gradients of
 * layers 3-4, 93 % of the trunk's parameter bytes, are final and their all-reduce can start), part 1 = layers 2, 1 and
 * the stem; part -1 = everything (== ph_resnet_backward).  The reference's DataParallel reduces after the whole
 * backward (utils.py:257-260). */
int ph_resnet_backward_part(const PhResnetPlan* plan, const void* const* params, const void* packed, void* workspace,
                            const float* g_f3, const float* g_f4, void* const* grads, int part, ph_stream_t stream);
/* Test access: the same backward cut off after `stop_after` >= 1 launch groups ("stages": the avgpool backward, then per
 * BasicBlock bn2 / wgrad(conv2) / dgrad(conv2) / bn1 / wgrad(conv1) / dgrad(conv1) [+ bn / wgrad / dgrad of the
 * downsample branch], then the stem's BatchNorm reduction, its apply pass, its weight gradient) - autograd of
 * resnets.py:58-74,219-222 one node at a time.  Only scratch buffers are written, so it can be re-run; with
 * the tensor-info call below - its `what` 4-9 are the scratch buffers, the BatchNorm statistics and the pool arg codes - a harness compares
 * every stage with a reference computed from that stage's own inputs (tests/test_gpu_fullsize.py). */
int ph_resnet_backward_debug(const PhResnetPlan* plan, const void* const* params, const void* packed, void* workspace,
                             const float* g_f3, const float* g_f4, void* const* grads, int stop_after, ph_stream_t stream);
/* Gradient with respect to the image [B,3,H,W] f32 of an EVAL-mode forward (flags bit1; BatchNorm backward is then
 * gamma * invstd * dz, no parameter gradients).  The reference needs it for the MIA-2023 stage-1 superpixel attention
 * masks ("MIA 2023/stage1_multi_modal_teacher/train_test_MT_SP_Masking.py":62-75: model.eval(); cost.backward();
 * x_path.grad).  ph_stem_dgrad is its last step (input gradient of the 7x7/2 conv; dy NHWC of the mode's type). */
int ph_resnet_backward_input(const PhResnetPlan* plan, const void* const* params, const void* packed, void* workspace,
                             const float* g_f3 /* may be NULL */, const float* g_f4, float* dx_nchw, ph_stream_t stream);
int ph_stem_dgrad(const void* dy_nhwc, const float* w_oihw, float* dx_nchw, int B, int H, int W, int prec,
                  ph_stream_t stream);
int ph_resnet_tensor_info(const PhResnetPlan* plan, int what, int id, size_t* byte_off, int* dims4);

/* ------------------------------------------------------------------------------------------------
 * Dense fp32 operators of the heads / SNN / fusion (resnets.py:165-169,239-250; networks_new.py:185-251;
 * fusion.py:36-63).  C[m][n] = act(sum_k A[m*sam+k*sak] * B[k*sbk+n*sbn] + bias[n]) (+C)
 * ---------------------------------------------------------------------------------------------- */
int ph_sgemm(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, long sam, long sak,
             long sbk, long sbn, long ldc, int act, int accumulate, ph_stream_t stream);
int ph_sgemm_splitk(const float* A, const float* B, const float* bias, float* C, float* part /* nsplit*M*N */,
                    int nsplit, int M, int N, int K, long sam, long sak, long sbk, long sbn, long ldc, int act,
                    ph_stream_t stream);
/* nn.BatchNorm1d training mode (+ReLU) and backward (resnets.py:165-167; fusion.py:29-32) */
int ph_bn1d_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* invstd,
                float* running_mean, float* running_var, int64_t* num_batches_tracked, int B, int C, float eps,
                float momentum, int relu, ph_stream_t stream);
/* eval mode (module.eval(): running statistics), used by the reference's test() (train_test_path_multi_distill.py:409-411) */
int ph_bn1d_eval(const float* x, const float* gamma, const float* beta, const float* running_mean,
                 const float* running_var, float* y, int B, int C, float eps, int relu, ph_stream_t stream);
/* backward of the eval-mode form (a fixed per-channel scale): needed when a gradient flows THROUGH an eval-mode net to its
 * input (train_test_MT_SP_Masking.py:62-75) */
int ph_bn1d_eval_bwd(const float* g, const float* y, const float* gamma, const float* running_var, float* dx, int B, int C,
                     float eps, int relu, ph_stream_t stream);
int ph_bn1d_bwd(const float* g, const float* y, const float* x, const float* mean, const float* invstd,
                const float* gamma, float* dx, float* dgamma, float* dbeta, int B, int C, int relu,
                ph_stream_t stream);
/* nn.LogSoftmax(dim=1) (networks_new.py:139-140), F.nll_loss mean (train_test_path_multi_distill.py:262) */
int ph_log_softmax(const float* x, float* y, int B, int C, ph_stream_t stream);
int ph_log_softmax_bwd(const float* g, const float* y, float* dx, int B, int C, ph_stream_t stream);
int ph_nll_fwd(const float* pred, const int64_t* grade, float* loss, int B, int C, float inv_bnorm,
               ph_stream_t stream);
int ph_nll_bwd(const float* gscalar, const int64_t* grade, float* dpred, int B, int C, float inv_bnorm,
               ph_stream_t stream);
/* DistillKL.forward / backward w.r.t. y_s (KD_loss.py:13-17) */
int ph_kl_fwd(const float* y_s, const float* y_t, float* loss, int B, int C, float T, float inv_bnorm,
              ph_stream_t stream);
int ph_kl_bwd(const float* gscalar, const float* y_s, const float* y_t, float* dy_s, int B, int C, float T,
              float inv_bnorm, ph_stream_t stream);
/* MIA-2023 per-sample DistillKL ("MIA 2023/stage2_unimodal_student/KD_loss.py":14-20) and its backward */
int ph_kl_rows_fwd(const float* y_s, const float* y_t, float* sample_loss, int B, int C, float T, ph_stream_t stream);
int ph_kl_rows_bwd(const float* g_rows, const float* y_s, const float* y_t, float* dy_s, int B, int C, float T,
                   ph_stream_t stream);
/* assign_sample_weights ("MIA 2023/stage2_unimodal_student/train_test_path_multi_distill.py":131-158), from logits */
int ph_conf_discrepancy(const float* logit_s, const float* logit_t, const int64_t* gt, float* out, int B, int C,
                        float max_discrep, ph_stream_t stream);
/* Normalize(2) of Embed (CL_utils/CRD_loss.py:263-267,276-279) */
int ph_l2norm_fwd(const float* x, float* y, float* norm, int B, int D, ph_stream_t stream);
int ph_l2norm_bwd(const float* g, const float* y, const float* norm, float* dx, int B, int D, ph_stream_t stream);
int ph_eltwise(const float* a, const float* b, float* out, size_t n, int op, ph_stream_t stream);
/* o1 (x) o2 with optional appended ones (fusion.py:56-58) / nn.Bilinear operand (fusion.py:43,50) */
int ph_outer(const float* o1, const float* o2, float* o12, int B, int D1, int D2, int append_one,
             ph_stream_t stream);
/* nn.Dropout / nn.AlphaDropout, training mode, counter-based RNG (fusion.py:22-32; networks_new.py:193) */
int ph_dropout(float* x, size_t n, float p, uint64_t seed, uint64_t offset, int alpha, ph_stream_t stream);
/* HIP-graph-replayable form: the per-step part of the counter is read from device memory */
int ph_dropout_dev(float* x, size_t n, float p, uint64_t seed, uint64_t site_offset, const uint64_t* step_counter,
                   int alpha, ph_stream_t stream);
int ph_counter_inc(uint64_t* counter, ph_stream_t stream);
/* Backward of the SNN / fusion operators (stage-1 teacher training, SURVEY row f-1: train_test_MT.py:42-337 needs the
 * gradients of networks_new.py:223-251 and fusion.py:36-63).  ph_dropout_bwd_dev re-creates the forward mask from the
 * same (seed, site_offset, step counter value); ph_gate_bwd: y = sigmoid(z) * h; ph_outer_bwd: the two operands of
 * ph_outer. */
int ph_dropout_bwd_dev(float* g, size_t n, float p, uint64_t seed, uint64_t site_offset, const uint64_t* step_counter,
                       int alpha, ph_stream_t stream);
/* the same two out of place (dst = f(src); src == dst allowed): one launch per dropout site instead of a copy + an in-place
 * launch on the latency-bound head chains of the step */
int ph_dropout_dev_to(const float* src, float* dst, size_t n, float p, uint64_t seed, uint64_t site_offset,
                      const uint64_t* step_counter, int alpha, ph_stream_t stream);
int ph_dropout_bwd_dev_to(const float* src, float* dst, size_t n, float p, uint64_t seed, uint64_t site_offset,
                          const uint64_t* step_counter, int alpha, ph_stream_t stream);
int ph_gate_bwd(const float* g, const float* z, const float* h, float* dz, float* dh, size_t n, ph_stream_t stream);
int ph_outer_bwd(const float* g, const float* o1, const float* o2, float* do1, float* do2, int B, int D1, int D2,
                 int append_one, ph_stream_t stream);
int ph_sum(const float* x, float* out, int n, float scale, ph_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * CRD memory bank (DC-Distill): ContrastMemory_v3.forward (CL_utils/memory_new.py:249-397) and
 * ContrastLoss_v2 (CL_utils/CRD_loss.py:221-244).  v1 = student embedding, v2 = teacher embedding,
 * mem1/mem2 = memory_v1/memory_v2 [n_data][128] f32, params = the module's `params` buffer
 * [K, T, Z_v1, Z_v2, momentum, P].
 * ---------------------------------------------------------------------------------------------- */
/* idx_bank2: optional second index array for memory_v2 (MIA-2023 v10: each bank has its own KNN positives); NULL = idx */
int ph_crd_score(const float* v1, const float* v2, const int64_t* idx /* [B][P+K] */, const int64_t* idx_bank2,
                 const float* mem1, const float* mem2, float* out1, float* out2, float* diff /* each [B][P+K] */,
                 int B, int PK, int feat_dim, float T, ph_stream_t stream);
/* ranks: the P2 host-RNG ranks of memory_new.py:311 (int32, device) or NULL for "hard" (:308) */
int ph_crd_select(const float* diff, const float* out1, const float* out2, const int* ranks, int* sel /* [B][P2+K2] */,
                  float* xs, float* xt /* gathered raw scores [B][P2+K2] */, int B, int P, int K, int P2, int K2,
                  int select_neg, int select_pos /* 0: keep every positive column in order (v3 / v10 banks) */,
                  ph_stream_t stream);
int ph_crd_zsum(const float* xs, const float* xt, float* sums2, int n, ph_stream_t stream);
int ph_crd_setz(float* params, const float* sums2, float count, float n_data, ph_stream_t stream);
/* loss partials lossp[B] (sum = s_loss + t_loss) and d loss/d v1, d loss/d v2 [B][128] */
/* posw_s / posw_t: optional per-positive weights [B][P2] (MIA-2023 ContrastLoss_v2: similarity / sum similarity,
 * "MIA 2023/stage2_unimodal_student/CL_utils/CRD_criterion_v10.py":281-314); NULL = 1/P2 */
int ph_crd_loss_grad(const float* xs, const float* xt, const int* sel, const int64_t* idx, const int64_t* idx_bank2,
                     const float* posw_s, const float* posw_t, const float* mem1, const float* mem2,
                     const float* params, float* lossp, float* dv1, float* dv2, int B, int PK, int P2, int K2,
                     int feat_dim, float n_data, float inv_bnorm, void* workspace /* may be NULL */, ph_stream_t stream);
size_t ph_crd_loss_grad_workspace_bytes(int B);
/* Bank-scan form of the CRD negatives: BASELINE configs[4] read as nce_k = 65536 negatives per query (SURVEY 8-e assumption (i);
 * the reference gathers them: "MIA 2023/stage2_unimodal_student/CL_utils/CRD_criterion_v10.py":68-70,106-107,140-141 index_select +
 * bmm over [B][K+1][128], and sums their terms in ContrastLoss_v2 :300-306).  With K at or above the number of bank rows the same
 * sums are taken over ALL rows weighted by multiplicity:
 *   ph_crd_neg_hist:  mult[b][r] = #{k : idx[b * row_stride + col0 + k] == r}, k < K  (mult [B][n_data] int32, every element written);
 *   scores S1 = v1 . bank2^T, S2 = v2 . bank1^T ([B][n_data], ph_sgemm);
 *   ph_crd_scan_neg with zsum_only = 1 (first call of a bank: zsums[0..1] += sum mult exp(S / T), the negatives' share of :146-153's means)
 *   ph_crd_scan_neg with zsum_only = 0: loss_neg[b] = -inv_bnorm sum_r mult (log(m Pn / (x1 + c)) + log(m Pn / (x2 + c))), x = exp(S / T) / Z,
 *     and S1, S2 overwritten by d loss / d S = inv_bnorm mult (x / (x + c)) / T, so that dv1 += S1 . bank2, dv2 += S2 . bank1 (ph_sgemm_splitk);
 *   ph_crd_loss_grad_pos: ph_crd_loss_grad over the P positive columns alone with the NCE constant m Pn of m_neg negatives.
 * params = the memory module's [K T Z_v1 Z_v2 momentum]; loss_neg [B]; zsums [2]. */
int ph_crd_neg_hist(const int64_t* idx, long row_stride, int col0, int K, int B, int n_data, int* mult, ph_stream_t stream);
size_t ph_crd_scan_neg_workspace_bytes(int B, int n_data);
int ph_crd_scan_neg(float* S1, float* S2, const int* mult, const float* params, void* workspace, float* loss_neg, float* zsums,
                    int B, int n_data, int m_neg, float inv_bnorm, int zsum_only, ph_stream_t stream);
int ph_crd_loss_grad_pos(const float* xs, const float* xt, const int* sel, const int64_t* idx, const int64_t* idx_bank2,
                         const float* posw_s, const float* posw_t, const float* mem1, const float* mem2, const float* params,
                         float* lossp, float* dv1, float* dv2, int B, int P, int m_neg, int feat_dim, float n_data, float inv_bnorm,
                         ph_stream_t stream);
/* MIA-2023 v10 KNN positives (CRD_criterion_v10.py:72-79,110-116): class-masked full-bank cosine top-num_pos of each
 * query's own bank row, for both banks; labels = class of every bank row (int32 [n_data]) */
size_t ph_crd_bank_topk_workspace_bytes(int B, int n_data);
int ph_crd_bank_topk(const float* mem1, const float* mem2, const int* labels, const int64_t* idx, int PK,
                     const int64_t* batch_label, int B, int n_data, int num_pos, int feat_dim, int64_t* nb1,
                     int64_t* nb2, float* sim1, float* sim2, void* workspace, ph_stream_t stream);
int ph_crd_update(float* mem1, float* mem2, const float* v1, const float* v2, const int64_t* y, const float* params,
                  int B, int feat_dim, ph_stream_t stream);
/* ContrastMemory_v3.forward as a STANDALONE call returning (out_v1, out_v2) [B][P2+K2] (memory_new.py:362-379: selected
 * scores / Z); also gathers the selected PRE-update bank rows (rows1 from memory_v1, rows2 from memory_v2, each
 * [B][P2+K2][128]) that its backward needs after the in-call momentum update (:382-395) has overwritten the bank. */
int ph_crd_outputs(const float* xs, const float* xt, const int* sel, const int64_t* idx, const int64_t* idx_bank2,
                   const float* mem1, const float* mem2, const float* params, float* out1, float* out2, float* rows1,
                   float* rows2, int B, int PK, int S2, int feat_dim, ph_stream_t stream);
/* backward of the above: dv1 = sum_j g1 out1 / T rows2, dv2 = sum_j g2 out2 / T rows1 (g1 / g2 may be NULL = zero) */
int ph_crd_outputs_bwd(const float* g1, const float* g2, const float* out1, const float* out2, const float* rows1,
                       const float* rows2, float T, float* dv1, float* dv2, int B, int S2, int feat_dim,
                       ph_stream_t stream);
/* ContrastLoss_v2.forward (CL_utils/CRD_loss.py:221-252) on x [B][S] with P positives first: rows[b] = the per-sample
 * loss of the sample_KD == "True" branch (:246-250); the "False" branch's scalar (:240-244) is sum(rows) / B.
 * dx[b][j] = d rows[b] / d x[b][j]. */
int ph_contrast_loss_v2(const float* x, float* rows, float* dx, int B, int S, int P, float n_data, ph_stream_t stream);
/* MIA-2023 v10 class-centre positives (`--pos_extra centers --nce_p 2`, CRD_criterion_v10.py:84-89,121-126: the mean
 * bank row of every class, recomputed per call).  mem_ext = a bank allocated with n_data + num_classes rows; row
 * n_data + c receives the mean of rows members[offsets[c] .. offsets[c+1]).  max_class_rows = the largest class. */
size_t ph_crd_class_centers_workspace_bytes(int num_classes, int max_class_rows);
int ph_crd_class_centers(float* mem_ext, const int* members, const int* offsets, int num_classes, int max_class_rows,
                         int n_data, int feat_dim, void* workspace, ph_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GK-Refine (AEKD_loss, train_test_path_multi_distill.py:41-70) and optimiser
 * (networks_new.py:85 Adam; train_test_path_multi_distill.py:34-38 update_ema_variables)
 * ---------------------------------------------------------------------------------------------- */
int ph_gram(const float* G /* [ng][n] */, float* gram /* [ng*ng] */, int ng, int n, ph_stream_t stream);
int ph_gk_scale(const float* gram, const float* const* losses /* device array of nl device scalars */, int ng, int nl,
                float mult, float* scale, float* total, ph_stream_t stream);
/* Fused forms used by the closed-form loss head of the stage-2 step (multimodal_learning_amd/loss_head.py; reference
 * train_test_path_multi_distill.py:262-313 and AEKD_loss :41-70).  ph_logit_losses: log-softmax, the two DistillKL terms
 * and the NLL of one logit matrix plus d(each loss)/d(logits) for a unit upstream gradient (losses[3], dl[3][B][C]).
 * ph_gk_finish: GK-Refine weights of five gradients in the order [div1, div2, CE, kd1, kd2] from their Gram matrix,
 * w = add + scale * coef, total = w . losses, scaled = losses * logc, scale_ext = scale as [div1, div2, kd1, kd2, CE]. */
int ph_logit_losses(const float* ys, const float* yt1, const float* yt2, const int64_t* grade, float* pred, float* losses,
                    float* dl, int B, int C, float T, float inv_bnorm, ph_stream_t stream);
int ph_gk_finish(const float* gram, const float* losses, const float* coef, const float* add, const float* logc, float mult,
                 float* scale_int, float* w, float* total, float* scaled, float* scale_ext, ph_stream_t stream);
/* momentum_AEKD_loss ("MIA 2022/train_test_path_multi_distill_v2.py":89-132): cosine Gram row sums without the
 * x len(list) factor, optional > thresh binarisation (:114-115), EMA of the weights (:121-124; *mo_init == 0 on the
 * first call, set to 1 by the kernel) */
int ph_gk_scale_momentum(const float* gram, int ng, int use_thresh, float thresh, float momentum, float* mo_scale,
                         int* mo_init, ph_stream_t stream);
/* The closed-form loss head's finish for that trainer (loss_head.py, variant mia2022): gram / losses in the head's internal order
 * [div1, div2, CE, kd1, kd2]; mo_scale (persistent, [5]) and scale_ext in the trainer's order [div1, div2, kd1, kd2, CE];
 * w = gradient weights (lam for CE, mult * state_i * c_i for the distillation terms, c = (alpha, alpha, beta e, beta e) with
 * e = *e_dev, the epoch weight of the CRD terms, :436-437), total = w . losses, scaled = losses * c (:452-455) */
int ph_gk_finish_momentum(const float* gram, const float* losses, float alpha, float beta, const float* e_dev, float lam,
                          float mult, int use_thresh, float thresh, float momentum, float* mo_scale, int* mo_init, float* w,
                          float* total, float* scaled, float* scale_ext, ph_stream_t stream);
/* GK_refine_thresh ("MIA 2023/stage2_unimodal_student/train_test_path_multi_distill.py":81-128): per-sample cosine
 * matrix of the ng gradients G[ng][B][128] -> all_scale[B][ng] */
int ph_gk_rows(const float* G, int ng, int B, int D, int use_thresh, float thresh, float* all_scale,
               ph_stream_t stream);
int ph_adam_ema_step(float* p, const float* g, float* m, float* v, float* ema /* may be NULL */, size_t n, double lr,
                     double beta1, double beta2, double eps, double weight_decay, int step, double ema_alpha,
                     ph_stream_t stream);
/* HIP-graph-replayable form: hyper = device float[5] {lr, 1-beta1^t, sqrt(1-beta2^t), ema_alpha, 1-ema_alpha}.  beta1 < 0: the
 * betas are read from device memory too, hyper = float[12] with [8..11] = {beta1, 1-beta1, beta2, 1-beta2} (a schedule that
 * cycles beta1 - lr_policy onecycle, networks_new.py:124-125 - reaches a launch replayed from a captured graph) */
int ph_adam_ema_step_dev(float* p, const float* g, float* m, float* v, float* ema, size_t n, double beta1,
                         double beta2, double eps, double weight_decay, const float* hyper, ph_stream_t stream);
/* torch.optim.Adagrad as define_optimizer builds it (reference MICCAI-2022/networks_new.py:86-87: lr, weight_decay,
 * initial_accumulator_value = 0.1, lr_decay 0, eps 1e-10): g' = g + wd p, sum += g'^2, p -= lr g' / (sqrt(sum) + eps), with the
 * mean-teacher EMA copy (train_test_path_multi_distill.py:34-38) fused as in ph_adam_ema_step_dev.  hyper: the same device
 * float[5] record ([0] = lr, [3] / [4] = EMA rate and complement). */
int ph_adagrad_ema_step_dev(float* p, const float* g, float* sum, float* ema /* may be NULL */, size_t n, double eps,
                            double weight_decay, const float* hyper, ph_stream_t stream);
int ph_ema_update(float* ema, const float* p, size_t n, float alpha, ph_stream_t stream);
/* define_reg (MICCAI-2022/networks_new.py:93-108 -> utils.py:60-198, the `lambda_reg * loss_reg` term of
 * train_test_MT.py:209-217 and train_test_path_multi_distill.py:312-313): L1 norm of a contiguous run of fp32
 * parameters.  ph_l1_sum: out[0] (+)= sum |w[i]|, `partials` = scratch of >= 1024 floats, fixed summation order.
 * ph_l1_sign_axpy: g[i] += coef * coef_dev[0] * sgn(w[i]) (coef_dev may be NULL = 1), its backward. */
int ph_l1_sum(const float* w, size_t n, float* partials, float* out, int accumulate, ph_stream_t stream);
int ph_l1_sign_axpy(const float* w, float* g, size_t n, const float* coef_dev, float coef, ph_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fine-grained convolution entry points (unit tests / other callers).  Activations NHWC in the precision
 * mode's type, weights OIHW f32.  `ws` must hold ph_conv2d_workspace_bytes().
 * ---------------------------------------------------------------------------------------------- */
size_t ph_conv2d_workspace_bytes(int B, int Cin, int IH, int IW, int Cout, int KS, int stride, int pad);
/* PH_PREC_FP16X3: the tensors a convolution READS (x of fwd / wgrad, dy of dgrad / wgrad) are half-pair tensors - layout
 * [..][C / 64][2][64] fp16 (per 64-channel slice a 128-B line of hi values, then one of lo values), 256-B aligned, C % 64 == 0
 * - and everything it writes is fp32.  These two convert (n elements, n % 64 == 0; `scale`: a power of two applied before the
 * split, 1 for activations).  Replace nothing in the reference: they are the storage format of this arithmetic. */
int ph_hp_pack(const float* src, void* dst, size_t n, float scale, ph_stream_t stream);
int ph_hp_unpack(const void* src, float* dst, size_t n, ph_stream_t stream);
int ph_conv2d_fwd(const void* x, const float* w_oihw, void* y, float* ch_sum /* [Cout] or NULL */,
                  float* ch_sumsq, int B, int Cin, int IH, int IW, int Cout, int KS, int stride, int pad, int prec,
                  void* ws, ph_stream_t stream);
int ph_conv2d_dgrad(const void* dy, const float* w_oihw, void* dx, int B, int Cin, int IH, int IW, int Cout, int KS,
                    int stride, int pad, int prec, void* ws, ph_stream_t stream);
/* dgrad with the residual term of a BasicBlock backward fused into its epilogue, as the trunk backward uses it (stride 1: the
 * identity shortcut; stride 2: the downsample path accumulates in place, res_g == dx, a parity class no tap reaches keeps dx)
 * (reference resnets.py:58-74 backward: d_x = dgrad(conv1) + d_out * (out > 0)):
 * dx = round(dgrad) + (res_a == NULL || res_a > 0 ? res_g : 0); res_g / res_a: [B][IH][IW][Cin] in the activation type. */
int ph_conv2d_dgrad_res(const void* dy, const float* w_oihw, void* dx, const void* res_g, const void* res_a, int B, int Cin,
                        int IH, int IW, int Cout, int KS, int stride, int pad, int prec, void* ws, ph_stream_t stream);
/* Test access: 3x3 stride-1 pad-1 perf-mode forward whose INPUT is the raw output of a convolution: relu(x * in_scale[c] +
 * in_shift[c]) (the BatchNorm + ReLU of reference resnets.py:58-66, rounded to bf16 as the stand-alone pass stores it) is applied to
 * every halo tile in LDS, as the forward-only networks of the distillation step run conv2 of layers 1-2. */
int ph_conv2d_fwd_fused_in(const void* x, const float* in_scale, const float* in_shift, const float* w_oihw, void* y,
                           float* ch_sum, float* ch_sumsq, int B, int Cin, int IH, int IW, int Cout, void* ws, ph_stream_t stream);
/* Test access: the same stride-1 3x3 perf-mode dgrad with the BatchNorm-backward sums of its OUTPUT taken in the epilogue
 * (the backward of `relu(bn(y))`: reference MICCAI-2022/resnets.py:58-74 through autograd - dgamma = sum dz xhat, dbeta = sum dz
 * with dz = dx * relu'), as ph_resnet_backward uses it for layer 1.  bst_y [B][IH][IW][Cin] = the BatchNorm's input y; mask =
 * (bst_a > 0) if bst_a else (bst_y * bst_scale + bst_shift > 0); sums [3][Cin] fp32 = sum dz | sum dz (bst_y - bst_mean) |
 * sum dz (bst_y2 - bst_mean2) (row 2 zero without bst_y2).  ws: ph_conv2d_workspace_bytes() + 3 * 4 * Cin * 1024 bytes. */
int ph_conv2d_dgrad_bnstat(const void* dy, const float* w_oihw, void* dx, const void* res_g, const void* res_a, const void* bst_y,
                           const void* bst_a, const void* bst_y2, const float* bst_scale, const float* bst_shift,
                           const float* bst_mean, const float* bst_mean2, float* sums, int B, int Cin, int IH, int IW, int Cout,
                           void* ws, ph_stream_t stream);
int ph_conv2d_wgrad(const void* x, const void* dy, float* dw_oihw, int B, int Cin, int IH, int IW, int Cout, int KS,
                    int stride, int pad, int prec, void* ws, ph_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * t-SVD low-rank constraint of the MIA-2022 stage-1 trainer ("MIA 2022/train_test_tSVD.py").
 * Adjacency = ph_sgemm (F F^T) + ph_l2norm_* (update_adj_tensor, :57-70).  Penalty mu/2 ||adj - aux||_F^2 (:418-431):
 *   out[0] = scale * sum (a - b)^2 ;  gradient out = gscalar[0] * alpha * (a - b).
 * ph_tsvd_update_aux replaces update_aux(adj, Lambda_global / mu) called at :382-391 (its source is absent from the
 * reference: this is the tensor-nuclear-norm proximal operator, see csrc/tsvd.hip): adj, aux are [V][B][B] (view-major),
 * V in {2,4,6,8}, B <= 128 (B <= 64: Jacobi on the embedding of X^H X; above: one-sided Jacobi on the slice); tnn[0] = (1/V) sum over frequency slices of the nuclear norm of the thresholded slice.
 * ---------------------------------------------------------------------------------------------- */
/* Relational distillation baselines of the distiller zoo (`--distill pkt|rkd`,
 * "MIA 2022/train_test_path_multi_distill_v2.py":339-342).  Each entry returns the loss AND its gradient with respect
 * to the student rows f_s [B, D] (the teacher rows f_t are constants), fixed summation order.
 * ph_pkt_loss_grad: "MIA 2022/distiller_zoo/PKT.py":17-46 (cosine-similarity probabilities, KL, eps 1e-7).
 * ph_rkd_loss_grad: "MIA 2022/distiller_zoo/RKD.py":15-58 (w_d * smooth-L1 of mean-normalised pairwise distances +
 * w_a * smooth-L1 of the B^3 angles); B <= 128, D <= 512. */
/* Cox negative partial log-likelihood of the survival task (MICCAI-2022/utils.py:361-376): loss and d loss / d theta;
 * dtheta may be NULL.  B <= 4096. */
int ph_cox_loss_grad(const float* theta, const float* survtime, const float* censor, float* loss, float* dtheta, int B,
                     ph_stream_t stream);
size_t ph_pkt_workspace_bytes(int B, int D);
int ph_pkt_loss_grad(const float* f_s, const float* f_t, float* loss, float* dx, int B, int D, void* workspace,
                     ph_stream_t stream);
size_t ph_rkd_workspace_bytes(int B, int D);
int ph_rkd_loss_grad(const float* f_s, const float* f_t, float* loss, float* dx, int B, int D, float w_d, float w_a,
                     void* workspace, ph_stream_t stream);

/* Superpixel attention masks of the MIA-2023 stage-1 trainer (SURVEY row f-4;
 * "MIA 2023/stage1_multi_modal_teacher/train_test_MT_SP_Masking.py":77-98), the part after the input gradients exist:
 * ph_superpixel_mask: grad_nchw [B,C,H,W] f32, sp_mask [B,H,W] labels in [0,N) -> mean gradient per superpixel
 * (sum over channels and pixels / (area + 1e-9); mean_out [B,N] may be NULL) and mask [B,H,W] = union of the K
 * superpixels with the largest mean (:88-93).  N <= 2048.  The reference does the aggregation on the HOST (:84-86).
 * ph_topk_threshold_mask: mask[b][i] = x[b][i] >= (K-th largest of row b)  (:96). */
int ph_superpixel_mask(const float* grad_nchw, const int64_t* sp_mask, float* mask, float* mean_out, int B, int C, int H,
                       int W, int N, int K, ph_stream_t stream);
int ph_topk_threshold_mask(const float* x, float* mask, int B, int D, int K, ph_stream_t stream);
/* out[b][c][p] = x[b][c][p] * (1 - mask[b][p])  (:201-202, the masked views; C = 1 for the omic vector) */
int ph_apply_mask(const float* x, const float* mask, float* out, int B, int C, size_t P, ph_stream_t stream);

/* Batch indices of DataLoader(shuffle=True, drop_last=True) (the training loader's sampler): rows q..q+B-1 of the keyed
 * permutation of epoch (*batch_no) / (n / B), q = ((*batch_no) % (n / B)) * B.  Distributional parity (host RNG stream). */
int ph_shuffle_indices(int64_t* out, int n, int B, uint64_t seed, const uint64_t* batch_no /* device pointer or NULL */,
                       ph_stream_t stream);

/* On-device input pipeline (reference MICCAI-2022/data_loaders_MT.py:168-175 TransformTwice(Compose([
 * RandomHorizontalFlip, RandomVerticalFlip, RandomCrop(S), ColorJitter(b, c, s, h), ToTensor, Normalize(0.5, 0.5)]))).
 * src: uint8 [n][SH][SW][3] tiles resident in HBM (the batch is rows[0..B) of it, or the first B tiles).  params: [B][2 views][16] f32 rows {flipH, flipV, top, left, brightness, contrast,
 * saturation, hue, order[4] (0 b, 1 c, 2 s, 3 h), grey mean (filled by ph_augment_apply), pad, 64-bit grey-sum accumulator (zero on entry)} - drawn on the device by
 * ph_augment_params - a counter RNG keyed by seed, *step, image and view - or supplied by the caller.  ph_augment_apply writes the
 * two views as f32 [B][3][S][S] in [-1, 1].  The colour arithmetic restates PIL / torchvision (absent here): parity
 * unpinned, see csrc/augment.hip and oracle/augment.py. */
int ph_augment_params(float* params, int B, uint64_t seed, const uint64_t* step /* device pointer or NULL */, int SH, int SW,
                      int S, float brightness, float contrast, float saturation, float hue, ph_stream_t stream);
int ph_augment_apply(const uint8_t* src, const int64_t* rows /* NULL - image b of the batch is src[b] - or src[rows[b]] */,
                     float* params, float* out0, float* out1, int B, int SH, int SW, int S, ph_stream_t stream);

/* On-device contrast-index sampler (reference MICCAI-2022/data_loaders_MT.py:229-249 and the neg_mode
 * variants of "MIA 2023/stage2_unimodal_student/data_loaders_MT.py":205-238).  out[b] = [positives | K negatives]:
 * pos_mode 0 'exact' (the query), 1 'relax' (one same-class row), 2 'multi_pos' (P distinct same-class rows, slot 0 :=
 * the query); neg_mode 0 'diff_class' (rows of the other classes), 1 'all_others' (every row but the query); with
 * replacement exactly when K exceeds the candidate list.  cls_pos / cls_neg: the per-class row lists concatenated,
 * *_off[c] .. *_off[c+1] delimiting class c.  `step` (device counter, may be NULL) and `seed` select the draw. */
int ph_contrast_sampler(const int64_t* index, const int64_t* grade, const int* cls_pos, const int* cls_pos_off,
                        const int* cls_neg, const int* cls_neg_off, int n_data, int B, int P, int K, int pos_mode,
                        int neg_mode, uint64_t seed, const uint64_t* step, int64_t* out, ph_stream_t stream);

/* ContrastMemory_v3.forward called WITHOUT contrast indices (reference MICCAI-2022/CL_utils/memory_new.py:265-267:
 * `idx = self.multinomial.draw(batchSize * (self.K + P)).view(batchSize, -1); idx.select(1, 0).copy_(y.data)`, AliasMethod
 * :401-458 over uniform unigrams): out [B][S] int64, column 0 = y[b], every other entry uniform in [0, n_data) with
 * replacement.  `seed` and the device counter `step` (may be NULL) select the draw; distributional parity. */
int ph_alias_uniform_draw(const int64_t* y, int64_t* out, int n_data, int B, int S, uint64_t seed, const uint64_t* step,
                          ph_stream_t stream);

/* Orthogonality loss of the stage-1 trainer (reference MICCAI-2022/CL_utils/orthogonal_loss.py:18-32): rows scaled by
 * a DETACHED 1/(||x||+eps) (:24-28); the D x D cross-correlation and its mean square are ph_sgemm + ph_sqdiff_sum. */
int ph_row_invnorm_scale(const float* x, float* y, float* inv /* [B] */, int B, int D, float eps, ph_stream_t stream);
int ph_row_scale(const float* x, const float* r /* [B] */, float* y, int B, int D, ph_stream_t stream);
int ph_sqdiff_sum(const float* a, const float* b, float* out, size_t n, float scale, ph_stream_t stream);
int ph_scaled_diff(const float* a, const float* b, const float* gscalar, float alpha, float* out, size_t n,
                   ph_stream_t stream);
/* the mixed-feature views of n_views = 6 / 8 (:305-307, :334-363): out = wa * a / max(a) + wb * b / max(b) over n elements */
int ph_maxnorm_mix(const float* a, const float* b, float* out, size_t n, float wa, float wb, ph_stream_t stream);
size_t ph_tsvd_workspace_bytes(int V, int B);
int ph_tsvd_update_aux(const float* adj, float* aux, float* tnn /* may be NULL */, int V, int B, float tau,
                       void* workspace, ph_stream_t stream);
/* the same with tau = tau_dev[0] read on the device (a HIP graph of the stage-1 step replays with the current mu) */
int ph_tsvd_update_aux_dev(const float* adj, float* aux, float* tnn /* may be NULL */, int V, int B, const float* tau_dev,
                           void* workspace, ph_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * In-library kernel timer for bench.py's `roofline` object: HIP events around every MFMA kernel launch on the
 * launch stream.  Classes: 0 first-generation tap-conv Cout=64 (dgrad parity classes), 1 first-generation tap-conv
 * Cout>=128 stride 1 (1x1, dgrad parity classes), 2 tap-conv stride 2, 3 wgrad, 4 stem forward, 5 stem wgrad,
 * 6 tapconv2 3x3 stride-1 Cout>=128 (fwd + dgrad), 7 tapconv2 3x3 stride-1 Cin=Cout=64 (layer 1).
 * out[cls*3 + {0,1,2}] = {launches, total ms, total algorithmic work}; 12 classes: 0-7 the MFMA kernels (work = FLOPs),
 * 8-11 the HBM-bound crd_score / crd_loss_grad / adam_ema / bn_apply kernels (work = algorithmic bytes).
 * ---------------------------------------------------------------------------------------------- */
int ph_prof_enable(int on);
int ph_prof_reset(void);
int ph_prof_summary(double* out, int nclasses);
/* out[cls*4 + {0,1,2,3}] = {launches, total ms, total algorithmic work, total algorithmic HBM bytes} */
int ph_prof_summary4(double* out, int nclasses);
/* Phase marker: a one-thread launch that stores the device's 100 MHz wall clock into *out (device memory) when `stream`
 * reaches it - capturable in a HIP graph, unlike events, so the phases of a replayed step can be timed. */
int ph_prof_stamp(unsigned long long* out, ph_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PATHOMIC_HIP_H_ */
