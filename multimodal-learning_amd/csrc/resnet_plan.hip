// Host-side runtime for the ResNet-18 trunk (reference resnets.py:126-236, BasicBlock :37-74):
// a static plan (layer table + workspace layout in one HBM arena, no allocation at run time) and the
// forward / backward launch sequences over the kernels in conv_*.hip and bn_act.hip.
//
// Data layout in HBM: activations NHWC of the precision mode's type T; every conv
// writes its RAW output y plus per-workgroup BN partial sums; BN is finalised per channel (fp64
// combine) into scale/shift; the consumer-side elementwise kernel applies BN(+residual)+ReLU.
// Saved for backward per BasicBlock: x_in, y1, a1, y2, (y_ds), out.
#include <atomic>
#include <cstdlib>
#include <iterator>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>
#include "ph_common.h"
#include "ph_kernels.h"
#include "ph_dense.h"

namespace {

struct Unit {
  int Cin, Cout, KS, S, pad;
  int IH, IW, OH, OW;
  size_t wf_off, wd_off;   // element offsets (bf16) of the fwd / dgrad hi planes in the packed buffer
  size_t wplane;           // elements per plane (lo plane follows hi)
  size_t y_off;            // byte offset of raw conv output in ws
  size_t st_off;           // byte offset of [mean, invstd, scale, shift] (4*C floats)
};
struct Block {
  int u1, u2, uds;
  size_t in_off, a1_off, out_off;   // byte offsets of the block input / post-bn1-relu / output activations
  // the same tensors as the ELEMENTWISE passes read them (shortcut term, ReLU masks, average pool): in PH_PREC_FP16X3 fp32
  // copies beside the half-pair MFMA operand images, otherwise the tensors themselves
  size_t in32_off, out32_off;
  int IH, IW, OH, OW, Cin, Cout;
};

constexpr size_t ALIGN = 256;
inline size_t up(size_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

}  // namespace

struct PhResnetPlan {
  int B, H, W, prec, es;
  std::vector<Unit> units;
  std::vector<Block> blocks;
  size_t x4_off, p0_off, p0raw_off, idx_off, parts_off, parts_bytes;
  size_t g0_off, g1_off, dy_off, da_off, slab_off, slab_bytes, bparts_off, cc_off, zero_off;
  size_t ws_bytes, packed_bytes;
  int PH0, PW0;   // pooled dims
  size_t act_max;   // max block-level activation bytes
  mutable int no_masked = 0;   // A/B and test switch, set by the last forward's flag bit3 and followed by its backward
  mutable int bwd_prec = -1;   // >= 0: arithmetic of the backward's dgrad / wgrad launches where it differs from `prec` (both split-plane)
  // pre-packed inputs (forward flag bit6) by the WORKSPACE of the forward that read them: the backward on that workspace
  // reads the same tensor again (stem wgrad).  Keyed by workspace, not cached per plan: a no_grad / eval forward or another
  // taped forward of a `_multi_forward` net on the same plan must not redirect an earlier forward's backward.
  struct X4Ent { const void* x4; unsigned long long seq; };
  mutable std::unordered_map<const void*, X4Ent> x4_by_ws;
  mutable unsigned long long x4_seq = 0;
  mutable std::mutex x4_mu;
  // Backward on two streams (backward_impl): the weight-gradient launches run on a side stream beside the BatchNorm-backward /
  // dgrad chain; `dy2_off` is the second dz buffer they need.
  size_t dy2_off = 0;
  mutable int bwd_overlap = 1;
  // PH_PREC_FP16X3: per-block max |dz| of a BatchNorm-backward reduction, and the {2^s, 2^-s} scale records of the two dz buffers
  size_t amax_off = 0, dzs_off = 0;
  // rows of a dgrad launch's fused BatchNorm-backward sums (PhTapConv::bst_y), [<= 1024][3][<= 512] floats, and the second
  // BatchNorm's (downsample branch) c1 / c2 when one launch reduced for two
  size_t fparts_off = 0, cc2_off = 0;
};

namespace {

// Workgroups a weight-gradient launch aims for.  Alone on the chip ~2 per CU is best (A/B on one box: 256 -> +0.2 ms / step,
// 384 and 768 -> +0.33).  On the side stream of the two-stream backward (backward_impl) ONE per CU: the BatchNorm-backward
// passes it runs beside need the other half of every SIMD's registers and wave slots (same box, ms per step: one stream
// 12.76 / two streams with 512: 12.63, 384: 12.47, 256: 12.31, 192: 12.2 on a box where 256 gave 11.95, 128: 13.3).
constexpr int WG_WANT_ALONE = 512, WG_WANT_BESIDE = 256;
int wgrad_chunks(const Unit& u, int B, int* tiles_per_chunk, int wg_want = WG_WANT_ALONE) {
  const int th = ph_wgrad_tile_h((u.KS == 1) ? 1 : u.S);
  const int ntiles = B * cdiv(u.OH, th) * cdiv(u.OW, 16);
  const int blocks = (u.Cout / 64) * (u.Cin / 64);
  int want = cdiv(wg_want, blocks);
  if (want > ntiles) want = ntiles;
  if (want < 1) want = 1;
  int tpc = cdiv(ntiles, want);
  *tiles_per_chunk = tpc;
  return cdiv(ntiles, tpc);
}
int stem_chunks(int B, int OH, int OW, int* tpc) {
  const int ntiles = B * cdiv(OH, 8) * cdiv(OW, 16);
  int want = ntiles < 1024 ? ntiles : 1024;   // 4 workgroups per CU: the per-tile staging latency is hidden by occupancy
  *tpc = cdiv(ntiles, want);
  return cdiv(ntiles, *tpc);
}

}  // namespace

extern "C" {

PhResnetPlan* ph_resnet_plan_create(int B, int H, int W, int prec) {
  if (B < 1 || H < 32 || W < 32 || (prec != PH_PREC_BF16 && !PH_IS_SPLIT_PREC(prec) && prec != PH_PREC_FP16X3)) return nullptr;
  if (prec == PH_PREC_FP16X3 && ((H | W) & 3)) return nullptr;   // (even stem output: the pooled form of the stem's BatchNorm-backward sums)
  PhResnetPlan* P = new (std::nothrow) PhResnetPlan();
  if (!P) return nullptr;
  P->B = B; P->H = H; P->W = W; P->prec = prec; P->es = prec == PH_PREC_BF16 ? 2 : 4;
  const size_t es = P->es;
  size_t off = 0, woff = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = up(off + bytes); return o; };
  P->x4_off = take((size_t)B * H * W * 4 * es);
  // stem
  Unit s{};
  s.Cin = 3; s.Cout = 64; s.KS = 7; s.S = 2; s.pad = 3; s.IH = H; s.IW = W;
  s.OH = (H + 6 - 7) / 2 + 1; s.OW = (W + 6 - 7) / 2 + 1;
  s.wplane = 7 * 64 * 32; s.wf_off = woff; woff += PH_NPLANES * s.wplane; s.wd_off = 0;
  s.y_off = take((size_t)B * s.OH * s.OW * 64 * es);
  s.st_off = take(4 * 64 * sizeof(float));
  P->units.push_back(s);
  P->PH0 = (s.OH + 1) / 2; P->PW0 = (s.OW + 1) / 2;
  P->p0_off = take((size_t)B * P->PH0 * P->PW0 * 64 * es);
  const bool hp = prec == PH_PREC_FP16X3;
  const size_t p032_off = hp ? take((size_t)B * P->PH0 * P->PW0 * 64 * es) : P->p0_off;
  P->idx_off = take((size_t)B * P->PH0 * P->PW0 * 64);
  P->p0raw_off = take((size_t)B * P->PH0 * P->PW0 * 64 * es);   // conv output at every pooling window's arg-max (training forwards)
  size_t parts_max = (size_t)std::max(ph_stem_stat_parts(B, s.OH, s.OW), ph_stem_pool_stat_parts(B, s.OH, s.OW)) * 2 * 64 * sizeof(float);
  size_t slab_max = 0;
  { int tpc; int nc = stem_chunks(B, s.OH, s.OW, &tpc); slab_max = (size_t)nc * 7 * 64 * 32 * sizeof(float); }
  size_t act_max = (size_t)B * P->PH0 * P->PW0 * 64 * es;
  int ih = P->PH0, iw = P->PW0, inpl = 64;
  size_t in_off = P->p0_off, in32_off = p032_off;
  const int planes[4] = {64, 128, 256, 512};
  for (int li = 0; li < 4; ++li)
    for (int bi = 0; bi < 2; ++bi) {
      const int stride = (li > 0 && bi == 0) ? 2 : 1;
      const int pl = planes[li];
      Block b{};
      b.IH = ih; b.IW = iw; b.Cin = inpl; b.Cout = pl;
      b.OH = (ih + 2 - 3) / stride + 1; b.OW = (iw + 2 - 3) / stride + 1;
      auto mk = [&](int cin, int cout, int ks, int st, int pad, int uih, int uiw) {
        Unit u{};
        u.Cin = cin; u.Cout = cout; u.KS = ks; u.S = st; u.pad = pad; u.IH = uih; u.IW = uiw;
        u.OH = (uih + 2 * pad - ks) / st + 1; u.OW = (uiw + 2 * pad - ks) / st + 1;
        u.wplane = (size_t)ks * ks * cin * cout;
        u.wf_off = woff; woff += PH_NPLANES * u.wplane;
        u.wd_off = woff; woff += PH_NPLANES * u.wplane;
        u.y_off = take((size_t)B * u.OH * u.OW * cout * es);
        u.st_off = take(4 * (size_t)cout * sizeof(float));
        PhTapConv tc{}; tc.B = B; tc.Cout = cout; tc.OHt = u.OH; tc.OWt = u.OW;
        parts_max = std::max(parts_max, (size_t)ph_tapconv_stat_parts(&tc, (ks == 1 ? 1 : st), prec) * 2 * cout * sizeof(float));
        int tpc; int nc = wgrad_chunks(u, B, &tpc);
        slab_max = std::max(slab_max, (size_t)nc * ks * ks * cin * cout * sizeof(float));
        P->units.push_back(u);
        return (int)P->units.size() - 1;
      };
      b.u1 = mk(inpl, pl, 3, stride, 1, ih, iw);
      b.u2 = mk(pl, pl, 3, 1, 1, b.OH, b.OW);
      b.uds = (stride != 1 || inpl != pl) ? mk(inpl, pl, 1, stride, 0, ih, iw) : -1;
      b.in_off = in_off; b.in32_off = in32_off;
      b.a1_off = take((size_t)B * b.OH * b.OW * pl * es);
      b.out_off = take((size_t)B * b.OH * b.OW * pl * es);
      b.out32_off = hp ? take((size_t)B * b.OH * b.OW * pl * es) : b.out_off;
      act_max = std::max(act_max, (size_t)B * b.OH * b.OW * pl * es);
      P->blocks.push_back(b);
      in_off = b.out_off; in32_off = b.out32_off; ih = b.OH; iw = b.OW; inpl = pl;
    }
  P->act_max = act_max;
  P->parts_bytes = parts_max;
  P->parts_off = take(parts_max);
  // backward scratch
  P->g0_off = take(act_max);
  P->g1_off = take(act_max);
  P->da_off = take(act_max);
  P->dy_off = take((size_t)B * s.OH * s.OW * 64 * es);
  P->dy2_off = take(act_max);      // second dz buffer of the two-stream backward (every unit but the stem fits)
  P->slab_bytes = slab_max;
  P->slab_off = take(slab_max);
  {  // BN-backward partial rows: ph_bn_bwd_parts() <= 1024 rows of [2][C <= 512]; the stem writes its own count of [2][64]
    const size_t stem_rows = (size_t)ph_stem_bwd_parts(B, s.OH);
    const size_t a = (size_t)1024 * 2 * 512, b = stem_rows * 2 * 64;
    P->bparts_off = take((a > b ? a : b) * sizeof(float));
  }
  P->cc_off = take(2 * 512 * sizeof(float));
  P->zero_off = take(256);
  {
    const size_t rows = std::max((size_t)1024, (size_t)ph_stem_bwd_parts(B, s.OH));
    P->amax_off = take(rows * sizeof(float));
    P->dzs_off = take(256);
  }
  P->fparts_off = take((size_t)1024 * 3 * 512 * sizeof(float));
  P->cc2_off = take(2 * 512 * sizeof(float));
  P->ws_bytes = off;
  P->packed_bytes = woff * sizeof(bf16);
  return P;
}

void ph_resnet_plan_destroy(PhResnetPlan* P) { delete P; }
size_t ph_resnet_workspace_bytes(const PhResnetPlan* P) { return P ? P->ws_bytes : 0; }
size_t ph_resnet_packed_bytes(const PhResnetPlan* P) { return P ? P->packed_bytes : 0; }
int ph_resnet_num_units(const PhResnetPlan* P) { return P ? (int)P->units.size() : 0; }
// shape of unit u's conv weight: out[0..3] = Cout, Cin, KS, has_downsample_role(0)
int ph_resnet_unit_shape(const PhResnetPlan* P, int u, int* out4) {
  if (!P || u < 0 || u >= (int)P->units.size()) return PH_EINVAL;
  out4[0] = P->units[u].Cout; out4[1] = P->units[u].Cin; out4[2] = P->units[u].KS; out4[3] = P->units[u].S;
  return PH_OK;
}

// params: per unit 6 pointers [w (OIHW f32), gamma, beta, running_mean, running_var, num_batches_tracked(i64)]
int ph_resnet_pack_weights(const PhResnetPlan* P, const void* const* params, void* packed, hipStream_t st) {
  if (!P || !params || !packed) return PH_EINVAL;
  bf16* pk = reinterpret_cast<bf16*>(packed);
  int rc = P->prec == PH_PREC_FP16X3
               ? ph_pack_w_stem_hp_launch(reinterpret_cast<const float*>(params[0]), pk + P->units[0].wf_off, st)
               : ph_pack_w_stem_launch(reinterpret_cast<const float*>(params[0]), pk + P->units[0].wf_off, st);
  if (rc) return rc;
  PhPackAll t{};
  size_t acc = 0;
  for (size_t i = 1; i < P->units.size(); ++i) {
    const Unit& u = P->units[i];
    const int k = t.n++;
    t.w[k] = reinterpret_cast<const float*>(params[i * 6 + 0]);
    t.dst_fwd[k] = u.wf_off; t.dst_dg[k] = u.wd_off;
    t.O[k] = u.Cout; t.I[k] = u.Cin; t.NT[k] = u.KS * u.KS;
    t.start[k] = acc;
    acc += u.wplane;
  }
  t.start[t.n] = acc;
  t.total = acc;
  return ph_pack_all_launch(&t, pk, P->prec == PH_PREC_BF16 ? 1 : (P->prec == PH_PREC_FP16X3 ? -3 : 3), st);
}

}  // extern "C"

namespace {

struct Ctx {
  const PhResnetPlan* P;
  const void* const* params;
  const bf16* pk;
  unsigned char* ws;
  hipStream_t st;
  int update_running;
  int eval;
  int no_masked = 0;      // A/B and test switch (forward flag bit3): first-generation kernels for the stride-2 convolutions
  int wg_want = WG_WANT_ALONE;   // workgroups a weight-gradient launch aims for (wgrad_chunks)
  float* stat(const Unit& u, int which) const {
    return reinterpret_cast<float*>(ws + u.st_off) + (size_t)which * u.Cout;
  }
  int bprec() const { return (P->bwd_prec >= 0 && P->prec != PH_PREC_BF16) ? P->bwd_prec : P->prec; }
  // PH_PREC_FP16X3: scale record {2^s, 2^-s} of dz buffer k (null in the other modes), and the amax scratch
  float* dzs(int k) const { return P->prec == PH_PREC_FP16X3 ? reinterpret_cast<float*>(ws + P->dzs_off) + 2 * k : nullptr; }
  float* amax() const { return P->prec == PH_PREC_FP16X3 ? reinterpret_cast<float*>(ws + P->amax_off) : nullptr; }
};

int conv_fwd(const Ctx& c, int ui, const void* in, const float* in_scale = nullptr, const float* in_shift = nullptr) {
  const PhResnetPlan* P = c.P;
  const Unit& u = P->units[ui];
  PhTapConv t{};
  t.in = in; t.w = c.pk + u.wf_off; t.wplane = u.wplane;
  t.in_scale = in_scale; t.in_shift = in_shift;
  t.out = c.ws + u.y_off; t.stats = c.eval ? nullptr : reinterpret_cast<float*>(c.ws + P->parts_off);
  t.B = P->B; t.IH = u.IH; t.IW = u.IW; t.Cin = u.Cin; t.Cout = u.Cout;
  t.OHt = u.OH; t.OWt = u.OW; t.OH = u.OH; t.OW = u.OW; t.os = 1; t.oa_h = 0; t.oa_w = 0;
  t.iy0 = -u.pad; t.ix0 = -u.pad; t.ntaps = u.KS * u.KS;
  for (int k = 0; k < t.ntaps; ++k) { t.dy[k] = k / u.KS; t.dx[k] = k % u.KS; t.wtap[k] = k; }
  int S = u.S;
  // 3x3 / stride 2 in perf mode: a stride-1 MASKED tap grid over the four pixel-parity planes of the input (conv_tap2.hip)
  // (round 6: conv_tap6b.hip takes the un-masked stride-2 descriptor itself)
  const bool tap6b = P->prec == PH_PREC_BF16 && u.KS == 3 && u.S == 2 && !c.no_masked && ph_tap6b_switch(-1) && ph_tapconv6b_eligible(&t);
  if (!tap6b && u.KS == 3 && u.S == 2 && u.pad == 1 && !c.no_masked && ph_tapconv2_setup_s2_fwd(&t, u.Cin, u.Cout, u.IH, u.IW, P->prec)) S = 1;
  if (u.KS == 1 && u.S == 2) {   // 1x1 / stride 2 == 1x1 / stride 1 over the even-pixel view of the input
    t.in_pix_stride = 2L * u.Cin; t.in_row_stride = 2L * u.IW * u.Cin; t.in_img_stride = (long)u.IH * u.IW * u.Cin;
    t.IH = u.OH; t.IW = u.OW; S = 1;
  }
  int rc = ph_tapconv_launch(&t, S, P->prec, c.st);
  if (rc || c.eval) return rc;      // (eval: scale/shift were preset from the running statistics)
  const int nparts = ph_tapconv_stat_parts(&t, S, P->prec);
  float* rm = c.update_running ? (float*)c.params[ui * 6 + 3] : nullptr;
  return ph_bn_finalize_launch(t.stats, nparts, u.Cout, (double)P->B * u.OH * u.OW, 1e-5f, 0.1f,
                               (const float*)c.params[ui * 6 + 1], (const float*)c.params[ui * 6 + 2], c.stat(u, 0),
                               c.stat(u, 1), c.stat(u, 2), c.stat(u, 3), rm, (float*)c.params[ui * 6 + 4],
                               (int64_t*)c.params[ui * 6 + 5], c.st);
}

// BatchNorm-backward sums a dgrad launch takes over the gradient it writes (PhTapConv::bst_y): of unit `u` (mask: its own ReLU
// when a == null, else a > 0) and optionally of unit `u2` (the downsample branch reading the same dz)
struct Bst { int u = -1, u2 = -1; const void* a = nullptr; };

// A/B and test switch: PH_BST=0 in the environment / ph_debug_set_bst(0) keeps every BatchNorm-backward reduction a pass of its own
int bst_switch(int set) {
  static int on = [] { const char* e = getenv("PH_BST"); return (e && e[0] == '0') ? 0 : 1; }();
  if (set >= 0) on = set ? 1 : 0;
  return on;
}

// dgrad of unit ui: in = dY [B][OH][OW][Cout] -> out = dX [B][IH][IW][Cin] (+ residual); dzs: dY's scale record (half-pair mode).
// bst (optional) + fused_parts: when the launch can take the sums, *fused_parts = the number of [3][C] rows it left in the plan's
// fparts buffer (0 = not fused: the caller runs the separate reduction)
int conv_dgrad(const Ctx& c, int ui, const void* dy, void* dx, const void* res_g, const void* res_a, const float* dzs = nullptr,
               const Bst* bst = nullptr, int* fused_parts = nullptr) {
  const PhResnetPlan* P = c.P;
  const Unit& u = P->units[ui];
  PhTapConv t{};
  t.in = dy; t.w = c.pk + u.wd_off; t.wplane = u.wplane; t.in_unscale = dzs;
  t.out = dx; t.stats = nullptr; t.res_g = res_g; t.res_a = res_a;
  t.B = P->B; t.IH = u.OH; t.IW = u.OW; t.Cin = u.Cout; t.Cout = u.Cin;
  t.OH = u.IH; t.OW = u.IW;
  if (fused_parts) *fused_parts = 0;
  if (u.S == 1) {
    t.OHt = u.IH; t.OWt = u.IW; t.os = 1; t.oa_h = 0; t.oa_w = 0;
    t.iy0 = -(u.KS - 1 - u.pad); t.ix0 = t.iy0; t.ntaps = u.KS * u.KS;
    for (int k = 0; k < t.ntaps; ++k) {
      const int dyy = k / u.KS, dxx = k % u.KS;
      t.dy[k] = dyy; t.dx[k] = dxx; t.wtap[k] = (u.KS - 1 - dyy) * u.KS + (u.KS - 1 - dxx);
    }
    if (bst && fused_parts && bst->u >= 0 && P->prec == PH_PREC_BF16 && bst_switch(-1)) {
      const Unit& bu = P->units[bst->u];
      PhTapConv f = t;
      f.bst_y = c.ws + bu.y_off; f.bst_mean = c.stat(bu, 0);
      f.bst_a = bst->a;
      if (!bst->a) { f.bst_scale = c.stat(bu, 2); f.bst_shift = c.stat(bu, 3); }
      if (bst->u2 >= 0) { f.bst_y2 = c.ws + P->units[bst->u2].y_off; f.bst_mean2 = c.stat(P->units[bst->u2], 0); }
      f.stats = reinterpret_cast<float*>(c.ws + P->fparts_off);
      // Where it pays (same-box A/B, profiles/EXPERIMENTS.md round 5): layer 1's bn1 (conv_tap4.hip, the mask re-derived from
      // y: one extra 8-byte load per piece; 114 us against 76 + a 65 us reduction pass).  Built, tested and OFF by default:
      // the mask-tensor form behind conv1's dgrad + residual (four epilogue operands per piece: PH_BST2=1) and the dense
      // kernel of layers 2-4 (conv_tap3.hip at 512 registers: 93 against 64 us per launch, more than the pass it replaces: PH_BST3=1)
      // Round 6: conv_tap7.hip has the registers conv_tap3.hip lacked (178 - 237 vector registers with the sums, no scratch), so layers
      // 2-4 CAN take the sums in the dgrad epilogue - measured same-box (profiles/EXPERIMENTS.md): B = 64: 10.16 ms without, 10.27 with
      // (the launches get slower by more than the 30 us pass they replace, which ran beside a weight gradient anyway); B = 256: trunk
      // 23.82 -> 23.60 ms, step 36.6 -> 36.3 ms.  Default: on from B = 128 where conv_tap7.hip takes the launch (PH_BST3 / PH_BST2 = 0 / 1
      // force it off / on, 1 also on conv_tap3.hip)
      static const int bst3 = [] { const char* e = getenv("PH_BST3"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
      static const int bst2 = [] { const char* e = getenv("PH_BST2"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
      const bool dense = ph_tap3_switch(-1) && ph_tapconv2_tile_h(&f, 1, P->prec) && ph_tapconv3_eligible(&f);
      const bool on7 = dense && ph_tap7_switch(-1) && ph_tapconv7_eligible(&f);
      const bool l1 = ph_tap4_switch(-1) && ph_tapconv4_eligible(&f);
      const bool big = P->B >= 128;
      const bool mask_ok = !bst->a || bst2 == 1 || (bst2 == -1 && on7 && big);      // the mask-tensor forms (bn2 / downsample behind conv1's dgrad)
      const bool ok = bu.Cout == t.Cout && bu.OH == t.OH && bu.OW == t.OW && mask_ok &&
                      (l1 || (dense && (bst3 == 1 || (bst3 == -1 && on7 && big))));
      if (ok) {
        const int rc = ph_tapconv_launch(&f, 1, c.bprec(), c.st);
        if (rc == PH_OK) *fused_parts = ph_tapconv2_stat_parts(&f);
        return rc;
      }
    }
    return ph_tapconv_launch(&t, 1, c.bprec(), c.st);
  }
  // stride 2: output parity classes (a,b); x row 2i+a receives kh with (a + pad - kh) even.  One launch per class, or - the
  // 3x3 convolutions (1 / 2 / 2 / 4 taps over the same dz, disjoint output pixels) - ONE launch for all four (PhTapConv::ncls,
  // round 6: as four launches in a row layers 3.0 / 4.0 put 256 / 128 workgroups on the 256 CUs each time; PH_S2_MERGE=0
  // keeps them apart, A/B and test switch)
  static const bool merge = [] { const char* e = getenv("PH_S2_MERGE"); return !(e && e[0] == '0'); }();
  PhTapConv mt = t;
  mt.ncls = 0;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      int nk = 0, khs[3], dhs[3], nw = 0, kws[3], dws[3];
      for (int kh = 0; kh < u.KS; ++kh)
        if (((a + u.pad - kh) & 1) == 0) { khs[nk] = kh; dhs[nk] = (a + u.pad - kh) / 2; ++nk; }
      for (int kw = 0; kw < u.KS; ++kw)
        if (((b + u.pad - kw) & 1) == 0) { kws[nw] = kw; dws[nw] = (b + u.pad - kw) / 2; ++nw; }
      t.OHt = (u.IH - a + 1) / 2; t.OWt = (u.IW - b + 1) / 2;
      if (t.OHt <= 0 || t.OWt <= 0) continue;
      t.os = 2; t.oa_h = a; t.oa_w = b; t.iy0 = 0; t.ix0 = 0;
      t.ntaps = nk * nw;
      if (t.ntaps == 0) {
        // no tap reaches this class (1x1 stride-2): gradient is zero there; with an in-place residual
        // (res_g == dx) the existing values already are the result, otherwise the caller pre-zeroed dx.
        continue;
      }
      int q = 0;
      for (int i = 0; i < nk; ++i)
        for (int j = 0; j < nw; ++j) {
          if (dhs[i] < 0 || dws[j] < 0 || dhs[i] > 2 || dws[j] > 2) return PH_EINVAL;
          t.dy[q] = dhs[i]; t.dx[q] = dws[j]; t.wtap[q] = khs[i] * u.KS + kws[j]; ++q;
        }
      if (merge && u.KS == 3 && t.ntaps <= 4 && mt.ncls < 4) {
        const int k = mt.ncls++;
        mt.c_ntaps[k] = t.ntaps; mt.c_oa_h[k] = a; mt.c_oa_w[k] = b; mt.c_OHt[k] = t.OHt; mt.c_OWt[k] = t.OWt;
        for (int e = 0; e < t.ntaps; ++e) { mt.c_dy[k][e] = t.dy[e]; mt.c_dx[k][e] = t.dx[e]; mt.c_wtap[k][e] = t.wtap[e]; }
        for (int e = t.ntaps; e < 4; ++e) { mt.c_dy[k][e] = 0; mt.c_dx[k][e] = 0; mt.c_wtap[k][e] = 0; }
        // the descriptor's own fields: the largest class (grid extent; profiler)
        if (k == 0 || t.OHt * t.OWt > mt.OHt * mt.OWt) { mt.OHt = t.OHt; mt.OWt = t.OWt; }
        if (k == 0 || t.ntaps > mt.ntaps) {
          mt.ntaps = t.ntaps;
          for (int e = 0; e < t.ntaps; ++e) { mt.dy[e] = t.dy[e]; mt.dx[e] = t.dx[e]; mt.wtap[e] = t.wtap[e]; }
        }
        mt.os = 2; mt.oa_h = 0; mt.oa_w = 0; mt.iy0 = 0; mt.ix0 = 0;
        continue;
      }
      int rc = ph_tapconv_launch(&t, 1, c.bprec(), c.st);
      if (rc) return rc;
    }
  if (mt.ncls >= 2) return ph_tapconv_launch(&mt, 1, c.bprec(), c.st);
  if (mt.ncls == 1) {      // (a single class that qualified: an ordinary launch)
    PhTapConv one = mt;
    one.ncls = 0;
    one.ntaps = mt.c_ntaps[0]; one.oa_h = mt.c_oa_h[0]; one.oa_w = mt.c_oa_w[0]; one.OHt = mt.c_OHt[0]; one.OWt = mt.c_OWt[0];
    for (int e = 0; e < one.ntaps; ++e) { one.dy[e] = mt.c_dy[0][e]; one.dx[e] = mt.c_dx[0][e]; one.wtap[e] = mt.c_wtap[0][e]; }
    return ph_tapconv_launch(&one, 1, c.bprec(), c.st);
  }
  return PH_OK;
}

int conv_wgrad(const Ctx& c, int ui, const void* x, const void* dy, float* dw, const float* dzs = nullptr) {
  const PhResnetPlan* P = c.P;
  const Unit& u = P->units[ui];
  PhWgrad w{};
  w.x = x; w.dy = dy; w.slab = reinterpret_cast<float*>(c.ws + P->slab_off);
  w.zeros = c.ws + P->zero_off;
  w.B = P->B; w.IH = u.IH; w.IW = u.IW; w.Cin = u.Cin; w.OH = u.OH; w.OW = u.OW; w.Cout = u.Cout;
  w.S = u.S; w.pad = u.pad; w.KS = u.KS;
  if (u.KS == 1 && u.S == 2) {   // strided view, as in conv_fwd
    w.x_pix_stride = 2L * u.Cin; w.x_row_stride = 2L * u.IW * u.Cin; w.x_img_stride = (long)u.IH * u.IW * u.Cin;
    w.IH = u.OH; w.IW = u.OW; w.S = 1;
  }
  w.nchunks = wgrad_chunks(u, P->B, &w.tiles_per_chunk, c.wg_want);
  int rc = ph_wgrad_launch(&w, c.bprec(), c.st);
  if (rc) return rc;
  return ph_wgrad_reduce_launch(w.slab, dw, w.nchunks, u.KS, u.Cout, u.Cin, dzs, c.st);
}

// BN backward of unit ui: dz = g * (a > 0) -> dgamma/dbeta, dy (into ws dy buffer).  self_mask: `a` is this unit's own
// relu(bn(y)) (bn1 of a block) - the mask is recomputed from y and `a` is not read.
// fused_parts > 0: the sums were taken by the dgrad launch that wrote `g` (conv_dgrad: [fused_parts][3][C] rows in the plan's
// fparts buffer, this unit's second sum in row `row2`) - no reduction pass.  cc2: the c1 / c2 pair lives in the plan's second
// buffer (a downsample BatchNorm finalised together with bn2, applied later).  apply = false: finalise only.
int bn_bwd(const Ctx& c, int ui, const void* g, const void* a, void* dy, float* dgamma, float* dbeta, bool self_mask = false,
           float* dzs = nullptr, int fused_parts = 0, int row2 = 1, bool cc2 = false, bool finalize = true, bool apply = true) {
  const PhResnetPlan* P = c.P;
  const Unit& u = P->units[ui];
  const size_t npix = (size_t)P->B * u.OH * u.OW;
  float* parts = reinterpret_cast<float*>(c.ws + P->bparts_off);
  float* c1 = reinterpret_cast<float*>(c.ws + (cc2 ? P->cc2_off : P->cc_off));
  float* c2 = c1 + 512;
  const void* y = c.ws + u.y_off;
  const float* ms = self_mask ? c.stat(u, 2) : nullptr;
  const float* mh = self_mask ? c.stat(u, 3) : nullptr;
  if (self_mask) a = nullptr;
  float* amax = dzs ? c.amax() : nullptr;
  int rc = PH_OK;
  if (finalize) {
    if (fused_parts > 0) {
      rc = ph_bn_bwd_finalize_fused_launch(reinterpret_cast<const float*>(c.ws + P->fparts_off), fused_parts, u.Cout, (double)npix,
                                           dgamma, dbeta, c1, c2, c.stat(u, 1), row2, c.st);
    } else {
      const int nparts = ph_bn_bwd_parts(npix, u.Cout);
      rc = ph_bn_bwd_reduce_launch(g, a, y, c.stat(u, 0), c.stat(u, 1), parts, npix, u.Cout, P->prec, ms, mh, amax, c.st);
      if (rc) return rc;
      rc = ph_bn_bwd_finalize_launch(parts, nparts, u.Cout, (double)npix, dgamma, dbeta, c1, c2, amax, nparts,
                                     (const float*)c.params[ui * 6 + 1], c.stat(u, 1), dzs, c.st);
    }
    if (rc) return rc;
  }
  if (!apply) return PH_OK;
  return ph_bn_bwd_apply_launch(g, a, y, c.stat(u, 0), c.stat(u, 1), (const float*)c.params[ui * 6 + 1], c1, c2, dy, npix,
                                u.Cout, P->prec, ms, mh, dzs, c.st);
}

}  // namespace

extern "C" {

// flags: bit0 = update BN running statistics (train mode); bit1 = eval mode (normalise with the running statistics);
// bit2 = forward only: no backward will read this workspace (the EMA and teacher networks of the distillation step) - in
// perf mode bn1 + ReLU of every block is then applied by conv2 itself while it stages its input (conv_tap2.hip,
// PhTapConv::in_scale): the a1 tensor is neither written nor read, 8 bn_apply launches per forward disappear;
// bit3 = A/B and test switch: separate bn_apply passes although bit2 is set;
// bit4 = A/B and test switch: the first-generation kernel for the 3x3 stride-2 convolutions (not the masked grid);
// bit5 = A/B and test switch: separate stem conv and BatchNorm + ReLU + max-pool passes although bit2 is set (perf mode
// otherwise pools the raw conv output inside the stem kernel, PhStemPool)
int ph_resnet_forward(const PhResnetPlan* P, const void* const* params, const void* packed, const float* x_nchw,
                      void* ws_, float* f3, float* f4, int flags, hipStream_t st) {
  if (!P || !params || !packed || !x_nchw || !ws_) return PH_EINVAL;
  Ctx c{P, params, reinterpret_cast<const bf16*>(packed), reinterpret_cast<unsigned char*>(ws_), st,
        (flags & 2) ? 0 : (flags & 1), (flags & 2) ? 1 : 0};
  P->no_masked = (flags & 16) ? 1 : 0;
  c.no_masked = P->no_masked;
  unsigned char* ws = c.ws;
  const bool hp = P->prec == PH_PREC_FP16X3;
  // half-pair mode, forward-only network (flag bit 2): nobody reads ReLU masks later, so the fp32 copies of the block outputs are
  // not written - the next block's shortcut term and the average pools read the half-pair operand images (22-23 significant
  // bits): 1.0 GB of writes per network and forward at B = 64 / 512 x 512.  PH_HP_OUT32=1 keeps the copies (A/B and test switch).
  static const bool keep32 = [] { const char* e = getenv("PH_HP_OUT32"); return e && e[0] == '1'; }();
  const bool fo_hp = hp && (flags & 4) && !keep32;
  // bit6: `x_nchw` is not the image but an NHWC4 tensor of the mode's activation type that ph_pack_input produced from it
  // (the student and the teacher of the distillation step read the same x_path: packed once, train_test_path_multi_distill.py:249,256)
  int rc = PH_OK;
  {
    std::lock_guard<std::mutex> lk(P->x4_mu);
    if ((flags & 64) && P->x4_by_ws.size() > 256) {
      // workspaces long gone (a backward erases its entry; these never had one): drop the OLDER half, never an entry a taped
      // forward of this step may still need (clear() also dropped live entries of `_multi_forward` nets)
      const unsigned long long cut = P->x4_seq - 128;
      for (auto it = P->x4_by_ws.begin(); it != P->x4_by_ws.end();) it = it->second.seq < cut ? P->x4_by_ws.erase(it) : std::next(it);
    }
    if (flags & 64) P->x4_by_ws[ws_] = PhResnetPlan::X4Ent{reinterpret_cast<const void*>(x_nchw), P->x4_seq++};
    else P->x4_by_ws.erase(ws_);
  }
  const unsigned char* x4p = (flags & 64) ? reinterpret_cast<const unsigned char*>(x_nchw) : ws + P->x4_off;
  if (!(flags & 64) && (rc = ph_pack_input_launch(x_nchw, ws + P->x4_off, P->B, P->H, P->W, P->prec, st))) return rc;
  if (c.eval) {
    PhBnEvalTable t{};
    t.n = (int)P->units.size();
    for (int i = 0; i < t.n; ++i) {
      const Unit& u = P->units[i];
      t.gamma[i] = (const float*)params[i * 6 + 1]; t.beta[i] = (const float*)params[i * 6 + 2];
      t.running_mean[i] = (const float*)params[i * 6 + 3]; t.running_var[i] = (const float*)params[i * 6 + 4];
      t.mean[i] = c.stat(u, 0); t.invstd[i] = c.stat(u, 1); t.scale[i] = c.stat(u, 2); t.shift[i] = c.stat(u, 3);
      t.C[i] = u.Cout;
    }
    if ((rc = ph_bn_eval_params_launch(&t, 1e-5f, st))) return rc;
  }
  bool p0_raw = false;
  if ((flags & 4) && !(flags & 32) && P->prec == PH_PREC_BF16) {
    // forward-only network, perf mode: the stem conv pools its own raw output (sign-aware max / min, conv_stem.hip
    // stem_fwd_pool_kernel) - the 537 MB conv output (B = 64, 512 x 512) is never written - and BatchNorm + ReLU is
    // applied to the pooled tensor in place; bitwise the separate passes' result (relu(scale * y + shift) is monotone)
    const Unit& u = P->units[0];
    PhStemPool s{};
    s.x4 = x4p; s.w = c.pk + u.wf_off; s.wplane = u.wplane;
    s.pooled = ws + P->p0_off; s.stats = c.eval ? nullptr : reinterpret_cast<float*>(ws + P->parts_off);
    s.gamma = (const float*)params[1];
    s.B = P->B; s.IH = P->H; s.IW = P->W; s.OH = u.OH; s.OW = u.OW; s.PH = P->PH0; s.PW = P->PW0;
    if ((rc = ph_stem_fwd_pool_launch(&s, st))) return rc;
    float* rm = c.update_running ? (float*)params[3] : nullptr;
    if (!c.eval && (rc = ph_bn_finalize_launch(s.stats, ph_stem_pool_stat_parts(P->B, u.OH, u.OW), 64, (double)P->B * u.OH * u.OW,
                                    1e-5f, 0.1f, (const float*)params[1], (const float*)params[2], c.stat(u, 0),
                                    c.stat(u, 1), c.stat(u, 2), c.stat(u, 3), rm, (float*)params[4],
                                    (int64_t*)params[5], st)))
      return rc;
    // BatchNorm + ReLU of the pooled tensor is applied by its two consumers - layer1.0.conv1 while it stages its input
    // (PhTapConv::in_scale, like conv2 of the forward-only blocks) and layer1.0's output pass on its shortcut term - so the
    // pooled tensor stays RAW and no pass over it remains (bit 3 of the flags keeps the separate pass, as for the blocks)
    p0_raw = !(flags & 8);
    if (!p0_raw && (rc = ph_bn_apply_launch(ws + P->p0_off, c.stat(u, 2), c.stat(u, 3), nullptr, nullptr, nullptr, nullptr,
                                            ws + P->p0_off, nullptr, (size_t)P->B * P->PH0 * P->PW0, 64, 1, P->prec, st)))
      return rc;
  } else
  {  // stem: conv7x7/2 -> BN stats -> fused BN+ReLU+maxpool
    const Unit& u = P->units[0];
    PhStem s{};
    s.x4 = x4p; s.w = c.pk + u.wf_off; s.wplane = u.wplane;
    s.out = ws + u.y_off; s.stats = c.eval ? nullptr : reinterpret_cast<float*>(ws + P->parts_off);
    s.B = P->B; s.IH = P->H; s.IW = P->W; s.OH = u.OH; s.OW = u.OW;
    if ((rc = ph_stem_fwd_launch(&s, P->prec, st))) return rc;
    float* rm = c.update_running ? (float*)params[3] : nullptr;
    if (!c.eval && (rc = ph_bn_finalize_launch(s.stats, ph_stem_stat_parts(P->B, u.OH, u.OW), 64, (double)P->B * u.OH * u.OW,
                                    1e-5f, 0.1f, (const float*)params[1], (const float*)params[2], c.stat(u, 0),
                                    c.stat(u, 1), c.stat(u, 2), c.stat(u, 3), rm, (float*)params[4],
                                    (int64_t*)params[5], st)))
      return rc;
    // (forward only: nobody scatters a gradient through the pooling windows - the argmax codes are not produced)
    if ((rc = ph_bn_relu_maxpool_launch(ws + u.y_off, c.stat(u, 2), c.stat(u, 3), ws + P->p0_off,
                                        (flags & 4) ? nullptr : ws + P->idx_off, (flags & 4) ? nullptr : ws + P->p0raw_off,
                                        (hp && !fo_hp) ? ws + P->blocks[0].in32_off : nullptr, P->B, u.OH, u.OW, 64, P->prec, st)))
      return rc;
  }
  // conv2 of every block is 3x3 / stride 1 with Cin = Cout in {64, 128, 256, 512}: always a second-generation kernel in perf mode
  const bool fuse_a1 = (flags & 4) && !(flags & 8) && P->prec == PH_PREC_BF16;
  for (size_t bi = 0; bi < P->blocks.size(); ++bi) {
    const Block& b = P->blocks[bi];
    const Unit& u1 = P->units[b.u1];
    const Unit& u2 = P->units[b.u2];
    const size_t npix = (size_t)P->B * b.OH * b.OW;
    const Unit& u0 = P->units[0];
    const bool raw_in = bi == 0 && p0_raw;      // block 0 reads the stem's pooled RAW output
    if ((rc = raw_in ? conv_fwd(c, b.u1, ws + b.in_off, c.stat(u0, 2), c.stat(u0, 3)) : conv_fwd(c, b.u1, ws + b.in_off))) return rc;
    // (measured per launch, B = 64, 512^2: the in-LDS pass costs the conv +10 us in layer 1 and +13 us in layers 2-4 - it
    // runs with the matrix pipe idle - against bn_apply launches of 49 / 25 / 13 / 8 us: fused where it pays)
    static const bool fuse_all = [] { const char* e = getenv("PH_FUSE_ALL"); return e && e[0] == '1'; }();      // A/B switch
    if (fuse_a1 && (b.Cout <= 128 || fuse_all)) {
      if ((rc = conv_fwd(c, b.u2, ws + u1.y_off, c.stat(u1, 2), c.stat(u1, 3)))) return rc;
    } else {
      if ((rc = ph_bn_apply_launch(ws + u1.y_off, c.stat(u1, 2), c.stat(u1, 3), nullptr, nullptr, nullptr, nullptr,
                                   ws + b.a1_off, nullptr, npix, b.Cout, 1, P->prec, st)))
        return rc;
      if ((rc = conv_fwd(c, b.u2, ws + b.a1_off))) return rc;
    }
    if (b.uds >= 0) {
      const Unit& ud = P->units[b.uds];
      if ((rc = conv_fwd(c, b.uds, ws + b.in_off))) return rc;
      rc = ph_bn_apply_launch(ws + u2.y_off, c.stat(u2, 2), c.stat(u2, 3), nullptr, ws + ud.y_off, c.stat(ud, 2),
                              c.stat(ud, 3), ws + b.out_off, (hp && !fo_hp) ? ws + b.out32_off : nullptr, npix, b.Cout, 1, P->prec, st);
    } else {
      rc = raw_in ? ph_bn_apply_launch(ws + u2.y_off, c.stat(u2, 2), c.stat(u2, 3), nullptr, ws + b.in_off, c.stat(u0, 2),
                                       c.stat(u0, 3), ws + b.out_off, nullptr, npix, b.Cout, 3, P->prec, st)
                  : ph_bn_apply_launch2(ws + u2.y_off, c.stat(u2, 2), c.stat(u2, 3), fo_hp ? ws + b.in_off : ws + b.in32_off, nullptr,
                                        nullptr, nullptr, ws + b.out_off, (hp && !fo_hp) ? ws + b.out32_off : nullptr, npix, b.Cout, 1,
                                        P->prec, fo_hp ? 1 : 0, st);
    }
    if (rc) return rc;
    if ((bi == 5 && f3) || (bi == 7 && f4)) {
      float* dst = bi == 5 ? f3 : f4;
      if ((rc = fo_hp ? ph_avgpool_launch_t(ws + b.out_off, dst, P->B, b.OH * b.OW, b.Cout, P->prec, st)
                      : ph_avgpool_launch(ws + b.out32_off, dst, P->B, b.OH * b.OW, b.Cout, P->prec, st)))
        return rc;
    }
  }
  return PH_OK;
}

}  // extern "C"

namespace {

// Every launch group of the backward is one numbered STAGE (1, 2, ...).  `stop` > 0 ends the call after that many stages:
// a test harness steps through the backward and compares each stage's output with a reference computed from that
// stage's own inputs (tests/test_gpu_fullsize.py); the product entry points pass 0.
#define PH_STAGE(call)                                   \
  do {                                                   \
    if ((rc = (call))) return rc;                        \
    if (stop > 0 && ++nst == stop) return PH_OK;         \
  } while (0)

// The backward runs on TWO streams: BatchNorm backward and dgrad form the dependent chain on the caller's stream; every
// weight gradient (wgrad + its slab reduction, 1.9 of the 5.6 ms) only needs its unit's dz and a saved activation and feeds
// nothing but the optimizer, so it goes to the library's side stream: its matrix-pipe work overlaps the HBM-bound BatchNorm
// passes and the tails of the chain's launches.  dz buffers alternate (dy / dy2) so that the chain can produce the next dz
// while the side stream still reads the last one; events order producer -> wgrad and wgrad -> the buffer's next writer.
// Both streams are joined before the call returns (also at the end of part 0: its gradients are then final).  Inside a
// stream capture the side stream joins the capture through the event waits (a parallel branch of the graph).  A weight
// gradient starts when the dgrad that reads the same dz is done (event after it): started earlier it takes CUs from the
// dgrad - a persistent kernel sized for the whole chip, one 512-register wave per SIMD - and the step gets SLOWER (same box:
// 12.4 one stream, 12.55-12.8 two streams); started there it runs beside the next unit's BatchNorm-backward passes.
// The chain's kernels and arguments are those of the one-stream sequence (BatchNorm gradients bitwise the same); the weight
// gradients are cut into half as many chunks (WG_WANT_BESIDE), i.e. the same sums in another fp32 order.
// The side stream and its events are PROCESS-WIDE (one set per device), created by the first backward that is not being
// captured and never destroyed: a stream that has taken part in a capture as a forked branch must outlive every later capture
// on the same origin stream (destroying the plan-owned stream of an earlier round-3 build produced rare segfaults inside
// hipStreamEndCapture of LATER captures), and streams of the torch pool are not destroyed either.  Every call reserves its own
// block of 64 events, so two backward passes in flight never share one.
struct SideRes {
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> evs;
  // A backward call holds one block of EV_BLOCK events until the work it enqueued has run: `done[b]` is recorded on the
  // caller's stream when the call returns and the block is handed out again only once that event has completed.  A call made
  // inside a stream capture frees its block at once (the capture turns its event waits into graph edges and keeps no
  // reference to the events).
  hipEvent_t done[16];
  bool busy[16];
  size_t cursor = 0;
  std::mutex mu;
};
constexpr size_t EV_BLOCK = 64, EV_BLOCKS = 16;
#define PH_EBUSY (-16)

SideRes* overlap_ready(const PhResnetPlan* P, hipStream_t st) {
  static SideRes* res[16] = {nullptr};
  static std::mutex mu;
  if (!P->bwd_overlap) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (res[dev]) return res[dev];
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;   // not while capturing
  SideRes* r = new (std::nothrow) SideRes();
  if (!r) return nullptr;
  if (hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking) != hipSuccess) { delete r; return nullptr; }
  r->evs.resize(EV_BLOCK * EV_BLOCKS + EV_BLOCKS);
  for (size_t i = 0; i < r->evs.size(); ++i)
    if (hipEventCreateWithFlags(&r->evs[i], hipEventDisableTiming) != hipSuccess) {
      for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(r->evs[j]);
      (void)hipStreamDestroy(r->side);      // (never used: safe to destroy)
      delete r;
      return nullptr;
    }
  for (size_t b = 0; b < EV_BLOCKS; ++b) { r->done[b] = r->evs[EV_BLOCK * EV_BLOCKS + b]; r->busy[b] = false; }
  res[dev] = r;
  return r;
}

int backward_impl(const PhResnetPlan* P, const void* const* params, const void* packed, void* ws_, const float* g_f3,
                  const float* g_f4, void* const* grads, int part, int stop, hipStream_t st) {
  if (!P || !params || !packed || !ws_ || !g_f4 || !grads || part < -1 || part > 1) return PH_EINVAL;
  const int bi_hi = part == 1 ? 3 : 7, bi_lo = part == 0 ? 4 : 0;
  Ctx c{P, params, reinterpret_cast<const bf16*>(packed), reinterpret_cast<unsigned char*>(ws_), st, 0};
  c.no_masked = P->no_masked;
  unsigned char* ws = c.ws;
  unsigned char* gcur = ws + P->g0_off;
  unsigned char* gnext = ws + P->g1_off;
  unsigned char* dab = ws + P->da_off;
  SideRes* const sr = stop == 0 ? overlap_ready(P, st) : nullptr;
  const bool ov = sr != nullptr;
  const hipStream_t side = ov ? sr->side : st;
  Ctx cs = c;                       // the weight-gradient launches' context
  // (PH_WG_BESIDE=n: A/B override of the side-stream weight gradients' workgroup target)
  static const int wg_beside = [] { const char* e = getenv("PH_WG_BESIDE"); const int v = e ? atoi(e) : 0; return v > 0 ? v : WG_WANT_BESIDE; }();
  if (ov) { cs.st = side; cs.wg_want = wg_beside; }
  // this call's block of events (see SideRes): eager calls take one of 15 blocks (waiting for the oldest call in flight when
  // all are held), captured calls the 16th
  size_t ev_base = 0;
  int ev_blk = -1;
  bool capturing = false;
  if (ov) {
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    capturing = hipStreamIsCapturing(st, &cst) == hipSuccess && cst != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lk(sr->mu);
    if (capturing) {
      // A captured call uses the block RESERVED for captures: the capture turns its event waits into graph edges and keeps no
      // reference to the events, so the block is free again when the call returns, and no hipEventQuery ever runs inside a
      // capture (in the default "global" capture mode a query from the capturing thread would invalidate it - ADVICE r04).
      ev_blk = (int)EV_BLOCKS - 1;
    } else {
      constexpr size_t NE = EV_BLOCKS - 1;      // blocks of eager calls
      for (size_t i = 0; i < NE && ev_blk < 0; ++i) {
        const size_t b = (sr->cursor + i) % NE;
        if (!sr->busy[b]) ev_blk = (int)b;
      }
      // none known free: ask the `done` events of the calls in flight
      for (size_t i = 0; i < NE && ev_blk < 0; ++i) {
        const size_t b = (sr->cursor + i) % NE;
        if (hipEventQuery(sr->done[b]) == hipSuccess) { sr->busy[b] = false; ev_blk = (int)b; }
      }
      (void)hipGetLastError();      // (hipErrorNotReady of a pending block is not an error of this call)
      if (ev_blk < 0) {
        // the host runs NE backward passes ahead of the GPU (an eager loop without a sync per step): back-pressure - wait for
        // the OLDEST call in flight (round-robin hand-out: the block at the cursor) instead of failing the step
        const size_t b = sr->cursor % NE;
        if (hipEventSynchronize(sr->done[b]) != hipSuccess) return PH_EBUSY;
        ev_blk = (int)b;
      }
      sr->busy[ev_blk] = true;      // (claimed; released below)
      sr->cursor = ((size_t)ev_blk + 1) % NE;
    }
    ev_base = (size_t)ev_blk * EV_BLOCK;
  }
  struct BlockGuard {      // every exit path: a captured call frees the block, an eager one marks it with `done` on the stream
    SideRes* r; int b; bool cap; hipStream_t s;
    ~BlockGuard() {
      if (!r || b < 0) return;
      std::lock_guard<std::mutex> lk(r->mu);
      if (cap) return;      // (the capture block is never marked busy)
      if (hipEventRecord(r->done[b], s) != hipSuccess) r->busy[b] = false;
    }
  } guard{ov ? sr : nullptr, ev_blk, capturing, st};
  size_t ev_used = 0;
  unsigned char* dzb[2] = {ws + P->dy_off, ov ? ws + P->dy2_off : ws + P->dy_off};
  hipEvent_t rd[2] = {nullptr, nullptr};    // recorded on the side stream after the last reader of dzb[k]
  int k = 0;
  auto next_ev = [&]() { return sr->evs[ev_base + (ev_used++ % EV_BLOCK)]; };      // (a call needs < 50)
  // the chain is about to overwrite dzb[k]: wait for the weight gradient that read it
  auto claim = [&](int kk) -> int {
    if (ov && rd[kk]) { if (hipStreamWaitEvent(st, rd[kk], 0) != hipSuccess) return PH_ELAUNCH; rd[kk] = nullptr; }
    return PH_OK;
  };
  // dz of unit ui is in dzb[kk] (produced on st): its weight gradient
  auto wgrad_of = [&](int ui, const void* x, int kk) -> int {
    if (!ov) return conv_wgrad(c, ui, x, dzb[kk], (float*)grads[ui * 3 + 0], c.dzs(kk));
    hipEvent_t e = next_ev();
    if (hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(side, e, 0) != hipSuccess) return PH_ELAUNCH;
    int r = conv_wgrad(cs, ui, x, dzb[kk], (float*)grads[ui * 3 + 0], c.dzs(kk));
    if (r) return r;
    hipEvent_t d = next_ev();
    if (hipEventRecord(d, side) != hipSuccess) return PH_ELAUNCH;
    rd[kk] = d;
    return PH_OK;
  };
  auto join = [&]() -> int {
    if (!ov) return PH_OK;
    hipEvent_t e = next_ev();
    if (hipEventRecord(e, side) != hipSuccess || hipStreamWaitEvent(st, e, 0) != hipSuccess) return PH_ELAUNCH;
    return PH_OK;
  };
  int rc, nst = 0;
  int fused_out = 0;      // rows of fused bn2 (+ downsample) sums the last dgrad left for the block about to be processed
  if (part != 1) {
    if (hipMemsetAsync(ws + P->zero_off, 0, 256, st) != hipSuccess) return PH_ELAUNCH;
    const Block& b = P->blocks[7];
    PH_STAGE(ph_avgpool_bwd_launch(g_f4, gcur, P->B, b.OH * b.OW, b.Cout, 0, P->prec, st));
  }
  for (int bi = bi_hi; bi >= bi_lo; --bi) {
    const Block& b = P->blocks[bi];
    if (bi == 5 && g_f3) {
      PH_STAGE(ph_avgpool_bwd_launch(g_f3, gcur, P->B, b.OH * b.OW, b.Cout, 1, P->prec, st));
      fused_out = 0;      // (sums taken before this addition would be stale: the separate reduction runs)
    }
    const void* out = ws + b.out32_off;      // (ReLU masks: the tensor as the elementwise passes read it)
    const void* a1 = ws + b.a1_off;
    const void* xin = ws + b.in_off;
    const void* xin32 = ws + b.in32_off;     // (the block input as the elementwise passes read it: the previous block's ReLU mask)
    // bn2 <- d_out * (out > 0); fused_out: the dgrad that wrote gcur (the next block's conv1 + residual) took bn2's sums - and
    // the downsample BatchNorm's, which reduces the same dz: both are finalised here, the downsample pair into the second buffer
    if ((rc = claim(k))) return rc;
    if (fused_out > 0 && b.uds >= 0 &&
        (rc = bn_bwd(c, b.uds, gcur, out, nullptr, (float*)grads[b.uds * 3 + 1], (float*)grads[b.uds * 3 + 2], false, nullptr,
                     fused_out, 2, true, true, false)))
      return rc;
    const bool ds_done = fused_out > 0 && b.uds >= 0;
    PH_STAGE(bn_bwd(c, b.u2, gcur, out, dzb[k], (float*)grads[b.u2 * 3 + 1], (float*)grads[b.u2 * 3 + 2], false, c.dzs(k), fused_out));
    fused_out = 0;
    if (!ov) PH_STAGE(wgrad_of(b.u2, a1, k));
    int fused_a1 = 0;      // conv2's dgrad writes d_a1 and takes bn1's sums over it (mask: bn1's own ReLU, re-derived from y1)
    {
      Bst bs; bs.u = b.u1;
      PH_STAGE(conv_dgrad(c, b.u2, dzb[k], dab, nullptr, nullptr, c.dzs(k), &bs, &fused_a1));
    }
    if (ov) PH_STAGE(wgrad_of(b.u2, a1, k));
    k ^= 1;
    // bn1 <- d_a1 * (a1 > 0)
    if ((rc = claim(k))) return rc;
    PH_STAGE(bn_bwd(c, b.u1, dab, a1, dzb[k], (float*)grads[b.u1 * 3 + 1], (float*)grads[b.u1 * 3 + 2], true, c.dzs(k), fused_a1));
    if (!ov) PH_STAGE(wgrad_of(b.u1, xin, k));
    if (b.uds < 0) {
      // identity shortcut: d_xin = dgrad(conv1) + d_out * (out > 0), fused in the dgrad epilogue - which also takes the sums of
      // the PREVIOUS block's bn2 (and downsample BatchNorm) over d_xin * (xin > 0).  Not where f3's gradient is added to
      // d_xin afterwards (block 5 with g_f3: its producer is a downsample block anyway).
      Bst bs;
      if (bi > bi_lo) {
        const Block& pb = P->blocks[bi - 1];
        bs.u = pb.u2; bs.u2 = pb.uds; bs.a = xin32;
      }
        PH_STAGE(conv_dgrad(c, b.u1, dzb[k], gnext, gcur, out, c.dzs(k), &bs, &fused_out));
      if (ov) PH_STAGE(wgrad_of(b.u1, xin, k));
      k ^= 1;
    } else {
        PH_STAGE(conv_dgrad(c, b.u1, dzb[k], gnext, nullptr, nullptr, c.dzs(k)));
      if (ov) PH_STAGE(wgrad_of(b.u1, xin, k));
      k ^= 1;
      if ((rc = claim(k))) return rc;
      PH_STAGE(bn_bwd(c, b.uds, gcur, out, dzb[k], (float*)grads[b.uds * 3 + 1], (float*)grads[b.uds * 3 + 2], false, c.dzs(k),
                      0, 1, ds_done, !ds_done));
      if (!ov) PH_STAGE(wgrad_of(b.uds, xin, k));
        PH_STAGE(conv_dgrad(c, b.uds, dzb[k], gnext, gnext, nullptr, c.dzs(k)));   // in-place accumulate
      if (ov) PH_STAGE(wgrad_of(b.uds, xin, k));
      k ^= 1;
    }
    unsigned char* t = gcur; gcur = gnext; gnext = t;
  }
  if (part != 0) {  // stem: d_pool -> (maxpool, relu, bn) backward -> wgrad.  No dgrad: the image needs no gradient.
    const Unit& u = P->units[0];
    const size_t npix = (size_t)P->B * u.OH * u.OW;
    float* parts = reinterpret_cast<float*>(ws + P->bparts_off);
    float* c1 = reinterpret_cast<float*>(ws + P->cc_off);
    float* c2 = c1 + 512;
    if ((rc = ph_stem_bwd_reduce_launch(gcur, ws + P->idx_off, ws + u.y_off, ws + P->p0raw_off, c.stat(u, 0), c.stat(u, 1),
                                        c.stat(u, 2), c.stat(u, 3), parts, P->B, u.OH, u.OW, 64, P->prec, c.amax(), st)))
      return rc;
    if (c.dzs(0) && (rc = claim(0))) return rc;      // (half-pair mode: the finalize pass rewrites buffer 0's scale record)
    PH_STAGE(ph_bn_bwd_finalize_launch(parts, ph_stem_bwd_parts(P->B, u.OH), 64, (double)npix, (float*)grads[1],
                                       (float*)grads[2], c1, c2, c.amax(), ph_stem_bwd_parts(P->B, u.OH),
                                       (const float*)params[1], c.stat(u, 1), c.dzs(0), st));
    if ((rc = claim(0))) return rc;       // the stem's dz is full resolution: only the first buffer holds it
    PH_STAGE(ph_stem_bwd_apply_launch(gcur, ws + P->idx_off, ws + u.y_off, c.stat(u, 0), c.stat(u, 1), c.stat(u, 2),
                                      c.stat(u, 3), (const float*)params[1], c1, c2, dzb[0], P->B, u.OH, u.OW, 64,
                                      P->prec, c.dzs(0), st));
    PhStemWgrad w{};
    const void* x4_ext = nullptr;
    {
      std::lock_guard<std::mutex> lk(P->x4_mu);
      auto it = P->x4_by_ws.find(ws_);
      if (it != P->x4_by_ws.end()) {
        x4_ext = it->second.x4;
        if (stop == 0) P->x4_by_ws.erase(it);      // consumed (the debug harness re-runs the backward on one forward: kept there)
      }
    }
    w.x4 = x4_ext ? x4_ext : ws + P->x4_off; w.dy = dzb[0]; w.slab = reinterpret_cast<float*>(ws + P->slab_off);
    w.B = P->B; w.IH = P->H; w.IW = P->W; w.OH = u.OH; w.OW = u.OW;
    w.nchunks = stem_chunks(P->B, u.OH, u.OW, &w.tiles_per_chunk);
    if (ov) {      // same stream as the other weight gradients: they share the slab
      hipEvent_t e = next_ev();
      if (hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(side, e, 0) != hipSuccess) return PH_ELAUNCH;
    }
    if ((rc = ph_stem_wgrad_launch(&w, c.bprec(), cs.st))) return rc;
    PH_STAGE(ph_stem_wgrad_reduce_launch(w.slab, (float*)grads[0], w.nchunks, c.dzs(0), cs.st));
  }
  return join();
}
#undef PH_STAGE

}  // namespace

extern "C" {

// Arithmetic of the BACKWARD's matrix kernels (dgrad / wgrad) where it differs from the plan's: PH_PREC_BF16X3 on a
// PH_PREC_BF16X6 plan = six-product forward (logits, losses and GK-Refine weights at parity-mode accuracy), three-product
// backward (gradients at ~1e-3 relative); -1 = follow the plan.  Both arithmetics read the same fp32 activations and the
// same packed weight planes.
int ph_resnet_plan_set_backward_prec(const PhResnetPlan* P, int prec) {
  // (a half-pair plan admits PH_PREC_FP16X1 - the hi planes' product alone in dgrad / wgrad - a split-plane plan the other
  // split-plane arithmetic)
  if (!P || (prec != -1 && P->prec == PH_PREC_BF16)) return PH_EINVAL;
  if (prec != -1 && !(P->prec == PH_PREC_FP16X3 ? (prec == PH_PREC_FP16X1 || prec == PH_PREC_FP16X3) : PH_IS_SPLIT_PREC(prec))) return PH_EINVAL;
  P->bwd_prec = prec;
  return PH_OK;
}

// A/B and test switch: 0 = the whole backward on the caller's stream (the round-2 sequence), 1 (default) = weight
// gradients on the side stream
// A/B and test switch (not part of the public C-ABI): 0 = every BatchNorm-backward reduction as a pass of its own
int ph_debug_set_bst(int on) { return bst_switch(on ? 1 : 0); }

int ph_resnet_plan_set_backward_overlap(const PhResnetPlan* P, int on) {
  if (!P) return PH_EINVAL;
  P->bwd_overlap = on ? 1 : 0;
  return PH_OK;
}

// image [B,3,H,W] f32 -> NHWC4 (channel 3 = 0) of the mode's activation type, B * H * W * 4 elements: the trunk's input
// layout, for ph_resnet_forward flag bit6
int ph_pack_input(const float* x_nchw, void* x4, int B, int H, int W, int prec, hipStream_t st) {
  if (!x_nchw || !x4 || B < 1 || H < 1 || W < 1 || (prec != PH_PREC_BF16 && !PH_IS_SPLIT_PREC(prec) && prec != PH_PREC_FP16X3))
    return PH_EINVAL;
  return ph_pack_input_launch(x_nchw, x4, B, H, W, prec, st);
}

// grads: per unit 3 pointers [dw (OIHW f32), dgamma, dbeta]; g_f3 may be null.  Needs the activations the
// matching ph_resnet_forward left in `ws`.
int ph_resnet_backward(const PhResnetPlan* P, const void* const* params, const void* packed, void* ws_,
                       const float* g_f3, const float* g_f4, void* const* grads, hipStream_t st) {
  return backward_impl(P, params, packed, ws_, g_f3, g_f4, grads, -1, 0, st);
}

// part -1: everything; part 0: layers 4 and 3 (blocks 7..4); part 1: layers 2 and 1 (blocks 3..0) and the stem.  After
// part 0 every gradient of layers 3-4 (93 % of the trunk's parameter bytes) is final: a data-parallel caller starts
// their all-reduce there and overlaps it with part 1.  An even number of blocks per part keeps the ping-pong buffers
// of the block gradient where the next part expects them.
int ph_resnet_backward_part(const PhResnetPlan* P, const void* const* params, const void* packed, void* ws_,
                            const float* g_f3, const float* g_f4, void* const* grads, int part, hipStream_t st) {
  return backward_impl(P, params, packed, ws_, g_f3, g_f4, grads, part, 0, st);
}

// test access: the whole backward, cut off after `stop_after` stages (>= 1; see PH_STAGE above for the numbering: the
// avgpool backward, then per block bn2 / wgrad2 / dgrad2 / bn1 / wgrad1 / dgrad1 (+ bn_ds / wgrad_ds / dgrad_ds), then the
// stem's BatchNorm reduction, its apply pass and its weight gradient).  Re-runnable: it only writes scratch buffers.
int ph_resnet_backward_debug(const PhResnetPlan* P, const void* const* params, const void* packed, void* ws_,
                             const float* g_f3, const float* g_f4, void* const* grads, int stop_after, hipStream_t st) {
  if (stop_after < 1) return PH_EINVAL;
  return backward_impl(P, params, packed, ws_, g_f3, g_f4, grads, -1, stop_after, st);
}

// Gradient with respect to the IMAGE of an eval-mode forward (flags bit1): BatchNorm is a fixed per-channel scale, so
// its backward is gamma * invstd * dz - the training kernel with zero correction terms - and no parameter gradient is
// produced.  Used by the MIA-2023 stage-1 superpixel attention masks
// ("MIA 2023/stage1_multi_modal_teacher/train_test_MT_SP_Masking.py":62-75: model.eval(), cost.backward(), x_path.grad).
int ph_resnet_backward_input(const PhResnetPlan* P, const void* const* params, const void* packed, void* ws_,
                             const float* g_f3, const float* g_f4, float* dx_nchw, hipStream_t st) {
  if (!P || !params || !packed || !ws_ || !g_f4 || !dx_nchw) return PH_EINVAL;
  if (P->prec == PH_PREC_FP16X3) return PH_EINVAL;      // (the image gradient is built for the bf16 / split-plane arithmetics)
  Ctx c{P, params, reinterpret_cast<const bf16*>(packed), reinterpret_cast<unsigned char*>(ws_), st, 0, 1};
  c.no_masked = P->no_masked;
  unsigned char* ws = c.ws;
  unsigned char* gcur = ws + P->g0_off;
  unsigned char* gnext = ws + P->g1_off;
  unsigned char* dyb = ws + P->dy_off;
  unsigned char* dab = ws + P->da_off;
  float* c1 = reinterpret_cast<float*>(ws + P->cc_off);
  float* c2 = c1 + 512;
  int rc;
  if (hipMemsetAsync(ws + P->zero_off, 0, 256, st) != hipSuccess) return PH_ELAUNCH;
  if (hipMemsetAsync(c1, 0, 1024 * sizeof(float), st) != hipSuccess) return PH_ELAUNCH;
  auto bn_eval_bwd = [&](int ui, const void* g, const void* a, void* dy) -> int {
    const Unit& u = P->units[ui];
    return ph_bn_bwd_apply_launch(g, a, ws + u.y_off, c.stat(u, 0), c.stat(u, 1), (const float*)params[ui * 6 + 1], c1, c2,
                                  dy, (size_t)P->B * u.OH * u.OW, u.Cout, P->prec, nullptr, nullptr, nullptr, st);
  };
  {
    const Block& b = P->blocks[7];
    if ((rc = ph_avgpool_bwd_launch(g_f4, gcur, P->B, b.OH * b.OW, b.Cout, 0, P->prec, st))) return rc;
  }
  for (int bi = 7; bi >= 0; --bi) {
    const Block& b = P->blocks[bi];
    if (bi == 5 && g_f3)
      if ((rc = ph_avgpool_bwd_launch(g_f3, gcur, P->B, b.OH * b.OW, b.Cout, 1, P->prec, st))) return rc;
    const void* out = ws + b.out_off;
    const void* a1 = ws + b.a1_off;
    if ((rc = bn_eval_bwd(b.u2, gcur, out, dyb))) return rc;
    if ((rc = conv_dgrad(c, b.u2, dyb, dab, nullptr, nullptr))) return rc;
    if ((rc = bn_eval_bwd(b.u1, dab, a1, dyb))) return rc;
    if (b.uds < 0) {
      if ((rc = conv_dgrad(c, b.u1, dyb, gnext, gcur, out))) return rc;
    } else {
      if ((rc = conv_dgrad(c, b.u1, dyb, gnext, nullptr, nullptr))) return rc;
      if ((rc = bn_eval_bwd(b.uds, gcur, out, dyb))) return rc;
      if ((rc = conv_dgrad(c, b.uds, dyb, gnext, gnext, nullptr))) return rc;
    }
    unsigned char* t = gcur; gcur = gnext; gnext = t;
  }
  const Unit& u = P->units[0];
  if ((rc = ph_stem_bwd_apply_launch(gcur, ws + P->idx_off, ws + u.y_off, c.stat(u, 0), c.stat(u, 1), c.stat(u, 2),
                                     c.stat(u, 3), (const float*)params[1], c1, c2, dyb, P->B, u.OH, u.OW, 64, P->prec, nullptr, st)))
    return rc;
  return ph_stem_dgrad_launch(dyb, (const float*)params[0], dx_nchw, P->B, P->H, P->W, P->prec, st);
}

// debug / test access: byte offset + dims of an intermediate activation in the workspace
//   what: 0 = unit raw output y (id = unit), 1 = block output (id = block), 2 = block a1, 3 = pooled stem, 4-9 below
int ph_resnet_tensor_info(const PhResnetPlan* P, int what, int id, size_t* byte_off, int* dims4) {
  if (!P) return PH_EINVAL;
  if (what == 0 && id >= 0 && id < (int)P->units.size()) {
    const Unit& u = P->units[id];
    *byte_off = u.y_off; dims4[0] = P->B; dims4[1] = u.OH; dims4[2] = u.OW; dims4[3] = u.Cout;
    return PH_OK;
  }
  if ((what == 1 || what == 2) && id >= 0 && id < (int)P->blocks.size()) {
    const Block& b = P->blocks[id];
    *byte_off = what == 1 ? b.out_off : b.a1_off;
    dims4[0] = P->B; dims4[1] = b.OH; dims4[2] = b.OW; dims4[3] = b.Cout;
    return PH_OK;
  }
  if (what == 3) {
    *byte_off = P->p0_off; dims4[0] = P->B; dims4[1] = P->PH0; dims4[2] = P->PW0; dims4[3] = 64;
    return PH_OK;
  }
  // backward scratch (dims are the caller's: the buffers are reused at every size): 4 / 5 = the two ping-pong block
  // gradient buffers, 6 = dy (BatchNorm-backward output), 7 = d_a1, 9 = the stem's max-pool arg codes (1 byte each)
  if (what >= 4 && what <= 7) {
    *byte_off = what == 4 ? P->g0_off : what == 5 ? P->g1_off : what == 6 ? P->dy_off : P->da_off;
    dims4[0] = dims4[1] = dims4[2] = dims4[3] = 0;
    return PH_OK;
  }
  if (what == 8 && id >= 0 && id < (int)P->units.size()) {   // fp32 [4][Cout]: mean, invstd, scale, shift of the unit's BatchNorm
    *byte_off = P->units[id].st_off; dims4[0] = 4; dims4[1] = P->units[id].Cout; dims4[2] = dims4[3] = 1;
    return PH_OK;
  }
  if (what == 9) {
    *byte_off = P->idx_off; dims4[0] = P->B; dims4[1] = P->PH0; dims4[2] = P->PW0; dims4[3] = 64;
    return PH_OK;
  }
  return PH_EINVAL;
}

}  // extern "C"
