// One whole refinement iteration of SCFlowDecoder.forward (models/decoder/scflow_decoder.py:196-243)
// behind ONE C entry point: the launch sequence -- 1/8 flow, lookup, motion encoder, SepConvGRU,
// flow / mask heads, delta-flow / mask encoders, full-resolution outputs, pose head, pose update,
// pose-induced flow -- lives here instead of in the caller's interpreter.  Nothing new is computed:
// every step is one of the library's own operator entry points, in the order and with the
// two-stream overlap scflow_amd/modules.py uses (bit-identical results; tests/test_gpu_refiner.py).
// At batch 1 an iteration is ~33 launches of a few microseconds each: sequenced from Python the pass
// is bound by the interpreter (~10 us per launch), sequenced here by the launch API alone.
#include "scf_common.h"

namespace {

// fork / join events of the optional side-stream branches: a small pool per (host thread, DEVICE), created on
// first use on that device (timing disabled) -- an event belongs to the device that was current when it was
// created, so a thread that drives cuda:0 and then cuda:1 must not reuse the first device's events.
// Re-recording an event after the wait that used it was enqueued is safe: hipStreamWaitEvent captures the
// record that precedes it.  The pools live as long as the thread (a handful of events per device).
struct IterEvents {
  hipEvent_t ev[7];
  bool ok = false;
  bool init() {
    if (ok) return true;
    for (int i = 0; i < 7; ++i)
      if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return false;
    return ok = true;
  }
};
IterEvents* iter_events() {
  static thread_local IterEvents pool[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  return pool[dev].init() ? &pool[dev] : nullptr;
}

bool fork_to(hipStream_t from, hipStream_t to, hipEvent_t ev) {
  return hipEventRecord(ev, from) == hipSuccess && hipStreamWaitEvent(to, ev, 0) == hipSuccess;
}

// A side-stream branch that is open when the call leaves early (a launch inside it failed) is joined on the way
// out: otherwise later work on the main stream would no longer be ordered behind the side stream's writes to the
// scratch buffers, and under hipGraph capture the unjoined stream would invalidate the capture with an unrelated
// error.
struct OpenBranch {
  hipStream_t mainq, sideq;
  hipEvent_t join_ev;
  bool open = false;
  ~OpenBranch() {
    if (open) (void)fork_to(sideq, mainq, join_ev);
  }
};

#define SCF_TRY(call)                \
  do {                               \
    const int rc_ = (call);          \
    if (rc_ != SCF_OK) return rc_;   \
  } while (0)

}  // namespace

// scf_tune(SCF_TUNE_ITER_MERGE, 0): the r4 launch sequence (copy of the 1/8 flow, the two up-samplings, pose update and
// re-projection as launches of their own) for A/B measurements; 1 (default): merged
static std::atomic<int> g_iter_merge{1};
int scf_iter_merge_set(int v) { return (v < 0 || v > 1) ? SCF_EINVAL : g_iter_merge.exchange(v); }

extern "C" int scf_scflow_iteration(const scf_scflow_iter* it, scf_stream_t stream) {
  if (!it) return SCF_EINVAL;
  const bool merge = g_iter_merge.load(std::memory_order_relaxed) != 0;
  if (it->struct_size != (int32_t)sizeof(scf_scflow_iter)) return SCF_EINVAL;      // header / binding mismatch
  if (!it->flow_in || !it->flow_out || !it->flow_pred || !it->mask_up || !it->flow_lr || !it->corr ||
      !it->R_in || !it->t_in || !it->R_out || !it->t_out || !it->d_rot || !it->d_trans || !it->hx ||
      it->N <= 0 || it->H <= 0 || it->W <= 0 || it->h <= 0 || it->w <= 0 || it->npass <= 0 || it->npass > 2)
    return SCF_EINVAL;
  if ((it->mask_flow || it->mask_corr) && !it->mask_prev) return SCF_EINVAL;
  // overlap_* : 0 = in order, 1 = the branch on the side stream, 2 (r6) = the branch's convolutions ride in the main branch's
  // launches (scf_conv2d_pair): same kernels, same results, no second stream
  const bool pair_flow = it->overlap_flow == 2, pair_mask = it->overlap_mask == 2;
  const bool any_overlap = it->overlap_flow == 1 || it->overlap_mask == 1 || it->overlap_up == 1;
  if (any_overlap && !it->side_stream) return SCF_EINVAL;
  hipStream_t mainq = scf_stream(stream), sideq = scf_stream(it->side_stream);
  IterEvents* Ep = any_overlap ? iter_events() : nullptr;
  if (any_overlap && !Ep) return SCF_ELAUNCH;
  static IterEvents none;                 // never dereferenced without overlap
  IterEvents& E = Ep ? *Ep : none;
  OpenBranch br{mainq, sideq, E.ev[6]};
  const int N = it->N, hw = it->h * it->w;
  const float scale = (float)it->H / (float)it->h;

  // ---- 1/8-resolution flow (scflow_decoder.py:196-197) ----
  // (r5: without flow masking the 1/8 flow is also the motion encoder's input, and its copy into the GRU input buffer
  // rides in the same launch: a second destination instead of a copy launch per iteration)
  const bool dual = merge && !it->mask_flow && it->flow_copy_dst != nullptr;
  SCF_TRY(scf_resize_bilinear_jobs(it->flow_in, nullptr, it->flow_lr, (int64_t)N * 2, 1.0f / scale, nullptr, nullptr, nullptr,
                                   0, 1.0f, dual ? it->flow_copy_dst : nullptr, 2, it->hx_nstride, it->H, it->W, it->h,
                                   it->w, stream));
  const float* flow_enc = it->flow_lr;              // what the motion encoder sees (:203-206)
  if (it->mask_flow) {
    if (!it->flow_masked) return SCF_EINVAL;
    SCF_TRY(scf_mul_mask(it->flow_lr, (int64_t)2 * hw, it->mask_prev, it->flow_masked, (int64_t)2 * hw, N, 2, hw, stream));
    flow_enc = it->flow_masked;
  }
  if (it->overlap_flow == 1) {
    if (!fork_to(mainq, sideq, E.ev[0])) return SCF_ELAUNCH;
    br.open = true;
  }
  // ---- correlation lookup (:198) ----
  if (it->lookup_timer) scf_timer_arm(it->lookup_timer);
  const int rl = scf_corr_lookup_ex(it->levels, it->flow_lr, it->corr, N, it->h, it->w, it->radius, it->L,
                                    it->tiled_levels, stream);
  if (it->lookup_timer) scf_timer_arm(nullptr);
  SCF_TRY(rl);
  if (it->mask_corr)
    SCF_TRY(scf_mul_mask(it->corr, (int64_t)it->corr_channels * hw, it->mask_prev, it->corr,
                         (int64_t)it->corr_channels * hw, N, it->corr_channels, hw, stream));
  // ---- motion encoder (raft_decoder.py:152-166): flow branch beside the correlation branch ----
  {
    scf_stream_t fq = it->overlap_flow == 1 ? it->side_stream : stream;
    scf_conv_desc f0 = it->flow0;
    f0.in0 = flow_enc;
    if (pair_flow) {            // the flow branch layer by layer beside the correlation branch: [corr0 | flow0], [corr1 | flow1]
      SCF_TRY(scf_conv2d_pair(&it->corr0, &f0, stream));
      SCF_TRY(scf_conv2d_pair(&it->corr1, &it->flow1, stream));
    } else {
      SCF_TRY(scf_conv2d(&f0, fq));
      SCF_TRY(scf_conv2d(&it->flow1, fq));
      SCF_TRY(scf_conv2d(&it->corr0, stream));
      SCF_TRY(scf_conv2d(&it->corr1, stream));
    }
    if (it->overlap_flow == 1) {
      br.open = false;
      if (!fork_to(sideq, mainq, E.ev[1])) return SCF_ELAUNCH;
    }
    SCF_TRY(scf_conv2d(&it->outn, stream));
    if (!dual)
      SCF_TRY(scf_copy_strided(flow_enc, (int64_t)2 * hw, it->flow_copy_dst, it->hx_nstride, N, (int64_t)2 * hw, stream));
  }
  // ---- SepConvGRU (:207-208), context part hoisted ----
  if (it->ctx[0])
    SCF_TRY(scf_sepconv_gru_ctx(it->hx, it->hx_nstride, N, it->Ch, it->Cc, it->Cx, it->h, it->w, it->gru, it->npass,
                                it->ctx, it->ctx_nstride, it->z, it->rh, stream));
  else
    SCF_TRY(scf_sepconv_gru(it->hx, it->hx_nstride, N, it->Ch, it->Cc + it->Cx, it->h, it->w, it->gru, it->npass,
                            it->z, it->rh, stream));
  // ---- heads (:210-213) and their encoders (:216-217) ----
  SCF_TRY(scf_conv2d(&it->heads, stream));
  // r5: with the mask branch on the side stream, the mask prediction goes with it -- [mask head -> mask encoder] beside
  // [flow head -> delta-flow encoder] instead of both predictions in a row on the main stream
  const bool mpred_aside = merge && it->overlap_mask == 1;
  if (mpred_aside) {                   // the side branch starts behind the heads' hidden layer
    if (!fork_to(mainq, sideq, E.ev[2])) return SCF_ELAUNCH;
    br.open = true;
  }
  if (pair_mask) {              // the two predictions read disjoint halves of the hidden tensor: one launch on small grids
    SCF_TRY(scf_conv2d_pair(&it->fpred, &it->mpred, stream));
  } else {
    SCF_TRY(scf_conv2d(&it->fpred, stream));
    if (!mpred_aside) SCF_TRY(scf_conv2d(&it->mpred, stream));
  }
  if (pair_mask) {              // [delta-flow encoder | mask encoder] layer by layer in shared launches
    SCF_TRY(scf_conv2d_pair(&it->denc0, &it->menc0, stream));
    SCF_TRY(scf_conv2d_pair(&it->denc1, &it->menc1, stream));
  } else {
    scf_stream_t mq = it->overlap_mask == 1 ? it->side_stream : stream;
    if (it->overlap_mask == 1 && !mpred_aside) {
      if (!fork_to(mainq, sideq, E.ev[2])) return SCF_ELAUNCH;
      br.open = true;
    }
    if (mpred_aside) SCF_TRY(scf_conv2d(&it->mpred, mq));
    SCF_TRY(scf_conv2d(&it->menc0, mq));
    SCF_TRY(scf_conv2d(&it->menc1, mq));
    SCF_TRY(scf_conv2d(&it->denc0, stream));
    SCF_TRY(scf_conv2d(&it->denc1, stream));
    if (it->overlap_mask == 1) {
      br.open = false;
      if (!fork_to(sideq, mainq, E.ev[3])) return SCF_ELAUNCH;
    }
  }
  // ---- full-resolution outputs (:222-227): they feed nothing, beside the pose head ----
  {
    scf_stream_t uq = it->overlap_up == 1 ? it->side_stream : stream;
    if (it->overlap_up == 1) {
      if (!fork_to(mainq, sideq, E.ev[4])) return SCF_ELAUNCH;
      br.open = true;
    }
    // flow + delta flow (x scale) and the mask, one launch (r5)
    if (!merge) {
      SCF_TRY(scf_resize_bilinear(it->flow_lr, it->fpred.out, it->flow_pred, (int64_t)N * 2, it->h, it->w, it->H, it->W,
                                  scale, uq));
      SCF_TRY(scf_resize_bilinear(it->mpred.out, nullptr, it->mask_up, (int64_t)N, it->h, it->w, it->H, it->W, 1.0f, uq));
    } else {
      SCF_TRY(scf_resize_bilinear_jobs(it->flow_lr, it->fpred.out, it->flow_pred, (int64_t)N * 2, scale, it->mpred.out, nullptr,
                                       it->mask_up, (int64_t)N, 1.0f, nullptr, 1, 0, it->h, it->w, it->H, it->W, uq));
    }
  }
  // ---- pose head (pose_head.py:201-211) ----
  for (int i = 0; i < 3; ++i) {
    SCF_TRY(scf_conv2d(&it->pose[i], stream));
    const scf_iter_gn& g = it->gn[i];
    if (it->fc_fused && i == 2) break;          // the last GroupNorm + ReLU is folded into fc1's operand load
    const scf_conv_desc& pc = it->pose[i];      // K split across blocks: the normalisation adds the partial tensors
    SCF_TRY(scf_group_norm_relu_parts(pc.out, pc.k_slices > 1 ? pc.k_slices : 1, pc.out_slice_stride, g.gamma, g.beta, g.out,
                                      N, g.C, g.HW, g.G, g.eps, stream));
  }
  if (it->fc_fused) {
    const scf_iter_gn& g = it->gn[2];
    scf_fc_desc f = {};
    f.x = it->pose[2].out; f.x_parts = it->pose[2].k_slices > 1 ? it->pose[2].k_slices : 1;
    f.x_part_stride = it->pose[2].out_slice_stride; f.gn_groups = g.G; f.gn_hw = g.HW; f.gn_gamma = g.gamma; f.gn_beta = g.beta;
    f.gn_eps = g.eps; f.N = N; f.K = it->fc1_K; f.W = it->fc1_w; f.y = it->fc1_out; f.O = it->fc1_O; f.slices = it->fc1_slices;
    SCF_TRY(scf_fc_splitk(&f, stream));
    scf_fc_desc f2 = {};
    f2.x = it->fc1_out; f2.x_parts = it->fc1_slices; f2.x_part_stride = (int64_t)N * it->fc1_O; f2.x_bias = it->fc1_b;
    f2.x_relu = 1; f2.N = N; f2.K = it->fc1_O; f2.W = it->fc2_w; f2.y = it->fc2_out; f2.O = it->fc2_O; f2.slices = it->fc2_slices;
    SCF_TRY(scf_fc_splitk(&f2, stream));
    scf_fc_desc f3 = {};
    f3.x = it->fc2_out; f3.x_parts = it->fc2_slices; f3.x_part_stride = (int64_t)N * it->fc2_O; f3.x_bias = it->fc2_b;
    f3.x_relu = 1; f3.N = N; f3.K = it->fc2_O; f3.W = it->rot_w; f3.bias = it->rot_b; f3.y = it->rot_all; f3.O = it->rot_O;
    f3.W2 = it->trans_w; f3.bias2 = it->trans_b; f3.y2 = it->trans_all; f3.O2 = it->trans_O; f3.act = SCF_ACT_NONE; f3.slices = 1;
    SCF_TRY(scf_fc_splitk(&f3, stream));
  } else {
    SCF_TRY(scf_linear(it->gn[2].out, it->fc1_w, it->fc1_b, it->fc1_out, N, it->fc1_K, it->fc1_O, SCF_ACT_RELU, stream));
    SCF_TRY(scf_linear(it->fc1_out, it->fc2_w, it->fc2_b, it->fc2_out, N, it->fc1_O, it->fc2_O, SCF_ACT_RELU, stream));
    SCF_TRY(scf_linear_pair(it->fc2_out, it->rot_w, it->rot_b, it->rot_all, it->rot_O, it->trans_w, it->trans_b,
                            it->trans_all, it->trans_O, N, it->fc2_O, SCF_ACT_NONE, stream));
  }
  // ---- pose update (pose.py:124-169) and the pose-induced flow (pose.py:44-88) ----
  // one launch (r5): every block of a sample recomputes the sample's pose update; an in-place update keeps two launches
  {
    const int rf = !merge ? SCF_EUNSUPPORTED : scf_pose_update_reproject(it->rot_all, it->trans_all, it->label, it->num_class, it->label_mode, it->R_in,
                                             it->t_in, it->d_rot, it->d_trans, it->R_out, it->t_out, it->depth, it->K, it->R0,
                                             it->t0, it->flow_out, N, it->H, it->W, it->invalid_flow_num, stream);
    if (rf == SCF_EUNSUPPORTED) {
      SCF_TRY(scf_pose_update(it->rot_all, it->trans_all, it->label, it->num_class, it->label_mode, it->R_in, it->t_in,
                              it->d_rot, it->d_trans, it->R_out, it->t_out, N, stream));
      SCF_TRY(scf_reproject_flow(it->depth, it->K, it->R0, it->t0, it->R_out, it->t_out, it->flow_out, N, it->H, it->W,
                                 it->invalid_flow_num, stream));
    } else {
      SCF_TRY(rf);
    }
  }
  if (it->overlap_up == 1) {
    br.open = false;
    if (!fork_to(sideq, mainq, E.ev[5])) return SCF_ELAUNCH;
  }
  return SCF_OK;
}
