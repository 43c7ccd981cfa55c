/*
 * catre_hip.h - C ABI of the MI355X (gfx950) implementation of CATRE's pose-refine hot path.
 *
 * The reference (THU-DA-6D-Pose-Group/CATRE) is 100 % Python on stock torch ops and has no FFI
 * of its own; each entry point below names the reference Python function (file:line, relative
 * to the reference root) whose arithmetic it replaces.  Host code binds these with ctypes
 * (catre_amd/hip.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless noted; tensors are dense row-major
 *     unless strides are passed explicitly (strides are in ELEMENTS);
 *   - B objects, N observed points, M shape-prior points per object; "cloud" c in [0,2B):
 *     c <  B -> observed cloud of object c      (N points)
 *     c >= B -> transformed prior of object c-B (M points)
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*); nothing
 *     allocates, frees or synchronises; scratch comes from the caller's `workspace`;
 *   - return value: CATRE_OK (0) or a negative catre_status; kernels never abort.
 */
#ifndef CATRE_HIP_H
#define CATRE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum catre_status {
  CATRE_OK = 0,
  CATRE_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, unsupported option */
  CATRE_ERR_WORKSPACE = -2,    /* workspace / packed buffer too small */
  CATRE_ERR_LAUNCH = -3,       /* hipGetLastError() != hipSuccess after a launch */
  CATRE_ERR_UNSUPPORTED = -4   /* configuration outside what the kernels implement */
} catre_status;

/* Parameter tensors, in the reference's state_dict layout.  Conv1d weights
 * are [out,in,1] == [out,in] row-major.  `params[CATRE_P_xxx]` is the device pointer. */
typedef enum catre_param {
  CATRE_P_STN_CONV1_W = 0, CATRE_P_STN_CONV1_B,   /* pcl_net.stn.conv1   [64,3]      */
  CATRE_P_STN_CONV2_W, CATRE_P_STN_CONV2_B,       /* pcl_net.stn.conv2   [128,64]    */
  CATRE_P_STN_CONV3_W, CATRE_P_STN_CONV3_B,       /* pcl_net.stn.conv3   [1024,128]  */
  CATRE_P_STN_FC1_W, CATRE_P_STN_FC1_B,           /* pcl_net.stn.fc1     [512,1024]  */
  CATRE_P_STN_FC2_W, CATRE_P_STN_FC2_B,           /* pcl_net.stn.fc2     [256,512]   */
  CATRE_P_STN_FC3_W, CATRE_P_STN_FC3_B,           /* pcl_net.stn.fc3     [9,256]     */
  CATRE_P_CONV1_W, CATRE_P_CONV1_B,               /* pcl_net.conv1       [64,3]      */
  CATRE_P_CONV2_W, CATRE_P_CONV2_B,               /* pcl_net.conv2       [128,64]    */
  CATRE_P_CONV3_W, CATRE_P_CONV3_B,               /* pcl_net.conv3       [512,128]   */
  CATRE_P_CONV4_W, CATRE_P_CONV4_B,               /* pcl_net.conv4       [1024,512]  */
  CATRE_P_FSTN_CONV1_W, CATRE_P_FSTN_CONV1_B,     /* pcl_net.fstn.conv1  [64,64]     */
  CATRE_P_FSTN_CONV2_W, CATRE_P_FSTN_CONV2_B,     /* pcl_net.fstn.conv2  [128,64]    */
  CATRE_P_FSTN_CONV3_W, CATRE_P_FSTN_CONV3_B,     /* pcl_net.fstn.conv3  [1024,128]  */
  CATRE_P_FSTN_FC1_W, CATRE_P_FSTN_FC1_B,         /* pcl_net.fstn.fc1    [512,1024]  */
  CATRE_P_FSTN_FC2_W, CATRE_P_FSTN_FC2_B,         /* pcl_net.fstn.fc2    [256,512]   */
  CATRE_P_FSTN_FC3_W, CATRE_P_FSTN_FC3_B,         /* pcl_net.fstn.fc3    [4096,256]  */
  /* rot_head.rot_head_x.* then rot_head.rot_head_y.* (12 tensors each) */
  CATRE_P_ROTX_L0_W, CATRE_P_ROTX_L0_B,           /* layers.0  [256,1088]            */
  CATRE_P_ROTX_GN0_W, CATRE_P_ROTX_GN0_B,         /* layers.1  [256]                 */
  CATRE_P_ROTX_L1_W, CATRE_P_ROTX_L1_B,           /* layers.3  [256,256]             */
  CATRE_P_ROTX_GN1_W, CATRE_P_ROTX_GN1_B,         /* layers.4  [256]                 */
  CATRE_P_ROTX_NECK_W, CATRE_P_ROTX_NECK_B,       /* neck.0    [3,256]               */
  CATRE_P_ROTX_CONVP_W, CATRE_P_ROTX_CONVP_B,     /* conv_p    [1,N+M] , [1] (bias may be NULL) */
  CATRE_P_ROTY_L0_W, CATRE_P_ROTY_L0_B,
  CATRE_P_ROTY_GN0_W, CATRE_P_ROTY_GN0_B,
  CATRE_P_ROTY_L1_W, CATRE_P_ROTY_L1_B,
  CATRE_P_ROTY_GN1_W, CATRE_P_ROTY_GN1_B,
  CATRE_P_ROTY_NECK_W, CATRE_P_ROTY_NECK_B,
  CATRE_P_ROTY_CONVP_W, CATRE_P_ROTY_CONVP_B,
  CATRE_P_TS_L0_W, CATRE_P_TS_L0_B,               /* ts_head.linears.0 [256,ts_in]   */
  CATRE_P_TS_GN0_W, CATRE_P_TS_GN0_B,             /* ts_head.linears.1 [256]         */
  CATRE_P_TS_L1_W, CATRE_P_TS_L1_B,               /* ts_head.linears.3 [256,256]     */
  CATRE_P_TS_GN1_W, CATRE_P_TS_GN1_B,             /* ts_head.linears.4 [256]         */
  CATRE_P_TS_FCT_W, CATRE_P_TS_FCT_B,             /* ts_head.fc_t      [3,256]       */
  CATRE_P_TS_FCS_W, CATRE_P_TS_FCS_B,             /* ts_head.fc_s      [3,256]       */
  CATRE_P_COUNT
} catre_param;

/* Options of one refine iteration: the flags `CATRE_disR_shared.forward` reads from
 * cfg.MODEL.CATRE.{ROT_HEAD,TS_HEAD} and cfg.INPUT (core/catre/models/CATRE_disR_shared.py:57-120). */
typedef struct catre_opts {
  int32_t feature_transform;   /* PCLNET.INIT_CFG.feature_transform (pointnet.py:105)            */
  int32_t with_kps_feature;    /* TS_HEAD.WITH_KPS_FEATURE  (CATRE_disR_shared.py:71-75)         */
  int32_t with_init_scale;     /* TS_HEAD.WITH_INIT_SCALE   (:78-79)                             */
  int32_t with_init_trans;     /* TS_HEAD.WITH_INIT_TRANS   (:80-82)                             */
  int32_t delta_t_space_3d;    /* 0: "image", 1: "3D"  (pose_scale_from_delta_init.py:51,72)     */
  int32_t delta_z_deepim;      /* 0: "cosypose", 1: "deepim" (:55-61)                            */
  int32_t k_aware;             /* T_TRANSFORM_K_AWARE (:63-69)                                   */
  int32_t scale_mul;           /* 0: "add" in SCLAE_TYPE, 1: multiplicative exp (:79-84)          */
  int32_t scale_base_mean;     /* 0: "iter" in SCLAE_TYPE (base = current scale), 1: mean_scales  */
  int32_t is_allo;             /* "allo" in ROT_TYPE (:87-90)                                    */
  int32_t refine_scale;        /* cfg.MODEL.REFINE_SCLAE (CATRE_disR_shared.py:119-120)          */
  int32_t zero_center;         /* cfg.INPUT.ZERO_CENTER_INPUT (engine/batch_test.py:84-97)       */
  float   delta_t_weight;      /* DELTA_T_WEIGHT (:48)                                           */
  float   allo_eps;            /* eps of allo_to_ego_mat_torch, 1e-4 (CATRE_disR_shared.py:112)  */
  int32_t ts_in_dim;           /* TS_HEAD.INIT_CFG.in_dim; must equal the gathered feature width */
  int32_t rot_input_is_matrix; /* catre_pose_update only: `rot6d` holds [B,3,3] matrices (get_rot_mat already applied) */
  int32_t compute_dtype;       /* catre_refine_iter / catre_refine_k: CATRE_DTYPE_F32 (default) or CATRE_DTYPE_BF16 -
                                * what torch.cuda.amp.autocast selects in the reference (engine.py:304, TEST.AMP_TEST):
                                * bf16 GEMM operands, fp32 accumulation / GroupNorm statistics / SO(3) update          */
  int32_t rot_type;            /* parametrisation of the rotation residual, the part of ROT_HEAD.ROT_TYPE after ego_/allo_
                                * (get_rot_mat, core/catre/models/model_utils.py:28-40): CATRE_ROT_6D (default, [B,6]),
                                * CATRE_ROT_QUAT ([B,4]), CATRE_ROT_LOG_QUAT ([B,3]), CATRE_ROT_LIE_VEC ([B,3]).  The fused
                                * drivers take the two rot heads' concatenated output [B, 2*rot_dim], so they accept the
                                * types whose width is even: 6D (rot_dim 3) and QUAT (rot_dim 2) - exactly the ones the
                                * reference's ConvOutPerRotHead can feed to get_rot_mat                                 */
} catre_opts;

enum { CATRE_ROT_6D = 0, CATRE_ROT_QUAT = 1, CATRE_ROT_LOG_QUAT = 2, CATRE_ROT_LIE_VEC = 3 };

/* CATRE_DTYPE_SPLIT: fp32 results from split-bf16 (hi + lo, three products) MFMAs on the layers holding 98 % of the FLOPs;
 * same parity bound as CATRE_DTYPE_F32 */
enum { CATRE_DTYPE_F32 = 0, CATRE_DTYPE_BF16 = 1, CATRE_DTYPE_SPLIT = 2 };
/* OR-ed into the compute_dtype of catre_op_gemm_rows_nr / catre_op_gemm_tn_bias_nr (with CATRE_DTYPE_BF16): the row-indexed
 * dense tensor - `mask` / `X` - holds bf16 rows (leading dimension in ELEMENTS): the activation rows the autocast encoder
 * forward saves (catre_train_*_fwd with CATRE_DTYPE_BF16) */
enum { CATRE_ROWS_BF16 = 0x100 };

/* ---- sizes ---------------------------------------------------------------------------- */

/* Bytes of scratch one refine iteration needs for (B,N,M).  0 on bad arguments. */
size_t catre_workspace_bytes(int B, int N, int M);

/* Floats of the fragment-packed weight image produced by catre_pack_weights. */
size_t catre_packed_floats(int N, int M, int ts_in_dim);

/* Re-lay the weights the MFMA kernels stream (conv / head GEMM operands) into wave-fragment
 * order, transpose the ts-head matrices and pre-reduce conv_p.  Must be re-run whenever the
 * parameters change (e.g. after an optimizer step).  No reference counterpart (layout only). */
int catre_pack_weights(const float* const* params, int N, int M, int ts_in_dim,
                       float* packed, size_t packed_floats, void* stream);

/* Same, restricted to the packs a caller needs (a training step that re-packs after every optimizer step only needs
 * the fp32 encoder image): OR of CATRE_PACK_*. */
enum {
  CATRE_PACK_F32_ENCODER = 1,
  CATRE_PACK_F32_HEADS = 2,  /* the rotation heads' two linears as fp32 fragments */
  CATRE_PACK_BF16 = 4,
  CATRE_PACK_SPLIT = 8,
  CATRE_PACK_F32_TAILS = 16, /* with _F32_HEADS: the ts head's transposed weights and the conv_p weight sums - read by the
                                inference tail kernels only, so a training forward (which re-packs every iteration) leaves
                                these four launches out */
  CATRE_PACK_ALL = 31
};
int catre_pack_weights_sel(const float* const* params, int N, int M, int ts_in_dim, float* packed, size_t packed_floats,
                           int sel, void* stream);

/* ---- single stages (each also usable on its own; tests check them one by one) ---------- */

/* a1: batch_updater_test core (core/catre/engine/batch_test.py:81-97) with
 * transform_normed_pts_batch (lib/pysixd/misc.py:1001-1026).
 * x_out [B,N,3] point-major = pcl - t (or pcl if !zero_center); kps_out [B,M,3] = R (kps*s) (+t). */
int catre_pose_apply(const float* pcl, const float* kps, const float* pose /*[B,3,4]*/,
                     const float* scale /*[B,3]*/, float* x_out, float* kps_out,
                     int B, int N, int M, int zero_center, void* stream);

/* Point arrays are described by a base pointer and element strides (batch, point, coord), so the
 * permuted [B,3,N] views the reference passes (SURVEY.md 8b "forward") are consumed in place. */
typedef struct catre_points {
  const float* obs;  int64_t obs_sb, obs_sn, obs_sc;   /* observed cloud  x        [B,N,3] */
  const float* kps;  int64_t kps_sb, kps_sn, kps_sc;   /* transformed prior tfd_kps [B,M,3] */
  /* apply_pose != 0: obs / kps describe the RAW batch["pcl"] / batch["obj_kps"] and the pose-apply of
   * batch_updater_test (engine/batch_test.py:81-97: x = pcl - t, tfd_kps = R (kps * s) [+ t]) is done on the fly as
   * the points are loaded, with pose [B,3,4] / scale [B,3] - what catre_refine_k does instead of materialising x and
   * tfd_kps every iteration.  0 (and NULL pointers): the arrays already hold x / tfd_kps. */
  const float* pose;  const float* scale;
  int32_t apply_pose, zero_center;
} catre_points;

/* a2: STN3d conv stack + max-pool (core/catre/models/pointnets/pointnet.py:24-29).
 * pooled [2B,1024]. */
int catre_stn3d_pool(const catre_points* pts, const float* const* params, const float* packed,
                     float* pooled, void* workspace, size_t ws_bytes, int B, int N, int M, void* stream);

/* a2/a4 tails: y = act(x W^T + b) (+ I_k flattened), torch.nn.functional.linear as used by
 * pointnet.py:31-40,64-77.  x [R,K] (ldx), W [J,K] (ldw), y [R,J] (ldy). K % 8 == 0. */
int catre_linear(const float* x, int ldx, const float* W, int ldw, const float* bias, float* y, int ldy,
                 int R, int J, int K, int relu, int add_identity_k, void* stream);
/* y[R,J] = x[R,K] W with W [K][J] row-major (ldw): catre_linear on the transposed weight without a transposed copy - the
 * data gradient of a small linear layer (autograd of the FC tails / FC_TransSizeHead, heads/fc_trans_size_head.py:98-116)
 * from the layer's own [out][in] weight.  K % 8 == 0.  xmask (optional, laid out like x): x .* (xmask > 0) replaces x (the
 * ReLU backward of the layer's output folded into the operand load). */
int catre_linear_t(const float* x, int ldx, const float* xmask, const float* W, int ldw, float* y, int ldy, int R, int J,
                   int K, void* stream);

/* a3+a4: x' = x T3, relu(conv1), STNkd conv stack + max-pool (pointnet.py:98-103,57-62).
 * trans3 [2B,3,3] -> pooled [2B,1024]. */
int catre_stnkd_pool(const catre_points* pts, const float* trans3, const float* const* params,
                     const float* packed, float* pooled, void* workspace, size_t ws_bytes,
                     int B, int N, int M, void* stream);

/* a3+a5: point feature transform, conv2..conv4, max-pool (pointnet.py:98-116).
 * trans64 [2B,64,64] (NULL when !feature_transform) -> gfeat [2B,1088] = [max_n conv4 | max_n pointfeat],
 * pointfeat written point-major: obs [B,N,64] then prior [B,M,64] (one buffer of B*(N+M)*64 floats). */
int catre_trunk(const catre_points* pts, const float* trans3, const float* trans64,
                const float* const* params, const float* packed, float* gfeat, float* pointfeat,
                void* workspace, size_t ws_bytes, int B, int N, int M, void* stream);

/* a7+a8: feature gather + FC_TransSizeHead.forward (CATRE_disR_shared.py:69-84,
 * heads/fc_trans_size_head.py:61-70).  -> trans_deltas [B,3], scale_deltas [B,3]. */
int catre_ts_head(const float* gfeat, const float* init_pose, const float* init_scale,
                  const float* const* params, const float* packed, const catre_opts* opts,
                  float* trans_deltas, float* scale_deltas, void* workspace, size_t ws_bytes, int B, void* stream);

/* a7+a9: ConvOutPerRotHead.forward on cat(pcl_feat,kps_feat,dim=2) without materialising it
 * (CATRE_disR_shared.py:86-88, heads/conv_out_per_rot_head.py:62-71,126-140). -> rot6d [B,6]. */
int catre_rot_head(const float* gfeat, const float* pointfeat, const float* const* params,
                   const float* packed, float* rot6d, void* workspace, size_t ws_bytes,
                   int B, int N, int M, void* stream);
/* Same with neck width rot_dim in {1,2,3} per head (RotHead's `rot_dim` constructor argument,
 * conv_out_per_rot_head.py:80,109) -> rot [B, 2*rot_dim]. */
int catre_rot_head_dim(const float* gfeat, const float* pointfeat, const float* const* params,
                       const float* packed, float* rot, void* workspace, size_t ws_bytes,
                       int B, int N, int M, int rot_dim, void* stream);

/* a10 on its own: get_rot_mat (core/catre/models/model_utils.py:28-40): rot [B,d] -> R [B,3,3] for
 * rot_type in CATRE_ROT_* (d = 6 / 4 / 3 / 3): rot6d_to_mat_batch (core/utils/rot_reps.py:34-55), quat2mat_torch
 * (core/utils/pose_utils.py:349-412), quat2mat_torch(qexp(.)) (core/utils/quaternion_lf.py:294-317), lie_vec_to_rot
 * (core/utils/lie_algebra.py:7-77); and its backward grad_R [B,3,3] -> grad_rot [B,d]. */
int catre_rot_to_mat(const float* rot, int rot_type, float* R_out, int B, void* stream);
int catre_rot_to_mat_bwd(const float* rot, int rot_type, const float* grad_R, float* grad_rot, int B, void* stream);

/* a10+a11+a12: rot6d_to_mat_batch (core/utils/rot_reps.py:34-55) and pose_scale_from_delta_init
 * (core/catre/models/pose_scale_from_delta_init.py:8-95).  Ks / mean_scales may be NULL when unused.
 * rot6d is [B,d] in the parametrisation opts->rot_type names (d = 6 / 4 / 3 / 3), or [B,3,3] rotation matrices when
 * opts->rot_input_is_matrix.
 * -> pose_out [B,3,4], scale_out [B,3]. */
int catre_pose_update(const float* rot6d, const float* trans_deltas, const float* scale_deltas,
                      const float* init_pose, const float* init_scale, const float* mean_scales,
                      const float* Ks, const catre_opts* opts, float* pose_out, float* scale_out,
                      int B, void* stream);

/* ---- fused drivers ------------------------------------------------------------------------ */

/* One CATRE_disR_shared.forward (test path, CATRE_disR_shared.py:57-124) on given x / tfd_kps.
 * Batches of up to 8 objects (the evaluator's one-image calls, catre_evaluator.py:292-311) take a latency path of 14-17
 * launches instead of 22 (csrc/catre_small.h); the result of an object does not depend on which path its batch took,
 * bit for bit. */
int catre_refine_iter(const catre_points* pts, const float* init_pose, const float* init_scale,
                      const float* mean_scales, const float* Ks, const float* const* params,
                      const float* packed, const catre_opts* opts, float* pose_out, float* scale_out,
                      void* workspace, size_t ws_bytes, int B, int N, int M, void* stream);

/* The K-loop of catre_inference_on_dataset (core/catre/engine/catre_evaluator.py:292-311):
 * for i in 1..n_iter: pose-apply (batch_test.py:63-99) -> forward -> feed back.
 * poses [n_iter+1,B,3,4] / scales [n_iter+1,B,3]: slot 0 holds the initial estimate on entry,
 * slots 1..n_iter are written.  pcl [B,N,3], kps [B,M,3] point-major contiguous. */
int catre_refine_k(const float* pcl, const float* kps, const float* mean_scales, const float* Ks,
                   const float* const* params, const float* packed, const catre_opts* opts,
                   float* poses, float* scales, void* workspace, size_t ws_bytes,
                   int B, int N, int M, int n_iter, void* stream);

/* The same loop with slot 0 as an OUTPUT: iteration 1 reads the caller's init_pose [B,3,4] / init_scale [B,3]
 * (`out_dict["pose_0"]`, catre_evaluator.py:292) and its pose-update kernel copies them into slot 0 - no copy launch
 * in front of the loop (n_iter == 0: two device-to-device copies). */
int catre_refine_k_from(const float* pcl, const float* kps, const float* init_pose, const float* init_scale,
                        const float* mean_scales, const float* Ks, const float* const* params, const float* packed,
                        const catre_opts* opts, float* poses, float* scales, void* workspace, size_t ws_bytes,
                        int B, int N, int M, int n_iter, void* stream);

/* ---- stand-alone memory-bound kernels (HBM roofline figures of SURVEY.md 8d) --------------- */

/* torch.max(x, 2)[0] for x [B,C,N] contiguous -> out [B,C]  (pointnet.py:28,61,115). */
int catre_colmax(const float* x, float* out, int B, int C, int N, void* stream);

/* ---- opt-in kernel timing: a MEASUREMENT AID for bench.py, fenced off from the data path's contract ---------------
 * Everything above keeps no state between calls and is re-entrant.  The two functions below are the exception and are
 * kept apart for that reason: they hold one process-global record table (mutex-guarded), do nothing unless
 * catre_profile_enable was called, and a product build may drop them altogether (-DCATRE_NO_PROFILING: both return
 * CATRE_ERR_UNSUPPORTED and the launch hooks compile to nothing).  catre_debug_trunk_trace (per-phase cycle stamps) only
 * works in the instrumented library (`make TRACE=1` -> libcatre_hip_trace.so); the product library returns
 * CATRE_ERR_UNSUPPORTED and its kernels carry no stamp code. */
typedef enum catre_kernel_id {
  CATRE_K_STN3D = 0, CATRE_K_STNKD, CATRE_K_TRUNK, CATRE_K_TS_HEAD, CATRE_K_ROT_L0_STATS, CATRE_K_ROT_L1,
  CATRE_K_ROT_OUT, CATRE_K_COLMAX, CATRE_K_COUNT
} catre_kernel_id;

/* Record a HIP event pair around every launch of `kernel_id` on its launch stream (up to max_records).
 * kernel_id < 0 or max_records <= 0 disables and frees the events. */
int catre_profile_enable(int kernel_id, int max_records);
/* Kernel-form switches (A/B measurements, tests): where a stage has more than one kernel form for full grids - all forms
 * give the same bits - `id` selects the switch (0: one-wave-per-SIMD trunk k_trunk4, 1: one-wave STN kernels, 2: STN kernels
 * on pairs of tiles, 3: one-wave rotation-head kernel k_rot_l1w, 4: one-launch FC tails of small batches k_fc_tail), `value` 1 / 0 sets it, value < 0 only queries.  Returns
 * the PREVIOUS setting (1 / 0), -1 for an unknown id.  Process-wide (an atomic word; defaults: 0-2 on, 3 and 4 off - they
 * measured slower - or what the environment says: CATRE_TRUNK4 / CATRE_STN4 / CATRE_STN_PAIR = 0, CATRE_ROTW / CATRE_FC_TAIL = 1); calls in flight keep the form they were
 * launched with. */
int catre_form_switch(int id, int value);
/* Experiment knobs of the instrumented library (id 0: start offset in cycles between the co-resident workgroups of the
 * STN kernels' first dispatch round); CATRE_ERR_UNSUPPORTED in the product library. */
int catre_debug_knob(int id, int value);
/* Wait for the recorded events, write per-launch durations (ms) and reset the record counter. */
int catre_profile_collect(float* ms_out, int max_out, int* n_out);

/* Debug aid: when non-NULL, every k_trunk workgroup writes 8 waves x 8 shader-clock stamps (u64) at its phase
 * boundaries into `device_buffer` ([tiles][8 waves][8]); NULL disables.  Process-global. */
int catre_debug_trunk_trace(void* device_buffer);

/* Identity of the stream capture `stream` is currently recording into (hipStreamGetCaptureInfo; 0 when the stream is not
 * capturing).  The host mirror keys its "this capture already recorded a weight-pack node" shortcut on it
 * (catre_amd/runtime.py; the reference has no counterpart: torch packs nothing). */
int catre_stream_capture_id(void* stream, unsigned long long* id_out);

const char* catre_status_string(int status);

/* ---- training ops (forward with saved activations + backward), chained by torch.autograd ------------------
 * The training path runs the layers unfused on point-major activation matrices [rows, channels] in HBM; each
 * op has a hand-written forward and backward kernel.  Replaced reference arithmetic: torch.nn.functional
 * conv1d(k=1)/linear, relu, max over points, bmm with the STN transforms, group_norm, gelu, the conv_p
 * weighted sum and the autograd of rot6d_to_mat_batch / pose_scale_from_delta_init - i.e. what
 * `losses.backward()` (core/catre/engine/engine.py:349) differentiates through.  Row orders: "cloud-major" =
 * B*N observed rows then B*M prior rows; "object-major" = [N observed | M prior] per object. */
int catre_op_pack(const float* src, int ld, int J, int K, int transpose, float* dst, void* stream);
int catre_op_gemm_rows(const float* X, int ldx, const float* Wp, const float* bias, const float* mask, int ldm,
                       float* Y, int ldy, int R, int J, int K, int relu, void* stream);
/* linear + max-pool over the points of each cloud, fused (no [R,J] intermediate); N, M multiples of 64 */
size_t catre_op_linear_maxpool_ws_bytes(int R, int J);
int catre_op_linear_maxpool(const float* X, int ldx, const float* Wp, const float* bias, float* out, int* idx, int J,
                            int K, int B, int N, int M, void* ws, size_t ws_bytes, void* stream);
/* catre_op_gemm_rows with the left operand masked on load, (X .* (xmask > 0)) Wl^T: the ReLU backward folded in */
int catre_op_gemm_rows_m(const float* X, int ldx, const float* xmask, int ldxm, const float* Wp, const float* bias,
                         const float* mask, int ldm, float* Y, int ldy, int R, int J, int K, int relu, void* stream);
/* mixed precision (torch.autocast around the training forward, core/catre/engine/engine.py:304): the row GEMMs with
 * bf16 operands and fp32 accumulation / outputs.  Wp from catre_op_pack_bf16 (J*K bf16); K in {64,128,256,512}. */
int catre_op_pack_bf16(const float* src, int ld, int J, int K, int transpose, void* dst, void* stream);
int catre_op_gemm_rows_bf16(const float* X, int ldx, const float* xmask, int ldxm, const void* Wp, const float* bias,
                            const float* mask, int ldm, float* Y, int ldy, int R, int J, int K, int relu, void* stream);
int catre_op_linear_maxpool_bf16(const float* X, int ldx, const void* Wp, const float* bias, float* out, int* idx, int J,
                                 int K, int B, int N, int M, void* ws, size_t ws_bytes, void* stream);
/* split mode (hi + lo bf16 operands, three products: fp32-grade results on the bf16 pipe) of the three entry points
 * above; dst / Wp hold 2 * J * K bf16 (hi pack, then lo pack) */
int catre_op_pack_split(const float* src, int ld, int J, int K, int transpose, void* dst, void* stream);
int catre_op_gemm_rows_split(const float* X, int ldx, const float* xmask, int ldxm, const void* Wp, const float* bias,
                             const float* mask, int ldm, float* Y, int ldy, int R, int J, int K, int relu, void* stream);
int catre_op_linear_maxpool_split(const float* X, int ldx, const void* Wp, const float* bias, float* out, int* idx, int J,
                                  int K, int B, int N, int M, void* ws, size_t ws_bytes, void* stream);
/* Y = X Wl^T + bias2d[cloud(row)]: row GEMM with a per-cloud bias (rot-head layer 0: the global-feature half of the
 * 1088 -> 256 conv, conv_out_per_rot_head.py:126-128, is constant per cloud).  Rows object-major (N observed then M prior
 * rows per object), N and M multiples of 64, bias2d [2B][J]; Wp packed for compute_dtype (catre_op_pack / _bf16 / _split). */
int catre_op_gemm_rows_cloudbias(const float* X, int ldx, const void* Wp, const float* bias2d, float* Y, int ldy, int J,
                                 int K, int B, int N, int M, int compute_dtype, void* stream);
/* The rot-head linears (conv_out_per_rot_head.py:126-134) with two optional epilogue extras: per_cloud != 0 - bias is
 * [2B][J], indexed by the row's cloud; gn_part != NULL (J == 256) - the per-64-row-tile GroupNorm(32,256) partials
 * [B*(N+M)/64][32][2] of Y, consumed by catre_op_gnp_gelu_fwd_pre instead of a statistics pass over Y. */
int catre_op_gemm_rows_gn(const float* X, int ldx, const void* Wp, const float* bias, int per_cloud, float* Y, int ldy,
                          int J, int K, int B, int N, int M, float* gn_part, int compute_dtype, void* stream);
int catre_op_gnp_gelu_fwd_pre(const float* Y, const float* part64, const float* gamma, const float* beta, float* A,
                              float* stat, int B, int P, void* stream);
size_t catre_op_gemm_tn_ws_bytes(int J, int K, int R);
int catre_op_gemm_tn(const float* dY, int ldy, const float* X, int ldx, float* dW, int J, int K, int R,
                     int accumulate, void* ws, size_t ws_bytes, void* stream);
/* weight AND bias gradient of y = x W^T + b in one pass over dY: dW = dY^T X, db = column sums of dY (db may be NULL) */
size_t catre_op_gemm_tn_bias_ws_bytes(int J, int K, int R);
int catre_op_gemm_tn_bias(const float* dY, int ldy, const float* X, int ldx, float* dW, float* db, int J, int K, int R,
                          int accumulate, void* ws, size_t ws_bytes, void* stream);
int catre_op_gemm_tn_bias_m(const float* dY, int ldy, const float* ymask, int ldym, const float* X, int ldx, float* dW,
                            float* db, int J, int K, int R, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* ... with the products on the matrix pipe compute_dtype names (CATRE_DTYPE_F32 / _BF16: operands rounded to bf16, the
 * backward of a layer that ran under torch.autocast / _SPLIT: hi + lo bf16 operands, three products); fp32 accumulation
 * and an fp32 bias gradient in every case */
int catre_op_gemm_tn_bias_lp(const float* dY, int ldy, const float* ymask, int ldym, const float* X, int ldx, float* dW,
                             float* db, int J, int K, int R, int accumulate, void* ws, size_t ws_bytes,
                             int compute_dtype, void* stream);
/* whole backward of a 64-channel layer fed by <= 4 live input columns (conv1 on 3-d points; the reference's autograd of
 * pointnet.py:103 `F.relu(self.conv1(x))`) in one pass over the rows: dv = (dY (+ dY2)) .* (ymask > 0); dW[64,4] = dv^T X,
 * db[64] = column sums (db must be dW + 256), dX[R,dxcols] = dv W[64,Kw] (columns >= Kw zero, dxcols 4 or 8).  dY2, ymask
 * and dX optional; ws >= catre_op_gemm_tn_bias_ws_bytes(64, 4, R) */
int catre_op_skinny_bwd(const float* dY, int ldy, const float* dY2, int ldy2, const float* ymask, int ldym, const float* X,
                        int ldx, const float* W, int ldw, int Kw, float* dW, float* db, float* dX, int lddx, int dxcols,
                        int R, void* ws, size_t ws_bytes, void* stream);
/* whole backward of a linear layer on few rows (R < 2048: the reference's autograd of the F.linear / nn.Linear layers whose
 * rows are clouds or objects - pointnet.py:31-33,64-66 fc1-fc3, fc_trans_size_head.py:61-70, the global half of
 * conv_out_per_rot_head.py:126) in ONE launch: dv = dY .* (YM > 0) (YM: the layer's output behind a ReLU, laid out like dY,
 * or NULL); dX[R,Kx] = dv W[J,Kw] (columns >= Kw zero), dW[J,Kw] = dv^T X[R,Kx] (columns >= Kx zero), db[J] = column sums
 * of dv.  dX, dW, db optional (db needs dW) and contiguous; any J, Kx, Kw, leading dimensions.  compute_dtype as
 * catre_op_gemm_tn_bias_lp for dW (dX in fp32).  Deterministic, no workspace. */
int catre_op_fc_bwd(const float* dY, int ldy, const float* YM, const float* X, int ldx, const float* W, int ldw, float* dX,
                    float* dW, float* db, int R, int J, int Kx, int Kw, int compute_dtype, void* stream);
int catre_op_colsum(const float* dY, int ld, int R, int J, float* out, int accumulate, void* ws, size_t ws_bytes,
                    void* stream);
int catre_op_reduce_splits(const float* part, float* out, int n, int splits, int accumulate, void* stream);
int catre_op_rowbias_add(float* Y, int ld, const float* bias, int J, int B, int N, int M, void* stream);
int catre_op_rowbias_bwd(const float* dY, int ld, float* dbias, int J, int B, int N, int M, void* stream);
int catre_op_maxpool_fwd(const float* Y, int ld, float* out, int* idx, int J, int B, int N, int M, void* stream);
int catre_op_maxpool_scatter(const float* dout, const int* idx, float* dY, int ld, int C, int J, void* stream);
int catre_op_maxlin_bwd_w(const float* dg, const int* idx, const float* X, int ldx, float* dW, float* db, int C,
                          int J, int K, void* stream);
int catre_op_maxlin_bwd_x(const float* dg, const int* idx, const float* W, int ldw, float* dX, int ldx, int C, int J,
                          int K, void* stream);
/* the same gradient written row by row (every row of dX[:, :K] exactly once, zeros included: no memset, no
 * read-modify-write); points per cloud <= 4096 */
int catre_op_maxlin_bwd_x_rows(const float* dg, const int* idx, const float* W, int ldw, float* dX, int ldx, int J, int K,
                               int B, int N, int M, void* stream);
int catre_op_cloud_matmul(const float* X, int ldx, const float* T, float* Y, int ldy, int kd, int B, int N, int M,
                          int transpose, void* stream);
int catre_op_cloud_matmul_bwd_t(const float* X, int ldx, const float* dY, int ldy, float* dT, int kd, int B, int N,
                                int M, void* stream);
/* out[R,K] = a + b + (c on its first Rc rows): a, b [R,K] contiguous, c [Rc,K] with leading dimension ldc, each optional
 * (NULL = zeros) - the
 * gradient of a tensor whose consumers include a row slice (the pooled feature: CATRE_disR_shared.py:69 feeds `[:B]` of it
 * to the ts head and all of it to both rotation heads), which autograd assembles from a zero-fill, a copy and an add per
 * further consumer */
int catre_op_sum_rows(const float* a, const float* b, const float* c, int ldc, float* out, int R, int Rc, int K,
                      void* stream);
int catre_op_relu_bwd(const float* dY, const float* Y, float* dX, size_t n, void* stream);
int catre_op_gnp_gelu_fwd(const float* Y, const float* gamma, const float* beta, float* A, float* stat, int B, int P,
                          void* stream);
int catre_op_gnp_gelu_bwd(const float* dA, const float* Y, const float* stat, const float* gamma, const float* beta,
                          float* dY, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, int B,
                          int P, void* stream);
/* GroupNorm + GELU + neck Conv1d(256 -> rot_dim <= 3) of a RotHead in one op (conv_out_per_rot_head.py:132-137): the [R,256]
 * activation between them is never stored.  Wn [3][256] (rows >= rot_dim zero), bn [3] or NULL, Y3 / dY3 [B*P][3],
 * part64 as catre_op_gnp_gelu_fwd_pre (P % 64 == 0).  Backward: dY [B*P,256], dparams [5][256] = dgamma, dbeta, dWn. */
int catre_op_gnp_gelu_neck_fwd(const float* Y, const float* part64, const float* gamma, const float* beta,
                               const float* Wn, const float* bn, float* Y3, float* stat, int B, int P, void* stream);
size_t catre_op_gnp_gelu_neck_bwd_ws_bytes(int B, int P);
int catre_op_gnp_gelu_neck_bwd(const float* dY3, const float* Y, const float* stat, const float* gamma,
                               const float* beta, const float* Wn, float* dY, float* dparams, int accumulate, void* ws,
                               size_t ws_bytes, int B, int P, void* stream);
/* Backward of a RotHead's first block - Conv1d(1088 -> 256) on cat(point feature, global feature) = a 64 -> 256 linear with
 * a per-cloud bias, GroupNorm(32,256), GELU (conv_out_per_rot_head.py:126-131) - from dA [R,256] in two passes over (dA, Y):
 * the [R,256] gradient of the linear's output stays in LDS.  X [R,64], W [256][64], rows object-major, N, M % 64 == 0.
 * accumulate_dx bit 0: dX += (the two RotHeads share X; the second head's call adds into the first one's result);
 * bit 1: the rows of X are cloud-major (B*N observed rows, then B*M prior rows - pointfeat as the encoder writes it), dX
 * stays object-major. */
size_t catre_op_rot_l0_bwd_ws_bytes(int B, int N, int M);
int catre_op_rot_l0_bwd(const float* dA, const float* Y, const float* stat, const float* gamma, const float* beta,
                        const float* X, int ldx, const float* W, float* dX, int lddx, float* dW, float* dbias2d,
                        float* dgamma, float* dbeta, int accumulate_dx, void* ws, size_t ws_bytes, int B, int N, int M,
                        void* stream);
/* ... with dX = dY W and dW = dY^T X on the bf16 matrix pipe (bf16 operands, fp32 accumulation; everything else fp32): the
 * backward torch.autocast gives the bf16 Conv1d (engine.py:304).  Same arguments, same workspace. */
int catre_op_rot_l0_bwd_lp(const float* dA, const float* Y, const float* stat, const float* gamma, const float* beta,
                           const float* X, int ldx, const float* W, float* dX, int lddx, float* dW, float* dbias2d,
                           float* dgamma, float* dbeta, int accumulate_dx, void* ws, size_t ws_bytes, int B, int N, int M,
                           void* stream);
/* ... and in split mode: hi + lo bf16 operands, three products - fp32-grade dX and dW (k_rot_l0_bwd_sp). */
int catre_op_rot_l0_bwd_sp(const float* dA, const float* Y, const float* stat, const float* gamma, const float* beta,
                           const float* X, int ldx, const float* W, float* dX, int lddx, float* dW, float* dbias2d,
                           float* dgamma, float* dbeta, int accumulate_dx, void* ws, size_t ws_bytes, int B, int N, int M,
                           void* stream);
/* Backward of a RotHead's second block - Conv1d(256 -> 256), GroupNorm, GELU, neck (conv_out_per_rot_head.py:129-137) - from
 * dY3 [R,3]: the sums pass, then one pass over (Y, A) that keeps the linear's output gradient in LDS.  A [R,256] the block's
 * input, W [256][256]; dA [R,256], dWb [256*256 + 256] = dW then db, dparams [5][256] = dgamma, dbeta, dWn.  P % 64 == 0. */
size_t catre_op_rot_l1_bwd_ws_bytes(int B, int P);
int catre_op_rot_l1_bwd(const float* dY3, const float* Y, const float* stat, const float* gamma, const float* beta,
                        const float* Wn, const float* A, const float* W, float* dA, float* dWb, float* dparams, void* ws,
                        size_t ws_bytes, int B, int P, void* stream);
/* The pair used when the neck output feeds conv_p only (out[b][j] = sum_p wp[p] Y3[b,p,j] + b, conv_out_per_rot_head.py:
 * 138-140, so dY3[b,p,:] = wp[p] dout[b,:]): the forward also leaves per-tile moments Spart [B*P/64][3][256]
 * (sum_p wp gelu', sum_p wp gelu' xhat, sum_p wp gelu per channel); the backward takes the GroupNorm-1 sums and dgamma /
 * dbeta / dWn from them and dout [B][3] instead of a reduction pass over Y.  dY3 must be catre_op_wsum_bwd's dY. */
int catre_op_gnp_gelu_neck_fwd_s(const float* Y, const float* part64, const float* gamma, const float* beta,
                                 const float* Wn, const float* bn, const float* wp, float* Y3, float* stat, float* Spart,
                                 int B, int P, void* stream);
int catre_op_rot_l1_bwd_s(const float* dY3, const float* dout, const float* Spart, const float* Y, const float* stat,
                          const float* gamma, const float* beta, const float* Wn, const float* A, const float* W, float* dA,
                          float* dWb, float* dparams, void* ws, size_t ws_bytes, int B, int P, void* stream);
/* catre_op_rot_l1_bwd / _s with the two GEMMs (dA = dY W, dW = dY^T A) on the bf16 matrix pipe: bf16 operands, fp32
 * accumulation - the backward torch.autocast gives a bf16 Conv1d (engine.py:304); GroupNorm / GELU', the bias gradient and all
 * outputs stay fp32.  dout and Spart: both (sums from the forward's moments) or both NULL (sums pass over Y). */
int catre_op_rot_l1_bwd_lp(const float* dY3, const float* dout, const float* Spart, const float* Y, const float* stat,
                           const float* gamma, const float* beta, const float* Wn, const float* A, const float* W, float* dA,
                           float* dWb, float* dparams, void* ws, size_t ws_bytes, int B, int P, void* stream);
/* ... and in split mode: hi + lo bf16 operands, three products per block - fp32-grade dA and dW on the bf16 pipe
 * (k_rot_l1_bwd_sp, 32-row half tiles). */
int catre_op_rot_l1_bwd_sp(const float* dY3, const float* dout, const float* Spart, const float* Y, const float* stat,
                           const float* gamma, const float* beta, const float* Wn, const float* A, const float* W, float* dA,
                           float* dWb, float* dparams, void* ws, size_t ws_bytes, int B, int P, void* stream);
/* The all-bf16-activation form of the autocast rotation heads: the [rows,256] tensors that travel between these kernels -
 * y0, a0, y1, dA - are bf16 rows (256 bf16 per row; `void*`), what torch.autocast's Conv1d outputs are (engine.py:304), so
 * every pass over them moves half the bytes; statistics, GroupNorm / GELU arithmetic, accumulation and every other output
 * stay fp32.  catre_op_gemm_rows_gn_h: catre_op_gemm_rows_gn on bf16 operands (Wp: catre_op_pack_bf16) with io bit 0: X is
 * bf16 rows, bit 1: Y is bf16 rows (ldx / ldy in elements); J = 256, K in {64, 256}; per_cloud == 2 (fp32 X only): per-cloud bias
 * and X rows in CLOUD-major order (pointfeat where the trunk kernel wrote it: no object-major copy), Y object-major.  The others: the op of the same name
 * without _h / with _lp, with the named tensors as bf16 rows (catre_op_rot_l1_bwd_h: Y, A, dA; catre_op_rot_l0_bwd_h: dA, Y). */
int catre_op_gemm_rows_gn_h(const void* X, int ldx, const void* Wp, const float* bias, int per_cloud, void* Y, int ldy, int J,
                            int K, int B, int N, int M, float* gn_part, int io, void* stream);
int catre_op_gnp_gelu_fwd_pre_h(const void* Y, const float* part64, const float* gamma, const float* beta, void* A,
                                float* stat, int B, int P, void* stream);
/* catre_op_gnp_gelu_fwd_pre_h followed by catre_op_gemm_rows_gn_h (K = 256, bf16 rows in and out) as one kernel behind the
 * statistics merge: the GroupNorm + GELU is applied while the GEMM stages its operand tile; A (bf16 rows) is still written
 * for the backward.  Same values as the two calls. */
int catre_op_gn_gelu_gemm_rows_h(const void* Y0, const float* part64, const float* gamma, const float* beta, void* A,
                                 float* stat, const void* Wp, const float* bias, void* Y1, float* gn_part, int B, int N, int M,
                                 void* stream);
int catre_op_gnp_gelu_neck_fwd_s_h(const void* Y, const float* part64, const float* gamma, const float* beta, const float* Wn,
                                   const float* bn, const float* wp, float* Y3, float* stat, float* Spart, int B, int P,
                                   void* stream);
int catre_op_rot_l1_bwd_h(const float* dY3, const float* dout, const float* Spart, const void* Y, const float* stat,
                          const float* gamma, const float* beta, const float* Wn, const void* A, const float* W, void* dA,
                          float* dWb, float* dparams, void* ws, size_t ws_bytes, int B, int P, void* stream);
int catre_op_rot_l0_bwd_h(const void* dA, const void* Y, const float* stat, const float* gamma, const float* beta,
                          const float* X, int ldx, const float* W, float* dX, int lddx, float* dW, float* dbias2d,
                          float* dgamma, float* dbeta, int accumulate_dx, void* ws, size_t ws_bytes, int B, int N, int M,
                          void* stream);
/* ... and the per-head form for the modes whose linear backward is not fused (autocast, split): sums and dparams from dout
 * and Spart, then the apply pass -> dY [B*P,256] (what catre_op_gnp_gelu_neck_bwd returns, without its reduction pass over Y).
 * ws as catre_op_gnp_gelu_neck_bwd_ws_bytes. */
int catre_op_gnp_gelu_neck_bwd_s(const float* dY3, const float* dout, const float* Spart, const float* Y, const float* stat,
                                 const float* gamma, const float* beta, const float* Wn, float* dY, float* dparams, void* ws,
                                 size_t ws_bytes, int B, int P, void* stream);
/* dst [rows][cols_pad] (contiguous) = src [rows][cols] (element strides: transposed views too) followed by zero columns:
 * the padding of small operands to the GEMM kernels' granularity in one launch (F.pad of the reference-side glue). */
int catre_op_pad_cols(const float* src, long stride_row, long stride_col, int rows, int cols, float* dst, int cols_pad,
                      void* stream);
int catre_op_gnr_gelu_fwd(const float* Y, const float* gamma, const float* beta, float* A, int R, void* stream);
int catre_op_gnr_gelu_bwd(const float* dA, const float* Y, const float* gamma, const float* beta, float* dY,
                          float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, int R, void* stream);
int catre_op_wsum_fwd(const float* Y, const float* w, const float* bias, float* out, int B, int P, void* stream);
int catre_op_wsum_bwd(const float* dout, const float* Y, const float* w, float* dY, float* dw, float* dbias,
                      int accumulate, void* ws, size_t ws_bytes, int B, int P, void* stream);
/* ... plus dbn[3] = column sums of dY - the bias gradient of the 256 -> 3 neck whose output gradient dY is
 * (conv_out_per_rot_head.py:131-140: neck, then conv_p) - as (sum_p w[p]) (sum_b dout[b][:]), in the launch that sums dbias */
int catre_op_wsum_bwd_n(const float* dout, const float* Y, const float* w, float* dY, float* dw, float* dbias, float* dbn,
                        int accumulate, void* ws, size_t ws_bytes, int B, int P, void* stream);
int catre_op_pose_update_bwd(const float* d_pose, const float* d_scale, const float* rot6d, const float* trans_deltas,
                             const float* scale_deltas, const float* init_pose, const float* init_scale,
                             const float* mean_scales, const float* Ks, const catre_opts* opts, float* d_rot6d,
                             float* d_dt, float* d_ds, int B, void* stream);

/* Row-sparse backward of linear + max-pool chains (the backward of core/catre/models/pointnets/pointnet.py:24-28, 57-61,
 * 112-116 as torch.autograd runs it, minus the zeros): only the arg-max row of a (cloud, channel) carries gradient, so the
 * gradient in front of a pool - and of every layer further up the conv stack until a dense side input joins - is zero on
 * every row that is nobody's arg-max (~70 % of the rows at N = M = 1024).  catre_op_rows_compact builds the ascending list of
 * live rows, the dense -> compact map and the live-row COUNT on the device (no host sync); catre_op_maxlin_bwd_x_compact
 * writes the pool's data gradient for the live rows only; catre_op_gemm_rows_n / catre_op_gemm_tn_bias_n are the row GEMMs
 * of catre_op_gemm_rows_m / catre_op_gemm_tn_bias_m on the first nrows_dev[0] rows; catre_op_gather_rows /
 * catre_op_scatter_rows move rows between the dense and the compact order.  rows, rowpos: [R] int32; count: [1] int32;
 * scratch: [2 * clouds] int32; clouds of at most 4096 points.  compute_dtype selects the matrix pipe of the two GEMMs
 * (CATRE_DTYPE_F32 / _BF16 / _SPLIT; Wp = the weight pack of that dtype: catre_op_pack / _pack_bf16 / _pack_split). */
int catre_op_rows_compact(const float* dg, const int32_t* idx, int J, int B, int N, int M, int32_t* rows, int32_t* rowpos,
                          int32_t* count, int32_t* scratch, void* stream);
int catre_op_maxlin_bwd_x_compact(const float* dg, const int32_t* idx, const float* W, int ldw, const int32_t* rowpos,
                                  const float* ymask, int ldym, float* dXc, int ldx, int J, int K, int B, int N, int M,
                                  void* stream);
/* catre_op_maxlin_bwd_x_compact / catre_op_maxlin_bwd_w with the saved activation - the ReLU mask / the gathered operand -
 * as bf16 rows (leading dimension in elements): the backward of the pooled layer behind the autocast encoder forward */
int catre_op_maxlin_bwd_x_compact_h(const float* dg, const int32_t* idx, const float* W, int ldw, const int32_t* rowpos,
                                    const void* ymask, int ldym, float* dXc, int ldx, int J, int K, int B, int N, int M,
                                    void* stream);
int catre_op_maxlin_bwd_w_h(const float* dg, const int32_t* idx, const void* X, int ldx, float* dW, float* db, int C, int J,
                            int K, void* stream);
/* Recompute instead of save (the STN stacks' row-sparse backward, fp32): catre_op_stn_recompute rebuilds
 * y1 = relu(conv1 x) [cap,64] and y2 = relu(conv2 y1) [cap,128] for the count[0] live rows as COMPACT rows with the forward
 * kernels' own device code (same bits), so catre_train_stn3d_fwd / _stnkd_fwd may be called without row buffers.
 * kind 0: STN3d (pointnet.py:24-26) - X point rows [R][ldx >= 3], w1 = conv1.weight [64,3], wp1 NULL; kind 1: STNkd
 * (:57-59) - X = relu(conv1) rows [R][64] as the trunk kernel saves them, wp1 = catre_op_pack of fstn.conv1.weight, w1 NULL.
 * wp2 = catre_op_pack of conv2.weight [128,64].  catre_op_maxlin_bwd_w_c / catre_op_maxlin_bwd_x_compact_cm are
 * catre_op_maxlin_bwd_w / catre_op_maxlin_bwd_x_compact reading those compact rows (dense row r at rowpos[r]). */
int catre_op_stn_recompute(int kind, const float* X, int ldx, const int32_t* rows, const int32_t* count, const float* w1,
                           const float* wp1, const float* b1, const float* wp2, const float* b2, float* y1c, float* y2c,
                           int cap, void* stream);
int catre_op_maxlin_bwd_w_c(const float* dg, const int32_t* idx, const int32_t* rowpos, const float* Xc, int ldx, float* dW,
                            float* db, int C, int J, int K, void* stream);
int catre_op_maxlin_bwd_x_compact_cm(const float* dg, const int32_t* idx, const float* W, int ldw, const int32_t* rowpos,
                                     const float* ymask_c, int ldym, float* dXc, int ldx, int J, int K, int B, int N, int M,
                                     void* stream);
int catre_op_gather_rows(const float* src, int lds, const int32_t* rows, const int32_t* count, float* dst, int ldd, int cols,
                         int cap, void* stream);
int catre_op_scatter_rows(const float* srcc, int lds, const int32_t* rowpos, float* dst, int ldd, int cols, int R,
                          void* stream);
/* catre_op_scatter_rows plus objsrc[object-major row of r] (the rows of B objects, [N observed | M prior] each - the order
 * the rotation heads work in, CATRE_disR_shared.py:86) and then dst[idx_max[c][j]][j] += dmax[c][j] (the backward of
 * max over points, :69): the three gradients pointfeat receives summed in two launches.  objsrc / dmax may be NULL. */
int catre_op_scatter_rows_merge(const float* srcc, int lds, const int32_t* rowpos, const float* objsrc, int ldo,
                                const float* dmax, const int32_t* idx_max, int Jm, float* dst, int ldd, int cols, int B, int N,
                                int M, void* stream);
int catre_op_gemm_rows_n(const float* X, int ldx, const float* xmask, int ldxm, const void* Wp, const float* bias,
                         const float* mask, int ldm, float* Y, int ldy, int R, int J, int K, int relu,
                         const int32_t* nrows_dev, int compute_dtype, void* stream);
int catre_op_gemm_tn_bias_n(const float* dY, int ldy, const float* ymask, int ldym, const float* X, int ldx, float* dW,
                            float* db, int J, int K, int R, int accumulate, void* ws, size_t ws_bytes,
                            const int32_t* nrows_dev, int compute_dtype, void* stream);
/* the two above with row indirection against DENSE tensors (no gathered copies of the saved activations in the row-sparse
 * backward): output row r of catre_op_gemm_rows_nr is masked with row mask_rows[r] of `mask`; catre_op_gemm_tn_bias_nr
 * contracts dY row r with row x_rows[r] of X.  Every compute_dtype; a null index array is the identity.
 * compute_dtype = CATRE_DTYPE_BF16 | CATRE_ROWS_BF16: `mask` / `X` are bf16 rows (ldm / ldx in elements) */
int catre_op_gemm_rows_nr(const float* X, int ldx, const float* xmask, int ldxm, const void* Wp, const float* bias,
                          const float* mask, int ldm, const int32_t* mask_rows, float* Y, int ldy, int R, int J, int K,
                          int relu, const int32_t* nrows_dev, int compute_dtype, void* stream);
int catre_op_gemm_tn_bias_nr(const float* dY, int ldy, const float* ymask, int ldym, const float* X, int ldx,
                             const int32_t* x_rows, float* dW, float* db, int J, int K, int R, int accumulate, void* ws,
                             size_t ws_bytes, const int32_t* nrows_dev, int compute_dtype, void* stream);

/* Training forward of the three encoder blocks on the FUSED kernels (the inference kernels with extra stores): the
 * pooled feature g [2B,1024] (bias added, no activation) with its arg-max rows idx [2B,1024], plus the activations the
 * layer-wise backward ops above read, as cloud-major point rows (B*N observed rows, then B*M prior rows):
 *   stn3d: a1 = relu(stn.conv1) [R,64], a2 = relu(stn.conv2) [R,128]            (pointnet.py:24-28)
 *   stnkd: f1 = relu(fstn.conv1) [R,64], f2 = relu(fstn.conv2) [R,128]           (pointnet.py:57-61)
 *   trunk: x1 = x T3 [R,8] (zero-padded), h1 = relu(conv1) [R,64], pf = h1 T64 [R,64], a2 = relu(conv2) [R,128],
 *          a3 = relu(conv3) [R,512]                                               (pointnet.py:98-116)
 * N and M multiples of 64; `workspace` as catre_workspace_bytes.  compute_dtype = CATRE_DTYPE_F32 (fp32 packs,
 * CATRE_PACK_F32_ENCODER, in `packed`) or CATRE_DTYPE_BF16 (what torch.autocast selects, engine.py:304: the bf16-operand
 * kernels, CATRE_PACK_BF16 packs; a1 / a2, f1 / f2 and the trunk's a2 / a3 are then bf16 ROWS - [R,width] bf16, half the
 * bytes, the values the reduced-precision dgrad / wgrad ops round their operands to anyway - which the row-sparse
 * backward reads in place (catre_op_maxlin_bwd_w_h / _x_compact_h, CATRE_ROWS_BF16); x1, h1 and pf stay fp32 rows holding
 * bf16-rounded values; trans64 required) or CATRE_DTYPE_SPLIT (the
 * split-bf16 kernels, CATRE_PACK_SPLIT | CATRE_PACK_F32_ENCODER packs - conv2 of the trunk stays an fp32 MFMA layer; the saved rows
 * hold hi + lo; trans64 required).
 * catre_train_stn3d_fwd / _stnkd_fwd, CATRE_DTYPE_F32 only: a1 = a2 = NULL (f1 = f2 = NULL) stores no activation rows - the
 * backward recomputes them on its live rows (catre_op_stn_recompute) - and full grids then run on the one-wave pair kernels
 * of the inference path with an arg-max epilogue. */
int catre_train_stn3d_fwd(const catre_points* pts, const float* const* params, const float* packed, float* a1, float* a2,
                          float* g, int32_t* idx, void* workspace, size_t ws_bytes, int B, int N, int M, int compute_dtype,
                          void* stream);
int catre_train_stnkd_fwd(const catre_points* pts, const float* trans3, const float* const* params, const float* packed,
                          float* f1, float* f2, float* g, int32_t* idx, void* workspace, size_t ws_bytes, int B, int N,
                          int M, int compute_dtype, void* stream);
int catre_train_trunk_fwd(const catre_points* pts, const float* trans3, const float* trans64, const float* const* params,
                          const float* packed, float* x1, float* h1, float* pf, float* a2, float* a3, float* g,
                          int32_t* idx, void* workspace, size_t ws_bytes, int B, int N, int M, int compute_dtype,
                          void* stream);

/* Training forward of both rotation heads up to the GroupNorm-1 input (heads/conv_out_per_rot_head.py:126-134) on the fused
 * inference kernels with saves: GN0 statistics from second moments of pointfeat, then layer 0 + GN0 + GELU + layer 1 per
 * 64-point tile.  pointfeat: cloud-major [B*N + B*M][64]; bias0 [2 heads][2B clouds][256] = W0[:, :1024] g_cloud + b0;
 * prm / packed: the parameter pointer array and a weight image holding CATRE_PACK_F32_HEADS (compute_dtype CATRE_DTYPE_F32)
 * or CATRE_PACK_SPLIT (CATRE_DTYPE_SPLIT: the two linears as split-bf16 MFMAs, everything stored is fp32).  Outputs, head-major:
 * y0, a0, y1 [2][B*(N+M)][256] (rows object-major), gn1_part [2][B*(N+M)/64][32][2] (what catre_op_gnp_gelu_neck_fwd takes),
 * stat0 [2][B][32][2] (mean, rstd).  N and M multiples of 64.  The per-head backward ops (catre_op_rot_l1_bwd,
 * catre_op_rot_l0_bwd) consume the head slices unchanged. */
size_t catre_train_rot_fwd_ws_bytes(int B);
int catre_train_rot_fwd(const float* pointfeat, const float* bias0, const float* const* params, const float* packed, float* y0,
                        float* a0, float* y1, float* gn1_part, float* stat0, void* workspace, size_t ws_bytes, int B, int N,
                        int M, int compute_dtype, void* stream);

/* f4: one fused multi-tensor Ranger step = RAdam + Lookahead + gradient centralization
 * (lib/torch_utils/solver/ranger.py:102-202) with the train loop's grad nan_to_num folded in
 * (core/catre/engine/engine.py:351-353).  `tensors`: device array of n_tensors 72-byte records
 * {float* p; const float* g; float* exp_avg; float* exp_avg_sq; float* slow; int numel, row_len, row_off;
 *  float lr_step, wd_lr; int adaptive, lookahead, pad;}; `chunks`: device array of {int tensor, offset} pairs
 * (4096 elements each); `row_tensor[total_rows]`: tensor index of every centralized gradient row.
 * beta1 / beta2 are doubles: the reference forms `1 - beta` in Python double precision before torch rounds it to the
 * tensors' fp32 (ranger.py:141-143), so the decay and the (1 - beta) weights are rounded separately here too. */
int catre_op_ranger_step(const void* tensors, int n_tensors, const void* chunks, int n_chunks, const int* row_tensor,
                         int total_rows, float* rowmean_ws, double beta1, double beta2, float eps, float alpha,
                         int clean_grads, float grad_limit, void* stream);

/* Build identification: "catre_hip gfx950 <version>" */
const char* catre_version(void);

/* ---- train-time batch glue on the device ----------------------------------------------- */

/* aug_3d_bbox followed by aug_RT (core/catre/engine/engine_utils.py:107-172; called from batch_data,
 * core/catre/engine/batching.py:84-88) over all B*N points in one launch:
 *   p' = dR (R (ratios .* R^T (p - t)) + t + dt),  pose' = [dR R | dR (t + dt)],  scale' = scale .* ratios.
 * bbox_ratios = HOST pointer to (ex, ey, ez) or NULL (no bbox aug); objects with sym_flags[b] != 0 use
 * ((ex+ez)/2, ey, (ex+ez)/2) (:126-128).  delta_r[9] / delta_t[3] = HOST pointers (get_rotation_torch output and
 * the translation shift) or both NULL.  pcl, pose, scale, sym_flags and the outputs are device pointers;
 * pcl_out may alias pcl; pose_out / scale_out must NOT alias pose / scale (other threads still read them). */
int catre_aug_points(const float* pcl, const float* pose, const float* scale, const int32_t* sym_flags,
                     const float* bbox_ratios, const float* delta_r, const float* delta_t, float* pcl_out,
                     float* pose_out, float* scale_out, int B, int N, void* stream);

/* aug_poses_normal (core/utils/pose_aug.py:59-101) and aug_scale_normal (:10-35) given the normal noise the caller
 * drew: euler_deg [B,3] ~ N(0, std_rot) (clamped to +-max_rot_deg here unless max_rot_deg < 0), trans_noise [B,3],
 * scale_noise [B,3].  Either half may be skipped with pose == NULL / scale == NULL. */
int catre_init_noise(const float* pose, const float* euler_deg, const float* trans_noise, float max_rot_deg,
                     float min_z, float* pose_out, const float* scale, const float* scale_noise, float min_s,
                     float max_s, float* scale_out, int B, void* stream);

/* ---- point-cloud preparation (the step before the path) -------------------------------- */

/* Candidate pixels of I instances of one depth map, all instances at once - replaces, per instance,
 * backproject_th (lib/pysixd/misc.py:360-378) + sample_bp_depth (core/utils/cat_data_utils.py:209-226) + the
 * radius search of crop_ball_from_pts (:289-304) as called from crop_ball_from_depth_image (:380-400) by the data
 * loader (core/catre/datasets/data_loader.py:576-603):
 *   valid = mask & (depth > 0);  radius = max(ratio * ||R s||, 0.05) grown x1.1 (at most 9 times) until >= 10 valid
 *   points lie within it of the pose centre; no point at all -> every valid pixel (the `distance <= 1e9` fallback).
 * use_ball = 0 keeps every valid pixel (crop_mask_depth_image, :352-377).  depth [H,W] fp32 metres (device),
 * K9 = HOST 3x3 intrinsics row-major, masks [I,H,W] bytes (device) or NULL, poses [I,3,4], scales [I,3] (device).
 * The ordered (row-major, = torch.nonzero order) candidate lists stay in `workspace`
 * (catre_pcl_workspace_bytes); counts_out [I] (device, optional) receives their lengths. */
size_t catre_pcl_workspace_bytes(int I, int H, int W);
int catre_pcl_candidates(const float* depth, const float* K9, const unsigned char* masks, const float* poses,
                         const float* scales, float ratio, int use_ball, int I, int H, int W, void* workspace,
                         size_t ws_bytes, int32_t* counts_out, void* stream);

/* N points per instance out of the candidate lists: the tail of crop_ball_from_pts (:305-320) - the list is tiled
 * by doubling to a length L >= N, slot i takes element sample_idx[inst][i] of it (sample_idx [I,N] int64 on the
 * device = the caller's torch.randperm(L)[:N], which reproduces the reference's random stream), or, with
 * sample_idx == NULL, element perm_seed(i) of a keyed pseudo-random permutation of [0,L) evaluated on the device
 * (no host round trip).  pcl_out [I,N,3]; pix_out [I,N] (optional) = the flat pixel index of every sample, for
 * gathering rgb / NOCS maps.  Instances without any candidate produce zeros and pix -1. */
int catre_pcl_sample(const float* depth, const float* K9, const void* workspace, size_t ws_bytes,
                     const long long* sample_idx, unsigned long long seed, int I, int H, int W, int N, float* pcl_out,
                     int32_t* pix_out, void* stream);
/* INPUT.FPS_SAMPLE (crop_ball_from_pts(..., fps_sample=True), core/utils/cat_data_utils.py:305-306 ->
 * core/utils/farthest_points_torch.py:6-62, init_center=True, pairwise_distance): farthest-point order of each
 * instance's tiled candidate list -> sample_idx_out [I][N] for catre_pcl_sample.  scratch: I * 4 * slot_cap floats
 * with slot_cap >= the largest tiled list (count doubled until >= N). */
int catre_pcl_fps(const float* depth, const float* K9, const void* workspace, size_t ws_bytes, int I, int H, int W, int N,
                  float* scratch, int slot_cap, long long* sample_idx_out, void* stream);

/* ---- the training loss on the device --------------------------------------------------- */

/* Flags of cfg.MODEL.CATRE.LOSS_CFG that CATRE_disR_shared.catre_loss reads
 * (core/catre/models/CATRE_disR_shared.py:168-288; PyPMLoss core/catre/losses/pm_loss.py:85-194, L1 / R-only form). */
typedef struct catre_loss_cfg {
  int32_t pm_on, pm_sym, pm_with_scale;     /* PM_LW > 0, PM_LOSS_SYM, PM_WITH_SCALE                              */
  int32_t rot_on, rot_l2, yaxis_smooth;     /* ROT_LW > 0, ROT_LOSS_TYPE == "L2" (else angular), ROT_YAXIS_LOSS_TYPE: 0 "L1",
                                             * 1 "smoothL1", 2 "L2", 3 "angular" (CATRE_disR_shared.py:232-243)           */
  int32_t trans_on, trans_mse, trans_split; /* TRANS_LW > 0, TRANS_LOSS_TYPE: 0 "L1", 1 "MSE", 2 "L2" (L2Loss,
                                             * core/catre/losses/l2_loss.py:5-28), TRANS_LOSS_DISENTANGLE                 */
  int32_t scale_on, scale_mse;              /* SCALE_LW > 0, SCALE_LOSS_TYPE: 0 "L1", 1 "MSE", 2 "L2"                      */
  float pm_lw, rot_lw, trans_lw, scale_lw;
} catre_loss_cfg;

/* losses[20]: [0..6) = {loss_PM_R, loss_rot, loss_yaxis_rot, loss_trans_xy (or loss_trans_LPnP), loss_trans_z,
 * loss_scale}; [6..20) = the scalars CATRE_disR_shared.forward logs per training iteration
 * (core/catre/models/CATRE_disR_shared.py:127-164, compute_mean_re_te models/model_utils.py:226-238), in its order:
 * error_R [deg], error_t [cm], |t_pred - t_gt| x,y,z of object 0 [cm], t_pred x,y,z, trans_deltas x,y,z
 * (trans_deltas [B,3] may be NULL -> 0), t_gt x,y,z - computed with the loss, so logging costs no launch and one copy.
 * pose [B,3,4] = [R|t] estimate, scale [B,3]; cands [B,S1,3,3] = symmetry rotations per object with the identity
 * first, valid [B,S1] bytes, is_sym [B]; the ground-truth rotation closest to the estimate among R_gt S_k
 * (get_closest_rot_batch, core/utils/pose_utils.py:472-528) is chosen on the device, its index kept in best [B] for
 * the backward, and counts [2] = {objects with, without symmetry info} (counted on the device, so nothing about the
 * batch composition is baked into a captured graph).  part_ws: B*8 floats of scratch.  All pointers are device pointers. */
int catre_loss_fwd(const float* pose, const float* scale, const float* gt_rot, const float* gt_trans,
                   const float* gt_scale, const float* kps, const float* cands, const unsigned char* valid,
                   const int32_t* is_sym, const catre_loss_cfg* cfg, int32_t* best, int32_t* counts, float* part_ws,
                   float* losses, const float* trans_deltas, int B, int M, int S1, void* stream);
/* dpose [B,3,4], dscale [B,3] = gradient of sum_i upstream[i] * losses[i] (upstream: 6 floats on the device) */
int catre_loss_bwd(const float* pose, const float* scale, const float* gt_rot, const float* gt_trans,
                   const float* gt_scale, const float* kps, const float* cands, const int32_t* is_sym,
                   const int32_t* best, const int32_t* counts, const float* upstream, const catre_loss_cfg* cfg,
                   float* dpose, float* dscale, int B, int M, int S1, void* stream);
/* The reference's train loop adds the loss dict up with python's sum() (core/catre/engine/engine.py:318
 * `losses = sum(loss_dict.values())`): one add kernel per term, and autograd's per-term bookkeeping on the way back.
 * catre_loss_fwd_sums also writes prefix[k] = ((0 + losses[terms[0]]) + losses[terms[1]]) + ... + losses[terms[k]] for
 * k < n_terms <= 6 (terms: HOST array of loss indices in the dict's order) - every intermediate that sum() builds, same
 * operations, same bits - and catre_loss_bwd_sums takes the upstream gradients of those prefix sums (up_prefix: HOST array
 * of n_terms device pointers to one float each, NULL entries = zero) next to the six per-loss ones (upstream [6], device);
 * either argument may be NULL (= zeros).  catre_amd/losses.py hands the dict's
 * values out as tensors that answer `a + b` along that chain with the precomputed prefix. */
int catre_loss_fwd_sums(const float* pose, const float* scale, const float* gt_rot, const float* gt_trans,
                        const float* gt_scale, const float* kps, const float* cands, const unsigned char* valid,
                        const int32_t* is_sym, const catre_loss_cfg* cfg, int32_t* best, int32_t* counts, float* part_ws,
                        float* losses, const float* trans_deltas, const int32_t* terms, int n_terms, float* prefix, int B,
                        int M, int S1, void* stream);
int catre_loss_bwd_sums(const float* pose, const float* scale, const float* gt_rot, const float* gt_trans,
                        const float* gt_scale, const float* kps, const float* cands, const int32_t* is_sym,
                        const int32_t* best, const int32_t* counts, const float* upstream,
                        const float* const* up_prefix, const int32_t* terms, int n_terms, const catre_loss_cfg* cfg,
                        float* dpose, float* dscale, int B, int M, int S1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CATRE_HIP_H */
