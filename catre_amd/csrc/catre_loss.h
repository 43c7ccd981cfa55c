// catre_loss.h - SURVEY.md row f1: the training loss of CATRE_disR_shared.catre_loss
// (core/catre/models/CATRE_disR_shared.py:168-288) with PyPMLoss (core/catre/losses/pm_loss.py:85-194, the shipped
// L1 / R-only form) and the symmetry-aware choice of the ground-truth rotation (core/utils/pose_utils.py:472-528),
// forward and backward, one workgroup per object.  The reference evaluates up to 314 candidate rotations per
// symmetric object in a numpy loop on the host and syncs ~20 scalars per iteration; the first device version used
// ~150 small torch kernels per iteration.  Here: k_loss_fwd + k_loss_reduce, and k_loss_bwd.
#pragma once

typedef catre_loss_cfg LossCfg;  // include/catre_hip.h

// per-object partial sums: 0 PM |est - tgt|, 1 rot (non-sym), 2 y-axis (sym), 3 trans xy (or xyz), 4 trans z, 5 scale,
// 6 rotation error re() in degrees, 7 translation error te() (lib/pysixd/pose_error.py:359-374,406-417) - the last
// two feed the forward-side logging scalars of CATRE_disR_shared.forward (reference :127-164)
#define LOSS_NP 8
#define LOSS_NVIS 14

__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// R_gt @ S_k for candidate k (k = 0 is the identity)
__device__ __forceinline__ void sym_candidate(const float* __restrict__ G, const float* __restrict__ S, float (&C)[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = G[i * 3] * S[j] + G[i * 3 + 1] * S[3 + j] + G[i * 3 + 2] * S[6 + j];
}

// best[b] = arg-max over the valid candidates of clamp((min(trace(P C^T), 3) - 1) / 2, -1, 1), first maximum
// (== the reference's strict `<` scan over re() that starts at the un-rotated ground truth)
__device__ __forceinline__ int closest_candidate(const float* __restrict__ P, const float* __restrict__ G,
                                                 const float* __restrict__ cands, const unsigned char* __restrict__ valid,
                                                 int S1, float* sval, int* sidx) {
  float bv = -3.f;
  int bi = 0x7fffffff;
  for (int k = threadIdx.x; k < S1; k += 256) {
    if (!valid[k]) continue;
    float C[9];
    sym_candidate(G, cands + (size_t)k * 9, C);
    float tr = 0.f;
#pragma unroll
    for (int e = 0; e < 9; ++e) tr = fmaf(P[e], C[e], tr);
    const float v = fminf(fmaxf(0.5f * (fminf(tr, 3.0f) - 1.0f), -1.0f), 1.0f);
    if (v > bv) {  // k increases within a thread: strict > keeps the first
      bv = v;
      bi = k;
    }
  }
  sval[threadIdx.x] = bv;
  sidx[threadIdx.x] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const float v2 = sval[threadIdx.x + o];
      const int i2 = sidx[threadIdx.x + o];
      if (v2 > sval[threadIdx.x] || (v2 == sval[threadIdx.x] && i2 < sidx[threadIdx.x])) {
        sval[threadIdx.x] = v2;
        sidx[threadIdx.x] = i2;
      }
    }
    __syncthreads();
  }
  return sidx[0];
}

__device__ __forceinline__ float smooth_l1(float d) {  // beta = 1
  const float a = fabsf(d);
  return a < 1.f ? 0.5f * d * d : a - 0.5f;
}

__global__ __launch_bounds__(256) void k_loss_fwd(const float* __restrict__ pose /*[B,3,4]*/,
                                                  const float* __restrict__ scale, const float* __restrict__ gt_rot,
                                                  const float* __restrict__ gt_trans, const float* __restrict__ gt_scale,
                                                  const float* __restrict__ kps /*[B,M,3]*/,
                                                  const float* __restrict__ cands /*[B,S1,3,3]*/,
                                                  const unsigned char* __restrict__ valid /*[B,S1]*/,
                                                  const int* __restrict__ is_sym, LossCfg cfg, int* __restrict__ best,
                                                  float* __restrict__ part /*[B][LOSS_NP]*/, int B, int M, int S1) {
  __shared__ float sval[256];
  __shared__ int sidx[256];
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* Pp = pose + b * 12;
  const float P[9] = {Pp[0], Pp[1], Pp[2], Pp[4], Pp[5], Pp[6], Pp[8], Pp[9], Pp[10]};
  const float t[3] = {Pp[3], Pp[7], Pp[11]};
  const float* G = gt_rot + b * 9;
  float out[LOSS_NP];
#pragma unroll
  for (int i = 0; i < LOSS_NP; ++i) out[i] = 0.f;
  if (cfg.pm_on) {
    int k = 0;
    if (cfg.pm_sym) k = closest_candidate(P, G, cands + (size_t)b * S1 * 9, valid + (size_t)b * S1, S1, sval, sidx);
    if (tid == 0) best[b] = k;
    float C[9];
    sym_candidate(G, cands + ((size_t)b * S1 + k) * 9, C);
    float se[3], sg[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      se[j] = cfg.pm_with_scale ? scale[b * 3 + j] : 1.f;
      sg[j] = cfg.pm_with_scale ? gt_scale[b * 3 + j] : 1.f;
    }
    float acc = 0.f;
    for (int m = tid; m < M; m += 256) {
      const float* q = kps + ((size_t)b * M + m) * 3;
      const float pe[3] = {q[0] * se[0], q[1] * se[1], q[2] * se[2]}, pt[3] = {q[0] * sg[0], q[1] * sg[1], q[2] * sg[2]};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float e = P[i * 3] * pe[0] + P[i * 3 + 1] * pe[1] + P[i * 3 + 2] * pe[2];
        const float g = C[i * 3] * pt[0] + C[i * 3 + 1] * pt[1] + C[i * 3 + 2] * pt[2];
        acc += fabsf(e - g);
      }
    }
    out[0] = block_sum256(acc, red);
  }
  if (tid == 0) {
    if (cfg.rot_on) {
      if (!is_sym[b]) {
        if (cfg.rot_l2) {
          float s = 0.f;
#pragma unroll
          for (int e = 0; e < 9; ++e) s += (P[e] - G[e]) * (P[e] - G[e]);
          out[1] = s;
        } else {
          float tr = 0.f;
#pragma unroll
          for (int e = 0; e < 9; ++e) tr += P[e] * G[e];
          out[1] = (1.f - (tr - 1.f) / 2.f) / 2.f;
        }
      } else {
        float s = 0.f;
        if (cfg.yaxis_smooth >= 2) {  // 2: L2Loss (l2_loss.py:5-28, per-object norm); 3: angular_distance_vec (rot_loss.py:33-42)
          float dd = 0.f, pg = 0.f, pp = 0.f, gg = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const float p = P[i * 3 + 1], g = G[i * 3 + 1];
            dd += (p - g) * (p - g);
            pg += p * g;
            pp += p * p;
            gg += g * g;
          }
          s = cfg.yaxis_smooth == 2 ? sqrtf(dd) : (1.f - pg / (sqrtf(pp) * sqrtf(gg))) * 0.5f;
        } else {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const float d = P[i * 3 + 1] - G[i * 3 + 1];
            s += cfg.yaxis_smooth ? smooth_l1(d) : fabsf(d);
          }
        }
        out[2] = s;
      }
    }
    if (cfg.trans_on) {
      const float d[3] = {t[0] - gt_trans[b * 3], t[1] - gt_trans[b * 3 + 1], t[2] - gt_trans[b * 3 + 2]};
      float f[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) f[i] = cfg.trans_mse == 1 ? d[i] * d[i] : fabsf(d[i]);
      if (cfg.trans_mse == 2) {  // L2Loss: per-object Euclidean norm
        out[3] = sqrtf(cfg.trans_split ? d[0] * d[0] + d[1] * d[1] : d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      } else {
        out[3] = cfg.trans_split ? f[0] + f[1] : f[0] + f[1] + f[2];
      }
      out[4] = f[2];
    }
    if (cfg.scale_on) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float d = scale[b * 3 + i] - gt_scale[b * 3 + i];
        s += cfg.scale_mse ? d * d : fabsf(d);
      }
      out[5] = cfg.scale_mse == 2 ? sqrtf(s) : s;
    }
    {  // compute_mean_re_te (models/model_utils.py:226-238): re against the plain ground truth, te
      float tr = 0.f;
#pragma unroll
      for (int e = 0; e < 9; ++e) tr = fmaf(P[e], G[e], tr);  // trace(R_est R_gt^T)
      tr = fminf(tr, 3.0f);
      out[6] = acosf(fminf(1.0f, fmaxf(-1.0f, 0.5f * (tr - 1.0f)))) * 57.29577951308232f;
      const float d0 = gt_trans[b * 3] - t[0], d1 = gt_trans[b * 3 + 1] - t[1], d2 = gt_trans[b * 3 + 2] - t[2];
      out[7] = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    }
#pragma unroll
    for (int i = 0; i < LOSS_NP; ++i) part[(size_t)b * LOSS_NP + i] = out[i];
  }
}

// losses[6] = PM_R, rot, yaxis_rot, trans_xy (or trans), trans_z, scale: objects summed in order, then normalised
// counts[2] = {objects with symmetry info, without}: taken from is_sym on the device, so a captured graph stays valid
// when the mix of objects changes from batch to batch
// losses[6 .. 6+14) = the reference's vis/ scalars in its own order: error_R [deg], error_t [cm], |t_pred - t_gt| of
// object 0 [cm] x3, t_pred x3, trans_deltas x3 (0 when no deltas are passed), t_gt x3 - all of object 0 like the
// reference (`pred_trans[0, 0]` ...)
// column sums of part [B][LOSS_NP] and the symmetric-object count: wave w of the 8 adds column w (lane l: objects l, l + 64,
// ... in order, then the butterfly over lanes - a fixed order), wave 0 also counts is_sym.  (A single wave walking the
// objects in order was 2 x 16 dependent L2 round trips: 13.5 us.)
__global__ __launch_bounds__(512) void k_loss_reduce(const float* __restrict__ part, const int* __restrict__ is_sym, LossCfg cfg,
                                                     float* __restrict__ losses, int* __restrict__ counts, int B, int M,
                                                     const float* __restrict__ pose, const float* __restrict__ gt_trans,
                                                     const float* __restrict__ trans_deltas, unsigned term_order = 0,
                                                     int n_terms = 0, float* __restrict__ prefix = nullptr) {
  static_assert(LOSS_NP == 8, "one wave per column of the partials");
  __shared__ float colsum[LOSS_NP];
  __shared__ float lossv[6];
  __shared__ int nsym_s;
  {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float s = 0.f, c = 0.f;
    for (int b = lane; b < B; b += 64) {
      s += part[(size_t)b * LOSS_NP + w];
      if (w == 0) c += is_sym[b] != 0 ? 1.f : 0.f;
    }
    s = wave_sum(s);
    if (w == 0) c = wave_sum(c);
    if (lane == 0) colsum[w] = s;
    if (lane == 0 && w == 0) nsym_s = (int)c;
  }
  __syncthreads();
  const int i = threadIdx.x;
  if (i >= 6 && i < 6 + LOSS_NVIS) {
    const int k = i - 6;
    float v;
    if (k < 2) {
      const float s = colsum[6 + k];
      v = s / (float)B * (k == 1 ? 100.f : 1.f);
    } else {
      const int c = (k - 2) % 3, what = (k - 2) / 3;
      const float tp = pose[c * 4 + 3], tg = gt_trans[c];
      v = what == 0 ? fabsf(tp - tg) * 100.f : what == 1 ? tp : what == 2 ? (trans_deltas ? trans_deltas[c] : 0.f) : tg;
    }
    losses[i] = v;
  }
  // prefix[k] = ((0 + l[t0]) + l[t1]) + ... + l[tk] over the terms the caller's loss dict holds, in its order: what python's
  // `sum(loss_dict.values())` (engine.py:318) builds one add kernel at a time - same operations, same bits
  if (i < 6) {
  const int n_sym = nsym_s;
  const int n_nonsym = B - n_sym;
  if (i == 0) {
    counts[0] = n_sym;
    counts[1] = n_nonsym;
  }
  const float s = colsum[i];
  float v = 0.f;
  switch (i) {
    case 0: v = 3.f * (s / ((float)B * M * 3.f)) * cfg.pm_lw; break;
    case 1: v = n_nonsym > 0 ? s / ((float)n_nonsym * (cfg.rot_l2 ? 9.f : 1.f)) * cfg.rot_lw : 0.f; break;
    case 2: v = n_sym > 0 ? s / ((float)n_sym * (cfg.yaxis_smooth >= 2 ? 1.f : 3.f)) * cfg.rot_lw : 0.f; break;
    case 3: v = s / ((float)B * (cfg.trans_mse == 2 ? 1.f : cfg.trans_split ? 2.f : 3.f)) * cfg.trans_lw; break;
    case 4: v = s / (float)B * cfg.trans_lw; break;
    case 5: v = s / ((float)B * (cfg.scale_mse == 2 ? 1.f : 3.f)) * cfg.scale_lw; break;
  }
  losses[i] = v;
  lossv[i] = v;
  }
  if (n_terms <= 0 || !prefix) return;
  __syncthreads();
  if (i == 0) {
    float acc = 0.f;
    for (int k = 0; k < n_terms; ++k) {
      acc += lossv[(term_order >> (4 * k)) & 15u];
      prefix[k] = acc;
    }
  }
}

// upstream gradients of the running sums (one device scalar each, null = zero)
struct LossUpPrefix {
  const float* p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

// d(sum_i up[i] * losses[i]) / d(pose, scale);  up = the six upstream gradients (device)
__global__ __launch_bounds__(256) void k_loss_bwd(const float* __restrict__ pose, const float* __restrict__ scale,
                                                  const float* __restrict__ gt_rot, const float* __restrict__ gt_trans,
                                                  const float* __restrict__ gt_scale, const float* __restrict__ kps,
                                                  const float* __restrict__ cands, const int* __restrict__ is_sym,
                                                  const int* __restrict__ best, const float* __restrict__ up_, LossCfg cfg,
                                                  const int* __restrict__ counts, float* __restrict__ dpose /*[B,3,4]*/,
                                                  float* __restrict__ dscale, int B, int M, int S1,
                                                  const LossUpPrefix up_prefix = LossUpPrefix{}, unsigned term_order = 0,
                                                  int n_terms = 0) {
  __shared__ float red[4];
  // effective upstream of loss i: its own (up_in, optional) plus that of every prefix sum it is part of (k >= its position)
  float up[6];
  {
    const float* up_in = up_;
#pragma unroll
    for (int i = 0; i < 6; ++i) up[i] = up_in ? up_in[i] : 0.f;
    {
      float tail = 0.f;
      for (int k = n_terms - 1; k >= 0; --k) {
        if (up_prefix.p[k]) tail += up_prefix.p[k][0];
        const int t = (term_order >> (4 * k)) & 15u;
#pragma unroll
        for (int i = 0; i < 6; ++i)
          if (i == t) up[i] += tail;
      }
    }
  }
  const int n_sym = counts[0], n_nonsym = counts[1];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* Pp = pose + b * 12;
  const float P[9] = {Pp[0], Pp[1], Pp[2], Pp[4], Pp[5], Pp[6], Pp[8], Pp[9], Pp[10]};
  const float t[3] = {Pp[3], Pp[7], Pp[11]};
  const float* G = gt_rot + b * 9;
  float dR[9], ds[3];
#pragma unroll
  for (int e = 0; e < 9; ++e) dR[e] = 0.f;
  ds[0] = ds[1] = ds[2] = 0.f;
  if (cfg.pm_on) {
    float C[9];
    sym_candidate(G, cands + ((size_t)b * S1 + best[b]) * 9, C);
    float se[3], sg[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      se[j] = cfg.pm_with_scale ? scale[b * 3 + j] : 1.f;
      sg[j] = cfg.pm_with_scale ? gt_scale[b * 3 + j] : 1.f;
    }
    const float c = up[0] * 3.f * cfg.pm_lw / ((float)B * M * 3.f);
    for (int m = tid; m < M; m += 256) {
      const float* q = kps + ((size_t)b * M + m) * 3;
      const float pe[3] = {q[0] * se[0], q[1] * se[1], q[2] * se[2]}, pt[3] = {q[0] * sg[0], q[1] * sg[1], q[2] * sg[2]};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float e = P[i * 3] * pe[0] + P[i * 3 + 1] * pe[1] + P[i * 3 + 2] * pe[2];
        const float g = C[i * 3] * pt[0] + C[i * 3 + 1] * pt[1] + C[i * 3 + 2] * pt[2];
        const float d = e - g;
        const float sgn = d > 0.f ? c : (d < 0.f ? -c : 0.f);  // torch: sign(0) = 0
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          dR[i * 3 + j] = fmaf(sgn, pe[j], dR[i * 3 + j]);
          ds[j] = fmaf(sgn * P[i * 3 + j], q[j], ds[j]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) dR[e] = block_sum256(dR[e], red);
#pragma unroll
    for (int j = 0; j < 3; ++j) ds[j] = cfg.pm_with_scale ? block_sum256(ds[j], red) : 0.f;
  }
  if (tid != 0) return;
  float dt[3] = {0.f, 0.f, 0.f};
  if (cfg.rot_on) {
    if (!is_sym[b]) {
      if (n_nonsym > 0) {
        if (cfg.rot_l2) {
          const float c = up[1] * cfg.rot_lw * 2.f / ((float)n_nonsym * 9.f);
#pragma unroll
          for (int e = 0; e < 9; ++e) dR[e] += c * (P[e] - G[e]);
        } else {
          const float c = -up[1] * cfg.rot_lw / (4.f * (float)n_nonsym);
#pragma unroll
          for (int e = 0; e < 9; ++e) dR[e] += c * G[e];
        }
      }
    } else if (n_sym > 0 && cfg.yaxis_smooth >= 2) {
      const float c = up[2] * cfg.rot_lw / (float)n_sym;
      float dd = 0.f, pg = 0.f, pp = 0.f, gg = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float p = P[i * 3 + 1], g = G[i * 3 + 1];
        dd += (p - g) * (p - g);
        pg += p * g;
        pp += p * p;
        gg += g * g;
      }
      if (cfg.yaxis_smooth == 2) {
        const float nrm = sqrtf(dd);
#pragma unroll
        for (int i = 0; i < 3; ++i) dR[i * 3 + 1] += nrm > 0.f ? c * (P[i * 3 + 1] - G[i * 3 + 1]) / nrm : 0.f;
      } else {
        const float np_ = sqrtf(pp), ng = sqrtf(gg), cs = pg / (np_ * ng);
#pragma unroll
        for (int i = 0; i < 3; ++i) dR[i * 3 + 1] += -0.5f * c * (G[i * 3 + 1] / (np_ * ng) - cs * P[i * 3 + 1] / pp);
      }
    } else if (n_sym > 0) {
      const float c = up[2] * cfg.rot_lw / ((float)n_sym * 3.f);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float d = P[i * 3 + 1] - G[i * 3 + 1];
        float gd;
        if (cfg.yaxis_smooth)
          gd = fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
        else
          gd = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        dR[i * 3 + 1] += c * gd;
      }
    }
  }
  if (cfg.trans_on) {
    const float d[3] = {t[0] - gt_trans[b * 3], t[1] - gt_trans[b * 3 + 1], t[2] - gt_trans[b * 3 + 2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float sg = d[i] > 0.f ? 1.f : (d[i] < 0.f ? -1.f : 0.f);
      float gd = cfg.trans_mse == 1 ? 2.f * d[i] : sg;
      float c;
      if (cfg.trans_mse == 2) {  // d ||d|| / d d_i = d_i / ||d|| over the components the norm spans
        const bool in_norm = !cfg.trans_split || i < 2;
        const float nrm = sqrtf(cfg.trans_split ? d[0] * d[0] + d[1] * d[1] : d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        gd = in_norm ? (nrm > 0.f ? d[i] / nrm : 0.f) : sg;
        c = (in_norm ? up[3] : up[4]) * cfg.trans_lw / (float)B;
      } else if (cfg.trans_split) {
        c = i < 2 ? up[3] * cfg.trans_lw / ((float)B * 2.f) : up[4] * cfg.trans_lw / (float)B;
      } else {
        c = up[3] * cfg.trans_lw / ((float)B * 3.f);
      }
      dt[i] = c * gd;
    }
  }
  if (cfg.scale_on) {
    const float c = up[5] * cfg.scale_lw / ((float)B * (cfg.scale_mse == 2 ? 1.f : 3.f));
    const float d3[3] = {scale[b * 3] - gt_scale[b * 3], scale[b * 3 + 1] - gt_scale[b * 3 + 1],
                         scale[b * 3 + 2] - gt_scale[b * 3 + 2]};
    const float nrm = sqrtf(d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float d = d3[i];
      ds[i] += c * (cfg.scale_mse == 2 ? (nrm > 0.f ? d / nrm : 0.f)
                                       : cfg.scale_mse == 1 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
    }
  }
  float* o = dpose + b * 12;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o[i * 4] = dR[i * 3];
    o[i * 4 + 1] = dR[i * 3 + 1];
    o[i * 4 + 2] = dR[i * 3 + 2];
    o[i * 4 + 3] = dt[i];
  }
  dscale[b * 3] = ds[0];
  dscale[b * 3 + 1] = ds[1];
  dscale[b * 3 + 2] = ds[2];
}
