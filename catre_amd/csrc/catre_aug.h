// catre_aug.h - SURVEY.md row f2: the train-time batch glue the reference runs as per-object Python loops of tiny
// torch.mm calls (core/catre/engine/engine_utils.py:107-172, core/utils/pose_aug.py:10-101), as two launches.
// Random numbers are NOT drawn here: the host wrapper draws them with the same torch / numpy calls as the reference
// (same generator consumption) and hands them in, so the kernels are deterministic functions that can be checked
// against the reference's own outputs.
#pragma once

struct AugParams {
  float ratios[3];   // aug_3d_bbox: (ex, ey, ez)                       engine_utils.py:110-113
  float delta_r[9];  // aug_RT: get_rotation_torch(rx, ry, rz)          engine_utils.py:149-154
  float delta_t[3];  //         (tx, ty, tz)
  int do_bbox, do_rt;
};

// per-object affine of one augmentation pass:  p' = dR * (R * (ratios .* (R^T (p - t))) + t + dt)
// plus the matching pose / scale update.  Operation order follows the reference functions line by line.
__global__ void k_aug_points(const float* __restrict__ pcl, const float* __restrict__ pose,
                             const float* __restrict__ scale, const int* __restrict__ sym, AugParams prm,
                             float* __restrict__ pcl_out, float* __restrict__ pose_out, float* __restrict__ scale_out,
                             int B, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  const int b = i / N;
  const float* P = pose + b * 12;
  const float R[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
  const float t[3] = {P[3], P[7], P[11]};
  float r3[3] = {1.f, 1.f, 1.f};
  if (prm.do_bbox) {
    if (sym && sym[b]) {  // y-axis symmetry: x and z share one ratio (engine_utils.py:126-128)
      const float exz = (prm.ratios[0] + prm.ratios[2]) / 2;
      r3[0] = exz;
      r3[1] = prm.ratios[1];
      r3[2] = exz;
    } else {
      r3[0] = prm.ratios[0];
      r3[1] = prm.ratios[1];
      r3[2] = prm.ratios[2];
    }
  }
  float p[3] = {pcl[(size_t)i * 3], pcl[(size_t)i * 3 + 1], pcl[(size_t)i * 3 + 2]};
  if (prm.do_bbox) {
    const float d[3] = {p[0] - t[0], p[1] - t[1], p[2] - t[2]};
    float q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = (R[k] * d[0] + R[3 + k] * d[1] + R[6 + k] * d[2]) * r3[k];  // R^T d, scaled
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = R[3 * k] * q[0] + R[3 * k + 1] * q[1] + R[3 * k + 2] * q[2] + t[k];
  }
  if (prm.do_rt) {
    const float d[3] = {p[0] + prm.delta_t[0], p[1] + prm.delta_t[1], p[2] + prm.delta_t[2]};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      p[k] = prm.delta_r[3 * k] * d[0] + prm.delta_r[3 * k + 1] * d[1] + prm.delta_r[3 * k + 2] * d[2];
  }
  pcl_out[(size_t)i * 3] = p[0];
  pcl_out[(size_t)i * 3 + 1] = p[1];
  pcl_out[(size_t)i * 3 + 2] = p[2];
  if (i % N == 0) {  // one thread per object also writes the updated pose / scale
    float* Po = pose_out + b * 12;
    if (prm.do_rt) {
      const float td[3] = {t[0] + prm.delta_t[0], t[1] + prm.delta_t[1], t[2] + prm.delta_t[2]};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          Po[r * 4 + c] = prm.delta_r[3 * r] * R[c] + prm.delta_r[3 * r + 1] * R[3 + c] + prm.delta_r[3 * r + 2] * R[6 + c];
        Po[r * 4 + 3] = prm.delta_r[3 * r] * td[0] + prm.delta_r[3 * r + 1] * td[1] + prm.delta_r[3 * r + 2] * td[2];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k) Po[k] = P[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) scale_out[b * 3 + k] = scale[b * 3 + k] * r3[k];
  }
}

// aug_poses_normal (pose_aug.py:59-101) and aug_scale_normal (:10-35) from caller-drawn normal noise.
//   R' = euler2mat_torch(clamp(euler_deg, +-max_rot) * pi/180) @ R   (xmat @ ymat @ zmat, pose_utils.py:266-296)
//   t' = t + trans_noise, t'_z = max(t'_z, max(min_z, 1e-4));  s' = clamp(s + scale_noise, max(min_s,1e-4), max_s)
__global__ void k_init_noise(const float* __restrict__ pose, const float* __restrict__ euler_deg,
                             const float* __restrict__ trans_noise, float max_rot, int clamp_rot, float min_z,
                             float* __restrict__ pose_out, const float* __restrict__ scale,
                             const float* __restrict__ scale_noise, float min_s, float max_s,
                             float* __restrict__ scale_out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (pose) {
    const float* P = pose + b * 12;
    float e[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v = euler_deg[b * 3 + k];
      if (clamp_rot) v = fminf(fmaxf(v, -max_rot), max_rot);
      e[k] = v * 3.14159265358979323846f / 180.0f;
    }
    const float cx = cosf(e[0]), sx = sinf(e[0]), cy = cosf(e[1]), sy = sinf(e[1]), cz = cosf(e[2]), sz = sinf(e[2]);
    // ymat @ zmat
    const float yz[9] = {cy * cz, -cy * sz, sy, sz, cz, 0.f, -sy * cz, sy * sz, cy};
    // xmat @ (ymat @ zmat)
    float Rn[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Rn[c] = yz[c];
      Rn[3 + c] = cx * yz[3 + c] - sx * yz[6 + c];
      Rn[6 + c] = sx * yz[3 + c] + cx * yz[6 + c];
    }
    float* Po = pose_out + b * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) Po[r * 4 + c] = Rn[3 * r] * P[c] + Rn[3 * r + 1] * P[4 + c] + Rn[3 * r + 2] * P[8 + c];
      Po[r * 4 + 3] = P[r * 4 + 3] + trans_noise[b * 3 + r];
    }
    Po[11] = fmaxf(Po[11], fmaxf(min_z, 1e-4f));
  }
  if (scale) {
    const float lo = fmaxf(min_s, 1e-4f);
#pragma unroll
    for (int k = 0; k < 3; ++k) scale_out[b * 3 + k] = fminf(fmaxf(scale[b * 3 + k] + scale_noise[b * 3 + k], lo), max_s);
  }
}
