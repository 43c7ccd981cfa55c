// catre_so3.h - a10: the four rotation parametrisations of get_rot_mat (core/catre/models/model_utils.py:28-40) and
// their reverse-mode derivatives, one object per thread (included by catre_kernels.hip before the pose-update kernels).
//
//   CATRE_ROT_6D        [B,6]  rot6d_to_mat_batch          core/utils/rot_reps.py:34-55
//   CATRE_ROT_QUAT      [B,4]  quat2mat_torch              core/utils/pose_utils.py:349-412   (w, x, y, z; eps = 0)
//   CATRE_ROT_LOG_QUAT  [B,3]  quat2mat_torch(qexp(.))     core/utils/quaternion_lf.py:294-317 (eps = 1e-8 clamp on theta)
//   CATRE_ROT_LIE_VEC   [B,3]  lie_vec_to_rot              core/utils/lie_algebra.py:7-77     (Rodrigues with the
//                              v / (theta + 1e-6) axis of the reference; first-order branch when theta^2 <= 1e-6)
//
// Gradient conventions follow what torch.autograd gives the reference wherever that is finite.  Where the reference's
// graph produces 0 * inf = NaN (lie_vec at exactly v = 0: sqrt'(0) under a zero mask; log_quat at v = 0: norm'(0)) the
// finite masked-branch value is returned instead (the first-order branch's gradient / zero).
#pragma once

__host__ __device__ inline int catre_rot_dim(int rot_type) {
  return rot_type == CATRE_ROT_QUAT ? 4 : (rot_type == CATRE_ROT_6D ? 6 : 3);
}

__device__ __forceinline__ void so3_cross(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// rot6d -> R, columns (x, y, z); F.normalize eps 1e-12
__device__ __forceinline__ void rot6d_to_mat(const float* r, float* R) {
  const float na = fmaxf(sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]), 1e-12f);
  const float x[3] = {r[0] / na, r[1] / na, r[2] / na};
  const float yr[3] = {r[3], r[4], r[5]};
  float z[3], y[3];
  so3_cross(x, yr, z);
  const float nz = fmaxf(sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]), 1e-12f);
  z[0] /= nz;
  z[1] /= nz;
  z[2] /= nz;
  so3_cross(z, x, y);
  R[0] = x[0]; R[1] = y[0]; R[2] = z[0];
  R[3] = x[1]; R[4] = y[1]; R[5] = z[1];
  R[6] = x[2]; R[7] = y[2]; R[8] = z[2];
}

__device__ __forceinline__ void rot6d_to_mat_bwd(const float* r, const float* gR, float* g) {
  const float bb[3] = {r[3], r[4], r[5]};
  const float na = fmaxf(sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]), 1e-12f);
  const float x[3] = {r[0] / na, r[1] / na, r[2] / na};
  float w[3];
  so3_cross(x, bb, w);
  const float nw = fmaxf(sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), 1e-12f);
  const float z[3] = {w[0] / nw, w[1] / nw, w[2] / nw};
  float gx[3] = {gR[0], gR[3], gR[6]}, gy[3] = {gR[1], gR[4], gR[7]}, gz[3] = {gR[2], gR[5], gR[8]};
  float t[3];
  so3_cross(x, gy, t);  // y = z cross x : dz += x cross gy ; dx += gy cross z
  gz[0] += t[0]; gz[1] += t[1]; gz[2] += t[2];
  so3_cross(gy, z, t);
  gx[0] += t[0]; gx[1] += t[1]; gx[2] += t[2];
  const float zg = z[0] * gz[0] + z[1] * gz[1] + z[2] * gz[2];  // z = w / |w|
  const float gw[3] = {(gz[0] - z[0] * zg) / nw, (gz[1] - z[1] * zg) / nw, (gz[2] - z[2] * zg) / nw};
  so3_cross(bb, gw, t);  // w = x cross b : dx += b cross gw ; db = gw cross x
  gx[0] += t[0]; gx[1] += t[1]; gx[2] += t[2];
  float gb[3];
  so3_cross(gw, x, gb);
  const float xg = x[0] * gx[0] + x[1] * gx[1] + x[2] * gx[2];
  g[0] = (gx[0] - x[0] * xg) / na;
  g[1] = (gx[1] - x[1] * xg) / na;
  g[2] = (gx[2] - x[2] * xg) / na;
  g[3] = gb[0];
  g[4] = gb[1];
  g[5] = gb[2];
}

// unit quaternion (w, x, y, z) -> R with the operation order of quat2mat_torch (pose_utils.py:378-405)
__device__ __forceinline__ void unit_quat_to_mat(float qw, float qx, float qy, float qz, float* R) {
  const float X = qx * 2.f, Y = qy * 2.f, Z = qz * 2.f;
  const float wX = qw * X, wY = qw * Y, wZ = qw * Z, xX = qx * X, xY = qx * Y, xZ = qx * Z;
  const float yY = qy * Y, yZ = qy * Z, zZ = qz * Z;
  R[0] = 1.f - (yY + zZ); R[1] = xY - wZ;         R[2] = xZ + wY;
  R[3] = xY + wZ;         R[4] = 1.f - (xX + zZ); R[5] = yZ - wX;
  R[6] = xZ - wY;         R[7] = yZ + wX;         R[8] = 1.f - (xX + yY);
}

// dL/d(unit quaternion) from dL/dR
__device__ __forceinline__ void unit_quat_to_mat_bwd(float qw, float qx, float qy, float qz, const float* gA, float* gq) {
  gq[0] = 2.f * (-qz * gA[1] + qy * gA[2] + qz * gA[3] - qx * gA[5] - qy * gA[6] + qx * gA[7]);
  gq[1] = 2.f * (qy * gA[1] + qz * gA[2] + qy * gA[3] - 2.f * qx * gA[4] - qw * gA[5] + qz * gA[6] + qw * gA[7] -
                 2.f * qx * gA[8]);
  gq[2] = 2.f * (-2.f * qy * gA[0] + qx * gA[1] + qw * gA[2] + qx * gA[3] + qz * gA[5] - qw * gA[6] + qz * gA[7] -
                 2.f * qy * gA[8]);
  gq[3] = 2.f * (-2.f * qz * gA[0] - qw * gA[1] + qx * gA[2] + qw * gA[3] - 2.f * qz * gA[4] + qy * gA[5] + qx * gA[6] +
                 qy * gA[7]);
}

// quat2mat_torch(q, eps = 0): q / |q| then unit_quat_to_mat
__device__ __forceinline__ void quat_to_mat(const float* q, float* R) {
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  unit_quat_to_mat(q[0] / n, q[1] / n, q[2] / n, q[3] / n, R);
}

__device__ __forceinline__ void quat_to_mat_bwd(const float* q, const float* gR, float* g) {
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float u[4] = {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
  float gu[4];
  unit_quat_to_mat_bwd(u[0], u[1], u[2], u[3], gR, gu);
  const float ug = u[0] * gu[0] + u[1] * gu[1] + u[2] * gu[2] + u[3] * gu[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) g[i] = (gu[i] - u[i] * ug) / n;
}

// qexp of a pure quaternion (0; v) (quaternion_lf.py:306-317): (cos th, sin th / clamp(th, 1e-8) * v)
__device__ __forceinline__ void qexp3(const float* v, float* q) {
  const float th = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float k = 1.0f / fmaxf(th, 1e-8f) * sinf(th);
  q[0] = cosf(th);
  q[1] = k * v[0];
  q[2] = k * v[1];
  q[3] = k * v[2];
}

__device__ __forceinline__ void log_quat_to_mat(const float* v, float* R) {
  float q[4];
  qexp3(v, q);
  quat_to_mat(q, R);
}

__device__ __forceinline__ void log_quat_to_mat_bwd(const float* v, const float* gR, float* g) {
  float q[4], gq[4];
  qexp3(v, q);
  quat_to_mat_bwd(q, gR, gq);
  const float th = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (!(th > 0.f)) {  // norm'(0) = 0 in torch; sin(0) = 0 kills the direct term too
    g[0] = g[1] = g[2] = 0.f;
    return;
  }
  const float s = sinf(th), c = cosf(th);
  const bool clamped = th < 1e-8f;
  const float inv = 1.0f / fmaxf(th, 1e-8f);
  const float k = inv * s;
  // q0 = cos th; q_i = k v_i, k = sin th / thc.  dk/dth = cos th / thc - (clamped ? 0 : sin th / th^2)
  const float dk = c * inv - (clamped ? 0.f : s * inv * inv);
  const float gv_dot = gq[1] * v[0] + gq[2] * v[1] + gq[3] * v[2];
  const float gth = -s * gq[0] + dk * gv_dot;
#pragma unroll
  for (int i = 0; i < 3; ++i) g[i] = k * gq[1 + i] + gth * v[i] / th;
}

// lie_vec_to_rot (lie_algebra.py:26-76)
__device__ __forceinline__ void lie_vec_to_mat(const float* v, float* R) {
  const float th2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (th2 > 1e-6f) {
    const float th = sqrtf(th2);
    const float a = th + 1e-6f;
    const float wx = v[0] / a, wy = v[1] / a, wz = v[2] / a;
    const float c = cosf(th), s = sinf(th), k = 1.0f - c;
    R[0] = c + wx * wx * k;       R[1] = wx * wy * k - wz * s;  R[2] = wy * s + wx * wz * k;
    R[3] = wz * s + wx * wy * k;  R[4] = c + wy * wy * k;       R[5] = -wx * s + wy * wz * k;
    R[6] = -wy * s + wx * wz * k; R[7] = wx * s + wy * wz * k;  R[8] = c + wz * wz * k;
  } else {
    R[0] = 1.f;   R[1] = -v[2]; R[2] = v[1];
    R[3] = v[2];  R[4] = 1.f;   R[5] = -v[0];
    R[6] = -v[1]; R[7] = v[0];  R[8] = 1.f;
  }
}

__device__ __forceinline__ void lie_vec_to_mat_bwd(const float* v, const float* gR, float* g) {
  const float th2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (th2 > 1e-6f) {
    const float th = sqrtf(th2);
    const float a = th + 1e-6f;
    const float w[3] = {v[0] / a, v[1] / a, v[2] / a};
    const float c = cosf(th), s = sinf(th), k = 1.0f - c;
    // R = c I + k w w^T + s [w]x
    float wgw = 0.f, gw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        t += (gR[i * 3 + j] + gR[j * 3 + i]) * w[j];
        wgw += gR[i * 3 + j] * w[i] * w[j];
      }
      gw[i] = k * t;
    }
    const float e[3] = {gR[7] - gR[5], gR[2] - gR[6], gR[3] - gR[1]};  // d/dw of s [w]x
    gw[0] += s * e[0];
    gw[1] += s * e[1];
    gw[2] += s * e[2];
    const float gc = (gR[0] + gR[4] + gR[8]) - wgw;
    const float gs = w[0] * e[0] + w[1] * e[1] + w[2] * e[2];
    const float gth = -s * gc + c * gs - (gw[0] * v[0] + gw[1] * v[1] + gw[2] * v[2]) / (a * a);
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = gw[i] / a + gth * v[i] / th;
  } else {
    g[0] = gR[7] - gR[5];
    g[1] = gR[2] - gR[6];
    g[2] = gR[3] - gR[1];
  }
}

__device__ __forceinline__ void rot_param_to_mat(const float* r, int rot_type, float* R) {
  switch (rot_type) {
    case CATRE_ROT_QUAT: quat_to_mat(r, R); break;
    case CATRE_ROT_LOG_QUAT: log_quat_to_mat(r, R); break;
    case CATRE_ROT_LIE_VEC: lie_vec_to_mat(r, R); break;
    default: rot6d_to_mat(r, R);
  }
}

__device__ __forceinline__ void rot_param_to_mat_bwd(const float* r, int rot_type, const float* gR, float* g) {
  switch (rot_type) {
    case CATRE_ROT_QUAT: quat_to_mat_bwd(r, gR, g); break;
    case CATRE_ROT_LOG_QUAT: log_quat_to_mat_bwd(r, gR, g); break;
    case CATRE_ROT_LIE_VEC: lie_vec_to_mat_bwd(r, gR, g); break;
    default: rot6d_to_mat_bwd(r, gR, g);
  }
}

// get_rot_mat as a stand-alone launch: rot [B,d] -> R [B,3,3]; and its backward gR [B,3,3] -> g [B,d]
__global__ void k_rot_to_mat(const float* __restrict__ rot, int rot_type, float* __restrict__ R, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int d = catre_rot_dim(rot_type);
  float r[6], m[9];
#pragma unroll
  for (int i = 0; i < 6; ++i) r[i] = i < d ? rot[(size_t)b * d + i] : 0.f;  // fixed trip count: r stays in registers
  rot_param_to_mat(r, rot_type, m);
#pragma unroll
  for (int i = 0; i < 9; ++i) R[(size_t)b * 9 + i] = m[i];
}

__global__ void k_rot_to_mat_bwd(const float* __restrict__ rot, int rot_type, const float* __restrict__ gR,
                                 float* __restrict__ g, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int d = catre_rot_dim(rot_type);
  float r[6], gm[9], go[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 6; ++i) r[i] = i < d ? rot[(size_t)b * d + i] : 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) gm[i] = gR[(size_t)b * 9 + i];
  rot_param_to_mat_bwd(r, rot_type, gm, go);
#pragma unroll
  for (int i = 0; i < 6; ++i)
    if (i < d) g[(size_t)b * d + i] = go[i];
}
