// catre_gram.h - GroupNorm-0 statistics of the rotation heads from second moments of pointfeat (included by
// catre_kernels.hip after catre_rot.h / catre_bf16.h).
//
// Layer 0 of a rotation head is linear in pointfeat:  y0[c,p] = w_c . pf_p + bias0[c, cloud(p)].  The statistics
// GroupNorm needs (mean and variance over 8 channels x all N+M points) therefore follow from the per-cloud mean mu
// and centred scatter matrix S = sum_p (pf_p - mu)(pf_p - mu)^T of pointfeat:
//     sum_p y0[c,p]            = n (w_c . mu + b_c)            =: n a_c
//     sum_p (y0[c,p] - a_c)^2  = w_c^T S w_c
// so the 2 x 256-channel layer (1024 MFMAs per 64-point tile, k_rot_l0_stats) is replaced by a 64 x 64 Gram matrix
// (128 MFMAs per tile) plus one small per-object kernel:
//   k_pf_gram        per tile: mean[64] and centred Gram[64][64] of the tile's valid points (fp32 MFMA; operands are
//                    read from the point-major LDS tile one float per lane, which is what 32x32x2 wants)
//   k_gram_merge     per (object, cloud): mu = weighted mean of the tile means, S = sum_t G_t + n_t d_t d_t^T with
//                    d_t = tile mean - mu (tiles summed in order: deterministic)
//   k_gn0_from_moments  per (object, head, 64 channels): the quadratic forms w^T S w, group statistics over both
//                    clouds, and the fused bias + GroupNorm affine table k_rot_l1 consumes (= k_gn0_affine's output).
// Used by the fp32 path; the bf16 path keeps k_rot_l0_stats_bf (at bf16 MFMA rates the recompute is already cheap).
// Everything is fp32; sums of squares are always taken about a mean (tile mean, then cloud mean), never raw.
#pragma once

template <bool BF>
__global__ __launch_bounds__(256) void k_pf_gram(const void* __restrict__ pointfeat, float* __restrict__ gram /*[B*T][64*64]*/,
                                                 float* __restrict__ tmean /*[B*T][64]*/, int B, int N, int M) {
  __shared__ float pf[TP * LD64];
  __shared__ float part[4][64];
  __shared__ float mean[64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const RotTile rt = rot_tile(blockIdx.x, B, N, M);
  if constexpr (BF) {
    const u32x4* src = reinterpret_cast<const u32x4*>(pointfeat) + (rt.pf_off / 64) * 8;
    for (int i = tid; i < TP * 8; i += 256) {
      const int row = i >> 3, c = i & 7;
      const u32x4 v = src[(size_t)min(row, rt.valid - 1) * 8 + c];
      float* d = pf + row * LD64;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d[bf_chunk_channel(c, 2 * e)] = bf_lo(v[e]);
        d[bf_chunk_channel(c, 2 * e + 1)] = bf_hi(v[e]);
      }
    }
  } else {
    const float* src = reinterpret_cast<const float*>(pointfeat) + rt.pf_off;
    for (int i = tid; i < TP * 16; i += 256) {
      const int row = i >> 4, c4 = i & 15;
      *reinterpret_cast<f32x4*>(pf + row * LD64 + c4 * 4) =
          *reinterpret_cast<const f32x4*>(src + (size_t)min(row, rt.valid - 1) * 64 + c4 * 4);
    }
  }
  __syncthreads();
  {  // tile mean over the valid points: thread = (channel, quarter of the points)
    const int ch = tid & 63, q = tid >> 6;
    float s = 0.f;
    for (int p = q * 16; p < q * 16 + 16; ++p) s += p < rt.valid ? pf[p * LD64 + ch] : 0.f;
    part[q][ch] = s;
  }
  __syncthreads();
  if (tid < 64) {
    const float m = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) / (float)rt.valid;
    mean[tid] = m;
    tmean[(size_t)blockIdx.x * 64 + tid] = m;
  }
  __syncthreads();
  // centred Gram, one 32x32 block per wave: G[i][j] = sum_p (pf[p][i] - m_i)(pf[p][j] - m_j)
  const int bi = wave >> 1, bj = wave & 1, i = lane & 31, h = lane >> 5;
  const float mi = mean[bi * 32 + i], mj = mean[bj * 32 + i];
  f32x16 acc = zero16();
#pragma unroll 8
  for (int t = 0; t < TP / 2; ++t) {
    const int p = 2 * t + h;  // lane half h supplies point 2t + h
    const bool ok = p < rt.valid;
    const float a = ok ? pf[p * LD64 + bi * 32 + i] - mi : 0.f;
    const float b = ok ? pf[p * LD64 + bj * 32 + i] - mj : 0.f;
    acc = mfma32(a, b, acc);
  }
  float* out = gram + (size_t)blockIdx.x * 4096;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = bi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    out[row * 64 + bj * 32 + i] = acc[reg];
  }
}

// per (object, cloud): mu = sum_t n_t m_t / n,  S = sum_t (G_t + n_t (m_t - mu)(m_t - mu)^T), tiles in order
__global__ __launch_bounds__(256) void k_gram_merge(const float* __restrict__ gram, const float* __restrict__ tmean,
                                                    float* __restrict__ Scl /*[B][2][4096]*/,
                                                    float* __restrict__ mucl /*[B][2][64]*/, int B, int N, int M) {
  extern __shared__ float dm[];  // [nt][64] tile means, then deviations from the cloud mean
  __shared__ float mu[64];
  const int obj = blockIdx.x, cl = blockIdx.y, tid = threadIdx.x;
  const int TN = (N + TP - 1) / TP, TM = (M + TP - 1) / TP, T = TN + TM;
  const int t0 = cl ? TN : 0, nt = cl ? TM : TN, npts = cl ? M : N;
  const size_t tile0 = (size_t)obj * T + t0;
  for (int i = tid; i < nt * 64; i += 256) dm[i] = tmean[tile0 * 64 + i];
  __syncthreads();
  if (tid < 64) {
    float s = 0.f;
    for (int t = 0; t < nt; ++t) s = fmaf((float)min(TP, npts - t * TP), dm[t * 64 + tid], s);
    const float m = s / (float)npts;
    mu[tid] = m;
    if (blockIdx.z == 0) mucl[((size_t)obj * 2 + cl) * 64 + tid] = m;
  }
  __syncthreads();
  for (int i = tid; i < nt * 64; i += 256) dm[i] -= mu[i & 63];
  __syncthreads();
  float* out = Scl + ((size_t)obj * 2 + cl) * 4096;
#pragma unroll
  for (int u = 0; u < 4; ++u) {  // blockIdx.z owns a quarter of the 4096 entries
    const int e = blockIdx.z * 1024 + tid + 256 * u, r = e >> 6, c = e & 63;
    float acc = 0.f;
    for (int t = 0; t < nt; ++t) {
      const float nb = (float)min(TP, npts - t * TP);
      acc += gram[(tile0 + t) * 4096 + e] + nb * dm[t * 64 + r] * dm[t * 64 + c];
    }
    out[e] = acc;
  }
}

// aff [B*2 (object, head)][2 (observed, prior)][2 (sc, sh)][256], as k_gn0_affine.  One workgroup per (object, head,
// 64 channels); thread = (channel, quarter of the rows of S) so that small batches still fill some of the chip.
__global__ __launch_bounds__(256) void k_gn0_from_moments(const float* __restrict__ Scl, const float* __restrict__ mucl,
                                                          const float* __restrict__ w0x, const float* __restrict__ w0y,
                                                          int ldw, int coloff, const float* __restrict__ bias0,
                                                          const float* __restrict__ gamx, const float* __restrict__ betx,
                                                          const float* __restrict__ gamy, const float* __restrict__ bety,
                                                          float* __restrict__ aff, int B, int N, int M) {
  __shared__ __attribute__((aligned(16))) float S[2][64 * 64];
  __shared__ float mu[2][64];
  __shared__ float qpart[2][4][64];
  const int obj = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x;
  const int cq = tid & 63, rq = tid >> 6, ch = blockIdx.z * 64 + cq;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(Scl + (size_t)obj * 2 * 4096);
    f32x4* dst = reinterpret_cast<f32x4*>(&S[0][0]);
    for (int i = tid; i < 2 * 1024; i += 256) dst[i] = src[i];
    if (tid < 128) mu[tid >> 6][tid & 63] = mucl[(size_t)obj * 128 + tid];
  }
  const float* wrow = (hd ? w0y : w0x) + (size_t)ch * ldw + coloff;
  float w[64];
#pragma unroll
  for (int k4 = 0; k4 < 16; ++k4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + k4 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) w[k4 * 4 + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int cl = 0; cl < 2; ++cl) {  // this thread's 16 rows of w^T S w
    float quad = 0.f;
#pragma unroll 4
    for (int r = rq * 16; r < rq * 16 + 16; ++r) {
      const f32x4* Sr = reinterpret_cast<const f32x4*>(S[cl] + r * 64);
      float t = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {
        const f32x4 s4 = Sr[k4];
        t = fmaf(s4[0], w[k4 * 4], t);
        t = fmaf(s4[1], w[k4 * 4 + 1], t);
        t = fmaf(s4[2], w[k4 * 4 + 2], t);
        t = fmaf(s4[3], w[k4 * 4 + 3], t);
      }
      quad = fmaf(wrow[r], t, quad);  // (a runtime index into w[] would put the whole array in scratch)
    }
    qpart[cl][rq][cq] = quad;
  }
  __syncthreads();
  if (rq != 0) return;  // wave 0 (64 channels) finishes
  const float nobs = (float)N, npri = (float)M, ntot = 8.f * (float)(N + M);
  float a[2], v2[2];
#pragma unroll
  for (int cl = 0; cl < 2; ++cl) {
    float dotm = 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k) dotm = fmaf(w[k], mu[cl][k], dotm);
    a[cl] = dotm + bias0[((size_t)hd * 2 * B + (cl ? B + obj : obj)) * 256 + ch];
    v2[cl] = fmaxf((qpart[cl][0][cq] + qpart[cl][1][cq]) + (qpart[cl][2][cq] + qpart[cl][3][cq]), 0.f);
  }
  // group = 8 consecutive channels = 8 consecutive lanes
  float s1 = nobs * a[0] + npri * a[1];
  s1 += __shfl_xor(s1, 1);
  s1 += __shfl_xor(s1, 2);
  s1 += __shfl_xor(s1, 4);
  const float gmean = s1 / ntot;
  const float d0 = a[0] - gmean, d1 = a[1] - gmean;
  float m2 = (v2[0] + nobs * d0 * d0) + (v2[1] + npri * d1 * d1);
  m2 += __shfl_xor(m2, 1);
  m2 += __shfl_xor(m2, 2);
  m2 += __shfl_xor(m2, 4);
  const float rstd = 1.0f / sqrtf(m2 / ntot + 1e-5f);
  const float sc = rstd * (hd ? gamy : gamx)[ch];
  const float sh0 = (hd ? bety : betx)[ch] - gmean * sc;
#pragma unroll
  for (int cl = 0; cl < 2; ++cl) {
    const float b0 = bias0[((size_t)hd * 2 * B + (cl ? B + obj : obj)) * 256 + ch];
    float* o = aff + ((((size_t)obj * 2 + hd) * 2 + cl) * 2) * 256;
    o[ch] = sc;
    o[256 + ch] = fmaf(b0, sc, sh0);
  }
}
