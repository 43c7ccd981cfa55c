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
//   k_pf_moments     per cloud: shifted second moments G' and first moments s1 of pointfeat over all the cloud's points
//                    (fp32 MFMA; operands read one float per lane from the point-major LDS tile, which is what 32x32x2
//                    wants), tiles accumulated in order in the MFMA accumulators: deterministic
//   k_gn0_from_moments  per (object, head, 64 channels): the quadratic forms w^T S w, group statistics over both
//                    clouds, and the fused bias + GroupNorm affine table k_rot_l1 consumes.
// Used by every compute mode (the bf16 path through k_pf_moments_bf, which reads its bf16 pointfeat buffer: the statistics
// are then those of the fp32 W0 on the rounded pointfeat, ~1e-4 relative from what the bf16 layer 0 computes).
// Everything is fp32; sums of squares are always taken about a mean (tile mean, then cloud mean), never raw.
#pragma once

// One workgroup per CLOUD (grid 2B): streams the cloud's pointfeat rows tile by tile (next tile's global loads in
// flight under the current tile's MFMAs, two LDS buffers) and accumulates, about a fixed per-cloud shift c (the mean of
// the cloud's first tile - any fixed vector keeps the algebra exact, a nearby one keeps it well conditioned):
//     G'[i][j] = sum_p (pf_p[i] - c_i)(pf_p[j] - c_j)          (4 waves x one 32x32 block, K = all points of the cloud)
//     s1[i]    = sum_p (pf_p[i] - c_i)
// so that  mu = c + s1 / n  and the centred scatter  S = G' - s1 s1^T / n  (applied by k_gn0_from_moments while it loads
// G').  Replaces the per-tile Gram kernel + merge kernel of round 1 (136 MB of per-tile moments written and re-read).
// Fixed summation order: deterministic.
// The cloud's tiles are always summed in PF_NG = 4 contiguous groups, each into its own partial (G', s1): with
// gridDim.y == 1 one workgroup sweeps all four groups, with gridDim.y == 4 (small batches: a handful of clouds would
// leave the chip idle) four workgroups take one group each.  The consumer adds the four partials in a fixed order, so
// an object's result does not depend on how many objects share its batch.
#define PF_NG 4
#define PF_MOM_SMEM (2 * TP * LD64 + 4 * 64 + 64)  // floats of LDS
#ifndef PF_SPLIT_ALWAYS
#define PF_SPLIT_ALWAYS 0
#endif
// workgroups per cloud of k_pf_moments (gridDim.y): the four tile groups on four workgroups, or one workgroup for all
inline int pf_groups(int B) { return (PF_SPLIT_ALWAYS || 2 * B * PF_NG <= 256) ? PF_NG : 1; }
// body for cloud `cloud`, tile group by of gy (gy == 1: all four groups); 256 threads
// BF: pointfeat is the bf16 point-major buffer of the reduced-precision path ([point][8 chunks of 8 bf16], channels in
// k-slot order inside a chunk - catre_bf16.h); a thread then stages chunk tid & 7 of rows tid >> 3 and (tid >> 3) + 32.
template <bool BF = false>
__device__ __forceinline__ void pf_moments_body(const float* __restrict__ pointfeat,
                                                float* __restrict__ Gc /*[2B][PF_NG][4096]*/,
                                                float* __restrict__ s1c /*[2B][PF_NG][64]*/,
                                                float* __restrict__ shc /*[2B][64]*/, int B, int N, int M, int cloud,
                                                int by, int gy, float* lds /*PF_MOM_SMEM*/) {
  float(*pf)[TP * LD64] = reinterpret_cast<float(*)[TP * LD64]>(lds);
  float(*part)[64] = reinterpret_cast<float(*)[64]>(lds + 2 * TP * LD64);
  float* shift = lds + 2 * TP * LD64 + 4 * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = cloud < B ? N : M;
  const float* src = pointfeat + (cloud < B ? (size_t)cloud * N : (size_t)B * N + (size_t)(cloud - B) * M) * 64;
  const int nt = (n + TP - 1) / TP, tpg = (nt + PF_NG - 1) / PF_NG;  // tiles per group
  // staging map: thread -> rows {r0, r0+16, r0+32, r0+48}, float4 column c4 (coalesced 256 B per row)
  const int r0 = tid >> 4, c4 = tid & 15;
  f32x4 nxt[4];
  // BF: rows rb, rb + 32, chunk cb -> channels [chb, chb + 4) and [chb + 8, chb + 12)
  const int rb = tid >> 3, cb = tid & 7, chb = 16 * (cb >> 1) + 4 * (cb & 1);
  u32x4 nxb[2];
  const u32x4* srcb = reinterpret_cast<const u32x4*>(pointfeat) +
                      (cloud < B ? (size_t)cloud * N : (size_t)B * N + (size_t)(cloud - B) * M) * 8;
  auto fetch = [&](int t) {
    if constexpr (BF) {
#pragma unroll
      for (int u = 0; u < 2; ++u) nxb[u] = srcb[(size_t)min(t * TP + rb + 32 * u, n - 1) * 8 + cb];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = t * TP + r0 + 16 * u;
        nxt[u] = *reinterpret_cast<const f32x4*>(src + (size_t)min(p, n - 1) * 64 + c4 * 4);
      }
    }
  };
  // the fetched tile t -> LDS buffer, minus the shift (sa / sb: the shift of this thread's channels), rows past the cloud's
  // end as zeros when `mask`
  auto stage = [&](float* buf, int t, bool mask, const f32x4& sa, const f32x4& sb) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BF) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bool ok = !mask || t * TP + rb + 32 * u < n;
        const f32x4 a = {bf_lo(nxb[u][0]), bf_hi(nxb[u][0]), bf_lo(nxb[u][1]), bf_hi(nxb[u][1])};
        const f32x4 b = {bf_lo(nxb[u][2]), bf_hi(nxb[u][2]), bf_lo(nxb[u][3]), bf_hi(nxb[u][3])};
        *reinterpret_cast<f32x4*>(buf + (rb + 32 * u) * LD64 + chb) = ok ? a - sa : z;
        *reinterpret_cast<f32x4*>(buf + (rb + 32 * u) * LD64 + chb + 8) = ok ? b - sb : z;
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = !mask || t * TP + r0 + 16 * u < n;
        *reinterpret_cast<f32x4*>(buf + (r0 + 16 * u) * LD64 + c4 * 4) = ok ? nxt[u] - sa : z;
      }
    }
  };
  fetch(0);
  {  // the cloud's first tile as it is -> buffer 1 (scratch use); column means of its valid points = the shift
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    stage(pf[1], 0, false, z4, z4);
    __syncthreads();
    const int ch = tid & 63, q = tid >> 6, v0 = min(TP, n);
    float s = 0.f;
    for (int p = q * 16; p < q * 16 + 16; ++p) s += p < v0 ? pf[1][p * LD64 + ch] : 0.f;
    part[q][ch] = s;
    __syncthreads();
    if (tid < 64) {
      const float m = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) / (float)v0;
      shift[tid] = m;
      if (by == 0) shc[(size_t)cloud * 64 + tid] = m;
    }
    __syncthreads();
  }
  const f32x4 sh4 = *reinterpret_cast<const f32x4*>(shift + (BF ? chb : c4 * 4));
  const f32x4 sh4b = *reinterpret_cast<const f32x4*>(shift + (BF ? chb + 8 : c4 * 4));
  const int bi = wave >> 1, bj = wave & 1, i = lane & 31, h = lane >> 5;
  const int g_lo = gy == 1 ? 0 : by, g_hi = gy == 1 ? PF_NG : by + 1;
#pragma unroll 1
  for (int g = g_lo; g < g_hi; ++g) {
    const int t_lo = min(g * tpg, nt), t_hi = min(t_lo + tpg, nt);
    f32x16 acc = zero16();
    float colsum = 0.f;  // thread (channel tid & 63, point quarter tid >> 6)
    if (t_lo < t_hi && !(g == g_lo && t_lo == 0)) fetch(t_lo);  // (tile 0 is still in registers for the first group)
    for (int t = t_lo; t < t_hi; ++t) {
      float* buf = pf[t & 1];
      stage(buf, t, true, sh4, sh4b);  // shifted tile -> LDS, rows past the cloud's end as zeros (they add nothing to G' or s1)
      __syncthreads();  // also orders the previous tile's reads of the other buffer before its next overwrite
      if (t + 1 < t_hi) fetch(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      {
        const float* col = buf + (tid >> 6) * 16 * LD64 + (tid & 63);
#pragma unroll
        for (int p = 0; p < 16; ++p) colsum += col[p * LD64];
      }
      const float* pa = buf + h * LD64 + bi * 32 + i;
      const float* pb = buf + h * LD64 + bj * 32 + i;
#pragma unroll 8
      for (int k = 0; k < TP / 2; ++k) acc = mfma32(pa[2 * k * LD64], pb[2 * k * LD64], acc);  // lane half h: point 2k + h
    }
    float* out = Gc + ((size_t)cloud * PF_NG + g) * 4096;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = bi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      out[row * 64 + bj * 32 + i] = acc[reg];
    }
    __syncthreads();  // every wave is done with both tile buffers and with `part`
    part[tid >> 6][tid & 63] = colsum;
    __syncthreads();
    if (tid < 64)
      s1c[((size_t)cloud * PF_NG + g) * 64 + tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
  }
}

__global__ __launch_bounds__(256) void k_pf_moments(const float* __restrict__ pointfeat, float* __restrict__ Gc,
                                                    float* __restrict__ s1c, float* __restrict__ shc, int B, int N,
                                                    int M) {
  __shared__ __attribute__((aligned(16))) float lds[PF_MOM_SMEM];
  pf_moments_body(pointfeat, Gc, s1c, shc, B, N, M, blockIdx.x, blockIdx.y, gridDim.y, lds);
}
// the same moments from the bf16 pointfeat buffer of the reduced-precision path
__global__ __launch_bounds__(256) void k_pf_moments_bf(const u32x4* __restrict__ pointfeat, float* __restrict__ Gc,
                                                       float* __restrict__ s1c, float* __restrict__ shc, int B, int N,
                                                       int M) {
  __shared__ __attribute__((aligned(16))) float lds[PF_MOM_SMEM];
  pf_moments_body<true>(reinterpret_cast<const float*>(pointfeat), Gc, s1c, shc, B, N, M, blockIdx.x, blockIdx.y,
                        gridDim.y, lds);
}

// aff [B*2 (object, head)][2 (observed, prior)][2 (sc, sh)][256]: gelu_in = acc * sc + sh with sc = rstd * gamma,
// sh = beta + (bias0[cloud][ch] - mean) * sc.  Workgroup = object x `gridDim.y`
// shares of the 8 (head, 64-channel block) combinations: 8 shares on small grids (one combination each: a single object
// still gets eight CUs), one share on large ones (the object's two 64 x 64 scatter matrices are then fetched and centred
// once instead of eight times - at B = 256 that traffic, 268 MB through L2, was the kernel's time).
// Per combination the quadratic forms run on the matrix pipe: T = W_blk S (64 x 64 x 64 per cloud, wave = (32-channel
// half, cloud): 64 MFMAs), then q_c = sum_j T[c][j] W[c][j] with a butterfly over the 32 column lanes.  (Round 2 first
// did them as 16 LDS-broadcast rows x 64 FMAs per thread: 2 k ds_read_b128 per workgroup, LDS-bound, 16 us at B = 1.)
#define GN0_SLD 96  // pitch of S in LDS: the two k rows an MFMA B-fragment reads (h = 0 / 1) land 32 banks apart
#define GN0_WLD 66  // pitch of the staged weight block: lane (i, h) reads bank 2 i + h
__global__ __launch_bounds__(256) void k_gn0_from_moments(const float* __restrict__ Gc /*[2B][PF_NG][4096]*/,
                                                          const float* __restrict__ s1c, const float* __restrict__ shc,
                                                          const float* __restrict__ w0x, const float* __restrict__ w0y,
                                                          int ldw, int coloff, const float* __restrict__ bias0,
                                                          const float* __restrict__ gamx, const float* __restrict__ betx,
                                                          const float* __restrict__ gamy, const float* __restrict__ bety,
                                                          float* __restrict__ aff, int B, int N, int M,
                                                          float* __restrict__ stat0 = nullptr /*[2][B][32][2]*/) {
  __shared__ __attribute__((aligned(16))) float S[2][64 * GN0_SLD];
  __shared__ __attribute__((aligned(16))) float Wl[64 * GN0_WLD];
  __shared__ float mu[2][64];
  __shared__ float s1[2][64];
  __shared__ float qv[2][64];
  const int obj = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = 8 / gridDim.y, combo0 = blockIdx.y * per;
  if (tid < 128) {
    const int cl = tid >> 6, k = tid & 63;
    const size_t cloud = cl ? (size_t)B + obj : obj;
    const float* sp = s1c + cloud * PF_NG * 64 + k;
    const float v = (sp[0] + sp[64]) + (sp[128] + sp[192]);  // the four tile groups, fixed order
    s1[cl][k] = v;
    mu[cl][k] = shc[cloud * 64 + k] + v / (float)(cl ? M : N);
  }
  __syncthreads();
#pragma unroll
  for (int cl = 0; cl < 2; ++cl) {  // S = G' - s1 s1^T / n while loading
    const f32x4* src = reinterpret_cast<const f32x4*>(Gc + (cl ? (size_t)B + obj : (size_t)obj) * PF_NG * 4096);
    const float inv_n = 1.0f / (float)(cl ? M : N);
    f32x4 gp[4][PF_NG];  // all 16 loads of this thread in flight before the first use
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int p = 0; p < PF_NG; ++p) gp[u][p] = src[p * 1024 + tid + 256 * u];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e4 = tid + 256 * u, r = e4 >> 4, c0 = (e4 & 15) * 4;
      f32x4 g = (gp[u][0] + gp[u][1]) + (gp[u][2] + gp[u][3]);
      const float sr = s1[cl][r] * inv_n;
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q] = fmaf(-sr, s1[cl][c0 + q], g[q]);
      *reinterpret_cast<f32x4*>(&S[cl][r * GN0_SLD + c0]) = g;
    }
  }
  const int i = lane & 31, h = lane >> 5, cb = wave & 1, wcl = wave >> 1;
  const float nobs = (float)N, npri = (float)M, ntot = 8.f * (float)(N + M);
#pragma unroll 1
  for (int combo = combo0; combo < combo0 + per; ++combo) {
    const int hd = combo >> 2, bz = combo & 3;
    const float* wblk = (hd ? w0y : w0x) + (size_t)(bz * 64) * ldw + coloff;
    f32x4 wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e4 = tid + 256 * u;
      wv[u] = *reinterpret_cast<const f32x4*>(wblk + (size_t)(e4 >> 4) * ldw + (e4 & 15) * 4);
    }
    // the finishing wave's per-channel scalars, requested with the weights
    const int ch = bz * 64 + lane;
    float b0v[2] = {0.f, 0.f}, gam = 0.f, bet = 0.f;
    if (wave == 0) {
      b0v[0] = bias0[((size_t)hd * 2 * B + obj) * 256 + ch];
      b0v[1] = bias0[((size_t)hd * 2 * B + B + obj) * 256 + ch];
      gam = (hd ? gamy : gamx)[ch];
      bet = (hd ? bety : betx)[ch];
    }
    __syncthreads();  // S is complete (first pass) / the previous combination is done with Wl and qv
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e4 = tid + 256 * u;
      float* d = Wl + (e4 >> 4) * GN0_WLD + (e4 & 15) * 4;
      d[0] = wv[u][0];
      d[1] = wv[u][1];
      d[2] = wv[u][2];
      d[3] = wv[u][3];
    }
    __syncthreads();
    {  // wave (cb, wcl): T = W[cb*32.., :] S_wcl as two 32 x 32 blocks, K = 64
      f32x16 acc0 = zero16(), acc1 = zero16();
      const float* ap = Wl + (cb * 32 + i) * GN0_WLD + h;
      const float* bp = S[wcl] + h * GN0_SLD + i;
#pragma unroll 8
      for (int kk = 0; kk < 32; ++kk) {
        const float a = ap[2 * kk];
        acc0 = mfma32(a, bp[2 * kk * GN0_SLD], acc0);
        acc1 = mfma32(a, bp[2 * kk * GN0_SLD + 32], acc1);
      }
      // q_c = sum_j T[c][j] W[c][j]: this lane's column j = i (and 32 + i) of its 16 rows, then a butterfly over i
      float pr[16];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float* wr = Wl + (cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * GN0_WLD + i;
        pr[reg] = fmaf(acc1[reg], wr[32], acc0[reg] * wr[0]);
      }
#pragma unroll
      for (int m = 1; m < 32; m <<= 1)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) pr[reg] += __shfl_xor(pr[reg], m);
      if (i == 0) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) qv[wcl][cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h] = pr[reg];
      }
    }
    __syncthreads();
    if (wave == 0) {  // 64 channels finish
      float a[2], v2[2];
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        float dotm = 0.f;
        const float* wrow = Wl + lane * GN0_WLD;
#pragma unroll
        for (int k = 0; k < 64; ++k) dotm = fmaf(wrow[k], mu[cl][k], dotm);
        a[cl] = dotm + b0v[cl];
        v2[cl] = fmaxf(qv[cl][lane], 0.f);
      }
      // group = 8 consecutive channels = 8 consecutive lanes
      float gs = nobs * a[0] + npri * a[1];
      gs += __shfl_xor(gs, 1);
      gs += __shfl_xor(gs, 2);
      gs += __shfl_xor(gs, 4);
      const float gmean = gs / ntot;
      const float d0 = a[0] - gmean, d1 = a[1] - gmean;
      float m2 = (v2[0] + nobs * d0 * d0) + (v2[1] + npri * d1 * d1);
      m2 += __shfl_xor(m2, 1);
      m2 += __shfl_xor(m2, 2);
      m2 += __shfl_xor(m2, 4);
      const float rstd = 1.0f / sqrtf(m2 / ntot + 1e-5f);
      const float sc = rstd * gam;
      const float sh0 = bet - gmean * sc;
      if (stat0 && (lane & 7) == 0) {  // training forward: (mean, rstd) of y0 per (head, object, group) for the backward
        float* so = stat0 + (((size_t)hd * B + obj) * 32 + (ch >> 3)) * 2;
        so[0] = gmean;
        so[1] = rstd;
      }
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        float* o = aff + ((((size_t)obj * 2 + hd) * 2 + cl) * 2) * 256;
        o[ch] = sc;
        o[256 + ch] = fmaf(b0v[cl], sc, sh0);
      }
    }
  }
}
