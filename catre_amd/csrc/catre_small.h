// catre_small.h - the latency path of small batches (included by catre_kernels.hip after catre_gram.h).
//
// The reference's evaluator refines one image at a time: a handful of objects per call (catre_evaluator.py:292-311).
// There a refine iteration is a chain of ~22 dependent launches of 4-18 us each - what it costs is the NUMBER of
// launches on the critical path, not their work.  For 2B <= SMALL_ROWS clouds catre_refine_iter therefore runs
// 14 launches (B <= 2; 17 up to B = 8) instead of 22, every one the SAME arithmetic in the same order as the large-batch kernels (the bodies
// are shared device functions), so an object's result still does not depend on the batch it came in, bit for bit:
//
//   * up to SMALL_FOLD_ROWS clouds (B <= 2) no k_reduce_pm launches: a consumer of a cloud's pooled feature takes the
//     maximum over the cloud's tile partials itself while it gathers its input (max is exact) - fc1 of both STN tails
//     (k_linear_pm), the ts head's layer 0 and the rot heads' global halves (k_heads_a).  Beyond that the staging - every
//     one of fc1's 16 workgroups reads all R rows x 16 tiles - costs more than the launch it saves (B = 8: 20 us against
//     5 + 6), and the three reductions stay launches of their own;
//   * k_heads_a: what only needs the trunk's outputs runs side by side in ONE launch, on disjoint workgroups -
//     pointfeat moments | ts-head layer 0 | both rot heads' global halves;
//   * k_heads_d: GN1 finalize + k_rot_out (every workgroup merges its (object, head)'s tile partials from LDS first: at
//     B <= 8 that is cheaper than a launch; at B = 256 it was not, see catre_rot.h) | the rest of the ts head, whose
//     result is not needed before the pose update;
//   * k_finish_update: the rot heads' tile sums (k_rot_finish) and the pose update in one launch.
#pragma once

#define SMALL_ROWS 16            // clouds (2B) up to which the latency path is taken
#define SMALL_FOLD_ROWS 4        // clouds up to which the consumers of a pooled feature reduce the tile partials themselves
#define XS_LD (1024 + 4)         // LDS pitch of a staged pooled-feature row
#define LINPM_SMEM (LIN_WAVES * 16 * 64 + SMALL_ROWS * XS_LD)  // floats: k_linear partial blocks + staged X rows

// xs[r][0..1023] = max over the tiles of cloud r of pm[tile][c]  (r < R <= SMALL_ROWS), all threads of the workgroup
__device__ __forceinline__ void stage_cloud_max(const float* __restrict__ pm, float* xs, int R, int B, int N, int M,
                                                int nthreads, int rpt = 1 /*partial rows per tile*/) {
  const int TN = (N + TP - 1) / TP, TM = (M + TP - 1) / TP;
  for (int idx = threadIdx.x; idx < R * 256; idx += nthreads) {
    const int r = idx >> 8, c4 = idx & 255;
    const int nt = (r < B ? TN : TM) * rpt;
    const size_t row0 = (r < B ? (size_t)r * TN : (size_t)B * TN + (size_t)(r - B) * TM) * rpt;
    const f32x4* src = reinterpret_cast<const f32x4*>(pm + row0 * PMW) + c4;
    f32x4 m = src[0];
    int t = 1;
    for (; t + 14 < nt; t += 15) {  // 15 partial rows in flight together
      f32x4 v[15];
#pragma unroll
      for (int u = 0; u < 15; ++u) v[u] = src[(size_t)(t + u) * (PMW / 4)];
#pragma unroll
      for (int u = 0; u < 15; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], v[u][q]);
    }
    for (; t + 3 < nt; t += 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[(size_t)(t + u) * (PMW / 4)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], v[u][q]);
    }
    for (; t < nt; ++t) {
      const f32x4 v = src[(size_t)t * (PMW / 4)];
#pragma unroll
      for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], v[q]);
    }
    *reinterpret_cast<f32x4*>(xs + r * XS_LD + c4 * 4) = m;
  }
}

// y = act(max_tiles(pm) W^T + b): k_linear on the pooled feature of R <= SMALL_ROWS clouds without the k_reduce_pm launch
// in front of it.  grid (1, J / 32, 1 or 2); gridDim.z == 2 as in k_linear.
__global__ __launch_bounds__(64 * LIN_WAVES) void k_linear_pm(const float* __restrict__ pm, int B, int N, int M,
                                                               const float* __restrict__ W, int ldw,
                                                               const float* __restrict__ bias, float* __restrict__ Y,
                                                               int ldy, int R, int J, int relu,
                                                               const float* __restrict__ W_z1,
                                                               const float* __restrict__ bias_z1,
                                                               float* __restrict__ Y_z1) {
  __shared__ __attribute__((aligned(16))) float smem[LINPM_SMEM];
  if (blockIdx.z == 1) {
    W = W_z1;
    bias = bias_z1;
    Y = Y_z1;
  }
  float* xs = smem + LIN_WAVES * 16 * 64;
  linear_body(xs, XS_LD, W, ldw, bias, Y, ldy, R, J, 1024, relu, 0, 0, blockIdx.y,
              reinterpret_cast<float(*)[16][64]>(smem), [&] {
                stage_cloud_max(pm, xs, R, B, N, M, 64 * LIN_WAVES);
                __syncthreads();
              });
}

// ---- one launch per FC tail (B <= 8) ------------------------------------------------------------------------------------
// relu(fc1 1024 -> 512) -> relu(fc2 512 -> 256) -> fc3 256 -> k*k (+ I_k) of an STN (pointnet.py:31-40 / 64-77) for R <=
// SMALL_ROWS clouds as ONE launch instead of three dependent ones (3 x ~7 us of launch + drain at B = 1, a quarter of an
// iteration's kernel time): every workgroup runs k_linear's own body (same fragments, same MFMA sequence, same LDS
// reduction: same bits) on its share of a layer's 32-column blocks, and a device-wide barrier separates the layers.
// The barrier: arrival counters in the workspace, zeroed by the encoder kernel launched in front of this one (k_stn3d /
// k_stnkd, `zero_bar`); the grid - 16 workgroups for STN3d, 128 for STNkd's 4096 outputs - is far below the chip's
// resident-workgroup capacity, so every workgroup arrives whatever else is running.
// MEASURED SLOWER than the three launches it replaces (one object, K = 4: 0.643 vs 0.620 ms; B = 8: 1.521 vs 1.479 ms,
// profiles/r06_fc_tail_ab.jsonl): a device-wide barrier on this chip is an agent-scope release + acquire across eight
// XCDs with private L2s (write-back, then invalidate), and two of them cost more than the two launch boundaries they stand
// in for.  Kept as an opt-in form (CATRE_FC_TAIL=1 / catre_form_switch 4) with its bit-equality test; default: off.
__device__ __forceinline__ void fct_grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this workgroup's stores of the layer's outputs
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__global__ __launch_bounds__(64 * LIN_WAVES) void k_fc_tail(const float* __restrict__ pooled, const float* __restrict__ pm,
                                                             int B, int N, int M, const float* __restrict__ W1,
                                                             const float* __restrict__ b1, const float* __restrict__ W2,
                                                             const float* __restrict__ b2, const float* __restrict__ W3,
                                                             const float* __restrict__ b3, float* __restrict__ h1,
                                                             float* __restrict__ h2, float* __restrict__ out, int k, int R,
                                                             unsigned* __restrict__ bar) {
  __shared__ __attribute__((aligned(16))) float smem[LINPM_SMEM];
  float(*part)[16][64] = reinterpret_cast<float(*)[16][64]>(smem);
  float* xs = smem + LIN_WAVES * 16 * 64;
  const int G = gridDim.x;
  // fc1: 16 column blocks; with pm the workgroup pools the tile partials itself (k_linear_pm), else X = the pooled feature
  for (int by = blockIdx.x; by < 16; by += G) {
    if (pm)
      linear_body(xs, XS_LD, W1, 1024, b1, h1, 512, R, 512, 1024, 1, 0, 0, by, part, [&] {
        stage_cloud_max(pm, xs, R, B, N, M, 64 * LIN_WAVES);
        __syncthreads();
      });
    else
      linear_body(pooled, 1024, W1, 1024, b1, h1, 512, R, 512, 1024, 1, 0, 0, by, part, [] {});
    __syncthreads();
  }
  fct_grid_barrier(bar, G);
  for (int by = blockIdx.x; by < 8; by += G) {
    linear_body(h1, 512, W2, 512, b2, h2, 256, R, 256, 512, 1, 0, 0, by, part, [] {});
    __syncthreads();
  }
  fct_grid_barrier(bar + 1, G);
  const int nb3 = (k * k + 31) / 32;
  for (int by = blockIdx.x; by < nb3; by += G) {
    linear_body(h2, 256, W3, 256, b3, out, k * k, R, k * k, 256, 0, k, 0, by, part, [] {});
    __syncthreads();
  }
}

// ---- after the trunk, launch 1 of 5 ----------------------------------------------------------------------------------
struct HeadsAArgs {
  // pointfeat moments (k_pf_moments)
  const float* pointfeat;
  float *Gc, *s1c, *shc;
  // ts-head layer 0 (k_ts_l0)
  const float *pose, *scale, *W0T;
  float* tspart;
  int in_dim, with_kps, with_scale, with_trans;
  // rot heads' global halves (k_linear, gridDim.z == 2)
  const float *w0x, *b0x, *w0y, *b0y;
  float* bias0;
  const float* pm;     // tile partials: the consumers pool them themselves (R <= SMALL_FOLD_ROWS) ...
  const float* gfeat;  // ... or nullptr: the pooled features [2B][PMW] that a k_reduce_pm launch wrote
  int B, N, M;
  int rpt;          // partial rows per tile in pm (2 after k_trunk_h)
  int n_mom, n_ts;  // workgroups of the first two roles
};

#define HEADS_A_SMEM (LINPM_SMEM > PF_MOM_SMEM ? LINPM_SMEM : PF_MOM_SMEM)
__global__ __launch_bounds__(64 * LIN_WAVES) void k_heads_a(HeadsAArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[HEADS_A_SMEM];
  int j = blockIdx.x;
  if (j < A.n_mom) {
    if (threadIdx.x >= 256) return;  // whole waves leave: the role's barriers count the surviving ones
    pf_moments_body(A.pointfeat, A.Gc, A.s1c, A.shc, A.B, A.N, A.M, j / PF_NG, j % PF_NG, PF_NG, smem);
    return;
  }
  j -= A.n_mom;
  if (j < A.n_ts) {
    if (threadIdx.x >= 256) return;
    ts_l0_body(A.gfeat, A.pm, A.pose, A.scale, A.W0T, A.tspart, A.B, A.N, A.M, A.in_dim, A.with_kps, A.with_scale,
               A.with_trans, j / TS_KS, j % TS_KS, smem, A.rpt);
    return;
  }
  j -= A.n_ts;  // (head z, column block by) of bias0[z][cloud][:] = W0_z[:, :1024] g_cloud + b0_z
  const int z = j >> 3, by = j & 7, R = 2 * A.B;
  float* xs = smem + LIN_WAVES * 16 * 64;
  if (A.pm)
    linear_body(xs, XS_LD, z ? A.w0y : A.w0x, PMW, z ? A.b0y : A.b0x, A.bias0 + (size_t)z * R * 256, 256, R, 256, 1024, 0, 0,
                0, by, reinterpret_cast<float(*)[16][64]>(smem), [&] {
                  stage_cloud_max(A.pm, xs, R, A.B, A.N, A.M, 64 * LIN_WAVES, A.rpt);
                  __syncthreads();
                });
  else
    linear_body(A.gfeat, PMW, z ? A.w0y : A.w0x, PMW, z ? A.b0y : A.b0x, A.bias0 + (size_t)z * R * 256, 256, R, 256, 1024, 0,
                0, 0, by, reinterpret_cast<float(*)[16][64]>(smem), [] {});
}

// ---- after k_rot_l1: GN1 finalize + k_rot_out | rest of the ts head ---------------------------------------------------
struct HeadsDArgs {
  const float *y1, *gn1;
  const float *gam1x, *bet1x, *gam1y, *bet1y, *neckx, *necky, *wpx, *wpy;
  float* rpart;
  int B, N, M, rd;
  int n_rot;  // workgroups of the rot role: B * T * 2
  TsHeadArgs ts;
};

__global__ __launch_bounds__(1024) void k_heads_d(HeadsDArgs A) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // max(T * 64 + 64 + 16, TS_OB * 1280) floats
  const int j = blockIdx.x;
  if (j >= A.n_rot) {
    ts_head_body(A.ts, j - A.n_rot, sm);
    return;
  }
  if (threadIdx.x >= 256) return;
  const int TN = (A.N + TP - 1) / TP, T = TN + (A.M + TP - 1) / TP;
  const int hd = j & 1;
  const RotTile rt = rot_tile(j >> 1, A.B, A.N, A.M);
  float* sp = sm;              // [T][64] the (object, head)'s tile partials
  float* stat = sm + T * 64;   // [32][2]
  float(*red)[4] = reinterpret_cast<float(*)[4]>(stat + 64);
  const float* src = A.gn1 + ((size_t)rt.obj * 2 + hd) * T * 64;
  for (int i = threadIdx.x; i < T * 64; i += 256) sp[i] = src[i];
  __syncthreads();
  if (threadIdx.x < 32) {  // k_gn_finalize's merge, same order
    float mean, rstd;
    merge_gn(sp, threadIdx.x, T, TN, A.N, A.M, mean, rstd);
    stat[threadIdx.x * 2] = mean;
    stat[threadIdx.x * 2 + 1] = rstd;
  }
  __syncthreads();
  rot_out_body(A.y1, stat, A.gam1x, A.bet1x, A.gam1y, A.bet1y, A.neckx, A.necky, A.wpx, A.wpy, A.rpart, A.B, A.N, A.M, A.rd,
               rt, hd, red);
}

// ---- k_rot_finish + k_pose_update: 8 objects per 64-thread workgroup --------------------------------------------------
__global__ __launch_bounds__(64) void k_finish_update(const float* __restrict__ rpart, const float* __restrict__ neckbx,
                                                      const float* __restrict__ neckby, const float* __restrict__ sumwp,
                                                      const float* __restrict__ cpbx, const float* __restrict__ cpby,
                                                      int T, int rd, const float* __restrict__ dtr,
                                                      const float* __restrict__ dsr, const float* __restrict__ pose0,
                                                      const float* __restrict__ scale0,
                                                      const float* __restrict__ mean_scales, const float* __restrict__ Ks,
                                                      catre_opts o, float* __restrict__ pose_out,
                                                      float* __restrict__ scale_out, int B,
                                                      float* __restrict__ pose_echo = nullptr,
                                                      float* __restrict__ scale_echo = nullptr) {
  __shared__ float rot[8][8];
  const int tid = threadIdx.x, nv = 2 * rd;
  if (tid < 8 * nv) {
    const int ob = tid / nv, i = tid % nv, b = blockIdx.x * 8 + ob;
    if (b < B) {  // k_rot_finish: the tile sums in tile order, then neck bias and conv_p bias
      const int hd = i / rd, c = i % rd;
      const float* rp = rpart + ((size_t)b * 2 + hd) * T * 4 + c;
      float s = 0.f;
      int t = 0;
      for (; t + 8 <= T; t += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = rp[(t + u) * 4];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; t < T; ++t) s += rp[t * 4];
      const float nb = (hd ? neckby : neckbx)[c];
      const float* cpb = hd ? cpby : cpbx;
      s = fmaf(nb, sumwp[hd], s);
      if (cpb) s += cpb[0];
      rot[ob][i] = s;
    }
  }
  __syncthreads();
  const int b = blockIdx.x * 8 + tid;
  if (tid < 8 && b < B)
    pose_update_obj(rot[tid], dtr, dsr, pose0, scale0, mean_scales, Ks, o, pose_out, scale_out, b, pose_echo, scale_echo);
}

// ---- the trunk on HALF tiles ----------------------------------------------------------------------------------------
// k_trunk with 32 points per workgroup: (tile, half, part) -> the conv1-conv3 prologue, which every workgroup of a tile
// repeats, is half as long (conv3 alone is 14 us per 64-point tile), and RSH workgroups per half tile share conv4's
// output channels.  Used while tiles * 2 * RSH <= 256 (B <= 4 at N = M = 1024).  Every output element sees the same
// operands in the same K order as in k_trunk, the pooled maxima are maxima of the same values: same bits.  The partial
// maxima go to pm[(tile * 2 + half)][PMW]: consumers walk two rows per tile (rpt = 2).
#define HP 32
template <int RSH>
__global__ __launch_bounds__(512) void k_trunk_h(catre_points P, const float* __restrict__ trans3,
                                                 const float* __restrict__ trans64, const float* __restrict__ Wc1,
                                                 const float* __restrict__ bc1, const f32x4* __restrict__ wp2,
                                                 const float* __restrict__ b2, const f32x4* __restrict__ wp3,
                                                 const float* __restrict__ b3, const f32x4* __restrict__ wp4,
                                                 const float* __restrict__ b4, float* __restrict__ pm,
                                                 float* __restrict__ pointfeat, int B, int N, int M) {
  __shared__ __attribute__((aligned(16))) float smem[HP * 512 + HP * 128];
  float* h1 = smem;                          // [32][68]
  float* t64 = smem + HP * LD64;             // [64][64]
  float* pf = smem + HP * LD64 + 4096;       // [32][68]
  float* scratch = smem + 2 * HP * LD64 + 4096;  // [8][64]
  float* a3 = smem;                          // [32][512] swizzled
  float* a2 = smem + HP * 512;               // [32][128] swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x / (2 * RSH), rem = blockIdx.x % (2 * RSH), hh = rem / RSH, part = rem % RSH;
  const TileInfo ti = tile_info(tile, B, N, M);
  const int valid_h = min(HP, max(ti.valid - hh * HP, 0));  // real points of this half (its other rows are duplicates)
  const bool ft = trans64 != nullptr;
  const int n = lane & 31, h = lane >> 5;

  const int mblk2 = wave & 3;  // conv2 64->128: 4 m-blocks on waves 0-3
  GemmPipe<1, 1, false, false, 8, 4> g2;
  g2.prefetch(wp2 + (mblk2 * 8) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, mblk2 * 32, lane);
  {
    float x, y, z;
    load_point(P, ti, hh * HP + n, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    conv3_relu_row<4>(x, y, z, Wc1, bc1, wave * 8 + h * 4, h1 + n * LD64);
    if (ft) {
      const f32x4* src = reinterpret_cast<const f32x4*>(trans64 + (size_t)ti.cloud * 4096);
      f32x4* dst = reinterpret_cast<f32x4*>(t64);
      dst[tid] = src[tid];
      dst[tid + 512] = src[tid + 512];
    }
  }
  __syncthreads();
  if (ft) {
    if (wave < 2) {  // pointfeat[j][n] = sum_i T64[i][j] h1[i][n]: two m-blocks, one point block
      const int mblk = wave;
      f32x16 acc = zero16();
      const float* xr = h1 + n * LD64 + 4 * h;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        const f32x4 bx = *reinterpret_cast<const f32x4*>(xr + kc * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float a = t64[(kc * 8 + 4 * h + s) * 64 + mblk * 32 + n];
          acc = mfma32(a, bx[s], acc);
        }
      }
      f32x16 accs[1][1] = {{acc}};
      store_tile_lds<1, 1, false>(accs, pf, LD64, mblk * 32, nullptr, lane);
    }
    __syncthreads();
  } else {
    pf = h1;
  }
  // conv3 128->512: 16 m-blocks, two per wave
  GemmPipe<2, 1, false, true, 16, 3, 1> g3;
  g3.prefetch(wp3 + (wave * 2 * 16) * 64 + lane, 16 * 64);
  f32x4 bv3[2][4];
  load_bias_quads<2>(bv3, b3, wave * 64, lane);
  __builtin_amdgcn_sched_barrier(0);
  const int pf_row = tid >> 4, pf_c4 = tid & 15;
  f32x4 pf_out = {0.f, 0.f, 0.f, 0.f};
  float pf_max = 0.f;
  {
    if (pf_row < valid_h) pf_out = reinterpret_cast<const f32x4*>(pf + pf_row * LD64)[pf_c4];
    {  // max_n pointfeat over this half: wave w reduces points [4w, 4w+4) for channel `lane`
      const float* col = pf + (wave * 4) * LD64 + lane;
      float m = col[0];
#pragma unroll
      for (int p = 1; p < 4; ++p) m = fmaxf(m, col[p * LD64]);
      scratch[wave * 64 + lane] = m;
      __syncthreads();
      if (tid < 64) {
        m = scratch[tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, scratch[w * 64 + tid]);
        pf_max = m;
      }
    }
    if (wave < 4) {
      f32x16 acc[1][1] = {{zero16()}};
      g2.run(acc, pf, LD64, lane);
      store_tile_lds_pre<1, 1, true, true>(acc, a2, 128, mblk2 * 32, bv2, lane);
    }
  }
  __syncthreads();
  constexpr int MB4 = 4 / RSH;
  const int mb0 = part * (32 / RSH) + wave * MB4;
  GemmPipe<MB4, 1, true, true, 64, 3, 1> g4;
  g4.prefetch(wp4 + ((size_t)mb0 * 64) * 64 + lane, 64 * 64);
  float bl4[MB4];
  load_bias_lane<MB4>(bl4, b4, mb0 * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 acc3[2][1];
    acc3[0][0] = acc3[1][0] = zero16();
    g3.run(acc3, a2, 128, lane);
    store_tile_lds_pre<2, 1, true, true>(acc3, a3, 512, wave * 64, bv3, lane);
  }
  __syncthreads();
  float* pmrow = pm + ((size_t)tile * 2 + hh) * PMW;
  {
    float* dstbase = pointfeat + (ti.is_obs ? ((size_t)ti.obj * N + ti.p0 + hh * HP) * 64
                                            : ((size_t)B * N + (size_t)ti.obj * M + ti.p0 + hh * HP) * 64);
    if (part == 0 && pf_row < valid_h) reinterpret_cast<f32x4*>(dstbase + pf_row * 64)[pf_c4] = pf_out;
    if (part == 0 && tid < 64) pmrow[1024 + tid] = pf_max;
  }
  f32x16 acc4[MB4][1];
#pragma unroll
  for (int mb = 0; mb < MB4; ++mb) acc4[mb][0] = zero16();
  g4.run(acc4, a3, 512, lane);
  max_tile_store_pre<MB4, 1>(acc4, pmrow, mb0 * 32, bl4, false, lane);
}
