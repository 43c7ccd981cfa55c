// catre_kernels.hip - hand-written gfx950 kernels for CATRE's pose-refine hot path + their C ABI.
// Built with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see Makefile).
// Reference citations (file:line) are relative to the CATRE tree; see include/catre_hip.h.
#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <mutex>

#include "catre_device.h"

#include "../../include/catre_hip.h"

#define CATRE_VERSION_STR "catre_hip gfx950 r1"
#define PMW 1088  // row pitch of the tile-partial-max buffer: [1024 conv max | 64 pointfeat max]

// ------------------------------------------------------------------------------------------
// tile bookkeeping
// ------------------------------------------------------------------------------------------
struct TileInfo {
  int obj;     // object index b
  int cloud;   // b (observed) or B+b (prior)
  int is_obs;
  int p0;      // first point of the tile inside its cloud
  int valid;   // number of real points in the tile (1..64); the rest are duplicates of the last one
};

// Tiles are enumerated observed-first: bid in [0, B*TN) -> observed, then B*TM prior tiles.
__device__ __forceinline__ TileInfo tile_info(int bid, int B, int N, int M) {
  const int TN = (N + TP - 1) / TP, TM = (M + TP - 1) / TP;
  TileInfo ti;
  if (bid < B * TN) {
    ti.obj = bid / TN;
    ti.cloud = ti.obj;
    ti.is_obs = 1;
    ti.p0 = (bid % TN) * TP;
    ti.valid = min(TP, N - ti.p0);
  } else {
    const int r = bid - B * TN;
    ti.obj = r / TM;
    ti.cloud = B + ti.obj;
    ti.is_obs = 0;
    ti.p0 = (r % TM) * TP;
    ti.valid = min(TP, M - ti.p0);
  }
  return ti;
}

// a1, one point: x = pcl - t (observed; only when ZERO_CENTER_INPUT) / tfd_kps = R (kps * s) (+ t otherwise)
// (engine/batch_test.py:81-97, lib/pysixd/misc.py:1014-1026).  Shared by k_pose_apply and the on-the-fly form in
// load_point, so both produce the same bits.
__device__ __forceinline__ void pose_apply_point(const float* __restrict__ p, const float* __restrict__ sc, int is_obs,
                                                 int zero_center, float& x, float& y, float& z) {
  if (is_obs) {
    if (zero_center) {
      x -= p[3];
      y -= p[7];
      z -= p[11];
    }
  } else {
    const float sx = x * sc[0], sy = y * sc[1], sz = z * sc[2];      // misc.py:1017
    // misc.py:1021 (row . column); explicit fma chain so that every caller contracts it the same way
    float ox = fmaf(p[2], sz, fmaf(p[1], sy, p[0] * sx));
    float oy = fmaf(p[6], sz, fmaf(p[5], sy, p[4] * sx));
    float oz = fmaf(p[10], sz, fmaf(p[9], sy, p[8] * sx));
    if (!zero_center) {
      ox += p[3];
      oy += p[7];
      oz += p[11];
    }
    x = ox;
    y = oy;
    z = oz;
  }
}

__device__ __forceinline__ void load_point(const catre_points& P, const TileInfo& ti, int p, float& x, float& y,
                                           float& z) {
  const int pi = ti.p0 + min(p, ti.valid - 1);  // clamp: duplicates never change a max-pool
  const float* base;
  int64_t sc;
  if (ti.is_obs) {
    base = P.obs + ti.obj * P.obs_sb + pi * P.obs_sn;
    sc = P.obs_sc;
  } else {
    base = P.kps + ti.obj * P.kps_sb + pi * P.kps_sn;
    sc = P.kps_sc;
  }
  x = base[0];
  y = base[sc];
  z = base[2 * sc];
  if (P.apply_pose) pose_apply_point(P.pose + ti.obj * 12, P.scale + ti.obj * 3, ti.is_obs, P.zero_center, x, y, z);
}

// out[ch] = relu(W[ch][0..2] . (x,y,z) + b[ch]) for NCH consecutive channels starting at ch0 (wave-uniform)
template <int NCH>
__device__ __forceinline__ void conv3_relu_row(float x, float y, float z, const float* __restrict__ W,
                                               const float* __restrict__ b, int ch0, float* out_row) {
#pragma unroll
  for (int c4 = 0; c4 < NCH / 4; ++c4) {
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = ch0 + c4 * 4 + q;
      float t = b[ch];
      t = fmaf(W[ch * 3 + 0], x, t);
      t = fmaf(W[ch * 3 + 1], y, t);
      t = fmaf(W[ch * 3 + 2], z, t);
      v[q] = fmaxf(t, 0.f);
    }
    *reinterpret_cast<f32x4*>(out_row + ch0 + c4 * 4) = v;
  }
}

// x' = x^T T3 : u'[j] = sum_i u[i] * T[i][j]   (pointnet.py:100-102)
__device__ __forceinline__ void apply_t3(const float* __restrict__ T, float& x, float& y, float& z) {
  const float nx = fmaf(z, T[6], fmaf(y, T[3], x * T[0]));
  const float ny = fmaf(z, T[7], fmaf(y, T[4], x * T[1]));
  const float nz = fmaf(z, T[8], fmaf(y, T[5], x * T[2]));
  x = nx;
  y = ny;
  z = nz;
}

// ------------------------------------------------------------------------------------------
// weight packing (layout only, no arithmetic)
// ------------------------------------------------------------------------------------------
__global__ void k_pack_frag(const float* __restrict__ src, int ld, int coloff, int rows, int K, float* __restrict__ dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * K) return;
  const int s = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
  const int nkc = K / 8;
  const int kc = rest % nkc, mb = rest / nkc;
  const int row = mb * 32 + (lane & 31), col = kc * 8 + 4 * (lane >> 5) + s;
  dst[idx] = src[(size_t)row * ld + coloff + col];
}

// several fp32 fragment packs in one launch (a training step re-packs the eight encoder matrices after every optimizer
// step: eight 5 us launches on the critical path of each iteration)
#define PACK_MAX_JOBS 12
struct PackJobs {
  const float* src[PACK_MAX_JOBS];
  float* dst[PACK_MAX_JOBS];
  int ld[PACK_MAX_JOBS], coloff[PACK_MAX_JOBS], K[PACK_MAX_JOBS];
  int end[PACK_MAX_JOBS];  // exclusive prefix end of the job's elements (rows * K) in the launch's index space
  int n;
};
__global__ void k_pack_frag_multi(PackJobs J) {
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= J.end[J.n - 1]) return;
  int j = 0;
  while (gi >= J.end[j]) ++j;
  const int idx = gi - (j ? J.end[j - 1] : 0);
  const int s = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
  const int nkc = J.K[j] / 8;
  const int kc = rest % nkc, mb = rest / nkc;
  const int row = mb * 32 + (lane & 31), col = kc * 8 + 4 * (lane >> 5) + s;
  J.dst[j][idx] = J.src[j][(size_t)row * J.ld[j] + J.coloff[j] + col];
}

__global__ void k_pack_transpose(const float* __restrict__ src, int J, int K, float* __restrict__ dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // dst[k*J + j] = src[j*K + k]
  if (idx >= J * K) return;
  const int j = idx % J, k = idx / J;
  dst[idx] = src[(size_t)j * K + k];
}

__global__ void k_sum(const float* __restrict__ src, int n, float* __restrict__ dst) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += src[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dst[0] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------
// a1: pose-apply (engine/batch_test.py:81-97, lib/pysixd/misc.py:1014-1026)
// ------------------------------------------------------------------------------------------
__global__ void k_pose_apply(const float* __restrict__ pcl, const float* __restrict__ kps,
                             const float* __restrict__ pose, const float* __restrict__ scale, float* __restrict__ xo,
                             float* __restrict__ ko, int B, int N, int M, int zero_center) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total_obs = B * N, total = B * (N + M);
  if (i >= total) return;
  const bool obs = i < total_obs;
  const int j = obs ? i : i - total_obs;
  const int b = obs ? j / N : j / M;
  const float* s = (obs ? pcl : kps) + (size_t)j * 3;
  float x = s[0], y = s[1], z = s[2];
  pose_apply_point(pose + b * 12, scale + b * 3, obs, zero_center, x, y, z);
  float* o = (obs ? xo : ko) + (size_t)j * 3;
  o[0] = x;
  o[1] = y;
  o[2] = z;
}

// Experiment knob of the instrumented build (make TRACE=1; catre_debug_knob): the co-resident workgroups of the first
// dispatch round can be started `g_dephase_cycles` apart, to test whether pairs that begin in lock step keep stalling
// in their load / thin-layer prologues at the same time.  Product build: no code.
#ifdef CATRE_DEBUG_TRACE
__device__ int g_ablate = 0;  // knob 1 (instrumented build): bit 0 = no weight loads, bit 1 = no LDS fragment loads in the bf16 pair trunk's conv4
__device__ int g_dephase_cycles = 0;
__device__ __forceinline__ void debug_dephase(int first_round_blocks) {
  const int cyc = g_dephase_cycles;
  if (cyc <= 0 || (int)blockIdx.x >= first_round_blocks) return;
  __shared__ int odd_slot;
  if (threadIdx.x == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    odd_slot = hw & 1;  // wave slot on the SIMD: the two co-resident workgroups' first waves sit in different slots
  }
  __syncthreads();
  if (odd_slot) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while ((long long)(__builtin_readcyclecounter() - t0) < cyc) __builtin_amdgcn_s_sleep(32);
  }
  __syncthreads();
}
#else
__device__ __forceinline__ void debug_dephase(int) {}
#endif

// ------------------------------------------------------------------------------------------
// a2: STN3d conv stack 3->64->128->1024 (+ReLU) and per-tile max  (pointnet.py:24-28)
// 256 threads = 4 waves, 2 workgroups per CU.
// ------------------------------------------------------------------------------------------
#define LD64 68
#define LD128 132
#define LD256 260
#define FCT_BAR_WORDS 4  // arrival counters of k_fc_tail's device-wide barriers (zeroed by the encoder kernel in front of it)

// SAVE (training forward, RS == 1, N and M multiples of 64): additionally writes what the layer-wise backward reads -
// the activation images as point rows (cloud-major: B*N observed rows, then B*M prior rows) and, instead of the tile
// maxima alone, the per-tile (max, arg-max row) pairs of the pooled layer (pitch 1024, merged by k_maxpool_tiles).
struct TrainSave {
  float *s1, *s2, *s3, *s4, *s5;  // kernel-specific activation rows (see the launchers)
  float* pmax;                    // [tiles][1024]
  int* pidx;                      // [tiles][1024]
};

// ONE (full grids, RS == 1): a single workgroup per CU with ONE wave per SIMD, the last layer as an MB8 x NB2 wave tile in
// one pass (256 accumulators) - the power-limited sweep form of k_trunk4; same K order per output element: same bits.
template <int RS, bool SAVE = false, bool ONE = false>
__global__ __launch_bounds__(256, ONE ? 1 : 2) void k_stn3d(catre_points P, const float* __restrict__ W1,
                                               const float* __restrict__ b1, const f32x4* __restrict__ wp2,
                                               const float* __restrict__ b2, const f32x4* __restrict__ wp3,
                                               const float* __restrict__ b3, float* __restrict__ pm, int B, int N,
                                               int M, TrainSave sv = TrainSave{}, unsigned* __restrict__ zero_bar = nullptr) {
  __shared__ __attribute__((aligned(16))) float smem[TP * LD64 + TP * LD128];
  float* a1 = smem;
  float* a2 = smem + TP * LD64;
  const int tid = threadIdx.x, lane = tid & 63;
  if (zero_bar && blockIdx.x == 0 && tid < FCT_BAR_WORDS) zero_bar[tid] = 0u;  // the fused FC tail behind this launch (k_fc_tail) counts arrivals here
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x / RS, part = blockIdx.x % RS;
  const TileInfo ti = tile_info(tile, B, N, M);
  debug_dephase(512);

  // weights / biases of the first MFMA layer are requested before anything else
  GemmPipe<1, 2, false, false, 8, 3> g2;
  g2.prefetch(wp2 + (wave * 8) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, wave * 32, lane);
  {  // conv1 3->64 on the VALU: thread = (point, 16-channel group)
    float x, y, z;
    load_point(P, ti, lane, x, y, z);
    conv3_relu_row<16>(x, y, z, W1, b1, wave * 16, a1 + lane * LD64);
  }
  __syncthreads();
  const size_t row0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE && sv.s1) save_tile_rows<64, 256, false>(a1, LD64, sv.s1 + row0 * 64, tid);  // (no rows: the backward recomputes them, k_stn_recompute)
  // conv3 128->1024 + max: wave owns channels [wave*256, +256) in two passes of 4 m-blocks (RS = 1); RS workgroups
  // per tile: 8/RS m-blocks per wave from mb0 in one pass (see k_trunk)
  constexpr int MB3 = ONE ? 8 : RS == 8 ? 1 : RS == 4 ? 2 : 4;
  static_assert(!ONE || RS == 1, "ONE is the full-grid form");
  constexpr bool TWO = RS == 1 && !ONE;  // two passes of 4 m-blocks
  const int mb0 = part * (32 / RS) + wave * (8 / RS);
  GemmPipe<MB3, 2, true, false, 16, RS == 8 ? 3 : 2, 1> g3a, g3b;
  float bl[2][MB3];
  g3a.prefetch(wp3 + ((size_t)mb0 * 16) * 64 + lane, 16 * 64);
  load_bias_lane<MB3>(bl[0], b3, mb0 * 32, lane);
  if (TWO) load_bias_lane<MB3>(bl[1], b3, (mb0 + 4) * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {  // conv2 64->128: wave -> m-block `wave`, both point blocks
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g2.run(acc, a1, LD64, lane);
    store_tile_lds_pre<1, 2, true, false>(acc, a2, LD128, wave * 32, bv2, lane);
  }
  __syncthreads();
  if (SAVE && sv.s2) save_tile_rows<128, 256, false>(a2, LD128, sv.s2 + row0 * 128, tid);
  float* out = pm + (size_t)tile * PMW;
  {
    f32x16 acc[MB3][2];
#pragma unroll
    for (int mb = 0; mb < MB3; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3a.run(acc, a2, LD128, lane);
    if (TWO) g3b.prefetch(wp3 + ((size_t)(mb0 + 4) * 16) * 64 + lane, 16 * 64);
    if (SAVE)
      argmax_tile_store<MB3, 2>(acc, sv.pmax + (size_t)tile * 1024, sv.pidx + (size_t)tile * 1024, mb0 * 32, bl[0],
                                (int)row0, lane);
    else
      max_tile_store_pre<MB3, 2>(acc, out, mb0 * 32, bl[0], true, lane);
  }
  if (TWO) {
    f32x16 acc[MB3][2];
#pragma unroll
    for (int mb = 0; mb < MB3; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3b.run(acc, a2, LD128, lane);
    if (SAVE)
      argmax_tile_store<MB3, 2>(acc, sv.pmax + (size_t)tile * 1024, sv.pidx + (size_t)tile * 1024, (mb0 + 4) * 32, bl[1],
                                (int)row0, lane);
    else
      max_tile_store_pre<MB3, 2>(acc, out, (mb0 + 4) * 32, bl[1], true, lane);
  }
}

// ------------------------------------------------------------------------------------------
// a3+a4: x' = x T3, relu(conv1), STNkd conv stack 64->64->128->1024 (+ReLU), per-tile max
// (pointnet.py:98-103, 57-61).  256 threads, 2 workgroups per CU.
// ------------------------------------------------------------------------------------------
template <int RS, bool SAVE = false, bool ONE = false>
__global__ __launch_bounds__(256, ONE ? 1 : 2) void k_stnkd(catre_points P, const float* __restrict__ trans3,
                                               const float* __restrict__ Wc1, const float* __restrict__ bc1,
                                               const f32x4* __restrict__ wpf1, const float* __restrict__ bf1,
                                               const f32x4* __restrict__ wpf2, const float* __restrict__ bf2,
                                               const f32x4* __restrict__ wpf3, const float* __restrict__ bf3,
                                               float* __restrict__ pm, int B, int N, int M,
                                               TrainSave sv = TrainSave{}, unsigned* __restrict__ zero_bar = nullptr) {
  __shared__ __attribute__((aligned(16))) float smem[2 * TP * LD64 + TP * LD128];
  float* h1 = smem;
  float* f1 = smem + TP * LD64;
  float* f2 = smem + 2 * TP * LD64;
  const int tid = threadIdx.x, lane = tid & 63;
  if (zero_bar && blockIdx.x == 0 && tid < FCT_BAR_WORDS) zero_bar[tid] = 0u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x / RS, part = blockIdx.x % RS;
  const TileInfo ti = tile_info(tile, B, N, M);
  debug_dephase(512);

  const int mblk1 = wave >> 1, nb1 = wave & 1;
  GemmPipe<1, 1, false, false, 8, 4> g1;
  g1.prefetch(wpf1 + (mblk1 * 8) * 64 + lane, 0);
  f32x4 bv1[1][4];
  load_bias_quads<1>(bv1, bf1, mblk1 * 32, lane);
  {
    float x, y, z;
    load_point(P, ti, lane, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    conv3_relu_row<16>(x, y, z, Wc1, bc1, wave * 16, h1 + lane * LD64);
  }
  __syncthreads();
  GemmPipe<1, 2, false, false, 8, 3> g2;
  g2.prefetch(wpf2 + (wave * 8) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, bf2, wave * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {  // fstn.conv1 64->64: 2 m-blocks x 2 point blocks, one per wave
    f32x16 acc[1][1] = {{zero16()}};
    g1.run(acc, h1 + nb1 * 32 * LD64, LD64, lane);
    store_tile_lds_pre<1, 1, true, false>(acc, f1 + nb1 * 32 * LD64, LD64, mblk1 * 32, bv1, lane);
  }
  __syncthreads();
  const size_t row0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE && sv.s1) save_tile_rows<64, 256, false>(f1, LD64, sv.s1 + row0 * 64, tid);
  constexpr int MB3 = ONE ? 8 : RS == 8 ? 1 : RS == 4 ? 2 : 4;
  static_assert(!ONE || RS == 1, "ONE is the full-grid form");
  constexpr bool TWO = RS == 1 && !ONE;  // two passes of 4 m-blocks
  const int mb0 = part * (32 / RS) + wave * (8 / RS);
  GemmPipe<MB3, 2, true, false, 16, RS == 8 ? 3 : 2, 1> g3a, g3b;
  float bl[2][MB3];
  g3a.prefetch(wpf3 + ((size_t)mb0 * 16) * 64 + lane, 16 * 64);
  load_bias_lane<MB3>(bl[0], bf3, mb0 * 32, lane);
  if (TWO) load_bias_lane<MB3>(bl[1], bf3, (mb0 + 4) * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {  // fstn.conv2 64->128
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g2.run(acc, f1, LD64, lane);
    store_tile_lds_pre<1, 2, true, false>(acc, f2, LD128, wave * 32, bv2, lane);
  }
  __syncthreads();
  if (SAVE && sv.s2) save_tile_rows<128, 256, false>(f2, LD128, sv.s2 + row0 * 128, tid);
  float* out = pm + (size_t)tile * PMW;
  {  // fstn.conv3 128->1024 + max, two passes of 4 m-blocks (RS = 1) or one pass of 8/RS
    f32x16 acc[MB3][2];
#pragma unroll
    for (int mb = 0; mb < MB3; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3a.run(acc, f2, LD128, lane);
    if (TWO) g3b.prefetch(wpf3 + ((size_t)(mb0 + 4) * 16) * 64 + lane, 16 * 64);
    if (SAVE)
      argmax_tile_store<MB3, 2>(acc, sv.pmax + (size_t)tile * 1024, sv.pidx + (size_t)tile * 1024, mb0 * 32, bl[0],
                                (int)row0, lane);
    else
      max_tile_store_pre<MB3, 2>(acc, out, mb0 * 32, bl[0], true, lane);
  }
  if (TWO) {
    f32x16 acc[MB3][2];
#pragma unroll
    for (int mb = 0; mb < MB3; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3b.run(acc, f2, LD128, lane);
    if (SAVE)
      argmax_tile_store<MB3, 2>(acc, sv.pmax + (size_t)tile * 1024, sv.pidx + (size_t)tile * 1024, (mb0 + 4) * 32, bl[1],
                                (int)row0, lane);
    else
      max_tile_store_pre<MB3, 2>(acc, out, (mb0 + 4) * 32, bl[1], true, lane);
  }
}

// ------------------------------------------------------------------------------------------
// The ONE form of the two STN kernels on PAIRS of tiles (128 points per workgroup): with one wave per SIMD nothing runs
// under a barrier, a point load or the dependent MFMA chains of the thin layers, so the prologue is done once for two tiles
// (four independent accumulators in conv2 instead of two, half the barriers and load round trips per tile) and the last
// layer sweeps the two tiles one after the other on the same 256 accumulators.  Every output element sees the operands of
// k_stn3d / k_stnkd in the same K order: same bits.  A cloud with an odd number of tiles ends in a pair of one tile: its
// second half holds clamped duplicates and is not swept.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_info32(int bid, int B, int N, int M, TileInfo& ti, int& tile0) {
  const int TN = (N + TP - 1) / TP, TM = (M + TP - 1) / TP, PN = (TN + 1) / 2, PM_ = (TM + 1) / 2;
  if (bid < B * PN) {
    ti.obj = bid / PN;
    ti.cloud = ti.obj;
    ti.is_obs = 1;
    const int pi = bid % PN;
    ti.p0 = pi * 2 * TP;
    ti.valid = min(2 * TP, N - ti.p0);
    tile0 = ti.obj * TN + 2 * pi;
  } else {
    const int r = bid - B * PN;
    ti.obj = r / PM_;
    ti.cloud = B + ti.obj;
    ti.is_obs = 0;
    const int pi = r % PM_;
    ti.p0 = pi * 2 * TP;
    ti.valid = min(2 * TP, M - ti.p0);
    tile0 = B * TN + ti.obj * TM + 2 * pi;
  }
}

// ARGMAX (training forward without activation saves: the row-sparse backward rebuilds the two thin layers on its live rows,
// k_stn_recompute): per tile the (max + bias, arg-max row) pairs of the pooled layer instead of the tile maxima.
// cloud-major row (B*N observed rows, then B*M prior rows) of the first point of a pair
__device__ __forceinline__ int pair_row0(const TileInfo& ti, int B, int N, int M) {
  return (ti.is_obs ? ti.obj * N : B * N + ti.obj * M) + ti.p0;
}

template <bool ARGMAX = false>
__global__ __launch_bounds__(256) void k_stn3d_pair(catre_points P, const float* __restrict__ W1,
                                                    const float* __restrict__ b1, const f32x4* __restrict__ wp2,
                                                    const float* __restrict__ b2, const f32x4* __restrict__ wp3,
                                                    const float* __restrict__ b3, float* __restrict__ pm, int B, int N,
                                                    int M, TrainSave sv = TrainSave{}) {
  __shared__ __attribute__((aligned(16))) float smem[2 * TP * LD64 + 2 * TP * LD128];
  float* a1 = smem;                   // [128][68]
  float* a2 = smem + 2 * TP * LD64;   // [128][132]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  TileInfo ti;
  int tile0;
  pair_info32(blockIdx.x, B, N, M, ti, tile0);
  const bool has2 = ti.valid > TP;

  GemmPipe<1, 4, false, false, 8, 3> g2;  // conv2 64->128: wave -> m-block `wave`, all four point blocks
  g2.prefetch(wp2 + (wave * 8) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, wave * 32, lane);
  {  // conv1 3->64 on the VALU: thread = (point, 32-channel half)
    const int p = (wave & 1) * TP + lane;
    float x, y, z;
    load_point(P, ti, p, x, y, z);
    conv3_relu_row<32>(x, y, z, W1, b1, (wave >> 1) * 32, a1 + p * LD64);
  }
  __syncthreads();
  const int mb0 = wave * 8;
  float bl[8];
  GemmPipe<8, 2, true, false, 16, 2, 1> g3;  // first sweep: its first weight chunks are requested in front of conv2
  g3.prefetch(wp3 + ((size_t)mb0 * 16) * 64 + lane, 16 * 64);
  load_bias_lane<8>(bl, b3, mb0 * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 acc[1][4] = {{zero16(), zero16(), zero16(), zero16()}};
    g2.run(acc, a1, LD64, lane);
    store_tile_lds_pre<1, 4, true, false>(acc, a2, LD128, wave * 32, bv2, lane);
  }
  __syncthreads();
  {  // the two tiles one after the other on the same accumulators.  (No weight ring carried across the epilogue - the 256
     // maxima pass through the VGPRs there and 64 prefetched registers next to them spill; a runtime loop over the two
     // sweeps spills as well.)
    f32x16 acc[8][2];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3.run(acc, a2, LD128, lane);
    if constexpr (ARGMAX)
      argmax_tile_store<8, 2>(acc, sv.pmax + (size_t)tile0 * 1024, sv.pidx + (size_t)tile0 * 1024, mb0 * 32, bl,
                              pair_row0(ti, B, N, M), lane);
    else
      max_tile_store_pre<8, 2>(acc, pm + (size_t)tile0 * PMW, mb0 * 32, bl, true, lane);
  }
  if (has2) {
    f32x16 acc[8][2];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    // opaque: otherwise the second sweep shares the first one's 32 fragment base pointers, which then stay live across the
    // epilogue in between - where the 256 maxima pass through the VGPRs - and spill
    unsigned lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    gemm_core<8, 2, true, false, 16, 2, 1>(acc, wp3 + ((size_t)mb0 * 16) * 64 + lane_o, 16 * 64, a2 + TP * LD128, LD128, lane);
    if constexpr (ARGMAX)
      argmax_tile_store<8, 2>(acc, sv.pmax + (size_t)(tile0 + 1) * 1024, sv.pidx + (size_t)(tile0 + 1) * 1024, mb0 * 32, bl,
                              pair_row0(ti, B, N, M) + TP, lane);
    else
      max_tile_store_pre<8, 2>(acc, pm + (size_t)(tile0 + 1) * PMW, mb0 * 32, bl, true, lane);
  }
}

template <bool ARGMAX = false>
__global__ __launch_bounds__(256) void k_stnkd_pair(catre_points P, const float* __restrict__ trans3,
                                                    const float* __restrict__ Wc1, const float* __restrict__ bc1,
                                                    const f32x4* __restrict__ wpf1, const float* __restrict__ bf1,
                                                    const f32x4* __restrict__ wpf2, const float* __restrict__ bf2,
                                                    const f32x4* __restrict__ wpf3, const float* __restrict__ bf3,
                                                    float* __restrict__ pm, int B, int N, int M,
                                                    TrainSave sv = TrainSave{}) {
  __shared__ __attribute__((aligned(16))) float smem[4 * TP * LD64 + 2 * TP * LD128];
  float* h1 = smem;                   // [128][68]
  float* f1 = smem + 2 * TP * LD64;   // [128][68]
  float* f2 = smem + 4 * TP * LD64;   // [128][132]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  TileInfo ti;
  int tile0;
  pair_info32(blockIdx.x, B, N, M, ti, tile0);
  const bool has2 = ti.valid > TP;

  const int mblk1 = wave >> 1, half1 = wave & 1;
  GemmPipe<1, 2, false, false, 8, 4> g1;  // fstn.conv1 64->64: wave -> (m-block, tile of the pair)
  g1.prefetch(wpf1 + (mblk1 * 8) * 64 + lane, 0);
  f32x4 bv1[1][4];
  load_bias_quads<1>(bv1, bf1, mblk1 * 32, lane);
  {
    const int p = (wave & 1) * TP + lane;
    float x, y, z;
    load_point(P, ti, p, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    conv3_relu_row<32>(x, y, z, Wc1, bc1, (wave >> 1) * 32, h1 + p * LD64);
  }
  __syncthreads();
  GemmPipe<1, 4, false, false, 8, 3> g2;
  g2.prefetch(wpf2 + (wave * 8) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, bf2, wave * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g1.run(acc, h1 + half1 * TP * LD64, LD64, lane);
    store_tile_lds_pre<1, 2, true, false>(acc, f1 + half1 * TP * LD64, LD64, mblk1 * 32, bv1, lane);
  }
  __syncthreads();
  const int mb0 = wave * 8;
  float bl[8];
  GemmPipe<8, 2, true, false, 16, 2, 1> g3;  // first sweep: its first weight chunks are requested in front of conv2
  g3.prefetch(wpf3 + ((size_t)mb0 * 16) * 64 + lane, 16 * 64);
  load_bias_lane<8>(bl, bf3, mb0 * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 acc[1][4] = {{zero16(), zero16(), zero16(), zero16()}};
    g2.run(acc, f1, LD64, lane);
    store_tile_lds_pre<1, 4, true, false>(acc, f2, LD128, wave * 32, bv2, lane);
  }
  __syncthreads();
  {  // the two tiles one after the other on the same accumulators.  (No weight ring carried across the epilogue - the 256
     // maxima pass through the VGPRs there and 64 prefetched registers next to them spill; a runtime loop over the two
     // sweeps spills as well.)
    f32x16 acc[8][2];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3.run(acc, f2, LD128, lane);
    if constexpr (ARGMAX)
      argmax_tile_store<8, 2>(acc, sv.pmax + (size_t)tile0 * 1024, sv.pidx + (size_t)tile0 * 1024, mb0 * 32, bl,
                              pair_row0(ti, B, N, M), lane);
    else
      max_tile_store_pre<8, 2>(acc, pm + (size_t)tile0 * PMW, mb0 * 32, bl, true, lane);
  }
  if (has2) {
    f32x16 acc[8][2];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    // opaque: otherwise the second sweep shares the first one's 32 fragment base pointers, which then stay live across the
    // epilogue in between - where the 256 maxima pass through the VGPRs - and spill
    unsigned lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    gemm_core<8, 2, true, false, 16, 2, 1>(acc, wpf3 + ((size_t)mb0 * 16) * 64 + lane_o, 16 * 64, f2 + TP * LD128, LD128, lane);
    if constexpr (ARGMAX)
      argmax_tile_store<8, 2>(acc, sv.pmax + (size_t)(tile0 + 1) * 1024, sv.pidx + (size_t)(tile0 + 1) * 1024, mb0 * 32, bl,
                              pair_row0(ti, B, N, M) + TP, lane);
    else
      max_tile_store_pre<8, 2>(acc, pm + (size_t)(tile0 + 1) * PMW, mb0 * 32, bl, true, lane);
  }
}

// ------------------------------------------------------------------------------------------
// a3+a5: trunk.  x' = x T3 -> relu(conv1) -> pointfeat = h1^T T64 -> relu(conv2) -> relu(conv3)
// -> conv4 -> max  (pointnet.py:98-116).  512 threads = 8 waves, 1 workgroup per CU.
// The whole 64-point x 512-channel conv4 input lives in LDS (128 KiB, XOR-swizzled, no padding) next to
// the conv3 input (32 KiB): exactly the 160 KiB of a CU.  conv4 is then ONE K=512 sweep per wave
// (1024 ch x 64 pts = 128 accumulator VGPRs per lane over 8 waves) with no barrier inside.
// ------------------------------------------------------------------------------------------
// Small grids (tiles * RS <= #CUs): RS workgroups share a tile - each repeats the cheap prologue (conv1-conv3, 12 % of
// the FLOPs) and sweeps 1/RS of conv4's output channels, so a handful of objects still spreads over the chip.  Every
// output channel sees the same K order for any RS: results do not depend on it.
// ------------------------------------------------------------------------------------------
#define TRUNK_SMEM (TP * 512 + TP * 128)

template <int RS, bool SAVE = false>
__global__ __launch_bounds__(512) void k_trunk(catre_points P, const float* __restrict__ trans3,
                                               const float* __restrict__ trans64, const float* __restrict__ Wc1,
                                               const float* __restrict__ bc1, const f32x4* __restrict__ wp2,
                                               const float* __restrict__ b2, const f32x4* __restrict__ wp3,
                                               const float* __restrict__ b3, const f32x4* __restrict__ wp4,
                                               const float* __restrict__ b4, float* __restrict__ pm,
                                               float* __restrict__ pointfeat, int B, int N, int M,
                                               unsigned long long* __restrict__ trace, TrainSave sv = TrainSave{}) {
  __shared__ __attribute__((aligned(16))) float smem[TRUNK_SMEM];
#define TRUNK_STAMP(i)                                                                     \
  do {                                                                                     \
    if (CATRE_TRACE_ON && trace && lane == 0) trace[((size_t)tile * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
  // phase-1/2 buffers alias the conv4 input image a3 (dead before a3 is first written)
  float* h1 = smem;                      // [64][68]
  float* t64 = smem + TP * LD64;         // [64][64]
  float* pf = smem + TP * LD64 + 4096;   // [64][68]
  float* a3 = smem;                      // [64][512] swizzled
  float* a2 = smem + TP * 512;           // [64][128] swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x / RS, part = blockIdx.x % RS;
  const TileInfo ti = tile_info(tile, B, N, M);
  const bool ft = trans64 != nullptr;
  TRUNK_STAMP(0);

  // conv2 64->128: 4 m-blocks x 2 point blocks over 8 waves.  Its first weight chunks and bias are
  // requested now, long before they are needed.
  const int mblk2 = wave >> 1, nb2 = wave & 1;
  GemmPipe<1, 1, false, false, 8, 4> g2;
  g2.prefetch(wp2 + (mblk2 * 8) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, mblk2 * 32, lane);
  {
    float x, y, z;
    load_point(P, ti, lane, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    if (SAVE && wave == 0) {  // x' = x T3 as a zero-padded 8-wide row: the input of the conv1 row GEMM in the backward
      float* xr = sv.s1 + ((ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0 + lane) * 8;
      *reinterpret_cast<f32x4*>(xr) = f32x4{x, y, z, 0.f};
      *reinterpret_cast<f32x4*>(xr + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    conv3_relu_row<8>(x, y, z, Wc1, bc1, wave * 8, h1 + lane * LD64);
    if (ft) {
      const f32x4* src = reinterpret_cast<const f32x4*>(trans64 + (size_t)ti.cloud * 4096);
      f32x4* dst = reinterpret_cast<f32x4*>(t64);
      dst[tid] = src[tid];
      dst[tid + 512] = src[tid + 512];
    }
  }
  __syncthreads();
  TRUNK_STAMP(1);
  const size_t trow0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE) save_tile_rows<64, 512, false>(h1, LD64, sv.s2 + trow0 * 64, tid);
  if (ft) {
    if (wave < 4) {  // pointfeat[j][n] = sum_i T64[i][j] h1[i][n]  (pointnet.py:107-109); A operand from LDS
      const int mblk = wave >> 1, nb = wave & 1;
      const int n = lane & 31, h = lane >> 5;
      f32x16 acc = zero16();
      const float* xr = h1 + (nb * 32 + n) * LD64 + 4 * h;
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
      store_tile_lds<1, 1, false>(accs, pf + nb * 32 * LD64, LD64, mblk * 32, nullptr, lane);
    }
    __syncthreads();
  } else {
    pf = h1;
  }
  TRUNK_STAMP(2);
  // conv3 128->512: 16 m-blocks, two per wave; request its first weight chunks and bias now
  GemmPipe<2, 2, false, true, 16, 3, 1> g3;
  g3.prefetch(wp3 + (wave * 2 * 16) * 64 + lane, 16 * 64);
  f32x4 bv3[2][4];
  load_bias_quads<2>(bv3, b3, wave * 64, lane);
  __builtin_amdgcn_sched_barrier(0);
  const int pf_row = tid >> 3, pf_c4 = tid & 7;
  f32x4 pf_out0 = {0.f, 0.f, 0.f, 0.f}, pf_out1 = {0.f, 0.f, 0.f, 0.f};
  float pf_max = 0.f;
  {
    // pointfeat tile -> registers now, -> HBM after the last barrier (a __syncthreads waits for this wave's
    // outstanding stores, so storing here would park all 8 waves behind a write acknowledge)
    if (pf_row < ti.valid) {
      const f32x4* s = reinterpret_cast<const f32x4*>(pf + pf_row * LD64);
      pf_out0 = s[pf_c4];
      pf_out1 = s[pf_c4 + 8];
    }
    {  // max_n pointfeat (second half of flat_pcl_feat, CATRE_disR_shared.py:69): wave w reduces points
       // [8w, 8w+8) for channel `lane`, then 64 threads merge the 8 partials (scratch sits in the still unused
       // tail of the a3 region; the extra barrier is cheap, all waves arrive together)
      float* scratch = smem + 2 * TP * LD64 + 4096;  // [8][64]
      const float* col = pf + (wave * 8) * LD64 + lane;
      float m = col[0];
#pragma unroll
      for (int p = 1; p < 8; ++p) m = fmaxf(m, col[p * LD64]);
      scratch[wave * 64 + lane] = m;
      __syncthreads();
      if (tid < 64) {
        m = scratch[tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, scratch[w * 64 + tid]);
        pf_max = m;
      }
    }
    f32x16 acc[1][1] = {{zero16()}};
    g2.run(acc, pf + nb2 * 32 * LD64, LD64, lane);
    store_tile_lds_pre<1, 1, true, true>(acc, a2 + nb2 * 32 * 128, 128, mblk2 * 32, bv2, lane);
  }
  __syncthreads();
  TRUNK_STAMP(3);
  if (SAVE) save_tile_rows<128, 512, true>(a2, 128, sv.s3 + trow0 * 128, tid);
  // conv4 512->1024: wave owns MB4 m-blocks from mb0 (RS = 1: out channels [wave*128, +128)); first weight chunks +
  // bias requested now
  // RS = 8 (a single object): four m-blocks per workgroup, wave -> (m-block wave / 2, point block wave % 2); the two point
  // blocks' maxima of a channel meet in LDS before the store (max is exact: same result)
  constexpr int MB4 = RS == 8 ? 1 : 4 / RS, NB4 = RS == 8 ? 1 : 2;
  const int mb0 = RS == 8 ? part * 4 + (wave >> 1) : part * (32 / RS) + wave * MB4;
  GemmPipe<MB4, NB4, true, true, 64, RS == 1 ? 2 : 3, 1> g4;
  g4.prefetch(wp4 + ((size_t)mb0 * 64) * 64 + lane, 64 * 64);
  float bl4[MB4];
  load_bias_lane<MB4>(bl4, b4, mb0 * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 acc3[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc3[mb][0] = acc3[mb][1] = zero16();
    g3.run(acc3, a2, 128, lane);
    store_tile_lds_pre<2, 2, true, true>(acc3, a3, 512, wave * 64, bv3, lane);
    TRUNK_STAMP(4);
  }
  __syncthreads();
  TRUNK_STAMP(5);
  {  // deferred stores of the pointfeat tile (point-major [cloud points][64], coalesced 16 KiB) and its max
    float* dstbase = pointfeat + (ti.is_obs ? ((size_t)ti.obj * N + ti.p0) * 64
                                            : ((size_t)B * N + (size_t)ti.obj * M + ti.p0) * 64);
    if (part == 0 && pf_row < ti.valid) {
      f32x4* d = reinterpret_cast<f32x4*>(dstbase + pf_row * 64);
      d[pf_c4] = pf_out0;
      d[pf_c4 + 8] = pf_out1;
    }
    if (part == 0 && tid < 64) pm[(size_t)tile * PMW + 1024 + tid] = pf_max;
  }
  if (SAVE) save_tile_rows<512, 512, true>(a3, 512, sv.s4 + trow0 * 512, tid);
  f32x16 acc4[MB4][NB4];
#pragma unroll
  for (int mb = 0; mb < MB4; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB4; ++nb) acc4[mb][nb] = zero16();
  g4.run(acc4, a3 + (RS == 8 ? (wave & 1) * 32 * 512 : 0), 512, lane);
  TRUNK_STAMP(6);
  if constexpr (RS == 8) {
    float m = acc4[0][0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc4[0][0][r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float* xch = a2 + (wave >> 1) * 32;  // a2 is dead since the barrier after conv3
    if ((wave & 1) && lane < 32) xch[lane] = m;
    __syncthreads();
    if (!(wave & 1) && lane < 32) pm[(size_t)tile * PMW + mb0 * 32 + lane] = fmaxf(m, xch[lane]) + bl4[0];
  } else if constexpr (SAVE) {
    argmax_tile_store<MB4, 2>(acc4, sv.pmax + (size_t)tile * 1024, sv.pidx + (size_t)tile * 1024, mb0 * 32, bl4,
                              (int)trow0, lane);
  } else {
    max_tile_store_pre<MB4, 2>(acc4, pm + (size_t)tile * PMW, mb0 * 32, bl4, false, lane);
  }
  TRUNK_STAMP(7);
#undef TRUNK_STAMP
}

// ------------------------------------------------------------------------------------------
// The trunk for grids that fill the chip (one workgroup per tile): 256 threads = ONE wave per SIMD, so that a wave may
// hold 512 registers and conv4 becomes an MB8 x NB2 wave tile (256 channels x 64 points = 256 accumulators): every LDS
// fragment feeds 8 MFMAs instead of 4.  The sweep is power-limited, not issue-limited (profiles/ubench/power.hip: the
// 8-wave MB4 x NB2 sweep saturates the matrix pipe and the chip answers with 2.0 GHz, 132-134 TFLOP/s; this form holds
// 2.36 GHz at 147).  Same operands in the same K order as k_trunk for every output element: same bits.
// ------------------------------------------------------------------------------------------
#ifndef TRUNK4_PFD
#define TRUNK4_PFD 2
#endif
#ifndef TRUNK4_PFB
#define TRUNK4_PFB 1
#endif
template <bool SAVE = false>
__global__ __launch_bounds__(256) void k_trunk4(catre_points P, const float* __restrict__ trans3,
                                                const float* __restrict__ trans64, const float* __restrict__ Wc1,
                                                const float* __restrict__ bc1, const f32x4* __restrict__ wp2,
                                                const float* __restrict__ b2, const f32x4* __restrict__ wp3,
                                                const float* __restrict__ b3, const f32x4* __restrict__ wp4,
                                                const float* __restrict__ b4, float* __restrict__ pm,
                                                float* __restrict__ pointfeat, int B, int N, int M,
                                                unsigned long long* __restrict__ trace, TrainSave sv = TrainSave{}) {
  __shared__ __attribute__((aligned(16))) float smem[TRUNK_SMEM];
#define TRUNK_STAMP(i)                                                                     \
  do {                                                                                     \
    if (CATRE_TRACE_ON && trace && lane == 0) trace[((size_t)tile * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
  float* h1 = smem;                      // [64][68]
  float* t64 = smem + TP * LD64;         // [64][64]
  float* pf = smem + TP * LD64 + 4096;   // [64][68]
  float* a3 = smem;                      // [64][512] swizzled
  float* a2 = smem + TP * 512;           // [64][128] swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;
  const TileInfo ti = tile_info(tile, B, N, M);
  const bool ft = trans64 != nullptr;
  TRUNK_STAMP(0);

  // conv2 64->128: wave -> m-block `wave`, both point blocks
  GemmPipe<1, 2, false, false, 8, 3> g2;
  g2.prefetch(wp2 + (wave * 8) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, wave * 32, lane);
  const size_t trow0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  {
    float x, y, z;
    load_point(P, ti, lane, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    if (SAVE && wave == 0) {
      float* xr = sv.s1 + (trow0 + lane) * 8;
      *reinterpret_cast<f32x4*>(xr) = f32x4{x, y, z, 0.f};
      *reinterpret_cast<f32x4*>(xr + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    conv3_relu_row<16>(x, y, z, Wc1, bc1, wave * 16, h1 + lane * LD64);
    if (ft) {
      const f32x4* src = reinterpret_cast<const f32x4*>(trans64 + (size_t)ti.cloud * 4096);
      f32x4* dst = reinterpret_cast<f32x4*>(t64);
#pragma unroll
      for (int u = 0; u < 4; ++u) dst[tid + 256 * u] = src[tid + 256 * u];
    }
  }
  __syncthreads();
  TRUNK_STAMP(1);
  if (SAVE) save_tile_rows<64, 256, false>(h1, LD64, sv.s2 + trow0 * 64, tid);
  if (ft) {
    {  // pointfeat[j][n] = sum_i T64[i][j] h1[i][n]  (pointnet.py:107-109): 2 m-blocks x 2 point blocks, one per wave
      const int mblk = wave >> 1, nb = wave & 1;
      const int n = lane & 31, h = lane >> 5;
      f32x16 acc = zero16();
      const float* xr = h1 + (nb * 32 + n) * LD64 + 4 * h;
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
      store_tile_lds<1, 1, false>(accs, pf + nb * 32 * LD64, LD64, mblk * 32, nullptr, lane);
    }
    __syncthreads();
  } else {
    pf = h1;
  }
  TRUNK_STAMP(2);
  // conv3 128->512: 16 m-blocks, four per wave
  GemmPipe<4, 2, false, true, 16, 2, 1> g3;
  g3.prefetch(wp3 + (wave * 4 * 16) * 64 + lane, 16 * 64);
  f32x4 bv3[4][4];
  load_bias_quads<4>(bv3, b3, wave * 128, lane);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 pf_out[4];
  float pf_max = 0.f;
  {
    // pointfeat tile -> registers now, -> HBM after the last barrier (see k_trunk); 16-byte chunk tid + 256 u of the tile
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u;
      pf_out[u] = *reinterpret_cast<const f32x4*>(pf + (i >> 4) * LD64 + (i & 15) * 4);
    }
    {  // max_n pointfeat: wave w reduces points [16w, 16w+16) for channel `lane`, 64 threads merge the 4 partials
      float* scratch = smem + 2 * TP * LD64 + 4096;  // [4][64]
      const float* col = pf + (wave * 16) * LD64 + lane;
      float m = col[0];
#pragma unroll
      for (int p = 1; p < 16; ++p) m = fmaxf(m, col[p * LD64]);
      scratch[wave * 64 + lane] = m;
      __syncthreads();
      if (tid < 64) {
        m = scratch[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) m = fmaxf(m, scratch[w * 64 + tid]);
        pf_max = m;
      }
    }
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g2.run(acc, pf, LD64, lane);
    store_tile_lds_pre<1, 2, true, true>(acc, a2, 128, wave * 32, bv2, lane);
  }
  __syncthreads();
  TRUNK_STAMP(3);
  if (SAVE) save_tile_rows<128, 256, true>(a2, 128, sv.s3 + trow0 * 128, tid);
  // conv4 512->1024: wave owns 8 m-blocks (channels [wave*256, +256)); first weight chunks + bias requested now
  const int mb0 = wave * 8;
  GemmPipe<8, 2, true, true, 64, TRUNK4_PFD, TRUNK4_PFB> g4;
  float bl4[8];
  {
    f32x16 acc3[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc3[mb][0] = acc3[mb][1] = zero16();
    g3.run(acc3, a2, 128, lane);
    // conv4's first weight chunks + bias are requested behind conv3's sweep: the L2 round trip hides behind the epilogue
    // and the barrier, and the 64 registers they land in are not live during the sweep
    g4.prefetch(wp4 + ((size_t)mb0 * 64) * 64 + lane, 64 * 64);
    load_bias_lane<8>(bl4, b4, mb0 * 32, lane);
    __builtin_amdgcn_sched_barrier(0);
    store_tile_lds_pre<4, 2, true, true>(acc3, a3, 512, wave * 128, bv3, lane);
    TRUNK_STAMP(4);
  }
  __syncthreads();
  TRUNK_STAMP(5);
  {  // deferred stores of the pointfeat tile (point-major [cloud points][64]: the tile is 16 KiB contiguous) and its max
    float* dstbase = pointfeat + trow0 * 64;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u;
      if ((i >> 4) < ti.valid) *reinterpret_cast<f32x4*>(dstbase + (size_t)i * 4) = pf_out[u];
    }
    if (tid < 64) pm[(size_t)tile * PMW + 1024 + tid] = pf_max;
  }
  if (SAVE) save_tile_rows<512, 256, true>(a3, 512, sv.s4 + trow0 * 512, tid);
  f32x16 acc4[8][2];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) acc4[mb][0] = acc4[mb][1] = zero16();
  g4.run(acc4, a3, 512, lane);
  TRUNK_STAMP(6);
  if constexpr (SAVE) {
    argmax_tile_store<8, 2>(acc4, sv.pmax + (size_t)tile * 1024, sv.pidx + (size_t)tile * 1024, mb0 * 32, bl4, (int)trow0,
                            lane);
  } else {
    max_tile_store_pre<8, 2>(acc4, pm + (size_t)tile * PMW, mb0 * 32, bl4, false, lane);
  }
  TRUNK_STAMP(7);
#undef TRUNK_STAMP
}

// ------------------------------------------------------------------------------------------
// tile partial maxima -> per-cloud max:  out[cloud][c] = max_t pm[row(cloud,t)][c]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_pm(const float* __restrict__ pm, float* __restrict__ out, int ldo, int C,
                                                   int B, int N, int M, int rpt = 1 /*partial rows per tile*/) {
  // grid (clouds, C / 256): one thread per (cloud, channel), so that a single object still spreads over several CUs
  const int cloud = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const int TN = (N + TP - 1) / TP, TM = (M + TP - 1) / TP;
  const int nt = (cloud < B ? TN : TM) * rpt;
  const size_t row0 = (cloud < B ? (size_t)cloud * TN : (size_t)B * TN + (size_t)(cloud - B) * TM) * rpt;
  const float* src = pm + row0 * PMW + c;
  float m = src[0];
  int t = 1;
  for (; t + 3 < nt; t += 4) {  // four loads in flight
    const float a = src[(size_t)t * PMW], b = src[(size_t)(t + 1) * PMW], d = src[(size_t)(t + 2) * PMW],
                e = src[(size_t)(t + 3) * PMW];
    m = fmaxf(fmaxf(m, fmaxf(a, b)), fmaxf(d, e));
  }
  for (; t < nt; ++t) m = fmaxf(m, src[(size_t)t * PMW]);
  out[(size_t)cloud * ldo + c] = m;
}

// ------------------------------------------------------------------------------------------
// generic y = act(x W^T + b) (+ I_k): one wave per 32x32 output block, operands straight from
// L2 (F.linear in pointnet.py:31-33,64-66 and the global-feature half of RotHead layer 0).
// ------------------------------------------------------------------------------------------
#define LIN_WAVES 8
// The body of k_linear for output block (bx, by): X is a global matrix or an LDS staging buffer (k_linear_pm and
// k_heads_a, catre_small.h) - the same fragment addressing, the same MFMA sequence, the same bits either way.
// `stage()` runs after the first trip's weight fragments have been requested and before X is read: the LDS-staged callers
// fill X there (+ barrier), so that their weight round trip overlaps the staging instead of following it.
// TW: W is given TRANSPOSED - element (output j, input k) at W[k * ldw + j] - and read with four 4-byte loads per chunk
// (consecutive lanes, consecutive j) instead of one 16-byte load: the dgrad of a small linear (dx = dy W) straight from
// the layer's own [J][K] weight, no transposed copy.
// XM (TW instances only): a matrix laid out like X; X is zeroed where XM <= 0 as it is loaded (ReLU backward folded in).
template <bool TW = false, class StageF>
__device__ __forceinline__ void linear_body(const float* X, int ldx, const float* __restrict__ W, int ldw,
                                            const float* __restrict__ bias, float* __restrict__ Y, int ldy, int R, int J,
                                            int K, int relu, int iden_k, int bx, int by, float (*part)[16][64],
                                            StageF stage, const float* __restrict__ XM = nullptr) {
  // 8 waves split K (interleaved 8-wide chunks), each with up to 8 chunk pairs in flight - the kernel is a chain of
  // L2 round trips, so the trip count (K/8/8/8 = 2 for K = 1024) is what sets its time; partial 32x32 blocks are
  // summed through LDS in wave order (deterministic).
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = min(bx * 32 + i, R - 1), j = min(by * 32 + i, J - 1);
  const f32x4* xa = reinterpret_cast<const f32x4*>(X + (size_t)r * ldx + 4 * h);
  const f32x4* xm = reinterpret_cast<const f32x4*>(XM + (size_t)r * ldx + 4 * h);
  auto xload = [&](int c) -> f32x4 {
    f32x4 v = xa[c * 2];
    if constexpr (TW) {
      if (XM) {
        const f32x4 m = xm[c * 2];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = m[q] > 0.f ? v[q] : 0.f;
      }
    }
    return v;
  };
  const f32x4* wb = reinterpret_cast<const f32x4*>(W + (size_t)j * ldw + 4 * h);
  const float* wt = W + (size_t)(4 * h) * ldw + j;  // TW: chunk c of this lane = wt[(8 c + s) * ldw], s = 0..3
  auto wload = [&](int c) -> f32x4 {
    if constexpr (TW) {
      const float* q = wt + (size_t)(8 * c) * ldw;
      return f32x4{q[0], q[ldw], q[2 * (size_t)ldw], q[3 * (size_t)ldw]};
    } else {
      return wb[c * 2];
    }
  };
  f32x16 acc = zero16();
  const int nkc = K / 8;
  int kc = wave;
  f32x4 b[8];
  const bool first = kc + 7 * LIN_WAVES < nkc;
  if (first) {
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = wload(kc + LIN_WAVES * u);
  }
  __builtin_amdgcn_sched_barrier(0);
  stage();
  for (; kc + 7 * LIN_WAVES < nkc; kc += 8 * LIN_WAVES) {  // 8 chunks of this wave per trip
    f32x4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = xload(kc + LIN_WAVES * u);
      if (kc != wave) b[u] = wload(kc + LIN_WAVES * u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma32(a[u][s], b[u][s], acc);  // D[row r][col j]
  }
  for (; kc + 3 * LIN_WAVES < nkc; kc += 4 * LIN_WAVES) {  // K = 256 (fc3 of the STNs): four chunks, one round trip
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = xload(kc + LIN_WAVES * u);
      b[u] = wload(kc + LIN_WAVES * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma32(a[u][s], b[u][s], acc);
  }
  for (; kc < nkc; kc += LIN_WAVES) {
    const f32x4 a = xload(kc), b = wload(kc);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma32(a[s], b[s], acc);
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) part[wave][reg][lane] = acc[reg];
  __syncthreads();
  const int col = by * 32 + i;
  if (col >= J) return;
  const float bv = bias ? bias[col] : 0.f;
  const float idv = (iden_k > 0 && col < iden_k * iden_k && (col % (iden_k + 1)) == 0) ? 1.f : 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {  // wave w finishes registers 2w, 2w+1 -> rows (reg&3) + 8(reg>>2) + 4h
    const int reg = wave * 2 + q;
    const int row = bx * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (row < R) {
      float v = part[0][reg][lane];
#pragma unroll
      for (int w = 1; w < LIN_WAVES; ++w) v += part[w][reg][lane];
      v += bv;
      if (relu) v = fmaxf(v, 0.f);
      Y[(size_t)row * ldy + col] = v + idv;
    }
  }
}


__global__ __launch_bounds__(64 * LIN_WAVES) void k_linear(const float* __restrict__ X, int ldx,
                                                            const float* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias, float* __restrict__ Y,
                                                            int ldy, int R, int J, int K, int relu, int iden_k,
                                                            const float* __restrict__ W_z1 = nullptr,
                                                            const float* __restrict__ bias_z1 = nullptr,
                                                            float* __restrict__ Y_z1 = nullptr) {
  // gridDim.z == 2: a second (W, bias, Y) on the same X in the same launch (the two rotation heads' global halves)
  if (blockIdx.z == 1) {
    W = W_z1;
    bias = bias_z1;
    Y = Y_z1;
  }
  __shared__ float part[LIN_WAVES][16][64];
  linear_body(X, ldx, W, ldw, bias, Y, ldy, R, J, K, relu, iden_k, blockIdx.x, blockIdx.y, part, [] {});
}

// Y = X W for W [K][J] row-major (the transposed-weight form of k_linear: dgrad of the FC tails / ts head in training)
__global__ __launch_bounds__(64 * LIN_WAVES) void k_linear_t(const float* __restrict__ X, int ldx,
                                                              const float* __restrict__ W, int ldw,
                                                              float* __restrict__ Y, int ldy, int R, int J, int K,
                                                              const float* __restrict__ XM) {
  __shared__ float part[LIN_WAVES][16][64];
  linear_body<true>(X, ldx, W, ldw, nullptr, Y, ldy, R, J, K, 0, 0, blockIdx.x, blockIdx.y, part, [] {}, XM);
}

// ------------------------------------------------------------------------------------------
// a7+a8: ts head (heads/fc_trans_size_head.py:61-70) on
// ts_feat = [flat_pcl_feat | (flat_kps_feat) | (init_scale) | (init_trans)] (CATRE_disR_shared.py:69-82)
// 4 objects per workgroup, thread = output channel; weights pre-transposed to [in][256].
// ------------------------------------------------------------------------------------------
#define TS_OB 4

__device__ __forceinline__ float group8_norm_gelu(float v, float gamma, float beta) {
  // GroupNorm(32,256) on a [B,256] row: statistics over 8 consecutive channels = 8 consecutive lanes
  float s = v;
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 4);
  const float mean = s * 0.125f;
  const float d = v - mean;
  float q = d * d;
  q += __shfl_xor(q, 1);
  q += __shfl_xor(q, 2);
  q += __shfl_xor(q, 4);
  const float rstd = 1.0f / sqrtf(q * 0.125f + 1e-5f);
  const float sc = rstd * gamma;
  return gelu_erf(fmaf(v, sc, beta - mean * sc));
}

// Layer 0 (in_dim -> 256) split over TS_KS workgroups per group of TS_OB objects: each gathers its K slice of
// ts_feat and writes a partial sum per (object, slice, channel).  With one workgroup the layer is a serial chain of
// ~1100 L2 round trips (33 us for a single object); eight slices cut it to ~140 and put 8x as many CUs to work.
#define TS_KS 8
// max over a cloud's tile partials (what k_reduce_pm writes to gfeat): out = max_t pm[row(cloud, t)][c]
// (rpt partial rows per tile: 2 after the half-tile trunk of catre_small.h, else 1)
__device__ __forceinline__ float cloud_max1(const float* __restrict__ pm, int cloud, int c, int B, int N, int M,
                                            int rpt) {
  const int TN = (N + TP - 1) / TP, TM = (M + TP - 1) / TP;
  const int nt = (cloud < B ? TN : TM) * rpt;
  const size_t row0 = (cloud < B ? (size_t)cloud * TN : (size_t)B * TN + (size_t)(cloud - B) * TM) * rpt;
  const float* src = pm + row0 * PMW + c;
  float m = src[0];
  int t = 1;
  for (; t + 7 < nt; t += 8) {  // eight loads in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(t + u) * PMW];
#pragma unroll
    for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
  }
  for (; t < nt; ++t) m = fmaxf(m, src[(size_t)t * PMW]);
  return m;
}

// Body for object group bx, K slice ks.  pm != nullptr (small batches, k_heads_a): gfeat does not exist yet - the gather
// takes the maximum over the cloud's tile partials itself (max is exact: the same values k_reduce_pm would have written).
__device__ __forceinline__ void ts_l0_body(const float* __restrict__ gfeat, const float* __restrict__ pm,
                                           const float* __restrict__ pose, const float* __restrict__ scale,
                                           const float* __restrict__ W0T, float* __restrict__ part /*[B][TS_KS][256]*/,
                                           int B, int N, int M, int in_dim, int with_kps, int with_scale, int with_trans,
                                           int bx, int ks, float* feat /*LDS [TS_OB][slice length]*/, int rpt = 1) {
  const int tid = threadIdx.x;
  const int b0i = bx * TS_OB;
  const int k0 = (in_dim * ks) / TS_KS, k1 = (in_dim * (ks + 1)) / TS_KS, len = k1 - k0;
  for (int o = 0; o < TS_OB; ++o) {
    const int b = min(b0i + o, B - 1);
    for (int k = k0 + tid; k < k1; k += 256) {
      int kk = k;
      float v;
      if (kk < PMW) {
        v = pm ? cloud_max1(pm, b, kk, B, N, M, rpt) : gfeat[(size_t)b * PMW + kk];
      } else {
        kk -= PMW;
        if (with_kps && kk < PMW) {
          v = pm ? cloud_max1(pm, B + b, kk, B, N, M, rpt) : gfeat[(size_t)(B + b) * PMW + kk];
        } else {
          if (with_kps) kk -= PMW;
          if (with_scale && kk < 3) {
            v = scale[b * 3 + kk];
          } else {
            if (with_scale) kk -= 3;
            v = pose[b * 12 + kk * 4 + 3];  // init_trans (with_trans)
          }
        }
      }
      feat[o * len + k - k0] = v;
    }
  }
  __syncthreads();
  float acc[TS_OB];
#pragma unroll
  for (int o = 0; o < TS_OB; ++o) acc[o] = 0.f;
  // 16 weight rows requested together, then used: left to itself hipcc waits for every row before asking for the next
  // (one L2 round trip, ~150 ns, per row - measured 24 us for 137 rows)
  const float* wcol = W0T + (size_t)k0 * 256 + tid;
  int k = 0;
  for (; k + 16 <= len; k += 16) {
    float w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = wcol[(size_t)(k + u) * 256];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int o = 0; o < TS_OB; ++o) acc[o] = fmaf(feat[o * len + k + u], w[u], acc[o]);
  }
  for (; k < len; ++k) {
    const float w = wcol[(size_t)k * 256];
#pragma unroll
    for (int o = 0; o < TS_OB; ++o) acc[o] = fmaf(feat[o * len + k], w, acc[o]);
  }
#pragma unroll
  for (int o = 0; o < TS_OB; ++o)
    if (b0i + o < B) part[((size_t)(b0i + o) * TS_KS + ks) * 256 + tid] = acc[o];
}

__global__ __launch_bounds__(256) void k_ts_l0(const float* __restrict__ gfeat, const float* __restrict__ pose,
                                               const float* __restrict__ scale, const float* __restrict__ W0T,
                                               float* __restrict__ part /*[B][TS_KS][256]*/, int B, int in_dim, int with_kps,
                                               int with_scale, int with_trans) {
  extern __shared__ __attribute__((aligned(16))) float feat[];  // [TS_OB][slice length]
  ts_l0_body(gfeat, nullptr, pose, scale, W0T, part, B, 0, 0, in_dim, with_kps, with_scale, with_trans, blockIdx.x,
             blockIdx.y, feat);
}

struct TsHeadArgs {
  const float *l0part /*[B][TS_KS][256]*/, *b0, *g0, *be0, *W1T, *b1, *g1, *be1, *Wt, *bt, *Ws, *bs;
  float *dt, *ds;
  int B;
};

// body for object group bx (1024 threads); sm: TS_OB * (256 + 4 * 256) floats of LDS
__device__ __forceinline__ void ts_head_body(const TsHeadArgs& A, int bx, float* sm) {
  const float* __restrict__ l0part = A.l0part;
  const float *__restrict__ b0 = A.b0, *__restrict__ g0 = A.g0, *__restrict__ be0 = A.be0, *__restrict__ W1T = A.W1T,
                           *__restrict__ b1 = A.b1, *__restrict__ g1 = A.g1, *__restrict__ be1 = A.be1,
                           *__restrict__ Wt = A.Wt, *__restrict__ bt = A.bt, *__restrict__ Ws = A.Ws,
                           *__restrict__ bs = A.bs;
  float *__restrict__ dt = A.dt, *__restrict__ ds = A.ds;
  const int B = A.B;
  float* hbuf = sm;                          // [TS_OB][256]
  float* kpart = hbuf + TS_OB * 256;         // [4][TS_OB][256] K-slice partial sums
  // 1024 threads: 4 K-slices x 256 output channels; slice partials are merged in slice order
  const int tid = threadIdx.x & 255, ks = threadIdx.x >> 8;
  const int b0i = bx * TS_OB;
  float acc[TS_OB];
  if (ks == 0) {
    const float ga = g0[tid], be = be0[tid], bb = b0[tid];
#pragma unroll
    for (int o = 0; o < TS_OB; ++o) {
      const float* p = l0part + (size_t)min(b0i + o, B - 1) * TS_KS * 256 + tid;
      const float v = bb + (((p[0] + p[256]) + (p[512] + p[768])) + ((p[1024] + p[1280]) + (p[1536] + p[1792])));
      hbuf[o * 256 + tid] = group8_norm_gelu(v, ga, be);
    }
  }
  __syncthreads();
  {
#pragma unroll
    for (int o = 0; o < TS_OB; ++o) acc[o] = 0.f;
#pragma unroll 1
    for (int kb = ks * 64; kb < ks * 64 + 64; kb += 16) {  // 16 rows of W1T in flight (see k_ts_l0)
      float w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = W1T[(kb + u) * 256 + tid];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int o = 0; o < TS_OB; ++o) acc[o] = fmaf(hbuf[o * 256 + kb + u], w[u], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < TS_OB; ++o) kpart[(ks * TS_OB + o) * 256 + tid] = acc[o];
  }
  __syncthreads();
  if (ks == 0) {
    const float ga = g1[tid], be = be1[tid], bb = b1[tid];
#pragma unroll
    for (int o = 0; o < TS_OB; ++o) {
      const float v = bb + (((kpart[o * 256 + tid] + kpart[(TS_OB + o) * 256 + tid]) +
                             kpart[(2 * TS_OB + o) * 256 + tid]) + kpart[(3 * TS_OB + o) * 256 + tid]);
      hbuf[o * 256 + tid] = group8_norm_gelu(v, ga, be);
    }
  }
  __syncthreads();
  // fc_t / fc_s: 32 lanes per output (object, coordinate): lane j takes k = j, j + 32, ... and a fixed shuffle tree adds
  // the 32 partials (a serial 256-term chain per output took most of this kernel's time)
  const int grp = threadIdx.x >> 5, j = threadIdx.x & 31;
  if (grp < TS_OB * 6) {
    const int o = grp / 6, c = grp % 6;
    const int b = b0i + o;
    const float* w = c < 3 ? Wt + c * 256 : Ws + (c - 3) * 256;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s = fmaf(hbuf[o * 256 + j + 32 * u], w[j + 32 * u], s);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (j == 0 && b < B) {
      s += c < 3 ? bt[c] : bs[c - 3];
      if (c < 3)
        dt[b * 3 + c] = s;
      else
        ds[b * 3 + c - 3] = s;
    }
  }
}

__global__ __launch_bounds__(1024) void k_ts_head(TsHeadArgs A) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  ts_head_body(A, blockIdx.x, sm);
}

// ------------------------------------------------------------------------------------------
// a9: rotation heads (heads/conv_out_per_rot_head.py:126-140) over the concatenated point
// sequence [observed N | prior M] (CATRE_disR_shared.py:86), never materialised.
//   layer 0 : y0 = W0[:,1024:] . pointfeat + (W0[:,:1024] . g_cloud + b0)      (bias0 from k_linear)
//   GN(32,256) couples all N+M points of an object.  GN0: y0 is linear in pointfeat, so its statistics come from the
//   per-cloud mean and scatter matrix of pointfeat (catre_gram.h) instead of a first pass over layer 0.  GN1: per-tile
//   (mean, M2) partials merged with Chan's formula in tile order (deterministic).
// Tiles never straddle the observed/prior boundary: T = ceil(N/64) + ceil(M/64) per object.
// ------------------------------------------------------------------------------------------
struct RotTile {
  int obj, t, is_obs, cloud, p0, valid, gp0;  // gp0 = index in the concatenated sequence
  size_t pf_off;                              // float offset of the tile inside the pointfeat buffer
};

__device__ __forceinline__ RotTile rot_tile(int bid, int B, int N, int M) {
  const int TN = (N + TP - 1) / TP, TM = (M + TP - 1) / TP, T = TN + TM;
  RotTile r;
  r.obj = bid / T;
  r.t = bid % T;
  r.is_obs = r.t < TN;
  r.cloud = r.is_obs ? r.obj : B + r.obj;
  r.p0 = (r.is_obs ? r.t : r.t - TN) * TP;
  r.valid = min(TP, (r.is_obs ? N : M) - r.p0);
  r.gp0 = r.is_obs ? r.p0 : N + r.p0;
  r.pf_off = r.is_obs ? ((size_t)r.obj * N + r.p0) * 64 : ((size_t)B * N + (size_t)r.obj * M + r.p0) * 64;
  return r;
}

// Merge per-tile (mean, M2) partials of one (object, head, group) in tile order -> (mean, rstd)
__device__ __forceinline__ void merge_gn(const float* __restrict__ part /*[T][64]*/, int group, int T, int TN, int N,
                                         int M, float& mean_out, float& rstd_out) {
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int t = 0; t < T; ++t) {
    const int p0 = (t < TN ? t : t - TN) * TP;
    const float nb = 8.f * (float)min(TP, (t < TN ? N : M) - p0);
    const float mb = part[t * 64 + group * 2], m2b = part[t * 64 + group * 2 + 1];
    const float nn = n + nb, delta = mb - mean;
    mean += delta * (nb / nn);
    m2 += m2b + delta * delta * (n * nb / nn);
    n = nn;
  }
  mean_out = mean;
  rstd_out = 1.0f / sqrtf(m2 / n + 1e-5f);
}

#include "catre_rot.h"

// rot[b][hd*rd+c] = sum_tiles rpart + neck_b[c] * sum_p w_p + conv_p.bias   (rd = RotHead.rot_dim: 3 for rot6d, 2 for quat)
__global__ void k_rot_finish(const float* __restrict__ rpart, const float* __restrict__ neckbx,
                             const float* __restrict__ neckby, const float* __restrict__ sumwp /*[2]*/,
                             const float* __restrict__ cpbx, const float* __restrict__ cpby, float* __restrict__ rot6d,
                             int B, int T, int rd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 2 * rd) return;
  const int b = i / (2 * rd), hd = (i % (2 * rd)) / rd, c = i % rd;
  const float* rp = rpart + ((size_t)b * 2 + hd) * T * 4 + c;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += rp[t * 4];
  const float nb = (hd ? neckby : neckbx)[c];
  const float* cpb = hd ? cpby : cpbx;
  s = fmaf(nb, sumwp[hd], s);
  if (cpb) s += cpb[0];
  rot6d[i] = s;
}

// ------------------------------------------------------------------------------------------
// a10-a12: rot6d -> R (rot_reps.py:46-55), pose/scale update (pose_scale_from_delta_init.py:48-93)
// ------------------------------------------------------------------------------------------
#include "catre_so3.h"

__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ __forceinline__ void normalize3(float* v) {  // F.normalize(p=2, eps=1e-12)
  const float nrm = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
  v[0] /= nrm;
  v[1] /= nrm;
  v[2] /= nrm;
}

// one object b; rotp: this object's rotation parameters (catre_rot_dim values, or 9 when rot_input_is_matrix) - global
// memory or LDS (k_finish_update, catre_small.h)
__device__ __forceinline__ void pose_update_obj(const float* rotp, const float* __restrict__ dtr,
                                                const float* __restrict__ dsr, const float* __restrict__ pose0,
                                                const float* __restrict__ scale0, const float* __restrict__ mean_scales,
                                                const float* __restrict__ Ks, const catre_opts& o,
                                                float* __restrict__ pose_out, float* __restrict__ scale_out, int b,
                                                float* __restrict__ pose_echo = nullptr,
                                                float* __restrict__ scale_echo = nullptr) {
  if (pose_echo) {  // catre_refine_k_from: slot 0 of the K-loop's output = the caller's initial estimate, no copy launch
#pragma unroll
    for (int i = 0; i < 12; ++i) pose_echo[b * 12 + i] = pose0[b * 12 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) scale_echo[b * 3 + i] = scale0[b * 3 + i];
  }
  float dR[9];
  if (o.rot_input_is_matrix) {
#pragma unroll
    for (int i = 0; i < 9; ++i) dR[i] = rotp[i];
  } else {  // get_rot_mat, models/model_utils.py:28-40
    const int rd = catre_rot_dim(o.rot_type);
    float r[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) r[i] = i < rd ? rotp[i] : 0.f;
    rot_param_to_mat(r, o.rot_type, dR);
  }

  const float* p0 = pose0 + b * 12;
  const float t0[3] = {p0[3], p0[7], p0[11]};
  float d[3] = {dtr[b * 3] * o.delta_t_weight, dtr[b * 3 + 1] * o.delta_t_weight, dtr[b * 3 + 2] * o.delta_t_weight};
  float t[3];
  if (!o.delta_t_space_3d) {
    const float zsrc = t0[2];
    const float ztgt = o.delta_z_deepim ? zsrc / expf(d[2]) : d[2] * zsrc;
    const float fx = o.k_aware ? Ks[b * 9 + 0] : 1.f, fy = o.k_aware ? Ks[b * 9 + 4] : 1.f;
    t[0] = ztgt * (d[0] / fx + t0[0] / zsrc);
    t[1] = ztgt * (d[1] / fy + t0[1] / zsrc);
    t[2] = ztgt;
  } else {
    t[0] = t0[0] + d[0];
    t[1] = t0[1] + d[1];
    t[2] = t0[2] + d[2];
  }
  const float* sb = o.scale_base_mean ? mean_scales + b * 3 : scale0 + b * 3;
  float s[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) s[i] = o.scale_mul ? sb[i] * expf(dsr[b * 3 + i]) : sb[i] + dsr[b * 3 + i];

  if (o.is_allo) {  // allo_to_ego_mat_torch, core/utils/utils.py:200-231
    const float nrm = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) + o.allo_eps;
    const float ray[3] = {t[0] / nrm, t[1] / nrm, t[2] / nrm};
    const float angle = acosf(ray[2]);
    float ax[3] = {-ray[1], ray[0], 0.f};  // (0,0,1) x ray
    const float an = sqrtf(ax[0] * ax[0] + ax[1] * ax[1]) + o.allo_eps;
    ax[0] /= an;
    ax[1] /= an;
    const float sh = sinf(angle * 0.5f);
    float q[4] = {cosf(angle * 0.5f), ax[0] * sh, ax[1] * sh, 0.f * sh};
    const float qn = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] /= qn;  // quat2mat_torch, core/utils/pose_utils.py:349-412
    const float X = q[1] * 2.f, Y = q[2] * 2.f, Z = q[3] * 2.f;
    const float wX = q[0] * X, wY = q[0] * Y, wZ = q[0] * Z, xX = q[1] * X, xY = q[1] * Y, xZ = q[1] * Z;
    const float yY = q[2] * Y, yZ = q[2] * Z, zZ = q[3] * Z;
    const float A[9] = {1.f - (yY + zZ), xY - wZ, xZ + wY, xY + wZ, 1.f - (xX + zZ), yZ - wX,
                        xZ - wY,         yZ + wX, 1.f - (xX + yY)};
    float E[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) E[i * 3 + j] = A[i * 3] * dR[j] + A[i * 3 + 1] * dR[3 + j] + A[i * 3 + 2] * dR[6 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) dR[i] = E[i];
  }
  float* po = pose_out + b * 12;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      po[i * 4 + j] = dR[i * 3] * p0[j] + dR[i * 3 + 1] * p0[4 + j] + dR[i * 3 + 2] * p0[8 + j];  // dR @ R0
    po[i * 4 + 3] = t[i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) scale_out[b * 3 + i] = o.refine_scale ? s[i] : scale0[b * 3 + i];
}

__global__ void k_pose_update(const float* __restrict__ rot6d, const float* __restrict__ dtr,
                              const float* __restrict__ dsr, const float* __restrict__ pose0,
                              const float* __restrict__ scale0, const float* __restrict__ mean_scales,
                              const float* __restrict__ Ks, catre_opts o, float* __restrict__ pose_out,
                              float* __restrict__ scale_out, int B, float* __restrict__ pose_echo = nullptr,
                              float* __restrict__ scale_echo = nullptr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  pose_update_obj(rot6d + (size_t)b * (o.rot_input_is_matrix ? 9 : catre_rot_dim(o.rot_type)), dtr, dsr, pose0, scale0,
                  mean_scales, Ks, o, pose_out, scale_out, b, pose_echo, scale_echo);
}

// ------------------------------------------------------------------------------------------
// stand-alone channel-wise max-pool [B,C,N] -> [B,C]: one wave per row, float4 streaming loads
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_colmax(const float* __restrict__ x, float* __restrict__ out, int rows,
                                                int N) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int nwaves = gridDim.x * wpb;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += nwaves) {
    const float* r = x + (size_t)row * N;
    float m = -INFINITY;
    if ((N & 3) == 0) {
      const f32x4* r4 = reinterpret_cast<const f32x4*>(r);
      const int n4 = N >> 2;
#pragma unroll 4
      for (int i = lane; i < n4; i += 64) {
        const f32x4 v = __builtin_nontemporal_load(r4 + i);
        m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
      }
    } else {
      for (int i = lane; i < N; i += 64) m = fmaxf(m, r[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) out[row] = m;
  }
}

#include "catre_bf16.h"
#include "catre_split.h"

// every bf16 (or hi + lo split) fragment pack of a weight image in ONE launch (k_pack_frag_multi's job table; the
// per-element arithmetic is k_pack_frag_bf's / k_pack_frag_split's): a training step re-packs the image after every
// optimizer step, and twelve 3-us launches sat on the critical path of each autocast / split iteration
template <bool SPLIT>
__global__ void k_pack_frag_lp_multi(PackJobs J) {
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= J.end[J.n - 1]) return;
  int j = 0;
  while (gi >= J.end[j]) ++j;
  const int base = j ? J.end[j - 1] : 0, idx = gi - base, total = J.end[j] - base;  // total = rows * K
  const int K = J.K[j];
  const int e = idx & 7, lane = (idx >> 3) & 63, rest = idx >> 9;
  const int nkc = K / 16;
  const int kc = rest % nkc, mb = rest / nkc;
  const int row = mb * 32 + (lane & 31), col = kc * 16 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
  const float w = J.src[j][(size_t)row * J.ld[j] + J.coloff[j] + col];
  unsigned short* dst = reinterpret_cast<unsigned short*>(J.dst[j]);
  const __bf16 h = (__bf16)w;
  dst[idx] = __builtin_bit_cast(unsigned short, h);
  if constexpr (SPLIT) dst[(size_t)total + idx] = __builtin_bit_cast(unsigned short, (__bf16)(w - (float)h));
}
#include "catre_gram.h"
#include "catre_small.h"
#include "catre_train.h"
#include "catre_aug.h"
#include "catre_pcl.h"
#include "catre_loss.h"

// ==========================================================================================
// host side: packed-weight and workspace layouts, launchers, C ABI
// ==========================================================================================
namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct PackLayout {
  size_t stn_c2, stn_c3, fstn_c1, fstn_c2, fstn_c3, c2, c3, c4, rot_l0[2], rot_l1[2], ts_w0t, ts_w1t, sumwp, total;
  // bf16 fragment packs of the same matrices (catre_bf16.h), offsets in floats
  size_t bf_stn_c2, bf_stn_c3, bf_fstn_c1, bf_fstn_c2, bf_fstn_c3, bf_c2, bf_c3, bf_c4, bf_rot_l0[2], bf_rot_l1[2];
  // hi + lo bf16 fragment packs of the three split-mode layers (catre_split.h), offsets in floats
  size_t sp_stn_c2, sp_stn_c3, sp_fstn_c1, sp_fstn_c2, sp_fstn_c3, sp_c3, sp_c4, sp_rot_l0[2], sp_rot_l1[2];
};

PackLayout pack_layout(int ts_in) {
  PackLayout L;
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t r = o;
    o += align_up(n, 64);
    return r;
  };
  L.stn_c2 = take(128 * 64);
  L.stn_c3 = take(1024 * 128);
  L.fstn_c1 = take(64 * 64);
  L.fstn_c2 = take(128 * 64);
  L.fstn_c3 = take(1024 * 128);
  L.c2 = take(128 * 64);
  L.c3 = take(512 * 128);
  L.c4 = take(1024 * 512);
  for (int h = 0; h < 2; ++h) {
    L.rot_l0[h] = take(256 * 64);
    L.rot_l1[h] = take(256 * 256);
  }
  L.sumwp = take(64);
  L.bf_stn_c2 = take(128 * 64 / 2);
  L.bf_stn_c3 = take(1024 * 128 / 2);
  L.bf_fstn_c1 = take(64 * 64 / 2);
  L.bf_fstn_c2 = take(128 * 64 / 2);
  L.bf_fstn_c3 = take(1024 * 128 / 2);
  L.bf_c2 = take(128 * 64 / 2);
  L.bf_c3 = take(512 * 128 / 2);
  L.bf_c4 = take(1024 * 512 / 2);
  for (int h = 0; h < 2; ++h) {
    L.bf_rot_l0[h] = take(256 * 64 / 2);
    L.bf_rot_l1[h] = take(256 * 256 / 2);
  }
  L.sp_stn_c2 = take(128 * 64);
  L.sp_stn_c3 = take(1024 * 128);
  L.sp_fstn_c1 = take(64 * 64);
  L.sp_fstn_c2 = take(128 * 64);
  L.sp_fstn_c3 = take(1024 * 128);
  L.sp_c3 = take(512 * 128);
  L.sp_c4 = take(1024 * 512);
  for (int h = 0; h < 2; ++h) {
    L.sp_rot_l0[h] = take(256 * 64);
    L.sp_rot_l1[h] = take(256 * 256);
  }
  // everything above is independent of ts_in (the stage entry points rely on that)
  L.ts_w0t = take((size_t)ts_in * 256);
  L.ts_w1t = take(256 * 256);
  L.total = o;
  return L;
}

struct WsLayout {
  size_t tspart, xbuf, kbuf, pm, pool, h1, h2, trans3, trans64, gfeat, pointfeat, dt, ds, rot6d, bias0, gn0, gn1, aff0, gn1stat, y1, rpart,
      bar, total;  // offsets in floats
};

WsLayout ws_layout(int B, int N, int M) {
  const size_t TN = (N + TP - 1) / TP, TM = (M + TP - 1) / TP, T = TN + TM;
  const size_t b = B, P = (size_t)N + M;
  WsLayout L;
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t r = o;
    o += align_up(n, 64);
    return r;
  };
  L.tspart = take(b * 8 * 256);  // first: catre_ts_head finds it without knowing N, M (TS_KS = 8 layer-0 partials)
  L.xbuf = take(b * N * 3);
  L.kbuf = take(b * M * 3);
  L.pm = take((2 * b <= SMALL_ROWS ? 2 : 1) * b * T * PMW);  // small batches: a partial row per HALF tile (k_trunk_h, catre_small.h)
  L.pool = take(2 * b * 1024);
  L.h1 = take(2 * b * 512);
  L.h2 = take(2 * b * 256);
  L.trans3 = take(2 * b * 9);
  L.trans64 = take(2 * b * 4096);
  L.gfeat = take(2 * b * PMW);
  L.pointfeat = take(b * P * 64);
  L.dt = take(b * 3);
  L.ds = take(b * 3);
  L.rot6d = take(b * 6);
  L.bias0 = take(2 * 2 * b * 256);
  L.gn0 = take(b * 2 * T * 64);
  L.gn1 = take(b * 2 * T * 64);
  L.aff0 = take(b * 2 * 2 * 2 * 256);
  L.gn1stat = take(b * 2 * 64);
  {  // y1 [b][2][P][256]; before it is written the region holds the pointfeat moments (catre_gram.h)
    const size_t y1n = b * 2 * P * 256, mom = b * 2 * (4 * (4096 + 64) + 64);  // PF_NG partial moments per cloud
    const size_t y1t = b * 2 * T * 256 * 32;  // bf16 mode: [b][2][T tiles][256 channels][64 points] bf16 (whole tiles)
    L.y1 = take(std::max(std::max(y1n, mom), y1t));
  }
  L.rpart = take(b * 2 * T * 4);
  L.bar = take(64);  // k_fc_tail's barrier counters
  L.total = o;
  return L;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? CATRE_OK : CATRE_ERR_LAUNCH; }

inline bool dims_ok(int B, int N, int M) { return B > 0 && N > 0 && M > 0; }
// the PointNet stages also run on a single cloud per object (M == 0: PointNetfeat.forward on its own)
inline bool dims_ok1(int B, int N, int M) { return B > 0 && N > 0 && M >= 0; }
inline int n_clouds(int B, int M) { return M > 0 ? 2 * B : B; }
// workgroups per 64-point tile of the encoder kernels: small grids split a tile's output channels over 2 or 4
// workgroups as long as that still fits one wave of workgroups on the 256 CUs
// RS_DISPATCH(rs, L): expands the launch macro L(RS) for the compile-time RS matching rs
#define RS_DISPATCH(rs, L) \
  switch (rs) {            \
    case 4: L(4); break;   \
    case 2: L(2); break;   \
    default: L(1);         \
  }
inline int row_split(int tiles) { return tiles * 4 <= 256 ? 4 : tiles * 2 <= 256 ? 2 : 1; }
// the fp32 encoder kernels also come with eight workgroups per tile (a single object: 32 tiles on 256 CUs)
#define RS_DISPATCH8(rs, L) \
  switch (rs) {             \
    case 8: L(8); break;    \
    case 4: L(4); break;    \
    case 2: L(2); break;    \
    default: L(1);          \
  }
inline int row_split8(int tiles) { return tiles * 8 <= 256 ? 8 : row_split(tiles); }
// workgroups per object of k_gn0_from_moments (catre_gram.h): its 8 (head, channel block) combinations on 8 / 4 / 2 / 1
inline int gn0_shares(int B) { return B * 8 <= 512 ? 8 : B * 4 <= 512 ? 4 : B * 2 <= 512 ? 2 : 1; }
// workgroups per HALF tile of k_trunk_h (0: too many tiles for it)
inline int trunk_h_split(int tiles) { return tiles * 8 <= 256 ? 4 : tiles * 4 <= 256 ? 2 : tiles * 2 <= 256 ? 1 : 0; }

#define REQUIRE(cond) \
  do {                \
    if (!(cond)) return CATRE_ERR_BAD_ARG; \
  } while (0)

inline const f32x4* pk4(const float* packed, size_t off) { return reinterpret_cast<const f32x4*>(packed + off); }
inline const u32x4* pkb(const float* packed, size_t off) { return reinterpret_cast<const u32x4*>(packed + off); }

inline TsHeadArgs ts_head_args(const float* l0part, const float* const* prm, const float* packed, const PackLayout& L,
                               float* dt, float* ds, int B) {
  return TsHeadArgs{l0part, prm[CATRE_P_TS_L0_B], prm[CATRE_P_TS_GN0_W], prm[CATRE_P_TS_GN0_B], packed + L.ts_w1t,
                    prm[CATRE_P_TS_L1_B], prm[CATRE_P_TS_GN1_W], prm[CATRE_P_TS_GN1_B], prm[CATRE_P_TS_FCT_W],
                    prm[CATRE_P_TS_FCT_B], prm[CATRE_P_TS_FCS_W], prm[CATRE_P_TS_FCS_B], dt, ds, B};
}

int stn_fc_tail(const float* pooled, const float* const* prm, int base /*CATRE_P_*_FC1_W*/, float* h1, float* h2,
                float* out, int k, int R, hipStream_t st, const float* pm = nullptr, int B = 0, int N = 0, int M = 0,
                unsigned* bar = nullptr) {
  // relu(fc1) -> relu(fc2) -> fc3 + I_k   (pointnet.py:31-40 / 64-77)
  if (bar) {  // small batches: the three layers in ONE launch (k_fc_tail, catre_small.h)
    const int nb3 = (k * k + 31) / 32, grid = nb3 > 16 ? nb3 : 16;
    hipLaunchKernelGGL(k_fc_tail, dim3(grid), dim3(64 * LIN_WAVES), 0, st, pooled, pm, B, N, M, prm[base], prm[base + 1],
                       prm[base + 2], prm[base + 3], prm[base + 4], prm[base + 5], h1, h2, out, k, R, bar);
    return check_launch();
  }
  if (pm)  // small batches (catre_small.h): fc1 pools the tile partials itself, no k_reduce_pm launch in front
    hipLaunchKernelGGL(k_linear_pm, dim3(1, 512 / 32), dim3(64 * LIN_WAVES), 0, st, pm, B, N, M, prm[base], 1024,
                       prm[base + 1], h1, 512, R, 512, 1, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL(k_linear, dim3((R + 31) / 32, 512 / 32), dim3(64 * LIN_WAVES), 0, st, pooled, 1024, prm[base], 1024,
                       prm[base + 1], h1, 512, R, 512, 1024, 1, 0);
  hipLaunchKernelGGL(k_linear, dim3((R + 31) / 32, 256 / 32), dim3(64 * LIN_WAVES), 0, st, h1, 512, prm[base + 2], 512,
                     prm[base + 3], h2, 256, R, 256, 512, 1, 0);
  hipLaunchKernelGGL(k_linear, dim3((R + 31) / 32, (k * k + 31) / 32), dim3(64 * LIN_WAVES), 0, st, h2, 256, prm[base + 4], 256,
                     prm[base + 5], out, k * k, R, k * k, 256, 0, k);
  return check_launch();
}

// ---- opt-in per-kernel timing with HIP events recorded on the launch stream -----------------
// bench.py uses this to measure the dominant kernel's average launch duration inside its timed
// region.  Process-global and not re-entrant by design (a measurement aid, off by default).
struct ProfState {
  int kernel = -1;
  int cap = 0, n = 0;
  hipEvent_t* ev = nullptr;  // 2 per record
};
ProfState g_prof;
unsigned long long* g_trunk_trace = nullptr;  // catre_debug_trunk_trace

// Smallest grid (in PAIRS of tiles) that takes the 128-point bf16 trunk; read once (CATRE_BF_PAIR_MIN overrides it for A/B
// measurements: 0 = always, a huge value = never).  512 pairs = two workgroups for each of the 256 CUs.
static int bf_pair_min() {
  static const int v = [] {
    const char* e = getenv("CATRE_BF_PAIR_MIN");
    return e ? atoi(e) : 512;
  }();
  return v;
}

// Which kernel FORM a full grid takes where more than one exists (all forms of a stage give the same bits; the switches
// exist for A/B measurements and for tests that compare the forms in one process).  Defaults: the encoder forms on,
// k_rot_l1w OFF (it measured 8 % slower than k_rot_l1<1>: profiles/r06_rotw_phases.txt), k_fc_tail OFF (one object:
// 0.643 vs 0.620 ms per K = 4 refine, profiles/r06_fc_tail_ab.jsonl); the environment (CATRE_TRUNK4 / CATRE_STN4 /
// CATRE_STN_PAIR = 0, CATRE_ROTW / CATRE_FC_TAIL = 1) sets the process default once, catre_form_switch changes it at run time.
enum { FORM_TRUNK4 = 1, FORM_STN4 = 2, FORM_STN_PAIR = 4, FORM_ROTW = 8, FORM_FC_TAIL = 16 };
static std::atomic<int> g_forms{-1};
static int forms() {
  int v = g_forms.load(std::memory_order_relaxed);
  if (v < 0) {
    auto on = [](const char* name, bool dflt = true) {
      const char* e = getenv(name);
      return e ? atoi(e) != 0 : dflt;
    };
    v = (on("CATRE_TRUNK4") ? FORM_TRUNK4 : 0) | (on("CATRE_STN4") ? FORM_STN4 : 0) |
        (on("CATRE_STN_PAIR") ? FORM_STN_PAIR : 0) | (on("CATRE_ROTW", false) ? FORM_ROTW : 0) |
        (on("CATRE_FC_TAIL", false) ? FORM_FC_TAIL : 0);
    g_forms.store(v, std::memory_order_relaxed);
  }
  return v;
}
static bool trunk4_on() { return forms() & FORM_TRUNK4; }  // one-wave-per-SIMD trunk (k_trunk4); off: the 8-wave kernel
inline int stn_pairs(int B, int N, int M) { return B * (((N + TP - 1) / TP + 1) / 2 + ((M + TP - 1) / TP + 1) / 2); }
static bool stn_pair_on() { return forms() & FORM_STN_PAIR; }  // off: one tile per workgroup
static bool stn4_on() { return forms() & FORM_STN4; }
static bool fc_tail_on() { return forms() & FORM_FC_TAIL; }  // B <= 8: an FC tail as one launch (k_fc_tail); off (default: the fused form measured SLOWER, profiles/r06_fc_tail_ab.jsonl): three k_linear launches
static bool rotw_on() { return forms() & FORM_ROTW; }  // rotation heads, one wave per SIMD (k_rot_l1w); off (default): k_rot_l1<1>

// The measurement hooks are the library's only process-global mutable state.  They are fenced: compiled out entirely
// with -DCATRE_NO_PROFILING (catre_profile_* then return CATRE_ERR_UNSUPPORTED), off unless catre_profile_enable was
// called, and the record table is guarded by a mutex so that concurrent callers of the data path cannot corrupt it.
#ifndef CATRE_NO_PROFILING
std::mutex g_prof_mu;
struct ProfScope {
  hipStream_t st;
  int slot = -1;
  ProfScope(int kernel_id, hipStream_t s) : st(s) {
    if (g_prof.kernel != kernel_id) return;  // the common case: one relaxed read, no lock
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.kernel == kernel_id && g_prof.n < g_prof.cap) {
      slot = g_prof.n++;
      (void)hipEventRecord(g_prof.ev[2 * slot], st);
    }
  }
  ~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < g_prof.cap) (void)hipEventRecord(g_prof.ev[2 * slot + 1], st);
  }
};
#else
struct ProfScope {
  ProfScope(int, hipStream_t) {}
};
#endif


// The three encoder kernels of one iteration (fp32 or split compute), tile partial maxima -> ws + W.pm
void launch_stn3d(const catre_points* pts, const float* const* prm, const float* packed, float* ws, const WsLayout& W,
                  int B, int N, int M, bool split, hipStream_t st, unsigned* zero_bar = nullptr) {
  const PackLayout L = pack_layout(1);  // conv offsets do not depend on ts_in
  const int tiles = B * ((N + TP - 1) / TP + (M + TP - 1) / TP);
  ProfScope ps(CATRE_K_STN3D, st);
  if (split) {
#define LAUNCH_(RS)                                                                                            \
  hipLaunchKernelGGL(k_stn3d_split<RS>, dim3(tiles * RS), dim3(256), 0, st, *pts, prm[CATRE_P_STN_CONV1_W],     \
                     prm[CATRE_P_STN_CONV1_B], pkb(packed, L.sp_stn_c2), prm[CATRE_P_STN_CONV2_B],              \
                     pkb(packed, L.sp_stn_c3), prm[CATRE_P_STN_CONV3_B], ws + W.pm, B, N, M)
    RS_DISPATCH(row_split(tiles), LAUNCH_)
#undef LAUNCH_
  } else if (row_split8(tiles) == 1 && stn4_on() && stn_pair_on() && stn_pairs(B, N, M) >= 256) {
    const int pairs = stn_pairs(B, N, M);  // (fewer pairs than CUs: one tile per workgroup keeps the whole chip busy)
    hipLaunchKernelGGL(k_stn3d_pair<false>, dim3(pairs), dim3(256), 0, st, *pts, prm[CATRE_P_STN_CONV1_W], prm[CATRE_P_STN_CONV1_B],
                       pk4(packed, L.stn_c2), prm[CATRE_P_STN_CONV2_B], pk4(packed, L.stn_c3), prm[CATRE_P_STN_CONV3_B],
                       ws + W.pm, B, N, M);
  } else if (row_split8(tiles) == 1 && stn4_on()) {
    hipLaunchKernelGGL((k_stn3d<1, false, true>), dim3(tiles), dim3(256), 0, st, *pts, prm[CATRE_P_STN_CONV1_W],
                       prm[CATRE_P_STN_CONV1_B], pk4(packed, L.stn_c2), prm[CATRE_P_STN_CONV2_B], pk4(packed, L.stn_c3),
                       prm[CATRE_P_STN_CONV3_B], ws + W.pm, B, N, M, TrainSave{}, zero_bar);
  } else {
#define LAUNCH_(RS)                                                                                      \
  hipLaunchKernelGGL(k_stn3d<RS>, dim3(tiles * RS), dim3(256), 0, st, *pts, prm[CATRE_P_STN_CONV1_W],     \
                     prm[CATRE_P_STN_CONV1_B], pk4(packed, L.stn_c2), prm[CATRE_P_STN_CONV2_B],           \
                     pk4(packed, L.stn_c3), prm[CATRE_P_STN_CONV3_B], ws + W.pm, B, N, M, TrainSave{}, zero_bar)
    RS_DISPATCH8(row_split8(tiles), LAUNCH_)
#undef LAUNCH_
  }
}

void launch_stnkd(const catre_points* pts, const float* trans3, const float* const* prm, const float* packed, float* ws,
                  const WsLayout& W, int B, int N, int M, bool split, hipStream_t st, unsigned* zero_bar = nullptr) {
  const PackLayout L = pack_layout(1);
  const int tiles = B * ((N + TP - 1) / TP + (M + TP - 1) / TP);
  ProfScope ps(CATRE_K_STNKD, st);
  if (split) {
#define LAUNCH_(RS)                                                                                                \
  hipLaunchKernelGGL(k_stnkd_split<RS>, dim3(tiles * RS), dim3(256), 0, st, *pts, trans3, prm[CATRE_P_CONV1_W],     \
                     prm[CATRE_P_CONV1_B], pkb(packed, L.sp_fstn_c1), prm[CATRE_P_FSTN_CONV1_B],                    \
                     pkb(packed, L.sp_fstn_c2), prm[CATRE_P_FSTN_CONV2_B], pkb(packed, L.sp_fstn_c3),               \
                     prm[CATRE_P_FSTN_CONV3_B], ws + W.pm, B, N, M)
    RS_DISPATCH(row_split(tiles), LAUNCH_)
#undef LAUNCH_
  } else if (row_split8(tiles) == 1 && stn4_on() && stn_pair_on() && stn_pairs(B, N, M) >= 256) {
    const int pairs = stn_pairs(B, N, M);
    hipLaunchKernelGGL(k_stnkd_pair<false>, dim3(pairs), dim3(256), 0, st, *pts, trans3, prm[CATRE_P_CONV1_W], prm[CATRE_P_CONV1_B],
                       pk4(packed, L.fstn_c1), prm[CATRE_P_FSTN_CONV1_B], pk4(packed, L.fstn_c2), prm[CATRE_P_FSTN_CONV2_B],
                       pk4(packed, L.fstn_c3), prm[CATRE_P_FSTN_CONV3_B], ws + W.pm, B, N, M);
  } else if (row_split8(tiles) == 1 && stn4_on()) {
    hipLaunchKernelGGL((k_stnkd<1, false, true>), dim3(tiles), dim3(256), 0, st, *pts, trans3, prm[CATRE_P_CONV1_W],
                       prm[CATRE_P_CONV1_B], pk4(packed, L.fstn_c1), prm[CATRE_P_FSTN_CONV1_B], pk4(packed, L.fstn_c2),
                       prm[CATRE_P_FSTN_CONV2_B], pk4(packed, L.fstn_c3), prm[CATRE_P_FSTN_CONV3_B], ws + W.pm, B, N, M,
                       TrainSave{}, zero_bar);
  } else {
#define LAUNCH_(RS)                                                                                                     \
  hipLaunchKernelGGL(k_stnkd<RS>, dim3(tiles * RS), dim3(256), 0, st, *pts, trans3, prm[CATRE_P_CONV1_W],                \
                     prm[CATRE_P_CONV1_B], pk4(packed, L.fstn_c1), prm[CATRE_P_FSTN_CONV1_B], pk4(packed, L.fstn_c2),    \
                     prm[CATRE_P_FSTN_CONV2_B], pk4(packed, L.fstn_c3), prm[CATRE_P_FSTN_CONV3_B], ws + W.pm, B, N, M,   \
                     TrainSave{}, zero_bar)
    RS_DISPATCH8(row_split8(tiles), LAUNCH_)
#undef LAUNCH_
  }
}

void launch_trunk(const catre_points* pts, const float* trans3, const float* trans64, const float* const* prm,
                  const float* packed, float* pointfeat, float* ws, const WsLayout& W, int B, int N, int M, bool split,
                  hipStream_t st) {
  const PackLayout L = pack_layout(1);
  const int tiles = B * ((N + TP - 1) / TP + (M + TP - 1) / TP);
  ProfScope ps(CATRE_K_TRUNK, st);
  if (split) {
#define LAUNCH_(RS)                                                                                           \
  hipLaunchKernelGGL(k_trunk_split<RS>, dim3(tiles * RS), dim3(512), 0, st, *pts, trans3, trans64,             \
                     prm[CATRE_P_CONV1_W], prm[CATRE_P_CONV1_B], pk4(packed, L.c2), prm[CATRE_P_CONV2_B],      \
                     pkb(packed, L.sp_c3), prm[CATRE_P_CONV3_B], pkb(packed, L.sp_c4), prm[CATRE_P_CONV4_B],   \
                     ws + W.pm, pointfeat, B, N, M, g_trunk_trace)
    RS_DISPATCH(row_split(tiles), LAUNCH_)
#undef LAUNCH_
  } else if (row_split8(tiles) == 1 && trunk4_on()) {
    hipLaunchKernelGGL(k_trunk4<false>, dim3(tiles), dim3(256), 0, st, *pts, trans3, trans64, prm[CATRE_P_CONV1_W],
                       prm[CATRE_P_CONV1_B], pk4(packed, L.c2), prm[CATRE_P_CONV2_B], pk4(packed, L.c3),
                       prm[CATRE_P_CONV3_B], pk4(packed, L.c4), prm[CATRE_P_CONV4_B], ws + W.pm, pointfeat, B, N, M,
                       g_trunk_trace);
  } else {
#define LAUNCH_(RS)                                                                                             \
  hipLaunchKernelGGL(k_trunk<RS>, dim3(tiles * RS), dim3(512), 0, st, *pts, trans3, trans64, prm[CATRE_P_CONV1_W], \
                     prm[CATRE_P_CONV1_B], pk4(packed, L.c2), prm[CATRE_P_CONV2_B], pk4(packed, L.c3),            \
                     prm[CATRE_P_CONV3_B], pk4(packed, L.c4), prm[CATRE_P_CONV4_B], ws + W.pm, pointfeat, B, N, M, \
                     g_trunk_trace)
    RS_DISPATCH8(row_split8(tiles), LAUNCH_)
#undef LAUNCH_
  }
}

}  // namespace

extern "C" {

const char* catre_version(void) { return CATRE_VERSION_STR; }

const char* catre_status_string(int s) {
  switch (s) {
    case CATRE_OK: return "ok";
    case CATRE_ERR_BAD_ARG: return "bad argument";
    case CATRE_ERR_WORKSPACE: return "workspace or packed-weight buffer too small";
    case CATRE_ERR_LAUNCH: return "kernel launch failed";
    case CATRE_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}

size_t catre_workspace_bytes(int B, int N, int M) {
  if (!dims_ok1(B, N, M)) return 0;
  return ws_layout(B, N, M).total * sizeof(float);
}

size_t catre_packed_floats(int N, int M, int ts_in_dim) {
  if (N <= 0 || M <= 0 || ts_in_dim <= 0) return 0;
  return pack_layout(ts_in_dim).total;
}

int catre_pack_weights(const float* const* prm, int N, int M, int ts_in, float* packed, size_t packed_floats,
                       void* stream) {
  return catre_pack_weights_sel(prm, N, M, ts_in, packed, packed_floats, CATRE_PACK_ALL, stream);
}

int catre_pack_weights_sel(const float* const* prm, int N, int M, int ts_in, float* packed, size_t packed_floats, int sel,
                           void* stream) {
  REQUIRE(prm && packed && N > 0 && M > 0 && ts_in > 0);
  const PackLayout L = pack_layout(ts_in);
  if (packed_floats < L.total) return CATRE_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  // a NULL source is skipped, so a sub-module (e.g. PointNetfeat alone) can pack just its own layers
  const bool enc32 = sel & CATRE_PACK_F32_ENCODER, head32 = sel & CATRE_PACK_F32_HEADS, bf = sel & CATRE_PACK_BF16,
             sp = sel & CATRE_PACK_SPLIT, tails = sel & CATRE_PACK_F32_TAILS;
  PackJobs jobs;
  jobs.n = 0;
  auto flush = [&]() {
    if (jobs.n) hipLaunchKernelGGL(k_pack_frag_multi, dim3((jobs.end[jobs.n - 1] + 255) / 256), dim3(256), 0, st, jobs);
    jobs.n = 0;
  };
  auto frag = [&](const float* src, int ld, int coloff, int rows, int K, size_t off) {  // queued: one launch for all
    if (!src) return;
    if (jobs.n == PACK_MAX_JOBS) flush();
    const int j = jobs.n++;
    jobs.src[j] = src;
    jobs.dst[j] = packed + off;
    jobs.ld[j] = ld;
    jobs.coloff[j] = coloff;
    jobs.K[j] = K;
    jobs.end[j] = (j ? jobs.end[j - 1] : 0) + rows * K;
  };
  PackJobs lpjobs;   // the bf16 / split packs: queued like the fp32 ones, one launch per kind
  lpjobs.n = 0;
  auto flush_lp = [&](bool split_kind) {
    if (lpjobs.n) {
      const dim3 grid((lpjobs.end[lpjobs.n - 1] + 255) / 256);
      if (split_kind)
        hipLaunchKernelGGL(k_pack_frag_lp_multi<true>, grid, dim3(256), 0, st, lpjobs);
      else
        hipLaunchKernelGGL(k_pack_frag_lp_multi<false>, grid, dim3(256), 0, st, lpjobs);
    }
    lpjobs.n = 0;
  };
  auto frag_lp = [&](bool split_kind, const float* src, int ld, int coloff, int rows, int K, size_t off) {
    if (!src) return;
    if (lpjobs.n == PACK_MAX_JOBS) flush_lp(split_kind);
    const int j = lpjobs.n++;
    lpjobs.src[j] = src;
    lpjobs.dst[j] = packed + off;
    lpjobs.ld[j] = ld;
    lpjobs.coloff[j] = coloff;
    lpjobs.K[j] = K;
    lpjobs.end[j] = (j ? lpjobs.end[j - 1] : 0) + rows * K;
  };
  auto frag_bf = [&](const float* src, int ld, int coloff, int rows, int K, size_t off) {
    frag_lp(false, src, ld, coloff, rows, K, off);
  };
  if (bf) {
  frag_bf(prm[CATRE_P_STN_CONV2_W], 64, 0, 128, 64, L.bf_stn_c2);
  frag_bf(prm[CATRE_P_STN_CONV3_W], 128, 0, 1024, 128, L.bf_stn_c3);
  frag_bf(prm[CATRE_P_FSTN_CONV1_W], 64, 0, 64, 64, L.bf_fstn_c1);
  frag_bf(prm[CATRE_P_FSTN_CONV2_W], 64, 0, 128, 64, L.bf_fstn_c2);
  frag_bf(prm[CATRE_P_FSTN_CONV3_W], 128, 0, 1024, 128, L.bf_fstn_c3);
  frag_bf(prm[CATRE_P_CONV2_W], 64, 0, 128, 64, L.bf_c2);
  frag_bf(prm[CATRE_P_CONV3_W], 128, 0, 512, 128, L.bf_c3);
  frag_bf(prm[CATRE_P_CONV4_W], 512, 0, 1024, 512, L.bf_c4);
  for (int h = 0; h < 2; ++h) {
    const int base = h ? CATRE_P_ROTY_L0_W : CATRE_P_ROTX_L0_W;
    frag_bf(prm[base], PMW, 1024, 256, 64, L.bf_rot_l0[h]);
    frag_bf(prm[base + 4], 256, 0, 256, 256, L.bf_rot_l1[h]);
  }
  flush_lp(false);
  }
  auto frag_sp = [&](const float* src, int ld, int rows, int K, size_t off, int coloff = 0) {
    frag_lp(true, src, ld, coloff, rows, K, off);
  };
  if (sp) {
  frag_sp(prm[CATRE_P_STN_CONV2_W], 64, 128, 64, L.sp_stn_c2);
  frag_sp(prm[CATRE_P_STN_CONV3_W], 128, 1024, 128, L.sp_stn_c3);
  frag_sp(prm[CATRE_P_FSTN_CONV1_W], 64, 64, 64, L.sp_fstn_c1);
  frag_sp(prm[CATRE_P_FSTN_CONV2_W], 64, 128, 64, L.sp_fstn_c2);
  frag_sp(prm[CATRE_P_FSTN_CONV3_W], 128, 1024, 128, L.sp_fstn_c3);
  frag_sp(prm[CATRE_P_CONV3_W], 128, 512, 128, L.sp_c3);
  frag_sp(prm[CATRE_P_CONV4_W], 512, 1024, 512, L.sp_c4);
  frag_sp(prm[CATRE_P_ROTX_L0_W], PMW, 256, 64, L.sp_rot_l0[0], 1024);
  frag_sp(prm[CATRE_P_ROTY_L0_W], PMW, 256, 64, L.sp_rot_l0[1], 1024);
  frag_sp(prm[CATRE_P_ROTX_L0_W + 4], 256, 256, 256, L.sp_rot_l1[0]);
  frag_sp(prm[CATRE_P_ROTY_L0_W + 4], 256, 256, 256, L.sp_rot_l1[1]);
  flush_lp(true);
  }
  if (enc32) {
  frag(prm[CATRE_P_STN_CONV2_W], 64, 0, 128, 64, L.stn_c2);
  frag(prm[CATRE_P_STN_CONV3_W], 128, 0, 1024, 128, L.stn_c3);
  frag(prm[CATRE_P_FSTN_CONV1_W], 64, 0, 64, 64, L.fstn_c1);
  frag(prm[CATRE_P_FSTN_CONV2_W], 64, 0, 128, 64, L.fstn_c2);
  frag(prm[CATRE_P_FSTN_CONV3_W], 128, 0, 1024, 128, L.fstn_c3);
  frag(prm[CATRE_P_CONV2_W], 64, 0, 128, 64, L.c2);
  frag(prm[CATRE_P_CONV3_W], 128, 0, 512, 128, L.c3);
  frag(prm[CATRE_P_CONV4_W], 512, 0, 1024, 512, L.c4);
  }
  for (int h = 0; h < 2 && head32; ++h) {
    const int base = h ? CATRE_P_ROTY_L0_W : CATRE_P_ROTX_L0_W;
    frag(prm[base], PMW, 1024, 256, 64, L.rot_l0[h]);  // W0[:, 1024:1088]
    frag(prm[base + 4], 256, 0, 256, 256, L.rot_l1[h]);
    if (prm[base + 10] && tails)
      hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, st, prm[base + 10], N + M, packed + L.sumwp + h);
  }
  flush();
  if (head32 && tails && prm[CATRE_P_TS_L0_W] && prm[CATRE_P_TS_L1_W]) {
    int n = ts_in * 256;
    hipLaunchKernelGGL(k_pack_transpose, dim3((n + 255) / 256), dim3(256), 0, st, prm[CATRE_P_TS_L0_W], 256, ts_in,
                       packed + L.ts_w0t);
    n = 256 * 256;
    hipLaunchKernelGGL(k_pack_transpose, dim3((n + 255) / 256), dim3(256), 0, st, prm[CATRE_P_TS_L1_W], 256, 256,
                       packed + L.ts_w1t);
  }
  return check_launch();
}

int catre_pose_apply(const float* pcl, const float* kps, const float* pose, const float* scale, float* x_out,
                     float* kps_out, int B, int N, int M, int zero_center, void* stream) {
  REQUIRE(pcl && kps && pose && scale && x_out && kps_out && dims_ok(B, N, M));
  const int total = B * (N + M);
  hipLaunchKernelGGL(k_pose_apply, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, pcl, kps, pose, scale,
                     x_out, kps_out, B, N, M, zero_center);
  return check_launch();
}

int catre_stn3d_pool(const catre_points* pts, const float* const* prm, const float* packed, float* pooled,
                     void* workspace, size_t ws_bytes, int B, int N, int M, void* stream) {
  REQUIRE(pts && prm && packed && pooled && workspace && dims_ok1(B, N, M));
  const WsLayout W = ws_layout(B, N, M);
  if (ws_bytes < W.total * sizeof(float)) return CATRE_ERR_WORKSPACE;
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  launch_stn3d(pts, prm, packed, ws, W, B, N, M, false, st);
  hipLaunchKernelGGL(k_reduce_pm, dim3(n_clouds(B, M), (1024 + 255) / 256), dim3(256), 0, st, ws + W.pm, pooled, 1024, 1024, B, N, M);
  return check_launch();
}

int catre_linear(const float* x, int ldx, const float* Wt, int ldw, const float* bias, float* y, int ldy, int R, int J,
                 int K, int relu, int add_identity_k, void* stream) {
  REQUIRE(x && Wt && y && R > 0 && J > 0 && K > 0 && (K % 8) == 0 && (ldx % 4) == 0 && (ldw % 4) == 0);
  hipLaunchKernelGGL(k_linear, dim3((R + 31) / 32, (J + 31) / 32), dim3(64 * LIN_WAVES), 0, (hipStream_t)stream, x, ldx, Wt, ldw,
                     bias, y, ldy, R, J, K, relu, add_identity_k);
  return check_launch();
}

// y[R,J] = x[R,K] W for W [K][J] row-major (ldw): the data gradient of a small linear from its own weight, no transposed copy
// xmask (optional, [R,K] with x's row pitch): x .* (xmask > 0) replaces x - the ReLU backward of the layer's output
int catre_linear_t(const float* x, int ldx, const float* xmask, const float* W, int ldw, float* y, int ldy, int R, int J,
                   int K, void* stream) {
  REQUIRE(x && W && y && R > 0 && J > 0 && K > 0 && (K % 8) == 0 && (ldx % 4) == 0 && ldw >= J);
  hipLaunchKernelGGL(k_linear_t, dim3((R + 31) / 32, (J + 31) / 32), dim3(64 * LIN_WAVES), 0, (hipStream_t)stream, x, ldx, W,
                     ldw, y, ldy, R, J, K, xmask);
  return check_launch();
}

int catre_stnkd_pool(const catre_points* pts, const float* trans3, const float* const* prm, const float* packed,
                     float* pooled, void* workspace, size_t ws_bytes, int B, int N, int M, void* stream) {
  REQUIRE(pts && trans3 && prm && packed && pooled && workspace && dims_ok1(B, N, M));
  const WsLayout W = ws_layout(B, N, M);
  if (ws_bytes < W.total * sizeof(float)) return CATRE_ERR_WORKSPACE;
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  launch_stnkd(pts, trans3, prm, packed, ws, W, B, N, M, false, st);
  hipLaunchKernelGGL(k_reduce_pm, dim3(n_clouds(B, M), (1024 + 255) / 256), dim3(256), 0, st, ws + W.pm, pooled, 1024, 1024, B, N, M);
  return check_launch();
}

int catre_trunk(const catre_points* pts, const float* trans3, const float* trans64, const float* const* prm,
                const float* packed, float* gfeat, float* pointfeat, void* workspace, size_t ws_bytes, int B, int N,
                int M, void* stream) {
  REQUIRE(pts && trans3 && prm && packed && gfeat && pointfeat && workspace && dims_ok1(B, N, M));
  const WsLayout W = ws_layout(B, N, M);
  if (ws_bytes < W.total * sizeof(float)) return CATRE_ERR_WORKSPACE;
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  launch_trunk(pts, trans3, trans64, prm, packed, pointfeat, ws, W, B, N, M, false, st);
  hipLaunchKernelGGL(k_reduce_pm, dim3(n_clouds(B, M), (PMW + 255) / 256), dim3(256), 0, st, ws + W.pm, gfeat, PMW, PMW, B, N, M);
  return check_launch();
}

int catre_ts_head(const float* gfeat, const float* init_pose, const float* init_scale, const float* const* prm,
                  const float* packed, const catre_opts* o, float* trans_deltas, float* scale_deltas, void* workspace,
                  size_t ws_bytes, int B, void* stream) {
  REQUIRE(gfeat && init_pose && init_scale && prm && packed && o && trans_deltas && scale_deltas && B > 0);
  const int expect = PMW * (o->with_kps_feature ? 2 : 1) + (o->with_init_scale ? 3 : 0) + (o->with_init_trans ? 3 : 0);
  if (o->ts_in_dim != expect) return CATRE_ERR_BAD_ARG;
  const PackLayout L = pack_layout(o->ts_in_dim);
  hipStream_t st = (hipStream_t)stream;
  if (!workspace || ws_bytes < (size_t)B * TS_KS * 256 * sizeof(float)) return CATRE_ERR_WORKSPACE;
  float* l0part = (float*)workspace;  // WsLayout::tspart sits at offset 0 for every (N, M)
  const int groups = (B + TS_OB - 1) / TS_OB;
  {
    ProfScope ps(CATRE_K_TS_HEAD, st);
    const size_t smem0 = (size_t)TS_OB * (o->ts_in_dim / TS_KS + 2) * sizeof(float);
    hipLaunchKernelGGL(k_ts_l0, dim3(groups, TS_KS), dim3(256), smem0, st, gfeat, init_pose, init_scale, packed + L.ts_w0t,
                       l0part, B, o->ts_in_dim, o->with_kps_feature, o->with_init_scale, o->with_init_trans);
    const size_t smem = (size_t)TS_OB * (256 + 4 * 256) * sizeof(float);
    hipLaunchKernelGGL(k_ts_head, dim3(groups), dim3(1024), smem, st,
                       ts_head_args(l0part, prm, packed, L, trans_deltas, scale_deltas, B));
  }
  return check_launch();
}

static int rot_head_impl(const float* gfeat, const float* pointfeat, const float* const* prm, const float* packed,
                         float* rot6d, float* ws, const WsLayout& W, int B, int N, int M, hipStream_t st,
                         bool split = false, int rd = 3) {
  const PackLayout L = pack_layout(1);
  const int T = (N + TP - 1) / TP + (M + TP - 1) / TP;
  float* bias0 = ws + W.bias0;
  // global-feature half of layer 0 for every cloud: bias0[hd][cloud][:] = W0[:, :1024] g_cloud + b0
  hipLaunchKernelGGL(k_linear, dim3((2 * B + 31) / 32, 256 / 32, 2), dim3(64 * LIN_WAVES), 0, st, gfeat, PMW,
                     prm[CATRE_P_ROTX_L0_W], PMW, prm[CATRE_P_ROTX_L0_W + 1], bias0, 256, 2 * B, 256, 1024, 0, 0,
                     prm[CATRE_P_ROTY_L0_W], prm[CATRE_P_ROTY_L0_W + 1], bias0 + (size_t)2 * B * 256);
  // GN0 statistics from second moments of pointfeat (catre_gram.h); the moment buffers borrow y1, which is only
  // written by k_rot_l1 afterwards
  float* Gc = ws + W.y1;
  float* s1c = Gc + (size_t)2 * B * PF_NG * 4096;
  float* shc = s1c + (size_t)2 * B * PF_NG * 64;
  {
    ProfScope ps(CATRE_K_ROT_L0_STATS, st);
    // a handful of clouds: the four tile groups of a cloud on four workgroups (same partial sums, same result)
    hipLaunchKernelGGL(k_pf_moments, dim3(2 * B, pf_groups(B)), dim3(256), 0, st, pointfeat, Gc, s1c,
                       shc, B, N, M);
    hipLaunchKernelGGL(k_gn0_from_moments, dim3(B, gn0_shares(B)), dim3(256), 0, st, Gc, s1c, shc, prm[CATRE_P_ROTX_L0_W],
                       prm[CATRE_P_ROTY_L0_W], PMW, 1024, bias0, prm[CATRE_P_ROTX_GN0_W], prm[CATRE_P_ROTX_GN0_B],
                       prm[CATRE_P_ROTY_GN0_W], prm[CATRE_P_ROTY_GN0_B], ws + W.aff0, B, N, M);
  }
  {
    ProfScope ps(CATRE_K_ROT_L1, st);
    if (split)
      hipLaunchKernelGGL(k_rot_l1_split<false>, dim3(B * T), dim3(256), 0, st, pointfeat, pkb(packed, L.sp_rot_l0[0]),
                         pkb(packed, L.sp_rot_l0[1]), ws + W.aff0, pkb(packed, L.sp_rot_l1[0]), pkb(packed, L.sp_rot_l1[1]),
                         prm[CATRE_P_ROTX_L1_B], prm[CATRE_P_ROTY_L1_B], ws + W.y1, ws + W.gn1, B, N, M,
                         g_trunk_trace ? g_trunk_trace + ((size_t)1 << 24) : nullptr);
    else if (B * T >= ROTW_MIN_TILES && rotw_on()) {
      // grids that fill the chip: one wave per SIMD, a tile per wave (same bits as k_rot_l1)
      hipLaunchKernelGGL(k_rot_l1w, dim3((B * T + 3) / 4), dim3(256), 0, st, pointfeat, pk4(packed, L.rot_l0[0]),
                         pk4(packed, L.rot_l0[1]), ws + W.aff0, pk4(packed, L.rot_l1[0]), pk4(packed, L.rot_l1[1]),
                         prm[CATRE_P_ROTX_L1_B], prm[CATRE_P_ROTY_L1_B], ws + W.y1, ws + W.gn1, B, N, M,
                         g_trunk_trace ? g_trunk_trace + ((size_t)1 << 24) : nullptr);
    } else {
#define LAUNCH_ROT_L1(RS)                                                                                               \
  hipLaunchKernelGGL(k_rot_l1<RS>, dim3(B * T * RS), dim3(256), 0, st, pointfeat, pk4(packed, L.rot_l0[0]),               \
                     pk4(packed, L.rot_l0[1]), ws + W.aff0, pk4(packed, L.rot_l1[0]), pk4(packed, L.rot_l1[1]),         \
                     prm[CATRE_P_ROTX_L1_B], prm[CATRE_P_ROTY_L1_B], ws + W.y1, ws + W.gn1, B, N, M,                     \
                     g_trunk_trace ? g_trunk_trace + ((size_t)1 << 24) : nullptr)
      RS_DISPATCH(row_split(B * T), LAUNCH_ROT_L1)
#undef LAUNCH_ROT_L1
    }
  }
  hipLaunchKernelGGL(k_gn_finalize, dim3(B * 2), dim3(256), (size_t)T * 64 * sizeof(float), st, ws + W.gn1, ws + W.gn1stat,
                     N, M);
  {
    ProfScope ps(CATRE_K_ROT_OUT, st);
    hipLaunchKernelGGL(k_rot_out, dim3(B * T, 2), dim3(256), 0, st, ws + W.y1, ws + W.gn1stat,
                       prm[CATRE_P_ROTX_GN1_W], prm[CATRE_P_ROTX_GN1_B], prm[CATRE_P_ROTY_GN1_W],
                       prm[CATRE_P_ROTY_GN1_B], prm[CATRE_P_ROTX_NECK_W], prm[CATRE_P_ROTY_NECK_W],
                       prm[CATRE_P_ROTX_CONVP_W], prm[CATRE_P_ROTY_CONVP_W], ws + W.rpart, B, N, M, rd);
  }
  hipLaunchKernelGGL(k_rot_finish, dim3((B * 6 + 255) / 256), dim3(256), 0, st, ws + W.rpart, prm[CATRE_P_ROTX_NECK_B],
                     prm[CATRE_P_ROTY_NECK_B], packed + L.sumwp, prm[CATRE_P_ROTX_CONVP_B], prm[CATRE_P_ROTY_CONVP_B],
                     rot6d, B, T, rd);
  return check_launch();
}

int catre_rot_head_dim(const float* gfeat, const float* pointfeat, const float* const* prm, const float* packed,
                       float* rot, void* workspace, size_t ws_bytes, int B, int N, int M, int rot_dim, void* stream) {
  REQUIRE(gfeat && pointfeat && prm && packed && rot && workspace && dims_ok(B, N, M) && rot_dim >= 1 && rot_dim <= 3);
  const WsLayout W = ws_layout(B, N, M);
  if (ws_bytes < W.total * sizeof(float)) return CATRE_ERR_WORKSPACE;
  return rot_head_impl(gfeat, pointfeat, prm, packed, rot, (float*)workspace, W, B, N, M, (hipStream_t)stream, false,
                       rot_dim);
}

int catre_rot_to_mat(const float* rot, int rot_type, float* R_out, int B, void* stream) {
  REQUIRE(rot && R_out && B > 0 && rot_type >= CATRE_ROT_6D && rot_type <= CATRE_ROT_LIE_VEC);
  hipLaunchKernelGGL(k_rot_to_mat, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, rot, rot_type, R_out, B);
  return check_launch();
}

int catre_rot_to_mat_bwd(const float* rot, int rot_type, const float* grad_R, float* grad_rot, int B, void* stream) {
  REQUIRE(rot && grad_R && grad_rot && B > 0 && rot_type >= CATRE_ROT_6D && rot_type <= CATRE_ROT_LIE_VEC);
  hipLaunchKernelGGL(k_rot_to_mat_bwd, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, rot, rot_type, grad_R,
                     grad_rot, B);
  return check_launch();
}

int catre_rot_head(const float* gfeat, const float* pointfeat, const float* const* prm, const float* packed,
                   float* rot6d, void* workspace, size_t ws_bytes, int B, int N, int M, void* stream) {
  REQUIRE(gfeat && pointfeat && prm && packed && rot6d && workspace && dims_ok(B, N, M));
  const WsLayout W = ws_layout(B, N, M);
  if (ws_bytes < W.total * sizeof(float)) return CATRE_ERR_WORKSPACE;
  return rot_head_impl(gfeat, pointfeat, prm, packed, rot6d, (float*)workspace, W, B, N, M, (hipStream_t)stream);
}

static int pose_update_impl(const float* rot6d, const float* trans_deltas, const float* scale_deltas,
                            const float* init_pose, const float* init_scale, const float* mean_scales, const float* Ks,
                            const catre_opts* o, float* pose_out, float* scale_out, int B, void* stream,
                            float* pose_echo, float* scale_echo) {
  REQUIRE(rot6d && trans_deltas && scale_deltas && init_pose && init_scale && o && pose_out && scale_out && B > 0);
  if (o->k_aware && !o->delta_t_space_3d && !Ks) return CATRE_ERR_BAD_ARG;
  if (o->scale_base_mean && !mean_scales) return CATRE_ERR_BAD_ARG;
  if (o->rot_type < CATRE_ROT_6D || o->rot_type > CATRE_ROT_LIE_VEC) return CATRE_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_pose_update, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, rot6d, trans_deltas,
                     scale_deltas, init_pose, init_scale, mean_scales, Ks, *o, pose_out, scale_out, B, pose_echo,
                     scale_echo);
  return check_launch();
}

int catre_pose_update(const float* rot6d, const float* trans_deltas, const float* scale_deltas, const float* init_pose,
                      const float* init_scale, const float* mean_scales, const float* Ks, const catre_opts* o,
                      float* pose_out, float* scale_out, int B, void* stream) {
  return pose_update_impl(rot6d, trans_deltas, scale_deltas, init_pose, init_scale, mean_scales, Ks, o, pose_out, scale_out,
                          B, stream, nullptr, nullptr);
}

// One refine iteration on the bf16-operand kernels (catre_bf16.h); same launch chain, same workspace (pointfeat and
// y1 hold bf16 in their fp32-sized slots), fp32 FC tails / ts head / pose update shared with the fp32 path.
static int refine_iter_bf(const catre_points* pts, const float* init_pose, const float* init_scale,
                          const float* mean_scales, const float* Ks, const float* const* prm, const float* packed,
                          const catre_opts* o, float* pose_out, float* scale_out, float* ws, const WsLayout& W, int B,
                          int N, int M, hipStream_t st, float* pose_echo, float* scale_echo) {
  const PackLayout L = pack_layout(1);
  const int TN = (N + TP - 1) / TP, TM = (M + TP - 1) / TP, T = TN + TM, tiles = B * T;
  const int rd = catre_rot_dim(o->rot_type) / 2;
  int rc;
  // grids that fill the chip twice over take PAIRS of tiles per workgroup in all three encoder kernels (same bits, fewer
  // prologues and weight fetches per MFMA); below that the 64-point kernels spread better
  const int pairs = B * ((TN + 1) / 2 + (TM + 1) / 2);
  const bool paired = pairs >= bf_pair_min();
  {
    ProfScope ps(CATRE_K_STN3D, st);
    if (paired)
      hipLaunchKernelGGL(k_stn3d_bf2<false>, dim3(pairs), dim3(256), 0, st, *pts, prm[CATRE_P_STN_CONV1_W],
                         prm[CATRE_P_STN_CONV1_B], pkb(packed, L.bf_stn_c2), prm[CATRE_P_STN_CONV2_B],
                         pkb(packed, L.bf_stn_c3), prm[CATRE_P_STN_CONV3_B], ws + W.pm, B, N, M);
    else
      hipLaunchKernelGGL((k_stn3d_bf<false>), dim3(tiles), dim3(256), 0, st, *pts, prm[CATRE_P_STN_CONV1_W],
                         prm[CATRE_P_STN_CONV1_B], pkb(packed, L.bf_stn_c2), prm[CATRE_P_STN_CONV2_B],
                         pkb(packed, L.bf_stn_c3), prm[CATRE_P_STN_CONV3_B], ws + W.pm, B, N, M);
  }
  hipLaunchKernelGGL(k_reduce_pm, dim3(2 * B, (1024 + 255) / 256), dim3(256), 0, st, ws + W.pm, ws + W.pool, 1024, 1024, B, N, M);
  if ((rc = stn_fc_tail(ws + W.pool, prm, CATRE_P_STN_FC1_W, ws + W.h1, ws + W.h2, ws + W.trans3, 3, 2 * B, st)))
    return rc;
  const float* t64 = nullptr;
  if (o->feature_transform) {
    {
      ProfScope ps(CATRE_K_STNKD, st);
      if (paired)
        hipLaunchKernelGGL(k_stnkd_bf2<false>, dim3(pairs), dim3(256), 0, st, *pts, ws + W.trans3, prm[CATRE_P_CONV1_W],
                           prm[CATRE_P_CONV1_B], pkb(packed, L.bf_fstn_c1), prm[CATRE_P_FSTN_CONV1_B],
                           pkb(packed, L.bf_fstn_c2), prm[CATRE_P_FSTN_CONV2_B], pkb(packed, L.bf_fstn_c3),
                           prm[CATRE_P_FSTN_CONV3_B], ws + W.pm, B, N, M);
      else
        hipLaunchKernelGGL((k_stnkd_bf<false>), dim3(tiles), dim3(256), 0, st, *pts, ws + W.trans3, prm[CATRE_P_CONV1_W],
                           prm[CATRE_P_CONV1_B], pkb(packed, L.bf_fstn_c1), prm[CATRE_P_FSTN_CONV1_B],
                           pkb(packed, L.bf_fstn_c2), prm[CATRE_P_FSTN_CONV2_B], pkb(packed, L.bf_fstn_c3),
                           prm[CATRE_P_FSTN_CONV3_B], ws + W.pm, B, N, M);
    }
    hipLaunchKernelGGL(k_reduce_pm, dim3(2 * B, (1024 + 255) / 256), dim3(256), 0, st, ws + W.pm, ws + W.pool, 1024, 1024, B, N, M);
    if ((rc = stn_fc_tail(ws + W.pool, prm, CATRE_P_FSTN_FC1_W, ws + W.h1, ws + W.h2, ws + W.trans64, 64, 2 * B, st)))
      return rc;
    t64 = ws + W.trans64;
  }
  u32x4* pointfeat = reinterpret_cast<u32x4*>(ws + W.pointfeat);
  {
    ProfScope ps(CATRE_K_TRUNK, st);
    if (paired)
      hipLaunchKernelGGL((k_trunk_bf2<false>), dim3(pairs), dim3(512), 0, st, *pts, ws + W.trans3, t64, prm[CATRE_P_CONV1_W],
                         prm[CATRE_P_CONV1_B], pkb(packed, L.bf_c2), prm[CATRE_P_CONV2_B], pkb(packed, L.bf_c3),
                         prm[CATRE_P_CONV3_B], pkb(packed, L.bf_c4), prm[CATRE_P_CONV4_B], ws + W.pm, pointfeat, B, N, M,
                         g_trunk_trace);
    else
      hipLaunchKernelGGL(k_trunk_bf, dim3(tiles), dim3(256), 0, st, *pts, ws + W.trans3, t64, prm[CATRE_P_CONV1_W],
                         prm[CATRE_P_CONV1_B], pkb(packed, L.bf_c2), prm[CATRE_P_CONV2_B], pkb(packed, L.bf_c3),
                         prm[CATRE_P_CONV3_B], pkb(packed, L.bf_c4), prm[CATRE_P_CONV4_B], ws + W.pm, pointfeat, B, N, M,
                         g_trunk_trace);
  }
  hipLaunchKernelGGL(k_reduce_pm, dim3(2 * B, (PMW + 255) / 256), dim3(256), 0, st, ws + W.pm, ws + W.gfeat, PMW, PMW, B, N, M);
  if ((rc = catre_ts_head(ws + W.gfeat, init_pose, init_scale, prm, packed, o, ws + W.dt, ws + W.ds, ws,
                          W.total * sizeof(float), B, (void*)st)))
    return rc;
  float* bias0 = ws + W.bias0;
  hipLaunchKernelGGL(k_linear, dim3((2 * B + 31) / 32, 256 / 32, 2), dim3(64 * LIN_WAVES), 0, st, ws + W.gfeat, PMW,
                     prm[CATRE_P_ROTX_L0_W], PMW, prm[CATRE_P_ROTX_L0_W + 1], bias0, 256, 2 * B, 256, 1024, 0, 0,
                     prm[CATRE_P_ROTY_L0_W], prm[CATRE_P_ROTY_L0_W + 1], bias0 + (size_t)2 * B * 256);
  {
    // GN0 statistics from second moments of the (bf16) pointfeat like the fp32 path (catre_gram.h) instead of a recompute of
    // layer 0 over every point: the statistics are those of W0 pf in fp32 on the bf16-rounded pointfeat - layer 0's own
    // bf16 weight rounding moves them by ~1e-4 relative, a fraction of what rounding a0 to bf16 does next.  The moment
    // buffers borrow y1, which k_rot_l1_bf writes afterwards.
    float* Gc = ws + W.y1;
    float* s1c = Gc + (size_t)2 * B * PF_NG * 4096;
    float* shc = s1c + (size_t)2 * B * PF_NG * 64;
    ProfScope ps(CATRE_K_ROT_L0_STATS, st);
    hipLaunchKernelGGL(k_pf_moments_bf, dim3(2 * B, pf_groups(B)), dim3(256), 0, st, pointfeat, Gc, s1c,
                       shc, B, N, M);
    hipLaunchKernelGGL(k_gn0_from_moments, dim3(B, gn0_shares(B)), dim3(256), 0, st, Gc, s1c, shc, prm[CATRE_P_ROTX_L0_W],
                       prm[CATRE_P_ROTY_L0_W], PMW, 1024, bias0, prm[CATRE_P_ROTX_GN0_W], prm[CATRE_P_ROTX_GN0_B],
                       prm[CATRE_P_ROTY_GN0_W], prm[CATRE_P_ROTY_GN0_B], ws + W.aff0, B, N, M);
  }
  unsigned short* y1 = reinterpret_cast<unsigned short*>(ws + W.y1);
  {
    ProfScope ps(CATRE_K_ROT_L1, st);
    hipLaunchKernelGGL(k_rot_l1_bf, dim3(B * T), dim3(256), 0, st, pointfeat, pkb(packed, L.bf_rot_l0[0]),
                       pkb(packed, L.bf_rot_l0[1]), ws + W.aff0, pkb(packed, L.bf_rot_l1[0]),
                       pkb(packed, L.bf_rot_l1[1]), prm[CATRE_P_ROTX_L1_B], prm[CATRE_P_ROTY_L1_B], y1, ws + W.gn1, B,
                       N, M, g_trunk_trace ? g_trunk_trace + ((size_t)1 << 24) : nullptr);
  }
  hipLaunchKernelGGL(k_gn_finalize, dim3(B * 2), dim3(256), (size_t)T * 64 * sizeof(float), st, ws + W.gn1, ws + W.gn1stat,
                     N, M);
  {
    ProfScope ps(CATRE_K_ROT_OUT, st);
    hipLaunchKernelGGL(k_rot_out_bf, dim3(B * T, 2), dim3(256), 0, st, y1, ws + W.gn1stat, prm[CATRE_P_ROTX_GN1_W],
                       prm[CATRE_P_ROTX_GN1_B], prm[CATRE_P_ROTY_GN1_W], prm[CATRE_P_ROTY_GN1_B],
                       prm[CATRE_P_ROTX_NECK_W], prm[CATRE_P_ROTY_NECK_W], prm[CATRE_P_ROTX_CONVP_W],
                       prm[CATRE_P_ROTY_CONVP_W], ws + W.rpart, B, N, M, rd);
  }
  hipLaunchKernelGGL(k_rot_finish, dim3((B * 6 + 255) / 256), dim3(256), 0, st, ws + W.rpart, prm[CATRE_P_ROTX_NECK_B],
                     prm[CATRE_P_ROTY_NECK_B], packed + L.sumwp, prm[CATRE_P_ROTX_CONVP_B], prm[CATRE_P_ROTY_CONVP_B],
                     ws + W.rot6d, B, T, rd);
  if ((rc = check_launch())) return rc;
  return pose_update_impl(ws + W.rot6d, ws + W.dt, ws + W.ds, init_pose, init_scale, mean_scales, Ks, o, pose_out,
                          scale_out, B, (void*)st, pose_echo, scale_echo);
}

static int refine_iter_impl(const catre_points* pts, const float* init_pose, const float* init_scale,
                            const float* mean_scales, const float* Ks, const float* const* prm, const float* packed,
                            const catre_opts* o, float* pose_out, float* scale_out, void* workspace, size_t ws_bytes, int B,
                            int N, int M, void* stream, float* pose_echo, float* scale_echo) {
  REQUIRE(pts && pts->obs && pts->kps && init_pose && init_scale && prm && packed && o && pose_out && scale_out &&
          workspace && dims_ok(B, N, M));
  const WsLayout W = ws_layout(B, N, M);
  if (ws_bytes < W.total * sizeof(float)) return CATRE_ERR_WORKSPACE;
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  // the two rot heads emit rot_dim values each: only even-width parametrisations can come out of them
  if (o->rot_type != CATRE_ROT_6D && o->rot_type != CATRE_ROT_QUAT) return CATRE_ERR_UNSUPPORTED;
  if (o->compute_dtype == CATRE_DTYPE_BF16)
    return refine_iter_bf(pts, init_pose, init_scale, mean_scales, Ks, prm, packed, o, pose_out, scale_out, ws, W, B, N, M,
                          st, pose_echo, scale_echo);
  if (o->compute_dtype != CATRE_DTYPE_F32 && o->compute_dtype != CATRE_DTYPE_SPLIT) return CATRE_ERR_UNSUPPORTED;
  const bool split = o->compute_dtype == CATRE_DTYPE_SPLIT;
  const int T = (N + TP - 1) / TP + (M + TP - 1) / TP, R = 2 * B;
  const int rd = catre_rot_dim(o->rot_type) / 2;
  // Small batches take the latency path (catre_small.h): the same arithmetic on fewer launches.
  const size_t smem_d = sizeof(float) * (size_t)std::max(T * 64 + 64 + 16, TS_OB * (256 + 4 * 256));
  const bool small = R <= SMALL_ROWS && smem_d <= 64 * 1024;
  // ... and while a consumer of a pooled feature can take the maximum over the tile partials itself: every one of fc1's
  // 16 workgroups stages all R rows x 16 tiles, which beats a k_reduce_pm launch up to R = 4 clouds (B = 4: 15 vs 5 + 5.5 us)
  const bool fold = small && R <= SMALL_FOLD_ROWS;
  const float* pm = fold ? ws + W.pm : nullptr;
  auto reduce_pm = [&](float* out, int ldo, int C, int rpt = 1) {
    hipLaunchKernelGGL(k_reduce_pm, dim3(R, (C + 255) / 256), dim3(256), 0, st, ws + W.pm, out, ldo, C, B, N, M, rpt);
  };
  // STN3d (pointnet.py:98) on both clouds
  // small fp32 batches: each FC tail as ONE launch (k_fc_tail); its barrier counters are zeroed by the encoder kernel in front
  unsigned* bar = small && !split && fc_tail_on() ? reinterpret_cast<unsigned*>(ws + W.bar) : nullptr;
  launch_stn3d(pts, prm, packed, ws, W, B, N, M, split, st, bar);
  if (!fold) reduce_pm(ws + W.pool, 1024, 1024);
  if ((rc = stn_fc_tail(ws + W.pool, prm, CATRE_P_STN_FC1_W, ws + W.h1, ws + W.h2, ws + W.trans3, 3, R, st, pm, B, N, M,
                        bar)))
    return rc;
  const float* t64 = nullptr;
  if (o->feature_transform) {  // STNkd (pointnet.py:105-106)
    launch_stnkd(pts, ws + W.trans3, prm, packed, ws, W, B, N, M, split, st, bar);
    if (!fold) reduce_pm(ws + W.pool, 1024, 1024);
    if ((rc = stn_fc_tail(ws + W.pool, prm, CATRE_P_FSTN_FC1_W, ws + W.h1, ws + W.h2, ws + W.trans64, 64, R, st, pm, B, N,
                          M, bar)))
      return rc;
    t64 = ws + W.trans64;
  }
  // a handful of objects, fp32: the trunk on HALF tiles (32 points per workgroup: half the per-tile prologue)
  const int rsh = small && !split ? trunk_h_split(B * T) : 0;
  if (rsh) {
    const PackLayout L = pack_layout(1);
    ProfScope ps(CATRE_K_TRUNK, st);
#define LAUNCH_(RSH)                                                                                                  \
  hipLaunchKernelGGL(k_trunk_h<RSH>, dim3(B * T * 2 * RSH), dim3(512), 0, st, *pts, (const float*)(ws + W.trans3), t64,  \
                     prm[CATRE_P_CONV1_W], prm[CATRE_P_CONV1_B], pk4(packed, L.c2), prm[CATRE_P_CONV2_B],               \
                     pk4(packed, L.c3), prm[CATRE_P_CONV3_B], pk4(packed, L.c4), prm[CATRE_P_CONV4_B], ws + W.pm,       \
                     ws + W.pointfeat, B, N, M)
    RS_DISPATCH(rsh, LAUNCH_)
#undef LAUNCH_
  } else {
    launch_trunk(pts, ws + W.trans3, t64, prm, packed, ws + W.pointfeat, ws, W, B, N, M, split, st);
  }
  if (!small) {
    reduce_pm(ws + W.gfeat, PMW, PMW);
    if ((rc = check_launch())) return rc;
    if ((rc = catre_ts_head(ws + W.gfeat, init_pose, init_scale, prm, packed, o, ws + W.dt, ws + W.ds, workspace, ws_bytes,
                            B, stream)))
      return rc;
    if ((rc = rot_head_impl(ws + W.gfeat, ws + W.pointfeat, prm, packed, ws + W.rot6d, ws, W, B, N, M, st, split, rd)))
      return rc;
    return pose_update_impl(ws + W.rot6d, ws + W.dt, ws + W.ds, init_pose, init_scale, mean_scales, Ks, o, pose_out,
                            scale_out, B, stream, pose_echo, scale_echo);
  }
  // ---- latency path after the trunk: 5 launches (catre_small.h), 6 when the pooled feature is reduced by its own ----
  if (!fold) reduce_pm(ws + W.gfeat, PMW, PMW, rsh ? 2 : 1);
  {
    const int expect = PMW * (o->with_kps_feature ? 2 : 1) + (o->with_init_scale ? 3 : 0) + (o->with_init_trans ? 3 : 0);
    if (o->ts_in_dim != expect) return CATRE_ERR_BAD_ARG;
    if (o->k_aware && !o->delta_t_space_3d && !Ks) return CATRE_ERR_BAD_ARG;
    if (o->scale_base_mean && !mean_scales) return CATRE_ERR_BAD_ARG;
  }
  const PackLayout L = pack_layout(o->ts_in_dim);
  float* bias0 = ws + W.bias0;
  float* Gc = ws + W.y1;  // the moment buffers borrow y1 (see rot_head_impl)
  float* s1c = Gc + (size_t)2 * B * PF_NG * 4096;
  float* shc = s1c + (size_t)2 * B * PF_NG * 64;
  const int groups = (B + TS_OB - 1) / TS_OB;
  {
    HeadsAArgs A;
    A.pointfeat = ws + W.pointfeat;
    A.Gc = Gc;
    A.s1c = s1c;
    A.shc = shc;
    A.pose = init_pose;
    A.scale = init_scale;
    A.W0T = packed + L.ts_w0t;
    A.tspart = ws + W.tspart;
    A.in_dim = o->ts_in_dim;
    A.with_kps = o->with_kps_feature;
    A.with_scale = o->with_init_scale;
    A.with_trans = o->with_init_trans;
    A.w0x = prm[CATRE_P_ROTX_L0_W];
    A.b0x = prm[CATRE_P_ROTX_L0_W + 1];
    A.w0y = prm[CATRE_P_ROTY_L0_W];
    A.b0y = prm[CATRE_P_ROTY_L0_W + 1];
    A.bias0 = bias0;
    A.pm = fold ? ws + W.pm : nullptr;
    A.gfeat = ws + W.gfeat;
    A.B = B;
    A.N = N;
    A.M = M;
    A.rpt = rsh ? 2 : 1;
    A.n_mom = R * PF_NG;
    A.n_ts = groups * TS_KS;
    ProfScope ps(CATRE_K_ROT_L0_STATS, st);
    hipLaunchKernelGGL(k_heads_a, dim3(A.n_mom + A.n_ts + 2 * 8), dim3(64 * LIN_WAVES), 0, st, A);
    hipLaunchKernelGGL(k_gn0_from_moments, dim3(B, gn0_shares(B)), dim3(256), 0, st, Gc, s1c, shc, prm[CATRE_P_ROTX_L0_W],
                       prm[CATRE_P_ROTY_L0_W], PMW, 1024, bias0, prm[CATRE_P_ROTX_GN0_W], prm[CATRE_P_ROTX_GN0_B],
                       prm[CATRE_P_ROTY_GN0_W], prm[CATRE_P_ROTY_GN0_B], ws + W.aff0, B, N, M);
  }
  {
    ProfScope ps(CATRE_K_ROT_L1, st);
    if (split)
      hipLaunchKernelGGL(k_rot_l1_split<false>, dim3(B * T), dim3(256), 0, st, ws + W.pointfeat, pkb(packed, L.sp_rot_l0[0]),
                         pkb(packed, L.sp_rot_l0[1]), ws + W.aff0, pkb(packed, L.sp_rot_l1[0]), pkb(packed, L.sp_rot_l1[1]),
                         prm[CATRE_P_ROTX_L1_B], prm[CATRE_P_ROTY_L1_B], ws + W.y1, ws + W.gn1, B, N, M,
                         g_trunk_trace ? g_trunk_trace + ((size_t)1 << 24) : nullptr);
    else {
#define LAUNCH_ROT_L1(RS)                                                                                               \
  hipLaunchKernelGGL(k_rot_l1<RS>, dim3(B * T * RS), dim3(256), 0, st, ws + W.pointfeat, pk4(packed, L.rot_l0[0]),        \
                     pk4(packed, L.rot_l0[1]), ws + W.aff0, pk4(packed, L.rot_l1[0]), pk4(packed, L.rot_l1[1]),         \
                     prm[CATRE_P_ROTX_L1_B], prm[CATRE_P_ROTY_L1_B], ws + W.y1, ws + W.gn1, B, N, M,                     \
                     g_trunk_trace ? g_trunk_trace + ((size_t)1 << 24) : nullptr)
      RS_DISPATCH(row_split(B * T), LAUNCH_ROT_L1)
#undef LAUNCH_ROT_L1
    }
  }
  {
    HeadsDArgs D;
    D.y1 = ws + W.y1;
    D.gn1 = ws + W.gn1;
    D.gam1x = prm[CATRE_P_ROTX_GN1_W];
    D.bet1x = prm[CATRE_P_ROTX_GN1_B];
    D.gam1y = prm[CATRE_P_ROTY_GN1_W];
    D.bet1y = prm[CATRE_P_ROTY_GN1_B];
    D.neckx = prm[CATRE_P_ROTX_NECK_W];
    D.necky = prm[CATRE_P_ROTY_NECK_W];
    D.wpx = prm[CATRE_P_ROTX_CONVP_W];
    D.wpy = prm[CATRE_P_ROTY_CONVP_W];
    D.rpart = ws + W.rpart;
    D.B = B;
    D.N = N;
    D.M = M;
    D.rd = rd;
    D.n_rot = B * T * 2;
    D.ts = ts_head_args(ws + W.tspart, prm, packed, L, ws + W.dt, ws + W.ds, B);
    ProfScope ps(CATRE_K_ROT_OUT, st);
    hipLaunchKernelGGL(k_heads_d, dim3(D.n_rot + groups), dim3(1024), smem_d, st, D);
  }
  hipLaunchKernelGGL(k_finish_update, dim3((B + 7) / 8), dim3(64), 0, st, ws + W.rpart, prm[CATRE_P_ROTX_NECK_B],
                     prm[CATRE_P_ROTY_NECK_B], packed + L.sumwp, prm[CATRE_P_ROTX_CONVP_B], prm[CATRE_P_ROTY_CONVP_B], T, rd,
                     ws + W.dt, ws + W.ds, init_pose, init_scale, mean_scales, Ks, *o, pose_out, scale_out, B, pose_echo,
                     scale_echo);
  return check_launch();
}

int catre_refine_iter(const catre_points* pts, const float* init_pose, const float* init_scale,
                      const float* mean_scales, const float* Ks, const float* const* prm, const float* packed,
                      const catre_opts* o, float* pose_out, float* scale_out, void* workspace, size_t ws_bytes, int B,
                      int N, int M, void* stream) {
  return refine_iter_impl(pts, init_pose, init_scale, mean_scales, Ks, prm, packed, o, pose_out, scale_out, workspace,
                          ws_bytes, B, N, M, stream, nullptr, nullptr);
}

// init_pose / init_scale == nullptr: slot 0 of poses / scales holds the initial estimate (catre_refine_k); otherwise the
// first iteration reads the caller's buffers and its pose-update kernel copies them into slot 0 (catre_refine_k_from)
static int refine_k_impl(const float* pcl, const float* kps, const float* init_pose, const float* init_scale,
                         const float* mean_scales, const float* Ks, const float* const* prm, const float* packed,
                         const catre_opts* o, float* poses, float* scales, void* workspace, size_t ws_bytes, int B, int N,
                         int M, int n_iter, void* stream) {
  REQUIRE(pcl && kps && prm && packed && o && poses && scales && workspace && dims_ok(B, N, M) && n_iter >= 0);
  REQUIRE((init_pose == nullptr) == (init_scale == nullptr));
  const WsLayout W = ws_layout(B, N, M);
  if (ws_bytes < W.total * sizeof(float)) return CATRE_ERR_WORKSPACE;
  float* ws = (float*)workspace;
  // Small batches (latency-bound: every launch counts): the raw clouds go straight to the encoder kernels and the
  // pose-apply of batch_updater_test happens as they load a point.  Large batches: x / tfd_kps are materialised once per
  // iteration (5 us) - three kernels x eight waves re-deriving every point's transform costs more than that there.
  // Same device function either way: same bits.
#ifndef CATRE_OTF_POINTS
#define CATRE_OTF_POINTS (64 * 1024)
#endif
  const bool on_the_fly = (size_t)B * (N + M) <= (size_t)CATRE_OTF_POINTS;
  catre_points pts;
  pts.obs = on_the_fly ? pcl : ws + W.xbuf;
  pts.obs_sb = (int64_t)N * 3;
  pts.obs_sn = 3;
  pts.obs_sc = 1;
  pts.kps = on_the_fly ? kps : ws + W.kbuf;
  pts.kps_sb = (int64_t)M * 3;
  pts.kps_sn = 3;
  pts.kps_sc = 1;
  pts.apply_pose = on_the_fly ? 1 : 0;
  pts.zero_center = o->zero_center;
  pts.pose = pts.scale = nullptr;
  if (init_pose && n_iter == 0) {  // nothing to ride on: plain copies
    if (hipMemcpyAsync(poses, init_pose, (size_t)B * 12 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) !=
            hipSuccess ||
        hipMemcpyAsync(scales, init_scale, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) !=
            hipSuccess)
      return CATRE_ERR_LAUNCH;
  }
  for (int i = 1; i <= n_iter; ++i) {
    const bool first = init_pose && i == 1;
    const float* pose_in = first ? init_pose : poses + (size_t)(i - 1) * B * 12;
    // batch_test.py:74-75: the scale estimate is only fed back when REFINE_SCLAE
    const float* scale_in = first ? init_scale : scales + (size_t)(o->refine_scale ? i - 1 : 0) * B * 3;
    if (on_the_fly) {
      pts.pose = pose_in;
      pts.scale = scale_in;
    } else {
      const int rc0 =
          catre_pose_apply(pcl, kps, pose_in, scale_in, ws + W.xbuf, ws + W.kbuf, B, N, M, o->zero_center, stream);
      if (rc0) return rc0;
    }
    const int rc = refine_iter_impl(&pts, pose_in, scale_in, mean_scales, Ks, prm, packed, o, poses + (size_t)i * B * 12,
                                    scales + (size_t)i * B * 3, workspace, ws_bytes, B, N, M, stream,
                                    first ? poses : nullptr, first ? scales : nullptr);
    if (rc) return rc;
  }
  return CATRE_OK;
}

int catre_refine_k(const float* pcl, const float* kps, const float* mean_scales, const float* Ks,
                   const float* const* prm, const float* packed, const catre_opts* o, float* poses, float* scales,
                   void* workspace, size_t ws_bytes, int B, int N, int M, int n_iter, void* stream) {
  return refine_k_impl(pcl, kps, nullptr, nullptr, mean_scales, Ks, prm, packed, o, poses, scales, workspace, ws_bytes, B, N,
                       M, n_iter, stream);
}

int catre_refine_k_from(const float* pcl, const float* kps, const float* init_pose, const float* init_scale,
                        const float* mean_scales, const float* Ks, const float* const* prm, const float* packed,
                        const catre_opts* o, float* poses, float* scales, void* workspace, size_t ws_bytes, int B, int N,
                        int M, int n_iter, void* stream) {
  REQUIRE(init_pose && init_scale);
  return refine_k_impl(pcl, kps, init_pose, init_scale, mean_scales, Ks, prm, packed, o, poses, scales, workspace, ws_bytes,
                       B, N, M, n_iter, stream);
}


int catre_form_switch(int id, int value) {
  if (id < 0 || id > 4) return -1;
  const int bit = 1 << id;
  const int cur = forms();
  if (value >= 0) g_forms.store(value ? (cur | bit) : (cur & ~bit), std::memory_order_relaxed);
  return (cur & bit) ? 1 : 0;
}

int catre_debug_knob(int id, int value) {
#ifdef CATRE_DEBUG_TRACE
  if (id == 0) return hipMemcpyToSymbol(HIP_SYMBOL(g_dephase_cycles), &value, sizeof(int)) == hipSuccess ? CATRE_OK : CATRE_ERR_LAUNCH;
  if (id == 1) return hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &value, sizeof(int)) == hipSuccess ? CATRE_OK : CATRE_ERR_LAUNCH;
  return CATRE_ERR_BAD_ARG;
#else
  (void)id;
  (void)value;
  return CATRE_ERR_UNSUPPORTED;
#endif
}

// Identity of the capture `stream` is recording into (0: not capturing).  HipRuntime keys "this capture already holds a
// weight-pack node" on it: two captures on one stream are two graphs, each needs its own pack node.
int catre_stream_capture_id(void* stream, unsigned long long* id_out) {
  REQUIRE(id_out);
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &status, &id) != hipSuccess) return CATRE_ERR_LAUNCH;
  *id_out = status == hipStreamCaptureStatusActive ? id : 0ull;
  return CATRE_OK;
}

int catre_debug_trunk_trace(void* device_buffer) {
#ifdef CATRE_DEBUG_TRACE
  g_trunk_trace = (unsigned long long*)device_buffer;
  return CATRE_OK;
#else
  (void)device_buffer;
  return CATRE_ERR_UNSUPPORTED;  // product build: no stamps in the kernels (make TRACE=1 builds the instrumented library)
#endif
}

#ifdef CATRE_NO_PROFILING
int catre_profile_enable(int, int) { return CATRE_ERR_UNSUPPORTED; }
int catre_profile_collect(float*, int, int*) { return CATRE_ERR_UNSUPPORTED; }
#else
int catre_profile_enable(int kernel_id, int max_records) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
  delete[] g_prof.ev;
  g_prof = ProfState();
  if (kernel_id < 0 || max_records <= 0) return CATRE_OK;
  if (kernel_id >= CATRE_K_COUNT) return CATRE_ERR_BAD_ARG;
  g_prof.ev = new hipEvent_t[2 * max_records];
  for (int i = 0; i < 2 * max_records; ++i)
    if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return CATRE_ERR_LAUNCH;
  g_prof.kernel = kernel_id;
  g_prof.cap = max_records;
  return CATRE_OK;
}

int catre_profile_collect(float* ms_out, int max_out, int* n_out) {
  REQUIRE(n_out && (ms_out || max_out == 0));
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = g_prof.n < max_out ? g_prof.n : max_out;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return CATRE_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms_out[i], g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return CATRE_ERR_LAUNCH;
  }
  *n_out = n;
  g_prof.n = 0;
  return CATRE_OK;
}
#endif

int catre_colmax(const float* x, float* out, int B, int C, int N, void* stream) {
  REQUIRE(x && out && B > 0 && C > 0 && N > 0);
  const int rows = B * C;
  const int grid = rows / 4 < 8192 ? (rows + 3) / 4 : 8192;
  {
    ProfScope ps(CATRE_K_COLMAX, (hipStream_t)stream);
    hipLaunchKernelGGL(k_colmax, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, rows, N);
  }
  return check_launch();
}

#include "catre_train_api.inc"

// ---- row f2: train-time batch glue --------------------------------------------------------------------------
int catre_aug_points(const float* pcl, const float* pose, const float* scale, const int32_t* sym_flags,
                     const float* bbox_ratios, const float* delta_r, const float* delta_t, float* pcl_out,
                     float* pose_out, float* scale_out, int B, int N, void* stream) {
  REQUIRE(pcl && pose && scale && pcl_out && pose_out && scale_out && B > 0 && N > 0);
  REQUIRE((delta_r == nullptr) == (delta_t == nullptr));
  AugParams prm;
  prm.do_bbox = bbox_ratios != nullptr;
  prm.do_rt = delta_r != nullptr;
  for (int i = 0; i < 3; ++i) prm.ratios[i] = prm.do_bbox ? bbox_ratios[i] : 1.f;
  for (int i = 0; i < 9; ++i) prm.delta_r[i] = prm.do_rt ? delta_r[i] : (i % 4 == 0 ? 1.f : 0.f);
  for (int i = 0; i < 3; ++i) prm.delta_t[i] = prm.do_rt ? delta_t[i] : 0.f;
  const int total = B * N;
  hipLaunchKernelGGL(k_aug_points, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, pcl, pose, scale,
                     sym_flags, prm, pcl_out, pose_out, scale_out, B, N);
  return check_launch();
}

// ---- row f3: point-cloud preparation -----------------------------------------------------------------------
namespace {
struct PclWs {
  size_t bins, choose, offsets, total, cand, total_ints;
  int nchunks;
};
PclWs pcl_ws(int I, int H, int W) {
  PclWs L;
  L.nchunks = (H * W + PCL_CHUNK - 1) / PCL_CHUNK;
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t r = o;
    o += align_up(n, 64);
    return r;
  };
  L.bins = take((size_t)I * L.nchunks * 12);
  L.choose = take(I);
  L.offsets = take((size_t)I * L.nchunks);
  L.total = take(I);
  L.cand = take((size_t)I * H * W);
  L.total_ints = o;
  return L;
}
inline PclCam pcl_cam(const float* K9) { return PclCam{K9[0], K9[4], K9[2], K9[5]}; }
}  // namespace

size_t catre_pcl_workspace_bytes(int I, int H, int W) {
  if (I <= 0 || H <= 0 || W <= 0) return 0;
  return pcl_ws(I, H, W).total_ints * sizeof(int);
}

int catre_pcl_candidates(const float* depth, const float* K9, const unsigned char* masks, const float* poses,
                         const float* scales, float ratio, int use_ball, int I, int H, int W, void* workspace,
                         size_t ws_bytes, int32_t* counts_out, void* stream) {
  REQUIRE(depth && K9 && poses && scales && workspace && I > 0 && H > 0 && W > 0 && (size_t)H * W < (1u << 30));
  const PclWs L = pcl_ws(I, H, W);
  if (ws_bytes < L.total_ints * sizeof(int)) return CATRE_ERR_WORKSPACE;
  int* ws = (int*)workspace;
  hipStream_t st = (hipStream_t)stream;
  const PclCam cam = pcl_cam(K9);
  hipLaunchKernelGGL(k_pcl_count, dim3(L.nchunks, I), dim3(256), 0, st, depth, masks, poses, scales, cam, ratio, use_ball,
                     H, W, L.nchunks, ws + L.bins);
  hipLaunchKernelGGL(k_pcl_pick, dim3(I), dim3(256), 0, st, ws + L.bins, L.nchunks, 1, ws + L.choose, ws + L.offsets,
                     ws + L.total);
  hipLaunchKernelGGL(k_pcl_compact, dim3(L.nchunks, I), dim3(256), 0, st, depth, masks, poses, scales, cam, ratio,
                     use_ball, H, W, L.nchunks, ws + L.choose, ws + L.offsets, ws + L.cand);
  if (counts_out &&
      hipMemcpyAsync(counts_out, ws + L.total, (size_t)I * sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess)
    return CATRE_ERR_LAUNCH;
  return check_launch();
}

int catre_pcl_sample(const float* depth, const float* K9, const void* workspace, size_t ws_bytes,
                     const long long* sample_idx, unsigned long long seed, int I, int H, int W, int N, float* pcl_out,
                     int32_t* pix_out, void* stream) {
  REQUIRE(depth && K9 && workspace && pcl_out && I > 0 && H > 0 && W > 0 && N > 0);
  const PclWs L = pcl_ws(I, H, W);
  if (ws_bytes < L.total_ints * sizeof(int)) return CATRE_ERR_WORKSPACE;
  const int* ws = (const int*)workspace;
  hipLaunchKernelGGL(k_pcl_gather, dim3((N + 255) / 256, I), dim3(256), 0, (hipStream_t)stream, depth, pcl_cam(K9), H, W,
                     ws + L.cand, ws + L.total, sample_idx, seed, N, pcl_out, pix_out);
  return check_launch();
}

// INPUT.FPS_SAMPLE: sample_idx_out [I][N] = farthest-point order of each instance's tiled candidate list (feed it to
// catre_pcl_sample).  scratch: I * 4 * slot_cap floats, slot_cap >= the largest tiled list (count doubled until >= N).
int catre_pcl_fps(const float* depth, const float* K9, const void* workspace, size_t ws_bytes, int I, int H, int W, int N,
                  float* scratch, int slot_cap, long long* sample_idx_out, void* stream) {
  REQUIRE(depth && K9 && workspace && scratch && sample_idx_out && I > 0 && H > 0 && W > 0 && N > 0 && slot_cap > 0);
  const PclWs L = pcl_ws(I, H, W);
  if (ws_bytes < L.total_ints * sizeof(int)) return CATRE_ERR_WORKSPACE;
  const int* ws = (const int*)workspace;
  hipLaunchKernelGGL(k_pcl_fps, dim3(I), dim3(1024), 0, (hipStream_t)stream, depth, pcl_cam(K9), H, W, ws + L.cand,
                     ws + L.total, N, scratch, slot_cap, sample_idx_out);
  return check_launch();
}

// ---- row f1: training loss ---------------------------------------------------------------------------------
int catre_loss_fwd(const float* pose, const float* scale, const float* gt_rot, const float* gt_trans,
                   const float* gt_scale, const float* kps, const float* cands, const unsigned char* valid,
                   const int32_t* is_sym, const catre_loss_cfg* cfg, int32_t* best, int32_t* counts, float* part_ws,
                   float* losses, const float* trans_deltas, int B, int M, int S1, void* stream) {
  return catre_loss_fwd_sums(pose, scale, gt_rot, gt_trans, gt_scale, kps, cands, valid, is_sym, cfg, best, counts, part_ws,
                             losses, trans_deltas, nullptr, 0, nullptr, B, M, S1, stream);
}

// ... plus prefix[k] = ((0 + losses[terms[0]]) + losses[terms[1]]) + ... + losses[terms[k]], k < n_terms <= 6 (terms: host
// array of loss indices in the order the caller's loss dict holds them): every intermediate of `sum(loss_dict.values())`
int catre_loss_fwd_sums(const float* pose, const float* scale, const float* gt_rot, const float* gt_trans,
                        const float* gt_scale, const float* kps, const float* cands, const unsigned char* valid,
                        const int32_t* is_sym, const catre_loss_cfg* cfg, int32_t* best, int32_t* counts, float* part_ws,
                        float* losses, const float* trans_deltas, const int32_t* terms, int n_terms, float* prefix, int B,
                        int M, int S1, void* stream) {
  REQUIRE(pose && scale && gt_rot && gt_trans && gt_scale && cfg && is_sym && best && counts && part_ws && losses && B > 0 &&
          S1 > 0);
  REQUIRE(!cfg->pm_on || (kps && cands && valid && M > 0));
  REQUIRE(n_terms >= 0 && n_terms <= 6 && (n_terms == 0 || (terms && prefix)));
  unsigned order = 0;
  for (int k = 0; k < n_terms; ++k) {
    REQUIRE(terms[k] >= 0 && terms[k] < 6);
    order |= (unsigned)terms[k] << (4 * k);
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_loss_fwd, dim3(B), dim3(256), 0, st, pose, scale, gt_rot, gt_trans, gt_scale, kps, cands, valid,
                     is_sym, *cfg, best, part_ws, B, M, S1);
  hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(512), 0, st, (const float*)part_ws, is_sym, *cfg, losses, counts, B, M,
                     pose, gt_trans, trans_deltas, order, n_terms, prefix);
  return check_launch();
}

int catre_loss_bwd(const float* pose, const float* scale, const float* gt_rot, const float* gt_trans,
                   const float* gt_scale, const float* kps, const float* cands, const int32_t* is_sym,
                   const int32_t* best, const int32_t* counts, const float* upstream, const catre_loss_cfg* cfg,
                   float* dpose, float* dscale, int B, int M, int S1, void* stream) {
  REQUIRE(upstream);
  return catre_loss_bwd_sums(pose, scale, gt_rot, gt_trans, gt_scale, kps, cands, is_sym, best, counts, upstream, nullptr,
                             nullptr, 0, cfg, dpose, dscale, B, M, S1, stream);
}

// ... with the upstream gradients of the prefix sums of catre_loss_fwd_sums added to those of the six losses (upstream [6]):
// up_prefix = HOST array of n_terms device pointers to one float each (NULL entries = zero); either argument may be NULL
int catre_loss_bwd_sums(const float* pose, const float* scale, const float* gt_rot, const float* gt_trans,
                        const float* gt_scale, const float* kps, const float* cands, const int32_t* is_sym,
                        const int32_t* best, const int32_t* counts, const float* upstream,
                        const float* const* up_prefix, const int32_t* terms, int n_terms, const catre_loss_cfg* cfg,
                        float* dpose, float* dscale, int B, int M, int S1, void* stream) {
  REQUIRE(pose && scale && gt_rot && gt_trans && gt_scale && cfg && is_sym && best && counts && (upstream || up_prefix) &&
          dpose && dscale && B > 0 && S1 > 0);
  REQUIRE(!cfg->pm_on || (kps && cands && M > 0));
  REQUIRE(n_terms >= 0 && n_terms <= 6 && (!up_prefix || (terms && n_terms > 0)));
  unsigned order = 0;
  LossUpPrefix upp;
  for (int k = 0; k < n_terms; ++k) {
    REQUIRE(terms[k] >= 0 && terms[k] < 6);
    order |= (unsigned)terms[k] << (4 * k);
    if (up_prefix) upp.p[k] = up_prefix[k];
  }
  hipLaunchKernelGGL(k_loss_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, pose, scale, gt_rot, gt_trans, gt_scale, kps,
                     cands, is_sym, best, upstream, *cfg, counts, dpose, dscale, B, M, S1, upp, order, up_prefix ? n_terms : 0);
  return check_launch();
}

int catre_init_noise(const float* pose, const float* euler_deg, const float* trans_noise, float max_rot_deg,
                     float min_z, float* pose_out, const float* scale, const float* scale_noise, float min_s,
                     float max_s, float* scale_out, int B, void* stream) {
  REQUIRE(B > 0 && (pose || scale));
  if (pose) REQUIRE(euler_deg && trans_noise && pose_out);
  if (scale) REQUIRE(scale_noise && scale_out);
  hipLaunchKernelGGL(k_init_noise, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose, euler_deg, trans_noise,
                     max_rot_deg, max_rot_deg >= 0.f ? 1 : 0, min_z, pose_out, scale, scale_noise, min_s, max_s, scale_out,
                     B);
  return check_launch();
}

}  // extern "C"
