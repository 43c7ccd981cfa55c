// catre_split.h - the "split" compute mode: fp32-accurate GEMMs on the bf16 matrix pipe for the three layers that
// hold 98 % of the path's FLOPs (all MFMA layers of the two STNs, trunk conv3 128->512 and conv4 512->1024, rot-head
// layers 0 64->256 and 1 256->256).
// Included by catre_kernels.hip after catre_bf16.h.
//
// Every fp32 operand x is written as x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 bits of mantissa.  A product
// is accumulated in fp32 as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  - three v_mfma_f32_32x32x16_bf16 (96 cycles per K=16)
// instead of eight v_mfma_f32_32x32x2_f32 (512 cycles); the dropped a_lo*b_lo term is 2^-16 relative.  The layers it
// is applied to feed a max-pool or a GroupNorm, and the refined pose moves by <= 1e-5 (measured: parity tests hold the
// same 2e-5 bound as the pure fp32 path; the reference contract is 1e-4).  Everything else - the thin layers, bias,
// ReLU, pools, GroupNorm statistics, GELU, FC tails, heads, SO(3) update - is the fp32 code of catre_kernels.hip.
//
// The hi / lo halves of an activation image take exactly the LDS bytes of the fp32 image they replace, so the
// workgroup shapes of the fp32 kernels are kept.  Images and weight fragments use the bf16 chunk / k-slot layout of
// catre_bf16.h; the lo pack follows the hi pack in the packed-weight buffer.
#pragma once

__device__ __forceinline__ void split_bf8(const float (&v)[8], u32x4& hi, u32x4& lo) {
  float r[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
    r[2 * i] = v[2 * i] - bf_lo(hi[i]);
    r[2 * i + 1] = v[2 * i + 1] - bf_hi(hi[i]);
  }
  lo = pack_bf8(r);
}

// weights -> hi and lo bf16 fragments in k-slot order: dst[0 .. rows*K) = hi, dst[rows*K .. 2 rows*K) = lo
__global__ void k_pack_frag_split(const float* __restrict__ src, int ld, int coloff, int rows, int K,
                                  unsigned short* __restrict__ dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * K) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, rest = idx >> 9;
  const int nkc = K / 16;
  const int kc = rest % nkc, mb = rest / nkc;
  const int row = mb * 32 + (lane & 31), col = kc * 16 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
  const float w = src[(size_t)row * ld + coloff + col];
  const __bf16 h = (__bf16)w;
  dst[idx] = __builtin_bit_cast(unsigned short, h);
  dst[(size_t)rows * K + idx] = __builtin_bit_cast(unsigned short, (__bf16)(w - (float)h));
}

// K-sweep with split operands.  wp: hi fragments of the wave's first m-block, the lo fragments sit `lo_off` u32x4
// further; xh / xl: hi and lo images (row of point 0 of the wave tile).  Per K=16 step and (mb, nb): 3 MFMAs.
template <int MB, int NB, bool SWAP, int CP, int PFD>
struct GemmPipeS {
  static constexpr int NKC = CP / 2;
  static_assert(PFD >= 1 && PFD <= NKC, "prefetch depth");
  static constexpr int RA = PFD + 1;
  u32x4 ah[RA][MB], al[RA][MB], bh[2][NB], bl[2][NB];
  const u32x4* wp;
  int wp_mb, lo_off;

  __device__ __forceinline__ void issue_a(int kc) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      ah[kc % RA][mb] = wp[mb * wp_mb + kc * 64];
      al[kc % RA][mb] = wp[lo_off + mb * wp_mb + kc * 64];
    }
  }
  __device__ __forceinline__ void prefetch(const u32x4* __restrict__ wp_, int wp_mb_, int lo_off_) {
    wp = wp_;
    wp_mb = wp_mb_;
    lo_off = lo_off_;
#pragma unroll
    for (int d = 0; d < PFD; ++d) issue_a(d);
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void run(f32x16 (&acc)[MB][NB], const u32x4* xh, const u32x4* xl, int lane) {
    const int n = lane & 31, h = lane >> 5, key = bf_key<CP>(n);
    const u32x4* rh = xh + n * CP;
    const u32x4* rl = xl + n * CP;
    auto issue_b = [&](int kc) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        bh[kc & 1][nb] = rh[nb * 32 * CP + ((2 * kc + h) ^ key)];
        bl[kc & 1][nb] = rl[nb * 32 * CP + ((2 * kc + h) ^ key)];
      }
    };
    issue_b(0);
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      if (kc + PFD < NKC) issue_a(kc + PFD);
      if (kc + 1 < NKC) issue_b(kc + 1);
      __builtin_amdgcn_sched_barrier(0);
      const int ca = kc % RA, cb = kc & 1;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if (SWAP) {
            acc[mb][nb] = mfma_bf(bh[cb][nb], al[ca][mb], acc[mb][nb]);
            acc[mb][nb] = mfma_bf(bl[cb][nb], ah[ca][mb], acc[mb][nb]);
            acc[mb][nb] = mfma_bf(bh[cb][nb], ah[ca][mb], acc[mb][nb]);
          } else {
            acc[mb][nb] = mfma_bf(al[ca][mb], bh[cb][nb], acc[mb][nb]);
            acc[mb][nb] = mfma_bf(ah[ca][mb], bl[cb][nb], acc[mb][nb]);
            acc[mb][nb] = mfma_bf(ah[ca][mb], bh[cb][nb], acc[mb][nb]);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// "normal"-orientation epilogue of an fp32 layer into split images: act(acc + bias) -> hi chunk + lo chunk
template <int MB, int NB, bool RELU, int CP>
__device__ __forceinline__ void store_tile_split(const f32x16 (&acc)[MB][NB], u32x4* ih, u32x4* il, int mblk0,
                                                 const f32x4 (&bv)[MB][4], int lane) {
  const int n = lane & 31, h = lane >> 5, key = bf_key<CP>(n);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int chunk = 4 * (mblk0 + mb) + 2 * s + h;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float v[8];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float t = acc[mb][nb][4 * (2 * s + g2) + q] + bv[mb][2 * s + g2][q];
            v[4 * g2 + q] = RELU ? fmaxf(t, 0.f) : t;
          }
        u32x4 hi, lo;
        split_bf8(v, hi, lo);
        const int off = (nb * 32 + n) * CP + (chunk ^ key);
        ih[off] = hi;
        il[off] = lo;
      }
    }
}

// Training forward (SAVE instances): hi + lo images [ROWS][C] -> fp32 rows dst[row][C] in channel order (the value the
// three-product arithmetic works with: hi + lo, 16-17 bits of mantissa), rows >= valid skipped.
template <int C, int NT, int ROWS>
__device__ __forceinline__ void save_tile_rows_split(const u32x4* __restrict__ ih, const u32x4* __restrict__ il,
                                                     float* __restrict__ dst, int valid, int tid) {
  constexpr int CP = C / 8;
#pragma unroll
  for (int u = 0; u < ROWS * CP / NT; ++u) {
    const int i = tid + NT * u, row = i / CP, c = i % CP;
    if (row >= valid) continue;
    const u32x4 a = ih[bf_off<CP>(row, c)], b = il[bf_off<CP>(row, c)];
    const f32x4 lo = {bf_lo(a[0]) + bf_lo(b[0]), bf_hi(a[0]) + bf_hi(b[0]), bf_lo(a[1]) + bf_lo(b[1]), bf_hi(a[1]) + bf_hi(b[1])};
    const f32x4 hi = {bf_lo(a[2]) + bf_lo(b[2]), bf_hi(a[2]) + bf_hi(b[2]), bf_lo(a[3]) + bf_lo(b[3]), bf_hi(a[3]) + bf_hi(b[3])};
    float* d = dst + (size_t)row * C + 16 * (c >> 1) + 4 * (c & 1);
    st_stream(reinterpret_cast<f32x4*>(d), lo);
    st_stream(reinterpret_cast<f32x4*>(d + 8), hi);
  }
}

// ------------------------------------------------------------------------------------------
// trunk: k_trunk with conv3 and conv4 on GemmPipeS (conv2 / conv3 epilogues write split images).  512 threads, 160 KiB.
//   a3 hi [64][512 ch] 64 KiB | a3 lo 64 KiB | a2 hi [64][128 ch] 16 KiB | a2 lo 16 KiB
// ------------------------------------------------------------------------------------------
// SAVE (training forward in split mode, catre_train_trunk_fwd): x1 (sv.s1), h1 (sv.s2), conv2 / conv3 outputs (sv.s3, sv.s4)
// as fp32 rows and the per-tile (max, arg-max row) pairs; pointfeat is the fp32 row output it always was.
template <int RS, bool SAVE = false>
__global__ __launch_bounds__(512) void k_trunk_split(catre_points P, const float* __restrict__ trans3,
                                                     const float* __restrict__ trans64, const float* __restrict__ Wc1,
                                                     const float* __restrict__ bc1, const f32x4* __restrict__ wp2,
                                                     const float* __restrict__ b2, const u32x4* __restrict__ wp3,
                                                     const float* __restrict__ b3, const u32x4* __restrict__ wp4,
                                                     const float* __restrict__ b4, float* __restrict__ pm,
                                                     float* __restrict__ pointfeat, int B, int N, int M,
                                                     unsigned long long* __restrict__ trace, TrainSave sv = TrainSave{}) {
  __shared__ __attribute__((aligned(16))) float smem[TRUNK_SMEM];
  float* h1 = smem;                      // [64][68]
  float* t64 = smem + TP * LD64;         // [64][64]
  float* pf = smem + TP * LD64 + 4096;   // [64][68]
  u32x4* a3h = reinterpret_cast<u32x4*>(smem);                // [64][64 chunks]
  u32x4* a3l = reinterpret_cast<u32x4*>(smem + TP * 256);     // + 64 KiB
  u32x4* a2h = reinterpret_cast<u32x4*>(smem + TP * 512);     // [64][16 chunks], 16 KiB
  u32x4* a2l = a2h + TP * 16;                                 // + 16 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x / RS, part = blockIdx.x % RS;
  const TileInfo ti = tile_info(tile, B, N, M);
  const bool ft = trans64 != nullptr;
#define TRUNKS_STAMP(i)                                                                                    \
  do {                                                                                                     \
    if (CATRE_TRACE_ON && trace && lane == 0) trace[((size_t)tile * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter();     \
  } while (0)
  TRUNKS_STAMP(0);

  const int mblk2 = wave >> 1, nb2 = wave & 1;
  GemmPipe<1, 1, false, false, 8, 4> g2;
  g2.prefetch(wp2 + (mblk2 * 8) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, mblk2 * 32, lane);
  {
    float x, y, z;
    load_point(P, ti, lane, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    if (SAVE && wave == 0 && lane < ti.valid) {
      const size_t r = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0 + lane;
      const f32x4 lo = {x, y, z, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(sv.s1 + r * 8) = lo;
      *reinterpret_cast<f32x4*>(sv.s1 + r * 8 + 4) = hi;
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
  TRUNKS_STAMP(1);
  const size_t srow0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE) save_tile_rows<64, 512, false>(h1, LD64, sv.s2 + srow0 * 64, tid);
  if (ft) {
    if (wave < 4) {
      const int mblk = wave >> 1, nb = wave & 1;
      const int n = lane & 31, h = lane >> 5;
      f32x16 acc = zero16();
      const float* xr = h1 + (nb * 32 + n) * LD64 + 4 * h;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        const f32x4 bx = *reinterpret_cast<const f32x4*>(xr + kc * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma32(t64[(kc * 8 + 4 * h + s) * 64 + mblk * 32 + n], bx[s], acc);
      }
      f32x16 accs[1][1] = {{acc}};
      store_tile_lds<1, 1, false>(accs, pf + nb * 32 * LD64, LD64, mblk * 32, nullptr, lane);
    }
    __syncthreads();
  } else {
    pf = h1;
  }
  TRUNKS_STAMP(2);
  // conv3 128->512 on the split pipe too (it would otherwise be a third of the kernel's matrix time)
  GemmPipeS<2, 2, false, 16, 2> g3;
  g3.prefetch(wp3 + ((wave * 2) * 8) * 64 + lane, 8 * 64, 512 * 128 / 8);
  f32x4 bv3[2][4];
  load_bias_quads<2>(bv3, b3, wave * 64, lane);
  __builtin_amdgcn_sched_barrier(0);
  const int pf_row = tid >> 3, pf_c4 = tid & 7;
  f32x4 pf_out0 = {0.f, 0.f, 0.f, 0.f}, pf_out1 = {0.f, 0.f, 0.f, 0.f};
  float pf_max = 0.f;
  {
    if (pf_row < ti.valid) {
      const f32x4* s = reinterpret_cast<const f32x4*>(pf + pf_row * LD64);
      pf_out0 = s[pf_c4];
      pf_out1 = s[pf_c4 + 8];
    }
    {
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
    store_tile_split<1, 1, true, 16>(acc, a2h + nb2 * 32 * 16, a2l + nb2 * 32 * 16, mblk2, bv2, lane);
  }
  __syncthreads();
  TRUNKS_STAMP(3);
  if (SAVE) save_tile_rows_split<128, 512, TP>(a2h, a2l, sv.s3 + srow0 * 128, ti.valid, tid);
  // conv4 512->1024 on the split pipe: wave owns m-blocks [4*wave, +4) in two passes of 2; K = 512 = 32 steps of 16.
  // RS workgroups per tile: 4/RS m-blocks from mb0 (one pass of 2, or of 1)
  constexpr int MB4 = RS == 4 ? 1 : 2;
  const int mb0 = part * (32 / RS) + wave * (4 / RS);
  GemmPipeS<MB4, 2, true, 64, SAVE ? 2 : 3> g4a, g4b;  // SAVE: the arg-max epilogue needs the registers of a ring slot
  {
    f32x16 acc3[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc3[mb][0] = acc3[mb][1] = zero16();
    g3.run(acc3, a2h, a2l, lane);
    // conv4's first weight fragments are requested once conv3's own ring is dead (register budget), still ahead of
    // the epilogue and the barrier
    g4a.prefetch(wp4 + ((size_t)mb0 * 32) * 64 + lane, 32 * 64, 1024 * 512 / 8);
    store_tile_split<2, 2, true, 64>(acc3, a3h, a3l, wave * 2, bv3, lane);
    TRUNKS_STAMP(4);
  }
  float bl4[2][MB4];
  load_bias_lane<MB4>(bl4[0], b4, mb0 * 32, lane);
  if (RS == 1) load_bias_lane<MB4>(bl4[1], b4, (mb0 + 2) * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  TRUNKS_STAMP(5);
  if (SAVE) save_tile_rows_split<512, 512, TP>(a3h, a3l, sv.s4 + srow0 * 512, ti.valid, tid);
  {
    float* dstbase = pointfeat + (ti.is_obs ? ((size_t)ti.obj * N + ti.p0) * 64
                                            : ((size_t)B * N + (size_t)ti.obj * M + ti.p0) * 64);
    if (part == 0 && pf_row < ti.valid) {
      f32x4* d = reinterpret_cast<f32x4*>(dstbase + pf_row * 64);
      d[pf_c4] = pf_out0;
      d[pf_c4 + 8] = pf_out1;
    }
    if (part == 0 && tid < 64) pm[(size_t)tile * PMW + 1024 + tid] = pf_max;
  }
  float* out = pm + (size_t)tile * PMW;
  {
    f32x16 acc4[MB4][2];
#pragma unroll
    for (int mb = 0; mb < MB4; ++mb) acc4[mb][0] = acc4[mb][1] = zero16();
    g4a.run(acc4, a3h, a3l, lane);
    if (RS == 1 && !SAVE) g4b.prefetch(wp4 + ((size_t)(mb0 + 2) * 32) * 64 + lane, 32 * 64, 1024 * 512 / 8);
    if (SAVE) {
      argmax_tile_store<MB4, 2>(acc4, sv.pmax + (size_t)tile * 1024, sv.pidx + (size_t)tile * 1024, mb0 * 32, bl4[0],
                                (int)srow0, lane);
      __builtin_amdgcn_sched_barrier(0);
      if (RS == 1) g4b.prefetch(wp4 + ((size_t)(mb0 + 2) * 32) * 64 + lane, 32 * 64, 1024 * 512 / 8);
    } else {
      max_tile_store_pre<MB4, 2>(acc4, out, mb0 * 32, bl4[0], false, lane);
    }
    TRUNKS_STAMP(6);
  }
  if (RS == 1) {
    f32x16 acc4[MB4][2];
#pragma unroll
    for (int mb = 0; mb < MB4; ++mb) acc4[mb][0] = acc4[mb][1] = zero16();
    g4b.run(acc4, a3h, a3l, lane);
    if (SAVE) {
      int lane2 = lane;  // opaque: keeps the epilogue's store addresses from being computed (and spilled) before the sweep
      asm volatile("" : "+v"(lane2));
      argmax_tile_store<MB4, 2>(acc4, sv.pmax + (size_t)tile * 1024, sv.pidx + (size_t)tile * 1024, (mb0 + 2) * 32, bl4[1],
                                (int)srow0, lane2);
    } else
      max_tile_store_pre<MB4, 2>(acc4, out, (mb0 + 2) * 32, bl4[1], false, lane);
  }
  TRUNKS_STAMP(7);
#undef TRUNKS_STAMP
}

// ------------------------------------------------------------------------------------------
// STN3d / STNkd: every MFMA layer on GemmPipeS (conv1 3->64 stays on the VALU and writes split images); conv3
// (128 -> 1024, 92 % of the kernels' FLOPs) in four passes of 2 m-blocks per wave.  Structure of k_stn3d / k_stnkd.
// ------------------------------------------------------------------------------------------
// RS workgroups per tile (small grids, see k_trunk): the wave owns 8/RS m-blocks from mb0 = 4/RS passes.
template <int RS, typename Img, bool SAVE = false>
__device__ __forceinline__ void stn_conv3_split(const u32x4* __restrict__ wp3, const float* __restrict__ b3, Img a2h,
                                                Img a2l, float* __restrict__ out, int part, int wave, int lane,
                                                float* __restrict__ pmax = nullptr, int* __restrict__ pidx = nullptr,
                                                int row0 = 0) {
  constexpr int NPS = 4 / RS;
  const int mb0 = part * (32 / RS) + wave * (8 / RS);
  GemmPipeS<2, 2, true, 16, 2> g[2];
  g[0].prefetch(wp3 + ((size_t)mb0 * 8) * 64 + lane, 8 * 64, 1024 * 128 / 8);
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    float bl[2];
    load_bias_lane<2>(bl, b3, (mb0 + ps * 2) * 32, lane);
    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g[ps & 1].run(acc, a2h, a2l, lane);
    if (ps < NPS - 1) g[(ps + 1) & 1].prefetch(wp3 + ((size_t)(mb0 + (ps + 1) * 2) * 8) * 64 + lane, 8 * 64, 1024 * 128 / 8);
    if (SAVE)
      argmax_tile_store<2, 2>(acc, pmax, pidx, (mb0 + ps * 2) * 32, bl, row0, lane);
    else
      max_tile_store_pre<2, 2>(acc, out, (mb0 + ps * 2) * 32, bl, true, lane);
  }
}

// conv 3 -> 16 channels [16*grp, +16) of one point on the VALU (fp32), ReLU -> the two split chunks of k-group grp
__device__ __forceinline__ void conv3_relu_chunks_split(float x, float y, float z, const float* __restrict__ W,
                                                        const float* __restrict__ b, int grp, u32x4* rh, u32x4* rl, int key) {
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ch = grp * 16 + r;
    float t = b[ch];
    t = fmaf(W[ch * 3 + 0], x, t);
    t = fmaf(W[ch * 3 + 1], y, t);
    t = fmaf(W[ch * 3 + 2], z, t);
    v[r] = fmaxf(t, 0.f);
  }
  const float c0[8] = {v[0], v[1], v[2], v[3], v[8], v[9], v[10], v[11]};
  const float c1[8] = {v[4], v[5], v[6], v[7], v[12], v[13], v[14], v[15]};
  u32x4 hi, lo;
  split_bf8(c0, hi, lo);
  rh[(2 * grp) ^ key] = hi;
  rl[(2 * grp) ^ key] = lo;
  split_bf8(c1, hi, lo);
  rh[(2 * grp + 1) ^ key] = hi;
  rl[(2 * grp + 1) ^ key] = lo;
}

template <int RS, bool SAVE = false>
__global__ __launch_bounds__(256, 2) void k_stn3d_split(catre_points P, const float* __restrict__ W1,
                                                        const float* __restrict__ b1, const u32x4* __restrict__ wp2,
                                                        const float* __restrict__ b2, const u32x4* __restrict__ wp3,
                                                        const float* __restrict__ b3, float* __restrict__ pm, int B,
                                                        int N, int M, TrainSave sv = TrainSave{}) {
  __shared__ u32x4 smem[2 * TP * 8 + 2 * TP * 16];  // a1 hi/lo [64][64 ch] 16 KiB + a2 hi/lo [64][128 ch] 32 KiB
  u32x4* a1h = smem;
  u32x4* a1l = smem + TP * 8;
  u32x4* a2h = smem + 2 * TP * 8;
  u32x4* a2l = a2h + TP * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x / RS, part = blockIdx.x % RS;
  const TileInfo ti = tile_info(tile, B, N, M);
  GemmPipeS<1, 2, false, 8, 2> g2;  // conv2 64->128: wave -> m-block `wave`
  g2.prefetch(wp2 + (wave * 4) * 64 + lane, 0, 128 * 64 / 8);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, wave * 32, lane);
  {
    float x, y, z;
    load_point(P, ti, lane, x, y, z);
    conv3_relu_chunks_split(x, y, z, W1, b1, wave, a1h + lane * 8, a1l + lane * 8, bf_key<8>(lane));
  }
  __syncthreads();
  const size_t row0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE) save_tile_rows_split<64, 256, TP>(a1h, a1l, sv.s1 + row0 * 64, ti.valid, tid);
  {
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g2.run(acc, a1h, a1l, lane);
    store_tile_split<1, 2, true, 16>(acc, a2h, a2l, wave, bv2, lane);
  }
  __syncthreads();
  if (SAVE) save_tile_rows_split<128, 256, TP>(a2h, a2l, sv.s2 + row0 * 128, ti.valid, tid);
  stn_conv3_split<RS, u32x4*, SAVE>(wp3, b3, a2h, a2l, pm + (size_t)tile * PMW, part, wave, lane,
                                    SAVE ? sv.pmax + (size_t)tile * 1024 : nullptr, SAVE ? sv.pidx + (size_t)tile * 1024 : nullptr,
                                    (int)row0);
}

template <int RS, bool SAVE = false>
__global__ __launch_bounds__(256, 2) void k_stnkd_split(catre_points P, const float* __restrict__ trans3,
                                                        const float* __restrict__ Wc1, const float* __restrict__ bc1,
                                                        const u32x4* __restrict__ wpf1, const float* __restrict__ bf1,
                                                        const u32x4* __restrict__ wpf2, const float* __restrict__ bf2,
                                                        const u32x4* __restrict__ wpf3, const float* __restrict__ bf3,
                                                        float* __restrict__ pm, int B, int N, int M,
                                                        TrainSave sv = TrainSave{}) {
  __shared__ u32x4 smem[4 * TP * 8 + 2 * TP * 16];  // h1, f1 hi/lo 32 KiB + f2 hi/lo 32 KiB
  u32x4* h1h = smem;
  u32x4* h1l = smem + TP * 8;
  u32x4* f1h = smem + 2 * TP * 8;
  u32x4* f1l = smem + 3 * TP * 8;
  u32x4* f2h = smem + 4 * TP * 8;
  u32x4* f2l = f2h + TP * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x / RS, part = blockIdx.x % RS;
  const TileInfo ti = tile_info(tile, B, N, M);
  const int mblk1 = wave >> 1, nb1 = wave & 1;
  GemmPipeS<1, 1, false, 8, 2> g1;  // fstn.conv1 64->64: 2 m-blocks x 2 point blocks
  g1.prefetch(wpf1 + (mblk1 * 4) * 64 + lane, 0, 64 * 64 / 8);
  f32x4 bv1[1][4];
  load_bias_quads<1>(bv1, bf1, mblk1 * 32, lane);
  {
    float x, y, z;
    load_point(P, ti, lane, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    conv3_relu_chunks_split(x, y, z, Wc1, bc1, wave, h1h + lane * 8, h1l + lane * 8, bf_key<8>(lane));
  }
  __syncthreads();
  GemmPipeS<1, 2, false, 8, 2> g2;
  g2.prefetch(wpf2 + (wave * 4) * 64 + lane, 0, 128 * 64 / 8);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, bf2, wave * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 acc[1][1] = {{zero16()}};
    g1.run(acc, h1h + nb1 * 32 * 8, h1l + nb1 * 32 * 8, lane);
    store_tile_split<1, 1, true, 8>(acc, f1h + nb1 * 32 * 8, f1l + nb1 * 32 * 8, mblk1, bv1, lane);
  }
  __syncthreads();
  const size_t row0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE) save_tile_rows_split<64, 256, TP>(f1h, f1l, sv.s1 + row0 * 64, ti.valid, tid);
  {
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g2.run(acc, f1h, f1l, lane);
    store_tile_split<1, 2, true, 16>(acc, f2h, f2l, wave, bv2, lane);
  }
  __syncthreads();
  if (SAVE) save_tile_rows_split<128, 256, TP>(f2h, f2l, sv.s2 + row0 * 128, ti.valid, tid);
  stn_conv3_split<RS, u32x4*, SAVE>(wpf3, bf3, f2h, f2l, pm + (size_t)tile * PMW, part, wave, lane,
                                    SAVE ? sv.pmax + (size_t)tile * 1024 : nullptr, SAVE ? sv.pidx + (size_t)tile * 1024 : nullptr,
                                    (int)row0);
}

// ------------------------------------------------------------------------------------------
// rotation head: pointfeat tile -> split images -> layer 0 (recomputed) on GemmPipeS -> fused bias+GN affine + GELU ->
// split image -> layer 1 on GemmPipeS -> y1 + GN1 partials.  Otherwise k_rot_l1.  256 threads, 80 KiB LDS.
// ------------------------------------------------------------------------------------------
// SAVE: the training forward of the split mode (catre_train_rot_fwd, as k_rot_l1<1, true>): y0 = layer 0's output with its
// per-cloud bias and a0 = gelu(GN0(y0)) stored as fp32 (from an extra sweep of layer 0, see the head loop), all outputs
// head-major.
template <bool SAVE = false>
__global__ __launch_bounds__(256, 2) void k_rot_l1_split(const float* __restrict__ pointfeat,
                                                         const u32x4* __restrict__ wpl0x, const u32x4* __restrict__ wpl0y,
                                                         const float* __restrict__ aff0 /*[B*2][2][2][256]*/,
                                                         const u32x4* __restrict__ wpl1x, const u32x4* __restrict__ wpl1y,
                                                         const float* __restrict__ b1x, const float* __restrict__ b1y,
                                                         float* __restrict__ y1, float* __restrict__ gn1, int B, int N,
                                                         int M, unsigned long long* __restrict__ trace,
                                                         const float* __restrict__ bias0 = nullptr /*[2][2B][256]*/,
                                                         float* __restrict__ y0s = nullptr,
                                                         float* __restrict__ a0s = nullptr) {
  __shared__ __attribute__((aligned(16))) float smem[TP * 64 + TP * 256];  // 80 KiB exactly
  int stamp_i = 0;
#define ROTS_STAMP()                                                                                     \
  do {                                                                                                   \
    if (CATRE_TRACE_ON && trace && (threadIdx.x & 63) == 0)                                                                \
      trace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + stamp_i] = __builtin_readcyclecounter(); \
    ++stamp_i;                                                                                           \
  } while (0)
  ROTS_STAMP();
  u32x4* pfh = reinterpret_cast<u32x4*>(smem);                // [64][8 chunks] hi, 8 KiB
  u32x4* pfl = pfh + TP * 8;                                  // lo
  u32x4* a0h = reinterpret_cast<u32x4*>(smem + TP * 64);      // [64][32 chunks]
  u32x4* a0l = a0h + TP * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const RotTile rt = rot_tile(blockIdx.x, B, N, M);
  const int T = (N + TP - 1) / TP + (M + TP - 1) / TP;
  const int P = N + M;
  for (int i = tid; i < TP * 8; i += 256) {  // pointfeat tile -> split images (chunk = 8 channels in k-slot order)
    const int row = i >> 3, c = i & 7;
    const float* src = pointfeat + rt.pf_off + (size_t)min(row, rt.valid - 1) * 64 + 16 * (c >> 1) + 4 * (c & 1);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 8);
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    u32x4 hi, lo;
    split_bf8(v, hi, lo);
    pfh[bf_off<8>(row, c)] = hi;
    pfl[bf_off<8>(row, c)] = lo;
  }
  __syncthreads();
  ROTS_STAMP();
  const int n = lane & 31, h = lane >> 5;
#pragma unroll 1
  for (int hd = 0; hd < 2; ++hd) {
    {
      const float* af = aff0 + ((((size_t)rt.obj * 2 + hd) * 2 + (rt.is_obs ? 0 : 1)) * 2) * 256 + wave * 64 + 4 * h;
      if constexpr (SAVE) {
        // Saves first, from a sweep of layer 0 in the SWAPPED orientation (lane = channel wave*64 + mb*32 + n): y0 and a0
        // leave as 4-byte stores covering 128 consecutive bytes of a row per half-wave - whole L2 lines.  The hi / lo LDS
        // image below needs 8 channels of a point in one lane, which only the normal orientation has, so layer 0 (K = 64,
        // a fifth of layer 1) runs twice in this instance.  (Storing from the normal orientation - 16 bytes per lane, 32
        // rows per instruction - made this kernel L2-request-bound: 1.65 ms against 0.6 ms without saves.)
        const float* afb = aff0 + ((((size_t)rt.obj * 2 + hd) * 2 + (rt.is_obs ? 0 : 1)) * 2) * 256 + wave * 64 + n;
        const float* bqb = bias0 + ((size_t)hd * 2 * B + rt.cloud) * 256 + wave * 64 + n;
        float scv[2], shv[2], b0v[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          scv[mb] = afb[mb * 32];
          shv[mb] = afb[256 + mb * 32];
          b0v[mb] = bqb[mb * 32];
        }
        f32x16 accs[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) accs[mb][0] = accs[mb][1] = zero16();
        {
          GemmPipeS<2, 2, true, 8, 2> gs;
          gs.prefetch((hd ? wpl0y : wpl0x) + ((wave * 2) * 4) * 64 + lane, 4 * 64, 256 * 64 / 8);
          gs.run(accs, pfh, pfl, lane);
        }
        const size_t rowg = ((size_t)hd * B + rt.obj) * P + rt.gp0 + 4 * h;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const int c = wave * 64 + mb * 32 + n;
          float* yp = y0s + rowg * 256 + c;
          float* ap = a0s + rowg * 256 + c;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = nb * 32 + (r & 3) + 8 * (r >> 2);
              const float v = accs[mb][nb][r];
              st_stream(yp + row * 256, v + b0v[mb]);
              st_stream(ap + row * 256, gelu_erf(fmaf(v, scv[mb], shv[mb])));
            }
        }
      }
      f32x4 scr[3], shr[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        scr[i] = *reinterpret_cast<const f32x4*>(af + (i >> 2) * 32 + 8 * (i & 3));
        shr[i] = *reinterpret_cast<const f32x4*>(af + 256 + (i >> 2) * 32 + 8 * (i & 3));
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc[2][2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) acc[mb][0] = acc[mb][1] = zero16();
      {
        GemmPipeS<2, 2, false, 8, 2> g0;  // layer 0: 64 -> 256, wave -> m-blocks 2*wave, 2*wave + 1
        g0.prefetch((hd ? wpl0y : wpl0x) + ((wave * 2) * 4) * 64 + lane, 4 * 64, 256 * 64 / 8);
        g0.run(acc, pfh, pfl, lane);
      }
      ROTS_STAMP();
      const int key = bf_key<32>(n);
      float zprev[2][4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // step i = (mb, register quad g): channels wave*64 + mb*32 + 8g + 4h .. +3
        const int mb = i >> 2, g = i & 3;
        if (i + 2 < 8) {
          const int j = i + 2;
          scr[j % 3] = *reinterpret_cast<const f32x4*>(af + (j >> 2) * 32 + 8 * (j & 3));
          shr[j % 3] = *reinterpret_cast<const f32x4*>(af + 256 + (j >> 2) * 32 + 8 * (j & 3));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          float z[4];
          gelu_affine4(acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1], acc[mb][nb][4 * g + 2], acc[mb][nb][4 * g + 3],
                       scr[i % 3], shr[i % 3], z);
          if ((g & 1) == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) zprev[nb][q] = z[q];
          } else {  // quads g-1 and g complete chunk 4*(2*wave+mb) + 2*(g>>1) + h
            const float v[8] = {zprev[nb][0], zprev[nb][1], zprev[nb][2], zprev[nb][3], z[0], z[1], z[2], z[3]};
            u32x4 hi, lo;
            split_bf8(v, hi, lo);
            const int off = (nb * 32 + n) * 32 + ((4 * (2 * wave + mb) + 2 * (g >> 1) + h) ^ key);
            a0h[off] = hi;
            a0l[off] = lo;
          }
        }
      }
    }
    ROTS_STAMP();
    __syncthreads();
    ROTS_STAMP();
    {
      f32x16 acc[2][2];  // start at the lane's channel bias ("swapped" orientation: one channel per lane)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float bb = (hd ? b1y : b1x)[wave * 64 + mb * 32 + n];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][0][r] = acc[mb][1][r] = bb;
      }
      GemmPipeS<2, 2, true, 32, 2> g1;
      g1.prefetch((hd ? wpl1y : wpl1x) + (wave * 2 * 16) * 64 + lane, 16 * 64, 256 * 256 / 8);
      g1.run(acc, a0h, a0l, lane);
      ROTS_STAMP();
      const float inv_cnt = 1.0f / (8.f * (float)rt.valid);
      int valid_h = rt.valid - 4 * h;  // point (r, nb) of this half-wave is real iff its in-tile index < valid_h
      asm volatile("" : "+v"(valid_h));
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int ch = wave * 64 + mb * 32 + n;
        float* dst = y1 + ((SAVE ? (size_t)hd * B + rt.obj : (size_t)rt.obj * 2 + hd) * P + rt.gp0) * 256 + ch;
        float s = 0.f;
        if (rt.valid == TP) {
          float* dh = dst + (size_t)(4 * h) * 256;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = acc[mb][nb][r];
              st_stream(dh + (nb * 32 + (r & 3) + 8 * (r >> 2)) * 256, v);
              s += v;
            }
        } else {
          // ragged tile (rare): same base pointer + compile-time offsets as above; the row bound is re-materialised per
          // head so that hipcc does not hoist 32 loop-invariant predicates / addresses out of the head loop (that cost
          // 24-34 spilled SGPRs here and 6 VGPRs + scratch in the split variant)
          float* dh = dst + (size_t)(4 * h) * 256;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = acc[mb][nb][r];
              if (nb * 32 + (r & 3) + 8 * (r >> 2) < valid_h) {
                st_stream(dh + (nb * 32 + (r & 3) + 8 * (r >> 2)) * 256, v);
                s += v;
              }
            }
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 32);
        const float mean = s * inv_cnt;
        float m2 = 0.f;
        if (rt.valid == TP) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = acc[mb][nb][r] - mean;
              m2 = fmaf(d, d, m2);
            }
        } else {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = acc[mb][nb][r] - mean;
              m2 += nb * 32 + (r & 3) + 8 * (r >> 2) < valid_h ? d * d : 0.f;
            }
        }
        m2 += __shfl_xor(m2, 1);
        m2 += __shfl_xor(m2, 2);
        m2 += __shfl_xor(m2, 4);
        m2 += __shfl_xor(m2, 32);
        if ((lane & 7) == 0 && h == 0) {
          float* out = gn1 + ((SAVE ? (size_t)hd * B + rt.obj : (size_t)rt.obj * 2 + hd) * T + rt.t) * 64 + (ch >> 3) * 2;
          out[0] = mean;
          out[1] = m2;
        }
      }
    }
    ROTS_STAMP();
    __syncthreads();  // the a0 images are rewritten for the second head
    ROTS_STAMP();
  }
#undef ROTS_STAMP
}
