// catre_bf16.h - the reduced-precision variant of the fused refine path (BASELINE config 5: "bf16 with fp32
// SO(3) accumulate"; the reference reaches it through torch.cuda.amp.autocast, engine.py:304 / TEST.AMP_TEST).
// Included by catre_kernels.hip after the fp32 kernels.
//
// What is bf16 and what is not
//   * operands of every 1x1-conv GEMM over points (STN conv2/3, STNkd conv1-3, the 64x64 feature transform, trunk
//     conv2-4, rot-head layers 0/1) are rounded to bf16 (RNE) - weights once at pack time, activations when they are
//     written to LDS; the products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate);
//   * bias, ReLU, max-pool, GroupNorm statistics, GELU, the FC tails, the ts head, rot6d->SO(3) and the pose update
//     stay fp32;
//   * the two activations that travel through HBM (pointfeat [P][64] and rot-head y1 [P][256]) are stored as bf16.
//
// LDS images.  An activation image is [64 points][C channels] bf16 stored as 16-byte CHUNKS of 8 channels,
// CP = C/8 chunks per row, chunk c of row r at chunk index c ^ key(r) (XOR swizzle: the 16 rows of a
// ds_read_b128 lane group land on 16 distinct 4-bank slots; key = r&15 for rows of >= 256 B, (r>>1)&7 for the
// 128-byte rows of the 64-channel images).  The channels inside a 16-wide k-group are stored in "k-slot" order:
//     chunk 2G   = channels 16G + {0,1,2,3, 8, 9,10,11}
//     chunk 2G+1 = channels 16G + {4,5,6,7,12,13,14,15}
// which is exactly what one lane holds of a 32x32 MFMA result in the "normal" orientation (rows = channels,
// cols = points: register quad g of half-wave h = channels 8g+4h..+3), so an epilogue writes ONE ds_write_b128
// per 8 results.  The contraction index of the next layer is permuted the same way in its packed weights
// (k_pack_frag_bf), which only re-orders an fp32 sum.
//
// Packed weights: u32x4 index ((mblk*(K/16) + kc)*64 + lane) holds, for row mblk*32 + (lane&31), the 8 bf16 of
// chunk 2kc + (lane>>5) in k-slot order - one coalesced 1 KiB load per wave feeds one K=16 MFMA.
#pragma once

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                                 0, 0);
}

__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {  // v_cvt_pk_bf16_f32 (RNE)
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ u32x4 pack_bf8(const float (&v)[8]) {
  u32x4 w;
  w[0] = pack_bf2(v[0], v[1]);
  w[1] = pack_bf2(v[2], v[3]);
  w[2] = pack_bf2(v[4], v[5]);
  w[3] = pack_bf2(v[6], v[7]);
  return w;
}

// rows of 256 channels stored as fp32 (HB = false) or bf16 (HB = true): four consecutive channels of row-quad index i4
// (= row * 64 + channel / 4).  The reduced-precision training heads keep their [rows,256] activations in bf16 - what
// torch.autocast's Conv1d outputs are (engine.py:304) - and every pass over them moves half the bytes.
template <bool HB>
__device__ __forceinline__ f32x4 ld_row4(const void* __restrict__ p, size_t i4) {
  if constexpr (HB) {
    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p) + i4);
    return f32x4{bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
  } else {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i4);
  }
}
// four consecutive channels of a bf16 row as floats, through the caches (operand rows that several workgroups re-read:
// the gathered rows of the row-sparse backward)
__device__ __forceinline__ f32x4 ld_bf4(const void* __restrict__ p, size_t i4) {
  const u32x2 v = reinterpret_cast<const u32x2*>(p)[i4];
  return f32x4{bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
}
template <bool HB>
__device__ __forceinline__ void st_row4(void* __restrict__ p, size_t i4, const f32x4& v) {
  if constexpr (HB)
    reinterpret_cast<u32x2*>(p)[i4] = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
  else
    reinterpret_cast<f32x4*>(p)[i4] = v;
}

// ... and the form for loads that are requested long before they are used (the prefetch rings of k_rot_l1_bwd_bf /
// k_rot_l0_bwd_bf): the RAW quad stays in the ring - a conversion right behind the load would wait for it - and is widened
// where it is consumed.  rowq_pair: column q of two rows as one packed bf16 pair (first row in the low half).
template <bool HB>
struct RowQ {
  typedef f32x4 T;
};
template <>
struct RowQ<true> {
  typedef u32x2 T;
};
template <bool HB>
__device__ __forceinline__ typename RowQ<HB>::T ld_rowq(const void* __restrict__ p, size_t i4) {
  return __builtin_nontemporal_load(reinterpret_cast<const typename RowQ<HB>::T*>(p) + i4);
}
__device__ __forceinline__ f32x4 rowq_f32(const f32x4& v) { return v; }
__device__ __forceinline__ f32x4 rowq_f32(const u32x2& v) { return f32x4{bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])}; }
__device__ __forceinline__ unsigned rowq_pair(const f32x4& a, const f32x4& b, int q) { return pack_bf2(a[q], b[q]); }
__device__ __forceinline__ unsigned rowq_pair(const u32x2& a, const u32x2& b, int q) {
  // v_perm_b32: bytes 4-7 = first operand, 0-3 = second
  return (q & 1) ? __builtin_amdgcn_perm(b[q >> 1], a[q >> 1], 0x07060302u) : __builtin_amdgcn_perm(b[q >> 1], a[q >> 1], 0x05040100u);
}

template <int CP>
__device__ __forceinline__ int bf_key(int row) {
  return CP >= 16 ? (row & 15) : ((row >> 1) & (CP - 1));
}
template <int CP>
__device__ __forceinline__ int bf_off(int row, int chunk) {
  return row * CP + (chunk ^ bf_key<CP>(row));
}

// weights -> bf16 fragments in k-slot order (layout only + the RNE rounding)
__global__ void k_pack_frag_bf(const float* __restrict__ src, int ld, int coloff, int rows, int K,
                               unsigned short* __restrict__ dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * K) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, rest = idx >> 9;
  const int nkc = K / 16;
  const int kc = rest % nkc, mb = rest / nkc;
  const int row = mb * 32 + (lane & 31), col = kc * 16 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
  dst[idx] = __builtin_bit_cast(unsigned short, (__bf16)src[(size_t)row * ld + coloff + col]);
}

// K-sweep of a wave tile of MB x NB 32x32 blocks, K = 8*CP (CP chunks per LDS row, NKC = CP/2 MFMA steps).
// Same software pipeline as GemmPipe: weight fragments stream L2 -> registers PFD steps ahead, activation
// fragments LDS -> registers PFB steps ahead, both pinned with sched_barrier.
// XA (K >= 128 only): the image base is aligned to its row pitch, so the swizzled fragment address is ONE v_xor of a
// per-lane base with a compile-time constant - (chunk ^ key) << 4 == (chunk << 4) ^ (key << 4), and neither overlaps the
// row bits - instead of eight (sixteen past 64 KiB) per-lane pointers held across the sweep.
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
template <int MB, int NB, bool SWAP, int CP, int PFD, int PFB = 1, bool XA = false>
struct GemmPipeB {
  static constexpr int NKC = CP / 2;
  static_assert(PFD >= 1 && PFD <= NKC && PFB >= 1 && PFB <= PFD, "prefetch depth");
  static constexpr int RA = PFD + 1, RB = PFB + 1;
  u32x4 a[RA][MB], b[RB][NB];
  const u32x4* wp;
  int wp_mb;
  int ablate = 0;  // instrumented build only (catre_debug_knob 1): skip operand loads to price them; always 0 in the product

  __device__ __forceinline__ void issue_a(int kc) {
    if (CATRE_TRACE_ON && (ablate & 1) && kc >= PFD) return;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a[kc % RA][mb] = wp[mb * wp_mb + kc * 64];
  }
  __device__ __forceinline__ void prefetch(const u32x4* __restrict__ wp_, int wp_mb_) {
    wp = wp_;
    wp_mb = wp_mb_;
#pragma unroll
    for (int d = 0; d < PFD; ++d) issue_a(d);
    __builtin_amdgcn_sched_barrier(0);
  }
  // x: image row of point 0 of this wave tile (a multiple of 32 rows into the image)
  __device__ __forceinline__ void run(f32x16 (&acc)[MB][NB], const u32x4* x, int lane) {
    run(acc, x, lane, [](int) {});
  }
  // hook(kc) runs before the loads of K-step kc are issued (k_trunk_bf2 re-balances the two waves of a SIMD there)
  template <class Hook>
  __device__ __forceinline__ void run(f32x16 (&acc)[MB][NB], const u32x4* x, int lane, Hook hook) {
    const int n = lane & 31, h = lane >> 5, key = bf_key<CP>(n);
    const u32x4* xrow = x + n * CP;
    // rows of >= 256 B: the XOR key (< 16) only touches the low four bits of the chunk index 2*kc + h, so eight per-lane
    // pointers (kc & 7) + compile-time offsets (kc >> 3, nb) address every fragment - no address arithmetic in the sweep
    // (left to itself the compiler keeps one swizzled index per kc in registers: 32 of them for K = 512)
    const u32x4* xl[8];
    if constexpr (CP >= 16 && !XA) {
#pragma unroll
      for (int lo = 0; lo < 8; ++lo) xl[lo] = xrow + ((2 * lo + h) ^ key);
    }
    unsigned xa0 = 0;  // XA: LDS byte address of (row n, chunk h ^ key)
    if constexpr (XA) {
      static_assert(CP >= 16, "XA needs rows of >= 256 B");
      xa0 = (unsigned)(size_t)(lds_cu32x4*)xrow ^ (unsigned)((h ^ key) << 4);
    }
    auto issue_b = [&](int kc) {
      if (CATRE_TRACE_ON && (ablate & 2) && kc >= PFB) return;
      if constexpr (XA) {
        // two point blocks (2 x 32 rows) per 16-bit ds offset window
        unsigned a;
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a) : "i"(kc << 5), "v"(xa0));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          constexpr int WIN = 65536 / (32 * CP * 16);  // point blocks per 64 KiB
          static_assert(WIN >= 1, "row block larger than the ds offset window");
          const unsigned aw = a + (unsigned)((nb / WIN) * WIN * 32 * CP * 16);
          b[kc % RB][nb] = *(lds_cu32x4*)(size_t)(aw + (unsigned)((nb % WIN) * 32 * CP * 16));
        }
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if constexpr (CP >= 16)
            b[kc % RB][nb] = xl[kc & 7][nb * 32 * CP + 16 * (kc >> 3)];
          else
            b[kc % RB][nb] = xrow[nb * 32 * CP + ((2 * kc + h) ^ key)];
        }
      }
    };
#pragma unroll
    for (int d = 0; d < PFB; ++d) issue_b(d);
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      hook(kc);
      if (kc + PFD < NKC) issue_a(kc + PFD);
      if (kc + PFB < NKC) issue_b(kc + PFB);
      __builtin_amdgcn_sched_barrier(0);
      const int ca = kc % RA, cb = kc % RB;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[mb][nb] = SWAP ? mfma_bf(b[cb][nb], a[ca][mb], acc[mb][nb]) : mfma_bf(a[ca][mb], b[cb][nb], acc[mb][nb]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// "normal"-orientation epilogue: act(acc + bias) -> bf16 image, one 16-byte chunk per 8 results.
// mblk0 = absolute index of the wave tile's first 32-channel block; img = image row of point 0 of the tile.
template <int MB, int NB, bool RELU, int CP>
__device__ __forceinline__ void store_tile_bf(const f32x16 (&acc)[MB][NB], u32x4* img, int mblk0,
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
        img[(nb * 32 + n) * CP + (chunk ^ key)] = pack_bf8(v);
      }
    }
}

// conv 3 -> 16 channels [16*grp, +16) of one point on the VALU (fp32), ReLU, -> the two chunks of k-group grp
__device__ __forceinline__ void conv3_relu_chunks(float x, float y, float z, const float* __restrict__ W,
                                                  const float* __restrict__ b, int grp, u32x4* row, int key) {
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
  row[(2 * grp) ^ key] = pack_bf8(c0);
  row[(2 * grp + 1) ^ key] = pack_bf8(c1);
}

// channel of element e of chunk c (k-slot order)
__device__ __forceinline__ int bf_chunk_channel(int c, int e) { return 16 * (c >> 1) + 8 * (e >> 2) + 4 * (c & 1) + (e & 3); }

// Training forward (SAVE instances): a bf16 LDS image [ROWS][C] (chunks in k-slot order, swizzled) -> fp32 rows dst[row][C] in
// channel order, rows >= valid skipped.  What the layer-wise backward reads; the values are the bf16-rounded activations - the
// reduced-precision dgrad / wgrad kernels round their operands to bf16 when they stage them anyway.
template <int C, int NT, int ROWS>
__device__ __forceinline__ void save_tile_rows_bf(const u32x4* __restrict__ img, float* __restrict__ dst, int valid, int tid) {
  constexpr int CP = C / 8;
#pragma unroll
  for (int u = 0; u < ROWS * CP / NT; ++u) {
    const int i = tid + NT * u, row = i / CP, c = i % CP;
    if (row >= valid) continue;
    const u32x4 v = img[bf_off<CP>(row, c)];
    const f32x4 lo = {bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
    const f32x4 hi = {bf_lo(v[2]), bf_hi(v[2]), bf_lo(v[3]), bf_hi(v[3])};
    float* d = dst + (size_t)row * C + 16 * (c >> 1) + 4 * (c & 1);
    st_stream(reinterpret_cast<f32x4*>(d), lo);
    st_stream(reinterpret_cast<f32x4*>(d + 8), hi);
  }
}

// The same image as bf16 ROWS dst[row][C] in channel order (dst: the row buffer, C / 2 floats per row): half the bytes of
// the fp32 form - the SAVE instances wait for these stores (k_trunk_bf2<true>: 1.5 GB of them as fp32 rows) - and nothing
// is lost: the fp32 rows held the bf16-rounded values, and the backward kernels that read them (catre_op_maxlin_bwd_*_h,
// catre_op_gemm_rows_nr / catre_op_gemm_tn_bias_nr with CATRE_ROWS_BF16) see the same numbers.  A thread takes a chunk
// PAIR - 16 consecutive channels: [even.e0-3 | odd.e0-3 | even.e4-7 | odd.e4-7] - and stores 32 contiguous bytes.
template <int C, int NT, int ROWS>
__device__ __forceinline__ void save_tile_rows_bf16(const u32x4* __restrict__ img, float* __restrict__ dst, int valid, int tid) {
  constexpr int CP = C / 8, PP = CP / 2;
  static_assert((ROWS * PP) % NT == 0 || ROWS * PP < NT, "whole trips");
#pragma unroll
  for (int u = 0; u < (ROWS * PP + NT - 1) / NT; ++u) {
    const int i = tid + NT * u, row = i / PP, q = i % PP;
    if (i >= ROWS * PP || row >= valid) continue;
    const u32x4 ev = img[bf_off<CP>(row, 2 * q)], od = img[bf_off<CP>(row, 2 * q + 1)];
    u32x4* d = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(dst) + (size_t)row * C + 16 * q);
    st_stream(d, u32x4{ev[0], ev[1], od[0], od[1]});
    st_stream(d + 1, u32x4{ev[2], ev[3], od[2], od[3]});
  }
}

// ------------------------------------------------------------------------------------------
// a2: STN3d conv stack (pointnet.py:24-28), bf16 operands.  256 threads, 24 KiB LDS.
// ------------------------------------------------------------------------------------------
template <bool SAVE = false>
__global__ __launch_bounds__(256, 2) void k_stn3d_bf(catre_points P, const float* __restrict__ W1,
                                                     const float* __restrict__ b1, const u32x4* __restrict__ wp2,
                                                     const float* __restrict__ b2, const u32x4* __restrict__ wp3,
                                                     const float* __restrict__ b3, float* __restrict__ pm, int B, int N,
                                                     int M, TrainSave sv = TrainSave{}) {
  __shared__ u32x4 smem[TP * 8 + TP * 16];
  u32x4* a1 = smem;           // [64][64 ch]
  u32x4* a2 = smem + TP * 8;  // [64][128 ch]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TileInfo ti = tile_info(blockIdx.x, B, N, M);

  GemmPipeB<1, 2, false, 8, 3> g2;  // conv2 64->128: wave -> m-block `wave`
  g2.prefetch(wp2 + (wave * 4) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, wave * 32, lane);
  {
    float x, y, z;
    load_point(P, ti, lane, x, y, z);
    conv3_relu_chunks(x, y, z, W1, b1, wave, a1 + lane * 8, bf_key<8>(lane));
  }
  __syncthreads();
  const size_t row0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE) save_tile_rows_bf16<64, 256, TP>(a1, sv.s1 + row0 * 32, ti.valid, tid);
  // conv3 128->1024 + max: wave owns m-blocks [8*wave, +8) in two passes of 4
  GemmPipeB<4, 2, true, 16, 2, 1> g3a, g3b;
  float bl[2][4];
  g3a.prefetch(wp3 + ((wave * 8) * 8) * 64 + lane, 8 * 64);
  load_bias_lane<4>(bl[0], b3, (wave * 8) * 32, lane);
  load_bias_lane<4>(bl[1], b3, (wave * 8 + 4) * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g2.run(acc, a1, lane);
    store_tile_bf<1, 2, true, 16>(acc, a2, wave, bv2, lane);
  }
  __syncthreads();
  if (SAVE) save_tile_rows_bf16<128, 256, TP>(a2, sv.s2 + row0 * 64, ti.valid, tid);
  float* out = pm + (size_t)blockIdx.x * PMW;
  float* pmax = SAVE ? sv.pmax + (size_t)blockIdx.x * 1024 : nullptr;
  int* pidx = SAVE ? sv.pidx + (size_t)blockIdx.x * 1024 : nullptr;
  {
    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3a.run(acc, a2, lane);
    g3b.prefetch(wp3 + ((wave * 8 + 4) * 8) * 64 + lane, 8 * 64);
    if (SAVE)
      argmax_tile_store<4, 2>(acc, pmax, pidx, (wave * 8) * 32, bl[0], (int)row0, lane);
    else
      max_tile_store_pre<4, 2>(acc, out, (wave * 8) * 32, bl[0], true, lane);
  }
  {
    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3b.run(acc, a2, lane);
    if (SAVE)
      argmax_tile_store<4, 2>(acc, pmax, pidx, (wave * 8 + 4) * 32, bl[1], (int)row0, lane);
    else
      max_tile_store_pre<4, 2>(acc, out, (wave * 8 + 4) * 32, bl[1], true, lane);
  }
}

// ------------------------------------------------------------------------------------------
// a3+a4: x T3 -> relu(conv1) -> STNkd conv stack 64->64->128->1024 (+ReLU) + per-tile max, bf16 operands.
// ------------------------------------------------------------------------------------------
template <bool SAVE = false>
__global__ __launch_bounds__(256, 2) void k_stnkd_bf(catre_points P, const float* __restrict__ trans3,
                                                     const float* __restrict__ Wc1, const float* __restrict__ bc1,
                                                     const u32x4* __restrict__ wpf1, const float* __restrict__ bf1,
                                                     const u32x4* __restrict__ wpf2, const float* __restrict__ bf2,
                                                     const u32x4* __restrict__ wpf3, const float* __restrict__ bf3,
                                                     float* __restrict__ pm, int B, int N, int M,
                                                     TrainSave sv = TrainSave{}) {
  __shared__ u32x4 smem[2 * TP * 8 + TP * 16];
  u32x4* h1 = smem;
  u32x4* f1 = smem + TP * 8;
  u32x4* f2 = smem + 2 * TP * 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TileInfo ti = tile_info(blockIdx.x, B, N, M);

  const int mblk1 = wave >> 1, nb1 = wave & 1;
  GemmPipeB<1, 1, false, 8, 4> g1;  // fstn.conv1 64->64: 2 m-blocks x 2 point blocks
  g1.prefetch(wpf1 + (mblk1 * 4) * 64 + lane, 0);
  f32x4 bv1[1][4];
  load_bias_quads<1>(bv1, bf1, mblk1 * 32, lane);
  {
    float x, y, z;
    load_point(P, ti, lane, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    conv3_relu_chunks(x, y, z, Wc1, bc1, wave, h1 + lane * 8, bf_key<8>(lane));
  }
  __syncthreads();
  GemmPipeB<1, 2, false, 8, 3> g2;
  g2.prefetch(wpf2 + (wave * 4) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, bf2, wave * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 acc[1][1] = {{zero16()}};
    g1.run(acc, h1 + nb1 * 32 * 8, lane);
    store_tile_bf<1, 1, true, 8>(acc, f1 + nb1 * 32 * 8, mblk1, bv1, lane);
  }
  __syncthreads();
  const size_t row0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE) save_tile_rows_bf16<64, 256, TP>(f1, sv.s1 + row0 * 32, ti.valid, tid);
  GemmPipeB<4, 2, true, 16, 2, 1> g3a, g3b;
  float bl[2][4];
  g3a.prefetch(wpf3 + ((wave * 8) * 8) * 64 + lane, 8 * 64);
  load_bias_lane<4>(bl[0], bf3, (wave * 8) * 32, lane);
  load_bias_lane<4>(bl[1], bf3, (wave * 8 + 4) * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g2.run(acc, f1, lane);
    store_tile_bf<1, 2, true, 16>(acc, f2, wave, bv2, lane);
  }
  __syncthreads();
  if (SAVE) save_tile_rows_bf16<128, 256, TP>(f2, sv.s2 + row0 * 64, ti.valid, tid);
  float* out = pm + (size_t)blockIdx.x * PMW;
  float* pmax = SAVE ? sv.pmax + (size_t)blockIdx.x * 1024 : nullptr;
  int* pidx = SAVE ? sv.pidx + (size_t)blockIdx.x * 1024 : nullptr;
  {
    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3a.run(acc, f2, lane);
    g3b.prefetch(wpf3 + ((wave * 8 + 4) * 8) * 64 + lane, 8 * 64);
    if (SAVE)
      argmax_tile_store<4, 2>(acc, pmax, pidx, (wave * 8) * 32, bl[0], (int)row0, lane);
    else
      max_tile_store_pre<4, 2>(acc, out, (wave * 8) * 32, bl[0], true, lane);
  }
  {
    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3b.run(acc, f2, lane);
    if (SAVE)
      argmax_tile_store<4, 2>(acc, pmax, pidx, (wave * 8 + 4) * 32, bl[1], (int)row0, lane);
    else
      max_tile_store_pre<4, 2>(acc, out, (wave * 8 + 4) * 32, bl[1], true, lane);
  }
}

// ------------------------------------------------------------------------------------------
// a3+a5: trunk (pointnet.py:98-116), bf16 operands.  256 threads and exactly 80 KiB of LDS so that TWO
// workgroups share a CU: while one sweeps conv4 (MFMA/L2-bound) the other runs its short, latency-bound
// prologue (conv1, feature transform, conv2, conv3) - with the 16x faster matrix rate those phases would
// otherwise be ~1/3 of a lone workgroup's time.
//   a3 [64][512 ch] 64 KiB | a2 [64][128 ch] 16 KiB; h1 / T64 image / pointfeat image alias the a3 region.
// pointfeat leaves as bf16 chunks [point][8] (k-slot order) - the layout the rotation head consumes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_trunk_bf(catre_points P, const float* __restrict__ trans3,
                                                     const float* __restrict__ trans64, const float* __restrict__ Wc1,
                                                     const float* __restrict__ bc1, const u32x4* __restrict__ wp2,
                                                     const float* __restrict__ b2, const u32x4* __restrict__ wp3,
                                                     const float* __restrict__ b3, const u32x4* __restrict__ wp4,
                                                     const float* __restrict__ b4, float* __restrict__ pm,
                                                     u32x4* __restrict__ pointfeat, int B, int N, int M,
                                                     unsigned long long* __restrict__ trace) {
  __shared__ u32x4 smem[TP * 64 + TP * 16];
#define TRUNKB_STAMP(i)                                                                                    \
  do {                                                                                                     \
    if (CATRE_TRACE_ON && trace && (threadIdx.x & 63) == 0)                                                                  \
      trace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_readcyclecounter();      \
  } while (0)
  u32x4* a3 = smem;
  u32x4* a2 = smem + TP * 64;
  u32x4* h1 = smem;                                            // [64][8]
  u32x4* tA = smem + TP * 8;                                   // [64 j][8]: T64 transposed, rows = out channel j
  u32x4* pf = smem + 2 * TP * 8;                               // [64][8]
  float* scratch = reinterpret_cast<float*>(smem + 3 * TP * 8);  // [4][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TileInfo ti = tile_info(blockIdx.x, B, N, M);
  const bool ft = trans64 != nullptr;
  const int n = lane & 31, h = lane >> 5;
  TRUNKB_STAMP(0);

  GemmPipeB<1, 2, false, 8, 3> g2;  // conv2 64->128: wave -> m-block `wave`, both point blocks
  g2.prefetch(wp2 + (wave * 4) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, wave * 32, lane);
  {
    float x, y, z;
    load_point(P, ti, lane, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    conv3_relu_chunks(x, y, z, Wc1, bc1, wave, (ft ? h1 : pf) + lane * 8, bf_key<8>(lane));
    if (ft) {  // A-operand image of the feature transform: row j holds T64[i][j] over i (pointnet.py:107-109)
      const float* src = trans64 + (size_t)ti.cloud * 4096 + (wave * 16) * 64 + lane;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = src[r * 64];
      const float c0[8] = {v[0], v[1], v[2], v[3], v[8], v[9], v[10], v[11]};
      const float c1[8] = {v[4], v[5], v[6], v[7], v[12], v[13], v[14], v[15]};
      const int key = bf_key<8>(lane);
      tA[lane * 8 + ((2 * wave) ^ key)] = pack_bf8(c0);
      tA[lane * 8 + ((2 * wave + 1) ^ key)] = pack_bf8(c1);
    }
  }
  __syncthreads();
  TRUNKB_STAMP(1);
  if (ft) {
    {  // pointfeat[j][p] = sum_i T64[i][j] h1[i][p]: 2 m-blocks x 2 point blocks, one per wave
      const int mblk = wave >> 1, nb = wave & 1, key = bf_key<8>(n);
      f32x16 acc[1][1] = {{zero16()}};
      const u32x4* ar = tA + (mblk * 32 + n) * 8;
      const u32x4* br = h1 + (nb * 32 + n) * 8;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) acc[0][0] = mfma_bf(ar[(2 * kc + h) ^ key], br[(2 * kc + h) ^ key], acc[0][0]);
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 zb[1][4] = {{z4, z4, z4, z4}};
      store_tile_bf<1, 1, false, 8>(acc, pf + nb * 32 * 8, mblk, zb, lane);
    }
    __syncthreads();
  }
  TRUNKB_STAMP(2);
  // conv3 128->512: wave owns m-blocks [4*wave, +4) in two passes of 2; first weights + bias requested now
  GemmPipeB<2, 2, false, 16, 3, 1> g3a, g3b;
  g3a.prefetch(wp3 + ((wave * 4) * 8) * 64 + lane, 8 * 64);
  f32x4 bv3[2][4];
  load_bias_quads<2>(bv3, b3, wave * 128, lane);
  __builtin_amdgcn_sched_barrier(0);
  // pointfeat tile -> registers (stored to HBM after the last barrier) and its per-channel max over the tile
  const int pf_cc = tid & 7, pf_row = tid >> 3;  // 8 chunks x 32 rows, rows r and r+32
  const u32x4 pfc0 = pf[bf_off<8>(pf_row, pf_cc)], pfc1 = pf[bf_off<8>(pf_row + 32, pf_cc)];
  {
    float m[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m[2 * i] = fmaxf(bf_lo(pfc0[i]), bf_lo(pfc1[i]));
      m[2 * i + 1] = fmaxf(bf_hi(pfc0[i]), bf_hi(pfc1[i]));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      m[e] = fmaxf(m[e], __shfl_xor(m[e], 8));
      m[e] = fmaxf(m[e], __shfl_xor(m[e], 16));
      m[e] = fmaxf(m[e], __shfl_xor(m[e], 32));
    }
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) scratch[wave * 64 + bf_chunk_channel(lane, e)] = m[e];
    }
  }
  {
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g2.run(acc, pf, lane);
    store_tile_bf<1, 2, true, 16>(acc, a2, wave, bv2, lane);
  }
  __syncthreads();
  float pf_max = 0.f;
  if (tid < 64) pf_max = fmaxf(fmaxf(scratch[tid], scratch[64 + tid]), fmaxf(scratch[128 + tid], scratch[192 + tid]));
  __syncthreads();  // scratch / pf / h1 live inside a3, which conv3 overwrites next
  TRUNKB_STAMP(3);
  {
    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3a.run(acc, a2, lane);
    g3b.prefetch(wp3 + ((wave * 4 + 2) * 8) * 64 + lane, 8 * 64);
    store_tile_bf<2, 2, true, 64>(acc, a3, wave * 4, bv3, lane);
    load_bias_quads<2>(bv3, b3, wave * 128 + 64, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g3b.run(acc, a2, lane);
    store_tile_bf<2, 2, true, 64>(acc, a3, wave * 4 + 2, bv3, lane);
  }
  TRUNKB_STAMP(4);
  // conv4 512->1024 + max: wave owns m-blocks [8*wave, +8) in two passes of 4
  GemmPipeB<4, 2, true, 64, 2, 1> g4a, g4b;
  g4a.prefetch(wp4 + ((wave * 8) * 32) * 64 + lane, 32 * 64);
  float bl4[2][4];
  load_bias_lane<4>(bl4[0], b4, (wave * 8) * 32, lane);
  load_bias_lane<4>(bl4[1], b4, (wave * 8 + 4) * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  TRUNKB_STAMP(5);
  {  // deferred HBM stores (a barrier would otherwise wait for their acknowledge)
    const size_t prow0 = ti.is_obs ? (size_t)ti.obj * N + ti.p0 : (size_t)B * N + (size_t)ti.obj * M + ti.p0;
    if (pf_row < ti.valid) pointfeat[(prow0 + pf_row) * 8 + pf_cc] = pfc0;
    if (pf_row + 32 < ti.valid) pointfeat[(prow0 + pf_row + 32) * 8 + pf_cc] = pfc1;
    if (tid < 64) pm[(size_t)blockIdx.x * PMW + 1024 + tid] = pf_max;
  }
  float* out = pm + (size_t)blockIdx.x * PMW;
  {
    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g4a.run(acc, a3, lane);
    g4b.prefetch(wp4 + ((wave * 8 + 4) * 32) * 64 + lane, 32 * 64);
    max_tile_store_pre<4, 2>(acc, out, (wave * 8) * 32, bl4[0], false, lane);
    TRUNKB_STAMP(6);
  }
  {
    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    g4b.run(acc, a3, lane);
    max_tile_store_pre<4, 2>(acc, out, (wave * 8 + 4) * 32, bl4[1], false, lane);
  }
  TRUNKB_STAMP(7);
#undef TRUNKB_STAMP
}

// ------------------------------------------------------------------------------------------
// a3+a5 for LARGE grids: the same trunk on PAIRS of tiles (128 points per workgroup, 512 threads, all 160 KiB of LDS).
// Why: at the bf16 matrix rate k_trunk_bf is bound by the L2 -> CU weight stream, not by the MFMAs - every 64-point
// tile pulls the whole 1.28 MB of bf16 weights (conv4: 512 B per MFMA with a 4 x 2 wave tile = 64 B/clk/CU at full
// matrix rate, above what the L2 sustains per CU), so its matrix pipes sit at 60 %.  Here a wave tile is 2 m-blocks x
// 4 point blocks: one weight fragment feeds four MFMAs (256 B per MFMA from L2, 512 B from LDS = half the LDS rate).
// Same contraction order per output as k_trunk_bf (kc ascending into one fp32 accumulator) and exact maxima, so the
// two kernels return the same bits; the pair's maxima are written to BOTH tiles' rows of the partial-max buffer
// (k_reduce_pm takes a maximum over them).  A pair never straddles clouds; a cloud with an odd tile count ends in a
// pair whose second half holds duplicates of the last point.
//   a3 [128][512 ch] 128 KiB | a2 [128][128 ch] 32 KiB; h1 / T64 image / pointfeat image / scratch alias a3.
// ------------------------------------------------------------------------------------------
template <int MB, int NB, bool RELU = false>
__device__ __forceinline__ void max_tile_store_pre2(const f32x16 (&acc)[MB][NB], float* __restrict__ out,
                                                    float* __restrict__ out2, int ch0, const float* __restrict__ bias,
                                                    int lane) {
  // the bias is requested here and consumed after the 16 * NB maxima + shuffle of each m-block (its latency hides
  // behind them; held across the sweep it would cost registers the 2 x 4 wave tile does not have)
  float bl[MB];
  int l31;  // lane & 31, recomputed here (opaque to CSE: kept live across the sweep it costs the register that spills)
  asm volatile("v_and_b32 %0, 31, %1" : "=v"(l31) : "v"(lane));
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) bl[mb] = bias[ch0 + mb * 32 + l31];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    float m = acc[mb][0][0];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[mb][nb][r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    if (lane < 32) {
      const float t = m + bl[mb];
      const float v = RELU ? fmaxf(t, 0.f) : t;
      out[ch0 + mb * 32 + l31] = v;
      if (out2) out2[ch0 + mb * 32 + l31] = v;
    }
  }
}

// arg-max form of the epilogue above (training): first maximum in point order wins, like torch.max; written to the rows of
// both tiles of the pair
template <int MB, int NB>
__device__ __forceinline__ void argmax_pair_store(const f32x16 (&acc)[MB][NB], float* __restrict__ pmax, int* __restrict__ pidx,
                                                  bool has2, int ch0, const float* __restrict__ bias, int row0, int lane) {
  int n, h;  // lane & 31, lane >> 5: recomputed here, opaque to CSE (held across the sweep they are what spills)
  asm volatile("v_and_b32 %0, 31, %1" : "=v"(n) : "v"(lane));
  asm volatile("v_lshrrev_b32 %0, 5, %1" : "=v"(h) : "v"(lane));
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const float bl = bias[ch0 + mb * 32 + n];
    float m = -INFINITY;
    int am = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // increasing point order inside a half-wave: strict > keeps the first
        const float v = acc[mb][nb][r] + bl;
        if (v > m) {
          m = v;
          am = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        }
      }
    const float mo = __shfl_xor(m, 32);
    const int ao = __shfl_xor(am, 32);
    if (mo > m || (mo == m && ao < am)) {
      m = mo;
      am = ao;
    }
    if (h == 0) {
      pmax[ch0 + mb * 32 + n] = m;
      pidx[ch0 + mb * 32 + n] = row0 + am;
      if (has2) {
        pmax[1024 + ch0 + mb * 32 + n] = m;
        pidx[1024 + ch0 + mb * 32 + n] = row0 + am;
      }
    }
  }
}

// pair bookkeeping: a TileInfo of up to 128 valid points + the index of its first 64-point tile
__device__ __forceinline__ void pair_info(int bid, int B, int N, int M, TileInfo& ti, int& tile0) {
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

// ------------------------------------------------------------------------------------------
// STN conv stacks on PAIRS of tiles (large grids, like k_trunk_bf2): 128 points and 256 threads per workgroup, two
// workgroups per CU.  At the bf16 rate the 64-point kernels spend half their time in the point loads / conv1 / conv2
// prologue and in the start-up of four K = 128 sweeps; a pair amortises both over twice the MFMAs (2 x 4 wave tiles: a
// weight fragment feeds four MFMAs).  Same contraction order per output, exact maxima: the bits of k_stn3d_bf / k_stnkd_bf.
// conv3 128 -> 1024 + ReLU + max: the wave owns m-blocks [8 wave, +8) in four passes of two.
// ------------------------------------------------------------------------------------------
typedef GemmPipeB<2, 4, true, 16, 3, 1, true> StnConv3Pipe;
// g[0] arrives with its first weight fragments already requested (stn_conv3_prefetch, issued by the caller BEFORE the conv2
// phase: the L2 round trip then runs under conv2 and its barrier instead of after them)
__device__ __forceinline__ void stn_conv3_prefetch(StnConv3Pipe (&g)[2], const u32x4* __restrict__ wp3, int wave, int lane) {
  g[0].prefetch(wp3 + ((size_t)(wave * 8) * 8) * 64 + lane, 8 * 64);
}
template <bool SAVE = false>
__device__ __forceinline__ void stn_conv3_pair_bf(StnConv3Pipe (&g)[2], const u32x4* __restrict__ wp3,
                                                  const float* __restrict__ b3, const u32x4* a2,
                                                  float* __restrict__ out, float* __restrict__ out2, int wave, int lane,
                                                  float* __restrict__ pmax = nullptr, int* __restrict__ pidx = nullptr,
                                                  bool has2 = false, int row0 = 0) {
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = zero16();
    g[ps & 1].run(acc, a2, lane);
    if (ps < 3) g[(ps + 1) & 1].prefetch(wp3 + ((size_t)(wave * 8 + (ps + 1) * 2) * 8) * 64 + lane, 8 * 64);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SAVE)  // training: (max, arg-max row) of the pre-ReLU values, the pooled ReLU is the graph's
      argmax_pair_store<2, 4>(acc, pmax, pidx, has2, (wave * 8 + ps * 2) * 32, b3, row0, lane);
    else
      max_tile_store_pre2<2, 4, true>(acc, out, out2, (wave * 8 + ps * 2) * 32, b3, lane);
  }
}

// SAVE (training forward under autocast, N and M multiples of 64): like k_stn3d_bf<true> / k_stnkd_bf<true> - the conv1 / conv2
// images as fp32 rows (bf16 values) and the per-tile (max, arg-max row) pairs, written to both tiles' rows of a pair.
template <bool SAVE = false>
__global__ __launch_bounds__(256, 2) void k_stn3d_bf2(catre_points P, const float* __restrict__ W1,
                                                      const float* __restrict__ b1, const u32x4* __restrict__ wp2,
                                                      const float* __restrict__ b2, const u32x4* __restrict__ wp3,
                                                      const float* __restrict__ b3, float* __restrict__ pm, int B, int N,
                                                      int M, TrainSave sv = TrainSave{}) {
  __shared__ __attribute__((aligned(1024))) u32x4 smem[2 * TP * 16 + 2 * TP * 8];
  u32x4* a2 = smem;                // [128][128 ch] (first: its rows are the XOR-addressed ones)
  u32x4* a1 = smem + 2 * TP * 16;  // [128][64 ch]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  TileInfo ti;
  int tile0;
  pair_info(blockIdx.x, B, N, M, ti, tile0);

  GemmPipeB<1, 4, false, 8, 3> g2;  // conv2 64->128: wave -> m-block `wave`, all four point blocks
  g2.prefetch(wp2 + (wave * 4) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, wave * 32, lane);
  {  // conv1 3->64 on the VALU: thread = (point, two 16-channel groups)
    const int p = (wave & 1) * TP + lane, g0 = (wave >> 1) * 2;
    float x, y, z;
    load_point(P, ti, p, x, y, z);
    conv3_relu_chunks(x, y, z, W1, b1, g0, a1 + p * 8, bf_key<8>(p));
    conv3_relu_chunks(x, y, z, W1, b1, g0 + 1, a1 + p * 8, bf_key<8>(p));
  }
  __syncthreads();
  const size_t row0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE) save_tile_rows_bf16<64, 256, 2 * TP>(a1, sv.s1 + row0 * 32, ti.valid, tid);
  StnConv3Pipe g3[2];
  stn_conv3_prefetch(g3, wp3, wave, lane);
  {
    f32x16 acc[1][4] = {{zero16(), zero16(), zero16(), zero16()}};
    g2.run(acc, a1, lane);
    store_tile_bf<1, 4, true, 16>(acc, a2, wave, bv2, lane);
  }
  __syncthreads();
  if (SAVE) save_tile_rows_bf16<128, 256, 2 * TP>(a2, sv.s2 + row0 * 64, ti.valid, tid);
  float* out = pm + (size_t)tile0 * PMW;
  if constexpr (SAVE)
    stn_conv3_pair_bf<true>(g3, wp3, b3, a2, out, nullptr, wave, lane, sv.pmax + (size_t)tile0 * 1024,
                            sv.pidx + (size_t)tile0 * 1024, ti.valid > TP, (int)row0);
  else
    stn_conv3_pair_bf(g3, wp3, b3, a2, out, ti.valid > TP ? out + PMW : nullptr, wave, lane);
}

template <bool SAVE = false>
__global__ __launch_bounds__(256, 2) void k_stnkd_bf2(catre_points P, const float* __restrict__ trans3,
                                                      const float* __restrict__ Wc1, const float* __restrict__ bc1,
                                                      const u32x4* __restrict__ wpf1, const float* __restrict__ bf1,
                                                      const u32x4* __restrict__ wpf2, const float* __restrict__ bf2,
                                                      const u32x4* __restrict__ wpf3, const float* __restrict__ bf3,
                                                      float* __restrict__ pm, int B, int N, int M,
                                                      TrainSave sv = TrainSave{}) {
  __shared__ __attribute__((aligned(1024))) u32x4 smem[2 * TP * 16 + 2 * 2 * TP * 8];
  u32x4* f2 = smem;                              // [128][128 ch]
  u32x4* h1 = smem + 2 * TP * 16;                // [128][64 ch]
  u32x4* f1 = smem + 2 * TP * 16 + 2 * TP * 8;   // [128][64 ch]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  TileInfo ti;
  int tile0;
  pair_info(blockIdx.x, B, N, M, ti, tile0);

  const int mblk1 = wave >> 1, half1 = wave & 1;
  GemmPipeB<1, 2, false, 8, 4> g1;  // fstn.conv1 64->64: 2 m-blocks x 4 point blocks, (m-block, point half) per wave
  g1.prefetch(wpf1 + (mblk1 * 4) * 64 + lane, 0);
  f32x4 bv1[1][4];
  load_bias_quads<1>(bv1, bf1, mblk1 * 32, lane);
  {
    const int p = (wave & 1) * TP + lane, g0 = (wave >> 1) * 2;
    float x, y, z;
    load_point(P, ti, p, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    conv3_relu_chunks(x, y, z, Wc1, bc1, g0, h1 + p * 8, bf_key<8>(p));
    conv3_relu_chunks(x, y, z, Wc1, bc1, g0 + 1, h1 + p * 8, bf_key<8>(p));
  }
  __syncthreads();
  GemmPipeB<1, 4, false, 8, 3> g2;
  g2.prefetch(wpf2 + (wave * 4) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, bf2, wave * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g1.run(acc, h1 + half1 * TP * 8, lane);
    store_tile_bf<1, 2, true, 8>(acc, f1 + half1 * TP * 8, mblk1, bv1, lane);
  }
  __syncthreads();
  const size_t row0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE) save_tile_rows_bf16<64, 256, 2 * TP>(f1, sv.s1 + row0 * 32, ti.valid, tid);
  StnConv3Pipe g3[2];
  stn_conv3_prefetch(g3, wpf3, wave, lane);
  {
    f32x16 acc[1][4] = {{zero16(), zero16(), zero16(), zero16()}};
    g2.run(acc, f1, lane);
    store_tile_bf<1, 4, true, 16>(acc, f2, wave, bv2, lane);
  }
  __syncthreads();
  if (SAVE) save_tile_rows_bf16<128, 256, 2 * TP>(f2, sv.s2 + row0 * 64, ti.valid, tid);
  float* out = pm + (size_t)tile0 * PMW;
  if constexpr (SAVE)
    stn_conv3_pair_bf<true>(g3, wpf3, bf3, f2, out, nullptr, wave, lane, sv.pmax + (size_t)tile0 * 1024,
                            sv.pidx + (size_t)tile0 * 1024, ti.valid > TP, (int)row0);
  else
    stn_conv3_pair_bf(g3, wpf3, bf3, f2, out, ti.valid > TP ? out + PMW : nullptr, wave, lane);
}

#define TRUNKB2_SMEM (2 * TP * 64 + 2 * TP * 16)
#ifndef CATRE_BF2_PFD
#define CATRE_BF2_PFD 4  // conv4 weight K-steps in flight per wave
#endif
// SAVE (training forward under autocast, catre_train_trunk_fwd): additionally writes the fp32 rows the layer-wise backward
// reads - x1 = x T3 (sv.s1, [rows,8]), h1 (sv.s2), conv2 / conv3 outputs (sv.s3, sv.s4), pointfeat (sv.s5, [rows,64]; the
// bf16 `pointfeat` buffer is then not written) - and the per-tile (max, arg-max row) pairs instead of the maxima.
template <bool SAVE = false>
__global__ __launch_bounds__(512) void k_trunk_bf2(catre_points P, const float* __restrict__ trans3,
                                                   const float* __restrict__ trans64, const float* __restrict__ Wc1,
                                                   const float* __restrict__ bc1, const u32x4* __restrict__ wp2,
                                                   const float* __restrict__ b2, const u32x4* __restrict__ wp3,
                                                   const float* __restrict__ b3, const u32x4* __restrict__ wp4,
                                                   const float* __restrict__ b4, float* __restrict__ pm,
                                                   u32x4* __restrict__ pointfeat, int B, int N, int M,
                                                   unsigned long long* __restrict__ trace, TrainSave sv = TrainSave{}) {
  __shared__ __attribute__((aligned(1024))) u32x4 smem[TRUNKB2_SMEM];
#define TRUNKB2_STAMP(i)                                                                                   \
  do {                                                                                                     \
    if (CATRE_TRACE_ON && trace && (threadIdx.x & 63) == 0)                                                \
      trace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_readcyclecounter();      \
  } while (0)
  constexpr int TP2 = 2 * TP;
  u32x4* a3 = smem;
  u32x4* a2 = smem + TP2 * 64;
  u32x4* h1 = smem;                                               // [128][8]
  u32x4* tA = smem + TP2 * 8;                                     // [64 j][8]: T64 transposed, rows = out channel j
  u32x4* pf = smem + TP2 * 8 + TP * 8;                            // [128][8]
  float* scratch = reinterpret_cast<float*>(smem + 2 * TP2 * 8 + TP * 8);  // [8][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  TileInfo ti;
  int tile0;
  pair_info(blockIdx.x, B, N, M, ti, tile0);
  const bool has2 = ti.valid > TP;
  const bool ft = trans64 != nullptr;
  TRUNKB2_STAMP(0);

  const int ph = wave >> 2, w4 = wave & 3;  // point half / quarter-of-the-channels roles of the prologue
  GemmPipeB<1, 2, false, 8, 3> g2;          // conv2 64->128: wave -> m-block w4, the two point blocks of half ph
  g2.prefetch(wp2 + (w4 * 4) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, w4 * 32, lane);
  {
    const int p = ph * TP + lane;
    float x, y, z;
    load_point(P, ti, p, x, y, z);
    apply_t3(trans3 + ti.cloud * 9, x, y, z);
    if (SAVE && w4 == 0 && p < ti.valid) {  // x1 rows, zero-padded to 8 columns (train_ops.cloud_matmul out_cols=8)
      const size_t r = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0 + p;
      const f32x4 lo = {x, y, z, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(sv.s1 + r * 8) = lo;
      *reinterpret_cast<f32x4*>(sv.s1 + r * 8 + 4) = hi;
    }
    conv3_relu_chunks(x, y, z, Wc1, bc1, w4, (ft ? h1 : pf) + p * 8, bf_key<8>(p));
    if (ft) {  // A-operand image of the feature transform: row j holds T64[i][j] over i (pointnet.py:107-109);
               // wave (G = w4, s = ph) fills chunk 2G+s: i = 16G + {4s..4s+3, 8+4s..8+4s+3}
      const float* src = trans64 + (size_t)ti.cloud * 4096 + (w4 * 16 + 4 * ph) * 64 + lane;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = src[r * 64];
        v[4 + r] = src[(8 + r) * 64];
      }
      tA[lane * 8 + ((2 * w4 + ph) ^ bf_key<8>(lane))] = pack_bf8(v);
    }
  }
  __syncthreads();
  TRUNKB2_STAMP(1);
  if (ft) {
    {  // pointfeat[j][p] = sum_i T64[i][j] h1[i][p]: 2 m-blocks x 4 point blocks, one per wave
      const int mblk = wave >> 2, nb = wave & 3, key = bf_key<8>(n);
      f32x16 acc[1][1] = {{zero16()}};
      const u32x4* ar = tA + (mblk * 32 + n) * 8;
      const u32x4* br = h1 + (nb * 32 + n) * 8;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) acc[0][0] = mfma_bf(ar[(2 * kc + h) ^ key], br[(2 * kc + h) ^ key], acc[0][0]);
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 zb[1][4] = {{z4, z4, z4, z4}};
      store_tile_bf<1, 1, false, 8>(acc, pf + nb * 32 * 8, mblk, zb, lane);
    }
    __syncthreads();
  }
  TRUNKB2_STAMP(2);
  const size_t srow0 = (ti.is_obs ? (size_t)ti.obj * N : (size_t)B * N + (size_t)ti.obj * M) + ti.p0;
  if (SAVE) {  // h1 and pointfeat images are both complete here (the feature transform reads one and writes the other)
    if (ft) save_tile_rows_bf<64, 512, 2 * TP>(h1, sv.s2 + srow0 * 64, ti.valid, tid);
    save_tile_rows_bf<64, 512, 2 * TP>(pf, sv.s5 + srow0 * 64, ti.valid, tid);
  }
  // conv3 128->512: wave owns m-blocks [2*wave, +2) over all four point blocks; first weights + bias requested now
  GemmPipeB<2, 4, false, 16, 2, 1> g3;
  g3.prefetch(wp3 + ((wave * 2) * 8) * 64 + lane, 8 * 64);
  f32x4 bv3[2][4];
  load_bias_quads<2>(bv3, b3, wave * 64, lane);
  __builtin_amdgcn_sched_barrier(0);
  // pointfeat tile -> registers (stored to HBM after the last barrier) and its per-channel max over the pair
  const int pf_cc = tid & 7, pf_row = tid >> 3;  // 8 chunks x 64 rows, rows r and r+64
  const u32x4 pfc0 = pf[bf_off<8>(pf_row, pf_cc)], pfc1 = pf[bf_off<8>(pf_row + TP, pf_cc)];
  {
    float m[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m[2 * i] = fmaxf(bf_lo(pfc0[i]), bf_lo(pfc1[i]));
      m[2 * i + 1] = fmaxf(bf_hi(pfc0[i]), bf_hi(pfc1[i]));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      m[e] = fmaxf(m[e], __shfl_xor(m[e], 8));
      m[e] = fmaxf(m[e], __shfl_xor(m[e], 16));
      m[e] = fmaxf(m[e], __shfl_xor(m[e], 32));
    }
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) scratch[wave * 64 + bf_chunk_channel(lane, e)] = m[e];
    }
  }
  {
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g2.run(acc, pf + ph * TP * 8, lane);
    store_tile_bf<1, 2, true, 16>(acc, a2 + ph * TP * 16, w4, bv2, lane);
  }
  __syncthreads();
  float pf_max = 0.f;
  if (tid < 64) {
    pf_max = scratch[tid];
#pragma unroll
    for (int w = 1; w < 8; ++w) pf_max = fmaxf(pf_max, scratch[w * 64 + tid]);
  }
  __syncthreads();  // scratch / pf / h1 live inside a3, which conv3 overwrites next
  TRUNKB2_STAMP(3);
  if (SAVE) save_tile_rows_bf16<128, 512, 2 * TP>(a2, sv.s3 + srow0 * 64, ti.valid, tid);
  {  // HBM stores of the pair's pointfeat rows and their maxima: issued here, a whole conv3 sweep before the next barrier
     // (which waits for their acknowledge), so that their 8 + 1 registers are free during the sweeps
    const size_t prow0 = ti.is_obs ? (size_t)ti.obj * N + ti.p0 : (size_t)B * N + (size_t)ti.obj * M + ti.p0;
    if (!SAVE) {
      if (pf_row < ti.valid) pointfeat[(prow0 + pf_row) * 8 + pf_cc] = pfc0;
      if (pf_row + TP < ti.valid) pointfeat[(prow0 + pf_row + TP) * 8 + pf_cc] = pfc1;
    }
    if (tid < 64) {
      float* o1 = pm + (size_t)tile0 * PMW + 1024 + tid;
      o1[0] = pf_max;
      if (has2) o1[PMW] = pf_max;
    }
  }
  // conv4 512->1024 + max: wave owns m-blocks [4*wave, +4) in two passes of 2 x 4 point blocks
  GemmPipeB<2, 4, true, 64, SAVE ? 3 : CATRE_BF2_PFD, 1, true> g4a, g4b;  // SAVE: the arg-max epilogue needs the registers of one ring slot
#ifdef CATRE_DEBUG_TRACE
  g4a.ablate = g4b.ablate = __builtin_amdgcn_readfirstlane(g_ablate);
#endif
  {
    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = zero16();
    g3.run(acc, a2, lane);
    store_tile_bf<2, 4, true, 64>(acc, a3, wave * 2, bv3, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  g4a.prefetch(wp4 + ((wave * 4) * 32) * 64 + lane, 32 * 64);  // in flight across the barrier
  __builtin_amdgcn_sched_barrier(0);
  TRUNKB2_STAMP(4);
  __syncthreads();
  TRUNKB2_STAMP(5);
  if (SAVE) save_tile_rows_bf16<512, 512, 2 * TP>(a3, sv.s4 + srow0 * 256, ti.valid, tid);
  float* out = pm + (size_t)tile0 * PMW;
  float* out2 = has2 ? out + PMW : nullptr;
  // SAVE: (max, arg-max row) of the PAIR into both tiles' rows of the per-tile tables - k_maxpool_tiles keeps the first
  // maximum in tile order, and both rows name the same point
  float* pmax = SAVE ? sv.pmax + (size_t)tile0 * 1024 : nullptr;
  int* pidx = SAVE ? sv.pidx + (size_t)tile0 * 1024 : nullptr;
  {
    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = zero16();
    g4a.run(acc, a3, lane);
    if (!SAVE) g4b.prefetch(wp4 + ((wave * 4 + 2) * 32) * 64 + lane, 32 * 64);
    __builtin_amdgcn_sched_barrier(0);
    if (SAVE) {  // the arg-max epilogue needs the registers the next pass's first weights would hold
      argmax_pair_store<2, 4>(acc, pmax, pidx, has2, (wave * 4) * 32, b4, (int)srow0, lane);
      __builtin_amdgcn_sched_barrier(0);
      g4b.prefetch(wp4 + ((wave * 4 + 2) * 32) * 64 + lane, 32 * 64);
    } else {
      max_tile_store_pre2<2, 4>(acc, out, out2, (wave * 4) * 32, b4, lane);
    }
    TRUNKB2_STAMP(6);
  }
  {
    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = zero16();
    g4b.run(acc, a3, lane);
    if (SAVE)
      argmax_pair_store<2, 4>(acc, pmax, pidx, has2, (wave * 4 + 2) * 32, b4, (int)srow0, lane);
    else
      max_tile_store_pre2<2, 4>(acc, out, out2, (wave * 4 + 2) * 32, b4, lane);
  }
  TRUNKB2_STAMP(7);
#undef TRUNKB2_STAMP
}

// ------------------------------------------------------------------------------------------
// a9: rotation heads, bf16 operands: the structure of k_rot_l1 / k_rot_out (GN0 statistics: k_pf_moments_bf, catre_gram.h).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_pf_tile_bf(const u32x4* __restrict__ pointfeat, const RotTile& rt, u32x4* pf, int tid,
                                                int nthreads) {
  for (int i = tid; i < TP * 8; i += nthreads) {
    const int row = i >> 3, c = i & 7;
    const int srow = min(row, rt.valid - 1);
    pf[bf_off<8>(row, c)] = pointfeat[(rt.pf_off / 64 + srow) * 8 + c];
  }
}

// layer 0 recompute -> fused (bias + GN0) affine -> GELU -> bf16 image -> layer 1 -> y1 (bf16, HBM) + GN1 partials.
// 256 threads, 50 KiB LDS (40 of images + a 2.5 KiB transposition stage per wave), 3 workgroups per CU.
#ifndef ROTBF_PFD1
#define ROTBF_PFD1 3  // layer-1 weight K-steps in flight per wave
#endif
#define ROTBF_PITCH 80                       // bytes per stage row: 32 points x 2 bytes + 16 (conflict-free 8-byte writes)
#define ROTBF_STAGE (32 * ROTBF_PITCH / 16)  // u32x4 per wave: one 32-channel x 32-point block
__global__ __launch_bounds__(256, 2) void k_rot_l1_bf(const u32x4* __restrict__ pointfeat,
                                                      const u32x4* __restrict__ wpl0x, const u32x4* __restrict__ wpl0y,
                                                      const float* __restrict__ aff0 /*[B*2][2][2][256]*/,
                                                      const u32x4* __restrict__ wpl1x, const u32x4* __restrict__ wpl1y,
                                                      const float* __restrict__ b1x, const float* __restrict__ b1y,
                                                      unsigned short* __restrict__ y1, float* __restrict__ gn1, int B,
                                                      int N, int M, unsigned long long* __restrict__ trace = nullptr) {
  __shared__ u32x4 smem[TP * 8 + TP * 32 + 4 * ROTBF_STAGE];
  int stamp_i = 0;
#define ROTB_STAMP()                                                                                     \
  do {                                                                                                   \
    if (CATRE_TRACE_ON && trace && (threadIdx.x & 63) == 0)                                              \
      trace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + stamp_i] = __builtin_readcyclecounter(); \
    ++stamp_i;                                                                                           \
  } while (0)
  ROTB_STAMP();
  int abl = 0;  // instrumented build: knob 1 switches phases off (timing only, results wrong)
#ifdef CATRE_DEBUG_TRACE
  abl = __builtin_amdgcn_readfirstlane(g_ablate);
#endif
  u32x4* pf = smem;           // [64][64 ch]
  u32x4* a0 = smem + TP * 8;  // [64][256 ch]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave-private transposition stage of the y1 epilogue: [32 channels][32 points] bf16, rows of 80 bytes
  unsigned* stage = reinterpret_cast<unsigned*>(smem + TP * 8 + TP * 32 + wave * ROTBF_STAGE);
  const RotTile rt = rot_tile(blockIdx.x, B, N, M);
  const int T = (N + TP - 1) / TP + (M + TP - 1) / TP;
  load_pf_tile_bf(pointfeat, rt, pf, tid, 256);
  __syncthreads();
  ROTB_STAMP();
  const int n = lane & 31, h = lane >> 5;
#pragma unroll 1
  for (int hd = 0; hd < 2; ++hd) {
    GemmPipeB<2, 2, true, 32, ROTBF_PFD1> g1;
    {
      // wave -> channels [wave*64, +64) = m-blocks 2*wave, 2*wave+1
      const float* af = aff0 + ((((size_t)rt.obj * 2 + hd) * 2 + (rt.is_obs ? 0 : 1)) * 2) * 256 + wave * 64 + 4 * h;
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
      GemmPipeB<2, 2, false, 8, 2> g0;
      g0.prefetch((hd ? wpl0y : wpl0x) + (wave * 2 * 4) * 64 + lane, 4 * 64);
      if (!(abl & 64)) g0.run(acc, pf, lane);
      ROTB_STAMP();
      // layer 1's first weight fragments are requested now: their L2 round trip runs under the GELU epilogue and the barrier
      g1.prefetch((hd ? wpl1y : wpl1x) + (wave * 2 * 16) * 64 + lane, 16 * 64);
      const int key = bf_key<32>(n);
      float zprev[2][4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // step i = (mb, register quad g): channels mb*32 + 8g + 4h .. +3
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
          if (abl & 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] = fmaf(acc[mb][nb][4 * g + q], scr[i % 3][q], shr[i % 3][q]);
          } else {
            gelu_affine4_lp(acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1], acc[mb][nb][4 * g + 2], acc[mb][nb][4 * g + 3],
                            scr[i % 3], shr[i % 3], z);
          }
          if ((g & 1) == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) zprev[nb][q] = z[q];
          } else {  // quads g-1 and g complete chunk 4*(2*wave+mb) + 2*(g>>1) + h
            const float v[8] = {zprev[nb][0], zprev[nb][1], zprev[nb][2], zprev[nb][3], z[0], z[1], z[2], z[3]};
            const int chunk = 4 * (2 * wave + mb) + 2 * (g >> 1) + h;
            a0[(nb * 32 + n) * 32 + (chunk ^ key)] = pack_bf8(v);
          }
        }
      }
    }
    ROTB_STAMP();
    __syncthreads();
    ROTB_STAMP();
    {
      // layer 1 (256->256), "swapped": lane owns channel wave*64 + mb*32 + n and 32 of the tile's points
      // (the accumulators start at the lane's channel bias - one value per lane in this orientation)
      f32x16 acc[2][2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float bb = (hd ? b1y : b1x)[wave * 64 + mb * 32 + n];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][0][r] = acc[mb][1][r] = bb;
      }
      if (!(abl & 32)) g1.run(acc, a0, lane);
      ROTB_STAMP();
      const float inv_cnt = 1.0f / (8.f * (float)rt.valid);
      int valid_h = rt.valid - 4 * h;
      asm volatile("" : "+v"(valid_h));
      // y1 leaves in BLOCK order [object][head][tile][2 point halves][256 channels][32 points] (bf16): a wave's 32-channel x
      // 32-point accumulator block is 2 KiB of consecutive bytes, written as two 1-KiB store instructions (16 bytes per
      // lane, whole lines) after a transposition through 2.5 KiB of wave-private LDS.  Straight from the accumulators (lane =
      // channel, registers = points) it took 64 two-byte stores per lane and head.  Points >= rt.valid hold clamped
      // duplicates; k_rot_out_bf masks them.
      u32x4* ytile = reinterpret_cast<u32x4*>(y1) + (((size_t)rt.obj * 2 + hd) * T + rt.t) * (2 * 256 * 4);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int ch = wave * 64 + mb * 32 + n;
        if (!(abl & 16)) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            unsigned* srow = stage + n * (ROTBF_PITCH / 4) + 2 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const u32x2 d = {pack_bf2(acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1]),
                               pack_bf2(acc[mb][nb][4 * g + 2], acc[mb][nb][4 * g + 3])};
              *reinterpret_cast<u32x2*>(srow + 4 * g) = d;
            }
            const u32x4* sv = reinterpret_cast<const u32x4*>(stage);
            u32x4* dst = ytile + ((size_t)nb * 256 + wave * 64 + mb * 32) * 4;
            const u32x4 r0 = sv[(lane >> 2) * (ROTBF_PITCH / 16) + (lane & 3)];
            const u32x4 r1 = sv[(16 + (lane >> 2)) * (ROTBF_PITCH / 16) + (lane & 3)];
            __builtin_nontemporal_store(r0, dst + lane);
            __builtin_nontemporal_store(r1, dst + 64 + lane);
          }
        }
        if (abl & 8) continue;
        float s = 0.f, m2 = 0.f, q = 0.f;
        // full tile: sum and sum of squares in ONE pass (VALU time adds to MFMA time here), taken around an
        // anchor - one value of the group (its first lane's first point), so that M2 = sum d^2 - (sum d)^2 / n does not
        // cancel when the group's mean is large against its spread (the layer's bias sits in the accumulators)
        const float anchor = __shfl(acc[mb][0][0], n & ~7);
        if (rt.valid == TP) {  // (wave-uniform: no predication)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = acc[mb][nb][r] - anchor;
              s += d;
              q = fmaf(d, d, q);
            }
        } else {  // ragged tile: see k_rot_l1
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += nb * 32 + (r & 3) + 8 * (r >> 2) < valid_h ? acc[mb][nb][r] : 0.f;
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 32);
        const float mean = rt.valid == TP ? fmaf(s, inv_cnt, anchor) : s * inv_cnt;
        if (rt.valid == TP) {
          m2 = q;  // reduced over the group below like the ragged form's M2, then q_group - s_group^2 / n
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
        if (rt.valid == TP) m2 = fmaxf(m2 - s * (s * inv_cnt), 0.f);
        if ((lane & 7) == 0 && h == 0) {
          float* out = gn1 + (((size_t)rt.obj * 2 + hd) * T + rt.t) * 64 + (ch >> 3) * 2;
          out[0] = mean;
          out[1] = m2;
        }
      }
    }
    ROTB_STAMP();
    __syncthreads();  // a0 is rewritten for the second head
    ROTB_STAMP();
  }
#undef ROTB_STAMP
}

// GN1 -> GELU -> neck (256->3) -> conv_p weighted sum over the tile's points; reads the bf16 y1 in the block order
// k_rot_l1_bf writes ([2 point halves][256 channels][32 points] per tile and head).  A lane takes 8 consecutive points
// (16 bytes) of 4 channels in both halves: every load instruction of the workgroup is 4 KiB of consecutive bytes, the conv_p
// weights of its 16 points stay in registers, and the neck is applied to the per-channel sums (sum_p w_p gelu(z_cp))
// instead of to every point.
__global__ __launch_bounds__(256) void k_rot_out_bf(const unsigned short* __restrict__ y1,
                                                    const float* __restrict__ gn1stat, const float* __restrict__ gam1x,
                                                    const float* __restrict__ bet1x, const float* __restrict__ gam1y,
                                                    const float* __restrict__ bet1y, const float* __restrict__ neckx,
                                                    const float* __restrict__ necky, const float* __restrict__ wpx,
                                                    const float* __restrict__ wpy, float* __restrict__ rpart, int B,
                                                    int N, int M, int rd) {
  __shared__ float cst[5][256];  // per channel: GN1 scale, shift, the three neck weights
  __shared__ float red[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = (N + TP - 1) / TP + (M + TP - 1) / TP;
  const int hd = blockIdx.y;
  const RotTile rt = rot_tile(blockIdx.x, B, N, M);
  const u32x4* src = reinterpret_cast<const u32x4*>(y1) + (((size_t)rt.obj * 2 + hd) * T + rt.t) * (2 * 256 * 4);
  // The small (L2-resident) operands are requested BEFORE the tile: vmcnt retires in order, so behind the 32 KiB of HBM
  // loads their consumers - the constant table and the barrier - would wait for the whole tile first.
  const float* gam = hd ? gam1y : gam1x;
  const float* bet = hd ? bet1y : bet1x;
  const float* neck = hd ? necky : neckx;
  const float* st = gn1stat + ((size_t)rt.obj * 2 + hd) * 64 + (tid >> 3) * 2;
  const float mean = st[0], rstd = st[1], gm = gam[tid], bt = bet[tid];
  float nk[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) nk[c] = c < rd ? neck[c * 256 + tid] : 0.f;
  // conv_p weights of the lane's 2 x 8 points (zero past the ragged end: those slots hold clamped duplicates)
  const float* wp = (hd ? wpy : wpx) + rt.gp0;
  const int pbase = (tid & 3) * 8;
  float w[2][8];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int e = 0; e < 8; ++e) w[nb][e] = nb * 32 + pbase + e < rt.valid ? wp[nb * 32 + pbase + e] : 0.f;
  __builtin_amdgcn_sched_barrier(0);
  u32x4 rows[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) rows[i] = __builtin_nontemporal_load(src + i * 256 + tid);
  __builtin_amdgcn_sched_barrier(0);
  {
    const float sc = rstd * gm;
    cst[0][tid] = sc;
    cst[1][tid] = bt - mean * sc;
#pragma unroll
    for (int c = 0; c < 3; ++c) cst[2 + c][tid] = nk[c];
  }
  __syncthreads();
  float a3[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // channel (k * 64 + tid / 4): chunks i = k (points 0..31) and i = 4 + k (points 32..63)
    const int c = k * 64 + (tid >> 2);
    const float sc1 = cst[0][c], sh1 = cst[1][c];
    const f32x4 sc = {sc1, sc1, sc1, sc1}, sh = {sh1, sh1, sh1, sh1};
    float t = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const u32x4 u = rows[nb * 4 + k];
      float z[4];
      gelu_affine4_lp(bf_lo(u[0]), bf_hi(u[0]), bf_lo(u[1]), bf_hi(u[1]), sc, sh, z);
      t = fmaf(w[nb][0], z[0], t);
      t = fmaf(w[nb][1], z[1], t);
      t = fmaf(w[nb][2], z[2], t);
      t = fmaf(w[nb][3], z[3], t);
      gelu_affine4_lp(bf_lo(u[2]), bf_hi(u[2]), bf_lo(u[3]), bf_hi(u[3]), sc, sh, z);
      t = fmaf(w[nb][4], z[0], t);
      t = fmaf(w[nb][5], z[1], t);
      t = fmaf(w[nb][6], z[2], t);
      t = fmaf(w[nb][7], z[3], t);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) a3[q] = fmaf(cst[2 + q][c], t, a3[q]);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) a3[c] = wave_sum(a3[c]);
  if (lane == 0) {
    red[wave][0] = a3[0];
    red[wave][1] = a3[1];
    red[wave][2] = a3[2];
  }
  __syncthreads();
  if (tid < 3) {
    rpart[(((size_t)rt.obj * 2 + hd) * T + rt.t) * 4 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  }
}
