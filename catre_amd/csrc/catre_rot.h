// catre_rot.h - rotation-head kernels after the layer-0 statistics pass (included by catre_kernels.hip).
//
// a9 (heads/conv_out_per_rot_head.py:126-140) per (object, head), over the P = N+M concatenated points:
//   y0 = W0[:,1024:] pointfeat + bias0(cloud)       -> GN0 statistics from the moments of pointfeat (catre_gram.h,
//                                                      every compute mode)
//   a0 = gelu(GN0(y0));  y1 = W1 a0 + b1             -> y1 to HBM, GN1 partials (k_rot_l1)
//   k_gn_finalize
//   out[c] = sum_p w_p * (neck gelu(GN1(y1)))[c,p]                             (k_rot_out, HBM-bound)
//
// k_rot_l1 is sized for TWO workgroups per CU: 256 threads, 64 accumulator VGPRs per layer, and exactly
// 80 KiB of LDS (pointfeat tile 16 KiB + a0 image 64 KiB, both XOR-swizzled instead of padded).
#pragma once

// (mean, M2) partials [rows][T][32][2] -> (mean, rstd) [rows][32][2]; one workgroup per row (row = object*2+head): the
// row's partials are staged in LDS with coalesced loads, then 32 threads merge them in tile order (the merge is a serial
// chain per group - from LDS it costs ~1 us instead of T dependent L2 round trips).  (Tried: merging inside k_rot_out
// instead of a separate launch - every one of its 16 k workgroups then waits on the chain: 190 -> 297 us at B = 256.)
__global__ __launch_bounds__(256) void k_gn_finalize(const float* __restrict__ part, float* __restrict__ stat, int N,
                                                     int M) {
  extern __shared__ float sp[];  // [T][64]
  const int TN = (N + TP - 1) / TP, T = TN + (M + TP - 1) / TP;
  const float* src = part + (size_t)blockIdx.x * T * 64;
  for (int i = threadIdx.x; i < T * 64; i += 256) sp[i] = src[i];
  __syncthreads();
  const int g = threadIdx.x;
  if (g >= 32) return;
  float mean, rstd;
  merge_gn(sp, g, T, TN, N, M, mean, rstd);
  stat[((size_t)blockIdx.x * 32 + g) * 2] = mean;
  stat[((size_t)blockIdx.x * 32 + g) * 2 + 1] = rstd;
}

__device__ __forceinline__ void load_pf_tile_swz(const float* __restrict__ pointfeat, const RotTile& rt, float* pf,
                                                 int tid) {
  // 256 threads: 64 rows x 4 lanes x 4 float4 (16 chunks per row)
  const int row = tid >> 2, c0 = tid & 3;
  const int srow = min(row, rt.valid - 1);
  const f32x4* s = reinterpret_cast<const f32x4*>(pointfeat + rt.pf_off + (size_t)srow * 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + 4 * i;
    *reinterpret_cast<f32x4*>(pf + swz_off(row, c, 64)) = s[c];
  }
}

// Small grids (tiles * RS <= #CUs, like the encoder kernels): RS = 2 puts the two heads of a tile on two workgroups,
// RS = 4 additionally halves layer 1's output channels (each workgroup repeats layer 0 + GELU, 20 % of the MFMAs, and
// sweeps 128 of the 256 layer-1 channels: one m-block per wave).  Every output sees the same K order for any RS.
//
// SAVE (training forward, catre_train_rot_fwd; N and M multiples of 64): additionally stores what the backward kernels
// read - y0 = layer 0's output WITH its per-cloud bias (the GroupNorm-0 input) and a0 = gelu(GN0(y0)) - straight from the
// layer-0 epilogue's registers (layer 0 runs in the swapped orientation in this instance: whole-line stores, see the head
// loop), and lays y0 / a0 / y1 / the GN1 partials out HEAD-major
// ([2][B*P][256], [2][B*P/64][32][2]) so that each head's slice is the [rows,256] matrix the per-head backward ops take.
template <int RS, bool SAVE = false>
__global__ __launch_bounds__(256, 2) void k_rot_l1(const float* __restrict__ pointfeat, const f32x4* __restrict__ wpl0x,
                                                   const f32x4* __restrict__ wpl0y,
                                                   const float* __restrict__ aff0 /*[B*2][2][2][256]*/,
                                                   const f32x4* __restrict__ wpl1x, const f32x4* __restrict__ wpl1y,
                                                   const float* __restrict__ b1x, const float* __restrict__ b1y,
                                                   float* __restrict__ y1, float* __restrict__ gn1, int B, int N,
                                                   int M, unsigned long long* __restrict__ trace,
                                                   const float* __restrict__ bias0 = nullptr /*[2][2B][256]*/,
                                                   float* __restrict__ y0s = nullptr, float* __restrict__ a0s = nullptr) {
  static_assert(!SAVE || RS == 1, "the training instance takes both heads of a tile");
  __shared__ __attribute__((aligned(16))) float smem[TP * 64 + TP * 256];  // 80 KiB exactly
  int stamp_i = 0;
#define ROT_STAMP()                                                                                      \
  do {                                                                                                   \
    if (CATRE_TRACE_ON && trace && (threadIdx.x & 63) == 0)                                                                \
      trace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + stamp_i] = __builtin_readcyclecounter(); \
    ++stamp_i;                                                                                           \
  } while (0)
  float* pf = smem;            // [64][64]  swizzled
  float* a0 = smem + TP * 64;  // [64][256] swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int part = blockIdx.x % RS;
  const RotTile rt = rot_tile(blockIdx.x / RS, B, N, M);
  const int T = (N + TP - 1) / TP + (M + TP - 1) / TP;
  const int P = N + M;
  ROT_STAMP();
  load_pf_tile_swz(pointfeat, rt, pf, tid);
  __syncthreads();
  ROT_STAMP();
  const int n = lane & 31, h = lane >> 5;
  constexpr int MB1 = RS == 4 ? 1 : 2;                       // layer-1 m-blocks per wave
  const int mblk1 = RS == 4 ? (part >> 1) * 4 + wave : wave * 2;  // first of them
  const int hd_lo = RS == 1 ? 0 : (part & 1), hd_hi = RS == 1 ? 2 : hd_lo + 1;
#pragma unroll 1
  for (int hd = hd_lo; hd < hd_hi; ++hd) {
    if constexpr (SAVE) {
      // Training instance: layer 0 in the SWAPPED orientation (lane owns channel wave*64 + mb*32 + n and 32 of the tile's
      // points, like layer 1): y0 / a0 then leave as 4-byte stores whose 32 lanes cover 128 consecutive bytes of a row -
      // whole L2 lines.  (The normal orientation's 16-byte stores put 32 rows x 32 bytes into one instruction: the same
      // 3.2 GB took 0.9 ms of L2 request time, k_rot_l1_split<true> is bound by exactly that.)  The a0 image layer 1 reads
      // is the same; it is written with ds_write_b32 here (lanes along channels: conflict-free up to the two half-waves).
      const float* afb = aff0 + ((((size_t)rt.obj * 2 + hd) * 2 + (rt.is_obs ? 0 : 1)) * 2) * 256 + wave * 64 + n;
      const float* bqb = bias0 + ((size_t)hd * 2 * B + rt.cloud) * 256 + wave * 64 + n;
      float scv[2], shv[2], b0v[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        scv[mb] = afb[mb * 32];
        shv[mb] = afb[256 + mb * 32];
        b0v[mb] = bqb[mb * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc[2][2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) acc[mb][0] = acc[mb][1] = zero16();
      gemm_core<2, 2, true, true, 8, 2>(acc, (hd ? wpl0y : wpl0x) + (wave * 2 * 8) * 64 + lane, 8 * 64, pf, 64, lane);
      ROT_STAMP();
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
            const int row = nb * 32 + (r & 3) + 8 * (r >> 2);  // + 4 h: in rowg / below
            const float v = acc[mb][nb][r];
            const float z = gelu_erf(fmaf(v, scv[mb], shv[mb]));
            st_stream(yp + row * 256, v + b0v[mb]);
            st_stream(ap + row * 256, z);
            a0[swz_off(row + 4 * h, c >> 2, 256) + (c & 3)] = z;
          }
      }
    } else {
      // layer 0 recompute: wave -> channels [wave*64, +64), "normal" orientation.  The fused bias+GN affine
      // of this (object, head, cloud) is requested before the GEMM so the epilogue never waits on HBM/L2.
      const float* af = aff0 + ((((size_t)rt.obj * 2 + hd) * 2 + (rt.is_obs ? 0 : 1)) * 2) * 256 + wave * 64 + 4 * h;
      // epilogue step i = (mb, g) = (i >> 2, i & 3) needs the sc/sh quads of channels mb*32 + 8g + 4h ..+3; a ring of
      // three keeps two steps in flight (the first two are requested before the GEMM)
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
      gemm_core<2, 2, false, true, 8, 2>(acc, (hd ? wpl0y : wpl0x) + (wave * 2 * 8) * 64 + lane, 8 * 64, pf, 64, lane);
      ROT_STAMP();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int mb = i >> 2, g = i & 3;
        if (i + 2 < 8) {
          const int j = i + 2;
          scr[j % 3] = *reinterpret_cast<const f32x4*>(af + (j >> 2) * 32 + 8 * (j & 3));
          shr[j % 3] = *reinterpret_cast<const f32x4*>(af + 256 + (j >> 2) * 32 + 8 * (j & 3));
        }
        __builtin_amdgcn_sched_barrier(0);
        const int c = wave * 64 + mb * 32 + 8 * g + 4 * h;  // first of 4 consecutive channels
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          float zz[4];
          gelu_affine4(acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1], acc[mb][nb][4 * g + 2], acc[mb][nb][4 * g + 3],
                       scr[i % 3], shr[i % 3], zz);
          const f32x4 z = {zz[0], zz[1], zz[2], zz[3]};
          *reinterpret_cast<f32x4*>(a0 + swz_off(nb * 32 + n, c >> 2, 256)) = z;
        }
      }
    }
    ROT_STAMP();
    __syncthreads();
    ROT_STAMP();
    {
      // layer 1 (256->256), "swapped": lane owns channels wave*64 + mb*32 + n and 32 of the tile's points.  The
      // accumulators start at the lane's channel bias (one value per lane in this orientation): no bias add afterwards.
      f32x16 acc[MB1][2];
#pragma unroll
      for (int mb = 0; mb < MB1; ++mb) {
        const float bb = (hd ? b1y : b1x)[(mblk1 + mb) * 32 + n];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][0][r] = acc[mb][1][r] = bb;
      }
      gemm_core<MB1, 2, true, true, 32, 2>(acc, (hd ? wpl1y : wpl1x) + ((size_t)mblk1 * 32) * 64 + lane, 32 * 64, a0, 256,
                                          lane);
      ROT_STAMP();
      const float inv_cnt = 1.0f / (8.f * (float)rt.valid);
      int valid_h = rt.valid - 4 * h;  // point (r, nb) of this half-wave is real iff its in-tile index < valid_h
      asm volatile("" : "+v"(valid_h));
#pragma unroll
      for (int mb = 0; mb < MB1; ++mb) {
        const int ch = (mblk1 + mb) * 32 + n;
        float* dst = y1 + ((SAVE ? (size_t)hd * B + rt.obj : (size_t)rt.obj * 2 + hd) * P + rt.gp0) * 256 + ch;
        float s = 0.f;
        if (rt.valid == TP) {  // full tile (wave-uniform): no per-store predication
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
        // GN group = 8 consecutive channels = 8 consecutive lanes, both half-waves
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
    ROT_STAMP();
    __syncthreads();  // a0 is rewritten for the second head
    ROT_STAMP();
  }
#undef ROT_STAMP
}

// ------------------------------------------------------------------------------------------
// k_rot_l1 for grids that fill the chip: ONE wave per SIMD (256 threads, one workgroup per CU), every wave carries its OWN
// 64-point tile through both heads with an MB8 x NB2 layer-1 wave tile - all 256 output channels x 64 points = 256
// accumulators.  Nothing is shared between waves (no barrier), and a0 never touches LDS: layer 0 runs in the "normal"
// orientation, whose result registers - 4 consecutive channels of one point per quad - ARE the B fragments layer 1's
// "swapped" MFMAs want (catre_device.h: lane (n, h) supplies channels 8 kc + 4 h + s of point n), so GroupNorm-0 + GELU turn
// a quad of layer-0 accumulators straight into the operand of the next 64 layer-1 MFMAs.  Layer 0 is produced in quarters of
// 64 channels (64 accumulators), each consumed by the matching K = 64 slice of layer 1's sweep.
//   * K order of every output element as in k_rot_l1 (layer 0: chunks 0..7; layer 1: chunks 0..31 in order, accumulators
//     starting at the bias): same bits.
//   * the 256 layer-1 accumulators fill the AGPRs; hipcc selects the AGPR form for EVERY MFMA of a function that needs
//     AGPRs, so layer 0's 64 accumulators (read by the VALU right away) would evict layer-1 blocks to scratch.  Its MFMAs
//     are therefore inline asm with VGPR operands (accumulate-in-place chains four MFMAs apart; 20 wait states before the
//     VALU reads the result, CDNA3 ISA 4.5: a 16-pass XDL write followed by a VALU read needs 18).
//   * the GELU of step i + 1 is interleaved with the MFMAs of step i two VALU instructions per MFMA - the rate at which
//     VALU issue is free next to fp32 MFMAs - instead of standing alone in an epilogue.
// Why it was built: k_rot_l1 issues at 88 % but runs at ~2.13 GHz (two waves per SIMD, an LDS fragment per 2 MFMAs) - the
// power signature k_trunk showed before k_trunk4.  What it measured (profiles/r06_rotw_phases.txt, r06_ab_rotw*.txt): the
// clock does go up (2.34 GHz) and the results are the same bits, but the kernel is 8 % SLOWER (1523 vs 1409 us): with one
// wave per SIMD nothing overlaps - 164 k cycles of MFMA issue per (tile, head) + 54 k of GELU / statistics VALU, weight
// load issue and 256 dword stores per head, where k_rot_l1's second wave hides the store and load issue of the first
// (187 k per tile pair).  fp32 MFMAs and VALU share the SIMD's issue whichever wave they come from, so the GELU cannot be
// hidden either way; the trunk's win came from a sweep with no VALU in it.  Kept as an opt-in form (CATRE_ROTW=1,
// catre_form_switch 3) with its bit-equality test; the default stays k_rot_l1<1>.
// ------------------------------------------------------------------------------------------
#ifndef ROTW_MIN_TILES
#define ROTW_MIN_TILES 1024  // one full round: 4 tiles per workgroup x 256 CUs
#endif
#ifndef ROTW_VALU_PER_MFMA
#define ROTW_VALU_PER_MFMA 2
#endif
#define ROTW_SMEM (4 * TP * 64)  // floats: four pointfeat tiles = 64 KiB
__device__ __forceinline__ void mfma32_vg(f32x16& c, float a, float b) {  // VGPR form, accumulate in place
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_vg0(f32x16& c, float a, float b) {  // VGPR form, c = a b
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));  // (vDst must not overlap srcA / srcB)
}
__global__ __launch_bounds__(256) void k_rot_l1w(const float* __restrict__ pointfeat, const f32x4* __restrict__ wpl0x,
                                                 const f32x4* __restrict__ wpl0y, const float* __restrict__ aff0,
                                                 const f32x4* __restrict__ wpl1x, const f32x4* __restrict__ wpl1y,
                                                 const float* __restrict__ b1x, const float* __restrict__ b1y,
                                                 float* __restrict__ y1, float* __restrict__ gn1, int B, int N, int M,
                                                 unsigned long long* __restrict__ trace) {
  __shared__ __attribute__((aligned(16))) float smem[ROTW_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = (N + TP - 1) / TP + (M + TP - 1) / TP;
  const int P = N + M;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= B * T) return;  // (wave-uniform; no barrier anywhere below)
  int stamp_i = 0;
#define ROTW_STAMP()                                                                                             \
  do {                                                                                                           \
    if (CATRE_TRACE_ON && trace && lane == 0) trace[((size_t)tile) * 32 + stamp_i] = __builtin_readcyclecounter(); \
    ++stamp_i;                                                                                                   \
  } while (0)
  ROTW_STAMP();
  const RotTile rt = rot_tile(tile, B, N, M);
  float* pf = smem + wave * (TP * 64);  // [64][64] swizzled, this wave's tile
  {  // the wave stages its own tile: row = lane, 16 chunks
    const int srow = min(lane, rt.valid - 1);
    const f32x4* s = reinterpret_cast<const f32x4*>(pointfeat + rt.pf_off + (size_t)srow * 64);
    f32x4 v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = s[c];
#pragma unroll
    for (int c = 0; c < 16; ++c) *reinterpret_cast<f32x4*>(pf + swz_off(lane, c, 64)) = v[c];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  ROTW_STAMP();
  const int n = lane & 31, h = lane >> 5, sw = lane & 15;
  // layer 0's B fragments: chunk 2 kc + h of rows n and 32 + n of the swizzled tile (GemmPipe::run's addressing)
  const float* xlow[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) xlow[j] = pf + n * 64 + (((2 * j + h) ^ sw) << 2);
#pragma unroll 1
  for (int hd = 0; hd < 2; ++hd) {
    const f32x4* wp0 = (hd ? wpl0y : wpl0x) + lane;
    const f32x4* wp1 = (hd ? wpl1y : wpl1x) + lane;
    const float* afh = aff0 + ((((size_t)rt.obj * 2 + hd) * 2 + (rt.is_obs ? 0 : 1)) * 2) * 256 + 4 * h;
    f32x16 acc1[8][2];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const float bb = (hd ? b1y : b1x)[mb * 32 + n];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[mb][0][r] = acc1[mb][1][r] = bb;
    }
    f32x4 a1[2][8];   // layer-1 weight chunks, ring of two (chunk kc in slot kc & 1): one step in flight
    f32x4 w0[3][2];   // layer-0 weight chunks, ring of three: two in flight
    f32x4 x0[2][2];   // layer-0 B fragments (LDS), ring of two
    f32x4 scr[3], shr[3];
    auto issue_a1 = [&](int kc) {
#pragma unroll
      for (int m = 0; m < 8; ++m) a1[kc & 1][m] = wp1[(m * 32 + kc) * 64];
    };
    auto issue_w0 = [&](int kq, int kc) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) w0[kc % 3][mb] = wp0[((kq * 2 + mb) * 8 + kc) * 64];
    };
    auto issue_x0 = [&](int kc) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) x0[kc & 1][nb] = *reinterpret_cast<const f32x4*>(xlow[kc] + nb * 32 * 64);
    };
    auto issue_aff = [&](int kq, int i) {  // step i = (mb, g): the sc / sh quads of channels 64 kq + 32 mb + 8 g + 4 h ..+3
      const float* af = afh + kq * 64 + (i >> 2) * 32 + 8 * (i & 3);
      scr[i % 3] = *reinterpret_cast<const f32x4*>(af);
      shr[i % 3] = *reinterpret_cast<const f32x4*>(af + 256);
    };
    issue_w0(0, 0);
    issue_w0(0, 1);
    issue_a1(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      // ---- layer 0, out channels [64 kq, +64): acc0[mb][nb], VGPR-form MFMAs
      issue_aff(kq, 0);
      issue_aff(kq, 1);
      issue_x0(0);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc0[2][2];
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        if (kc + 2 < 8) issue_w0(kq, kc + 2);
        if (kc + 1 < 8) issue_x0(kc + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
              if (kc == 0 && s == 0)
                mfma32_vg0(acc0[mb][nb], w0[kc % 3][mb][s], x0[kc & 1][nb][s]);
              else
                mfma32_vg(acc0[mb][nb], w0[kc % 3][mb][s], x0[kc & 1][nb][s]);
            }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next quarter's first layer-0 weight chunks: in flight under this quarter's layer-1 slice
      if (kq < 3) {
        issue_w0(kq + 1, 0);
        issue_w0(kq + 1, 1);
      }
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // XDL write -> VALU read of acc0
      __builtin_amdgcn_sched_barrier(0);
      ROTW_STAMP();
      // ---- GroupNorm-0 + GELU on one register quad per point block = layer 1's B fragment of chunk 8 kq + i, then the
      //      64 MFMAs of that chunk; the GELU of step i + 1 is issued between the MFMAs of step i
      f32x4 bq[2][2];  // ring of two steps x two point blocks
      auto make_b = [&](int i) {
        const int mb = i >> 2, g = i & 3;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          float zz[4];
          gelu_affine4(acc0[mb][nb][4 * g], acc0[mb][nb][4 * g + 1], acc0[mb][nb][4 * g + 2], acc0[mb][nb][4 * g + 3],
                       scr[i % 3], shr[i % 3], zz);
          bq[i & 1][nb] = f32x4{zz[0], zz[1], zz[2], zz[3]};
        }
      };
      make_b(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kc = kq * 8 + i;
        if (kc + 1 < 32) issue_a1(kc + 1);
        if (i + 2 < 8) issue_aff(kq, i + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < 8) make_b(i + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc1[m][nb] = mfma32(bq[i & 1][nb][s], a1[kc & 1][m][s], acc1[m][nb]);
        if (i + 1 < 8) {
          // one MFMA, then two VALU instructions, ... : the next step's GELU rides in the MFMAs' free issue slots
#pragma unroll
          for (int u = 0; u < 64; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, ROTW_VALU_PER_MFMA, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      ROTW_STAMP();
    }
    // ---- y1 + GroupNorm-1 partials: k_rot_l1's epilogue for all eight m-blocks
    const float inv_cnt = 1.0f / (8.f * (float)rt.valid);
    int valid_h = rt.valid - 4 * h;
    asm volatile("" : "+v"(valid_h));
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      // one m-block at a time: left alone the scheduler reads all 256 accumulators out of the AGPRs up front
      __builtin_amdgcn_sched_barrier(0);
      const int ch = mb * 32 + n;
      float* dst = y1 + (((size_t)rt.obj * 2 + hd) * P + rt.gp0) * 256 + ch;
      float* dh = dst + (size_t)(4 * h) * 256;
      float s = 0.f;
      if (rt.valid == TP) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc1[mb][nb][r];
            st_stream(dh + (nb * 32 + (r & 3) + 8 * (r >> 2)) * 256, v);
            s += v;
          }
      } else {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc1[mb][nb][r];
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
            const float d = acc1[mb][nb][r] - mean;
            m2 = fmaf(d, d, m2);
          }
      } else {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc1[mb][nb][r] - mean;
            m2 += nb * 32 + (r & 3) + 8 * (r >> 2) < valid_h ? d * d : 0.f;
          }
      }
      m2 += __shfl_xor(m2, 1);
      m2 += __shfl_xor(m2, 2);
      m2 += __shfl_xor(m2, 4);
      m2 += __shfl_xor(m2, 32);
      if ((lane & 7) == 0 && h == 0) {
        float* out = gn1 + (((size_t)rt.obj * 2 + hd) * T + rt.t) * 64 + (ch >> 3) * 2;
        out[0] = mean;
        out[1] = m2;
      }
    }
    ROTW_STAMP();
  }
#undef ROTW_STAMP
}

// GN1 -> GELU -> neck (256->3) -> conv_p weighted sum over the tile's points; HBM-bound read of y1.
// Body for tile bx, head hd.  stat_oh: the 32 (mean, rstd) pairs of this (object, head) - k_gn_finalize's output or an
// LDS copy (k_heads_d, catre_small.h).
__device__ __forceinline__ void rot_out_body(const float* __restrict__ y1, const float* stat_oh,
                                             const float* __restrict__ gam1x, const float* __restrict__ bet1x,
                                             const float* __restrict__ gam1y, const float* __restrict__ bet1y,
                                             const float* __restrict__ neckx, const float* __restrict__ necky,
                                             const float* __restrict__ wpx, const float* __restrict__ wpy,
                                             float* __restrict__ rpart /*[B][2][T][4]*/, int B, int N, int M,
                                             int rd /* neck rows: RotHead.rot_dim <= 3 */, const RotTile& rt, int hd,
                                             float (*red)[4]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = (N + TP - 1) / TP + (M + TP - 1) / TP, P = N + M;
  const int c0 = lane * 4;  // this lane's 4 channels
  const float* gam = hd ? gam1y : gam1x;
  const float* bet = hd ? bet1y : bet1x;
  const float* neck = hd ? necky : neckx;
  const float* wp = hd ? wpy : wpx;
  const float* st = stat_oh + (c0 >> 3) * 2;
  const float mean = st[0], rstd = st[1];
  f32x4 sc, sh;
  float nk[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * gam[c0 + q];
    sh[q] = bet[c0 + q] - mean * sc[q];
#pragma unroll
    for (int c = 0; c < 3; ++c) nk[c][q] = c < rd ? neck[c * 256 + c0 + q] : 0.f;
  }
  const float* src = y1 + (((size_t)rt.obj * 2 + hd) * P + rt.gp0) * 256 + c0;
  float a3[3] = {0.f, 0.f, 0.f};
#ifndef ROT_OUT_UNROLL
#define ROT_OUT_UNROLL 4
#endif
#pragma unroll ROT_OUT_UNROLL
  for (int p = wave; p < rt.valid; p += 4) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)p * 256));
    const float w = wp[rt.gp0 + p];
    float z[4];
    gelu_affine4(v[0], v[1], v[2], v[3], sc, sh, z);
    // The neck sums stay SCALAR instructions (the empty asm statements keep the SLP vectoriser from pairing components 0
    // and 1 into v_pk_fma_f32 with op_sel operands).  Paired, THIS kernel returned a wrong first component - the low half of
    // the packed accumulator - in ~3 % of its runs whenever kernels issuing bf16 MFMAs were co-resident on the CUs (refines
    // in split / bf16 mode on a second stream; never on one stream, never next to fp32 kernels: profiles/soak_victim.py,
    // soak_streams.py).  The packed ISA is hazard-clean as far as LLVM's gfx950 tables go and the cause was not established;
    // with scalar sums 0 of 2400 runs differ (the packed GELU above is not involved: it stays).
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(z[q]));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float t = nk[c][0] * z[0];
      t = fmaf(nk[c][1], z[1], t);
      t = fmaf(nk[c][2], z[2], t);
      t = fmaf(nk[c][3], z[3], t);
      asm volatile("" : "+v"(t));
      a3[c] = fmaf(w, t, a3[c]);
      asm volatile("" : "+v"(a3[c]));
    }
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

__global__ __launch_bounds__(256) void k_rot_out(const float* __restrict__ y1, const float* __restrict__ gn1stat,
                                                 const float* __restrict__ gam1x, const float* __restrict__ bet1x,
                                                 const float* __restrict__ gam1y, const float* __restrict__ bet1y,
                                                 const float* __restrict__ neckx, const float* __restrict__ necky,
                                                 const float* __restrict__ wpx, const float* __restrict__ wpy,
                                                 float* __restrict__ rpart /*[B][2][T][4]*/, int B, int N, int M,
                                                 int rd /* neck rows: RotHead.rot_dim <= 3 */) {
  __shared__ float red[4][4];
  const int hd = blockIdx.y;
  const RotTile rt = rot_tile(blockIdx.x, B, N, M);
  rot_out_body(y1, gn1stat + ((size_t)rt.obj * 2 + hd) * 64, gam1x, bet1x, gam1y, bet1y, neckx, necky, wpx, wpy, rpart, B,
               N, M, rd, rt, hd, red);
}
