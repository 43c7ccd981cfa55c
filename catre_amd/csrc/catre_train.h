// catre_train.h - building blocks of the TRAINING path (forward with saved activations + backward).
//
// Round-1 design: correctness and coverage first.  Training runs the layers UNFUSED on point-major
// activation matrices [rows = points of all clouds, channels] in HBM (288 GB makes the ~10 GB of saved
// activations at B=256 a non-issue); every op below has a hand-written forward and backward kernel and is
// chained by torch.autograd (catre_amd/train_ops.py).  The fused inference kernels are untouched.
//
//   gemm_rows : Y[R,J]  = act(X[R,K] W^T + b) (* mask)     MFMA, 64-row tiles, X chunk in LDS, packed W from L2
//               (also the data-gradient GEMM: dX = dY W with W packed transposed, mask = ReLU mask)
//   gemm_tn   : dW[J,K] = dY[R,J]^T X[R,K]                 MFMA, 128x128 tiles, rows split over workgroups,
//               deterministic two-stage reduction
//   the rest  : column sums, per-cloud bias, max-pool with argmax (+ sparse gather/scatter backward),
//               per-cloud 3x3 / 64x64 transforms, GroupNorm over points / rows, GELU, conv_p weighted sum,
//               rot6d + pose-update backward.
#pragma once

// ------------------------------------------------------------------------------------------------
// weight packing for gemm_rows: logical matrix Wl[J][K]; transpose=1 reads Wl[j][k] = src[k*ld + j]
// ------------------------------------------------------------------------------------------------
__global__ void k_op_pack(const float* __restrict__ src, int ld, int J, int K, int transpose, float* __restrict__ dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= J * K) return;
  const int s = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
  const int nkc = K / 8;
  const int kc = rest % nkc, mb = rest / nkc;
  const int row = mb * 32 + (lane & 31), col = kc * 8 + 4 * (lane >> 5) + s;
  dst[idx] = transpose ? src[(size_t)col * ld + row] : src[(size_t)row * ld + col];
}

// ------------------------------------------------------------------------------------------------
// gemm_rows: 512 threads, 64 rows per workgroup, K processed in chunks of 8*NKC staged in LDS (swizzled),
// wave w owns m-blocks {w + 8*i}, i < MB (J <= 256*MB).  Store form: "normal" orientation (4 consecutive channels/lane)
// in the fp32 and split kernels, "swapped" (lane = channel, whole-line stores) in the bf16-operand kernel.
// ------------------------------------------------------------------------------------------------
// optional per-cloud bias of the row GEMMs (rot-head layer 0: the global-feature half of the 1088 -> 256 conv is a bias
// that depends on the cloud a row belongs to): rows object-major [N observed | M prior] per object, N and M multiples
// of 64, bias [2B][J].  B == 0: plain bias [J].
struct CloudBias {
  int B, N, M;
  float* gn_part;  // optional: per-tile GroupNorm partials of the output [tiles][32][2] (J == 256, no ReLU / mask)
  int xcm;         // the rows of X are CLOUD-major (B*N observed rows, then B*M prior rows) while the output rows stay
                   // object-major: the autocast rotation heads read pointfeat where the trunk wrote it, no re-ordered copy
};
// object-major row r ([N observed | M prior] per object) -> the cloud-major row of the same point
__device__ __forceinline__ int cloud_major_row(int r, const CloudBias& cb) {
  const int P = cb.N + cb.M, obj = r / P, w = r - obj * P;
  return w < cb.N ? obj * cb.N + w : cb.B * cb.N + obj * cb.M + (w - cb.N);
}
__device__ __forceinline__ const float* cloud_bias(const float* bias, CloudBias cb, int r0, int J) {
  if (cb.B <= 0 || !bias) return bias;
  const int P = cb.N + cb.M, obj = r0 / P, within = r0 - obj * P;
  return bias + (size_t)(within < cb.N ? obj : cb.B + obj) * J;
}

// cross-lane move inside a 16-lane row on the VALU (DPP), no LDS round trip like ds_bpermute
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// epilogue shared by the fp32 and bf16-operand row GEMMs: bias / ReLU / output mask -> Y, or (MAXP) the per-tile
// max / arg-max over the 64 rows
template <int MB, bool MAXP, bool YB = false, bool MH = false>  // YB: Y holds bf16 rows (ldy in elements; full tiles only); MH: `mask` does (ldm in elements)
__device__ __forceinline__ void gemm_rows_epilogue(f32x16 (&acc)[MB][2], const float* __restrict__ bias,
                                                   const float* __restrict__ mask, int ldm, float* __restrict__ Y,
                                                   int ldy, int R, int relu, int r0, int nblk, int wave, int lane,
                                                   int blk_off = 0, float* __restrict__ gn_part = nullptr,
                                                   int mri = -1) {
  // mri >= 0 (all lanes): lane l holds the row of `mask` that output row r0 + l takes its mask from (row-compacted outputs
  // against a dense mask, looked up by the kernel before its sweep); -1: output row = mask row
  const int n = lane & 31, h = lane >> 5;
  if constexpr (MAXP) {
    // D[row = point][col = channel]: lane owns channel blk*32 + n and points (r&3) + 8(r>>2) + 4h + 32nb
    int* amax = reinterpret_cast<int*>(const_cast<float*>(mask));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int blk = blk_off + wave + 8 * mb;
      if (blk >= nblk) break;
      const int ch = blk * 32 + n;
      const float bv = bias ? bias[ch] : 0.f;
      float m = -INFINITY;
      int am = 0;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // increasing point order inside a half-wave: strict > keeps the first
          const int pt = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = acc[mb][nb][r] + bv;
          if (r0 + pt < R && v > m) {
            m = v;
            am = pt;
          }
        }
      const float mo = __shfl_xor(m, 32);
      const int ao = __shfl_xor(am, 32);
      if (mo > m || (mo == m && ao < am)) {
        m = mo;
        am = ao;
      }
      if (h == 0) {
        Y[(size_t)blockIdx.x * ldy + ch] = m;
        amax[(size_t)blockIdx.x * ldy + ch] = r0 + am;
      }
    }
    return;
  }
  // Store form - also in the "swapped" orientation (lane owns channel blk*32 + n and rows (r&3) + 8(r>>2) + 4h + 32nb): Y
  // leaves as 4-byte stores whose 32 lanes cover 128 consecutive bytes of a row, whole L2 lines.  (Until round 3 this was
  // the normal orientation with 16-byte stores, lanes along the rows: 32 rows x 32 bytes per instruction, four times the
  // L2 write requests - what bounded the [rows,256] GEMMs of the reduced-precision training modes.)
  const bool full = r0 + TP <= R;  // wave-uniform: no per-row predicates on full tiles
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int blk = blk_off + wave + 8 * mb;
    if (blk >= nblk) break;
    const int ch = blk * 32 + n;
    const float bv = bias ? bias[ch] : 0.f;
    float* yo = Y + (size_t)(r0 + 4 * h) * ldy + ch;
    const size_t mo0 = (mri >= 0 ? (size_t)0 : (size_t)(r0 + 4 * h) * ldm) + ch;  // element offset of the lane's first mask row
    const float* mo = mask ? mask + mo0 : nullptr;
    const short* moh = reinterpret_cast<const short*>(mask) + mo0;
    int lim = R - r0 - 4 * h;  // row (nb, r) of this half-wave exists iff its in-tile index < lim
    asm volatile("" : "+v"(lim));
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = nb * 32 + (r & 3) + 8 * (r >> 2);
        float t = acc[mb][nb][r] + bv;
        t = relu ? fmaxf(t, 0.f) : t;
        size_t mrow = (size_t)row;
        if (mri >= 0) mrow = (size_t)__shfl(mri, 4 * h + row);
        if (full || row < lim) {
          if constexpr (MH) {
            if (mo) t = moh[mrow * ldm] > 0 ? t : 0.f;  // (a bf16 is > 0 iff its bits, as int16, are)
          } else {
            if (mo) t = mo[mrow * ldm] > 0.f ? t : 0.f;
          }
          if constexpr (!YB) yo[(size_t)row * ldy] = t;
          // bf16 rows: the GroupNorm partials below are those of the ROUNDED values - the tensor that is normalised and that
          // the backward re-reads (torch.autocast's GroupNorm sees the Conv1d's bf16 output too)
          if constexpr (YB) t = bf_lo(pack_bf2(t, 0.f));
          s += t;
        } else {
          t = 0.f;
        }
        acc[mb][nb][r] = t;
      }
    if constexpr (YB) {
      // bf16 rows: lane pairs (channels ch, ch + 1) exchange one value per pair of registers (rows row, row + 1); the even
      // lane stores both channels of the first row, the odd lane of the second - one 4-byte store per two results
      unsigned* yb = reinterpret_cast<unsigned*>(Y) + ((size_t)(r0 + 4 * h + (n & 1)) * ldy + (ch & ~1)) / 2;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float t0 = acc[mb][nb][r], t1 = acc[mb][nb][r + 1];
          const float got = dpp_move<0xB1>((n & 1) ? t0 : t1);  // quad_perm [1,0,3,2]: the pair's other lane
          const unsigned w = (n & 1) ? pack_bf2(got, t1) : pack_bf2(t0, got);
          yb[(size_t)(nb * 32 + (r & 3) + 8 * (r >> 2)) * (ldy / 2)] = w;
        }
    }
    if (gn_part) {
      // GroupNorm(32, 256) partials of this 64-row tile for group blk*4 + (n>>3) (8 consecutive lanes, both half-waves):
      // (mean, M2), merged per object by k_gnp_stats_final in tile order
      const float cnt = 8.f * (float)min(TP, R - r0);
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 32);
      const float mean = s / cnt;
      float m2 = 0.f;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[mb][nb][r] - mean;
          m2 += (full || nb * 32 + (r & 3) + 8 * (r >> 2) < lim) ? d * d : 0.f;
        }
      m2 += __shfl_xor(m2, 1);
      m2 += __shfl_xor(m2, 2);
      m2 += __shfl_xor(m2, 4);
      m2 += __shfl_xor(m2, 32);
      if ((lane & 7) == 0 && h == 0) {
        float* o = gn_part + ((size_t)blockIdx.x * 32 + blk * 4 + (n >> 3)) * 2;
        o[0] = mean;
        o[1] = m2;
      }
    }
  }
}

// The store epilogue in the NORMAL orientation (lane owns a row and 4 consecutive channels per register quad: 16-byte
// stores) - for k_gemm_rows (fp32: the requests hide under the slower MFMAs) and k_gemm_rows_sp (its three-product sweep in
// the swapped orientation spills ~120 SGPRs and ran 15-60 % slower than the L2 requests it saved; one-box A/B, round 3).
template <int MB>
__device__ __forceinline__ void gemm_rows_epilogue_n(f32x16 (&acc)[MB][2], const float* __restrict__ bias,
                                                     const float* __restrict__ mask, int ldm, float* __restrict__ Y,
                                                     int ldy, int R, int relu, int r0, int nblk, int wave, int lane,
                                                     int blk_off = 0, float* __restrict__ gn_part = nullptr,
                                                     int mr0 = -1, int mr1 = -1) {
  const int n = lane & 31, h = lane >> 5;
  // mr0 / mr1 >= 0: the rows of `mask` that output rows r0 + n and r0 + 32 + n take their mask from (row-compacted outputs
  // against a dense mask; looked up by the kernel before its sweep)
  const size_t mr[2] = {mr0 >= 0 ? (size_t)mr0 : (size_t)(r0 + n), mr1 >= 0 ? (size_t)mr1 : (size_t)(r0 + 32 + n)};
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int blk = blk_off + wave + 8 * mb;
    if (blk >= nblk) break;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = blk * 32 + 8 * g + 4 * h;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ch);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int r = r0 + nb * 32 + n;
        if (r < R) {
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float t = acc[mb][nb][4 * g + q] + bv[q];
            v[q] = relu ? fmaxf(t, 0.f) : t;
          }
          if (mask) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(mask + mr[nb] * ldm + ch);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = m[q] > 0.f ? v[q] : 0.f;
          }
          *reinterpret_cast<f32x4*>(Y + (size_t)r * ldy + ch) = v;
        }
      }
      if (gn_part) {
        // GroupNorm(32, 256) partials of this 64-row tile for group blk*4 + g (its 8 channels = this register quad of
        // both half-waves): (mean, M2) about a shift, merged per object by k_gnp_stats_final in tile order
        const float shift = __shfl(acc[mb][0][4 * g] + bv[0], 0);
        float sd = 0.f, sq = 0.f;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          if (r0 + nb * 32 + n < R) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float d = (acc[mb][nb][4 * g + q] + bv[q]) - shift;
              sd += d;
              sq = fmaf(d, d, sq);
            }
          }
        sd = wave_sum(sd);
        sq = wave_sum(sq);
        if (lane == 0) {
          const float cnt = 8.f * (float)min(TP, R - r0);
          float* o = gn_part + ((size_t)blockIdx.x * 32 + blk * 4 + g) * 2;
          o[0] = shift + sd / cnt;
          o[1] = sq - sd * sd / cnt;
        }
      }
    }
  }
}

// MAXP: instead of storing Y, reduce the tile's 64 rows to a per-channel (max, arg-max row) pair - the forward of
// linear + max-pool over points without materialising the [R, J] matrix (Y = float maxima [tiles][J], mask = the int
// arg-max rows [tiles][J]).  Uses the "swapped" MFMA orientation so that a lane owns a channel and the reduction is
// in-register; the first maximum wins, like torch.max.
// KS = 2 (MB = 1, J <= 128: at most four m-blocks for eight waves): the waves that would only help staging take the second
// half of every K chunk of the same m-blocks, and the two partial tiles meet in LDS behind the last chunk (fixed order:
// first half + second half).  The launcher instantiates it for K = 128, J <= 128 only - the conv2-class dgrads of the
// row-sparse chains (53 -> 43 us each); for the conv3 dgrad (J = 128, K = 512) the split measured slower and is not used.
// Those K = 128 dgrads are therefore RE-ASSOCIATED against the single-pass form: (first half of every chunk) + (second half).
template <int MB, int NKC, bool MAXP = false, int KS = 1>
__global__ __launch_bounds__(512, MB == 1 ? 4 : 2) void k_gemm_rows(const float* __restrict__ X, int ldx, const f32x4* __restrict__ Wp,
                                                   const float* __restrict__ bias, const float* __restrict__ mask,
                                                   int ldm, float* __restrict__ Y, int ldy, int R, int J, int K,
                                                   int relu, const float* __restrict__ xmask, int ldxm, CloudBias cb,
                                                   const int* __restrict__ Rdev = nullptr,
                                                   const int* __restrict__ mrows = nullptr) {
  constexpr int KC = 8 * NKC;                       // floats per chunk (64, 128 or 256)
  constexpr int LDX = KC < 64 ? 64 : KC;            // swizzle needs a row pitch that is a multiple of 64 floats
  __shared__ __attribute__((aligned(16))) float xs[TP * LDX];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * TP;
  if (Rdev) {  // row count known only on the device (row-compacted operands, catre_op_rows_compact): tiles past it leave
    R = min(R, *Rdev);
    if (r0 >= R) return;
  }
  const int nblk = J / 32, nkc_total = K / 8, nchunks = K / KC;
  // mask rows of this lane's two output rows (row-compacted outputs against a dense mask): requested before the sweep
  int mr0 = -1, mr1 = -1;
  if (mrows) {
    mr0 = mrows[min(r0 + (lane & 31), R - 1)];
    mr1 = mrows[min(r0 + 32 + (lane & 31), R - 1)];
  }
  f32x16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb][0] = acc[mb][1] = zero16();
  static_assert(KS == 1 || (KS == 2 && MB == 1 && !MAXP && (NKC / KS) % 8 == 0), "K split: swizzled images shift by whole 8-chunk groups");
  const int blk_w = KS > 1 ? wave % nblk : wave, ks = KS > 1 ? wave / nblk : 0;
  const bool active = KS > 1 ? wave < KS * nblk : wave < nblk;  // waves beyond that only help staging
  for (int c = 0; c < nchunks; ++c) {
    if (c) __syncthreads();
    // stage X[r0..r0+64, c*KC .. +KC) : coalesced float4 along K, swizzled rows
    constexpr int F4 = KC / 4;
    constexpr int NU = (TP * F4 + 511) / 512;
    // all of this thread's loads are requested before the first LDS store: as a plain loop hipcc waits for every load
    // before issuing the next (up to 8 HBM round trips in a row per workgroup - the row GEMMs were latency-bound on it)
    // (the instances with 128 accumulator registers have no room for it and keep the plain loop)
    constexpr int G = NU > 4 ? 4 : NU;  // eight loads + eight masks at once do not fit 128 VGPRs next to the sweep
    // opaque zero, new in every chunk: keeps hipcc from hoisting the 2 x NU 64-bit row addresses out of the chunk loop
    // (32 VGPRs held across the MFMA sweep - the K-chunk-256 instance then needs 158 and only one workgroup fits a CU)
    int rz = 0;
    asm volatile("" : "+v"(rz));  // eight at once would push the K-chunk-256 instance over 128 VGPRs (one workgroup per CU)
    if constexpr (MB >= 4) {
      for (int i = tid; i < TP * F4; i += 512) {
        const int row = i / F4, ch = i % F4;
        const int gr = min(r0 + row, R - 1);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(X + (size_t)gr * ldx + c * KC + ch * 4);
        if (xmask) {
          const f32x4 m = *reinterpret_cast<const f32x4*>(xmask + (size_t)gr * ldxm + c * KC + ch * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) v1[q] = m[q] > 0.f ? v1[q] : 0.f;
        }
        *reinterpret_cast<f32x4*>(xs + swz_off(row, ch, LDX)) = v1;
      }
    } else
#pragma unroll
    for (int u0 = 0; u0 < NU; u0 += G) {
      f32x4 v[G], mk[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int i = tid + 512 * (u0 + u);
        if (u0 + u < NU && i < TP * F4) {
          const int row = i / F4, ch = i % F4;
          const int gr = min(r0 + row + rz, R - 1);
          v[u] = *reinterpret_cast<const f32x4*>(X + (size_t)gr * ldx + c * KC + ch * 4);
          if (xmask) mk[u] = *reinterpret_cast<const f32x4*>(xmask + (size_t)gr * ldxm + c * KC + ch * 4);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int i = tid + 512 * (u0 + u);
        if (u0 + u < NU && i < TP * F4) {
          const int row = i / F4, ch = i % F4;
          if (xmask) {  // ReLU backward folded into the operand load: X .* (xmask > 0)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[u][q] = mk[u][q] > 0.f ? v[u][q] : 0.f;
          }
          *reinterpret_cast<f32x4*>(xs + swz_off(row, ch, LDX)) = v[u];
        }
      }
    }
    __syncthreads();
    if (active) {
      // m-blocks of this wave: wave + 8*i; packed stride between them = 8 m-blocks
      // two weight chunks in flight, except the widest instance (4 m-blocks x K-chunk 256): 32 MFMAs per chunk cover
      // one chunk's L2 round trip, and the third ring slot would not fit 256 VGPRs (2 spills)
      GemmPipe<MB, 2, MAXP, true, NKC / KS, (NKC >= 4 && !(MB == 4 && NKC == 32) ? 2 : 1), 1> g;
      // the launcher guarantees J <= 256 (MB = 1, waves >= J/32 idle) or J % 256 == 0 (every wave owns MB blocks)
      g.prefetch(Wp + ((size_t)blk_w * nkc_total + c * NKC + ks * (NKC / KS)) * 64 + lane, 8 * nkc_total * 64);
      g.run(acc, xs + ks * (NKC / KS) * 8, LDX, lane);
    }
  }
  if constexpr (KS > 1) {
    __syncthreads();  // the last chunk's image is free: [m-block][point block][register][lane] partials of the second halves
    if (active && ks == 1) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) xs[((blk_w * 2 + nb) * 16 + r) * 64 + lane] = acc[0][nb][r];
    }
    __syncthreads();
    if (!active || ks != 0) return;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][nb][r] += xs[((blk_w * 2 + nb) * 16 + r) * 64 + lane];
  }
  if (!active) return;
  // (store form: normal orientation - at the fp32 MFMA rate the L2 write requests of its 16-byte stores are hidden, and
  // the swapped form measured 3-10 % slower on the training shapes; the bf16-operand kernel below is the one that needs it)
  if constexpr (MAXP)
    gemm_rows_epilogue<MB, true>(acc, cloud_bias(bias, cb, r0, J), mask, ldm, Y, ldy, R, relu, r0, nblk, wave, lane, 0,
                                 cb.gn_part);
  else
    gemm_rows_epilogue_n<MB>(acc, cloud_bias(bias, cb, r0, J), mask, ldm, Y, ldy, R, relu, r0, nblk, wave, lane, 0,
                             cb.gn_part, mr0, mr1);
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: dW[J,K] = sum_r dY[r,J]^T X[r,K] over rows [row_lo, row_hi) of this split.
// 512 threads, 128x128 output tile: wave w -> j-block w>>1 (of 4), k-blocks {2(w&1), 2(w&1)+1}.
// 32-row slabs are staged transposed in LDS ([col][row], row contiguous, +4 skew) so both MFMA fragments
// are 16-byte reads along the contraction (row) index.  Partials go to part[split][J*K].
// ------------------------------------------------------------------------------------------------
// ---- mixed precision (torch.autocast around the training forward, engine.py:304): the forward and dgrad row GEMMs
// with bf16 operands and fp32 accumulation / outputs.  Weights: bf16 fragments in plain k order; the 64-row X tile is
// converted while it is staged (whole K at once: 64 rows x 512 x 2 B = 64 KiB at most).
__global__ void k_op_pack_bf(const float* __restrict__ src, int ld, int J, int K, int transpose,
                             unsigned short* __restrict__ dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= J * K) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, rest = idx >> 9;
  const int nkc = K / 16;
  const int kc = rest % nkc, mb = rest / nkc;
  const int row = mb * 32 + (lane & 31), col = kc * 16 + 8 * (lane >> 5) + e;
  const float v = transpose ? src[(size_t)col * ld + row] : src[(size_t)row * ld + col];
  dst[idx] = __builtin_bit_cast(unsigned short, (__bf16)v);
}

// IO bit 0: X holds bf16 rows (ldx in elements, already the operand format: staged with one 16-byte copy per chunk);
// bit 1: Y holds bf16 rows (full tiles, no masks) - the all-bf16 activations of train_ops._RotHeadLP;
// bit 3: `mask` holds bf16 rows (ldm in elements) - the activation rows the autocast encoder forward saves;
// bit 2 (with bit 0, K = 256): X is the INPUT of a GroupNorm(32,256) + GELU whose output is this GEMM's operand - a chunk of 8
// channels is one GroupNorm group: the staging thread widens it, applies k_gnp_gelu_fwd's arithmetic with (mean, rstd) of
// its object from xf_stat [B][32][2], rounds to bf16 into LDS and stores the same 16 bytes to xf_out (bf16 rows: the
// activation the backward needs) - the GroupNorm + GELU pass between the two linears of a RotHead without its own launch
// and without re-reading what it wrote.
template <int MB, int CP, bool MAXP, int IO = 0>
__global__ __launch_bounds__(512) void k_gemm_rows_bf(const float* __restrict__ X, int ldx, const u32x4* __restrict__ Wp,
                                                      const float* __restrict__ bias, const float* __restrict__ mask,
                                                      int ldm, float* __restrict__ Y, int ldy, int R, int J, int relu,
                                                      const float* __restrict__ xmask, int ldxm, CloudBias cb,
                                                      const int* __restrict__ Rdev = nullptr,
                                                      const int* __restrict__ mrows = nullptr,
                                                      const float* __restrict__ xf_stat = nullptr,
                                                      const float* __restrict__ xf_gamma = nullptr,
                                                      const float* __restrict__ xf_beta = nullptr,
                                                      unsigned short* __restrict__ xf_out = nullptr, int xf_P = 1) {
  constexpr int NKC = CP / 2;  // K = 8 * CP
  __shared__ u32x4 xs[TP * CP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * TP;
  if (Rdev) {
    R = min(R, *Rdev);
    if (r0 >= R) return;
  }
  const int mri = mrows ? mrows[min(r0 + lane, R - 1)] : -1;  // mask row of output row r0 + lane (see gemm_rows_epilogue)
  const int nblk = J / 32;
  for (int i = tid; i < TP * CP; i += 512) {
    const int row = i / CP, ch = i % CP;
    const int gr = min(r0 + row, R - 1);
    if constexpr (IO & 1) {
      u32x4 v = __builtin_nontemporal_load(
          reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(X) + (size_t)gr * ldx) + ch);
      if constexpr (IO & 4) {
        static_assert(CP == 32, "the GroupNorm + GELU staging is for 256-channel rows");
        const float* st2 = xf_stat + ((size_t)(gr / xf_P) * 32 + ch) * 2;
        const float mean = st2[0], rstd = st2[1];
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(xf_gamma + ch * 8), g1 = *reinterpret_cast<const f32x4*>(xf_gamma + ch * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(xf_beta + ch * 8), b1 = *reinterpret_cast<const f32x4*>(xf_beta + ch * 8 + 4);
        const float y[8] = {bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1]), bf_lo(v[2]), bf_hi(v[2]), bf_lo(v[3]), bf_hi(v[3])};
        float a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float sc = rstd * (q < 4 ? g0[q & 3] : g1[q & 3]);
          a[q] = gelu_erf_lp(fmaf(y[q], sc, (q < 4 ? b0[q & 3] : b1[q & 3]) - mean * sc));  // k_gnp_gelu_fwd<true>'s operation sequence
        }
        v = pack_bf8(a);
        reinterpret_cast<u32x4*>(xf_out + (size_t)gr * 256)[ch] = v;
      }
      xs[bf_off<CP>(row, ch)] = v;
      continue;
    }
    const float* src = X + (size_t)(cb.xcm ? cloud_major_row(gr, cb) : gr) * ldx + ch * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    if (xmask) {
      const float* ms = xmask + (size_t)gr * ldxm + ch * 8;
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(ms), m1 = *reinterpret_cast<const f32x4*>(ms + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = m0[q] > 0.f ? v[q] : 0.f;
        v[4 + q] = m1[q] > 0.f ? v[4 + q] : 0.f;
      }
    }
    xs[bf_off<CP>(row, ch)] = pack_bf8(v);
  }
  __syncthreads();
  if (wave >= nblk) return;
  f32x16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb][0] = acc[mb][1] = zero16();
  GemmPipeB<MB, 2, true, CP, (NKC >= 4 ? 2 : 1), 1> g;  // swapped: lane = channel (gemm_rows_epilogue)
  g.prefetch(Wp + ((size_t)wave * NKC) * 64 + lane, 8 * NKC * 64);
  g.run(acc, xs, lane);
  gemm_rows_epilogue<MB, MAXP, (IO & 2) != 0, (IO & 8) != 0>(acc, cloud_bias(bias, cb, r0, J), mask, ldm, Y, ldy, R, relu, r0,
                                                             nblk, wave, lane, 0, cb.gn_part, MAXP ? -1 : mri);
}

// ---- split mode for the same row GEMMs: every operand hi + lo bf16, three products - fp32-grade results
// at the speed of the bf16 kernels (both are bound by the fp32 activations they stream).  Weights: hi pack, then lo pack
// (k_op_pack_split); the X tile is split while it is staged.  Wave w owns m-blocks {w + 8 i}; they are swept in passes
// of two (four accumulator blocks + both fragment rings would not fit 256 VGPRs).
__global__ void k_op_pack_split(const float* __restrict__ src, int ld, int J, int K, int transpose,
                                unsigned short* __restrict__ dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= J * K) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, rest = idx >> 9;
  const int nkc = K / 16;
  const int kc = rest % nkc, mb = rest / nkc;
  const int row = mb * 32 + (lane & 31), col = kc * 16 + 8 * (lane >> 5) + e;
  const float v = transpose ? src[(size_t)col * ld + row] : src[(size_t)row * ld + col];
  const __bf16 hi = (__bf16)v;
  dst[idx] = __builtin_bit_cast(unsigned short, hi);
  dst[(size_t)J * K + idx] = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)hi));
}

template <int MB, int CP, bool MAXP>
__global__ __launch_bounds__(512) void k_gemm_rows_sp(const float* __restrict__ X, int ldx, const u32x4* __restrict__ Wp,
                                                      const float* __restrict__ bias, const float* __restrict__ mask,
                                                      int ldm, float* __restrict__ Y, int ldy, int R, int J, int relu,
                                                      const float* __restrict__ xmask, int ldxm, CloudBias cb,
                                                      const int* __restrict__ Rdev = nullptr,
                                                      const int* __restrict__ mrows = nullptr) {
  constexpr int NKC = CP / 2;  // K = 8 * CP
  constexpr int PMB = MB >= 2 ? 2 : 1;
  __shared__ u32x4 xs[2][TP * CP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * TP;
  if (Rdev) {
    R = min(R, *Rdev);
    if (r0 >= R) return;
  }
  int mr0 = -1, mr1 = -1;  // mask rows of this lane's two output rows (gemm_rows_epilogue_n)
  if (mrows) {
    mr0 = mrows[min(r0 + (lane & 31), R - 1)];
    mr1 = mrows[min(r0 + 32 + (lane & 31), R - 1)];
  }
  const int nblk = J / 32;
  for (int i = tid; i < TP * CP; i += 512) {
    const int row = i / CP, ch = i % CP;
    const int gr = min(r0 + row, R - 1);
    const float* src = X + (size_t)gr * ldx + ch * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    if (xmask) {
      const float* ms = xmask + (size_t)gr * ldxm + ch * 8;
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(ms), m1 = *reinterpret_cast<const f32x4*>(ms + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = m0[q] > 0.f ? v[q] : 0.f;
        v[4 + q] = m1[q] > 0.f ? v[4 + q] : 0.f;
      }
    }
    u32x4 hi, lo;
    split_bf8(v, hi, lo);
    xs[0][bf_off<CP>(row, ch)] = hi;
    xs[1][bf_off<CP>(row, ch)] = lo;
  }
  __syncthreads();
  const int lo_off = J * CP;  // u32x4 units: J*K bf16 = J*K/8 chunks, K = 8 CP
  // the pass count is hidden from the optimiser: with a visible trip count of 1 (MB == 2) the sweep is merged with the
  // staging code above and the K = 512 variant spills 72 VGPRs; as a real loop it fits like the MB == 4 instances do
  int passes = MB / PMB;
  asm volatile("" : "+s"(passes));
#pragma unroll 1
  for (int p = 0; p < passes; ++p) {
    const int blk0 = 8 * PMB * p;
    if (blk0 + wave >= nblk) return;
    f32x16 acc[PMB][2];
#pragma unroll
    for (int mb = 0; mb < PMB; ++mb) acc[mb][0] = acc[mb][1] = zero16();
    GemmPipeS<PMB, 2, MAXP, CP, (NKC >= 4 ? 2 : 1)> g;
    g.prefetch(Wp + ((size_t)(blk0 + wave) * NKC) * 64 + lane, 8 * NKC * 64, lo_off);
    g.run(acc, xs[0], xs[1], lane);
    if constexpr (MAXP)
      gemm_rows_epilogue<PMB, true>(acc, cloud_bias(bias, cb, r0, J), mask, ldm, Y, ldy, R, relu, r0, nblk, wave, lane, blk0,
                                    cb.gn_part);
    else
      gemm_rows_epilogue_n<PMB>(acc, cloud_bias(bias, cb, r0, J), mask, ldm, Y, ldy, R, relu, r0, nblk, wave, lane, blk0,
                                cb.gn_part, mr0, mr1);
  }
}

#define TN_ROWS 64  // rows of dY / X staged per step
// dW partial [split][J][K] = dY[rows of the split, J]^T X[rows, K]; 128 x 128 output tile per workgroup, 8 waves x
// (32 x 64).  The contraction runs over ROWS, and v_mfma_f32_32x32x2_f32 takes one A and one B value per lane per
// instruction (A[i][k = lane>>5]), so the operands are staged ROW-MAJOR exactly as they sit in HBM (ds_write_b128,
// conflict-free) and read back with ds_read_b32 - consecutive lanes, consecutive columns.  (The first version
// staged them transposed with scalar stores: 16-way bank conflicts, 24 % of the fp32 MFMA rate.)  The next step's
// global loads are issued into registers before the current step's 64 MFMAs.
// KB = k-blocks of 32 per wave: 2 -> 128 x 128 tile; 1 -> 128 x 64 for K <= 64 (the 64-channel point features: on the
// wide tile half of every MFMA there is padding and the launch is bound by it; the narrow one also leaves LDS for a
// third workgroup per CU).
template <int KB>
__global__ __launch_bounds__(512) void k_gemm_tn(const float* __restrict__ dY, int ldy, const float* __restrict__ X,
                                                 int ldx, float* __restrict__ part, int J, int K, int R,
                                                 int rows_per_split, float* __restrict__ colpart,
                                                 const float* __restrict__ ymask, int ldym, size_t pitch,
                                                 const int* __restrict__ Rdev = nullptr,
                                                 const int* __restrict__ xrows = nullptr) {
  // xrows: row r of the contraction takes its X row from xrows[r] (dY row-compacted, X dense: no gathered copy of X)
  if (Rdev) {  // device-side row count: the splits share the rows that exist (empty splits write zero partials)
    R = min(R, *Rdev);
    rows_per_split = (int)((R + gridDim.z * TN_ROWS - 1) / (gridDim.z * TN_ROWS)) * TN_ROWS;
  }
  constexpr int KT = 64 * KB;      // tile width along K
  constexpr int XQ = KT / 4;       // float4 per staged X row
  constexpr int XU = TN_ROWS * XQ / 512;
  __shared__ __attribute__((aligned(16))) float ys[TN_ROWS * 128];
  __shared__ __attribute__((aligned(16))) float xs[TN_ROWS * KT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j0 = blockIdx.x * 128, k0 = blockIdx.y * KT;
  const int row_lo = blockIdx.z * rows_per_split, row_hi = min(R, row_lo + rows_per_split);
  const int jb = wave >> 1, kb0 = KB * (wave & 1);
  f32x16 acc[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) acc[kb] = zero16();
  const int i = lane & 31, h = lane >> 5;
  // bias gradient for free: the k-tile-0 workgroups also column-sum the dY tile they stage anyway
  // (thread = column tid&127, 16-row slice tid>>7), which saves a separate pass over dY
  const bool do_col = colpart != nullptr && blockIdx.y == 0;
  float csum = 0.f;
  // staging: float4 slot e = tid + 512*u  ->  dY: row e>>5, float4 column e&31 (u = 0..3); X: row e / XQ, column e % XQ
  f32x4 vy[4], vx[XU];
  int xi[XU];
#pragma unroll
  for (int u = 0; u < XU; ++u) xi[u] = 0;
  auto fetch = [&](int rs) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 512 * u, row = e >> 5, c4 = e & 31, gr = rs + row;
      vy[u] = z;
      const int jc = j0 + c4 * 4;
      if (gr < row_hi && jc < J) {
        vy[u] = *reinterpret_cast<const f32x4*>(dY + (size_t)gr * ldy + jc);
        if (ymask) {  // ReLU backward folded into the operand load: dY .* (ymask > 0)
          const f32x4 m = *reinterpret_cast<const f32x4*>(ymask + (size_t)gr * ldym + jc);
#pragma unroll
          for (int q = 0; q < 4; ++q) vy[u][q] = m[q] > 0.f ? vy[u][q] : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int e = tid + 512 * u, row = e / XQ, c4 = e % XQ, gr = rs + row;
      vx[u] = z;
      const int kc = k0 + c4 * 4;
      if (gr < row_hi && kc < K) {
        const size_t xr = xrows ? (size_t)xi[u] : (size_t)gr;
        vx[u] = *reinterpret_cast<const f32x4*>(X + xr * ldx + kc);
      }
    }
  };
  // row indices of the step AFTER the one being fetched: requested a whole MFMA phase before the X loads that need them
  // (loaded inside fetch, every step would wait one L2 round trip for them before it could issue its X loads)
  auto fetch_idx = [&](int rs) {
    if (!xrows) return;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int gr = rs + (tid + 512 * u) / XQ;
      xi[u] = gr < row_hi ? xrows[gr] : 0;
    }
  };
  fetch_idx(row_lo);
  fetch(row_lo);
  fetch_idx(row_lo + TN_ROWS);
  for (int rs = row_lo; rs < row_hi; rs += TN_ROWS) {
    __syncthreads();  // the previous step's reads are done
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(ys + (tid + 512 * u) * 4) = vy[u];
#pragma unroll
    for (int u = 0; u < XU; ++u) *reinterpret_cast<f32x4*>(xs + (tid + 512 * u) * 4) = vx[u];
    __syncthreads();
    if (rs + TN_ROWS < row_hi) {
      fetch(rs + TN_ROWS);  // in flight during the MFMAs below
      fetch_idx(rs + 2 * TN_ROWS);
    }
    if (do_col) {
      const float* yc = ys + (tid >> 7) * 16 * 128 + (tid & 127);
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += yc[r * 128];
      csum += t;
    }
    const float* ya = ys + h * 128 + jb * 32 + i;
    const float* xb = xs + h * KT + kb0 * 32 + i;
    // MFMA step t contracts rows 2t (h = 0) and 2t + 1 (h = 1); operands two steps ahead of their use, pinned
    float an[3], xn[KB][3];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      an[d] = ya[d * 256];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) xn[kb][d] = xb[d * 2 * KT + kb * 32];
    }
#pragma unroll
    for (int t = 0; t < TN_ROWS / 2; ++t) {
      if (t + 2 < TN_ROWS / 2) {
        an[(t + 2) % 3] = ya[(t + 2) * 256];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) xn[kb][(t + 2) % 3] = xb[(t + 2) * 2 * KT + kb * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) acc[kb] = mfma32(an[t % 3], xn[kb][t % 3], acc[kb]);  // D[j][k]
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (do_col) {
    __syncthreads();
    ys[tid] = csum;  // [4 row slices][128 columns]
    __syncthreads();
    const int j = j0 + tid;
    if (tid < 128 && j < J)
      colpart[(size_t)blockIdx.z * pitch + j] = (ys[tid] + ys[128 + tid]) + (ys[256 + tid] + ys[384 + tid]);
  }
  // D[row = j][col = k]: lane holds col k = lane&31, rows (reg&3)+8(reg>>2)+4h
  float* out = part + (size_t)blockIdx.z * pitch;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int k = k0 + (kb0 + kb) * 32 + i;
    if (k >= K) continue;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int j = j0 + jb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (j < J) out[(size_t)j * K + k] = acc[kb][reg];
    }
  }
}

// The same partials for a SKINNY right operand (K <= 8: the layers fed by 3-d points - conv1 of the STN and of the
// trunk).  On the 128 x 128 MFMA tile above 97 % of the products are padding and the launch is bound by them (0.23 ms
// for a 64 x 3 gradient whose operands stream in 0.02 ms); here a thread owns four dY columns, 256 / (J/4) row lanes
// share the rows, the X row is a uniform 16/32-byte load, and four rows per lane are requested together.
template <int KS>
__global__ __launch_bounds__(256) void k_gemm_tn_skinny(const float* __restrict__ dY, int ldy, const float* __restrict__ X,
                                                        int ldx, float* __restrict__ part, int J, int K, int R,
                                                        int rows_per_split, float* __restrict__ colpart,
                                                        const float* __restrict__ ymask, int ldym, size_t pitch,
                                                        const int* __restrict__ Rdev = nullptr,
                                                        const int* __restrict__ xrows = nullptr) {
  if (Rdev) {
    R = min(R, *Rdev);
    rows_per_split = (int)((R + gridDim.x * 32 - 1) / (gridDim.x * 32)) * 32;
  }
  extern __shared__ float red[];  // [row lane][column quad][4 * KS products + 4 column sums]
  constexpr int E = 4 * KS + 4;
  const int JQ = J >> 2, RL = 256 / JQ;
  const int tid = threadIdx.x, q = tid % JQ, rl = tid / JQ;
  const int lo = blockIdx.x * rows_per_split, hi = min(R, lo + rows_per_split);
  float acc[4][KS], cs[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    cs[c] = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) acc[c][k] = 0.f;
  }
  if (rl < RL) {
    for (int r = lo + rl; r < hi; r += 4 * RL) {
      f32x4 d[4], m[4], xa[4], xb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = min(r + u * RL, hi - 1);
        d[u] = *reinterpret_cast<const f32x4*>(dY + (size_t)rr * ldy + 4 * q);
        if (ymask) m[u] = *reinterpret_cast<const f32x4*>(ymask + (size_t)rr * ldym + 4 * q);
        const size_t xr = xrows ? (size_t)xrows[rr] : (size_t)rr;
        xa[u] = *reinterpret_cast<const f32x4*>(X + xr * ldx);
        if (KS == 8) xb[u] = *reinterpret_cast<const f32x4*>(X + xr * ldx + 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r + u * RL < hi) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float dv = (ymask && !(m[u][c] > 0.f)) ? 0.f : d[u][c];
            cs[c] += dv;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[c][k] = fmaf(dv, xa[u][k], acc[c][k]);
            if (KS == 8) {
#pragma unroll
              for (int k = 0; k < 4; ++k) acc[c][4 + k] = fmaf(dv, xb[u][k], acc[c][4 + k]);
            }
          }
        }
      }
    }
    float* o = red + (size_t)(rl * JQ + q) * E;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int k = 0; k < KS; ++k) o[c * KS + k] = acc[c][k];
      o[4 * KS + c] = cs[c];
    }
  }
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * pitch;
  for (int e = tid; e < JQ * E; e += 256) {
    float sum = 0.f;
    for (int l = 0; l < RL; ++l) sum += red[(size_t)l * JQ * E + e];  // fixed order
    const int qq = e / E, i = e % E;
    if (i < 4 * KS) {
      const int c = i / KS, k = i % KS;
      if (k < K) out[(size_t)(4 * qq + c) * K + k] = sum;
    } else if (colpart) {
      colpart[(size_t)blockIdx.x * pitch + 4 * qq + (i - 4 * KS)] = sum;
    }
  }
}

// Whole backward of a 64-channel layer fed by 3-d points (conv1 of the trunk: x1 [R,8] -> h1 [R,64] + ReLU) in ONE pass
// over the rows: the output gradient arrives as up to two tensors (h1 feeds the STNkd stack and the feature transform; the
// node adds them here instead of autograd), the ReLU mask is applied on load, and the same registers feed
//   dW[j][k] = sum_r dv[r][j] X[r][k],  db[j] = sum_r dv[r][j]   (partials per workgroup, merged like k_gemm_tn_skinny's)
//   dX[r][k] = sum_j dv[r][j] W[j][k]   (16 lanes of a row hold 4 channels each: DPP reduction over the 16-lane row)
// instead of add + relu-backward + a K=64 row GEMM with 3 live outputs + pad + skinny wgrad (0.27 ms -> 0.07 ms at
// R = 524 k; the op is bound by its 3 x 134 MB of operand reads).  J = 64 only (one 16-lane group per row), K <= 4.
__global__ __launch_bounds__(256) void k_skinny_bwd(const float* __restrict__ dY, int ldy, const float* __restrict__ dY2,
                                                    int ldy2, const float* __restrict__ ymask, int ldym,
                                                    const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                    int ldw, int Kw, float* __restrict__ part, float* __restrict__ colpart,
                                                    size_t pitch, float* __restrict__ dX, int lddx, int dxcols, int R,
                                                    int rows_per_split) {
  constexpr int KS = 4, JQ = 16, RL = 16, E = 4 * KS + 4;
  __shared__ float red[RL * JQ * E];  // [row lane][column quad][4 * KS products + 4 column sums]
  const int tid = threadIdx.x, q = tid & 15, rl = tid >> 4;
  const int lo = blockIdx.x * rows_per_split, hi = min(R, lo + rows_per_split);
  float acc[4][KS], cs[4], wv[4][KS];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    cs[c] = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      acc[c][k] = 0.f;
      wv[c][k] = (W && k < Kw) ? W[(size_t)(4 * q + c) * ldw + k] : 0.f;
    }
  }
  for (int r = lo + rl; r < hi; r += 4 * RL) {
    f32x4 d[4], d2[4], m[4], xa[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = min(r + u * RL, hi - 1);
      d[u] = *reinterpret_cast<const f32x4*>(dY + (size_t)rr * ldy + 4 * q);
      if (dY2) d2[u] = *reinterpret_cast<const f32x4*>(dY2 + (size_t)rr * ldy2 + 4 * q);
      if (ymask) m[u] = *reinterpret_cast<const f32x4*>(ymask + (size_t)rr * ldym + 4 * q);
      xa[u] = *reinterpret_cast<const f32x4*>(X + (size_t)rr * ldx);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool live = r + u * RL < hi;  // uniform over the 16 lanes of a row
      float dx[KS];
#pragma unroll
      for (int k = 0; k < KS; ++k) dx[k] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float dv = dY2 ? d[u][c] + d2[u][c] : d[u][c];
        if ((ymask && !(m[u][c] > 0.f)) || !live) dv = 0.f;
        cs[c] += dv;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          acc[c][k] = fmaf(dv, xa[u][k], acc[c][k]);
          dx[k] = fmaf(dv, wv[c][k], dx[k]);
        }
      }
      if (dX) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          dx[k] += dpp_move<0xB1>(dx[k]);   // quad_perm [1,0,3,2]
          dx[k] += dpp_move<0x4E>(dx[k]);   // quad_perm [2,3,0,1]: every lane of a quad holds the quad's sum
          dx[k] += dpp_move<0x124>(dx[k]);  // row_ror:4
          dx[k] += dpp_move<0x128>(dx[k]);  // row_ror:8: all four quads of the 16-lane row
        }
        if (live && q == 0) {
          float* o = dX + (size_t)(r + u * RL) * lddx;
          const f32x4 v = {dx[0], dx[1], dx[2], dx[3]}, z = {0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(o) = v;
          if (dxcols > 4) *reinterpret_cast<f32x4*>(o + 4) = z;
        }
      }
    }
  }
  float* o = red + (size_t)(rl * JQ + q) * E;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int k = 0; k < KS; ++k) o[c * KS + k] = acc[c][k];
    o[4 * KS + c] = cs[c];
  }
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * pitch;
  for (int e = tid; e < JQ * E; e += 256) {
    float sum = 0.f;
    for (int l = 0; l < RL; ++l) sum += red[(size_t)l * JQ * E + e];  // fixed order
    const int qq = e / E, i = e % E;
    if (i < 4 * KS)
      out[(size_t)(4 * qq + i / KS) * KS + (i % KS)] = sum;
    else if (colpart)
      colpart[(size_t)blockIdx.x * pitch + 4 * qq + (i - 4 * KS)] = sum;
  }
}

// (Tried in round 2: 256 x 256 / 256 x 128 output tiles per workgroup for the big layers - each operand read once instead
// of twice, half the LDS reads per MFMA: 0.72 ms for the 256 x 256 x 524 k weight gradient either way, no gain.)
// ------------------------------------------------------------------------------------------------
// k_gemm_tn on the bf16 matrix pipe.  SPLIT = false: bf16 operands (torch.autocast runs the backward of a bf16 layer
// in bf16 as well); SPLIT = true: every operand as hi + lo bf16 and three products (fp32-grade gradients).
// 256 threads, 128 x 128 output tile, wave -> 64 x 64 (2 x 2 blocks: one 16-byte LDS read per MFMA).
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive contraction indices (= rows) per lane, so the 64-row slabs are
// transposed while they are staged: a thread loads 8 rows x 4 columns (row-major, coalesced), converts, and writes
// four 16-byte chunks "column c, rows 8g..8g+7".  Slot of (column, chunk) in the image - both the fragment reads (32
// consecutive columns, one chunk) and the staging writes (columns 4 apart, one chunk) are bank-conflict free:
//   slot = (c >> 1) * 16 + 8 * ((c ^ (c >> 2)) & 1) + ((chunk ^ (c >> 1) ^ (c >> 4)) & 7)
// The bias gradient (column sums of dY) is taken from the fp32 values before they are rounded.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_slot(int c, int chunk) {
  return (c >> 1) * 16 + 8 * ((c ^ (c >> 2)) & 1) + ((chunk ^ (c >> 1) ^ (c >> 4)) & 7);
}

template <bool SPLIT, bool XH = false>  // XH: X holds bf16 rows (ldx in elements) - the saved activations of the autocast encoder forward
__global__ __launch_bounds__(256, 2) void k_gemm_tn_lp(const float* __restrict__ dY, int ldy, const float* __restrict__ X,
                                                       int ldx, float* __restrict__ part, int J, int K, int R,
                                                       int rows_per_split, float* __restrict__ colpart,
                                                       const float* __restrict__ ymask, int ldym, size_t pitch,
                                                       const int* __restrict__ Rdev = nullptr,
                                                       const int* __restrict__ xrows = nullptr) {
  if (Rdev) {
    R = min(R, *Rdev);
    rows_per_split = (int)((R + gridDim.z * 64 - 1) / (gridDim.z * 64)) * 64;
  }
  constexpr int NI = SPLIT ? 2 : 1;
  __shared__ u32x4 ys[NI][128 * 8];  // [column][8 chunks of 8 rows] bf16, 16 KiB per image
  __shared__ u32x4 xs[NI][128 * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
  const int row_lo = blockIdx.z * rows_per_split, row_hi = min(R, row_lo + rows_per_split);
  const int jb0 = 2 * (wave >> 1), kb0 = 2 * (wave & 1);
  const int i = lane & 31, h = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a) acc[a][0] = acc[a][1] = zero16();
  const bool do_col = colpart != nullptr && blockIdx.y == 0;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  // staging: thread -> column quad c4 (columns 4 c4 .. +3) and row group g (rows 8g .. 8g+7 of the slab)
  const int c4 = tid & 31, g = tid >> 5;
  const int jc = j0 + c4 * 4, kc = k0 + c4 * 4;
  f32x4 vy[8], vx[8];
  int xi[8];  // xrows: the X rows of the NEXT fetch (k_gemm_tn: requested a step ahead of the loads that need them)
#pragma unroll
  for (int u = 0; u < 8; ++u) xi[u] = 0;
  auto fetch_idx = [&](int rs) {
    if (!xrows) return;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int gr = rs + g * 8 + u;
      xi[u] = gr < row_hi ? xrows[gr] : 0;
    }
  };
  auto fetch = [&](int rs) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int gr = rs + g * 8 + u;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      vy[u] = vx[u] = z;
      if (gr < row_hi) {
        if (jc < J) {
          vy[u] = *reinterpret_cast<const f32x4*>(dY + (size_t)gr * ldy + jc);
          if (ymask) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(ymask + (size_t)gr * ldym + jc);
#pragma unroll
            for (int q = 0; q < 4; ++q) vy[u][q] = m[q] > 0.f ? vy[u][q] : 0.f;
          }
        }
        if (kc < K) {
          const size_t xo = (xrows ? (size_t)xi[u] : (size_t)gr) * ldx + kc;
          if constexpr (XH) {
            // the RAW 8 bytes (four bf16) ride in the first two words until stage(): a conversion here would wait for the
            // load, and these loads are meant to fly during the MFMAs of the slab before
            const u32x2 r = reinterpret_cast<const u32x2*>(X)[xo >> 2];
            vx[u][0] = __uint_as_float(r[0]);
            vx[u][1] = __uint_as_float(r[1]);
          } else {
            vx[u] = *reinterpret_cast<const f32x4*>(X + xo);
          }
        }
      }
    }
  };
  auto stage = [&](const f32x4 (&v)[8], u32x4 (&img)[NI][128 * 8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float col[8] = {v[0][q], v[1][q], v[2][q], v[3][q], v[4][q], v[5][q], v[6][q], v[7][q]};
      const int slot = tn_slot(c4 * 4 + q, g);
      if (SPLIT) {
        u32x4 hi, lo;
        split_bf8(col, hi, lo);
        img[0][slot] = hi;
        img[NI - 1][slot] = lo;
      } else {
        img[0][slot] = pack_bf8(col);
      }
    }
  };
  fetch_idx(row_lo);
  fetch(row_lo);
  fetch_idx(row_lo + TN_ROWS);
  for (int rs = row_lo; rs < row_hi; rs += TN_ROWS) {
    __syncthreads();  // the previous slab's fragment reads are done
    if (do_col) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) t += vy[u][q];
        csum[q] += t;
      }
    }
    stage(vy, ys);
    if constexpr (XH) {
      // bf16 rows are the operand already: the eight rows' halves of channel q, paired as pack_bf8 pairs them
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x4 w;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const unsigned a = __float_as_uint(vx[2 * pr][q >> 1]), b = __float_as_uint(vx[2 * pr + 1][q >> 1]);
          w[pr] = (q & 1) ? (a >> 16) | (b & 0xffff0000u) : (a & 0xffffu) | (b << 16);
        }
        xs[0][tn_slot(c4 * 4 + q, g)] = w;
      }
    } else {
      stage(vx, xs);
    }
    __syncthreads();
    if (rs + TN_ROWS < row_hi) {
      fetch(rs + TN_ROWS);  // in flight during the MFMAs below
      fetch_idx(rs + 2 * TN_ROWS);
    }
#pragma unroll
    for (int ks = 0; ks < TN_ROWS / 16; ++ks) {
      u32x4 a[2], b[2], al[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int sa = tn_slot((jb0 + t) * 32 + i, 2 * ks + h), sb = tn_slot((kb0 + t) * 32 + i, 2 * ks + h);
        a[t] = ys[0][sa];
        b[t] = xs[0][sb];
        if (SPLIT) {
          al[t] = ys[NI - 1][sa];
          bl[t] = xs[NI - 1][sb];
        }
      }
#pragma unroll
      for (int ja = 0; ja < 2; ++ja)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          if (SPLIT) {
            acc[ja][kb] = mfma_bf(al[ja], b[kb], acc[ja][kb]);
            acc[ja][kb] = mfma_bf(a[ja], bl[kb], acc[ja][kb]);
          }
          acc[ja][kb] = mfma_bf(a[ja], b[kb], acc[ja][kb]);  // D[j][k]
        }
    }
  }
  if (do_col) {  // thread (c4, g) holds the sums of its 4 columns over its row groups: merge the 8 groups in order
    __syncthreads();
    float* red = reinterpret_cast<float*>(&ys[0][0]);  // [8 groups][128 columns]
#pragma unroll
    for (int q = 0; q < 4; ++q) red[g * 128 + c4 * 4 + q] = csum[q];
    __syncthreads();
    const int j = j0 + tid;
    if (tid < 128 && j < J) {
      float t = red[tid];
#pragma unroll
      for (int gg = 1; gg < 8; ++gg) t += red[gg * 128 + tid];
      colpart[(size_t)blockIdx.z * pitch + j] = t;
    }
  }
  float* out = part + (size_t)blockIdx.z * pitch;
#pragma unroll
  for (int ja = 0; ja < 2; ++ja)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int k = k0 + (kb0 + kb) * 32 + i;
      if (k >= K) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int j = j0 + (jb0 + ja) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (j < J) out[(size_t)j * K + k] = acc[ja][kb][reg];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// The whole backward of a linear layer on FEW rows (R < 2048: the STNs' FC tails, the ts head, the rotation heads' global
// halves - rows are clouds or objects) as ONE launch, in the latency form of k_linear (catre_kernels.hip): a workgroup per
// 32 x 32 output block, its 8 waves split the contraction in interleaved 8-wide chunks with up to 8 chunks in flight per
// wave, operands straight from L2, partial blocks summed through LDS in wave order (deterministic, no split buffers, no
// merge launch).  dv = dY .* (YM > 0):
//   workgroups [0, ndx)        : dX[R, Kx]  = dv W[J, Kw]                 (columns >= Kw zero)
//   workgroups [ndx, ndx + ndw): dW[J, Kw]  = dv^T X[R, Kx]               (columns >= Kx zero);  k-block 0 also db = colsum dv
// No alignment demands: every load is guarded, so J = 3 / 9 and K = 1091 need no padded copies.  LP: the weight gradient's
// operands rounded to bf16 (the backward of a layer that ran under autocast; db from the unrounded dv, dX in fp32 like
// k_linear_t before it).  It replaced, per layer: k_relu_bwd or masked loads, k_linear_t, k_pad_cols x 1-3,
// k_gemm_tn + k_reduce_splits (+ k_colsum) - 4 to 7 launches of 4-19 us on a dependent chain.
// ------------------------------------------------------------------------------------------------
struct FcBwdArgs {
  const float* dY;  // [R, J], ld = ldy
  const float* YM;  // layer output (ReLU mask), same layout as dY, or null
  const float* X;   // [R, Kx], ld = ldx
  const float* W;   // [J, Kw], ld = ldw
  float* dX;        // [R, Kx] contiguous or null
  float* dW;        // [J, Kw] contiguous or null
  float* db;        // [J] or null (needs dW's workgroups: dW may be null only when db is)
  int ldy, ldx, ldw, R, J, Kx, Kw;
  int ndx, dx_rb;  // dX workgroups, row blocks among them
  int dw_jb;       // j blocks of the dW part
};
#define FCB_WAVES 8
template <bool LP>
__global__ __launch_bounds__(64 * FCB_WAVES) void k_fc_bwd(const FcBwdArgs A) {
  __shared__ float part[FCB_WAVES][16][64];
  __shared__ float colp[FCB_WAVES][64];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool dx_role = (int)blockIdx.x < A.ndx;
  f32x16 acc = zero16();
  float cs = 0.f;
  int row0, col0, nrow, ncol, ldo;  // output block origin, output extents, leading dimension
  float* out;
  bool do_col = false;
  // (addresses are 32-bit element offsets from wave-uniform bases: R * ld < 2^23 here, and one register per load in
  // flight instead of two.  Every load of a trip is unconditional - clamped indices, validity applied afterwards - so that
  // the whole trip is ONE round trip: a guarded load whose select sits in the same block waits for itself.)
  if (dx_role) {
    const int rb = blockIdx.x % A.dx_rb, kb = blockIdx.x / A.dx_rb;
    row0 = rb * 32, col0 = kb * 32, nrow = A.R, ncol = A.Kx, ldo = A.Kx, out = A.dX;
    const unsigned r = min(row0 + i, A.R - 1), k = min(col0 + i, A.Kw - 1);
    const bool kok = col0 + i < A.Kw;
    const float* __restrict__ Y = A.dY;
    const float* __restrict__ M = A.YM;
    const float* __restrict__ W = A.W;
    const unsigned yo = r * (unsigned)A.ldy;
    const int nkc = (A.J + 7) / 8;
    auto trips = [&](auto vec_c, auto mask_c) {
      constexpr bool VEC = decltype(vec_c)::value, MASK = decltype(mask_c)::value;
      constexpr int U = MASK ? 4 : 8;  // chunks in flight per wave (three operands with the mask: 128 registers hold 4)
#pragma unroll 1
      for (int kc = wave; kc < nkc; kc += U * FCB_WAVES) {
        f32x4 a[U], m[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = min(kc + FCB_WAVES * u, nkc - 1);
          const int j = 8 * c + 4 * h;
          if constexpr (VEC) {
            a[u] = *reinterpret_cast<const f32x4*>(Y + (yo + j));
            if constexpr (MASK) m[u] = *reinterpret_cast<const f32x4*>(M + (yo + j));
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const unsigned jj = min(j + s, A.J - 1);
              a[u][s] = Y[yo + jj];
              if constexpr (MASK) m[u][s] = M[yo + jj];
            }
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) b[u][s] = W[(unsigned)min(j + s, A.J - 1) * (unsigned)A.ldw + k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = kc + FCB_WAVES * u;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            bool ok = c < nkc && (VEC || 8 * c + 4 * h + s < A.J);
            if constexpr (MASK) ok = ok && m[u][s] > 0.f;
            a[u][s] = ok ? a[u][s] : 0.f;
            b[u][s] = kok ? b[u][s] : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = mfma32(a[u][s], b[u][s], acc);  // D[row r][col k]
      }
    };
    // 16-byte loads along the contraction when every chunk is whole and aligned
    const bool vec = (A.ldy & 3) == 0 && (A.J & 7) == 0;
    if (vec) {
      if (M) trips(std::true_type{}, std::true_type{});
      else trips(std::true_type{}, std::false_type{});
    } else {
      if (M) trips(std::false_type{}, std::true_type{});
      else trips(std::false_type{}, std::false_type{});
    }
  } else {
    const int bw = blockIdx.x - A.ndx;
    const int jb = bw % A.dw_jb, kb = bw / A.dw_jb;
    row0 = jb * 32, col0 = kb * 32, nrow = A.J, ncol = A.Kw, ldo = A.Kw, out = A.dW;
    do_col = A.db != nullptr && kb == 0;
    const bool jok = row0 + i < A.J, kok = col0 + i < A.Kx;
    const unsigned j = min(row0 + i, A.J - 1), k = min(col0 + i, A.Kx - 1);
    const float* __restrict__ Y = A.dY;
    const float* __restrict__ M = A.YM;
    const float* __restrict__ X = A.X;
    const int nkc = (A.R + 7) / 8;
    auto trips = [&](auto mask_c) {
      constexpr bool MASK = decltype(mask_c)::value;
#pragma unroll 1
      for (int kc = wave; kc < nkc; kc += 4 * FCB_WAVES) {
        f32x4 a[4], m[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = min(kc + FCB_WAVES * u, nkc - 1);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const unsigned r = min(8 * c + 4 * h + s, A.R - 1);
            a[u][s] = Y[r * (unsigned)A.ldy + j];
            if constexpr (MASK) m[u][s] = M[r * (unsigned)A.ldy + j];
            b[u][s] = X[r * (unsigned)A.ldx + k];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = kc + FCB_WAVES * u;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            bool ok = jok && c < nkc && 8 * c + 4 * h + s < A.R;
            if constexpr (MASK) ok = ok && m[u][s] > 0.f;
            a[u][s] = ok ? a[u][s] : 0.f;
            b[u][s] = kok ? b[u][s] : 0.f;
          }
        }
        if (do_col) {
#pragma unroll
          for (int u = 0; u < 4; ++u) cs += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float av = LP ? bf_lo(pack_bf2(a[u][s], 0.f)) : a[u][s];
            const float bv = LP ? bf_lo(pack_bf2(b[u][s], 0.f)) : b[u][s];
            acc = mfma32(av, bv, acc);  // D[row j][col k]
          }
      }
    };
    if (M) trips(std::true_type{});
    else trips(std::false_type{});
    if (do_col) colp[wave][lane] = cs;
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) part[wave][reg][lane] = acc[reg];
  __syncthreads();
  if (do_col && tid < 32 && row0 + tid < A.J) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < FCB_WAVES; ++w) v += colp[w][tid] + colp[w][tid + 32];
    A.db[row0 + tid] = v;
  }
  const int col = col0 + i;
  if (col >= ncol) return;
#pragma unroll
  for (int q = 0; q < 2; ++q) {  // wave w finishes registers 2w, 2w+1 -> rows (reg&3) + 8(reg>>2) + 4h
    const int reg = wave * 2 + q;
    const int row = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (row < nrow) {
      float v = part[0][reg][lane];
#pragma unroll
      for (int w = 1; w < FCB_WAVES; ++w) v += part[w][reg][lane];
      out[(size_t)row * ldo + col] = v;
    }
  }
}

// out[i] = sum_s part[s][i]  (fixed order: deterministic)
__global__ void k_reduce_splits(const float* __restrict__ part, float* __restrict__ out, int n, int splits,
                                int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? out[i] : 0.f;
#pragma unroll 8
  for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
  out[i] = s;
}

// the same sum for FEW outputs and MANY partials (a bias / affine gradient over 256 objects: k_reduce_splits would walk the
// partials in one dependent chain per thread on a handful of workgroups): 64 outputs x 16 partial groups per workgroup, group
// g adds partials g, g + 16, ... in order, the 16 group sums are added in order.  Outputs [0, n_a) go to out_a, the rest to
// out_b (the dgamma | dbeta pairs of the GroupNorm backward kernels).  Deterministic.
__global__ __launch_bounds__(1024) void k_reduce_splits_wide(const float* __restrict__ part, float* __restrict__ out_a,
                                                             float* __restrict__ out_b, int n, int n_a, int splits,
                                                             int accumulate, size_t pitch) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  float s = 0.f;
  if (i < n) {
    int k = g;
    for (; k + 48 < splits; k += 64) {  // four partials of this group requested together
      const float v0 = part[(size_t)k * pitch + i], v1 = part[(size_t)(k + 16) * pitch + i];
      const float v2 = part[(size_t)(k + 32) * pitch + i], v3 = part[(size_t)(k + 48) * pitch + i];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; k < splits; k += 16) s += part[(size_t)k * pitch + i];
  }
  red[g][c] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    float* o = i < n_a ? out_a + i : out_b + (i - n_a);
    float t = accumulate ? *o : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][c];
    *o = t;
  }
}

// column sums: part[split][J] = sum over rows of the split of dY[r][j]
// same with an explicit row pitch of the partial buffer
__global__ void k_reduce_splits_p(const float* __restrict__ part, float* __restrict__ out, int n, int splits,
                                  int accumulate, size_t pitch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? out[i] : 0.f;
#pragma unroll 8
  for (int k = 0; k < splits; ++k) s += part[(size_t)k * pitch + i];
  out[i] = s;
}

// dst [rows][cols_pad] = src [rows][cols] (element strides sr, sc: transposed views too) with zero columns behind: the
// zero-padding of small operands to the GEMM kernels' granularity in ONE launch (torch's F.pad is a fill plus a copy)
__global__ void k_pad_cols(const float* __restrict__ src, long sr, long sc, int rows, int cols, float* __restrict__ dst,
                           int cols_pad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * cols_pad) return;
  const int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
  dst[i] = c < cols ? src[(size_t)r * sr + (size_t)c * sc] : 0.f;
}

__device__ __forceinline__ void colsum_body(const float* __restrict__ dY, int ld, float* __restrict__ part, int R, int J,
                                            int rows_per_split, int bx, int by, float* red /*[256]*/) {
  // a block covers JB = min(J, 256) columns with 256/JB row lanes, so narrow matrices still use every thread
  const int JB = J < 256 ? J : 256;
  const int RL = 256 / JB;  // row lanes (>= 1); threads beyond RL*JB idle
  const int col = threadIdx.x % JB, rl = threadIdx.x / JB;
  const int j = bx * 256 + col;
  const int lo = by * rows_per_split, hi = min(R, lo + rows_per_split);
  float s = 0.f;
  if (rl < RL && j < J) {
    // eight rows requested before the first add (same order of additions): as a plain loop every load was waited for
    // before the next was issued - 2.3 TB/s on the 67 MB partial buffers of the rot-head weight gradients
    int r = lo + rl;
    for (; r + 7 * RL < hi; r += 8 * RL) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(dY + (size_t)(r + u * RL) * ld + j);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < hi; r += RL) s += dY[(size_t)r * ld + j];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && j < J) {
    for (int k = 1; k < RL; ++k) s += red[k * JB + col];
    part[(size_t)by * J + j] = s;
  }
}

__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ dY, int ld, float* __restrict__ part, int R,
                                                int J, int rows_per_split) {
  __shared__ float red[256];
  colsum_body(dY, ld, part, R, J, rows_per_split, blockIdx.x, blockIdx.y, red);
}

// Several two-stage column reductions behind one kernel (the partial buffers a backward kernel leaves: weight gradient,
// affine gradients, bias gradient) as TWO launches instead of two per buffer: k_colsum_multi = k_colsum's body per job
// (64-row column sums into the job's stage), k_reduce_multi = the fixed-order merge of the stage rows into out_a[0, n_a) |
// out_b[0, J - n_a).  Same operations in the same order as the single-job kernels: same bits.
#define RED_MAX_JOBS 4
struct RedJob {
  const float* src;  // [R][ld]
  float* stage;      // [nsp][J]
  float* out_a;
  float* out_b;      // outputs n_a .. J-1 (or null when n_a == J)
  int ld, R, J, n_a, nsp, rps, accumulate;
  int sb0, fb0;      // first workgroup of the job in the stage / final launch
};
struct RedJobs {
  RedJob j[RED_MAX_JOBS];
  int n;
};
__global__ __launch_bounds__(256) void k_colsum_multi(const RedJobs A) {
  __shared__ float red[256];
  int q = 0;
#pragma unroll
  for (int t = 1; t < RED_MAX_JOBS; ++t)
    if (t < A.n && (int)blockIdx.x >= A.j[t].sb0) q = t;
  const RedJob& J = A.j[q];
  const int gx = (J.J + 255) / 256, local = blockIdx.x - J.sb0;
  colsum_body(J.src, J.ld, J.stage, J.R, J.J, J.rps, local % gx, local / gx, red);
}
__global__ __launch_bounds__(256) void k_reduce_multi(const RedJobs A) {
  int q = 0;
#pragma unroll
  for (int t = 1; t < RED_MAX_JOBS; ++t)
    if (t < A.n && (int)blockIdx.x >= A.j[t].fb0) q = t;
  const RedJob& J = A.j[q];
  const int i = (blockIdx.x - J.fb0) * 256 + threadIdx.x;
  if (i >= J.J) return;
  float* o = i < J.n_a ? J.out_a + i : J.out_b + (i - J.n_a);
  float s = J.accumulate ? *o : 0.f;
  int k = 0;
  for (; k + 8 <= J.nsp; k += 8) {  // eight stage rows requested together, added in order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = J.stage[(size_t)(k + u) * J.J + i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < J.nsp; ++k) s += J.stage[(size_t)k * J.J + i];
  *o = s;
}

// ------------------------------------------------------------------------------------------------
// cloud layout helper: rows of all observed clouds first (B*N), then all prior clouds (B*M)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cloud_of_row(int r, int B, int N, int M) { return r < B * N ? r / N : B + (r - B * N) / M; }
__device__ __forceinline__ void cloud_rows(int c, int B, int N, int M, int& r0, int& n) {
  if (c < B) {
    r0 = c * N;
    n = N;
  } else {
    r0 = B * N + (c - B) * M;
    n = M;
  }
}

// Y[r][j] (+)= bias[cloud(r)][j]   (rot-head layer 0: global-feature half as a per-cloud bias)
__global__ void k_rowbias_add(float* __restrict__ Y, int ld, const float* __restrict__ bias, int R, int J, int B, int N,
                              int M) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)R * (J / 4)) return;
  const int r = idx / (J / 4), j4 = idx % (J / 4);
  // rows here are ordered per OBJECT ([obs N | prior M] per object): cloud = obs or prior of object r / (N+M)
  const int obj = r / (N + M), p = r % (N + M);
  const int c = p < N ? obj : B + obj;
  f32x4* y = reinterpret_cast<f32x4*>(Y + (size_t)r * ld) + j4;
  const f32x4 b = *(reinterpret_cast<const f32x4*>(bias + (size_t)c * J) + j4);
  f32x4 v = *y;
  v[0] += b[0];
  v[1] += b[1];
  v[2] += b[2];
  v[3] += b[3];
  *y = v;
}

// dbias[c][j] = sum over the rows of cloud c (object-major row order as above) of dY[r][j]
// 64 columns x 4 row lanes per workgroup, 8 rows in flight per thread (one thread per column walking the cloud's rows
// alone was latency-bound: 1.2 TB/s); the four lane sums are merged in lane order
__global__ __launch_bounds__(256) void k_rowbias_bwd(const float* __restrict__ dY, int ld, float* __restrict__ dbias,
                                                     int J, int B, int N, int M) {
  __shared__ float red[256];
  const int c = blockIdx.x, j = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int obj = c < B ? c : c - B;
  const int r0 = obj * (N + M) + (c < B ? 0 : N), n = c < B ? N : M;
  float s = 0.f;
  if (j < J) {
    const float* col = dY + (size_t)r0 * ld + j;
    int r = rl;
    for (; r + 28 < n; r += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(r + 4 * u) * ld];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < n; r += 4) s += col[(size_t)r * ld];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && j < J)
    dbias[(size_t)c * J + j] = ((red[threadIdx.x] + red[64 + threadIdx.x]) + red[128 + threadIdx.x]) + red[192 + threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// max-pool over the points of each cloud with argmax (cloud-major rows), and its sparse backward
// ------------------------------------------------------------------------------------------------
// per-tile (max, arg-max row) partials [tiles][J] -> per-cloud; tiles never straddle clouds (N, M multiples of 64)
__global__ __launch_bounds__(256) void k_maxpool_tiles(const float* __restrict__ pmax, const int* __restrict__ pidx,
                                                       float* __restrict__ out, int* __restrict__ idx, int J, int B,
                                                       int N, int M) {
  const int c = blockIdx.x, j = blockIdx.y * 256 + threadIdx.x;
  if (j >= J) return;
  int r0, n;
  cloud_rows(c, B, N, M, r0, n);
  const int t0 = r0 / TP, nt = n / TP;
  float m = pmax[(size_t)t0 * J + j];
  int am = pidx[(size_t)t0 * J + j];
  for (int t = 1; t < nt; ++t) {
    const float v = pmax[(size_t)(t0 + t) * J + j];
    if (v > m) {  // tiles in row order: strict > keeps the first maximum
      m = v;
      am = pidx[(size_t)(t0 + t) * J + j];
    }
  }
  out[(size_t)c * J + j] = m;
  idx[(size_t)c * J + j] = am;
}

// same shape as k_rowbias_bwd: 64 columns x 4 row lanes, 8 rows in flight; first maximum wins (like torch.max)
__global__ __launch_bounds__(256) void k_maxpool_fwd(const float* __restrict__ Y, int ld, float* __restrict__ out,
                                                     int* __restrict__ idx, int J, int B, int N, int M) {
  __shared__ float redm[256];
  __shared__ int reda[256];
  const int c = blockIdx.x, j = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  int r0, n;
  cloud_rows(c, B, N, M, r0, n);
  float m = -INFINITY;
  int am = n;  // "none yet": loses every tie
  if (j < J) {
    const float* col = Y + (size_t)r0 * ld + j;
    int r = rl;
    for (; r + 28 < n; r += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(r + 4 * u) * ld];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v[u] > m) {
          m = v[u];
          am = r + 4 * u;
        }
    }
    for (; r < n; r += 4) {
      const float v = col[(size_t)r * ld];
      if (v > m) {
        m = v;
        am = r;
      }
    }
  }
  redm[threadIdx.x] = m;
  reda[threadIdx.x] = am;
  __syncthreads();
  if (rl == 0 && j < J) {
#pragma unroll
    for (int l = 1; l < 4; ++l) {
      const float mo = redm[l * 64 + threadIdx.x];
      const int ao = reda[l * 64 + threadIdx.x];
      if (mo > m || (mo == m && ao < am)) {
        m = mo;
        am = ao;
      }
    }
    if (am >= n) {  // a column of NaNs / -inf: row 0 and its value, like a serial walk
      am = 0;
      m = Y[(size_t)r0 * ld + j];
    }
    out[(size_t)c * J + j] = m;
    idx[(size_t)c * J + j] = r0 + am;
  }
}

// scatter: dY[idx[c][j]][j] = dout[c][j] on a zeroed dY (dense form, for the narrow pools)
__global__ void k_maxpool_scatter(const float* __restrict__ dout, const int* __restrict__ idx, float* __restrict__ dY,
                                  int ld, int C, int J, int accumulate = 0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * J) return;
  const int j = i % J;
  float* d = dY + (size_t)idx[i] * ld + j;  // (cloud, channel) -> one (row, column): no two threads share a target
  *d = accumulate ? *d + dout[i] : dout[i];
}

// Sparse backward of  Y = X W^T + b ; g = max_rows Y  without materialising dY:
//   dW[j][:] = sum_c dg[c][j] * X[idx[c][j]][:]      (one workgroup per output channel j)
__global__ __launch_bounds__(256) void k_maxlin_bwd_w(const float* __restrict__ dg, const int* __restrict__ idx,
                                                      const float* __restrict__ X, int ldx, float* __restrict__ dW,
                                                      float* __restrict__ db, int C, int J, int K) {
  // dW[j][:] = sum_c dg[c][j] * X[argmax row of (c, j)][:]; one workgroup per channel, clouds in order (deterministic).
  // The chain dg/idx -> row pointer -> X is two dependent global round trips per cloud: 8 clouds are in flight at once.
  const int j = blockIdx.x;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // K <= 1024 with 256 threads
  float sb = 0.f;
  for (int c0 = 0; c0 < C; c0 += 8) {
    float g[8];
    int row[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = min(c0 + u, C - 1);
      g[u] = c0 + u < C ? dg[(size_t)c * J + j] : 0.f;
      row[u] = idx[(size_t)c * J + j];
    }
    float xv[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = threadIdx.x + 256 * q;
        xv[u][q] = k < K ? X[(size_t)row[u] * ldx + k] : 0.f;
      }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      sb += g[u];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = fmaf(g[u], xv[u][q], acc[q]);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = threadIdx.x + 256 * u;
    if (k < K) dW[(size_t)j * K + k] = acc[u];
  }
  if (threadIdx.x == 0 && db) db[j] = sb;
}
// The same gradient for K % 4 == 0, K <= 1024 with the workgroup split into CL = 256 / (K/4) CLOUD LANES: lane l walks the
// clouds l, l + CL, ... with eight of them in flight, a thread owns one float4 column of its lane's rows, and the lanes'
// partial sums are merged through LDS in lane order (deterministic).  The walk above is one chain of C / 8 steps of two
// dependent global round trips whatever K is (150 us for K = 128 as for K = 512); here it is C / (8 CL) steps.
// rowpos != nullptr: X holds COMPACT rows (the live rows of a row-sparse chain, recomputed by k_stn_recompute) - dense row r
// sits at rowpos[r]; an arg-max row with dg == 0 is not live (rowpos < 0) and is read as row 0 (its product is an exact zero).
template <bool XH = false>  // XH: X holds bf16 rows (ldx in elements)
__global__ __launch_bounds__(256) void k_maxlin_bwd_w4(const float* __restrict__ dg, const int* __restrict__ idx,
                                                       const float* __restrict__ X, int ldx, float* __restrict__ dW,
                                                       float* __restrict__ db, int C, int J, int K,
                                                       const int* __restrict__ rowpos = nullptr) {
  __shared__ f32x4 red[256];
  __shared__ float reds[256];
  const int j = blockIdx.x, Q = K >> 2, CL = 256 / Q, q = threadIdx.x % Q, cl = threadIdx.x / Q;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float sb = 0.f;
  if (cl < CL) {
    for (int c0 = cl; c0 < C; c0 += 8 * CL) {
      float g[8];
      int row[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u * CL, cc = min(c, C - 1);
        g[u] = c < C ? dg[(size_t)cc * J + j] : 0.f;
        row[u] = idx[(size_t)cc * J + j];
      }
      if (rowpos) {
#pragma unroll
        for (int u = 0; u < 8; ++u) row[u] = max(rowpos[row[u]], 0);
      }
      f32x4 xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if constexpr (XH) xv[u] = ld_bf4(X, ((size_t)row[u] * ldx >> 2) + q);
        else xv[u] = *reinterpret_cast<const f32x4*>(X + (size_t)row[u] * ldx + 4 * q);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        sb += g[u];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(g[u], xv[u][e], acc[e]);
      }
    }
  }
  red[threadIdx.x] = acc;
  reds[threadIdx.x] = sb;
  __syncthreads();
  if (cl == 0) {
    for (int l = 1; l < CL; ++l) {
      const f32x4 o = red[l * Q + q];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += o[e];
    }
    *reinterpret_cast<f32x4*>(dW + (size_t)j * K + 4 * q) = acc;
    if (q == 0 && db) {
      for (int l = 1; l < CL; ++l) sb += reds[l * Q];
      db[j] = sb;
    }
  }
}

// dX of linear + max-pool for one cloud per workgroup, every row written exactly once (no read-modify-write, no
// pre-zeroed buffer, no barrier per channel):
//   1. histogram of the arg-max rows of the cloud's J channels (LDS), exclusive scan -> bucket offsets
//   2. one wave drops the channels into their row's bucket in ascending channel order (ballot ranking inside each
//      64-channel chunk), so the per-row summation order - and with it every bit of the result - is fixed
//   3. the waves walk the rows: dX[row][:] = sum over the row's channels of dg * W[channel][:], zeros for rows that
//      were nobody's arg-max.
#define MLX_MAXN 4096
// rowpos != nullptr (row-sparse chains, below): only the rows that are somebody's arg-max are written, row r of the cloud at
// compact row rowpos[r0 + r]; ymask (the layer's own ReLU output, dense rows) zeroes what that ReLU killed.
__global__ __launch_bounds__(512) void k_maxlin_bwd_x_rows(const float* __restrict__ dg, const int* __restrict__ idx,
                                                           const float* __restrict__ W, int ldw, float* __restrict__ dX,
                                                           int ldx, int J, int K, int B, int N, int M,
                                                           const int* __restrict__ rowpos = nullptr,
                                                           const float* __restrict__ ymask = nullptr, int ldym = 0,
                                                           int ymask_compact = 0 /*ymask holds the compact rows*/,
                                                           int ymask_bf16 = 0 /*ymask: bf16 rows, ldym in elements*/) {
  __shared__ int start[MLX_MAXN + 1];
  __shared__ int fill[MLX_MAXN];
  __shared__ int lst[1024];
  __shared__ int part[256];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int r0, n;
  cloud_rows(c, B, N, M, r0, n);
  const float* g = dg + (size_t)c * J;
  const int* ix = idx + (size_t)c * J;
  for (int i = tid; i < n; i += 512) {
    start[i] = 0;
    fill[i] = 0;
  }
  __syncthreads();
  for (int j = tid; j < J; j += 512)
    if (g[j] != 0.f) atomicAdd(&start[ix[j] - r0], 1);
  __syncthreads();
  {  // exclusive scan of start[0..n): thread t < 256 owns a contiguous run of bins
    const int per = (n + 255) / 256, lo = min(n, tid * per), hi = tid < 256 ? min(n, lo + per) : lo;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += start[i];
    if (tid < 256) part[tid] = s;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int t = 0; t < 256; ++t) {
        const int v = part[t];
        part[t] = run;
        run += v;
      }
      start[n] = run;
    }
    __syncthreads();
    int run = tid < 256 ? part[tid] : 0;
    for (int i = lo; i < hi; ++i) {
      const int v = start[i];
      start[i] = run;
      run += v;
    }
  }
  __syncthreads();
  {
    // buckets in ascending (row, channel) order = a sort of the keys row << 10 | channel (dead channels: a sentinel that
    // sorts last).  Bitonic network over 1024 keys in LDS, one compare-exchange per thread and step (55 steps) - the
    // per-row summation order, and with it every bit of the result, is fixed.  (The first version let ONE wave drop
    // the channels into their buckets with a ballot loop per distinct row: ~800 serial LDS round trips, 70 us of a
    // 200 us kernel, and the reason several workgroups per cloud did not pay.)
    for (int j = tid; j < 1024; j += 512) {
      const bool live = j < J && g[min(j, J - 1)] != 0.f;
      lst[j] = live ? ((ix[j] - r0) << 10) | j : 0x7fffffff;
    }
    __syncthreads();
    for (int k = 2; k <= 1024; k <<= 1)
      for (int jj = k >> 1; jj > 0; jj >>= 1) {
        const int i = ((tid / jj) * 2 * jj) + (tid % jj), p = i + jj;
        const int a = lst[i], b = lst[p];
        const bool up = (i & k) == 0;
        if ((a > b) == up) {
          lst[i] = b;
          lst[p] = a;
        }
        __syncthreads();
      }
    for (int j = tid; j < 1024; j += 512) lst[j] &= 1023;  // bucket entries: the channel (sentinels are never read)
    __syncthreads();
  }
  const int nf4 = K / 4;  // K % 4 == 0 (checked by the launcher); K <= 512: at most two float4 per lane
  // rows are dealt round-robin to (workgroup of the cloud, wave): with few clouds the launcher gives every cloud
  // several workgroups (each repeats the cheap bucketing above) so that the row walk is not one long serial chain.
  // A row is a chain of dependent loads (bucket entry -> dg, W row): eight waves walk rows side by side and a lane's
  // two column slices are requested together - the walk is bound by that latency, not by the bytes it writes.
  // (the channel gradients of the cloud sit in LDS - `fill` is free after the bucketing - so that a row's chain is bucket
  // entry (LDS) -> W row (L2) and not bucket entry -> dg (L2) -> W row (L2); and a row's channels are fetched FOUR at a
  // time: most rows own one to four channels, so a row costs one L2 round trip instead of one per channel.  The sums run
  // in the same ascending channel order as before: not a bit changes.)
  float* gl = reinterpret_cast<float*>(fill);
  for (int j = tid; j < 1024; j += 512) gl[j] = j < J ? g[j] : 0.f;
  __syncthreads();
  for (int r = blockIdx.y * 8 + wave; r < n; r += 8 * gridDim.y) {
    const int b = start[r], e = start[r + 1];
    if (rowpos && b == e) continue;  // compact destination: rows without a channel do not exist
    float* xr = dX + (size_t)(rowpos ? rowpos[r0 + r] : r0 + r) * ldx;
    const int q0 = lane, q1 = lane + 64;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int t = b; t < e; t += 4) {
      int jj[4];
      float gv[4];
      f32x4 w0[4], w1[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bool on = t + d < e;  // wave-uniform
        jj[d] = lst[min(t + d, e - 1)];
        gv[d] = on ? gl[jj[d]] : 0.f;
        w0[d] = on && q0 < nf4 ? *reinterpret_cast<const f32x4*>(W + (size_t)jj[d] * ldw + q0 * 4) : z;
        w1[d] = on && q1 < nf4 ? *reinterpret_cast<const f32x4*>(W + (size_t)jj[d] * ldw + q1 * 4) : z;
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        if (t + d < e) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc0[u] = fmaf(gv[d], w0[d][u], acc0[u]);
            acc1[u] = fmaf(gv[d], w1[d][u], acc1[u]);
          }
        }
      }
    }
    if (ymask) {
      const size_t mro = (size_t)(ymask_compact ? rowpos[r0 + r] : r0 + r) * ldym;
      const float* mr = ymask + mro;
      if (q0 < nf4) {
        const f32x4 m = ymask_bf16 ? ld_bf4(ymask, (mro >> 2) + q0) : *reinterpret_cast<const f32x4*>(mr + q0 * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc0[u] = m[u] > 0.f ? acc0[u] : 0.f;
      }
      if (q1 < nf4) {
        const f32x4 m = ymask_bf16 ? ld_bf4(ymask, (mro >> 2) + q1) : *reinterpret_cast<const f32x4*>(mr + q1 * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc1[u] = m[u] > 0.f ? acc1[u] : 0.f;
      }
    }
    if (q0 < nf4) *reinterpret_cast<f32x4*>(xr + q0 * 4) = acc0;
    if (q1 < nf4) *reinterpret_cast<f32x4*>(xr + q1 * 4) = acc1;
  }
}

// ------------------------------------------------------------------------------------------------
// Row-sparse backward of linear + max-pool chains.  Only the arg-max row of a (cloud, channel) carries gradient, so the
// gradient of the layer in front of a pool - and of every layer further up the conv stack, until it meets a dense side
// input - is zero on every row that is nobody's arg-max: ~70 % of the rows at N = M = 1024 (profiles/argmax_row_fraction.py).
// These kernels build the ascending list of live rows and a dense -> compact map on the device (no host sync: the
// count stays in `count[0]`, the row GEMMs read it - Rdev above), so that dgrad / wgrad run on the compacted rows only.
//   k_rows_count   : per cloud, how many distinct rows are the arg-max of a channel with dg != 0
//   k_rows_fill    : base[c] = the counts of the clouds in front of c, summed by the workgroup itself (no scan launch);
//                    rows[base[c] + rank] = r (ascending), rowpos[r] = base[c] + rank or -1; the last cloud writes count[0]
// One workgroup per cloud, flags in LDS (clouds of <= MLX_MAXN points).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int rows_flag_scan(const float* __restrict__ g, const int* __restrict__ ix, int J, int r0, int n,
                                              int* flag /*[MLX_MAXN]*/, int* part /*[256]*/, int tid) {
  for (int i = tid; i < n; i += 512) flag[i] = 0;
  __syncthreads();
  for (int j = tid; j < J; j += 512)
    if (g[j] != 0.f) flag[ix[j] - r0] = 1;
  __syncthreads();
  // exclusive scan of flag[0..n) in place (flag[i] becomes the rank, bit 30 keeps "live"): thread t < 256 owns a run
  const int per = (n + 255) / 256, lo = min(n, tid * per), hi = tid < 256 ? min(n, lo + per) : lo;
  int s = 0;
  for (int i = lo; i < hi; ++i) s += flag[i];
  if (tid < 256) part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) {
      const int v = part[t];
      part[t] = run;
      run += v;
    }
    flag[MLX_MAXN] = run;  // total of the cloud
  }
  __syncthreads();
  int run = tid < 256 ? part[tid] : 0;
  for (int i = lo; i < hi; ++i) {
    const int v = flag[i];
    flag[i] = v ? run : -1;
    run += v;
  }
  __syncthreads();
  return flag[MLX_MAXN];
}

__global__ __launch_bounds__(512) void k_rows_count(const float* __restrict__ dg, const int* __restrict__ idx, int J, int B,
                                                    int N, int M, int* __restrict__ cnt) {
  __shared__ int flag[MLX_MAXN + 1];
  __shared__ int part[256];
  const int c = blockIdx.x;
  int r0, n;
  cloud_rows(c, B, N, M, r0, n);
  const int total = rows_flag_scan(dg + (size_t)c * J, idx + (size_t)c * J, J, r0, n, flag, part, threadIdx.x);
  if (threadIdx.x == 0) cnt[c] = total;
}

// (the cloud's base - the live rows of all clouds in front of it - is summed here from the per-cloud counts, 2 KB per
// workgroup: the single-workgroup scan launch that sat between count and fill is gone; the last cloud also writes the total)
__global__ __launch_bounds__(512) void k_rows_fill(const float* __restrict__ dg, const int* __restrict__ idx, int J, int B,
                                                   int N, int M, const int* __restrict__ cnt, int C, int* __restrict__ rows,
                                                   int* __restrict__ rowpos, int* __restrict__ count) {
  __shared__ int flag[MLX_MAXN + 1];
  __shared__ int part[256];
  __shared__ int bsum[8];
  const int c = blockIdx.x, tid = threadIdx.x;
  int r0, n;
  cloud_rows(c, B, N, M, r0, n);
  int mine = 0;
  for (int i = tid; i < c; i += 512) mine += cnt[i];  // requested before the flag scan's own loads
  const int total = rows_flag_scan(dg + (size_t)c * J, idx + (size_t)c * J, J, r0, n, flag, part, tid);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if ((tid & 63) == 0) bsum[tid >> 6] = mine;
  __syncthreads();
  const int b = ((bsum[0] + bsum[1]) + (bsum[2] + bsum[3])) + ((bsum[4] + bsum[5]) + (bsum[6] + bsum[7]));
  if (c == C - 1 && tid == 0) count[0] = b + total;
  for (int i = tid; i < n; i += 512) {
    const int rk = flag[i];
    rowpos[r0 + i] = rk < 0 ? -1 : b + rk;
    if (rk >= 0) rows[b + rk] = r0 + i;
  }
}

// dst[i][:cols] = src[rows[i]][:cols] for i < count[0]  (cols % 4 == 0); one wave per row, grid-stride
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ src, int lds_, const int* __restrict__ rows,
                                                     const int* __restrict__ count, float* __restrict__ dst, int ldd,
                                                     int cols) {
  const int n = count[0], lane = threadIdx.x & 63, q = cols >> 2;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
    const float* s = src + (size_t)rows[i] * lds_;
    float* d = dst + (size_t)i * ldd;
    for (int c4 = lane; c4 < q; c4 += 64)
      *reinterpret_cast<f32x4*>(d + 4 * c4) = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s) + c4);
  }
}

// dense[r][:cols] = rowpos[r] >= 0 ? srcc[rowpos[r]][:cols] : 0 for every r < R: every row written once, no memset.
// objsrc (optional, [B (N + M)][cols] in OBJECT-major row order): added on the way - the gradient the rotation heads send
// to the same cloud-major rows (train_ops.object_major), so that the sum of a tensor's consumers' gradients is one pass.
__global__ __launch_bounds__(256) void k_scatter_rows(const float* __restrict__ srcc, int lds_, const int* __restrict__ rowpos,
                                                      float* __restrict__ dst, int ldd, int cols, int R,
                                                      const float* __restrict__ objsrc = nullptr, int ldo = 0, int B = 0,
                                                      int N = 0, int M = 0) {
  // a wave carries RPW rows at a time: 64 / (cols / 4) of them when a row is narrower than 64 float4 (the 64-column rows of
  // pointfeat / h1: four rows per wave - one row per wave left 48 of 64 lanes idle and the kernel at 1.7 TB/s of pure writes)
  const int lane = threadIdx.x & 63, q = cols >> 2;
  const int rpw = q >= 64 ? 1 : 64 / q, sub = q >= 64 ? 0 : lane / q, c0 = q >= 64 ? lane : lane % q;
  const bool live = q >= 64 || sub < rpw;  // (q not a divisor of 64: the last lanes sit out)
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw; r0 < R; r0 += gridDim.x * 4 * rpw) {
    const int r = r0 + sub;
    if (!live || r >= R) continue;
    const int p = rowpos[r];
    float* d = dst + (size_t)r * ldd;
    const float* o = nullptr;
    if (objsrc) {
      const int ro = r < B * N ? (r / N) * (N + M) + r % N : ((r - B * N) / M) * (N + M) + N + (r - B * N) % M;
      o = objsrc + (size_t)ro * ldo;
    }
    for (int c4 = c0; c4 < q; c4 += 64) {
      f32x4 v = p >= 0 ? reinterpret_cast<const f32x4*>(srcc + (size_t)p * lds_)[c4] : z;
      if (o) {
        const f32x4 a = reinterpret_cast<const f32x4*>(o)[c4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += a[e];
      }
      *reinterpret_cast<f32x4*>(d + 4 * c4) = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_maxlin_bwd_x(const float* __restrict__ dg, const int* __restrict__ idx,
                                                      const float* __restrict__ W, int ldw, float* __restrict__ dX,
                                                      int ldx, int J, int K) {
  const int c = blockIdx.x;
  for (int j = 0; j < J; ++j) {
    const float g = dg[(size_t)c * J + j];
    if (g == 0.f) continue;  // wave-uniform
    float* xr = dX + (size_t)idx[(size_t)c * J + j] * ldx;
    const float* wr = W + (size_t)j * ldw;
    for (int k = threadIdx.x; k < K; k += 256) xr[k] = fmaf(g, wr[k], xr[k]);
    __syncthreads();  // the next channel may hit the same row
  }
}

// ------------------------------------------------------------------------------------------------
// Recompute instead of save (STN stacks): the row-sparse backward of a pooled chain reads the two thin layers'
// activations - y1 = relu(conv1 x), y2 = relu(conv2 y1) - on the LIVE rows only (~30 % at N = M = 1024), so the training
// forward stores neither (470 MB per launch, which kept it off the one-wave pair kernels) and this kernel rebuilds them for
// the n = count[0] live rows as COMPACT rows, 64 per workgroup, with the forward kernels' own device code (k_stn3d /
// k_stnkd: same operands, same K order - same bits, so every ReLU mask agrees with the forward's).
//   KIND 0 (STN3d): X = the point rows [R][>= 3]; conv1 3 -> 64 on the VALU (conv3_relu_row)
//   KIND 1 (STNkd): X = h1 rows [R][64] (saved by the trunk kernel); conv1 64 -> 64 as an MFMA layer
// ------------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256, 2) void k_stn_recompute(const float* __restrict__ X, int ldx, const int* __restrict__ rows,
                                                          const int* __restrict__ count, const float* __restrict__ W1,
                                                          const f32x4* __restrict__ wp1, const float* __restrict__ b1,
                                                          const f32x4* __restrict__ wp2, const float* __restrict__ b2,
                                                          float* __restrict__ y1c, float* __restrict__ y2c) {
  __shared__ __attribute__((aligned(16))) float smem[2 * TP * LD64 + TP * LD128];
  float* h1 = smem;                    // KIND 1: gathered input rows
  float* a1 = smem + TP * LD64;        // y1 tile [64][68]
  float* a2 = smem + 2 * TP * LD64;    // y2 tile [64][132]
  const int n = count[0], t0 = blockIdx.x * TP;
  if (t0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  GemmPipe<1, 2, false, false, 8, 3> g2;
  g2.prefetch(wp2 + (wave * 8) * 64 + lane, 0);
  f32x4 bv2[1][4];
  load_bias_quads<1>(bv2, b2, wave * 32, lane);
  if constexpr (KIND == 0) {
    const float* xr = X + (size_t)rows[min(t0 + lane, n - 1)] * ldx;
    conv3_relu_row<16>(xr[0], xr[1], xr[2], W1, b1, wave * 16, a1 + lane * LD64);
    __syncthreads();
  } else {
    const int mblk1 = wave >> 1, nb1 = wave & 1;
    GemmPipe<1, 1, false, false, 8, 4> g1;
    g1.prefetch(wp1 + (mblk1 * 8) * 64 + lane, 0);
    f32x4 bv1[1][4];
    load_bias_quads<1>(bv1, b1, mblk1 * 32, lane);
    {  // gather: thread -> row tid / 4, float4 columns tid % 4 + 4 u
      const int r = tid >> 2, c0 = tid & 3;
      const f32x4* src = reinterpret_cast<const f32x4*>(X + (size_t)rows[min(t0 + r, n - 1)] * ldx);
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(h1 + r * LD64 + (c0 + 4 * u) * 4) = src[c0 + 4 * u];
    }
    __syncthreads();
    f32x16 acc[1][1] = {{zero16()}};
    g1.run(acc, h1 + nb1 * 32 * LD64, LD64, lane);
    store_tile_lds_pre<1, 1, true, false>(acc, a1 + nb1 * 32 * LD64, LD64, mblk1 * 32, bv1, lane);
    __syncthreads();
  }
  {
    f32x16 acc[1][2] = {{zero16(), zero16()}};
    g2.run(acc, a1, LD64, lane);
    store_tile_lds_pre<1, 2, true, false>(acc, a2, LD128, wave * 32, bv2, lane);
  }
  __syncthreads();
  // (rows past count[0] of the last tile hold duplicates of the last live row: inside the buffers' capacity, never read)
  save_tile_rows<64, 256, false>(a1, LD64, y1c + (size_t)t0 * 64, tid);
  save_tile_rows<128, 256, false>(a2, LD128, y2c + (size_t)t0 * 128, tid);
}

// ------------------------------------------------------------------------------------------------
// per-cloud transforms  Y[r][:] = X[r][:] T[c]   (k = 3: input transform, k = 64: feature transform)
// transpose=1 multiplies by T[c]^T (the data gradient).  VALU kernel: 0.2 % of the FLOPs of the path.
// ------------------------------------------------------------------------------------------------
template <int KD>
__global__ __launch_bounds__(256) void k_cloud_matmul(const float* __restrict__ X, int ldx, const float* __restrict__ T,
                                                      float* __restrict__ Y, int ldy, int R, int B, int N, int M,
                                                      int transpose) {
  __shared__ float ts[KD * KD];
  // one workgroup handles 256/KD rows... keep it simple: block = 64 rows of ONE cloud
  const int c = blockIdx.y;
  int r0, n;
  cloud_rows(c, B, N, M, r0, n);
  const int rb = blockIdx.x * 64;
  if (rb >= n) return;
  for (int i = threadIdx.x; i < KD * KD; i += 256) {
    const int a = i / KD, b = i % KD;
    ts[i] = transpose ? T[(size_t)c * KD * KD + b * KD + a] : T[(size_t)c * KD * KD + i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * KD; e += 256) {
    const int row = e / KD, j = e % KD;
    if (rb + row >= n) continue;
    const float* xr = X + (size_t)(r0 + rb + row) * ldx;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < KD; ++i) s = fmaf(xr[i], ts[i * KD + j], s);
    Y[(size_t)(r0 + rb + row) * ldy + j] = s;
  }
}

// kd = 64 on the matrix pipe (the VALU form above ran at 0.9 TB/s): 64 rows of one cloud per workgroup,
// D[row][j] = sum_i X[row][i] T[i][j] with T (or T^T) as the B operand straight from LDS, wave -> (32 channels, 32 rows)
__global__ __launch_bounds__(256) void k_cloud_matmul64(const float* __restrict__ X, int ldx, const float* __restrict__ T,
                                                        float* __restrict__ Y, int ldy, int B, int N, int M,
                                                        int transpose, int tpw) {
  // a workgroup takes `tpw` consecutive 64-row tiles of one cloud: the 16 KiB transform is staged once, and the next
  // tile's rows are requested before the current tile's MFMAs (one tile per workgroup re-read the transform for every
  // tile and waited for each of its three dependent phases: 126 us for 0.4 GB at B = 256)
  __shared__ __attribute__((aligned(16))) float ts[64 * 64];
  __shared__ __attribute__((aligned(16))) float xs[64 * 68];
  const int c = blockIdx.y;
  int r0, n;
  cloud_rows(c, B, N, M, r0, n);
  const int rb0 = blockIdx.x * tpw * 64;
  if (rb0 >= n) return;
  const int rb1 = min(n, rb0 + tpw * 64);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* Tc = T + (size_t)c * 4096;
  const int srow = tid >> 2, c0 = (tid & 3) * 16;
  f32x4 v[4];
  auto fetch = [&](int rb) {
    const bool ok = rb + srow < n;
    const float* src = X + (size_t)(r0 + rb + (ok ? srow : 0)) * ldx + c0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok) v[u] = *reinterpret_cast<const f32x4*>(src + 4 * u);
    }
  };
  fetch(rb0);
  if (transpose) {
    for (int i = tid; i < 4096; i += 256) ts[i] = Tc[(i & 63) * 64 + (i >> 6)];
  } else {
    for (int i = tid; i < 1024; i += 256) reinterpret_cast<f32x4*>(ts)[i] = reinterpret_cast<const f32x4*>(Tc)[i];
  }
  const int mblk = wave >> 1, nb = wave & 1, nn = lane & 31, h = lane >> 5;
  for (int rb = rb0; rb < rb1; rb += 64) {
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(xs + srow * 68 + c0 + 4 * u) = v[u];
    __syncthreads();
    if (rb + 64 < rb1) fetch(rb + 64);
    f32x16 acc = zero16();
    const float* xr = xs + (nb * 32 + nn) * 68 + 4 * h;
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
      const f32x4 bx = *reinterpret_cast<const f32x4*>(xr + kc * 8);
#pragma unroll
      // swapped operand order: D[row][j] - the lane owns channel mblk*32 + nn and 16 of the tile's rows, so that the
      // stores below cover 128 consecutive bytes of a row per half-wave (whole L2 lines; see k_rot_l1<1, true>)
      for (int q = 0; q < 4; ++q) acc = mfma32(bx[q], ts[(kc * 8 + 4 * h + q) * 64 + mblk * 32 + nn], acc);
    }
    float* dst = Y + (size_t)(r0 + rb + nb * 32 + 4 * h) * ldy + mblk * 32 + nn;
    const int lim = n - rb - nb * 32 - 4 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2);
      if (row < lim) dst[(size_t)row * ldy] = acc[r];
    }
    __syncthreads();  // xs is rewritten for the next tile
  }
}

// dT[c] = X[c]^T dY[c] for kd = 64 on the matrix pipe: one workgroup per cloud, 64-row slabs of both operands staged
// row-major (the contraction runs over rows, as in k_gemm_tn), wave -> one 32 x 32 block of the 64 x 64 result
__global__ __launch_bounds__(256) void k_cloud_matmul64_bwd_t(const float* __restrict__ X, int ldx,
                                                              const float* __restrict__ dY, int ldy,
                                                              float* __restrict__ dT, int B, int N, int M) {
  __shared__ __attribute__((aligned(16))) float xs[64 * 64];
  __shared__ __attribute__((aligned(16))) float ys[64 * 64];
  const int c = blockIdx.x;
  int r0, n;
  cloud_rows(c, B, N, M, r0, n);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ib = wave >> 1, jb = wave & 1, i = lane & 31, h = lane >> 5;
  f32x16 acc = zero16();
  const int row = tid >> 2, c0 = (tid & 3) * 16;
  f32x4 vx[4], vy[4];
  auto fetch = [&](int rs) {
    const bool ok = rs + row < n;
    const size_t gr = (size_t)(r0 + rs + (ok ? row : 0));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      vx[u] = ok ? *reinterpret_cast<const f32x4*>(X + gr * ldx + c0 + 4 * u) : z;
      vy[u] = ok ? *reinterpret_cast<const f32x4*>(dY + gr * ldy + c0 + 4 * u) : z;
    }
  };
  fetch(0);
  for (int rs = 0; rs < n; rs += 64) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      *reinterpret_cast<f32x4*>(xs + row * 64 + c0 + 4 * u) = vx[u];
      *reinterpret_cast<f32x4*>(ys + row * 64 + c0 + 4 * u) = vy[u];
    }
    __syncthreads();
    if (rs + 64 < n) fetch(rs + 64);
    const float* xa = xs + h * 64 + ib * 32 + i;
    const float* yb = ys + h * 64 + jb * 32 + i;
#pragma unroll 8
    for (int t = 0; t < 32; ++t) acc = mfma32(xa[t * 128], yb[t * 128], acc);  // rows 2t (h = 0), 2t + 1 (h = 1)
  }
  // D[row = i][col = j]: lane holds col j = lane&31, rows (reg&3) + 8(reg>>2) + 4h
  float* out = dT + (size_t)c * 4096;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg)
    out[(ib * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * 64 + jb * 32 + i] = acc[reg];
}

// dT[c][i][j] = sum_r X[r][i] dY[r][j] over the rows of cloud c
template <int KD>
__global__ __launch_bounds__(256) void k_cloud_matmul_bwd_t(const float* __restrict__ X, int ldx,
                                                            const float* __restrict__ dY, int ldy,
                                                            float* __restrict__ dT, int B, int N, int M) {
  __shared__ float xs[32 * KD], ys[32 * KD];
  const int c = blockIdx.x;
  int r0, n;
  cloud_rows(c, B, N, M, r0, n);
  constexpr int PER = (KD * KD + 255) / 256;
  float acc[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) acc[u] = 0.f;
  for (int rs = 0; rs < n; rs += 32) {
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * KD; e += 256) {
      const int row = e / KD, i = e % KD;
      const bool ok = rs + row < n;
      xs[e] = ok ? X[(size_t)(r0 + rs + row) * ldx + i] : 0.f;
      ys[e] = ok ? dY[(size_t)(r0 + rs + row) * ldy + i] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int o = threadIdx.x + 256 * u;
      if (o < KD * KD) {
        const int i = o / KD, j = o % KD;
        float s = acc[u];
        for (int row = 0; row < 32; ++row) s = fmaf(xs[row * KD + i], ys[row * KD + j], s);
        acc[u] = s;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int o = threadIdx.x + 256 * u;
    if (o < KD * KD) dT[(size_t)c * KD * KD + o] = acc[u];
  }
}

// ------------------------------------------------------------------------------------------------
// elementwise
// ------------------------------------------------------------------------------------------------
// out[r][:] = a[r][:] + b[r][:] + (r < Rc ? c[r][:] : 0), any of a / b / c null (= zeros): the gradient of a tensor with
// several consumers, one of which reads only its first Rc rows (train_ops._Hub) - autograd's own route is a zero-fill and a
// copy for the slice plus one add per further consumer
__global__ void k_sum_rows(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int ldc,
                           float* __restrict__ out, size_t n, size_t nc, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = a ? a[i] : 0.f;
  if (b) v += b[i];
  if (c && i < nc) v += c[(i / K) * ldc + (i % K)];
  out[i] = v;
}

__global__ void k_relu_bwd(const float* __restrict__ dY, const float* __restrict__ Y, float* __restrict__ dX, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dX[i] = Y[i] > 0.f ? dY[i] : 0.f;
}

__device__ __forceinline__ float gelu_grad(float v) {
  // d/dv [0.5 v (1 + erf(v/sqrt2))] = 0.5 (1 + erf(v/sqrt2)) + v * exp(-v^2/2) / sqrt(2 pi)
  return fmaf(v * 0.39894228040143267794f, __expf(-0.5f * v * v), gelu_cdf(v));
}
// LP = the autocast kernels on bf16 rows (HB instances): Phi from the P3 / Q3 rational of the bf16-operand inference path
// (catre_device.h gelu_erf_lp: max error 1.7e-5, against the 2e-3 relative of rounding the result to bf16) - four FMAs less
// per evaluation in kernels that are VALU-bound on exactly this (r04_rot_l1_bwd_bf_phases.txt).  The fp32-row instances keep
// the fp32-accurate rational.
template <bool LP>
__device__ __forceinline__ float gelu_fwd_t(float v) {
  return LP ? gelu_erf_lp(v) : gelu_erf(v);
}
template <bool LP>
__device__ __forceinline__ float gelu_grad_t(float v) {
  return fmaf(v * 0.39894228040143267794f, __expf(-0.5f * v * v), LP ? gelu_cdf_lp(v) : gelu_cdf(v));
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(32, 256) + GELU over the P points of an object (rows object-major, 256 channels)
//   stats  : per (object, group) mean / rstd over P x 8 values       (two-pass, one workgroup per (object, group))
//   forward: a = gelu(y*sc + sh)
//   backward (dy from da): dyhat = da*gelu'(yhat); dxhat = dyhat*gamma;
//              dy = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat*xhat));  dgamma, dbeta column sums
// ------------------------------------------------------------------------------------------------
// GroupNorm(32, 256) statistics over [P points x 8 channels] per (object, group), in two steps that read whole
// 1 KiB rows (the one-workgroup-per-group version below fetched 32 B of every row per workgroup: 1.7 TB/s):
//   k_gnp_stats_chunk  (object, chunk of 128 rows): per group n, mean, M2 from sums shifted by the chunk's first value
//   k_gnp_stats_final  per object: Chan merge of the chunks in order -> (mean, rstd)
#define GNS_CH 128
__global__ __launch_bounds__(256) void k_gnp_stats_chunk(const float* __restrict__ Y, float* __restrict__ part /*[B][nch][32][2]*/,
                                                         int P, int nch) {
  __shared__ float rs[4][32], rq[4][32];
  const int obj = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
  const int c4 = tid & 63, rl = tid >> 6, grp = c4 >> 1;
  const int p0 = chunk * GNS_CH, p1 = min(P, p0 + GNS_CH);
  const float* base = Y + ((size_t)obj * P) * 256;
  const float shift = base[(size_t)p0 * 256 + grp * 8];
  float s = 0.f, q = 0.f;
  for (int p = p0 + rl; p < p1; p += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * 256 + c4 * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float d = v[u] - shift;
      s += d;
      q = fmaf(d, d, q);
    }
  }
  s += __shfl_xor(s, 1);  // the group's other 4 channels
  q += __shfl_xor(q, 1);
  if ((c4 & 1) == 0) {
    rs[rl][grp] = s;
    rq[rl][grp] = q;
  }
  __syncthreads();
  if (tid < 32) {
    const float S = (rs[0][tid] + rs[1][tid]) + (rs[2][tid] + rs[3][tid]);
    const float Q = (rq[0][tid] + rq[1][tid]) + (rq[2][tid] + rq[3][tid]);
    const float n = 8.f * (float)(p1 - p0);
    const float sh = base[(size_t)p0 * 256 + tid * 8];
    float* o = part + (((size_t)obj * nch + chunk) * 32 + tid) * 2;
    o[0] = sh + S / n;       // chunk mean
    o[1] = Q - S * S / n;    // chunk M2
  }
}

#define GNP_FINAL_MAXCH 128  // chunks of an object staged in LDS (P / 64 <= 128: clouds of up to 8192 points in total)
__global__ __launch_bounds__(256) void k_gnp_stats_final(const float* __restrict__ part, float* __restrict__ stat, int P,
                                                         int nch, int rows_per_chunk) {
  // the object's partials are staged in LDS with coalesced loads, all in flight at once; the merge itself is a serial chain
  // per group and was a chain of nch dependent L2 round trips from global memory (16 us for 32 chunks; k_gn_finalize does
  // the same for the inference path).  Same operations in the same order: same bits.
  __shared__ float sp[GNP_FINAL_MAXCH * 64];
  const int obj = blockIdx.x, g = threadIdx.x;
  const bool staged = nch <= GNP_FINAL_MAXCH;
  const float* src = part + (size_t)obj * nch * 64;
  if (staged) {
    // 256 threads, eight loads each requested before the first LDS store (64 threads in a plain loop were 32 dependent
    // round trips: 15-19 us of a kernel whose arithmetic is 1 us)
    const int n = nch * 64;
    for (int i0 = 0; i0 < n; i0 += 2048) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256 + g;
        v[u] = i < n ? src[i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256 + g;
        if (i < n) sp[i] = v[u];
      }
    }
    __syncthreads();
  }
  if (g >= 32) return;
  const float* base = staged ? sp : src;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int c = 0; c < nch; ++c) {
    const float nb = 8.f * (float)(min(P, (c + 1) * rows_per_chunk) - c * rows_per_chunk);
    const float* o = base + ((size_t)c * 32 + g) * 2;
    const float nn = n + nb, delta = o[0] - mean;
    mean += delta * (nb / nn);
    m2 += o[1] + delta * delta * (n * nb / nn);
    n = nn;
  }
  stat[((size_t)obj * 32 + g) * 2] = mean;
  stat[((size_t)obj * 32 + g) * 2 + 1] = 1.0f / sqrtf(m2 / n + 1e-5f);
}

__global__ __launch_bounds__(256) void k_gnp_stats(const float* __restrict__ Y, float* __restrict__ stat, int P) {
  __shared__ float red[8];
  const int obj = blockIdx.x, g = blockIdx.y;
  const float* base = Y + (size_t)obj * P * 256 + g * 8;
  const int n = P * 8;
  float s = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) s += base[(size_t)(e >> 3) * 256 + (e & 7)];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)n;
  float q = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const float d = base[(size_t)(e >> 3) * 256 + (e & 7)] - mean;
    q = fmaf(d, d, q);
  }
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float var = (red[4] + red[5] + red[6] + red[7]) / (float)n;
    stat[((size_t)obj * 32 + g) * 2] = mean;
    stat[((size_t)obj * 32 + g) * 2 + 1] = 1.0f / sqrtf(var + 1e-5f);
  }
}

template <bool HB = false>  // HB: Y and A are bf16 rows (ld_row4 / st_row4)
__global__ void k_gnp_gelu_fwd(const void* __restrict__ Y, const float* __restrict__ stat,
                               const float* __restrict__ gamma, const float* __restrict__ beta, void* __restrict__ A,
                               int P, size_t total4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int c4 = i & 63;  // float4 index within the 256-channel row
  const size_t row = i >> 6;
  const int obj = row / P, g = c4 >> 1;
  const float mean = stat[((size_t)obj * 32 + g) * 2], rstd = stat[((size_t)obj * 32 + g) * 2 + 1];
  const f32x4 y = HB ? ld_row4<HB>(Y, i) : reinterpret_cast<const f32x4*>(Y)[i];
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c4], be = reinterpret_cast<const f32x4*>(beta)[c4];
  f32x4 a;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float sc = rstd * ga[q];
    a[q] = gelu_fwd_t<HB>(fmaf(y[q], sc, be[q] - mean * sc));
  }
  st_row4<HB>(A, i, a);
}

// pass 1 of the backward: per (object, group) sums S1 = sum dxhat, S2 = sum dxhat*xhat, and per-(object, channel)
// partial dgamma / dbeta (summed over objects by k_reduce_splits)
#define GNP_CH 128  // rows per workgroup in the backward reduction pass
template <bool HB = false>  // HB: dA and Y are bf16 rows
__global__ __launch_bounds__(256) void k_gnp_bwd_sums(const void* __restrict__ dA, const void* __restrict__ Y,
                                                      const float* __restrict__ stat, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ sums_part,
                                                      float* __restrict__ dgb_part, int P) {
  // workgroup = (object, chunk of GNP_CH rows); wave = every fourth row, lane = 4 channels (16-byte loads, eight of them
  // in flight per lane: the one-channel-per-thread form of round 2 kept 256 B per wave in flight and ran at 4.4 TB/s).
  // Partials are merged in wave order here and in chunk order by the finalize kernels.
  __shared__ float red[4][4][256];
  const int obj = blockIdx.x, chunk = blockIdx.y, nch = gridDim.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = lane * 4, g = c0 >> 3;
  const float mean = stat[((size_t)obj * 32 + g) * 2], rstd = stat[((size_t)obj * 32 + g) * 2 + 1];
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[lane], be = reinterpret_cast<const f32x4*>(beta)[lane];
  f32x4 sc, sh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * ga[q];
    sh[q] = be[q] - mean * sc[q];
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, dga[4] = {0.f, 0.f, 0.f, 0.f}, dbe[4] = {0.f, 0.f, 0.f, 0.f};
  const int p0 = chunk * GNP_CH, p1 = min(P, p0 + GNP_CH);
  const size_t q0 = (size_t)obj * P * 64 + lane;  // row-quad index of (first row of the object, this lane's channels)
  for (int p = p0 + wave; p < p1; p += 16) {
    f32x4 yv[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = min(p + 4 * u, p1 - 1);
      yv[u] = ld_row4<HB>(Y, q0 + (size_t)pp * 64);
      dv[u] = ld_row4<HB>(dA, q0 + (size_t)pp * 64);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p + 4 * u >= p1) break;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = (yv[u][q] - mean) * rstd;
        const float dyh = dv[u][q] * gelu_grad_t<HB>(fmaf(yv[u][q], sc[q], sh[q]));
        dga[q] = fmaf(dyh, xh, dga[q]);
        dbe[q] += dyh;
        const float dxh = dyh * ga[q];
        s1[q] += dxh;
        s2[q] = fmaf(dxh, xh, s2[q]);
      }
    }
  }
  *reinterpret_cast<f32x4*>(&red[wave][0][c0]) = f32x4{s1[0], s1[1], s1[2], s1[3]};
  *reinterpret_cast<f32x4*>(&red[wave][1][c0]) = f32x4{s2[0], s2[1], s2[2], s2[3]};
  *reinterpret_cast<f32x4*>(&red[wave][2][c0]) = f32x4{dga[0], dga[1], dga[2], dga[3]};
  *reinterpret_cast<f32x4*>(&red[wave][3][c0]) = f32x4{dbe[0], dbe[1], dbe[2], dbe[3]};
  __syncthreads();
  const int ch = tid;
  float t1 = (red[0][0][ch] + red[1][0][ch]) + (red[2][0][ch] + red[3][0][ch]);
  float t2 = (red[0][1][ch] + red[1][1][ch]) + (red[2][1][ch] + red[3][1][ch]);
  // group = 8 consecutive channels = 8 consecutive lanes
  t1 += __shfl_xor(t1, 1);
  t1 += __shfl_xor(t1, 2);
  t1 += __shfl_xor(t1, 4);
  t2 += __shfl_xor(t2, 1);
  t2 += __shfl_xor(t2, 2);
  t2 += __shfl_xor(t2, 4);
  const size_t slot = (size_t)obj * nch + chunk;
  if ((ch & 7) == 0) {
    sums_part[(slot * 32 + (ch >> 3)) * 2] = t1;
    sums_part[(slot * 32 + (ch >> 3)) * 2 + 1] = t2;
  }
  dgb_part[(slot * 2) * 256 + ch] = (red[0][2][ch] + red[1][2][ch]) + (red[2][2][ch] + red[3][2][ch]);
  dgb_part[(slot * 2 + 1) * 256 + ch] = (red[0][3][ch] + red[1][3][ch]) + (red[2][3][ch] + red[3][3][ch]);
}

// sums[obj][32][2] = sum over chunks of sums_part[obj][chunk][32][2]
__global__ void k_gnp_bwd_sums_finalize(const float* __restrict__ sums_part, float* __restrict__ sums, int B, int nch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 64) return;
  const int obj = i / 64, e = i % 64;
  float s = 0.f;
#pragma unroll 8
  for (int c = 0; c < nch; ++c) s += sums_part[((size_t)obj * nch + c) * 64 + e];
  sums[i] = s;
}

__global__ void k_gnp_bwd_apply(const float* __restrict__ dA, const float* __restrict__ Y,
                                const float* __restrict__ stat, const float* __restrict__ sums,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                float* __restrict__ dY, int P, size_t total4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int c4 = i & 63;
  const size_t row = i >> 6;
  const int obj = row / P, g = c4 >> 1;
  const float mean = stat[((size_t)obj * 32 + g) * 2], rstd = stat[((size_t)obj * 32 + g) * 2 + 1];
  const float inv_m = 1.0f / (8.f * (float)P);
  const float m1 = sums[((size_t)obj * 32 + g) * 2] * inv_m, m2 = sums[((size_t)obj * 32 + g) * 2 + 1] * inv_m;
  const f32x4 y = reinterpret_cast<const f32x4*>(Y)[i], da = reinterpret_cast<const f32x4*>(dA)[i];
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c4], be = reinterpret_cast<const f32x4*>(beta)[c4];
  f32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float sc = rstd * ga[q];
    const float xh = (y[q] - mean) * rstd;
    const float dxh = da[q] * gelu_grad(fmaf(y[q], sc, be[q] - mean * sc)) * ga[q];
    o[q] = rstd * (dxh - m1 - xh * m2);
  }
  reinterpret_cast<f32x4*>(dY)[i] = o;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm + GELU + neck (Conv1d 256 -> rot_dim <= 3, conv_out_per_rot_head.py:134-137) as ONE op for training: the
// [R,256] activation between them is never written and its gradient never exists - d a[r][c] = sum_k dY3[r][k] Wn[k][c]
// is rebuilt from the three floats of its row wherever it is needed.  Per head that removes a 0.5 GiB store + three
// 0.5 GiB loads forward and backward, and the two padded GEMMs (a 3 x 256 weight gradient on 128 x 128 MFMA tiles).
// ------------------------------------------------------------------------------------------------
template <bool LP = false>
__device__ __forceinline__ void gelu_both(float v, float& g, float& dg) {
  const float cdf = LP ? gelu_cdf_lp(v) : gelu_cdf(v);
  g = v * cdf;  // the operation sequences of gelu_erf / gelu_grad on one evaluation of Phi
  dg = fmaf(v * 0.39894228040143267794f, __expf(-0.5f * v * v), cdf);
}

// workgroup = 64 consecutive rows of one object (P % 64 == 0); wave = row, lane = 4 channels; Y3 [R][3]
__global__ __launch_bounds__(256) void k_gnp_gelu_neck_fwd(const float* __restrict__ Y, const float* __restrict__ stat,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ Wn, const float* __restrict__ bn,
                                                           float* __restrict__ Y3, int P) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t r0 = (size_t)blockIdx.x * 64;
  const int obj = (int)(r0 / P), c0 = lane * 4;
  const float* st = stat + ((size_t)obj * 32 + (c0 >> 3)) * 2;
  const float mean = st[0], rstd = st[1];
  f32x4 sc, sh;
  float nk[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * gamma[c0 + q];
    sh[q] = beta[c0 + q] - mean * sc[q];
#pragma unroll
    for (int k = 0; k < 3; ++k) nk[k][q] = Wn[k * 256 + c0 + q];
  }
  const float b0 = bn ? bn[0] : 0.f, b1 = bn ? bn[1] : 0.f, b2 = bn ? bn[2] : 0.f;
  const float* src = Y + (r0 + wave) * 256 + c0;
  // four rows of this wave per step: the loads are requested together, the 12 row sums share one butterfly
  for (int it = 0; it < 4; ++it) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(16 * it + 4 * u) * 256));
    float t[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float z[4];
      gelu_affine4(v[u][0], v[u][1], v[u][2], v[u][3], sc, sh, z);
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(z[q]));  // scalar neck sums on purpose: see rot_out_body
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float a = nk[k][0] * z[0];
        a = fmaf(nk[k][1], z[1], a);
        a = fmaf(nk[k][2], z[2], a);
        a = fmaf(nk[k][3], z[3], a);
        asm volatile("" : "+v"(a));
        t[u][k] = a;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) t[u][k] = wave_sum(t[u][k]);
    if (lane < 4) {
      // lane u stores row u of the step (wave_sum leaves the totals in every lane)
      float o0 = t[0][0], o1 = t[0][1], o2 = t[0][2];
#pragma unroll
      for (int u = 1; u < 4; ++u)
        if (lane == u) {
          o0 = t[u][0];
          o1 = t[u][1];
          o2 = t[u][2];
        }
      float* dst = Y3 + (r0 + wave + 16 * it + 4 * lane) * 3;
      dst[0] = o0 + b0;
      dst[1] = o1 + b1;
      dst[2] = o2 + b2;
    }
  }
}

// The same forward that ALSO leaves what the GroupNorm-1 backward's reduction pass would have to recompute from Y.  The
// neck output feeds only conv_p (out[b][j] = sum_p wp[p] Y3[b,p,j] + b, conv_out_per_rot_head.py:138-140), so the gradient
// of every row is dY3[b,p,:] = wp[p] dout[b,:] and, with u[b][c] = sum_j Wn[j][c] dout[b][j], everything that pass sums over
// the points factors through three per-(object, channel) moments that do not depend on dout:
//   S1 = sum_p wp[p] gelu'(z[p,c]),   S2 = sum_p wp[p] gelu'(z[p,c]) xhat[p,c],   S3 = sum_p wp[p] gelu(z[p,c])
//   sum dxhat = gamma_c u S1,  sum dxhat xhat = gamma_c u S2,  dgamma_c = sum_b u S2,  dbeta_c = sum_b u S1,
//   dWn[j][c] = sum_b dout[b][j] S3
// Spart [tile = 64 rows][3][256]: this workgroup's share; k_neck_sums_from_s finishes them in the backward - which then
// never reads Y for the sums (k_gnp_neck_bwd_sums: a 0.5 GiB pass per head).
template <bool HB = false>  // HB: Y is bf16 rows
__global__ __launch_bounds__(256) void k_gnp_gelu_neck_fwd_s(const void* __restrict__ Y, const float* __restrict__ stat,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ Wn, const float* __restrict__ bn,
                                                             const float* __restrict__ wp, float* __restrict__ Y3,
                                                             float* __restrict__ Spart, int P) {
  __shared__ float red[4][3][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t r0 = (size_t)blockIdx.x * 64;
  const int obj = (int)(r0 / P), c0 = lane * 4;
  const int p0 = (int)(r0 - (size_t)obj * P);
  const float* st = stat + ((size_t)obj * 32 + (c0 >> 3)) * 2;
  const float mean = st[0], rstd = st[1];
  f32x4 sc, sh;
  float nk[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * gamma[c0 + q];
    sh[q] = beta[c0 + q] - mean * sc[q];
#pragma unroll
    for (int k = 0; k < 3; ++k) nk[k][q] = Wn[k * 256 + c0 + q];
  }
  const float b0 = bn ? bn[0] : 0.f, b1 = bn ? bn[1] : 0.f, b2 = bn ? bn[2] : 0.f;
  const size_t src4 = (r0 + wave) * 64 + lane;  // row-quad index
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, s3[4] = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < 4; ++it) {
    f32x4 v[4];
    float wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = ld_row4<HB>(Y, src4 + (size_t)(16 * it + 4 * u) * 64);
      wv[u] = wp[p0 + wave + 16 * it + 4 * u];
    }
    float t[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float z[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float dg;
        gelu_both<HB>(fmaf(v[u][q], sc[q], sh[q]), z[q], dg);
        const float wd = wv[u] * dg;
        s1[q] += wd;
        s2[q] = fmaf(wd, (v[u][q] - mean) * rstd, s2[q]);
        s3[q] = fmaf(wv[u], z[q], s3[q]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float a = nk[k][0] * z[0];
        a = fmaf(nk[k][1], z[1], a);
        a = fmaf(nk[k][2], z[2], a);
        a = fmaf(nk[k][3], z[3], a);
        t[u][k] = a;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) t[u][k] = wave_sum(t[u][k]);
    if (lane < 4) {
      float o0 = t[0][0], o1 = t[0][1], o2 = t[0][2];
#pragma unroll
      for (int u = 1; u < 4; ++u)
        if (lane == u) {
          o0 = t[u][0];
          o1 = t[u][1];
          o2 = t[u][2];
        }
      float* dst = Y3 + (r0 + wave + 16 * it + 4 * lane) * 3;
      dst[0] = o0 + b0;
      dst[1] = o1 + b1;
      dst[2] = o2 + b2;
    }
  }
  *reinterpret_cast<f32x4*>(&red[wave][0][c0]) = f32x4{s1[0], s1[1], s1[2], s1[3]};
  *reinterpret_cast<f32x4*>(&red[wave][1][c0]) = f32x4{s2[0], s2[1], s2[2], s2[3]};
  *reinterpret_cast<f32x4*>(&red[wave][2][c0]) = f32x4{s3[0], s3[1], s3[2], s3[3]};
  __syncthreads();
  float* o = Spart + (size_t)blockIdx.x * 768;
#pragma unroll
  for (int k = 0; k < 3; ++k) o[k * 256 + tid] = (red[0][k][tid] + red[1][k][tid]) + (red[2][k][tid] + red[3][k][tid]);
}

// backward of the above: the GroupNorm sums [B][32][2] and the per-object (dgamma, dbeta, dWn[0..2]) partials
// dgb [B][5][256] (slot layout of k_gnp_neck_bwd_sums with one chunk per object) from dout [B][3] and the tile moments.
__global__ __launch_bounds__(256) void k_neck_sums_from_s(const float* __restrict__ dout, const float* __restrict__ Spart,
                                                          const float* __restrict__ gamma, const float* __restrict__ Wn,
                                                          float* __restrict__ sums, float* __restrict__ dgb, int T) {
  const int obj = blockIdx.x, ch = threadIdx.x;
  const float* sp = Spart + (size_t)obj * T * 768 + ch;
  float S1 = 0.f, S2 = 0.f, S3 = 0.f;
#pragma unroll 8
  for (int t = 0; t < T; ++t) {  // tile order: fixed
    S1 += sp[(size_t)t * 768];
    S2 += sp[(size_t)t * 768 + 256];
    S3 += sp[(size_t)t * 768 + 512];
  }
  const float d0 = dout[obj * 3], d1 = dout[obj * 3 + 1], d2 = dout[obj * 3 + 2];
  const float u = fmaf(Wn[512 + ch], d2, fmaf(Wn[256 + ch], d1, Wn[ch] * d0));
  const float ga = gamma[ch];
  float s1 = ga * u * S1, s2 = ga * u * S2;
  s1 += __shfl_xor(s1, 1);
  s1 += __shfl_xor(s1, 2);
  s1 += __shfl_xor(s1, 4);
  s2 += __shfl_xor(s2, 1);
  s2 += __shfl_xor(s2, 2);
  s2 += __shfl_xor(s2, 4);
  if ((ch & 7) == 0) {
    sums[((size_t)obj * 32 + (ch >> 3)) * 2] = s1;
    sums[((size_t)obj * 32 + (ch >> 3)) * 2 + 1] = s2;
  }
  float* o = dgb + (size_t)obj * 5 * 256 + ch;
  o[0] = u * S2;
  o[256] = u * S1;
  o[512] = d0 * S3;
  o[768] = d1 * S3;
  o[1024] = d2 * S3;
}

// backward pass 1: k_gnp_bwd_sums with d a rebuilt from dY3, plus the neck's weight-gradient partials.
// dgb_part [slot][5][256] = dgamma, dbeta, dWn[0..2] of the chunk.
__global__ __launch_bounds__(256) void k_gnp_neck_bwd_sums(const float* __restrict__ dY3, const float* __restrict__ Y,
                                                           const float* __restrict__ stat, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ Wn,
                                                           float* __restrict__ sums_part, float* __restrict__ dgb_part,
                                                           int P) {
  // workgroup = (object, chunk of GNP_CH rows); wave = every fourth row, lane = 4 channels (16-byte loads: see
  // k_gnp_bwd_sums); partials merged in wave order
  __shared__ float red[4][7][256];
  const int obj = blockIdx.x, chunk = blockIdx.y, nch = gridDim.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = lane * 4, g = c0 >> 3;
  const float mean = stat[((size_t)obj * 32 + g) * 2], rstd = stat[((size_t)obj * 32 + g) * 2 + 1];
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[lane], be = reinterpret_cast<const f32x4*>(beta)[lane];
  const f32x4 w0 = reinterpret_cast<const f32x4*>(Wn)[lane], w1 = reinterpret_cast<const f32x4*>(Wn)[64 + lane],
              w2 = reinterpret_cast<const f32x4*>(Wn)[128 + lane];
  f32x4 sc, sh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * ga[q];
    sh[q] = be[q] - mean * sc[q];
  }
  float acc[7][4];
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[k][q] = 0.f;
  const int p0 = chunk * GNP_CH, p1 = min(P, p0 + GNP_CH);
  const f32x4* y = reinterpret_cast<const f32x4*>(Y + (size_t)obj * P * 256) + lane;
  const float* d3 = dY3 + (size_t)obj * P * 3;
  for (int p = p0 + wave; p < p1; p += 16) {
    f32x4 yv[4];
    float d[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = min(p + 4 * u, p1 - 1);
      yv[u] = __builtin_nontemporal_load(y + (size_t)pp * 64);
      d[u][0] = d3[pp * 3];  // uniform over the wave
      d[u][1] = d3[pp * 3 + 1];
      d[u][2] = d3[pp * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p + 4 * u >= p1) break;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = (yv[u][q] - mean) * rstd;
        float a, dg;
        gelu_both(fmaf(yv[u][q], sc[q], sh[q]), a, dg);
        const float da = fmaf(w2[q], d[u][2], fmaf(w1[q], d[u][1], w0[q] * d[u][0]));
        const float dyh = da * dg;
        acc[2][q] = fmaf(dyh, xh, acc[2][q]);  // dgamma
        acc[3][q] += dyh;                      // dbeta
        const float dxh = dyh * ga[q];
        acc[0][q] += dxh;
        acc[1][q] = fmaf(dxh, xh, acc[1][q]);
        acc[4][q] = fmaf(d[u][0], a, acc[4][q]);  // dWn rows
        acc[5][q] = fmaf(d[u][1], a, acc[5][q]);
        acc[6][q] = fmaf(d[u][2], a, acc[6][q]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) *reinterpret_cast<f32x4*>(&red[wave][k][c0]) = f32x4{acc[k][0], acc[k][1], acc[k][2], acc[k][3]};
  __syncthreads();
  const int ch = tid;
  float t[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) t[k] = (red[0][k][ch] + red[1][k][ch]) + (red[2][k][ch] + red[3][k][ch]);
  float s1 = t[0], s2 = t[1];
  s1 += __shfl_xor(s1, 1);
  s1 += __shfl_xor(s1, 2);
  s1 += __shfl_xor(s1, 4);
  s2 += __shfl_xor(s2, 1);
  s2 += __shfl_xor(s2, 2);
  s2 += __shfl_xor(s2, 4);
  const size_t slot = (size_t)obj * nch + chunk;
  if ((ch & 7) == 0) {
    sums_part[(slot * 32 + (ch >> 3)) * 2] = s1;
    sums_part[(slot * 32 + (ch >> 3)) * 2 + 1] = s2;
  }
  float* o = dgb_part + slot * 5 * 256 + ch;
  o[0] = t[2];
  o[256] = t[3];
  o[512] = t[4];
  o[768] = t[5];
  o[1024] = t[6];
}

// backward pass 2: dY = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat * xhat)) with d a rebuilt from dY3
__global__ void k_gnp_neck_bwd_apply(const float* __restrict__ dY3, const float* __restrict__ Y,
                                     const float* __restrict__ stat, const float* __restrict__ sums,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ Wn, float* __restrict__ dY, int P, size_t total4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int c4 = i & 63;
  const size_t row = i >> 6;
  const int obj = row / P, g = c4 >> 1;
  const float mean = stat[((size_t)obj * 32 + g) * 2], rstd = stat[((size_t)obj * 32 + g) * 2 + 1];
  const float inv_m = 1.0f / (8.f * (float)P);
  const float m1 = sums[((size_t)obj * 32 + g) * 2] * inv_m, m2 = sums[((size_t)obj * 32 + g) * 2 + 1] * inv_m;
  const f32x4 y = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Y) + i);
  const float d0 = dY3[row * 3], d1 = dY3[row * 3 + 1], d2 = dY3[row * 3 + 2];
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c4], be = reinterpret_cast<const f32x4*>(beta)[c4];
  const f32x4 w0 = reinterpret_cast<const f32x4*>(Wn)[c4], w1 = reinterpret_cast<const f32x4*>(Wn)[64 + c4],
              w2 = reinterpret_cast<const f32x4*>(Wn)[128 + c4];
  f32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float sc = rstd * ga[q];
    const float xh = (y[q] - mean) * rstd;
    const float da = fmaf(w2[q], d2, fmaf(w1[q], d1, w0[q] * d0));
    const float dxh = da * gelu_grad(fmaf(y[q], sc, be[q] - mean * sc)) * ga[q];
    o[q] = rstd * (dxh - m1 - xh * m2);
  }
  reinterpret_cast<f32x4*>(dY)[i] = o;
}

// ------------------------------------------------------------------------------------------------
// Backward of a rot head's layer 0 block (per-cloud-bias linear 64 -> 256, GroupNorm, GELU; conv_out_per_rot_head.py:
// 126-131) behind the GroupNorm sums, in ONE pass over (dA, Y): a workgroup walks the 64-row tiles of one cloud, rebuilds
// dY = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)) straight into LDS (the [R,256] gradient never reaches
// HBM) and takes from that tile
//   waves 0-3: dX tile [64 x 64] = dY W          (one 32 x 32 block each, K = 256 from packed W^T fragments)
//   waves 4-7: dW [256 x 64] += dY^T X           (four 32 x 32 blocks each, contraction over the tile's rows)
//   all waves: the cloud's bias gradient          (column sums of the values they staged)
// instead of k_gnp_bwd_apply (2 loads + 1 store of [R,256]) + k_rowbias_bwd + dgrad + wgrad (one load of it each).
// Rows object-major, N and M multiples of 64.  Clouds: blockIdx.x < B observed, else prior (the row order of bias2d).
// ------------------------------------------------------------------------------------------------
#define L0B_LDP 96  // pitch of the staged X tile: rows 2t / 2t+1 land in different bank halves
__global__ __launch_bounds__(512) void k_rot_l0_bwd(const float* __restrict__ dA, const float* __restrict__ Y,
                                                    const float* __restrict__ stat, const float* __restrict__ sums,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ X, int ldx, const f32x4* __restrict__ WpT,
                                                    float* __restrict__ dX, int lddx, float* __restrict__ wpart,
                                                    float* __restrict__ dbias, int B, int N, int M, int acc_dx) {
  __shared__ __attribute__((aligned(16))) float dys[TP * 256];
  __shared__ __attribute__((aligned(16))) float pfs[TP * L0B_LDP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = N + M;
  const int obj = blockIdx.x % B, prior = blockIdx.x / B;
  const int ntile = (prior ? M : N) / TP;
  const size_t row0 = (size_t)obj * P + (prior ? N : 0);
  // staging: wave -> rows wave + 8u of the tile, lane -> float4 column (channels 4 lane .. +3, GroupNorm group lane>>1)
  const int grp = lane >> 1;
  const float mean = stat[((size_t)obj * 32 + grp) * 2], rstd = stat[((size_t)obj * 32 + grp) * 2 + 1];
  const float inv_m = 1.0f / (8.f * (float)P);
  const float m1 = sums[((size_t)obj * 32 + grp) * 2] * inv_m, m2 = sums[((size_t)obj * 32 + grp) * 2 + 1] * inv_m;
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[lane], be = reinterpret_cast<const f32x4*>(beta)[lane];
  f32x4 sc, sh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * ga[q];
    sh[q] = be[q] - mean * sc[q];
  }
  f32x4 vd[8], vy[8], vp[2];
  // X rows of this cloud: object-major like everything else here, or (acc_dx & 2) cloud-major - pointfeat as the encoder
  // wrote it, so that the training forward needs no object-major copy of it
  const size_t xrow0 = (acc_dx & 2) ? (prior ? (size_t)B * N + (size_t)obj * M : (size_t)obj * N) : row0;
  auto fetch = [&](int t) {
    const size_t r = row0 + (size_t)t * TP;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t o = ((r + wave + 8 * u) * 64 + lane);
      vd[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dA) + o);
      vy[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Y) + o);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 512 * u;
      vp[u] = *reinterpret_cast<const f32x4*>(X + (xrow0 + (size_t)t * TP + (e >> 4)) * ldx + 4 * (e & 15));
    }
  };
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  f32x16 wacc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a) wacc[a][0] = wacc[a][1] = zero16();
  // weight-gradient waves: j-blocks 2q, 2q+1 (q = wave - 4) x both k-blocks.  Element (row, col) of the swizzled dY image
  // sits at row * 256 + (((col >> 2) ^ (row & 15)) << 2) + (col & 3); with row = 2t + h the XOR is (chunk ^ h) ^ (2t & 15):
  // eight per-lane offsets cover the sweep, and the second j-block (chunk ^ 8) reuses them.
  const int i = lane & 31, h = lane >> 5;
  int offA[8];
  {
    const int col = ((wave & 3) * 2) * 32 + i;
#pragma unroll
    for (int v = 0; v < 8; ++v) offA[v] = h * 256 + ((((col >> 2) ^ h) ^ (2 * v)) << 2) + (col & 3);
  }
  for (int t = 0; t < ntile; ++t) {
    // (no prefetch across the MFMA phase: 72 staging registers next to 64 accumulators and the sweep's rings spill)
    fetch(t);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      f32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = (vy[u][q] - mean) * rstd;
        const float dxh = vd[u][q] * gelu_grad(fmaf(vy[u][q], sc[q], sh[q])) * ga[q];
        o[q] = rstd * (dxh - m1 - xh * m2);  // the operation sequence of k_gnp_bwd_apply
        cs[q] += o[q];
      }
      *reinterpret_cast<f32x4*>(dys + swz_off(wave + 8 * u, lane, 256)) = o;
      __builtin_amdgcn_sched_barrier(0);  // one row at a time: interleaving all eight costs ~100 VGPRs of temporaries
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 512 * u;
      *reinterpret_cast<f32x4*>(pfs + (e >> 4) * L0B_LDP + 4 * (e & 15)) = vp[u];
    }
    __syncthreads();
    if (wave < 4) {
      const int mbk = wave & 1, nb = wave >> 1;
      f32x16 acc[1][1];
      acc[0][0] = zero16();
      GemmPipe<1, 1, false, true, 32, 2, 1> gp;
      // opaque per tile: otherwise the 32 fragment addresses of the sweep are loop-invariant, get hoisted out of the tile
      // loop as 64-bit pairs and spill
      unsigned lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      gp.prefetch(WpT + ((size_t)mbk * 32) * 64 + lane_o, 0);
      gp.run(acc, dys + nb * 32 * 256, 256, lane);
      // (the swapped orientation - whole-line stores, as in k_rot_l1_bwd - needs 16 store addresses here and spills; dX is
      // 134 MB per head, its 16-byte stores are not what bounds this kernel)
      float* o = dX + (row0 + (size_t)t * TP + nb * 32 + i) * lddx + mbk * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[0][0][4 * g], acc[0][0][4 * g + 1], acc[0][0][4 * g + 2], acc[0][0][4 * g + 3]};
        if (acc_dx & 1) v += *reinterpret_cast<const f32x4*>(o + 8 * g);  // second head of a pair: dX += (the heads share X)
        *reinterpret_cast<f32x4*>(o + 8 * g) = v;
      }
    } else {
      const float* pb = pfs + h * L0B_LDP + i;
      // operands two steps ahead of their use, pinned (left alone hipcc hoists all 128 LDS reads above the MFMAs)
      float a0[3], a1[3], b0[3], b1[3];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        a0[d] = dys[2 * d * 256 + offA[d & 7]];
        a1[d] = dys[2 * d * 256 + offA[(d & 7) ^ 4]];
        b0[d] = pb[2 * d * L0B_LDP];
        b1[d] = pb[2 * d * L0B_LDP + 32];
      }
#pragma unroll
      for (int tt = 0; tt < TP / 2; ++tt) {
        if (tt + 2 < TP / 2) {
          const int n = tt + 2;
          a0[n % 3] = dys[2 * n * 256 + offA[n & 7]];
          a1[n % 3] = dys[2 * n * 256 + offA[(n & 7) ^ 4]];
          b0[n % 3] = pb[2 * n * L0B_LDP];
          b1[n % 3] = pb[2 * n * L0B_LDP + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        wacc[0][0] = mfma32(a0[tt % 3], b0[tt % 3], wacc[0][0]);
        wacc[0][1] = mfma32(a0[tt % 3], b1[tt % 3], wacc[0][1]);
        wacc[1][0] = mfma32(a1[tt % 3], b0[tt % 3], wacc[1][0]);
        wacc[1][1] = mfma32(a1[tt % 3], b1[tt % 3], wacc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  if (wave >= 4) {
    float* out = wpart + (size_t)blockIdx.x * (256 * 64);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int j = ((wave & 3) * 2 + a) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          out[(size_t)j * 64 + kb * 32 + i] = wacc[a][kb][reg];
        }
  }
  // bias gradient of the cloud: eight row slices (one per wave) of 256 column sums, merged in wave order
  *reinterpret_cast<f32x4*>(dys + wave * 256 + 4 * lane) = f32x4{cs[0], cs[1], cs[2], cs[3]};
  __syncthreads();
  if (tid < 256) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += dys[w * 256 + tid];
    dbias[(size_t)blockIdx.x * 256 + tid] = sum;
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of a rot head's SECOND block (256 -> 256 linear, GroupNorm, GELU, neck; conv_out_per_rot_head.py:129-137)
// behind the GroupNorm sums, in one pass: a workgroup walks 64-row tiles of one object, rebuilds the linear's output
// gradient dY from (dY3, Y) into LDS - the k_gnp_neck_bwd_apply arithmetic - stages the layer's input tile A next to it,
// and takes dA = dY W (64 x 256, K = 256) and dW += dY^T A (256 x 256, contraction over the tile's rows) from the pair:
// dY [R,256] is never written, nor read back by a dgrad and a wgrad launch.  256 threads = one wave per SIMD with the
// whole register file: 256 weight-gradient accumulators (sixteen 32 x 32 blocks) + 64 for the data gradient per wave.
// LDS images are row-major with pitches 260 / 288 floats (16-byte fragment reads of the dgrad and the 4-byte operand
// reads of the wgrad both conflict-free).
// ------------------------------------------------------------------------------------------------
#define L1B_LDY 260
#define L1B_LDA 288
__global__ __launch_bounds__(256) void k_rot_l1_bwd(const float* __restrict__ dY3, const float* __restrict__ Y,
                                                    const float* __restrict__ stat, const float* __restrict__ sums,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ Wn, const float* __restrict__ A,
                                                    const f32x4* __restrict__ WpT, float* __restrict__ dA,
                                                    float* __restrict__ part /*[wg][256*256 + 256]*/, int P, int tpw,
                                                    unsigned long long* __restrict__ trace = nullptr) {
#define L1B_STAMP(i)                                                                                       \
  do {                                                                                                     \
    if (CATRE_TRACE_ON && trace && (threadIdx.x & 63) == 0)                                                \
      trace[(((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 4 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dys = lds;                  // [64][260]
  float* as = lds + TP * L1B_LDY;    // [64][288]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int obj = blockIdx.x, T = P / TP;
  const int t0 = blockIdx.y * tpw, t1 = min(T, t0 + tpw);
  const int grp = lane >> 1;
  const float mean = stat[((size_t)obj * 32 + grp) * 2], rstd = stat[((size_t)obj * 32 + grp) * 2 + 1];
  const float inv_m = 1.0f / (8.f * (float)P);
  const float m1 = sums[((size_t)obj * 32 + grp) * 2] * inv_m, m2 = sums[((size_t)obj * 32 + grp) * 2 + 1] * inv_m;
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[lane], be = reinterpret_cast<const f32x4*>(beta)[lane];
  const f32x4 w0 = reinterpret_cast<const f32x4*>(Wn)[lane], w1 = reinterpret_cast<const f32x4*>(Wn)[64 + lane],
              w2 = reinterpret_cast<const f32x4*>(Wn)[128 + lane];
  f32x4 sc, sh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * ga[q];
    sh[q] = be[q] - mean * sc[q];
  }
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  f32x16 wacc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) wacc[a][kb] = zero16();
  const int i = lane & 31, h = lane >> 5;
  // staging: wave -> rows wave + 4 j (j = 0..15), lane -> float4 column, in four batches of four rows.  Two batches of
  // (Y, A) rows are in flight while a batch is being transformed: a batch is requested when the batch two before it has
  // been consumed, and the first batch of the NEXT tile before this tile's MFMA phases - with one wave per SIMD nothing
  // else can run under a global round trip, so every load that is waited for right after its issue is ~2 us of idle CU.
  f32x4 vy[2][4], va[2][4];
  auto request_y = [&](int tt, int bb) {
    const size_t rr = (size_t)obj * P + (size_t)tt * TP;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      vy[bb & 1][u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Y) + (rr + wave + 4 * (4 * bb + u)) * 64 + lane);
  };
  auto request_a = [&](int tt, int bb) {
    const size_t rr = (size_t)obj * P + (size_t)tt * TP;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      va[bb & 1][u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A) + (rr + wave + 4 * (4 * bb + u)) * 64 + lane);
  };
  auto request = [&](int tt, int bb) {
    request_y(tt, bb);
    request_a(tt, bb);
  };
  if (t0 < t1) request_y(t0, 0);
  for (int t = t0; t < t1; ++t) {
    const size_t r0 = (size_t)obj * P + (size_t)t * TP;
    L1B_STAMP(0);
    request_a(t, 0);
    request(t, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      float d3v[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* d3 = dY3 + (r0 + wave + 4 * (4 * bb + u)) * 3;
        d3v[u][0] = d3[0];
        d3v[u][1] = d3[1];
        d3v[u][2] = d3[2];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = wave + 4 * (4 * bb + u);
        const float d0 = d3v[u][0], d1 = d3v[u][1], d2 = d3v[u][2];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float yv = vy[bb & 1][u][q];
          const float xh = (yv - mean) * rstd;
          const float da = fmaf(w2[q], d2, fmaf(w1[q], d1, w0[q] * d0));
          const float dxh = da * gelu_grad(fmaf(yv, sc[q], sh[q])) * ga[q];
          o[q] = rstd * (dxh - m1 - xh * m2);  // the operation sequence of k_gnp_neck_bwd_apply
          cs[q] += o[q];
        }
        *reinterpret_cast<f32x4*>(dys + row * L1B_LDY + 4 * lane) = o;
        *reinterpret_cast<f32x4*>(as + row * L1B_LDA + 4 * lane) = va[bb & 1][u];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (bb < 2) request(t, bb + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    L1B_STAMP(1);
    __syncthreads();
    L1B_STAMP(2);
    if (t + 1 < t1) request_y(t + 1, 0);  // (16 registers across the MFMA phases: the A rows too, or two batches, would spill)
    __builtin_amdgcn_sched_barrier(0);
    {  // dA tile = dY W: m-blocks 2 wave, 2 wave + 1 of the 256 input channels, both 32-row halves
      f32x16 acc[2][2];
      acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = zero16();
      // "swapped" MFMA orientation: a lane owns input channel (2 wave + mb) * 32 + i and 32 of the tile's rows, so the
      // stores below put 32 lanes on 128 consecutive bytes of a row - whole L2 lines (16-byte stores with the lanes along
      // the rows cost four times the L2 requests: see k_rot_l1<1, true>)
      GemmPipe<2, 2, true, false, 32, 2, 1> gp;
      unsigned lane_o = lane;  // opaque per tile: keeps the 64 fragment addresses of the sweep out of the tile loop's preheader
      asm volatile("" : "+v"(lane_o));
      gp.prefetch(WpT + ((size_t)(2 * wave) * 32) * 64 + lane_o, 32 * 64);
      gp.run(acc, dys, L1B_LDY, lane);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        float* o = dA + (r0 + 4 * h) * 256 + (2 * wave + mb) * 32 + i;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) st_stream(o + (nb * 32 + (r & 3) + 8 * (r >> 2)) * 256, acc[mb][nb][r]);
      }
    }
    L1B_STAMP(3);
    {  // dW += dY^T A: j-blocks 2 wave, 2 wave + 1 x all eight k-blocks; operands two steps ahead, pinned
      const float* pa = dys + h * L1B_LDY + (2 * wave) * 32 + i;
      const float* pb = as + h * L1B_LDA + i;
      float a0[3], a1[3], b[3][8];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        a0[d] = pa[2 * d * L1B_LDY];
        a1[d] = pa[2 * d * L1B_LDY + 32];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) b[d][kb] = pb[2 * d * L1B_LDA + kb * 32];
      }
#pragma unroll
      for (int tt = 0; tt < TP / 2; ++tt) {
        if (tt + 2 < TP / 2) {
          const int n = tt + 2;
          a0[n % 3] = pa[2 * n * L1B_LDY];
          a1[n % 3] = pa[2 * n * L1B_LDY + 32];
#pragma unroll
          for (int kb = 0; kb < 8; ++kb) b[n % 3][kb] = pb[2 * n * L1B_LDA + kb * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          wacc[0][kb] = mfma32(a0[tt % 3], b[tt % 3][kb], wacc[0][kb]);
          wacc[1][kb] = mfma32(a1[tt % 3], b[tt % 3][kb], wacc[1][kb]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    L1B_STAMP(4);
    __syncthreads();
    L1B_STAMP(5);
  }
#undef L1B_STAMP
  float* out = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (256 * 256 + 256);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int j = (2 * wave + a) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        out[(size_t)j * 256 + kb * 32 + i] = wacc[a][kb][reg];
      }
  // bias gradient partial: four row slices (one per wave) of 256 column sums, merged in wave order
  *reinterpret_cast<f32x4*>(dys + wave * 256 + 4 * lane) = f32x4{cs[0], cs[1], cs[2], cs[3]};
  __syncthreads();
  out[256 * 256 + tid] = (dys[tid] + dys[256 + tid]) + (dys[512 + tid] + dys[768 + tid]);
}

// ------------------------------------------------------------------------------------------------
// k_rot_l1_bwd on the bf16 matrix pipe (torch.autocast: the backward of a bf16 linear runs in bf16 as well).  Same pass,
// same fp32 GroupNorm / GELU' arithmetic, same outputs; the two GEMMs take bf16 operands (fp32 accumulation):
//   * dY is written to LDS twice - row-major [64][256] in GemmPipeB's swizzled chunk layout (the dgrad contracts over the
//     256 output channels: 8 consecutive channels per lane) and TRANSPOSED [256 columns][8 chunks of 8 rows] in
//     k_gemm_tn_lp's tn_slot layout next to the transposed A tile (the wgrad contracts over the tile's rows, and
//     v_mfma_f32_32x32x16_bf16 wants 8 consecutive contraction indices per lane);
//   * a thread stages 4 columns x 8 consecutive rows per 32-row half (wave w: rows 8w..8w+7), in two batches of four rows:
//     row-major 8-byte writes per row, one 16-byte transposed chunk per column and half.
// 96 KiB of LDS, one wave per SIMD (256 weight-gradient accumulators per lane, as in the fp32 kernel).
// WpT: bf16 fragments of W^T (k_op_pack_bf, transpose = 1: rows = input channels, K = output channels).
// ------------------------------------------------------------------------------------------------
// HB: Y, A and dA are bf16 rows (the all-bf16 activations of train_ops._RotHeadLP) instead of fp32 ones.
#define L1L_IMG (TP * 32)  // 16-byte slots of one 64 x 256 bf16 image
template <bool HB>
__global__ __launch_bounds__(256) void k_rot_l1_bwd_bf(const float* __restrict__ dY3, const void* __restrict__ Y,
                                                       const float* __restrict__ stat, const float* __restrict__ sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ Wn, const void* __restrict__ A,
                                                       const u32x4* __restrict__ WpT, void* __restrict__ dA,
                                                       float* __restrict__ part /*[wg][256*256 + 256]*/, int P, int tpw,
                                                       unsigned long long* __restrict__ trace = nullptr) {
#define L1L_STAMP(i)                                                                                       \
  do {                                                                                                     \
    if (CATRE_TRACE_ON && trace && (threadIdx.x & 63) == 0)                                                \
      trace[(((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 4 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
  extern __shared__ __attribute__((aligned(16))) u32x4 ldsq[];
  u32x4* dys = ldsq;                // row-major, bf_off<32>(row, chunk)
  u32x4* dyt = ldsq + L1L_IMG;      // transposed, tn_slot(column, chunk of 8 rows)
  u32x4* ast = ldsq + 2 * L1L_IMG;  // transposed A tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int obj = blockIdx.x, T = P / TP;
  const int t0 = blockIdx.y * tpw, t1 = min(T, t0 + tpw);
  const int grp = lane >> 1;
  const float mean = stat[((size_t)obj * 32 + grp) * 2], rstd = stat[((size_t)obj * 32 + grp) * 2 + 1];
  const float inv_m = 1.0f / (8.f * (float)P);
  const float m1 = sums[((size_t)obj * 32 + grp) * 2] * inv_m, m2 = sums[((size_t)obj * 32 + grp) * 2 + 1] * inv_m;
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[lane], be = reinterpret_cast<const f32x4*>(beta)[lane];
  const f32x4 w0 = reinterpret_cast<const f32x4*>(Wn)[lane], w1 = reinterpret_cast<const f32x4*>(Wn)[64 + lane],
              w2 = reinterpret_cast<const f32x4*>(Wn)[128 + lane];
  f32x4 sc, sh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * ga[q];
    sh[q] = be[q] - mean * sc[q];
  }
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  f32x16 wacc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) wacc[a][kb] = zero16();
  const int i = lane & 31, h = lane >> 5;
  // batch bb of a tile: rows (bb >> 1) * 32 + 8 wave + 4 (bb & 1) + u, u = 0..3; two batches of (Y, A) rows in flight while
  // one is transformed, the first Y batch of the next tile across the MFMA phases (as in k_rot_l1_bwd)
  auto row_of = [&](int bb, int u) { return (bb >> 1) * 32 + 8 * wave + 4 * (bb & 1) + u; };
  // (bf16 rows, whole-tile ring of four batches: measured, no gain - the sweeps wait for weight fragments, not for rows)
  constexpr int RD = 2;
  typename RowQ<HB>::T vy[RD][4], va[RD][4];
  auto request_y = [&](int tt, int bb) {
    const size_t rr = (size_t)obj * P + (size_t)tt * TP;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      vy[bb % RD][u] = ld_rowq<HB>(Y, (rr + row_of(bb, u)) * 64 + lane);
  };
  auto request_a = [&](int tt, int bb) {
    const size_t rr = (size_t)obj * P + (size_t)tt * TP;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      va[bb % RD][u] = ld_rowq<HB>(A, (rr + row_of(bb, u)) * 64 + lane);
  };
  auto request = [&](int tt, int bb) {
    request_y(tt, bb);
    request_a(tt, bb);
  };
  u32x2* dys2 = reinterpret_cast<u32x2*>(dys);
  if (t0 < t1) request_y(t0, 0);
  for (int t = t0; t < t1; ++t) {
    const size_t r0 = (size_t)obj * P + (size_t)t * TP;
    L1L_STAMP(0);
    request_a(t, 0);
    request(t, 1);
    __builtin_amdgcn_sched_barrier(0);
    // opaque per tile: keeps the ~70 LDS addresses of a tile (row-major and transposed staging slots, fragment slots) out
    // of the tile loop's preheader, where they would sit in registers across all three phases
    unsigned lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    unsigned hy[4][2], ha[4][2];  // rows 0..3 of the chunk, packed, until rows 4..7 arrive
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      unsigned py[4][2];  // this batch's two row pairs per column
#pragma unroll
      for (int up = 0; up < 2; ++up) {  // a pair of rows at a time: their temporaries do not overlap the next pair's
        float d3v[2][3];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const float* d3 = dY3 + (r0 + row_of(bb, 2 * up + w)) * 3;
          d3v[w][0] = d3[0];
          d3v[w][1] = d3[1];
          d3v[w][2] = d3[2];
        }
        f32x4 o[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const int u = 2 * up + w, row = row_of(bb, u);
          const float d0 = d3v[w][0], d1 = d3v[w][1], d2 = d3v[w][2];
          const f32x4 y4 = rowq_f32(vy[bb % RD][u]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float yv = y4[q];
            const float xh = (yv - mean) * rstd;
            const float da = fmaf(w2[q], d2, fmaf(w1[q], d1, w0[q] * d0));
            const float dxh = da * gelu_grad_t<HB>(fmaf(yv, sc[q], sh[q])) * ga[q];
            o[w][q] = rstd * (dxh - m1 - xh * m2);  // the operation sequence of k_gnp_neck_bwd_apply
            cs[q] += o[w][q];
          }
          const u32x2 pr = {pack_bf2(o[w][0], o[w][1]), pack_bf2(o[w][2], o[w][3])};
          dys2[(row * 32 + ((lane_o >> 1) ^ (row & 15))) * 2 + (lane_o & 1)] = pr;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) py[q][up] = pack_bf2(o[0][q], o[1][q]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if ((bb & 1) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          hy[q][0] = py[q][0];
          hy[q][1] = py[q][1];
          ha[q][0] = rowq_pair(va[bb % RD][0], va[bb % RD][1], q);
          ha[q][1] = rowq_pair(va[bb % RD][2], va[bb % RD][3], q);
        }
      } else {
        const int chunk = (bb >> 1) * 4 + wave;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int slot = tn_slot(4 * (int)lane_o + q, chunk);
          dyt[slot] = u32x4{hy[q][0], hy[q][1], py[q][0], py[q][1]};
          ast[slot] = u32x4{ha[q][0], ha[q][1], rowq_pair(va[bb % RD][0], va[bb % RD][1], q),
                            rowq_pair(va[bb % RD][2], va[bb % RD][3], q)};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (bb < 2) request(t, bb + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    L1L_STAMP(1);
    __syncthreads();
    L1L_STAMP(2);
    if (t + 1 < t1) request_y(t + 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    {  // dA tile = dY W: input-channel blocks 2 wave, 2 wave + 1, both 32-row halves; a lane owns one input channel and
       // 32 of the tile's rows (whole-line stores, see k_rot_l1_bwd)
      f32x16 acc[2][2];
      acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = zero16();
      GemmPipeB<2, 2, true, 32, 2, 1> gp;  // (weight fragments four K-steps ahead: measured, no gain)
      gp.prefetch(WpT + ((size_t)(2 * wave) * 16) * 64 + lane_o, 16 * 64);
      gp.run(acc, dys, (int)lane_o);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        if constexpr (HB) {
          // bf16 rows: a lane pair (channels c, c + 1) exchanges one value per pair of registers (rows r, r + 1) - the even
          // lane stores both channels of row r, the odd lane of row r + 1: one 4-byte store per two results
          unsigned* o = reinterpret_cast<unsigned*>(dA) + ((r0 + 4 * h + (i & 1)) * 256 + (2 * wave + mb) * 32 + (i & ~1)) / 2;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const float t0 = acc[mb][nb][r], t1 = acc[mb][nb][r + 1];
              const float got = dpp_move<0xB1>((i & 1) ? t0 : t1);  // quad_perm [1,0,3,2]: the pair's other lane
              const unsigned w = (i & 1) ? pack_bf2(got, t1) : pack_bf2(t0, got);
              // (plain stores: the two 32-channel blocks of a wave are the two halves of one 128-byte line, written by
              // different instructions - streamed past the cache they leave as partial lines, 2x the kernel time)
              o[(size_t)(nb * 32 + (r & 3) + 8 * (r >> 2)) * 128] = w;
            }
        } else {
          float* o = reinterpret_cast<float*>(dA) + (r0 + 4 * h) * 256 + (2 * wave + mb) * 32 + i;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st_stream(o + (nb * 32 + (r & 3) + 8 * (r >> 2)) * 256, acc[mb][nb][r]);
        }
      }
    }
    L1L_STAMP(3);
    {  // dW += dY^T A: output-channel blocks 2 wave, 2 wave + 1 x all eight input-channel blocks; 16 rows per step.
       // tn_slot(32 kb + i, 2 ks + h) = 256 kb + (b0 ^ (2 (kb ^ ks) & 7)) with b0 = tn_slot(i, h): one per-lane base, the
       // block and the step only through compile-time (kb: wave-uniform) constants
      const unsigned io = lane_o & 31, ho = lane_o >> 5;
      const unsigned b0 = (io >> 1) * 16 + 8 * ((io ^ (io >> 2)) & 1) + ((ho ^ (io >> 1) ^ (io >> 4)) & 7);
      u32x4 fa[2][2], fb[2][8];
      auto frags = [&](int ks) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int jb = 2 * wave + a;
          fa[ks & 1][a] = dyt[jb * 256 + (b0 ^ ((2 * (jb ^ ks)) & 7))];
        }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) fb[ks & 1][kb] = ast[kb * 256 + (b0 ^ ((2 * (kb ^ ks)) & 7))];
      };
      frags(0);
#pragma unroll
      for (int ks = 0; ks < TP / 16; ++ks) {
        if (ks + 1 < TP / 16) frags(ks + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          wacc[0][kb] = mfma_bf(fa[ks & 1][0], fb[ks & 1][kb], wacc[0][kb]);
          wacc[1][kb] = mfma_bf(fa[ks & 1][1], fb[ks & 1][kb], wacc[1][kb]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    L1L_STAMP(4);
    __syncthreads();
    L1L_STAMP(5);
  }
#undef L1L_STAMP
  float* out = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (256 * 256 + 256);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int j = (2 * wave + a) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        out[(size_t)j * 256 + kb * 32 + i] = wacc[a][kb][reg];
      }
  // bias gradient partial (fp32 values, before the bf16 rounding): four row slices of 256 column sums, merged in wave order
  float* red = reinterpret_cast<float*>(ldsq);
  *reinterpret_cast<f32x4*>(red + wave * 256 + 4 * lane) = f32x4{cs[0], cs[1], cs[2], cs[3]};
  __syncthreads();
  out[256 * 256 + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
}

// ------------------------------------------------------------------------------------------------
// k_rot_l1_bwd in split mode: the one-pass structure of k_rot_l1_bwd_bf with every MFMA operand as hi + lo bf16
// and three products - fp32-grade dA and dW on the bf16 matrix pipe.  The hi and lo images of a 64-row tile would be
// 192 KiB, so a tile is taken as two 32-row HALVES (row-major dY hi / lo for the dgrad, transposed dY and A hi / lo for the
// wgrad: 6 x 16 KiB): stage half, dgrad (32 rows), wgrad (two 16-row steps), next half.  Transposed images of 4 chunks per
// column: tn4_slot (conflict-free for the staging writes - columns 4 apart, one chunk - and the fragment reads - 32
// consecutive columns, one chunk; found by search, profiles/ubench/tn4_swizzle.py).  Y, A, dA: fp32 rows.
// WpT: catre_op_pack_split of W^T (hi pack, then lo pack 256 * 256 / 8 u32x4 further).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn4_slot(int c, int chunk) {
  return (c >> 2) * 16 + (((c & 3) ^ ((c >> 4) & 3)) << 2) + ((chunk ^ (c >> 2)) & 3);
}
#define L1S_IMG (32 * 32)  // 16-byte slots of one 32 x 256 bf16 image
__global__ __launch_bounds__(256) void k_rot_l1_bwd_sp(const float* __restrict__ dY3, const float* __restrict__ Y,
                                                       const float* __restrict__ stat, const float* __restrict__ sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ Wn, const float* __restrict__ A,
                                                       const u32x4* __restrict__ WpT, float* __restrict__ dA,
                                                       float* __restrict__ part /*[wg][256*256 + 256]*/, int P, int tpw) {
  extern __shared__ __attribute__((aligned(16))) u32x4 ldsq[];
  u32x4* dysh = ldsq;                // row-major dY, hi / lo: bf_off<32>(row, chunk)
  u32x4* dysl = ldsq + L1S_IMG;
  u32x4* dyth = ldsq + 2 * L1S_IMG;  // transposed dY, hi / lo: tn4_slot(column, chunk of 8 rows)
  u32x4* dytl = ldsq + 3 * L1S_IMG;
  u32x4* asth = ldsq + 4 * L1S_IMG;  // transposed A tile, hi / lo
  u32x4* astl = ldsq + 5 * L1S_IMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int obj = blockIdx.x, T = P / TP;
  const int t0 = blockIdx.y * tpw, t1 = min(T, t0 + tpw);
  const int grp = lane >> 1;
  const float mean = stat[((size_t)obj * 32 + grp) * 2], rstd = stat[((size_t)obj * 32 + grp) * 2 + 1];
  const float inv_m = 1.0f / (8.f * (float)P);
  const float m1 = sums[((size_t)obj * 32 + grp) * 2] * inv_m, m2 = sums[((size_t)obj * 32 + grp) * 2 + 1] * inv_m;
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[lane], be = reinterpret_cast<const f32x4*>(beta)[lane];
  const f32x4 w0 = reinterpret_cast<const f32x4*>(Wn)[lane], w1 = reinterpret_cast<const f32x4*>(Wn)[64 + lane],
              w2 = reinterpret_cast<const f32x4*>(Wn)[128 + lane];
  f32x4 sc, sh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * ga[q];
    sh[q] = be[q] - mean * sc[q];
  }
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  f32x16 wacc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) wacc[a][kb] = zero16();
  const int i = lane & 31, h = lane >> 5;
  // batch bb (0..3) of a tile: rows (bb >> 1) * 32 + 8 wave + 4 (bb & 1) + u, u = 0..3 - half bb >> 1, the wave's 8-row chunk
  auto row_of = [&](int bb, int u) { return (bb >> 1) * 32 + 8 * wave + 4 * (bb & 1) + u; };
  f32x4 vy[2][4], va[2][4];
  auto request_y = [&](int tt, int bb) {
    const size_t rr = (size_t)obj * P + (size_t)tt * TP;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      vy[bb & 1][u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Y) + (rr + row_of(bb, u)) * 64 + lane);
  };
  auto request_a = [&](int tt, int bb) {
    const size_t rr = (size_t)obj * P + (size_t)tt * TP;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      va[bb & 1][u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A) + (rr + row_of(bb, u)) * 64 + lane);
  };
  auto request = [&](int tt, int bb) {
    request_y(tt, bb);
    request_a(tt, bb);
  };
  u32x2* dysh2 = reinterpret_cast<u32x2*>(dysh);
  u32x2* dysl2 = reinterpret_cast<u32x2*>(dysl);
  // hi / lo of a pair of fp32 values as packed bf16 pairs
  auto split2 = [](float a, float b, unsigned& hi, unsigned& lo) {
    hi = pack_bf2(a, b);
    lo = pack_bf2(a - bf_lo(hi), b - bf_hi(hi));
  };
  if (t0 < t1) request_y(t0, 0);
  for (int t = t0; t < t1; ++t) {
    const size_t r0 = (size_t)obj * P + (size_t)t * TP;
    request_a(t, 0);
    request(t, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      unsigned lane_o = lane;  // opaque per half: keeps the LDS addresses out of the loop preheaders (k_rot_l1_bwd_bf)
      asm volatile("" : "+v"(lane_o));
      unsigned hyh[4][2], hyl[4][2], hah[4][2], hal[4][2];  // rows 0..3 of the chunk until rows 4..7 arrive
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int bb = 2 * hh + b2;
        unsigned pyh[4][2], pyl[4][2];  // this batch's two row pairs per column, hi / lo
#pragma unroll
        for (int up = 0; up < 2; ++up) {  // a pair of rows at a time
          float d3v[2][3];
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const float* d3 = dY3 + (r0 + row_of(bb, 2 * up + w)) * 3;
            d3v[w][0] = d3[0];
            d3v[w][1] = d3[1];
            d3v[w][2] = d3[2];
          }
          f32x4 o[2];
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const int u = 2 * up + w, row = 8 * wave + 4 * b2 + u;  // row inside the half
            const float d0 = d3v[w][0], d1 = d3v[w][1], d2 = d3v[w][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float yv = vy[bb & 1][u][q];
              const float xh = (yv - mean) * rstd;
              const float da = fmaf(w2[q], d2, fmaf(w1[q], d1, w0[q] * d0));
              const float dxh = da * gelu_grad(fmaf(yv, sc[q], sh[q])) * ga[q];
              o[w][q] = rstd * (dxh - m1 - xh * m2);  // the operation sequence of k_gnp_neck_bwd_apply
              cs[q] += o[w][q];
            }
            unsigned ph0, pl0, ph1, pl1;
            split2(o[w][0], o[w][1], ph0, pl0);
            split2(o[w][2], o[w][3], ph1, pl1);
            const u32x2 ph = {ph0, ph1}, pl = {pl0, pl1};
            const int off = (row * 32 + ((lane_o >> 1) ^ (row & 15))) * 2 + (lane_o & 1);
            dysh2[off] = ph;
            dysl2[off] = pl;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) split2(o[0][q], o[1][q], pyh[q][up], pyl[q][up]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (b2 == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            hyh[q][0] = pyh[q][0];
            hyh[q][1] = pyh[q][1];
            hyl[q][0] = pyl[q][0];
            hyl[q][1] = pyl[q][1];
            split2(va[0][0][q], va[0][1][q], hah[q][0], hal[q][0]);
            split2(va[0][2][q], va[0][3][q], hah[q][1], hal[q][1]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int slot = tn4_slot(4 * (int)lane_o + q, wave);
            dyth[slot] = u32x4{hyh[q][0], hyh[q][1], pyh[q][0], pyh[q][1]};
            dytl[slot] = u32x4{hyl[q][0], hyl[q][1], pyl[q][0], pyl[q][1]};
            unsigned h2, l2, h3, l3;
            split2(va[1][0][q], va[1][1][q], h2, l2);
            split2(va[1][2][q], va[1][3][q], h3, l3);
            asth[slot] = u32x4{hah[q][0], hah[q][1], h2, h3};
            astl[slot] = u32x4{hal[q][0], hal[q][1], l2, l3};
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (bb < 2) request(t, bb + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      if (hh == 1 && t + 1 < t1) request_y(t + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      {  // dA half tile = dY W: input-channel blocks 2 wave, 2 wave + 1 x the half's 32 rows
        f32x16 acc[2][1];
        acc[0][0] = acc[1][0] = zero16();
        GemmPipeS<2, 1, true, 32, 1> gp;  // (two K-steps of weight fragments ahead: 16 more registers, spills)
        gp.prefetch(WpT + ((size_t)(2 * wave) * 16) * 64 + lane_o, 16 * 64, 256 * 256 / 8);
        gp.run(acc, dysh, dysl, (int)lane_o);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          float* o = dA + (r0 + 32 * hh + 4 * h) * 256 + (2 * wave + mb) * 32 + i;
#pragma unroll
          for (int r = 0; r < 16; ++r) st_stream(o + ((r & 3) + 8 * (r >> 2)) * 256, acc[mb][0][r]);
        }
      }
      {  // dW += dY^T A over the half's rows: 16 rows per step, three products per block
        const unsigned io = lane_o & 31, ho = lane_o >> 5;
        const unsigned b0 = (io >> 2) * 16 + (((io & 3) ^ (io >> 4)) << 2) + ((ho ^ (io >> 2)) & 3);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 fah[2], fal[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const int jb = 2 * wave + a;
            const unsigned sl = jb * 128 + (b0 ^ (((2 * jb) & 3) << 2) ^ (2 * ks));
            fah[a] = dyth[sl];
            fal[a] = dytl[sl];
          }
#pragma unroll
          for (int kq = 0; kq < 2; ++kq) {  // four input-channel blocks at a time (all eight: 64 fragment registers, spills)
            u32x4 fbh[4], fbl[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
              const int kb = 4 * kq + k4;
              const unsigned sl = kb * 128 + (b0 ^ (((2 * kb) & 3) << 2) ^ (2 * ks));
              fbh[k4] = asth[sl];
              fbl[k4] = astl[sl];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
              for (int a = 0; a < 2; ++a) {
                wacc[a][4 * kq + k4] = mfma_bf(fal[a], fbh[k4], wacc[a][4 * kq + k4]);
                wacc[a][4 * kq + k4] = mfma_bf(fah[a], fbl[k4], wacc[a][4 * kq + k4]);
                wacc[a][4 * kq + k4] = mfma_bf(fah[a], fbh[k4], wacc[a][4 * kq + k4]);
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __syncthreads();
    }
  }
  float* out = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (256 * 256 + 256);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int j = (2 * wave + a) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        out[(size_t)j * 256 + kb * 32 + i] = wacc[a][kb][reg];
      }
  float* red = reinterpret_cast<float*>(ldsq);
  *reinterpret_cast<f32x4*>(red + wave * 256 + 4 * lane) = f32x4{cs[0], cs[1], cs[2], cs[3]};
  __syncthreads();
  out[256 * 256 + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
}

// ------------------------------------------------------------------------------------------------
// k_rot_l0_bwd on the bf16 matrix pipe (autocast), built like k_rot_l1_bwd_bf: dY goes to LDS row-major (dgrad: dX = dY W,
// K = 256) and transposed next to the transposed X tile (wgrad: dW += dY^T X over the tile's rows); a thread stages the 8
// rows of wave w's chunk x 4 columns in two batches of four rows, the batch registers are re-requested for the NEXT tile as
// soon as a batch is transformed (the MFMA phase is ~1.5 k cycles here: the loads have to be in flight across the staging).
// Every thread also transposes two rows x four columns of the 64 x 64 X tile.  72 KiB of LDS.  WpT: bf16 fragments of W^T (rows = 64 input channels,
// K = 256 output channels).  Same outputs and flags as k_rot_l0_bwd.
// ------------------------------------------------------------------------------------------------
// HB: dA and Y are bf16 rows.
#define L0L_IMG (TP * 32)  // 16-byte slots of one 64 x 256 bf16 image
template <bool HB>
__global__ __launch_bounds__(512) void k_rot_l0_bwd_bf(const void* __restrict__ dA, const void* __restrict__ Y,
                                                       const float* __restrict__ stat, const float* __restrict__ sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ X, int ldx, const u32x4* __restrict__ WpT,
                                                       float* __restrict__ dX, int lddx, float* __restrict__ wpart,
                                                       float* __restrict__ dbias, int B, int N, int M, int acc_dx) {
  extern __shared__ __attribute__((aligned(16))) u32x4 ldsq[];
  u32x4* dys = ldsq;                // row-major, bf_off<32>(row, chunk)
  u32x4* dyt = ldsq + L0L_IMG;      // transposed, tn_slot(column, chunk of 8 rows)
  u32x4* xt = ldsq + 2 * L0L_IMG;   // transposed X tile: 64 columns x 8 chunks
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = N + M;
  const int obj = blockIdx.x % B, prior = blockIdx.x / B;
  const int ntile = (prior ? M : N) / TP;
  const size_t row0 = (size_t)obj * P + (prior ? N : 0);
  const size_t xrow0 = (acc_dx & 2) ? (prior ? (size_t)B * N + (size_t)obj * M : (size_t)obj * N) : row0;
  const int grp = lane >> 1;
  const float mean = stat[((size_t)obj * 32 + grp) * 2], rstd = stat[((size_t)obj * 32 + grp) * 2 + 1];
  const float inv_m = 1.0f / (8.f * (float)P);
  const float m1 = sums[((size_t)obj * 32 + grp) * 2] * inv_m, m2 = sums[((size_t)obj * 32 + grp) * 2 + 1] * inv_m;
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[lane], be = reinterpret_cast<const f32x4*>(beta)[lane];
  f32x4 sc, sh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * ga[q];
    sh[q] = be[q] - mean * sc[q];
  }
  // batch bb of a tile: rows 8 wave + 4 bb + u, u = 0..3
  typename RowQ<HB>::T vd[2][4], vy[2][4];
  f32x4 vx[2];
  auto request = [&](int t, int bb) {
    const size_t r = row0 + (size_t)t * TP + 8 * wave + 4 * bb;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      vd[bb][u] = ld_rowq<HB>(dA, (r + u) * 64 + lane);
      vy[bb][u] = ld_rowq<HB>(Y, (r + u) * 64 + lane);
    }
  };
  const int xc4 = tid & 15, xrp = tid >> 4;  // X tile: column quad and row pair (rows 2 xrp, 2 xrp + 1) of this thread
  auto request_x = [&](int t) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      vx[u] = *reinterpret_cast<const f32x4*>(X + (xrow0 + (size_t)t * TP + 2 * xrp + u) * ldx + 4 * xc4);
  };
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  f32x16 wacc[2];  // weight gradient: output-channel block `wave` x both input-channel blocks
  wacc[0] = wacc[1] = zero16();
  const int i = lane & 31, h = lane >> 5;
  u32x2* dys2 = reinterpret_cast<u32x2*>(dys);
  if (ntile > 0) {
    request(0, 0);
    request(0, 1);
    request_x(0);
  }
  for (int t = 0; t < ntile; ++t) {
    unsigned lane_o = lane;  // opaque per tile: keeps the tile's LDS addresses out of the loop preheader (k_rot_l1_bwd_bf)
    asm volatile("" : "+v"(lane_o));
    unsigned hy[4][4];  // the chunk's four row pairs per column, packed as they are produced
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
      for (int up = 0; up < 2; ++up) {  // a pair of rows at a time (a row's temporaries do not overlap the next one's)
        f32x4 o[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const int u = 2 * up + w, row = 8 * wave + 4 * bb + u;
          const f32x4 y4 = rowq_f32(vy[bb][u]), d4 = rowq_f32(vd[bb][u]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float yv = y4[q];
            const float xh = (yv - mean) * rstd;
            const float dxh = d4[q] * gelu_grad_t<HB>(fmaf(yv, sc[q], sh[q])) * ga[q];
            o[w][q] = rstd * (dxh - m1 - xh * m2);  // the operation sequence of k_gnp_bwd_apply
            cs[q] += o[w][q];
          }
          const u32x2 pr = {pack_bf2(o[w][0], o[w][1]), pack_bf2(o[w][2], o[w][3])};
          dys2[(row * 32 + ((lane_o >> 1) ^ (row & 15))) * 2 + (lane_o & 1)] = pr;
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) hy[q][2 * bb + up] = pack_bf2(o[0][q], o[1][q]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < ntile) request(t + 1, bb);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) dyt[tn_slot(4 * (int)lane_o + q, wave)] = u32x4{hy[q][0], hy[q][1], hy[q][2], hy[q][3]};
    {  // X tile transposed: rows 2 xrp, 2 xrp + 1 of column 4 xc4 + q are one dword of chunk xrp >> 2
      unsigned* xt32 = reinterpret_cast<unsigned*>(xt);
#pragma unroll
      for (int q = 0; q < 4; ++q) xt32[tn_slot(4 * xc4 + q, xrp >> 2) * 4 + (xrp & 3)] = pack_bf2(vx[0][q], vx[1][q]);
      if (t + 1 < ntile) request_x(t + 1);
    }
    __syncthreads();
    if (wave < 4) {  // dX tile [64 x 64] = dY W: one 32 x 32 block per wave; a lane owns one row and 16 of the block's channels
      const int mbk = wave & 1, nb = wave >> 1;
      f32x16 acc[1][1];
      acc[0][0] = zero16();
      GemmPipeB<1, 1, false, 32, 2, 1> gp;
      gp.prefetch(WpT + ((size_t)mbk * 16) * 64 + lane_o, 0);
      gp.run(acc, dys + nb * 32 * 32, (int)lane_o);
      float* o = dX + (row0 + (size_t)t * TP + nb * 32 + i) * lddx + mbk * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[0][0][4 * g], acc[0][0][4 * g + 1], acc[0][0][4 * g + 2], acc[0][0][4 * g + 3]};
        if (acc_dx & 1) v += *reinterpret_cast<const f32x4*>(o + 8 * g);  // second head of a pair: dX += (the heads share X)
        *reinterpret_cast<f32x4*>(o + 8 * g) = v;
      }
    }
    {  // dW [256 x 64] += dY^T X: every wave its output-channel block x both input-channel blocks, 16 rows per step
      const unsigned io = lane_o & 31, ho = lane_o >> 5;
      const unsigned b0 = (io >> 1) * 16 + 8 * ((io ^ (io >> 2)) & 1) + ((ho ^ (io >> 1) ^ (io >> 4)) & 7);
#pragma unroll
      for (int ks = 0; ks < TP / 16; ++ks) {
        const u32x4 fa = dyt[wave * 256 + (b0 ^ ((2 * (wave ^ ks)) & 7))];
        u32x4 fb[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) fb[kb] = xt[kb * 256 + (b0 ^ ((2 * (kb ^ ks)) & 7))];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) wacc[kb] = mfma_bf(fa, fb[kb], wacc[kb]);
      }
    }
    __syncthreads();
  }
  {
    float* out = wpart + (size_t)blockIdx.x * (256 * 64);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int j = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        out[(size_t)j * 64 + kb * 32 + i] = wacc[kb][reg];
      }
  }
  // bias gradient of the cloud (fp32 values, before the bf16 rounding): eight row slices of 256 column sums, in wave order
  float* red = reinterpret_cast<float*>(ldsq);
  *reinterpret_cast<f32x4*>(red + wave * 256 + 4 * lane) = f32x4{cs[0], cs[1], cs[2], cs[3]};
  __syncthreads();
  if (tid < 256) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += red[w * 256 + tid];
    dbias[(size_t)blockIdx.x * 256 + tid] = sum;
  }
}

// k_rot_l0_bwd in split mode: k_rot_l0_bwd_bf's pass with every MFMA operand as hi + lo bf16 and three products
// (fp32-grade dX and dW): hi and lo images of the row-major and the transposed dY tile and of the transposed X tile, 144 KiB
// of LDS.  dA, Y: fp32 rows.  WpT: catre_op_pack_split of W^T (lo pack 64 * 256 / 8 u32x4 behind the hi pack).
__global__ __launch_bounds__(512) void k_rot_l0_bwd_sp(const float* __restrict__ dA, const float* __restrict__ Y,
                                                       const float* __restrict__ stat, const float* __restrict__ sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ X, int ldx, const u32x4* __restrict__ WpT,
                                                       float* __restrict__ dX, int lddx, float* __restrict__ wpart,
                                                       float* __restrict__ dbias, int B, int N, int M, int acc_dx) {
  extern __shared__ __attribute__((aligned(16))) u32x4 ldsq[];
  u32x4* dys = ldsq;                  // row-major hi, bf_off<32>(row, chunk); lo image L0L_IMG slots further
  u32x4* dyt = ldsq + 2 * L0L_IMG;    // transposed hi, tn_slot(column, chunk of 8 rows); lo L0L_IMG further
  u32x4* xt = ldsq + 4 * L0L_IMG;     // transposed X tile hi: 64 columns x 8 chunks; lo 512 slots further
  auto split2 = [](float a, float b, unsigned& hi, unsigned& lo) {
    hi = pack_bf2(a, b);
    lo = pack_bf2(a - bf_lo(hi), b - bf_hi(hi));
  };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = N + M;
  const int obj = blockIdx.x % B, prior = blockIdx.x / B;
  const int ntile = (prior ? M : N) / TP;
  const size_t row0 = (size_t)obj * P + (prior ? N : 0);
  const size_t xrow0 = (acc_dx & 2) ? (prior ? (size_t)B * N + (size_t)obj * M : (size_t)obj * N) : row0;
  const int grp = lane >> 1;
  const float mean = stat[((size_t)obj * 32 + grp) * 2], rstd = stat[((size_t)obj * 32 + grp) * 2 + 1];
  const float inv_m = 1.0f / (8.f * (float)P);
  const float m1 = sums[((size_t)obj * 32 + grp) * 2] * inv_m, m2 = sums[((size_t)obj * 32 + grp) * 2 + 1] * inv_m;
  const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[lane], be = reinterpret_cast<const f32x4*>(beta)[lane];
  f32x4 sc, sh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * ga[q];
    sh[q] = be[q] - mean * sc[q];
  }
  // batch bb of a tile: rows 8 wave + 4 bb + u, u = 0..3
  f32x4 vd[2][4], vy[2][4], vx[2];
  auto request = [&](int t, int bb) {
    const size_t r = row0 + (size_t)t * TP + 8 * wave + 4 * bb;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      vd[bb][u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dA) + (r + u) * 64 + lane);
      vy[bb][u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Y) + (r + u) * 64 + lane);
    }
  };
  const int xc4 = tid & 15, xrp = tid >> 4;  // X tile: column quad and row pair (rows 2 xrp, 2 xrp + 1) of this thread
  auto request_x = [&](int t) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      vx[u] = *reinterpret_cast<const f32x4*>(X + (xrow0 + (size_t)t * TP + 2 * xrp + u) * ldx + 4 * xc4);
  };
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  f32x16 wacc[2];  // weight gradient: output-channel block `wave` x both input-channel blocks
  wacc[0] = wacc[1] = zero16();
  const int i = lane & 31, h = lane >> 5;
  u32x2* dys2 = reinterpret_cast<u32x2*>(dys);
  if (ntile > 0) {
    request(0, 0);
    request(0, 1);
    request_x(0);
  }
  for (int t = 0; t < ntile; ++t) {
    unsigned lane_o = lane;  // opaque per tile: keeps the tile's LDS addresses out of the loop preheader (k_rot_l1_bwd_bf)
    asm volatile("" : "+v"(lane_o));
    unsigned hy[4][4], hl[4][4];  // the chunk's four row pairs per column, hi / lo, packed as they are produced
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
      for (int up = 0; up < 2; ++up) {  // a pair of rows at a time (a row's temporaries do not overlap the next one's)
        f32x4 o[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const int u = 2 * up + w, row = 8 * wave + 4 * bb + u;
          const f32x4 y4 = vy[bb][u], d4 = vd[bb][u];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float yv = y4[q];
            const float xh = (yv - mean) * rstd;
            const float dxh = d4[q] * gelu_grad(fmaf(yv, sc[q], sh[q])) * ga[q];
            o[w][q] = rstd * (dxh - m1 - xh * m2);  // the operation sequence of k_gnp_bwd_apply
            cs[q] += o[w][q];
          }
          unsigned ph0, pl0, ph1, pl1;
          split2(o[w][0], o[w][1], ph0, pl0);
          split2(o[w][2], o[w][3], ph1, pl1);
          const int off = (row * 32 + ((lane_o >> 1) ^ (row & 15))) * 2 + (lane_o & 1);
          dys2[off] = u32x2{ph0, ph1};
          dys2[2 * L0L_IMG + off] = u32x2{pl0, pl1};
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) split2(o[0][q], o[1][q], hy[q][2 * bb + up], hl[q][2 * bb + up]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < ntile) request(t + 1, bb);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int slot = tn_slot(4 * (int)lane_o + q, wave);
      dyt[slot] = u32x4{hy[q][0], hy[q][1], hy[q][2], hy[q][3]};
      dyt[L0L_IMG + slot] = u32x4{hl[q][0], hl[q][1], hl[q][2], hl[q][3]};
    }
    {  // X tile transposed: rows 2 xrp, 2 xrp + 1 of column 4 xc4 + q are one dword of chunk xrp >> 2
      unsigned* xt32 = reinterpret_cast<unsigned*>(xt);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned xh, xl;
        split2(vx[0][q], vx[1][q], xh, xl);
        const int e = tn_slot(4 * xc4 + q, xrp >> 2) * 4 + (xrp & 3);
        xt32[e] = xh;
        xt32[512 * 4 + e] = xl;
      }
      if (t + 1 < ntile) request_x(t + 1);
    }
    __syncthreads();
    if (wave < 4) {  // dX tile [64 x 64] = dY W: one 32 x 32 block per wave; a lane owns one row and 16 of the block's channels
      const int mbk = wave & 1, nb = wave >> 1;
      f32x16 acc[1][1];
      acc[0][0] = zero16();
      GemmPipeS<1, 1, false, 32, 2> gp;
      gp.prefetch(WpT + ((size_t)mbk * 16) * 64 + lane_o, 0, 64 * 256 / 8);
      gp.run(acc, dys + nb * 32 * 32, dys + L0L_IMG + nb * 32 * 32, (int)lane_o);
      float* o = dX + (row0 + (size_t)t * TP + nb * 32 + i) * lddx + mbk * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[0][0][4 * g], acc[0][0][4 * g + 1], acc[0][0][4 * g + 2], acc[0][0][4 * g + 3]};
        if (acc_dx & 1) v += *reinterpret_cast<const f32x4*>(o + 8 * g);  // second head of a pair: dX += (the heads share X)
        *reinterpret_cast<f32x4*>(o + 8 * g) = v;
      }
    }
    {  // dW [256 x 64] += dY^T X: every wave its output-channel block x both input-channel blocks, 16 rows per step
      const unsigned io = lane_o & 31, ho = lane_o >> 5;
      const unsigned b0 = (io >> 1) * 16 + 8 * ((io ^ (io >> 2)) & 1) + ((ho ^ (io >> 1) ^ (io >> 4)) & 7);
#pragma unroll
      for (int ks = 0; ks < TP / 16; ++ks) {
        const int sa = wave * 256 + (b0 ^ ((2 * (wave ^ ks)) & 7));
        const u32x4 fa = dyt[sa], fal = dyt[L0L_IMG + sa];
        u32x4 fb[2], fbl[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int sb = kb * 256 + (b0 ^ ((2 * (kb ^ ks)) & 7));
          fb[kb] = xt[sb];
          fbl[kb] = xt[512 + sb];
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          wacc[kb] = mfma_bf(fal, fb[kb], wacc[kb]);
          wacc[kb] = mfma_bf(fa, fbl[kb], wacc[kb]);
          wacc[kb] = mfma_bf(fa, fb[kb], wacc[kb]);
        }
      }
    }
    __syncthreads();
  }
  {
    float* out = wpart + (size_t)blockIdx.x * (256 * 64);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int j = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        out[(size_t)j * 64 + kb * 32 + i] = wacc[kb][reg];
      }
  }
  // bias gradient of the cloud (fp32 values, before the bf16 rounding): eight row slices of 256 column sums, in wave order
  float* red = reinterpret_cast<float*>(ldsq);
  *reinterpret_cast<f32x4*>(red + wave * 256 + 4 * lane) = f32x4{cs[0], cs[1], cs[2], cs[3]};
  __syncthreads();
  if (tid < 256) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += red[w * 256 + tid];
    dbias[(size_t)blockIdx.x * 256 + tid] = sum;
  }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(32,256) + GELU on rows [R,256] (ts head): groups of 8 channels inside a row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gnr_gelu_fwd(const float* __restrict__ Y, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ A, int R) {
  const int r = blockIdx.x, ch = threadIdx.x;
  A[(size_t)r * 256 + ch] = group8_norm_gelu(Y[(size_t)r * 256 + ch], gamma[ch], beta[ch]);
}

__global__ __launch_bounds__(256) void k_gnr_gelu_bwd(const float* __restrict__ dA, const float* __restrict__ Y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ dY, float* __restrict__ dgb_part, int R) {
  const int r = blockIdx.x, ch = threadIdx.x;
  const float v = Y[(size_t)r * 256 + ch];
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
  const float xh = d * rstd, ga = gamma[ch];
  const float dyh = dA[(size_t)r * 256 + ch] * gelu_grad(fmaf(xh, ga, beta[ch]));
  const float dxh = dyh * ga;
  float s1 = dxh, s2 = dxh * xh;
  s1 += __shfl_xor(s1, 1);
  s1 += __shfl_xor(s1, 2);
  s1 += __shfl_xor(s1, 4);
  s2 += __shfl_xor(s2, 1);
  s2 += __shfl_xor(s2, 2);
  s2 += __shfl_xor(s2, 4);
  dY[(size_t)r * 256 + ch] = rstd * (dxh - s1 * 0.125f - xh * s2 * 0.125f);
  dgb_part[((size_t)r * 2) * 256 + ch] = dyh * xh;
  dgb_part[((size_t)r * 2 + 1) * 256 + ch] = dyh;
}

// ------------------------------------------------------------------------------------------------
// conv_p: out[b][c] = sum_p w[p] * Y[b*P + p][c] + bias   (Y [B*P, 3] after the neck), and backward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wsum_fwd(const float* __restrict__ Y, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ out, int P) {
  __shared__ float red[4][3];
  const int b = blockIdx.x;
  float a[3] = {0.f, 0.f, 0.f};
  for (int p = threadIdx.x; p < P; p += 256) {
    const float wp = w[p];
    const float* y = Y + ((size_t)b * P + p) * 3;
    a[0] = fmaf(wp, y[0], a[0]);
    a[1] = fmaf(wp, y[1], a[1]);
    a[2] = fmaf(wp, y[2], a[2]);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) a[c] = wave_sum(a[c]);
  if ((threadIdx.x & 63) == 0)
    for (int c = 0; c < 3; ++c) red[threadIdx.x >> 6][c] = a[c];
  __syncthreads();
  if (threadIdx.x < 3)
    out[b * 3 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x] + (bias ? bias[0] : 0.f);
}

// dY[b*P+p][c] = w[p]*dout[b][c];  dw_part[b][p] = sum_c dout[b][c]*Y[b*P+p][c]
__global__ void k_wsum_bwd(const float* __restrict__ dout, const float* __restrict__ Y, const float* __restrict__ w,
                           float* __restrict__ dY, float* __restrict__ dw_part, int B, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * P) return;
  const int b = i / P, p = i % P;
  const float* d = dout + b * 3;
  const float* y = Y + (size_t)i * 3;
  const float wp = w[p];
  dY[(size_t)i * 3 + 0] = wp * d[0];
  dY[(size_t)i * 3 + 1] = wp * d[1];
  dY[(size_t)i * 3 + 2] = wp * d[2];
  dw_part[i] = d[0] * y[0] + d[1] * y[1] + d[2] * y[2];
}

// the two bias gradients behind catre_op_wsum_bwd in one workgroup: dbias = sum(dout) (k_sum_acc's operations), and the
// neck's bias gradient dbn[c] = sum_{b,p} dY[b*P+p][c] = (sum_p w[p]) (sum_b dout[b][c]) - the sum over 3 B P products
// factors, so no pass over dY (a 256-split column sum + its merge before)
__global__ __launch_bounds__(256) void k_wsum_bias(const float* __restrict__ dout, const float* __restrict__ w, int B, int P,
                                                   float* __restrict__ dbias, float* __restrict__ dbn, int accumulate) {
  __shared__ float red[4][5];
  const int tid = threadIdx.x;
  float s = 0.f, ws = 0.f, d[3] = {0.f, 0.f, 0.f};
  for (int i = tid; i < B * 3; i += 256) s += dout[i];
  if (dbn) {
    for (int p = tid; p < P; p += 256) ws += w[p];
    for (int b = tid; b < B; b += 256) {
      d[0] += dout[b * 3], d[1] += dout[b * 3 + 1], d[2] += dout[b * 3 + 2];
    }
  }
  s = wave_sum(s), ws = wave_sum(ws);
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = wave_sum(d[c]);
  if ((tid & 63) == 0) {
    float* r = red[tid >> 6];
    r[0] = s, r[1] = ws, r[2] = d[0], r[3] = d[1], r[4] = d[2];
  }
  __syncthreads();
  if (tid == 0 && dbias) dbias[0] = (accumulate ? dbias[0] : 0.f) + ((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
  if (tid < 3 && dbn) {
    const float W = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const float D = (red[0][2 + tid] + red[1][2 + tid]) + (red[2][2 + tid] + red[3][2 + tid]);
    dbn[tid] = (accumulate ? dbn[tid] : 0.f) + W * D;
  }
}

// ------------------------------------------------------------------------------------------------
// backward of get_rot_mat (catre_so3.h) and pose_scale_from_delta_init (ego and allo rotation types)
// ------------------------------------------------------------------------------------------------
__global__ void k_pose_update_bwd(const float* __restrict__ d_pose, const float* __restrict__ d_scale,
                                  const float* __restrict__ rot6d, const float* __restrict__ dtr,
                                  const float* __restrict__ dsr, const float* __restrict__ pose0,
                                  const float* __restrict__ scale0, const float* __restrict__ mean_scales,
                                  const float* __restrict__ Ks, catre_opts o, float* __restrict__ d_rot6d,
                                  float* __restrict__ d_dt, float* __restrict__ d_ds, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* p0 = pose0 + b * 12;
  const float* dp = d_pose + b * 12;
  // R' = dR @ R0  ->  d(dR)[i][k] = sum_j dR'[i][j] R0[k][j]
  float gR[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) gR[i * 3 + k] = dp[i * 4 + 0] * p0[k * 4 + 0] + dp[i * 4 + 1] * p0[k * 4 + 1] + dp[i * 4 + 2] * p0[k * 4 + 2];
  // translation target (needed first: the allo -> ego rotation is a function of it)
  const float t0[3] = {p0[3], p0[7], p0[11]};
  float gt[3] = {dp[3], dp[7], dp[11]};
  // the rotation residual and its matrix (get_rot_mat)
  const int rd = catre_rot_dim(o.rot_type);
  float rp[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) rp[i] = i < rd ? rot6d[b * rd + i] : 0.f;
  if (o.is_allo) {
    // ego = A(t') dR with A = quat2mat(axis-angle from the optical axis to t'), core/utils/utils.py:200-231.
    // gR so far is dL/d ego:  dL/d dR = A^T gR,  dL/dA = gR dR^T -> q -> (angle, axis) -> ray -> t'.
    const float d0 = dtr[b * 3] * o.delta_t_weight, d1 = dtr[b * 3 + 1] * o.delta_t_weight,
                d2 = dtr[b * 3 + 2] * o.delta_t_weight;
    float tt[3];
    if (!o.delta_t_space_3d) {
      const float zsrc = t0[2];
      const float ztgt = o.delta_z_deepim ? zsrc / expf(d2) : d2 * zsrc;
      const float fx = o.k_aware ? Ks[b * 9 + 0] : 1.f, fy = o.k_aware ? Ks[b * 9 + 4] : 1.f;
      tt[0] = ztgt * (d0 / fx + t0[0] / zsrc);
      tt[1] = ztgt * (d1 / fy + t0[1] / zsrc);
      tt[2] = ztgt;
    } else {
      tt[0] = t0[0] + d0;
      tt[1] = t0[1] + d1;
      tt[2] = t0[2] + d2;
    }
    const float tn = sqrtf(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]);
    const float nrm = tn + o.allo_eps;
    const float ray[3] = {tt[0] / nrm, tt[1] / nrm, tt[2] / nrm};
    const float angle = acosf(ray[2]);
    const float axr[2] = {-ray[1], ray[0]};
    const float axn = sqrtf(axr[0] * axr[0] + axr[1] * axr[1]);
    const float an = axn + o.allo_eps;
    const float ax[2] = {axr[0] / an, axr[1] / an};
    const float sh = sinf(angle * 0.5f), chf = cosf(angle * 0.5f);
    const float qr[4] = {chf, ax[0] * sh, ax[1] * sh, 0.f};
    const float qn = sqrtf(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2]);
    const float qw = qr[0] / qn, qx = qr[1] / qn, qy = qr[2] / qn, qz = 0.f;
    const float A[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qw * qz), 2.f * (qx * qz + qw * qy),
                        2.f * (qx * qy + qw * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qw * qx),
                        2.f * (qx * qz - qw * qy), 2.f * (qy * qz + qw * qx), 1.f - 2.f * (qx * qx + qy * qy)};
    float dRm[9];
    rot_param_to_mat(rp, o.rot_type, dRm);
    float gA[9], gD[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gA[r * 3 + c] = gR[r * 3] * dRm[c * 3] + gR[r * 3 + 1] * dRm[c * 3 + 1] + gR[r * 3 + 2] * dRm[c * 3 + 2];
        gD[r * 3 + c] = A[r] * gR[c] + A[3 + r] * gR[3 + c] + A[6 + r] * gR[6 + c];
      }
#pragma unroll
    for (int e = 0; e < 9; ++e) gR[e] = gD[e];
    // quat2mat backward (q normalised)
    float gq[4];
    gq[0] = 2.f * (-qz * gA[1] + qy * gA[2] + qz * gA[3] - qx * gA[5] - qy * gA[6] + qx * gA[7]);
    gq[1] = 2.f * (qy * gA[1] + qz * gA[2] + qy * gA[3] - 2.f * qx * gA[4] - qw * gA[5] + qz * gA[6] + qw * gA[7] -
                   2.f * qx * gA[8]);
    gq[2] = 2.f * (-2.f * qy * gA[0] + qx * gA[1] + qw * gA[2] + qx * gA[3] + qz * gA[5] - qw * gA[6] + qz * gA[7] -
                   2.f * qy * gA[8]);
    gq[3] = 2.f * (-2.f * qz * gA[0] - qw * gA[1] + qx * gA[2] + qw * gA[3] - 2.f * qz * gA[4] + qy * gA[5] + qx * gA[6] +
                   qy * gA[7]);
    const float qg = qw * gq[0] + qx * gq[1] + qy * gq[2] + qz * gq[3];
    const float gqr[3] = {(gq[0] - qw * qg) / qn, (gq[1] - qx * qg) / qn, (gq[2] - qy * qg) / qn};
    const float g_angle = 0.5f * (-sh * gqr[0] + chf * (ax[0] * gqr[1] + ax[1] * gqr[2]));
    const float gax[2] = {sh * gqr[1], sh * gqr[2]};
    float gaxr[2] = {gax[0] / an, gax[1] / an};
    if (axn > 0.f) {
      const float k = (gax[0] * axr[0] + gax[1] * axr[1]) / (an * an * axn);
      gaxr[0] -= k * axr[0];
      gaxr[1] -= k * axr[1];
    }
    float gray[3] = {gaxr[1], -gaxr[0], 0.f};
    const float sz = sqrtf(fmaxf(1.f - ray[2] * ray[2], 0.f));
    if (sz > 0.f) gray[2] = -g_angle / sz;
    const float rg = gray[0] * tt[0] + gray[1] * tt[1] + gray[2] * tt[2];
#pragma unroll
    for (int e = 0; e < 3; ++e) gt[e] += gray[e] / nrm - (tn > 0.f ? rg * tt[e] / (nrm * nrm * tn) : 0.f);
  }
  {
    float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    rot_param_to_mat_bwd(rp, o.rot_type, gR, g);
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (i < rd) d_rot6d[b * rd + i] = g[i];
  }
  // translation
  float gd[3];
  if (!o.delta_t_space_3d) {
    const float d0 = dtr[b * 3] * o.delta_t_weight, d1 = dtr[b * 3 + 1] * o.delta_t_weight,
                d2 = dtr[b * 3 + 2] * o.delta_t_weight;
    const float zsrc = t0[2];
    const float ztgt = o.delta_z_deepim ? zsrc / expf(d2) : d2 * zsrc;
    const float fx = o.k_aware ? Ks[b * 9 + 0] : 1.f, fy = o.k_aware ? Ks[b * 9 + 4] : 1.f;
    const float ux = d0 / fx + t0[0] / zsrc, uy = d1 / fy + t0[1] / zsrc;
    const float gzt = gt[2] + gt[0] * ux + gt[1] * uy;  // d loss / d ztgt
    gd[0] = gt[0] * ztgt / fx;
    gd[1] = gt[1] * ztgt / fy;
    gd[2] = o.delta_z_deepim ? -gzt * ztgt : gzt * zsrc;
  } else {
    gd[0] = gt[0];
    gd[1] = gt[1];
    gd[2] = gt[2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) d_dt[b * 3 + i] = gd[i] * o.delta_t_weight;
  // scale
  const float* sb = o.scale_base_mean ? mean_scales + b * 3 : scale0 + b * 3;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float g = o.refine_scale ? d_scale[b * 3 + i] : 0.f;
    d_ds[b * 3 + i] = o.scale_mul ? g * sb[i] * expf(dsr[b * 3 + i]) : g;
  }
}

// part [S][2][256] -> out_a[256] (blockIdx.x == 0) / out_b[256] (blockIdx.x == 1), fixed order
__global__ __launch_bounds__(256) void k_reduce_splits2(const float* __restrict__ part, float* __restrict__ out_a,
                                                        float* __restrict__ out_b, int S, int accumulate) {
  const int which = blockIdx.x, ch = threadIdx.x;
  float* out = which ? out_b : out_a;
  float s = accumulate ? out[ch] : 0.f;
  int k = 0;
  for (; k + 8 <= S; k += 8) {  // eight rows requested together (a plain loop is one L2 round trip per row), added in order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(k + u) * 2 + which) * 256 + ch];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < S; ++k) s += part[((size_t)k * 2 + which) * 256 + ch];
  out[ch] = s;
}

// dst[0] (=|+=) sum(src[0..n))
__global__ void k_sum_acc(const float* __restrict__ src, int n, float* __restrict__ dst, int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += src[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dst[0] = (accumulate ? dst[0] : 0.f) + ((red[0] + red[1]) + (red[2] + red[3]));
}

// ------------------------------------------------------------------------------------------------
// f4: fused multi-tensor Ranger step (RAdam + Lookahead + gradient centralization) with the train loop's
// grad nan_to_num folded in.  Replaces the per-tensor Python loop of lib/torch_utils/solver/ranger.py:102-202
// (~70 tensors x ~15 torch ops, K times per data batch) and core/catre/engine/engine.py:351-353.
// The host computes the scalar step sizes; the device table carries one RangerTensor per parameter.
// ------------------------------------------------------------------------------------------------
struct RangerTensor {
  float* p;
  const float* g;
  float* m;      // exp_avg
  float* v;      // exp_avg_sq
  float* slow;   // lookahead slow weights
  int numel;
  int row_len;   // > 0: centralize the gradient over rows of this length (dims 1.. of a conv / fc weight)
  int row_off;   // first slot of this tensor in the row-mean buffer
  float lr_step; // step_size * lr
  float wd_lr;   // weight_decay * lr
  int adaptive;  // N_sma > threshold: divide by sqrt(v) + eps
  int lookahead; // step % k == 0: merge into the slow weights
  int pad;
};
static_assert(sizeof(RangerTensor) == 72, "host packs this struct with the same layout");

__device__ __forceinline__ float ranger_clean(float g, int clean, float lim) {
  // torch.nan_to_num(g, nan=0, posinf=lim, neginf=-lim) (engine.py:351-353): finite values pass unchanged
  if (!clean) return g;
  if (g != g) return 0.f;
  return g == INFINITY ? lim : (g == -INFINITY ? -lim : g);
}

// one wave per (tensor, row): mean of the (cleaned) gradient row
__global__ __launch_bounds__(64) void k_ranger_rowmean(const RangerTensor* __restrict__ T, const int* __restrict__ row_tensor,
                                                       float* __restrict__ rowmean, int clean, float lim) {
  const int row = blockIdx.x;
  const RangerTensor t = T[row_tensor[row]];
  const int r = row - t.row_off;
  const float* g = t.g + (size_t)r * t.row_len;
  float s = 0.f;
  int i = threadIdx.x;
  for (; i + 192 < t.row_len; i += 256) {  // four loads requested together, added in the loop's order (same bits)
    const float a = g[i], b = g[i + 64], c = g[i + 128], d = g[i + 192];
    s += ranger_clean(a, clean, lim);
    s += ranger_clean(b, clean, lim);
    s += ranger_clean(c, clean, lim);
    s += ranger_clean(d, clean, lim);
  }
  for (; i < t.row_len; i += 64) s += ranger_clean(g[i], clean, lim);
  s = wave_sum(s);
  if (threadIdx.x == 0) rowmean[row] = s / (float)t.row_len;
}

#define RANGER_CHUNK 4096
__global__ __launch_bounds__(256) void k_ranger_update(const RangerTensor* __restrict__ T, const int2* __restrict__ chunks,
                                                       const float* __restrict__ rowmean, float beta1, float omb1,
                                                       float beta2, float omb2, float eps, float alpha, int clean,
                                                       float lim) {
  const int2 c = chunks[blockIdx.x];
  const RangerTensor t = T[c.x];
  const int end = min(t.numel, c.y + RANGER_CHUNK);
  // four elements per thread and trip, every load of the trip requested before the first store (the tensors may alias as far
  // as the compiler knows: as a plain loop each element waited for the previous one's stores - 16 dependent round trips)
  for (int i0 = c.y + threadIdx.x; i0 < end; i0 += 1024) {
    float g[4], v[4], m[4], p[4], sl[4], rm[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + 256 * u, end - 1);
      g[u] = t.g[i], v[u] = t.v[i], m[u] = t.m[i], p[u] = t.p[i];
      sl[u] = t.lookahead ? t.slow[i] : 0.f;
      rm[u] = t.row_len > 0 ? rowmean[t.row_off + i / t.row_len] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u;
      if (i < end) {
        float gg = ranger_clean(g[u], clean, lim);
        if (t.row_len > 0) gg -= rm[u];
        // omb = fp32(1 - beta) formed in double on the host like the reference's `1 - beta2` (1.f - 0.999f is off by 4.7e-5)
        const float vv = v[u] * beta2 + omb2 * gg * gg;
        const float mm = m[u] * beta1 + omb1 * gg;
        t.v[i] = vv;
        t.m[i] = mm;
        float pp = p[u];
        if (t.wd_lr != 0.f) pp -= t.wd_lr * pp;
        pp -= t.adaptive ? t.lr_step * (mm / (sqrtf(vv) + eps)) : t.lr_step * mm;
        if (t.lookahead) {
          const float s = sl[u] + alpha * (pp - sl[u]);
          t.slow[i] = s;
          pp = s;
        }
        t.p[i] = pp;
      }
    }
  }
}
