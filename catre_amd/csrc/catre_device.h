// catre_device.h - device-side building blocks shared by the CATRE hot-path kernels (gfx950 only).
//
// Data layout conventions used by every MFMA kernel in this directory
// -------------------------------------------------------------------
// * A workgroup owns one TILE of TP = 64 points of one cloud and carries it through a whole
//   chain of 1x1-conv layers; activations never leave LDS between layers.
// * Activations live in LDS point-major: act[point][channel] with a row pitch of C+4 floats.
//   The +4 (16 B) skew makes the 16-byte fragment reads below conflict-free: the 16-lane
//   groups of ds_read_b128 hit 16 distinct 4-bank slots (bank = 4*point + k mod 64).
// * Weights are streamed from L2 straight into registers in "fragment-packed" order
//   (catre_pack_weights): float4 index ((mblk*(K/8) + kc)*64 + lane) holds
//       W[mblk*32 + (lane&31)][kc*8 + 4*(lane>>5) + {0,1,2,3}]
//   so one coalesced 1 KiB global_load_dwordx4 per wave feeds four v_mfma_f32_32x32x2_f32.
//   The activation fragment uses the same k-slot mapping (lane>>5 selects k-offset 4h+s within
//   an 8-wide chunk); the k order inside a chunk is permuted identically for both operands,
//   which only re-associates the fp32 sum.
// * v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]; the
//   result D[i][j] sits at col j = l&31, row i = (reg&3) + 8*(reg>>2) + 4*(l>>5).
//     "normal"  call mfma(w, x): rows = out-channels, cols = points  (4 consecutive channels per
//               register quad -> one ds_write_b128 per quad into the next layer's LDS image)
//     "swapped" call mfma(x, w): rows = points, cols = out-channels  (a lane owns ONE channel and
//               16 points -> the max-pool over points is in-register)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TP 64  // points per tile

// Per-phase cycle stamps (profiles/trace_*.py) are compiled into the instrumented library only
// (`make TRACE=1` -> libcatre_hip_trace.so); in the product library the stamp macros fold to nothing.
#ifdef CATRE_DEBUG_TRACE
#define CATRE_TRACE_ON 1
#else
#define CATRE_TRACE_ON 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 zero16() {
  f32x16 v;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.f;
  return v;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// erf(x) in fp32 as x * P(x^2) / Q(x^2) on the clamped range |x| <= 4 (the single-precision rational
// minimax fit published with Eigen / XLA).  Max abs error 4.5e-7 evaluated in fp32 - the same class as
// torch's own fp32 erf-GELU (measured: GELU max abs error 1.36e-6 vs 1.21e-6 for torch, both against
// fp64) - at 16 VALU ops and one v_rcp instead of the ~45 ops of the two-branch libm erff, which made the
// rotation head VALU-bound (GELU phase as long as its 256x256 MFMA layer).
__device__ __forceinline__ float erf_rational(float x) {
  x = __builtin_amdgcn_fmed3f(x, -4.f, 4.f);  // clamp in one op
  const float x2 = x * x;
  float p = fmaf(x2, -2.72614225801306e-10f, 2.77068142495902e-08f);
  p = fmaf(x2, p, -2.10102402082508e-06f);
  p = fmaf(x2, p, -5.69250639462346e-05f);
  p = fmaf(x2, p, -7.34990630326855e-04f);
  p = fmaf(x2, p, -2.95459980854025e-03f);
  p = fmaf(x2, p, -1.60960333262415e-02f);
  float q = fmaf(x2, -1.45660718464996e-05f, -2.13374055278905e-04f);
  q = fmaf(x2, q, -1.68282697438203e-03f);
  q = fmaf(x2, q, -7.37332916720468e-03f);
  q = fmaf(x2, q, -1.42647390514189e-02f);
  return x * p * __builtin_amdgcn_rcpf(q);
}

// Phi(v) = 0.5 (1 + erf(v / sqrt 2)) from the same rational, with the argument scaling (1/sqrt 2), the x^2 = v^2 / 2 and the
// final 0.5 (1 + .) folded into the coefficients (powers of two: exact; the common factor 0.5/sqrt 2 is one rounding per
// numerator coefficient): clamp, square, 6 + 4 FMAs, v_rcp, one multiply, one FMA - two issue slots fewer per GELU than
// 0.5 v (1 + erf_rational(v/sqrt 2)), same 1.5e-6 absolute accuracy over |v| <= 8 (checked against scipy in fp32 emulation).
__device__ __forceinline__ float gelu_cdf(float v) {
  const float c = __builtin_amdgcn_fmed3f(v, -5.656854f, 5.656854f);
  const float t = c * c;
  float p = fmaf(t, -1.5059951e-12f, 3.0611993e-10f);
  p = fmaf(t, p, -4.6426507e-08f);
  p = fmaf(t, p, -2.515756e-06f);
  p = fmaf(t, p, -6.496461e-05f);
  p = fmaf(t, p, -5.223044e-04f);
  p = fmaf(t, p, -5.690807e-03f);
  float q = fmaf(t, -9.1037947e-07f, -2.6671756e-05f);
  q = fmaf(t, q, -4.2070675e-04f);
  q = fmaf(t, q, -3.6866646e-03f);
  q = fmaf(t, q, -1.4264739e-02f);
  return fmaf(c * p, __builtin_amdgcn_rcpf(q), 0.5f);
}

__device__ __forceinline__ float gelu_erf(float v) {
  // nn.GELU() default: 0.5 * v * (1 + erf(v / sqrt(2))) = v * Phi(v)
  return v * gelu_cdf(v);
}

// Two GELUs per instruction stream: the same operation sequence as gelu_erf (bit-identical results) written on
// float2 (the packed-fp32 forms it once compiled to are disabled library-wide; the pairing still gives the
// scheduler two independent chains).  Used where GELU is the bottleneck (rotation head: 2 x 2 x 256 x (N+M) evaluations per
// object).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float a) {
  f32x2 v = {a, a};
  return v;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 gelu_cdf2(f32x2 v) {
  f32x2 c;
  c[0] = __builtin_amdgcn_fmed3f(v[0], -5.656854f, 5.656854f);
  c[1] = __builtin_amdgcn_fmed3f(v[1], -5.656854f, 5.656854f);
  const f32x2 t = c * c;
  f32x2 p = pk_fma(t, splat2(-1.5059951e-12f), splat2(3.0611993e-10f));
  p = pk_fma(t, p, splat2(-4.6426507e-08f));
  p = pk_fma(t, p, splat2(-2.515756e-06f));
  p = pk_fma(t, p, splat2(-6.496461e-05f));
  p = pk_fma(t, p, splat2(-5.223044e-04f));
  p = pk_fma(t, p, splat2(-5.690807e-03f));
  f32x2 q = pk_fma(t, splat2(-9.1037947e-07f), splat2(-2.6671756e-05f));
  q = pk_fma(t, q, splat2(-4.2070675e-04f));
  q = pk_fma(t, q, splat2(-3.6866646e-03f));
  q = pk_fma(t, q, splat2(-1.4264739e-02f));
  const f32x2 r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  return pk_fma(c * p, r, splat2(0.5f));
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) { return v * gelu_cdf2(v); }
// z[0..3] = gelu(v * sc + sh) on a register quad
__device__ __forceinline__ void gelu_affine4(float v0, float v1, float v2, float v3, const f32x4& sc, const f32x4& sh,
                                             float (&z)[4]) {
  const f32x2 a = {v0, v1}, b = {v2, v3}, sca = {sc[0], sc[1]}, scb = {sc[2], sc[3]}, sha = {sh[0], sh[1]},
              shb = {sh[2], sh[3]};
  const f32x2 za = gelu_erf2(pk_fma(a, sca, sha)), zb = gelu_erf2(pk_fma(b, scb, shb));
  z[0] = za[0];
  z[1] = za[1];
  z[2] = zb[0];
  z[3] = zb[1];
}

// Reduced-precision modes only (bf16 operands: csrc/catre_bf16.h): erf as x P3(x^2) / Q3(x^2) on |x| <= 3.6 - 6 FMAs + v_rcp
// instead of 10, GELU max abs error 1.7e-5 (fitted and evaluated on the goldens by tests/emulate_gelu.py: 3.3e-5 on
// (R, t, s), an order of magnitude inside what rounding the operands to bf16 costs; NOT used by the fp32 / split kernels,
// whose 2e-5 bar it would eat).  The bf16 rotation-head kernels are VALU-bound on this function.
__device__ __forceinline__ float gelu_cdf_lp(float v) {
  // Phi(v) with the scalings folded into the coefficients like gelu_cdf
  const float c = __builtin_amdgcn_fmed3f(v, -5.091169f, 5.091169f);
  const float t = c * c;
  float p = fmaf(t, 2.9459565e-05f, 3.7487173e-03f);
  p = fmaf(t, p, 2.9940134e-02f);
  p = fmaf(t, p, 3.9886147e-01f);
  float q = fmaf(t, 1.0829841e-03f, 2.4883246e-02f);
  q = fmaf(t, q, 2.4130477e-01f);
  q = fmaf(t, q, 1.0f);
  return fmaf(c * p, __builtin_amdgcn_rcpf(q), 0.5f);
}
__device__ __forceinline__ float gelu_erf_lp(float v) {
  // v * Phi(v): 15 issue slots instead of 18
  return v * gelu_cdf_lp(v);
}
__device__ __forceinline__ void gelu_affine4_lp(float v0, float v1, float v2, float v3, const f32x4& sc, const f32x4& sh,
                                                float (&z)[4]) {
  z[0] = gelu_erf_lp(fmaf(v0, sc[0], sh[0]));
  z[1] = gelu_erf_lp(fmaf(v1, sc[1], sh[1]));
  z[2] = gelu_erf_lp(fmaf(v2, sc[2], sh[2]));
  z[3] = gelu_erf_lp(fmaf(v3, sc[3], sh[3]));
}

// One K-sweep of a wave tile: MB x NB blocks of 32x32, K = 8*NKC.
//   wp   : fragment-packed weights, already offset to [first m-block][first k-chunk][lane]
//   wp_mb: float4 stride between consecutive m-blocks  (= (Ktotal/8)*64)
//   x    : LDS image of point block 0 of this wave tile, [point][ld]; SWZ selects the XOR-swizzled layout
// Software pipeline: the A fragments (L2) and B fragments (LDS) of chunk kc+PFD are issued BEFORE the
// 4*MB*NB MFMAs of chunk kc and pinned there with sched_barrier - left alone, hipcc sinks the loads down
// to their first use (to save registers) and every chunk then eats a full L2 round trip.  PFD chunks are
// in flight: 1 is enough behind 32 MFMAs (MB=4), the thin MB=1/2 layers need 2-3 to cover L2 latency.
// NKC is a template parameter so the ring-buffer indices below are compile-time constants (a runtime-indexed
// register array would go to scratch).
template <int MB, int NB, bool SWAP, bool SWZ, int NKC, int PFD, int PFB = 1>
struct GemmPipe {
  // PFD: chunks of A (weights, L2 latency ~1 us under load) in flight; PFB: chunks of B (LDS) in flight.
  static_assert(PFD >= 1 && PFD <= NKC && PFB >= 1 && PFB <= PFD, "prefetch depth");
  static constexpr int RA = PFD + 1, RB = PFB + 1;
  f32x4 a[RA][MB], b[RB][NB];
  const f32x4* wp;
  int wp_mb;

  __device__ __forceinline__ void issue_a(int kc) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a[kc % RA][mb] = wp[mb * wp_mb + kc * 64];
  }
  // Issue the first PFD weight chunks.  They do not depend on LDS, so a kernel calls this BEFORE the
  // epilogue + barrier of the previous layer: the L2 round trip then overlaps that work instead of
  // sitting exposed at the head of the layer.
  __device__ __forceinline__ void prefetch(const f32x4* __restrict__ wp_, int wp_mb_) {
    wp = wp_;
    wp_mb = wp_mb_;
#pragma unroll
    for (int d = 0; d < PFD; ++d) issue_a(d);
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void run(f32x16 (&acc)[MB][NB], const float* x, int ld, int lane) {
    const int n = lane & 31, h = lane >> 5, sw = lane & 15;
    const float* xrow = x + n * ld;
    // Swizzled image: chunk c = 2 kc + h of row r sits at chunk c ^ (r & 15).  The XOR only touches the low four bits, so
    // c ^ sw = (c & ~15) | ((c & 15) ^ sw): eight per-lane row pointers (kc & 7) cover the sweep and the rest of the
    // offset, 64 floats per 8 chunks, is a compile-time immediate of the ds_read - no address VALU inside the K loop
    // (every VALU instruction costs the fp32 MFMA stream ~4 cycles: profiles/ubench/mfma_valu.hip).
    constexpr int NLOW = SWZ ? (NKC < 8 ? NKC : 8) : 1;
    const float* xlow[NLOW];
    if (SWZ) {
#pragma unroll
      for (int j = 0; j < NLOW; ++j) xlow[j] = xrow + (((2 * j + h) ^ sw) << 2);
    }
    auto issue_b = [&](int kc) {
      const float* src = SWZ ? xlow[kc % NLOW] + (kc / 8) * 64 : xrow + (kc * 8 + 4 * h);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) b[kc % RB][nb] = *reinterpret_cast<const f32x4*>(src + nb * 32 * ld);
    };
#pragma unroll
    for (int d = 0; d < PFB; ++d) issue_b(d);
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      if (kc + PFD < NKC) issue_a(kc + PFD);
      if (kc + PFB < NKC) issue_b(kc + PFB);
      __builtin_amdgcn_sched_barrier(0);
      const int ca = kc % RA, cb = kc % RB;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] =
                SWAP ? mfma32(b[cb][nb][s], a[ca][mb][s], acc[mb][nb]) : mfma32(a[ca][mb][s], b[cb][nb][s], acc[mb][nb]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// one-shot form: prefetch + run back to back
template <int MB, int NB, bool SWAP, bool SWZ, int NKC, int PFD, int PFB = 1>
__device__ __forceinline__ void gemm_core(f32x16 (&acc)[MB][NB], const f32x4* __restrict__ wp, int wp_mb,
                                          const float* x, int ld, int lane) {
  GemmPipe<MB, NB, SWAP, SWZ, NKC, PFD, PFB> g;
  g.prefetch(wp, wp_mb);
  g.run(acc, x, ld, lane);
}

// XOR-swizzled LDS images (no row padding): the 16-byte chunk c of row r lives at chunk c ^ (r & 15).
// For row pitches that are multiples of 64 floats this spreads the 16 rows of a ds_read_b128 lane
// group over 16 distinct 4-bank slots exactly like the +4 skew does, but costs no LDS bytes - which
// is what lets two 80 KiB workgroups share one CU in the rotation head.
__device__ __forceinline__ int swz_off(int row, int chunk, int ld) { return row * ld + ((chunk ^ (row & 15)) << 2); }

// "normal"-orientation epilogue: out[point][ch] = act(acc + bias[ch]) as float4 per register quad.
template <int MB, int NB, bool RELU>
__device__ __forceinline__ void store_tile_lds(const f32x16 (&acc)[MB][NB], float* out, int ldo, int ch0,
                                               const float* __restrict__ bias, int lane) {
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = ch0 + mb * 32 + 8 * g + 4 * h;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ch);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float t = acc[mb][nb][4 * g + q] + bv[q];
          v[q] = RELU ? fmaxf(t, 0.f) : t;
        }
        *reinterpret_cast<f32x4*>(out + (nb * 32 + n) * ldo + ch) = v;
      }
    }
}

// Same epilogue into an XOR-swizzled image: `out` is the [64][ldo] tile base, channels are absolute.
template <int MB, int NB, bool RELU>
__device__ __forceinline__ void store_tile_lds_swz(const f32x16 (&acc)[MB][NB], float* out, int ldo, int ch0,
                                                   const float* __restrict__ bias, int lane) {
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = ch0 + mb * 32 + 8 * g + 4 * h;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ch);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float t = acc[mb][nb][4 * g + q] + bv[q];
          v[q] = RELU ? fmaxf(t, 0.f) : t;
        }
        *reinterpret_cast<f32x4*>(out + swz_off(nb * 32 + n, ch >> 2, ldo)) = v;
      }
    }
}

// Bias quads of a "normal"-orientation wave tile, loaded ahead of the layer (pinned by the caller's
// sched_barrier) so the epilogue does not pay a global round trip.
template <int MB>
__device__ __forceinline__ void load_bias_quads(f32x4 (&bv)[MB][4], const float* __restrict__ bias, int ch0, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[mb][g] = *reinterpret_cast<const f32x4*>(bias + ch0 + mb * 32 + 8 * g + 4 * h);
}

template <int MB, int NB, bool RELU, bool SWZ>
__device__ __forceinline__ void store_tile_lds_pre(const f32x16 (&acc)[MB][NB], float* out, int ldo, int ch0,
                                                   const f32x4 (&bv)[MB][4], int lane) {
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = ch0 + mb * 32 + 8 * g + 4 * h;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float t = acc[mb][nb][4 * g + q] + bv[mb][g][q];
          v[q] = RELU ? fmaxf(t, 0.f) : t;
        }
        const int off = SWZ ? swz_off(nb * 32 + n, ch >> 2, ldo) : (nb * 32 + n) * ldo + ch;
        *reinterpret_cast<f32x4*>(out + off) = v;
      }
    }
}

// "swapped"-orientation epilogue: channel-wise max over the tile's 64 points, then bias (+ReLU).
// max_n act(y_n + b) == act(max_n y_n + b) because both are monotone.
template <int MB, int NB>
__device__ __forceinline__ void max_tile_store(const f32x16 (&acc)[MB][NB], float* __restrict__ out, int ch0,
                                               const float* __restrict__ bias, bool relu, int lane) {
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    float m = acc[mb][0][0];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[mb][nb][r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    if (lane < 32) {
      const int ch = ch0 + mb * 32 + lane;
      float v = m + bias[ch];
      out[ch] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

// per-lane bias of a "swapped"-orientation wave tile (lane owns channel ch0 + mb*32 + lane%32)
template <int MB>
__device__ __forceinline__ void load_bias_lane(float (&bl)[MB], const float* __restrict__ bias, int ch0, int lane) {
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) bl[mb] = bias[ch0 + mb * 32 + (lane & 31)];
}

template <int MB, int NB>
__device__ __forceinline__ void max_tile_store_pre(const f32x16 (&acc)[MB][NB], float* __restrict__ out, int ch0,
                                                   const float (&bl)[MB], bool relu, int lane) {
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    float m = acc[mb][0][0];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[mb][nb][r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    if (lane < 32) {
      const float v = m + bl[mb];
      out[ch0 + mb * 32 + lane] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}


// Stores of large activation tensors that the next launch (or the backward, much later) reads back: non-temporal.  A plain
// store leaves its line dirty in the 256 MiB Infinity Cache; behind > 256 MiB of them the NEXT kernel's reads run against
// their write-back (profiles/ubench/mall_reread.hip: 1 GiB written with plain stores reads back at 4.1 TB/s, with
// non-temporal stores at 5.7 TB/s; a 256 MiB buffer behind 1 GiB of other plain-store traffic at 2.5 TB/s).
#ifndef CATRE_NT_SAVES
#define CATRE_NT_SAVES 1
#endif
template <class T>
__device__ __forceinline__ void st_stream(T* p, const T& v) {
#if CATRE_NT_SAVES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// ---- training forward of the fused encoder kernels (SAVE variants): what the layer-wise backward needs ----------------
// LDS activation image [TP][ld] (point-major) -> global rows dst[TP][C], coalesced float4 stores by all NT threads
template <int C, int NT, bool SWZ>
__device__ __forceinline__ void save_tile_rows(const float* __restrict__ img, int ld, float* __restrict__ dst, int tid) {
  constexpr int F4 = C / 4;
#pragma unroll
  for (int u = 0; u < TP * F4 / NT; ++u) {
    const int i = tid + NT * u, row = i / F4, c4 = i % F4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(img + (SWZ ? swz_off(row, c4, ld) : row * ld + c4 * 4));
    st_stream(reinterpret_cast<f32x4*>(dst + (size_t)row * C + c4 * 4), v);
  }
}

// "swapped"-orientation epilogue with the arg-max: per channel the maximum of (acc + bias) over the tile's 64 points
// and the ROW it came from (row0 + point; the first maximum wins, like torch.max) - the forward of linear + max-pool
// for training, where the backward gathers / scatters at exactly these rows.
template <int MB, int NB>
__device__ __forceinline__ void argmax_tile_store(const f32x16 (&acc)[MB][NB], float* __restrict__ pmax,
                                                  int* __restrict__ pidx, int ch0, const float (&bl)[MB], int row0,
                                                  int lane) {
  // Per accumulator: half a v_max3 for the maximum, then a compare + select that walks the points BACKWARDS looking for
  // it (the last hit of the walk is the first occurrence: torch.max's choice) - 2.5 issue slots instead of the 4 of
  // "add bias, compare, select value, select index" per accumulator.  The bias joins after the maximum: x -> fl(x + b) is
  // monotonic, so max_r fl(v_r + b) == fl(max_r v_r + b) - the pooled value keeps its bits (and equals the inference
  // kernels'); only a tie that the rounding of v_r + b would have created is now seen as the strict order of the v_r.
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    float m = acc[mb][0][0];
#pragma unroll
    for (int i = 1; i + 1 < NB * 16; i += 2) m = fmaxf(fmaxf(m, acc[mb][i >> 4][i & 15]), acc[mb][(i + 1) >> 4][(i + 1) & 15]);
    m = fmaxf(m, acc[mb][NB - 1][15]);
    int am = 0;
#pragma unroll
    for (int nb = NB - 1; nb >= 0; --nb)
#pragma unroll
      for (int r = 15; r >= 0; --r)  // decreasing point order inside a half-wave: the first occurrence is the last hit
        am = acc[mb][nb][r] == m ? nb * 32 + (r & 3) + 8 * (r >> 2) : am;  // (an inline constant; the half-wave's 4h joins once)
    am += 4 * h;
    m += bl[mb];
    const float mo = __shfl_xor(m, 32);
    const int ao = __shfl_xor(am, 32);
    if (mo > m || (mo == m && ao < am)) {
      m = mo;
      am = ao;
    }
    if (h == 0) {
      pmax[ch0 + mb * 32 + n] = m;
      pidx[ch0 + mb * 32 + n] = row0 + am;
    }
  }
}
