// Reproducer attempt (MI355X): k_rot_out's ORIGINAL inner loop - packed GELU and packed neck sums exactly as hipcc paired
// them - on waves 0-3 of a workgroup, while waves 4-7 (one per SIMD next to a victim wave) run nothing / fp32 MFMAs /
// bf16 MFMAs.  The victim's per-wave sums must not depend on what the neighbour wave does.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../catre_amd/csrc -o pk_race pk_race.hip && ./pk_race
#ifndef VAR
#define VAR 0
#endif
#ifndef REPS
#define REPS 200
#endif
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "catre_device.h"
// gelu_affine4 of catre_device.h with wait states pinned right behind the two v_rcp_f32 (VAR 11-13)
#ifndef RCP_NOP
#define RCP_NOP "s_nop 0"
#endif
__device__ __forceinline__ f32x2 erf_rational2n(f32x2 x) {
  x[0] = __builtin_amdgcn_fmed3f(x[0], -4.f, 4.f);
  x[1] = __builtin_amdgcn_fmed3f(x[1], -4.f, 4.f);
  const f32x2 x2 = x * x;
  f32x2 p = pk_fma(x2, splat2(-2.72614225801306e-10f), splat2(2.77068142495902e-08f));
  p = pk_fma(x2, p, splat2(-2.10102402082508e-06f));
  p = pk_fma(x2, p, splat2(-5.69250639462346e-05f));
  p = pk_fma(x2, p, splat2(-7.34990630326855e-04f));
  p = pk_fma(x2, p, splat2(-2.95459980854025e-03f));
  p = pk_fma(x2, p, splat2(-1.60960333262415e-02f));
  f32x2 q = pk_fma(x2, splat2(-1.45660718464996e-05f), splat2(-2.13374055278905e-04f));
  q = pk_fma(x2, q, splat2(-1.68282697438203e-03f));
  q = pk_fma(x2, q, splat2(-7.37332916720468e-03f));
  q = pk_fma(x2, q, splat2(-1.42647390514189e-02f));
  float r0 = __builtin_amdgcn_rcpf(q[0]), r1 = __builtin_amdgcn_rcpf(q[1]);
  asm volatile(RCP_NOP : "+v"(r0), "+v"(r1));
  const f32x2 r = {r0, r1};
  return x * p * r;
}
__device__ __forceinline__ void gelu_affine4n(float v0, float v1, float v2, float v3, const f32x4& sc, const f32x4& sh,
                                              float (&z)[4]) {
  const f32x2 a = {v0, v1}, b = {v2, v3}, sca = {sc[0], sc[1]}, scb = {sc[2], sc[3]}, sha = {sh[0], sh[1]}, shb = {sh[2], sh[3]};
  const f32x2 ua = pk_fma(a, sca, sha), ub = pk_fma(b, scb, shb);
  const f32x2 ha = ua * splat2(0.5f), hb = ub * splat2(0.5f);
  const f32x2 za = pk_fma(ha, erf_rational2n(ua * splat2(0.70710678118654752440f)), ha);
  const f32x2 zb = pk_fma(hb, erf_rational2n(ub * splat2(0.70710678118654752440f)), hb);
  z[0] = za[0]; z[1] = za[1]; z[2] = zb[0]; z[3] = zb[1];
}
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(const float* __restrict__ y1, const float* __restrict__ stat,
                                         const float* __restrict__ gam, const float* __restrict__ bet,
                                         const float* __restrict__ neck, const float* __restrict__ wp,
                                         float* __restrict__ out, float* __restrict__ sink, int mode, int aggr_iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave >= 4) {  // the neighbour
    if (mode == 1) {
      f32x16 acc = zero16();
      const float a = 1.0f + lane * 1e-3f;
      for (int i = 0; i < aggr_iters; ++i) acc = mfma32(a, 0.5f, acc);
      if (acc[3] == 123.f) sink[tid] = acc[0];
    } else if (mode == 2) {
      f32x16 acc0 = zero16(), acc1 = zero16();
      u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f003f00u, 0x3f803f80u}, b = {0x3f003f00u, 0x3f003f00u, 0x3f003f00u, 0x3f003f00u};
      for (int i = 0; i < aggr_iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b), __builtin_bit_cast(bf16x8_t, a), acc1, 0, 0, 0);
      }
      if (acc0[3] == 123.f) sink[tid] = acc0[0] + acc1[1];
    }
    return;
  }
  // the victim: rot_out_body's loop as it was (catre_rot.h before the scalar-neck change)
  const int c0 = lane * 4;
  const float* st = stat + (c0 >> 3) * 2;
  const float mean = st[0], rstd = st[1];
  f32x4 sc, sh;
  float nk[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = rstd * gam[c0 + q];
    sh[q] = bet[c0 + q] - mean * sc[q];
#pragma unroll
    for (int c = 0; c < 3; ++c) nk[c][q] = neck[c * 256 + c0 + q];
  }
  const float* src = y1 + (size_t)blockIdx.x * 64 * 256 + c0;
  float a3[3] = {0.f, 0.f, 0.f};
#pragma unroll 4
  for (int p = wave; p < 64; p += 4) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)p * 256));
#if VAR == 1
    const float w = 0.01f + 1e-4f * p;  // no memory-loaded operand in the packed accumulate
#else
    const float w = wp[blockIdx.x * 64 + p];
#endif
    float z[4];
#if VAR == 10  // no GELU (no v_rcp, no packed polynomial): only the affine in front of the neck sums
#pragma unroll
    for (int q = 0; q < 4; ++q) z[q] = fmaf(v[q], sc[q], sh[q]);
#elif VAR >= 11
    gelu_affine4n(v[0], v[1], v[2], v[3], sc, sh, z);  // wait states behind the v_rcp_f32: -DRCP_NOP='"s_nop 3"'
#else
    gelu_affine4(v[0], v[1], v[2], v[3], sc, sh, z);
#endif
#if VAR == 9  // the form the library ships: packed GELU, scalar neck sums
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(z[q]));
#endif
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float t = nk[c][0] * z[0];
      t = fmaf(nk[c][1], z[1], t);
      t = fmaf(nk[c][2], z[2], t);
      t = fmaf(nk[c][3], z[3], t);
#if VAR == 9
      asm volatile("" : "+v"(t));
#endif
#if VAR == 2
      asm volatile("s_nop 7");  // wait states in front of the accumulate
#endif
      a3[c] = fmaf(w, t, a3[c]);
#if VAR == 9
      asm volatile("" : "+v"(a3[c]));
#endif
    }
#if VAR == 3
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // wait states after each row's packed block
#elif VAR == 4
    asm volatile("s_nop 0" ::: "memory");
#elif VAR == 5
    asm volatile("s_nop 1" ::: "memory");
#elif VAR == 6
    asm volatile("s_nop 3" ::: "memory");
#elif VAR == 7
    asm volatile("s_nop 7" ::: "memory");
#elif VAR == 8
    asm volatile("" ::: "memory");  // scheduling barrier only, no wait state
#endif
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) a3[c] = wave_sum(a3[c]);
  if (lane == 0) {
    float* o = out + ((size_t)blockIdx.x * 4 + wave) * 4;
    o[0] = a3[0];
    o[1] = a3[1];
    o[2] = a3[2];
  }
}

int main() {
  const int G = 4096;
  std::vector<float> hy((size_t)G * 64 * 256), hw((size_t)G * 64), hn(768), hg(256), hb(256), hs(64);
  unsigned s = 12345;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hy) v = 2.f * rnd();
  for (auto& v : hw) v = 0.01f * rnd();
  for (auto& v : hn) v = 0.1f * rnd();
  for (int i = 0; i < 256; ++i) { hg[i] = 1.f + 0.1f * rnd(); hb[i] = 0.1f * rnd(); }
  for (int i = 0; i < 32; ++i) { hs[2 * i] = 0.1f * rnd(); hs[2 * i + 1] = 1.f + 0.2f * rnd(); }
  float *y, *w, *n, *g, *b, *st, *out, *sink;
  hipMalloc(&y, hy.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&n, 768 * 4); hipMalloc(&g, 1024); hipMalloc(&b, 1024);
  hipMalloc(&st, 256); hipMalloc(&out, (size_t)G * 16 * 4); hipMalloc(&sink, 4096);
  hipMemcpy(y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(n, hn.data(), 768 * 4, hipMemcpyHostToDevice); hipMemcpy(g, hg.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(st, hs.data(), 256, hipMemcpyHostToDevice);
  std::vector<float> ref((size_t)G * 16), got((size_t)G * 16);
  hipMemset(out, 0, ref.size() * 4);
  hipLaunchKernelGGL(k, dim3(G), dim3(512), 0, 0, y, st, g, b, n, w, out, sink, 0, 0);
  hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost);
  const char* names[3] = {"idle", "fp32 MFMA", "bf16 MFMA"};
  for (int mode = 0; mode < 3; ++mode)
    for (int iters : {200, 2000}) {
      long bad[3] = {0, 0, 0}, runs = 0;
      for (int rep = 0; rep < REPS; ++rep) {
        hipLaunchKernelGGL(k, dim3(G), dim3(512), 0, 0, y, st, g, b, n, w, out, sink, mode, iters);
        hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost);
        ++runs;
        for (size_t i = 0; i < got.size(); ++i)
          if ((i & 3) < 3 && memcmp(&got[i], &ref[i], 4)) ++bad[i & 3];
      }
      printf("{\"neighbour\": \"%s\", \"mfma_per_neighbour_wave\": %d, \"launches\": %ld, \"wave_sums_checked\": %ld, "
             "\"wrong_component\": [%ld, %ld, %ld]}\n", names[mode], iters, runs, runs * (long)G * 4, bad[0], bad[1], bad[2]);
      if (mode == 0) break;
      if (mode == 1) break;
    }
  return 0;
}
