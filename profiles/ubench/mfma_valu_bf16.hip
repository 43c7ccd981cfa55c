// Micro-benchmark (MI355X): does fp32 VALU work on a SIMD take throughput away from v_mfma_f32_32x32x16_bf16?
// (the bf16 twin of mfma_valu.hip)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_bf16 mfma_valu_bf16.hip && ./mfma_valu_bf16
// Every workgroup = 512 threads = 8 waves = 2 waves per SIMD on its CU, one workgroup per CU (grid 256), LDS 160 KiB
// requested so that nothing else becomes resident.  Waves 0-3 run role A, waves 4-7 role B (one of each per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 mk(float a) { bf16x8 v; for (int i = 0; i < 8; ++i) v[i] = (__bf16)(a + i); return v; }

enum Role { IDLE = 0, MFMA = 1, VALU = 2, PKVALU = 3, MIX = 4, TRANS = 5 };

template <int NACC>
__device__ __forceinline__ float run_mfma(int iters, float a, float b) {
  const bf16x8 av = mk(a), bv = mk(b);
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
  return s;
}

__device__ __forceinline__ float run_valu(int iters, float a, float b) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], b, a);  // 64 independent-ish v_fma_f32 per iteration
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  return s;
}

__device__ __forceinline__ float run_pkvalu(int iters, float a, float b) {
  f32x2 x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = f32x2{a + i, a - i};
  const f32x2 bb = {b, b}, aa = {a, a};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_elementwise_fma(x[i], bb, aa);  // 64 v_pk_fma_f32
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i][0] + x[i][1];
  return s;
}

__device__ __forceinline__ float run_trans(int iters, float a, float b) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = a + i + 2.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_rcpf(x[i]) + b;  // 64 v_rcp_f32 + 64 v_add
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  return s;
}

// MIX: one wave issuing NV VALU fmas after every MFMA
template <int NV>
__device__ __forceinline__ float run_mix(int iters, float a, float b) {
  const bf16x8 av = mk(a), bv = mk(b);
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) x[(i + v) & 7] = fmaf(x[(i + v) & 7], b, a);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + x[i];
  return s;
}

template <int RA, int RB, int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters_a, int iters_b, float a, float b, unsigned long long* cyc) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? RA : RB;
  const int iters = wave < 4 ? iters_a : iters_b;
  const unsigned long long t0 = __builtin_readcyclecounter();
  float s = 0.f;
  if (role == MFMA) s = run_mfma<8>(iters, a, b);
  else if (role == VALU) s = run_valu(iters, a, b);
  else if (role == PKVALU) s = run_pkvalu(iters, a, b);
  else if (role == TRANS) s = run_trans(iters, a, b);
  else if (role == MIX) s = run_mix<NV>(iters, a, b);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (s == 123.456f) out[threadIdx.x] = s + lds[threadIdx.x];
}

template <int RA, int RB, int NV = 0>
void run(const char* name, int ia, int ib, double mfma_per_iter_a, double mfma_per_iter_b) {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 4096);
  hipMalloc(&cyc, 256 * 8 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<RA, RB, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<RA, RB, NV>), dim3(256), dim3(512), 160 * 1024, 0, out, ia, ib, 1.0f, 0.5f, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
  double ca = 0, cb = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? ca : cb) += (double)h[b * 8 + w];
  ca /= 1024;
  cb /= 1024;
  const double mfmas = 256.0 * 4 * (ia * mfma_per_iter_a + ib * mfma_per_iter_b);
  printf("%-44s %8.3f ms  waveA %9.0f cyc  waveB %9.0f cyc  bf16-MFMA %7.1f TFLOP/s  cyc/MFMA(A) %6.1f\n", name, ms, ca, cb,
         mfmas * 32768 / (ms * 1e-3) / 1e12, mfma_per_iter_a > 0 ? ca / (ia * mfma_per_iter_a) : 0.0);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  const int I = 40000;
  run<MFMA, IDLE>("A: 1 bf16-MFMA wave / SIMD", I, 0, 8, 0);
  run<MFMA, MFMA>("D: 2 bf16-MFMA waves / SIMD", I, I, 8, 8);
  run<VALU, IDLE>("V: 1 VALU wave / SIMD (64 fma / iter)", I, 0, 0, 0);
  run<VALU, VALU>("VV: 2 VALU waves / SIMD (64 fma / iter each)", I, I, 0, 0);
  run<MFMA, VALU>("B: MFMA wave + VALU wave (same iteration count)", I, I, 8, 0);
  run<MFMA, VALU>("B2: MFMA wave + VALU wave (2x VALU iterations)", I, I * 2, 8, 0);
  run<MFMA, TRANS>("T: MFMA wave + rcp/add wave", I, I, 8, 0);
  run<MIX, IDLE, 1>("E1: one wave, 1 v_fma after each MFMA", I, 0, 8, 0);
  run<MIX, IDLE, 2>("E2: one wave, 2 v_fma after each MFMA", I, 0, 8, 0);
  run<MIX, IDLE, 4>("E4: one wave, 4 v_fma after each MFMA", I, 0, 8, 0);
  run<MIX, IDLE, 8>("E8: one wave, 8 v_fma after each MFMA", I, 0, 8, 0);
  run<MIX, IDLE, 16>("E16: one wave, 16 v_fma after each MFMA", I, 0, 8, 0);
  run<MIX, MIX, 4>("F4: two waves, 4 v_fma after each MFMA", I, I, 8, 8);
  run<MIX, MIX, 8>("F8: two waves, 8 v_fma after each MFMA", I, I, 8, 8);
  run<MIX, MIX, 16>("F16: two waves, 16 v_fma after each MFMA", I, I, 8, 8);
  return 0;
}
