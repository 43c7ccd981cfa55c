// Micro-benchmark (MI355X): the K-sweep building block of the fp32 kernels in isolation - GemmPipe<4,2> / <2,2> with
// weight fragments from L2 and activation fragments from LDS, nothing else (no prologue, no epilogue work, no barriers).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../catre_amd/csrc -o gemmpipe gemmpipe.hip && ./gemmpipe
// Tells how much of a kernel's distance to 64 cycles/MFMA belongs to the sweep itself.
#include "catre_device.h"
#include <cstdio>
#include <vector>

template <int MB, int NKC, int PFD, bool SWZ, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const f32x4* __restrict__ wp, float* out, int reps, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int LD = SWZ ? 8 * NKC : 8 * NKC + 4;
  for (int i = tid; i < TP * LD; i += WAVES * 64) lds[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb][0] = acc[mb][1] = zero16();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    GemmPipe<MB, 2, true, SWZ, NKC, PFD, 1> g;
    const f32x4* w = wp + ((size_t)((wave * MB + r % 3) * NKC)) * 64 + lane;
    asm volatile("" : "+v"(w));
    g.prefetch(w, NKC * 64);
    g.run(acc, lds, LD, lane);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) s += acc[mb][0][0] + acc[mb][1][5];
  if (s == 12345.678f) out[tid] = s;
  if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int MB, int NKC, int PFD, bool SWZ, int WAVES>
void run(const char* name, int wg_per_cu, size_t lds_bytes) {
  const int reps = 64;
  f32x4* wp;
  float* out;
  unsigned long long* cyc;
  const size_t wfloats = (size_t)(WAVES * MB + 4) * NKC * 64 * 4 + 4096;
  hipMalloc(&wp, wfloats * 4);
  hipMemset(wp, 0, wfloats * 4);
  hipMalloc(&out, 1 << 16);
  const int grid = 256 * wg_per_cu;
  hipMalloc(&cyc, grid * WAVES * 8);
  hipFuncSetAttribute((const void*)k<MB, NKC, PFD, SWZ, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MB, NKC, PFD, SWZ, WAVES>), dim3(grid), dim3(WAVES * 64), lds_bytes, 0, wp, out, reps, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * WAVES);
  hipMemcpy(h.data(), cyc, grid * WAVES * 8, hipMemcpyDeviceToHost);
  double c = 0;
  for (auto v : h) c += v;
  c /= h.size();
  const double mf = (double)reps * NKC * 4 * MB * 2;  // MFMAs per wave
  const double waves_per_simd = WAVES * wg_per_cu / 4.0;
  printf("%-58s %7.3f ms  %6.1f TFLOP/s  %6.1f cyc per MFMA per wave  -> %5.1f cyc per MFMA per SIMD\n", name, ms,
         mf * grid * WAVES * 4096 / (ms * 1e-3) / 1e12, c / mf, c / mf / waves_per_simd);
  hipFree(wp);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<4, 64, 2, true, 8>("trunk conv4: MB4 NB2 K512 swz, 8 waves, 1 WG/CU (160K)", 1, 160 * 1024);
  run<4, 64, 2, true, 4>("same sweep, 4 waves, 1 WG/CU (1 wave/SIMD)", 1, 160 * 1024);
  run<4, 16, 2, false, 4>("STN conv3 pass: MB4 NB2 K128 padded, 4 waves, 2 WG/CU", 2, 64 * 132 * 4);
  run<4, 16, 2, false, 4>("same, 1 WG/CU", 1, 80 * 1024 + 64 * 132 * 4);
  run<2, 32, 2, true, 4>("rot layer 1: MB2 NB2 K256 swz, 4 waves, 2 WG/CU", 2, 64 * 1024);
  run<2, 32, 3, true, 4>("same, 3 chunks in flight", 2, 64 * 1024);
  run<4, 64, 3, true, 8>("trunk conv4 with 3 chunks in flight", 1, 160 * 1024);
  return 0;
}
