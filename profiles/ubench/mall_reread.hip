// Micro-benchmark (MI355X): is a buffer that was just written by the previous kernel read back faster when it fits the
// 256 MiB Infinity Cache?  (The rotation heads write y1 [B,2,P,256] and read it back in the next launch: 1.08 GB in fp32,
// 0.58 GB in bf16 mode - would processing the batch in object chunks keep that round trip on the die?)
//   hipcc --offload-arch=gfx950 -O3 -o mall_reread mall_reread.hip && ./mall_reread
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_write(f32x4* __restrict__ p, size_t n4, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 x = {v, v + 1.f, v + 2.f, (float)i};
    if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x;
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ p, size_t n4, float* __restrict__ out) {
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256 * 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t j = i + (size_t)u * gridDim.x * 256;
      const size_t jj = j < n4 ? j : i;
      v[u] = NT ? __builtin_nontemporal_load(p + jj) : p[jj];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) a += v[u];
  }
  if (a[0] + a[1] + a[2] + a[3] == 123.456f) out[threadIdx.x] = a[0];
}

template <bool NTW, bool NTR>
void run(const char* name, size_t mb, f32x4* buf, f32x4* other, size_t other_n4, float* out, bool flush) {
  const size_t n4 = mb * 1024 * 1024 / 16;
  hipEvent_t e0, e1, e2;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  float wms = 0, rms = 0;
  const int reps = 5;
  for (int r = 0; r < reps + 1; ++r) {
    if (flush) hipLaunchKernelGGL((k_write<false>), dim3(2048), dim3(256), 0, 0, other, other_n4, 3.f);  // evict
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_write<NTW>), dim3(2048), dim3(256), 0, 0, buf, n4, 1.f);
    hipEventRecord(e1);
    if (flush) hipLaunchKernelGGL((k_write<false>), dim3(2048), dim3(256), 0, 0, other, other_n4, 2.f);  // evict between
    hipEventRecord(e1);
    hipLaunchKernelGGL((k_read<NTR>), dim3(2048), dim3(256), 0, 0, buf, n4, out);
    hipEventRecord(e2);
    hipEventSynchronize(e2);
    float a, b;
    hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
    if (r) { wms += a; rms += b; }
  }
  printf("%-46s %5zu MiB  read %7.1f us = %6.2f TB/s\n", name, mb, rms / reps * 1e3, mb * 1.048576e6 / (rms / reps * 1e-3) / 1e12);
}

int main() {
  f32x4 *buf, *other; float* out;
  hipMalloc(&buf, (size_t)2048 << 20); hipMalloc(&other, (size_t)1024 << 20); hipMalloc(&out, 4096);
  const size_t on4 = ((size_t)1024 << 20) / 16;
  for (size_t mb : {32, 64, 96, 128, 192, 256, 512, 1024}) {
    run<false, false>("write, then read back (plain / plain)", mb, buf, other, on4, out, false);
    run<false, true>("write, then read back (plain / nt load)", mb, buf, other, on4, out, false);
    run<true, true>("write, then read back (nt store / nt load)", mb, buf, other, on4, out, false);
    run<false, false>("write, 1 GiB of other traffic, read (cold)", mb, buf, other, on4, out, true);
  }
  return 0;
}
