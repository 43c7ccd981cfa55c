// catre_pcl.h - SURVEY.md row f3: point-cloud preparation, the step right before the refine path.  Per instance the
// reference's data loader (core/catre/datasets/data_loader.py:576-603) back-projects the depth map
// (lib/pysixd/misc.py:360-378), keeps the pixels of the instance mask with depth > 0
// (core/utils/cat_data_utils.py:209-226), crops a ball around the pose centre whose radius grows x1.1 until it
// holds >= 10 points (:289-311) and samples NUM_PCL of them (:313-320) - all on the CPU, instance by instance.
// Here: all instances of an image in four launches.
//   k_pcl_count   per (instance, 4096-pixel chunk): how many valid pixels fall into each of the 10 candidate radii
//   k_pcl_pick    per instance: choose the radius like the reference's loop, exclusive scan of the chunk counts
//   k_pcl_compact per (instance, chunk): ordered (row-major, = torch.nonzero order) list of candidate pixels
//   k_pcl_gather  per sample slot: pixel -> 3-D point; the slot -> candidate map is either the caller's
//                 torch.randperm (reference-exact random stream) or a keyed Feistel permutation (no host sync)
#pragma once

#define PCL_CHUNK 4096
#define PCL_NR 10  // radii tried by crop_ball_from_pts: r0 * 1.1^i, i = 0..9

struct PclCam {
  float fx, fy, cx, cy;
};

__device__ __forceinline__ void pcl_point(const float* __restrict__ depth, int pix, int W, const PclCam& cam, float& x,
                                          float& y, float& z) {
  const int v = pix / W, u = pix - v * W;
  z = depth[pix];
  x = ((float)u - cam.cx) * z / cam.fx;  // X * depth / K[0,0]   (misc.py:378)
  y = ((float)v - cam.cy) * z / cam.fy;
}

// radii of instance i as the reference computes them: radius = ratio * ||R s||, max(radius, 0.05), then *= 1.10
// (fp32 tensor arithmetic when radius > 0.05, python-float (double) arithmetic when the 0.05 floor wins)
__device__ __forceinline__ void pcl_radii(const float* __restrict__ pose, const float* __restrict__ scale, float ratio,
                                          float (&r)[PCL_NR]) {
  float v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = pose[k * 4] * scale[0] + pose[k * 4 + 1] * scale[1] + pose[k * 4 + 2] * scale[2];
  const float r0 = ratio * sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (r0 > 0.05f) {
    float t = r0;
#pragma unroll
    for (int i = 0; i < PCL_NR; ++i) {
      r[i] = t;
      t *= 1.10f;
    }
  } else {
    double t = 0.05;
#pragma unroll
    for (int i = 0; i < PCL_NR; ++i) {
      r[i] = (float)t;
      t *= 1.10;
    }
  }
}

// bin of one pixel: 0..9 = smallest radius index that contains it, 10 = valid but outside all, -1 = not a candidate
__device__ __forceinline__ int pcl_bin(const float* __restrict__ depth, const unsigned char* __restrict__ mask, int pix,
                                       int W, const PclCam& cam, const float* centre, const float (&r)[PCL_NR],
                                       int use_ball) {
  if (mask && !mask[pix]) return -1;
  float x, y, z;
  pcl_point(depth, pix, W, cam, x, y, z);
  if (!(z > 0.f)) return -1;
  if (!use_ball) return 0;
  const float dx = x - centre[0], dy = y - centre[1], dz = z - centre[2];
  const float d = sqrtf(dx * dx + dy * dy + dz * dz);
  int b = PCL_NR;
#pragma unroll
  for (int i = PCL_NR - 1; i >= 0; --i)
    if (d <= r[i]) b = i;
  return b;
}

__global__ __launch_bounds__(256) void k_pcl_count(const float* __restrict__ depth,
                                                   const unsigned char* __restrict__ masks, const float* __restrict__ poses,
                                                   const float* __restrict__ scales, PclCam cam, float ratio, int use_ball,
                                                   int H, int W, int nchunks, int* __restrict__ bins /*[I][nchunks][12]*/) {
  __shared__ int cnt[12];
  const int inst = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  if (tid < 12) cnt[tid] = 0;
  __syncthreads();
  float r[PCL_NR];
  pcl_radii(poses + inst * 12, scales + inst * 3, ratio, r);
  const float centre[3] = {poses[inst * 12 + 3], poses[inst * 12 + 7], poses[inst * 12 + 11]};
  const unsigned char* m = masks ? masks + (size_t)inst * H * W : nullptr;
  const int p0 = chunk * PCL_CHUNK + tid * 16, HW = H * W;
  int local[PCL_NR + 1];
#pragma unroll
  for (int i = 0; i <= PCL_NR; ++i) local[i] = 0;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int pix = p0 + k;
    if (pix < HW) {
      const int b = pcl_bin(depth, m, pix, W, cam, centre, r, use_ball);
#pragma unroll
      for (int i = 0; i <= PCL_NR; ++i) local[i] += (b == i);
    }
  }
#pragma unroll
  for (int i = 0; i <= PCL_NR; ++i)
    if (local[i]) atomicAdd(&cnt[i], local[i]);
  __syncthreads();
  if (tid < 12) bins[((size_t)inst * nchunks + chunk) * 12 + tid] = tid <= PCL_NR ? cnt[tid] : 0;
}

// choose[inst] = k (0..9) or 10 (all valid pixels: the `distance <= 1e9` fallback);  offsets = exclusive scan of the
// per-chunk candidate counts under that choice; total[inst].
__global__ __launch_bounds__(256) void k_pcl_pick(const int* __restrict__ bins, int nchunks, int need_points,
                                                  int* __restrict__ choose, int* __restrict__ offsets,
                                                  int* __restrict__ total) {
  __shared__ int tot[12];
  __shared__ int part[256];
  __shared__ int sel;
  const int inst = blockIdx.x, tid = threadIdx.x;
  const int* b = bins + (size_t)inst * nchunks * 12;
  if (tid < 12) tot[tid] = 0;
  __syncthreads();
  int loc[PCL_NR + 1];
#pragma unroll
  for (int i = 0; i <= PCL_NR; ++i) loc[i] = 0;
  for (int c = tid; c < nchunks; c += 256)
#pragma unroll
    for (int i = 0; i <= PCL_NR; ++i) loc[i] += b[c * 12 + i];
#pragma unroll
  for (int i = 0; i <= PCL_NR; ++i)
    if (loc[i]) atomicAdd(&tot[i], loc[i]);
  __syncthreads();
  if (tid == 0) {
    // for i in range(10): idx = where(d <= radius); if len(idx) >= 10 or num_points is None: break; radius *= 1.1
    int k = PCL_NR - 1, cum = 0;
    for (int i = 0; i < PCL_NR; ++i) {
      cum += tot[i];
      if (cum >= 10 || !need_points) {
        k = i;
        break;
      }
    }
    int n = 0;
    for (int i = 0; i <= k; ++i) n += tot[i];
    if (n == 0 && need_points) k = PCL_NR;  // `if len(idx) == 0: idx = where(distance <= 1e9)`
    sel = k;
    choose[inst] = k;
  }
  __syncthreads();
  const int k = sel;
  // exclusive scan of per-chunk counts (chunks per thread = ceil(nchunks/256), sequential inside a thread)
  const int per = (nchunks + 255) / 256;
  int s = 0;
  for (int c = tid * per; c < min(nchunks, (tid + 1) * per); ++c)
    for (int i = 0; i <= k; ++i) s += b[c * 12 + i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) {
      const int v = part[t];
      part[t] = run;
      run += v;
    }
    total[inst] = run;
  }
  __syncthreads();
  int run = part[tid];
  for (int c = tid * per; c < min(nchunks, (tid + 1) * per); ++c) {
    offsets[(size_t)inst * nchunks + c] = run;
    for (int i = 0; i <= k; ++i) run += b[c * 12 + i];
  }
}

__global__ __launch_bounds__(256) void k_pcl_compact(const float* __restrict__ depth,
                                                     const unsigned char* __restrict__ masks,
                                                     const float* __restrict__ poses, const float* __restrict__ scales,
                                                     PclCam cam, float ratio, int use_ball, int H, int W, int nchunks,
                                                     const int* __restrict__ choose, const int* __restrict__ offsets,
                                                     int* __restrict__ cand /*[I][H*W]*/) {
  __shared__ int wsum[4];
  const int inst = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float r[PCL_NR];
  pcl_radii(poses + inst * 12, scales + inst * 3, ratio, r);
  const float centre[3] = {poses[inst * 12 + 3], poses[inst * 12 + 7], poses[inst * 12 + 11]};
  const unsigned char* m = masks ? masks + (size_t)inst * H * W : nullptr;
  const int k = choose[inst];
  const int p0 = chunk * PCL_CHUNK + tid * 16, HW = H * W;
  unsigned keep = 0;
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const int pix = p0 + j;
    if (pix < HW) {
      const int b = pcl_bin(depth, m, pix, W, cam, centre, r, use_ball);
      if (b >= 0 && b <= k) keep |= 1u << j;
    }
  }
  const int mine = __popc(keep);
  // exclusive scan over the 256 threads (thread order = pixel order)
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = offsets[(size_t)inst * nchunks + chunk];
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int pos = base + incl - mine;
  int* out = cand + (size_t)inst * HW;
#pragma unroll 4
  for (int j = 0; j < 16; ++j)
    if (keep & (1u << j)) out[pos++] = p0 + j;
}

// keyed 4-round Feistel permutation of [0, 2^bits) with cycle walking down to [0, L): slot i -> a distinct index
__device__ __forceinline__ unsigned pcl_mix(unsigned x, unsigned key) {
  x ^= key;
  x *= 0x9E3779B1u;
  x ^= x >> 15;
  x *= 0x85EBCA77u;
  x ^= x >> 13;
  return x;
}
__device__ __forceinline__ unsigned pcl_perm(unsigned i, unsigned L, unsigned long long seed, unsigned inst) {
  int bits = 2;
  while ((1u << bits) < L) ++bits;
  bits += bits & 1;  // even number of bits: two equal halves
  const int hb = bits >> 1;
  const unsigned hm = (1u << hb) - 1;
  unsigned x = i;
  do {
    unsigned l = x >> hb, r = x & hm;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const unsigned f = pcl_mix(r, (unsigned)(seed >> (8 * rd)) ^ (inst * 0x632BE5ABu) ^ (rd * 0x27D4EB2Fu)) & hm;
      const unsigned nl = r;
      r = l ^ f;
      l = nl;
    }
    x = (l << hb) | r;
  } while (x >= L);
  return x;
}

__global__ __launch_bounds__(256) void k_pcl_gather(const float* __restrict__ depth, PclCam cam, int H, int W,
                                                    const int* __restrict__ cand, const int* __restrict__ total,
                                                    const long long* __restrict__ sample /*[I][N] or null*/,
                                                    unsigned long long seed, int N, float* __restrict__ pcl_out,
                                                    int* __restrict__ pix_out) {
  const int inst = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int cnt = total[inst];
  float x = 0.f, y = 0.f, z = 0.f;
  int pix = -1;
  if (cnt > 0) {
    unsigned L = (unsigned)cnt;  // `while len(idx) < num_points: idx = cat([idx, idx])` -> idx tiled to length L
    while (L < (unsigned)N) L <<= 1;
    const unsigned j = sample ? (unsigned)sample[(size_t)inst * N + i] : pcl_perm((unsigned)i, L, seed, (unsigned)inst);
    pix = cand[(size_t)inst * H * W + (j % (unsigned)cnt)];
    pcl_point(depth, pix, W, cam, x, y, z);
  }
  float* o = pcl_out + ((size_t)inst * N + i) * 3;
  o[0] = x;
  o[1] = y;
  o[2] = z;
  if (pix_out) pix_out[(size_t)inst * N + i] = pix;
}

// ------------------------------------------------------------------------------------------------
// INPUT.FPS_SAMPLE: farthest point sampling of the (tiled) candidate list, as the data loader runs it -
// crop_ball_from_pts(..., device="cpu", fps_sample=True) -> core/utils/farthest_points_torch.py:6-62 with
// init_center=True and dist_func = F.pairwise_distance (|| a - b + 1e-6 ||_2):
//   distances = d(mean(points), points);  N times: c = argmax(distances) (first maximum), distances = min(distances, d(points[c], points)).
// One 1024-thread workgroup per instance; the tiled candidates' coordinates and running distances live in the
// caller's scratch (4 floats per slot).  N >= L returns 0 .. L-1 like the reference.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pcl_pdist(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = (ax - bx) + 1e-6f, dy = (ay - by) + 1e-6f, dz = (az - bz) + 1e-6f;
  return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
}

__global__ __launch_bounds__(1024) void k_pcl_fps(const float* __restrict__ depth, PclCam cam, int H, int W,
                                                  const int* __restrict__ cand, const int* __restrict__ total, int N,
                                                  float* __restrict__ scratch, int Lcap,
                                                  long long* __restrict__ sample_out) {
  __shared__ float red_v[16];
  __shared__ int red_i[16];
  __shared__ float red_s[3][16];
  __shared__ float cen[3];
  const int inst = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cnt = total[inst];
  long long* out = sample_out + (size_t)inst * N;
  unsigned L = cnt > 0 ? (unsigned)cnt : 0u;
  while (L && L < (unsigned)N) L <<= 1;
  if (L == 0 || (unsigned)N >= L || L > (unsigned)Lcap) {  // nothing to choose from / everything is chosen
    for (int i = tid; i < N; i += 1024) out[i] = L ? (long long)((unsigned)i % L) : 0;
    return;
  }
  float* px = scratch + (size_t)inst * 4 * Lcap;
  float* py = px + Lcap;
  float* pz = py + Lcap;
  float* dist = pz + Lcap;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (unsigned j = tid; j < L; j += 1024) {
    float x, y, z;
    pcl_point(depth, cand[(size_t)inst * H * W + (j % (unsigned)cnt)], W, cam, x, y, z);
    px[j] = x;
    py[j] = y;
    pz[j] = z;
    sx += x;
    sy += y;
    sz += z;
  }
  sx = wave_sum(sx);
  sy = wave_sum(sy);
  sz = wave_sum(sz);
  if (lane == 0) {
    red_s[0][wave] = sx;
    red_s[1][wave] = sy;
    red_s[2][wave] = sz;
  }
  __syncthreads();
  if (tid < 3) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red_s[tid][w];
    cen[tid] = t / (float)L;
  }
  __syncthreads();
  {
    const float cx = cen[0], cy = cen[1], cz = cen[2];
    for (unsigned j = tid; j < L; j += 1024) dist[j] = pcl_pdist(cx, cy, cz, px[j], py[j], pz[j]);
  }
  for (int it = 0; it < N; ++it) {
    // arg-max with "first maximum wins": per thread in increasing j, then (value, index) pairs
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (unsigned j = tid; j < L; j += 1024) {
      const float v = dist[j];
      if (v > bv) {
        bv = v;
        bi = (int)j;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      red_v[wave] = bv;
      red_i[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      float v = red_v[0];
      int i = red_i[0];
      for (int w = 1; w < 16; ++w)
        if (red_v[w] > v || (red_v[w] == v && red_i[w] < i)) {
          v = red_v[w];
          i = red_i[w];
        }
      out[it] = i;
      cen[0] = px[i];
      cen[1] = py[i];
      cen[2] = pz[i];
    }
    __syncthreads();
    const float cx = cen[0], cy = cen[1], cz = cen[2];
    for (unsigned j = tid; j < L; j += 1024) dist[j] = fminf(dist[j], pcl_pdist(cx, cy, cz, px[j], py[j], pz[j]));
    __syncthreads();
  }
}
