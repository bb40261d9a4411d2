// tools/microbench/gather_lane.hip — how fast is "one lane per query" candidate streaming on gfx950?
// Every lane walks its own contiguous list of LEN 16-byte grid entries (random start inside its sequence's 480 KB grid, the way
// a lane-per-query k_associate would walk a 3x3x3 block of hash buckets), computes the squared distance and keeps the minimum.
// Compared with the wave-per-query form: the 64 lanes of a wave read 64 CONSECUTIVE entries of one list.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_lane gather_lane.hip ; run: ./gather_lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kSeq = 512, kEntries = 30000, kQueries = 1224, kLen = 150;

__global__ __launch_bounds__(256) void k_lane(const float4* __restrict__ grid, const int* __restrict__ starts, float* __restrict__ out, int len) {
  const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  if (q >= kQueries) return;
  const float4* g = grid + (size_t)b * kEntries;
  const int s = starts[b * kQueries + q];
  const float sx = 1.f + q * 1e-3f, sy = 2.f, sz = 0.5f;
  float best = 3e38f; int bi = -1;
  for (int k = 0; k < len; k += 4) {
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = g[s + k + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dx = p[u].x - sx, dy = p[u].y - sy, dz = p[u].z - sz;
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < best) { best = d; bi = __float_as_int(p[u].w); }
    }
  }
  out[b * kQueries + q] = best + bi;
}

// wave per query: 64 lanes read 64 consecutive entries, then a wave-min
__global__ __launch_bounds__(256) void k_wave(const float4* __restrict__ grid, const int* __restrict__ starts, float* __restrict__ out, int len) {
  const int b = blockIdx.y, q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= kQueries) return;
  const float4* g = grid + (size_t)b * kEntries;
  const int s = starts[b * kQueries + q];
  const float sx = 1.f + q * 1e-3f, sy = 2.f, sz = 0.5f;
  float best = 3e38f;
  for (int k = lane; k < len; k += 64) {
    const float4 p = g[s + k];
    const float dx = p.x - sx, dy = p.y - sy, dz = p.z - sz;
    best = fminf(best, dx * dx + dy * dy + dz * dz);
  }
  for (int d = 32; d > 0; d >>= 1) best = fminf(best, __shfl_xor(best, d, 64));
  if (lane == 0) out[b * kQueries + q] = best;
}

int main() {
  std::vector<float4> h((size_t)kSeq * kEntries);
  for (size_t i = 0; i < h.size(); ++i) h[i] = make_float4(drand48() * 100, drand48() * 100, drand48() * 3, (float)i);
  std::vector<int> st((size_t)kSeq * kQueries);
  for (auto& v : st) v = (int)(drand48() * (kEntries - 2 * kLen));
  float4* d; int* ds; float* o;
  hipMalloc(&d, h.size() * 16); hipMalloc(&ds, st.size() * 4); hipMalloc(&o, st.size() * 4);
  hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice); hipMemcpy(ds, st.data(), st.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int len : {48, 152, 304}) {
    for (int mode = 0; mode < 2; ++mode) {
      float best_ms = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k_lane, dim3((kQueries + 255) / 256, kSeq), dim3(256), 0, 0, d, ds, o, len);
        else hipLaunchKernelGGL(k_wave, dim3((kQueries + 3) / 4, kSeq), dim3(256), 0, 0, d, ds, o, len);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best_ms) best_ms = ms;
      }
      printf("%s len=%d: %.3f ms  (%.1f G candidates/s)\n", mode == 0 ? "lane-per-query" : "wave-per-query", len, best_ms, (double)kSeq * kQueries * len / best_ms / 1e6);
    }
  }
  return 0;
}
