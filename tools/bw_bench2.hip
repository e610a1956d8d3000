// Dev microbenchmark (GPU box): interleaved-median comparison of HBM copy patterns, 16 GiB each way.
// build: hipcc --offload-arch=gfx950 -O3 tools/bw_bench2.hip -o tools/_bin/bw_bench2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int UN>
__global__ void __launch_bounds__(256) k_oneshot(const f4* __restrict__ s, f4* __restrict__ d)
{
   size_t i = (size_t)blockIdx.x * 256 * UN + threadIdx.x;
   f4 v[UN];
#pragma unroll
   for (int u = 0; u < UN; ++u) v[u] = __builtin_nontemporal_load(s + i + u * 256);
#pragma unroll
   for (int u = 0; u < UN; ++u) __builtin_nontemporal_store(v[u], d + i + u * 256);
}

// generic walk: lane owns W floats; element (t, lane g) at  base(g) + t * tstride
template <int W, int U, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_walk(const float* __restrict__ s, float* __restrict__ d, size_t n_lanes,
                                                size_t tile_lanes, size_t tstride, size_t tile_stride, int T)
{
   typedef float vw __attribute__((ext_vector_type(W)));
   const size_t g = (size_t)blockIdx.x * BLOCK + threadIdx.x;
   if (g >= n_lanes) return;
   const size_t base = (g / tile_lanes) * tile_stride + (g % tile_lanes) * W;
   for (int t = 0; t < T; t += U) {
      vw v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const vw*)(s + base + (size_t)(t + u) * tstride));
#pragma unroll
      for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], (vw*)(d + base + (size_t)(t + u) * tstride));
   }
}

struct Case { std::string name; std::function<void()> run; std::vector<float> ms; };

int main(int argc, char** argv)
{
   const size_t gib = argc > 1 ? atol(argv[1]) : 16;
   const int rounds = argc > 2 ? atoi(argv[2]) : 7;
   const size_t bytes = gib << 30, nf = bytes / 4;
   float *s, *d;
   hipMalloc(&s, bytes); hipMalloc(&d, bytes);
   hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
   const int T = 4096;
   const size_t row = nf / T;                      // floats per time-major row
   std::vector<Case> cases;
   cases.push_back({"oneshot un4", [&] { k_oneshot<4><<<dim3(nf / 4 / 1024), dim3(256)>>>((const f4*)s, (f4*)d); }, {}});
   cases.push_back({"oneshot un2", [&] { k_oneshot<2><<<dim3(nf / 4 / 512), dim3(256)>>>((const f4*)s, (f4*)d); }, {}});
#define ROWS(W, U, B) cases.push_back({"rows  W=" #W " U=" #U " blk=" #B, [&] { k_walk<W, U, B><<<dim3((row / W + B - 1) / B), dim3(B)>>>(s, d, row / W, row / W, row, 0, T); }, {}});
#define TILED(W, U, B, TL) cases.push_back({"tiled W=" #W " U=" #U " blk=" #B " tile=" #TL "lanes", [&] { k_walk<W, U, B><<<dim3((row / W + B - 1) / B), dim3(B)>>>(s, d, row / W, TL, (size_t)TL * W, (size_t)TL * W * T, T); }, {}});
   ROWS(1, 16, 256) ROWS(2, 8, 256)
   if (argc > 3) {   // scan of tile widths
      TILED(2, 8, 256, 512) TILED(2, 8, 256, 1024) TILED(2, 8, 256, 2048) TILED(2, 8, 256, 4096) TILED(2, 8, 256, 8192) TILED(2, 8, 256, 16384)
      TILED(2, 8, 256, 32768) TILED(2, 8, 256, 131072) TILED(2, 8, 256, 262144)
      TILED(1, 16, 256, 1024) TILED(1, 16, 256, 2048) TILED(1, 16, 256, 4096) TILED(1, 16, 256, 8192) TILED(1, 16, 256, 16384) TILED(1, 16, 256, 32768) TILED(1, 16, 256, 65536)
      TILED(4, 4, 256, 512) TILED(4, 4, 256, 1024) TILED(4, 4, 256, 2048) TILED(4, 4, 256, 4096) TILED(4, 4, 256, 8192)
   }
   for (auto& c : cases) c.run();
   hipDeviceSynchronize();
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   for (int r = 0; r < rounds; ++r)
      for (auto& c : cases) {
         hipEventRecord(e0); c.run(); hipEventRecord(e1); hipEventSynchronize(e1);
         float ms; hipEventElapsedTime(&ms, e0, e1); c.ms.push_back(ms);
      }
   printf("copy %zu GiB each way, T=%d rows of %zu floats, %d interleaved rounds (median / min)\n", gib, T, row, rounds);
   for (auto& c : cases) {
      std::sort(c.ms.begin(), c.ms.end());
      float med = c.ms[c.ms.size() / 2], mn = c.ms[0];
      printf("%-44s %7.3f ms  %7.1f GB/s   (best %7.1f)\n", c.name.c_str(), med, 2.0 * bytes / med / 1e6, 2.0 * bytes / mn / 1e6);
   }
   return 0;
}
