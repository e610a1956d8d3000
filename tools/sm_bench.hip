// Dev microbenchmark (GPU box): what HBM gives a STREAM-MAJOR walk -- every wave owns 64 adjacent streams
// ([stream][t] buffers, one contiguous row per stream) and moves them chunk by chunk, R floats per stream and
// chunk, as float4 pieces laid along the rows (the access pattern of fz_block_kernel's FZ_VF_STREAM_MAJOR body).
// Pure copy (same piece mapping in and out), so no LDS: this isolates the memory-side ceiling as a function of
// the run length R, the number of chunks in flight and the workgroup shape.
// build: hipcc --offload-arch=gfx950 -O3 tools/sm_bench.hip -o tools/_bin/sm_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// R floats per stream and chunk, D chunks in flight (register buffers), BLOCK threads, SW streams per wave
template <int R, int D, int BLOCK, int SW>
__global__ void __launch_bounds__(BLOCK) k_sm(const float* __restrict__ src, float* __restrict__ dst, unsigned n_streams, unsigned T)
{
   constexpr int PP = R / 4;                 // float4 pieces per stream and chunk
   constexpr int NP = SW * PP / 64;          // pieces per lane and chunk
   unsigned blk = blockIdx.x;
   {
      const unsigned nb = gridDim.x, xcd = blk & 7u, idx = blk >> 3, q = nb >> 3, r = nb & 7u;
      blk = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
   }
   const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
   const size_t s_base = ((size_t)blk * (BLOCK / 64) + wave) * SW;
   if (s_base >= n_streams) return;
   const unsigned nch = T / R;
   f4 buf[D][NP];
   size_t off[NP];
#pragma unroll
   for (int i = 0; i < NP; ++i) {
      const unsigned e = i * 64 + lane, sl = e / PP, q = e - sl * PP;
      off[i] = (s_base + sl) * (size_t)T + q * 4;
   }
#pragma unroll
   for (int d = 0; d < D - 1; ++d)
      if (d < (int)nch)
#pragma unroll
         for (int i = 0; i < NP; ++i) buf[d][i] = __builtin_nontemporal_load((const f4*)(src + off[i] + (size_t)d * R));
   for (unsigned c = 0; c < nch; c += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
         const unsigned cc = c + d;
         if (cc + D - 1 < nch)
#pragma unroll
            for (int i = 0; i < NP; ++i) buf[(d + D - 1) % D][i] = __builtin_nontemporal_load((const f4*)(src + off[i] + (size_t)(cc + D - 1) * R));
         if (cc < nch)
#pragma unroll
            for (int i = 0; i < NP; ++i) __builtin_nontemporal_store(buf[d][i], (f4*)(dst + off[i] + (size_t)cc * R));
      }
   }
}

// frames with the stream index fastest, tiled (the frame kernel's pattern): lane owns W floats
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
   const unsigned ns = argc > 1 ? atol(argv[1]) : (1u << 20);
   const unsigned T = argc > 2 ? atoi(argv[2]) : 4096;
   const int rounds = argc > 3 ? atoi(argv[3]) : 5;
   const size_t nf = (size_t)ns * T, bytes = nf * 4;
   float *s, *d;
   hipMalloc(&s, bytes); hipMalloc(&d, bytes);
   hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
   std::vector<Case> cases;
#define SM(R, D, B, SW) cases.push_back({"stream-major R=" #R " (" + std::to_string(R * 4) + " B runs) D=" #D " blk=" #B " streams/wave=" #SW, \
      [&] { k_sm<R, D, B, SW><<<dim3((ns / SW + B / 64 - 1) / (B / 64)), dim3(B)>>>(s, d, ns, T); }, {}});
   SM(16, 2, 256, 64) SM(32, 2, 256, 64) SM(64, 2, 256, 64) SM(128, 2, 256, 64)
   SM(32, 3, 256, 64) SM(32, 4, 256, 64) SM(64, 3, 256, 64)
   SM(32, 2, 64, 64) SM(32, 2, 128, 64) SM(64, 2, 64, 64) SM(64, 2, 128, 64)
   SM(64, 2, 256, 32) SM(128, 2, 256, 32) SM(128, 2, 256, 16) SM(256, 2, 256, 16) SM(256, 2, 256, 8)
   const size_t row = ns;
   cases.push_back({"frames tiled 8192 W=1 U=16", [&] { k_walk<1, 16, 256><<<dim3((row + 255) / 256), dim3(256)>>>(s, d, row, 8192, 8192, (size_t)8192 * T, T); }, {}});
   cases.push_back({"frames tiled 8192 W=2 U=16", [&] { k_walk<2, 16, 256><<<dim3((row / 2 + 255) / 256), dim3(256)>>>(s, d, row / 2, 4096, 8192, (size_t)8192 * T, T); }, {}});
   for (auto& c : cases) c.run();
   hipDeviceSynchronize();
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   for (int r = 0; r < rounds; ++r)
      for (auto& c : cases) {
         hipEventRecord(e0); c.run(); hipEventRecord(e1); hipEventSynchronize(e1);
         float ms; hipEventElapsedTime(&ms, e0, e1); c.ms.push_back(ms);
      }
   printf("# tools/sm_bench: copy of %u streams x %u samples (%.1f GiB each way), %d interleaved rounds (median / best)\n", ns, T, bytes / 1073741824.0, rounds);
   for (auto& c : cases) {
      std::sort(c.ms.begin(), c.ms.end());
      float med = c.ms[c.ms.size() / 2], mn = c.ms[0];
      printf("%-72s %7.3f ms  %7.1f GB/s   (best %7.1f)\n", c.name.c_str(), med, 2.0 * bytes / med / 1e6, 2.0 * bytes / mn / 1e6);
   }
   return 0;
}
