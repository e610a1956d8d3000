// Dev microbenchmark (GPU box): stream-major buffers WITHOUT the LDS transposition -- a lane owns one stream and fetches its
// own run of R floats with R/4 float4 loads (every load instruction: 64 lanes x 16 B in 64 different lines, 16 KiB apart; eight
// consecutive instructions walk the same 64 lines), keeps it in registers and writes it back the same way.  Pure copy: what does
// the memory system make of that pattern?   build: hipcc --offload-arch=gfx950 -O3 tools/sm_direct_bench.hip -o tools/_bin/sm_direct_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int R, bool NT_LD, bool NT_ST, int OCC_LDS>
__global__ void __launch_bounds__(256) k_direct(const float* __restrict__ src, float* __restrict__ dst, unsigned n_streams, unsigned T)
{
   __shared__ float pad[OCC_LDS > 0 ? OCC_LDS : 1];
   if (T == 0xFFFFFFFFu) { pad[threadIdx.x] = 1.f; __syncthreads(); dst[threadIdx.x] = pad[(threadIdx.x * 7 + 1) % (OCC_LDS > 0 ? OCC_LDS : 1)]; }
   unsigned blk = blockIdx.x;
   {
      const unsigned nb = gridDim.x, xcd = blk & 7u, idx = blk >> 3, q = nb >> 3, r = nb & 7u;
      blk = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
   }
   const size_t s = (size_t)blk * 256 + threadIdx.x;
   if (s >= n_streams) return;
   constexpr int NP = R / 4;
   const f4* in = (const f4*)(src + s * (size_t)T);
   f4* out = (f4*)(dst + s * (size_t)T);
   f4 a[NP], b[NP];
   const unsigned nph = T / R;
#define LD(buf, p) _Pragma("unroll") for (int i = 0; i < NP; ++i) buf[i] = NT_LD ? __builtin_nontemporal_load(in + (size_t)(p) * NP + i) : in[(size_t)(p) * NP + i];
#define ST(buf, p) _Pragma("unroll") for (int i = 0; i < NP; ++i) { if (NT_ST) __builtin_nontemporal_store(buf[i], out + (size_t)(p) * NP + i); else out[(size_t)(p) * NP + i] = buf[i]; }
   LD(a, 0)
   for (unsigned p = 0; p < nph; p += 2) {
      if (p + 1 < nph) { LD(b, p + 1) }
      ST(a, p)
      if (p + 2 < nph) { LD(a, p + 2) }
      if (p + 1 < nph) { ST(b, p + 1) }
   }
}

int main(int argc, char** argv)
{
   const unsigned ns = argc > 1 ? atoi(argv[1]) : 1 << 20, T = argc > 2 ? atoi(argv[2]) : 4096;
   float *s, *d;
   const size_t n = (size_t)ns * T;
   hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
   hipMemset(s, 1, n * 4); hipMemset(d, 0, n * 4);
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   auto run = [&](const char* name, auto f) {
      f(); hipDeviceSynchronize();
      std::vector<float> ts;
      for (int r = 0; r < 5; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
      std::sort(ts.begin(), ts.end());
      printf("%-78s %7.3f ms  %7.1f GB/s\n", name, ts[2], 2.0 * n * 4 / ts[2] / 1e6);
   };
   const dim3 g((ns + 255) / 256), b(256);
   run("direct R=128 (512 B per lane), nt loads, nt stores, free occupancy", [&] { k_direct<128, true, true, 0><<<g, b>>>(s, d, ns, T); });
   run("direct R=128, nt loads, plain stores (write-back L2), free occupancy", [&] { k_direct<128, true, false, 0><<<g, b>>>(s, d, ns, T); });
   run("direct R=128, plain loads, plain stores, free occupancy", [&] { k_direct<128, false, false, 0><<<g, b>>>(s, d, ns, T); });
   run("direct R=128, plain loads, plain stores, ONE block of 4 waves per CU", [&] { k_direct<128, false, false, 35000><<<g, b>>>(s, d, ns, T); });
   run("direct R=128, nt loads, plain stores, ONE block per CU", [&] { k_direct<128, true, false, 35000><<<g, b>>>(s, d, ns, T); });
   run("direct R=128, nt loads, nt stores, ONE block per CU", [&] { k_direct<128, true, true, 35000><<<g, b>>>(s, d, ns, T); });
   run("direct R=64 (256 B per lane), plain, TWO blocks per CU", [&] { k_direct<64, false, false, 18000><<<g, b>>>(s, d, ns, T); });
   run("direct R=64, plain, free occupancy", [&] { k_direct<64, false, false, 0><<<g, b>>>(s, d, ns, T); });
   run("direct R=32 (128 B per lane), plain, free occupancy", [&] { k_direct<32, false, false, 0><<<g, b>>>(s, d, ns, T); });
   return 0;
}
