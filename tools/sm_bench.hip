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
__global__ void __launch_bounds__(BLOCK) k_sm(const float* __restrict__ src, float* __restrict__ dst, unsigned n_streams, unsigned T, int dshift = 0)
{
   dst += dshift;                            // write runs start `dshift` floats off the run boundary (dshift = 8: 32 B)
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

// k_sm with the occupancy of the long-run kernel body: a 140 KB LDS allocation lets ONE block (4 waves) live on a CU
template <int R, int D, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_sm_occ(const float* __restrict__ src, float* __restrict__ dst, unsigned n_streams, unsigned T, int dshift)
{
   __shared__ float pad[35840];
   if (T == 0xFFFFFFFFu) { pad[threadIdx.x] = 1.f; __syncthreads(); dst[threadIdx.x] = pad[(threadIdx.x * 7 + 1) % 35840]; }
   dst += dshift;
   constexpr int PP = R / 4, NP = 64 * PP / 64;
   unsigned blk = blockIdx.x;
   {
      const unsigned nb = gridDim.x, xcd = blk & 7u, idx = blk >> 3, q = nb >> 3, r = nb & 7u;
      blk = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
   }
   const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
   const size_t s_base = ((size_t)blk * (BLOCK / 64) + wave) * 64;
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

// read runs of RI floats per stream and phase, write runs of RO floats (RI = k * RO): the wave reads a long run, then
// writes it back as k shorter runs spread over k sub-phases (other streams' pieces in between), or the reverse
template <int RI, int RO, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_sm_rw(const float* __restrict__ src, float* __restrict__ dst, unsigned n_streams, unsigned T)
{
   constexpr int R = RI > RO ? RI : RO;      // floats per stream and phase
   constexpr int PP = R / 4, NP = 64 * PP / 64;
   unsigned blk = blockIdx.x;
   {
      const unsigned nb = gridDim.x, xcd = blk & 7u, idx = blk >> 3, q = nb >> 3, r = nb & 7u;
      blk = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
   }
   const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
   const size_t s_base = ((size_t)blk * (BLOCK / 64) + wave) * 64;
   if (s_base >= n_streams) return;
   const unsigned nch = T / R;
   // piece i of lane: read mapping uses runs of RI, write mapping runs of RO
   auto off_of = [&](int i, int RUN, unsigned c) -> size_t {
      // the phase moves R floats per stream as R/RUN sub-runs; sub-run j of all 64 streams is one "sweep"
      constexpr int dummy = 0; (void)dummy;
      const int ppr = RUN / 4;                       // pieces per run
      const unsigned e = (unsigned)i * 64u + lane;   // piece index in the phase: [sub-run j][stream sl][piece q]
      const unsigned j = e / (64u * ppr), r = e - j * 64u * ppr, sl = r / ppr, q = r - sl * ppr;
      return (s_base + sl) * (size_t)T + (size_t)c * R + j * RUN + q * 4;
   };
   f4 a[NP], b[NP];
   if (nch > 0)
#pragma unroll
      for (int i = 0; i < NP; ++i) a[i] = __builtin_nontemporal_load((const f4*)(src + off_of(i, RI, 0)));
   for (unsigned c = 0; c < nch; c += 2) {
      if (c + 1 < nch)
#pragma unroll
         for (int i = 0; i < NP; ++i) b[i] = __builtin_nontemporal_load((const f4*)(src + off_of(i, RI, c + 1)));
      // (values are written where the WRITE mapping says: a pure bandwidth test, contents do not matter)
#pragma unroll
      for (int i = 0; i < NP; ++i) __builtin_nontemporal_store(a[i], (f4*)(dst + off_of(i, RO, c)));
      if (c + 2 < nch)
#pragma unroll
         for (int i = 0; i < NP; ++i) a[i] = __builtin_nontemporal_load((const f4*)(src + off_of(i, RI, c + 2)));
      if (c + 1 < nch)
#pragma unroll
         for (int i = 0; i < NP; ++i) __builtin_nontemporal_store(b[i], (f4*)(dst + off_of(i, RO, c + 1)));
   }
}

// read-only (sum into one float per lane) / write-only walks with runs of R floats
template <int R, int BLOCK, bool WRITE>
__global__ void __launch_bounds__(BLOCK) k_sm_one(const float* __restrict__ src, float* __restrict__ dst, unsigned n_streams, unsigned T)
{
   constexpr int PP = R / 4, NP = 64 * PP / 64;
   unsigned blk = blockIdx.x;
   {
      const unsigned nb = gridDim.x, xcd = blk & 7u, idx = blk >> 3, q = nb >> 3, r = nb & 7u;
      blk = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
   }
   const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
   const size_t s_base = ((size_t)blk * (BLOCK / 64) + wave) * 64;
   if (s_base >= n_streams) return;
   const unsigned nch = T / R;
   size_t off[NP];
#pragma unroll
   for (int i = 0; i < NP; ++i) {
      const unsigned e = i * 64 + lane, sl = e / PP, q = e - sl * PP;
      off[i] = (s_base + sl) * (size_t)T + q * 4;
   }
   f4 acc = {0, 0, 0, 0};
   for (unsigned c = 0; c < nch; ++c) {
      if (WRITE) {
#pragma unroll
         for (int i = 0; i < NP; ++i) __builtin_nontemporal_store(acc, (f4*)(dst + off[i] + (size_t)c * R));
      } else {
         f4 v[NP];
#pragma unroll
         for (int i = 0; i < NP; ++i) v[i] = __builtin_nontemporal_load((const f4*)(src + off[i] + (size_t)c * R));
#pragma unroll
         for (int i = 0; i < NP; ++i) acc += v[i];
      }
   }
   if (!WRITE && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) dst[s_base + lane] = acc[0];
}

// 128-byte pieces per stream and chunk as the kernel reads them today, but every 4th chunk the wave first TOUCHES the
// next 512-byte run of every stream (one dword per 128-byte line, default cache policy) so that DRAM sees long runs and
// the later piece loads hit in L2 / Infinity Cache
template <int BLOCK, bool TOUCH>
__global__ void __launch_bounds__(BLOCK) k_sm_touch(const float* __restrict__ src, float* __restrict__ dst, unsigned n_streams, unsigned T)
{
   constexpr int R = 32, PP = 8, NP = 8;
   unsigned blk = blockIdx.x;
   {
      const unsigned nb = gridDim.x, xcd = blk & 7u, idx = blk >> 3, q = nb >> 3, r = nb & 7u;
      blk = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
   }
   const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
   const size_t s_base = ((size_t)blk * (BLOCK / 64) + wave) * 64;
   if (s_base >= n_streams) return;
   const unsigned nch = T / R;
   size_t off[NP];
#pragma unroll
   for (int i = 0; i < NP; ++i) {
      const unsigned e = i * 64 + lane, sl = e / PP, q = e - sl * PP;
      off[i] = (s_base + sl) * (size_t)T + q * 4;
   }
   float sink = 0.f;
   f4 a[NP], b[NP];
#pragma unroll
   for (int i = 0; i < NP; ++i) a[i] = __builtin_nontemporal_load((const f4*)(src + off[i]));
   for (unsigned c = 0; c < nch; c += 2) {
      if (TOUCH && (c & 3u) == 0 && c + 8 <= nch) {
         // lines of chunks c+4 .. c+7 (the run after the next one): 64 streams x 4 lines = 256 touches = 4 per lane
#pragma unroll
         for (int k = 0; k < 4; ++k) {
            const unsigned e = k * 64 + lane, sl = e >> 2, ln = e & 3u;
            sink += src[(s_base + sl) * (size_t)T + (size_t)(c + 4 + ln) * R];
         }
      }
      if (c + 1 < nch)
#pragma unroll
         for (int i = 0; i < NP; ++i) b[i] = __builtin_nontemporal_load((const f4*)(src + off[i] + (size_t)(c + 1) * R));
#pragma unroll
      for (int i = 0; i < NP; ++i) __builtin_nontemporal_store(a[i], (f4*)(dst + off[i] + (size_t)c * R));
      if (c + 2 < nch)
#pragma unroll
         for (int i = 0; i < NP; ++i) a[i] = __builtin_nontemporal_load((const f4*)(src + off[i] + (size_t)(c + 2) * R));
      if (c + 1 < nch)
#pragma unroll
         for (int i = 0; i < NP; ++i) __builtin_nontemporal_store(b[i], (f4*)(dst + off[i] + (size_t)(c + 1) * R));
   }
   if (sink == 12345.678f) dst[s_base + lane] = sink;
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
#define RW(RI, RO, B) cases.push_back({"stream-major read runs " + std::to_string(RI * 4) + " B, write runs " + std::to_string(RO * 4) + " B, blk=" #B, \
      [&] { k_sm_rw<RI, RO, B><<<dim3((ns / 64 + B / 64 - 1) / (B / 64)), dim3(B)>>>(s, d, ns, T); }, {}});
   RW(128, 128, 256) RW(128, 32, 256) RW(32, 128, 256) RW(128, 64, 256) RW(64, 128, 256) RW(32, 32, 256)
#define ONE(R, B, W) cases.push_back({std::string(W ? "write-only" : "read-only") + " runs " + std::to_string(R * 4) + " B blk=" #B " (GB/s of ONE direction x2 shown)", \
      [&] { k_sm_one<R, B, W><<<dim3((ns / 64 + B / 64 - 1) / (B / 64)), dim3(B)>>>(s, d, ns, T); }, {}});
   ONE(32, 256, false) ONE(128, 256, false) ONE(32, 256, true) ONE(128, 256, true)
   cases.push_back({"128 B pieces, no touch (baseline of the next line)", [&] { k_sm_touch<256, false><<<dim3(ns / 256), dim3(256)>>>(s, d, ns, T); }, {}});
   cases.push_back({"128 B pieces + 512 B run touched one run ahead", [&] { k_sm_touch<256, true><<<dim3(ns / 256), dim3(256)>>>(s, d, ns, T); }, {}});
   cases.push_back({"stream-major R=128 (512 B runs), WRITE runs shifted by 32 B (T-128: rate shown is 3 % high)", [&] { k_sm<128, 2, 256, 64><<<dim3(ns / 256), dim3(256)>>>(s, d, ns, T - 128, 8); }, {}});
   cases.push_back({"stream-major R=128 (512 B runs), blk=64", [&] { k_sm<128, 2, 64, 64><<<dim3(ns / 64), dim3(64)>>>(s, d, ns, T); }, {}});
   cases.push_back({"R=128 (512 B runs) D=2, ONE block of 4 waves per CU (140 KB LDS)", [&] { k_sm_occ<128, 2, 256><<<dim3(ns / 256), dim3(256)>>>(s, d, ns, T, 0); }, {}});
   cases.push_back({"R=128 (512 B runs) D=2, ONE block per CU, write runs shifted by 128 B (whole line; T-128: 3 % high)", [&] { k_sm_occ<128, 2, 256><<<dim3(ns / 256), dim3(256)>>>(s, d, ns, T - 128, 32); }, {}});
   cases.push_back({"R=32 (128 B runs) D=2, ONE block per CU", [&] { k_sm_occ<32, 2, 256><<<dim3(ns / 256), dim3(256)>>>(s, d, ns, T, 0); }, {}});
   cases.push_back({"R=128 (512 B runs), write runs shifted by 128 B (whole line; T-128: 3 % high)", [&] { k_sm<128, 2, 256, 64><<<dim3(ns / 256), dim3(256)>>>(s, d, ns, T - 128, 32); }, {}});
   cases.push_back({"R=128 (512 B runs), write runs shifted by 64 B (T-128: 3 % high)", [&] { k_sm<128, 2, 256, 64><<<dim3(ns / 256), dim3(256)>>>(s, d, ns, T - 128, 16); }, {}});
   cases.push_back({"R=128 (512 B runs), READ runs shifted by 32 B, writes aligned (T-128: 3 % high)", [&] { k_sm<128, 2, 256, 64><<<dim3(ns / 256), dim3(256)>>>(s + 8, d, ns, T - 128, 0); }, {}});
   cases.push_back({"R=128 (512 B runs), READ runs shifted by 16 B, writes aligned (T-128: 3 % high)", [&] { k_sm<128, 2, 256, 64><<<dim3(ns / 256), dim3(256)>>>(s + 4, d, ns, T - 128, 0); }, {}});
   cases.push_back({"R=128, ONE block per CU, READ runs shifted by 32 B (T-128: 3 % high)", [&] { k_sm_occ<128, 2, 256><<<dim3(ns / 256), dim3(256)>>>(s + 8, d, ns, T - 128, 0); }, {}});
   cases.push_back({"R=128 aligned, T-128 (reference for the shifted lines: 3 % high)", [&] { k_sm<128, 2, 256, 64><<<dim3(ns / 256), dim3(256)>>>(s, d, ns, T - 128, 0); }, {}});
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
