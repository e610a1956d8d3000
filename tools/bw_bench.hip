// Dev microbenchmark (GPU box): achievable HBM copy bandwidth on MI355X for several copy shapes.
// build: hipcc --offload-arch=gfx950 -O3 tools/bw_bench.hip -o tools/_bin/bw_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

// grid-stride, UN loads in flight per thread
template <int UN, bool NT>
__global__ void __launch_bounds__(256) k_gs(const f4* __restrict__ s, f4* __restrict__ d, size_t n4)
{
   size_t i = (size_t)blockIdx.x * 256 * UN + threadIdx.x;
   const size_t stride = (size_t)gridDim.x * 256 * UN;
   for (; i + (UN - 1) * 256 < n4; i += stride) {
      f4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * 256) : s[i + u * 256];
#pragma unroll
      for (int u = 0; u < UN; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * 256); else d[i + u * 256] = v[u]; }
   }
}

// one contiguous chunk per block (no grid stride): blocks = n4 / (256*UN*ITER)
template <int UN, bool NT>
__global__ void __launch_bounds__(256) k_chunk(const f4* __restrict__ s, f4* __restrict__ d, size_t n4, int iters)
{
   size_t i = (size_t)blockIdx.x * 256 * UN * iters + threadIdx.x;
   for (int it = 0; it < iters; ++it, i += 256 * UN) {
      f4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * 256) : s[i + u * 256];
#pragma unroll
      for (int u = 0; u < UN; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * 256); else d[i + u * 256] = v[u]; }
   }
}

// time-major row walk like fz_block_kernel: a wave owns 64*W floats of every row (W = 1, 2, 4)
template <int W, int U, bool NT>
__global__ void __launch_bounds__(256) k_rows(const float* __restrict__ s, float* __restrict__ d, size_t row, int T)
{
   typedef float vw __attribute__((ext_vector_type(W)));
   const size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) * W;
   if (g >= row) return;
   for (int t = 0; t < T; t += U) {
      vw v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const vw* p = (const vw*)(s + (size_t)(t + u) * row + g); v[u] = NT ? __builtin_nontemporal_load(p) : *p; }
#pragma unroll
      for (int u = 0; u < U; ++u) { vw* p = (vw*)(d + (size_t)(t + u) * row + g); if (NT) __builtin_nontemporal_store(v[u], p); else *p = v[u]; }
   }
}

// row walk with a padded row pitch (pitch >= row), to probe DRAM channel/bank aliasing of 2^k strides
template <int W, int U, bool NT>
__global__ void __launch_bounds__(256) k_rows_pitch(const float* __restrict__ s, float* __restrict__ d, size_t row, size_t pitch, int T)
{
   typedef float vw __attribute__((ext_vector_type(W)));
   const size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) * W;
   if (g >= row) return;
   for (int t = 0; t < T; t += U) {
      vw v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const vw* p = (const vw*)(s + (size_t)(t + u) * pitch + g); v[u] = NT ? __builtin_nontemporal_load(p) : *p; }
#pragma unroll
      for (int u = 0; u < U; ++u) { vw* p = (vw*)(d + (size_t)(t + u) * pitch + g); if (NT) __builtin_nontemporal_store(v[u], p); else *p = v[u]; }
   }
}

// tiled walk: a wave owns a CONTIGUOUS region [T][64*W] and streams through it
template <int W, int U, bool NT>
__global__ void __launch_bounds__(256) k_tiled(const float* __restrict__ s, float* __restrict__ d, size_t n_waves, int T)
{
   typedef float vw __attribute__((ext_vector_type(W)));
   const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
   if (wave >= n_waves) return;
   const size_t base = wave * (size_t)T * 64 * W + (threadIdx.x & 63) * W;
   for (int t = 0; t < T; t += U) {
      vw v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const vw* p = (const vw*)(s + base + (size_t)(t + u) * 64 * W); v[u] = NT ? __builtin_nontemporal_load(p) : *p; }
#pragma unroll
      for (int u = 0; u < U; ++u) { vw* p = (vw*)(d + base + (size_t)(t + u) * 64 * W); if (NT) __builtin_nontemporal_store(v[u], p); else *p = v[u]; }
   }
}

static size_t g_bytes;
template <typename L> static void time_it(const char* label, L launch)
{
   launch(); hipDeviceSynchronize();
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   hipEventRecord(e0); for (int r = 0; r < 3; ++r) launch();
   hipEventRecord(e1); hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
   printf("%-44s %8.3f ms  %7.1f GB/s\n", label, ms, 2.0 * g_bytes / ms / 1e6);
}
#define TIME(label, ...) time_it(label, [&] { __VA_ARGS__; });

int main(int argc, char** argv)
{
   const size_t gib = argc > 1 ? atol(argv[1]) : 16;
   const size_t bytes = gib << 30, n4 = bytes / 16;
   g_bytes = bytes;
   f4 *s, *d;
   hipMalloc(&s, bytes); hipMalloc(&d, bytes);
   hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
   printf("copy %zu GiB -> %zu GiB (read+write counted)\n", gib, gib);
   TIME("hipMemcpyDtoD", hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0))
   TIME("gridstride 2048 blk un1 plain", k_gs<1, false><<<dim3(2048), dim3(256)>>>(s, d, n4))
   TIME("gridstride 2048 blk un1 nt", k_gs<1, true><<<dim3(2048), dim3(256)>>>(s, d, n4))
   TIME("gridstride 2048 blk un4 plain", k_gs<4, false><<<dim3(2048), dim3(256)>>>(s, d, n4))
   TIME("gridstride 2048 blk un4 nt", k_gs<4, true><<<dim3(2048), dim3(256)>>>(s, d, n4))
   TIME("gridstride 8192 blk un4 nt", k_gs<4, true><<<dim3(8192), dim3(256)>>>(s, d, n4))
   TIME("gridstride 1024 blk un8 nt", k_gs<8, true><<<dim3(1024), dim3(256)>>>(s, d, n4))
   TIME("chunk/block un4 it16 plain", k_chunk<4, false><<<dim3(n4 / (256 * 4 * 16)), dim3(256)>>>(s, d, n4, 16))
   TIME("chunk/block un4 it16 nt", k_chunk<4, true><<<dim3(n4 / (256 * 4 * 16)), dim3(256)>>>(s, d, n4, 16))
   TIME("chunk/block un4 it1 nt (one shot)", k_chunk<4, true><<<dim3(n4 / (256 * 4)), dim3(256)>>>(s, d, n4, 1))
   TIME("chunk/block un8 it64 nt", k_chunk<8, true><<<dim3(n4 / (256 * 8 * 64)), dim3(256)>>>(s, d, n4, 64))
   const int T = argc > 2 ? atoi(argv[2]) : 4096;
   printf("row walks: T = %d rows of %zu floats\n", T, bytes / 4 / T);
   const size_t row = bytes / 4 / T;
   TIME("rows W=1 U=8 nt", k_rows<1, 8, true><<<dim3(row / 256), dim3(256)>>>((const float*)s, (float*)d, row, T))
   TIME("rows W=1 U=16 nt", k_rows<1, 16, true><<<dim3(row / 256), dim3(256)>>>((const float*)s, (float*)d, row, T))
   TIME("rows W=2 U=8 nt", k_rows<2, 8, true><<<dim3(row / 512), dim3(256)>>>((const float*)s, (float*)d, row, T))
   TIME("rows W=2 U=8 plain", k_rows<2, 8, false><<<dim3(row / 512), dim3(256)>>>((const float*)s, (float*)d, row, T))
   TIME("rows W=4 U=8 nt", k_rows<4, 8, true><<<dim3(row / 1024), dim3(256)>>>((const float*)s, (float*)d, row, T))
   TIME("rows W=4 U=4 nt", k_rows<4, 4, true><<<dim3(row / 1024), dim3(256)>>>((const float*)s, (float*)d, row, T))
   TIME("tiled W=1 U=8 nt", k_tiled<1, 8, true><<<dim3(row / 256), dim3(256)>>>((const float*)s, (float*)d, row / 64, T))
   TIME("tiled W=2 U=8 nt", k_tiled<2, 8, true><<<dim3(row / 512), dim3(256)>>>((const float*)s, (float*)d, row / 128, T))
   TIME("tiled W=4 U=8 nt", k_tiled<4, 8, true><<<dim3(row / 1024), dim3(256)>>>((const float*)s, (float*)d, row / 256, T))
   TIME("tiled W=4 U=4 plain", k_tiled<4, 4, false><<<dim3(row / 1024), dim3(256)>>>((const float*)s, (float*)d, row / 256, T))
   for (size_t pad : {(size_t)0, (size_t)64, (size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536 + 256}) {
      const size_t rowp = row - 65536 - 256;           // keep the padded walk inside the allocation
      const size_t pitch = rowp + pad;
      char lab[96]; snprintf(lab, sizeof lab, "rows-pitch W=1 U=16 nt pad=%zu floats", pad);
      g_bytes = rowp * 4 * T;
      TIME(lab, k_rows_pitch<1, 16, true><<<dim3((rowp + 255) / 256), dim3(256)>>>((const float*)s, (float*)d, rowp, pitch, T))
      snprintf(lab, sizeof lab, "rows-pitch W=2 U=8 nt pad=%zu floats", pad);
      TIME(lab, k_rows_pitch<2, 8, true><<<dim3((rowp / 2 + 255) / 256), dim3(256)>>>((const float*)s, (float*)d, rowp, pitch, T))
   }
   return 0;
}
