// Dev microbenchmark (GPU box): issue rate of scalar vs packed f32 mul/add on gfx950, no FMA.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/valu_bench.hip -o /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CHAINS>
__global__ void __launch_bounds__(256) k_scalar(float* out, float a, float b, int iters)
{
   float v[CHAINS];
#pragma unroll
   for (int c = 0; c < CHAINS; ++c) v[c] = threadIdx.x * 0.001f + c;
   for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
         for (int c = 0; c < CHAINS; ++c) { v[c] = v[c] * a; v[c] = v[c] + b; }
      }
   }
   float s = 0;
#pragma unroll
   for (int c = 0; c < CHAINS; ++c) s += v[c];
   out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void __launch_bounds__(256) k_packed(float* out, float a, float b, int iters)
{
   f2 v[CHAINS];
#pragma unroll
   for (int c = 0; c < CHAINS; ++c) v[c] = (f2){threadIdx.x * 0.001f + c, threadIdx.x * 0.002f - c};
   for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
         for (int c = 0; c < CHAINS; ++c) { v[c] = v[c] * a; v[c] = v[c] + b; }
      }
   }
   f2 s = {0, 0};
#pragma unroll
   for (int c = 0; c < CHAINS; ++c) s += v[c];
   out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <typename K>
static double run(K kern, int blocks, int iters, float* out)
{
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
   hipEventRecord(e0);
   hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
   hipEventRecord(e1);
   hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1);
   return ms * 1e-3;
}

int main()
{
   float* out; hipMalloc(&out, 256 * 256 * 32 * 4);
   const int iters = 4096;
   for (int wpc : {4, 8, 16, 32}) {           // waves per CU
      int blocks = 256 * wpc / 4;
      double lanes = (double)blocks * 256;
#define REPORT(name, kern, CH, width)                                                           \
      { double s = run(kern<CH>, blocks, iters, out);                                           \
        double insts = lanes / 64 * iters * 8 * CH * 2;                                         \
        printf("%-10s chains=%d waves/CU=%2d : %.3f ms  %.2f T lane-ops/s  %.2f cyc/inst/SIMD@2.4GHz\n", name, CH, wpc, \
               s * 1e3, insts * 64 * width / s / 1e12, s * 2.4e9 / (insts / 1024)); }
      REPORT("scalar", k_scalar, 1, 1) REPORT("scalar", k_scalar, 4, 1) REPORT("scalar", k_scalar, 8, 1)
      REPORT("packed", k_packed, 1, 2) REPORT("packed", k_packed, 4, 2) REPORT("packed", k_packed, 8, 2)
   }
   return 0;
}
