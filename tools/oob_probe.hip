// Dev probe (GPU box), round 4: what gfx950 does with raw-buffer accesses that (a) run partly out of the descriptor's range,
// (b) are only 4-byte aligned -- the two things a four-streams-per-lane row walk meets when the stream count is not a multiple
// of four -- and what the misaligned row walk then gets from HBM under the store cache policies the kernels can choose.
// build: hipcc --offload-arch=gfx950 -O3 tools/oob_probe.hip -o tools/_bin/oob_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const float* base, unsigned bytes)
{
   return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, (int)bytes, 0x00020000);
}

// (1) correctness: n floats in range; lane i moves the four floats [4 i, 4 i + 4) -- the last lane straddles the end
__global__ void k_oob(const float* src, float* dst, unsigned n, float* seen)
{
   const unsigned i = threadIdx.x;
   const rsrc_t ri = make_rsrc(src, n * 4u), ro = make_rsrc(dst, n * 4u);
   const u4 q = __builtin_amdgcn_raw_buffer_load_b128(ri, (int)(i * 16u), 0, 0);
   const f4 v = __builtin_bit_cast(f4, q);
   for (int j = 0; j < 4; ++j) seen[4 * i + j] = v[j];
   f4 w = v;
   for (int j = 0; j < 4; ++j) w[j] = w[j] + 1000.f;        // what is written: data + 1000 (0 + 1000 for the dwords that read as zero)
   __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, w), ro, (int)(i * 16u), 0, 0);
}

// (2) the lockstep row walk of the headline kernel, arithmetic-free: one workgroup of BLOCK lanes per CU, four streams per lane,
// one row per load, three rows in flight, a barrier per row; pitch = floats per row (any), lanes past the row end rely on the
// descriptor's range.  AUX_ST: cache policy of the stores (18 = nt|sc1 as the kernels use, 2 = nt, 0 = plain, 16 = sc1)
template <int AUX_ST>
__global__ void __launch_bounds__(1024) k_walk(const float* __restrict__ in, float* __restrict__ out, unsigned pitch, unsigned T, unsigned lanes_per_block)
{
   const unsigned lane = blockIdx.x * lanes_per_block + threadIdx.x;
   if (threadIdx.x >= lanes_per_block) return;
   const unsigned off = lane * 16u;
   if (lane * 4u >= pitch) return;                          // (whole lanes beyond the row; barriers below: a bare s_barrier counts arrived waves only... keep whole waves)
   f4 a, b, c;
   auto ld = [&](unsigned t) {
      const rsrc_t r = make_rsrc(in + (size_t)(t < T ? t : 0) * pitch, t < T ? pitch * 4u : 0u);
      return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 2));
   };
   auto st = [&](unsigned t, f4 v) {
      const rsrc_t r = make_rsrc(out + (size_t)t * pitch, pitch * 4u);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, (int)off, 0, AUX_ST);
   };
   a = ld(0);
   b = ld(1);
   for (unsigned t = 0; t + 3 <= T; t += 3) {
      c = ld(t + 2);
      st(t, a);
      __builtin_amdgcn_s_barrier();
      a = ld(t + 3);
      st(t + 1, b);
      __builtin_amdgcn_s_barrier();
      b = ld(t + 4);
      st(t + 2, c);
      __builtin_amdgcn_s_barrier();
   }
}

int main()
{
   // ---- (1) partial out-of-range b128 accesses, aligned and 4-byte-aligned bases ----------------------------------------
   for (unsigned shift = 0; shift < 2; ++shift)
      for (unsigned n : {253u, 254u, 255u, 256u}) {
         const unsigned lanes = 64, total = 4 * lanes + 64;
         float *src, *dst, *seen;
         hipMalloc(&src, (total + 4) * 4);
         hipMalloc(&dst, (total + 4) * 4);
         hipMalloc(&seen, 4 * lanes * 4);
         std::vector<float> h(total + 4);
         for (unsigned i = 0; i < total + 4; ++i) h[i] = (float)(i + 1);
         hipMemcpy(src, h.data(), (total + 4) * 4, hipMemcpyHostToDevice);
         std::vector<float> z(total + 4, -7.f);
         hipMemcpy(dst, z.data(), (total + 4) * 4, hipMemcpyHostToDevice);
         k_oob<<<1, lanes>>>(src + shift, dst + shift, n, seen);
         std::vector<float> s(4 * lanes), d(total + 4);
         hipMemcpy(s.data(), seen, 4 * lanes * 4, hipMemcpyDeviceToHost);
         hipMemcpy(d.data(), dst, (total + 4) * 4, hipMemcpyDeviceToHost);
         unsigned bad_ld = 0, bad_st = 0;
         for (unsigned i = 0; i < 4 * lanes; ++i) {
            const float want = i < n ? h[i + shift] : 0.f;
            if (s[i] != want) ++bad_ld;
         }
         for (unsigned i = 0; i < total; ++i) {
            const float want = i < n ? h[i + shift] + 1000.f : -7.f;
            if (d[i + shift] != want) ++bad_st;
         }
         printf("range check: base %s, %u floats in range: loads %s (%u wrong), stores %s (%u wrong)  [tail lane sees %g %g %g %g]\n",
                shift ? "4-byte aligned" : "16-byte aligned", n, bad_ld ? "NOT per dword" : "per dword ok", bad_ld, bad_st ? "NOT per dword" : "per dword ok", bad_st,
                s[4 * ((n - 1) / 4)], s[4 * ((n - 1) / 4) + 1], s[4 * ((n - 1) / 4) + 2], s[4 * ((n - 1) / 4) + 3]);
         hipFree(src); hipFree(dst); hipFree(seen);
      }
   // ---- (2) the row walk at an aligned and at an odd pitch ---------------------------------------------------------------
   const unsigned T = 4095;                                   // (a multiple of three rows)
   const size_t cap = (size_t)(1048576 + 1024) * 4096 * 4;
   float *in, *out;
   hipMalloc(&in, cap);
   hipMalloc(&out, cap);
   hipMemset(in, 1, cap);
   hipMemset(out, 0, cap);
   struct Case { const char* name; unsigned pitch; int aux; std::vector<float> ms; };
   std::vector<Case> cases;
   for (unsigned pitch : {1048576u, 1048577u, 1048578u, 1048580u, 1000000u})
      for (int aux : {18, 2, 0, 16}) cases.push_back({"", pitch, aux, {}});
   auto run = [&](Case& c) {
      const unsigned lanes = (c.pitch + 3) / 4, per = (lanes + 255) / 256, lpb = (per + 63) / 64 * 64;
      const unsigned blocks = (lanes + lpb - 1) / lpb;
      switch (c.aux) {
         case 18: k_walk<18><<<blocks, lpb>>>(in, out, c.pitch, T, lpb); break;
         case 2: k_walk<2><<<blocks, lpb>>>(in, out, c.pitch, T, lpb); break;
         case 0: k_walk<0><<<blocks, lpb>>>(in, out, c.pitch, T, lpb); break;
         default: k_walk<16><<<blocks, lpb>>>(in, out, c.pitch, T, lpb); break;
      }
   };
   for (auto& c : cases) run(c);
   hipDeviceSynchronize();
   hipEvent_t e0, e1;
   hipEventCreate(&e0);
   hipEventCreate(&e1);
   for (int r = 0; r < 5; ++r)
      for (auto& c : cases) {
         hipEventRecord(e0);
         run(c);
         hipEventRecord(e1);
         hipEventSynchronize(e1);
         float ms;
         hipEventElapsedTime(&ms, e0, e1);
         c.ms.push_back(ms);
      }
   printf("row walk, 4 streams per lane, one workgroup per CU in lockstep, %u rows (median of 5 interleaved rounds)\n", T);
   for (auto& c : cases) {
      std::sort(c.ms.begin(), c.ms.end());
      const float med = c.ms[c.ms.size() / 2];
      printf("pitch %8u floats  store policy %2d   %7.3f ms  %7.1f GB/s\n", c.pitch, c.aux, med, 2.0 * c.pitch * 4.0 * T / med / 1e6);
   }
   return 0;
}
