// AOT utility kernels of libflowz_hip (gfx950): synthetic input fill, the copy-bandwidth yardstick, the RBJ low-pass
// coefficient generator and the stream-major <-> frames layout adapter, with their C entry points.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "fz_runtime.hpp"

namespace fz {

// ---- AOT utility kernels ---------------------------------------------------------------------------------------
typedef float fzr_f4 __attribute__((ext_vector_type(4)));

// sin and cos of a FLOAT argument, each rounded to float once: argument reduction by multiples of pi/2 (a two-part pi/2 whose head has
// 33 bits: k * head is exact), then the Taylor polynomials of sin / cos on [-pi/4, pi/4] by Horner's rule in r^2 -- IEEE double additions
// and multiplications in a fixed order, no FMA (this file is built with -ffp-contract=off), no libm: the tests' C checker
// spells the same operations and gets the same bits.  The double result is within 2^-60 of the true value, so the float is the correctly
// rounded one except for arguments within that distance of a rounding boundary; |x| >= 2^20 (k * head no longer exact): NaN on both sides.
__device__ __forceinline__ void fz_sincos_f32(float xf, float* sn, float* cs)
{
   const double x = (double)xf;
   if (!(fabs(x) < 0x1p20)) {
      *sn = *cs = __builtin_nanf("");
      return;
   }
   const double t = x * 0x1.45f306dc9c883p-1;                               // x * 2/pi
   const int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));                         // nearest integer (conversion truncates)
   const double kd = (double)k;
   double r = x - kd * 0x1.921fb54400000p+0;                                 // pi/2, first 33 bits: the product is exact
   r = r - kd * 0x1.0b4611a600000p-34;                                       // ... next 33 bits
   r = r - kd * 0x1.3198a2e037073p-69;                                       // ... the rest
   const double z = r * r;
   double ps = 0x1.952c77030ad4ap-49;                                        // 1/17!
   ps = -0x1.ae7f3e733b81fp-41 + z * ps;
   ps = 0x1.6124613a86d09p-33 + z * ps;
   ps = -0x1.ae64567f544e4p-26 + z * ps;
   ps = 0x1.71de3a556c734p-19 + z * ps;
   ps = -0x1.a01a01a01a01ap-13 + z * ps;
   ps = 0x1.1111111111111p-7 + z * ps;
   ps = -0x1.5555555555555p-3 + z * ps;
   const double s = r + r * (z * ps);
   double pc = -0x1.6827863b97d97p-53;                                       // -1/18!
   pc = 0x1.ae7f3e733b81fp-45 + z * pc;
   pc = -0x1.93974a8c07c9dp-37 + z * pc;
   pc = 0x1.1eed8eff8d898p-29 + z * pc;
   pc = -0x1.27e4fb7789f5cp-22 + z * pc;
   pc = 0x1.a01a01a01a01ap-16 + z * pc;
   pc = -0x1.6c16c16c16c17p-10 + z * pc;
   pc = 0x1.5555555555555p-5 + z * pc;
   pc = -0x1.0000000000000p-1 + z * pc;
   const double c = 1.0 + z * pc;
   const int q = k & 3;
   *sn = (float)(q == 0 ? s : q == 1 ? c : q == 2 ? -s : -c);
   *cs = (float)(q == 0 ? c : q == 1 ? -s : q == 2 ? -c : s);
}

// reactive_equations/reactive_filter_coeff.cpp:38-58, one stream per thread
__global__ void __launch_bounds__(256) fz_rbj_lowpass_kernel(const float* __restrict__ freq, const float* __restrict__ q, float sr,
                                                             unsigned long long n, float* raw6, float* df1)
{
   const unsigned long long s = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
   if (s >= n) return;
   const float two_pi = (float)(8. * 0.78539816339744830962);     // const float two_pi = 8. * std::atan(1.)
   const float w0 = two_pi * freq[s] / sr;
   float sinw0, cosw0;                                              // std::sin / std::cos of a float (see fz_sincos_f32)
   fz_sincos_f32(w0, &sinw0, &cosw0);
   const float alpha = (float)(sinw0 / (2. * q[s]));
   const float b0 = (float)((1. - cosw0) / 2.);
   const float b1 = (float)(1. - cosw0);
   const float b2 = (float)((1. - cosw0) / 2.);
   const float a0 = (float)(1. + alpha);
   const float a1 = (float)(-2. * cosw0);
   const float a2 = (float)(1. - alpha);
   if (raw6) {
      raw6[0 * n + s] = a0; raw6[1 * n + s] = a1; raw6[2 * n + s] = a2;
      raw6[3 * n + s] = b0; raw6[4 * n + s] = b1; raw6[5 * n + s] = b2;
   }
   if (df1) {
      df1[0 * n + s] = b0 / a0; df1[1 * n + s] = b1 / a0; df1[2 * n + s] = b2 / a0;
      df1[3 * n + s] = -a1 / a0; df1[4 * n + s] = -a2 / a0;
   }
}

__device__ __forceinline__ unsigned fmix32(unsigned h)
{
   h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
   return h;
}

// dst[t][s][w] for one row t per blockIdx.y; a thread produces 4 consecutive floats of the row
__global__ void __launch_bounds__(256) fz_synth_fill_kernel(float* dst, unsigned long long row_floats, unsigned n_wires,
                                                            unsigned seed, unsigned long long stream0,
                                                            unsigned long long t0, unsigned n_rows,
                                                            unsigned long long tile_floats)
{
   // row_floats = n_streams * n_wires of the logical time-major row; tile_floats = floats of one
   // tile's row segment (== row_floats when untiled).  Logical element i of row t is stored at
   // (i / tile_floats) * n_rows * tile_floats + t * tile_floats + i % tile_floats.
   const unsigned long long i0 = ((unsigned long long)blockIdx.x * 256u + threadIdx.x) * 4ull;
   if (i0 >= row_floats) return;
   const unsigned long long tl = i0 / tile_floats, within = i0 - tl * tile_floats;
   for (unsigned t = blockIdx.y; t < n_rows; t += gridDim.y) {
      const unsigned tt = (unsigned)((t0 + t) * 0x85EBCA6Bull);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
         const unsigned long long sid = stream0 * n_wires + i0 + j;     // (stream0+s)*n_wires + w
         const unsigned h = fmix32(fmix32(seed ^ (unsigned)(sid * 0x9E3779B9ull) ^ tt));
         v[j] = (float)(int)(h >> 8) * 0x1p-23f - 1.0f;
      }
      float* row = dst + (size_t)tl * n_rows * tile_floats + (size_t)t * tile_floats;
      if (i0 + 4 <= row_floats && (tile_floats & 3ull) == 0) {   // segments stay 16-byte aligned
         fzr_f4 q = {v[0], v[1], v[2], v[3]};
         __builtin_nontemporal_store(q, reinterpret_cast<fzr_f4*>(row + within));
      } else {
         for (int j = 0; j < 4 && i0 + j < row_floats; ++j) {
            const unsigned long long i = i0 + j, tj = i / tile_floats;
            dst[(size_t)tj * n_rows * tile_floats + (size_t)t * tile_floats + (i - tj * tile_floats)] = v[j];
         }
      }
   }
}

// one-shot float4 copy, four independent nt loads in flight per lane before the stores: the fastest
// plain copy of profiles/r01/hbm_copy_patterns_microbench.txt (5.8-6.0 TB/s; a 2048-block grid-stride
// loop and hipMemcpyDtoD stay at 4.8-4.9)
__global__ void __launch_bounds__(256) fz_copy_kernel(const fzr_f4* __restrict__ src, fzr_f4* __restrict__ dst,
                                                      unsigned long long n4)
{
   const unsigned long long base = (unsigned long long)blockIdx.x * 1024u + threadIdx.x;
   fzr_f4 v[4];
#pragma unroll
   for (int k = 0; k < 4; ++k)
      if (base + 256u * k < n4) v[k] = __builtin_nontemporal_load(src + base + 256u * k);
#pragma unroll
   for (int k = 0; k < 4; ++k)
      if (base + 256u * k < n4) __builtin_nontemporal_store(v[k], dst + base + 256u * k);
}

// Stream-major <-> frame layout adapter.  Callers of the reference hold one contiguous sample buffer
// per closure ([stream][t][wire], the loop of test/benchmark.cpp:137-147); the block kernel wants
// frames with the stream index fastest ([t][stream][wire], optionally tiled).  One workgroup moves a
// 64-stream x CT-column patch (CT = whole frames, <= 64 floats) through LDS so that both the reads and
// the writes are contiguous runs: rows of the stream-major side, (stream, wire) runs of the frame side.
template <bool TO_STREAM_MAJOR>
__global__ void __launch_bounds__(256) fz_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           unsigned long long n_streams, unsigned n_samples, unsigned W,
                                                           unsigned tile_streams, unsigned nt /* frames per patch */,
                                                           unsigned gx, unsigned gy)
{
   __shared__ float patch[64][65];
   // workgroups that run at the same time cover a 16 x 16 block of patches, so that each side sees
   // 4 KiB runs (16 patches x 256 B) instead of isolated 256 B pieces
   constexpr unsigned SX = 16, SY = 16;
   const unsigned sbx = (gx + SX - 1) / SX;
   const unsigned long long b = blockIdx.x;
   const unsigned long long sup = b / (SX * SY);
   const unsigned within = (unsigned)(b % (SX * SY));
   const unsigned px = (unsigned)(sup % sbx) * SX + within % SX, py = (unsigned)(sup / sbx) * SY + within / SX;
   if (px >= gx || py >= gy) return;
   const unsigned long long s0 = (unsigned long long)px * 64u;
   const unsigned t0 = py * nt;
   const unsigned tid = threadIdx.x;
   const unsigned long long TW = (unsigned long long)n_samples * W;
   const unsigned ns_here = (unsigned)(n_streams - s0 < 64u ? n_streams - s0 : 64u);
   const unsigned nt_here = n_samples - t0 < nt ? n_samples - t0 : nt;
   // frame side: element (t, s, w) at fbase + (t0 + t) * row_streams * W + s * W + w
   const unsigned long long tile = tile_streams ? s0 / tile_streams : 0u;
   const unsigned long long row_streams = tile_streams ? tile_streams : n_streams;
   const unsigned long long s_in_tile = tile_streams ? s0 % tile_streams : s0;
   const unsigned long long fbase = tile * (unsigned long long)n_samples * row_streams * W + s_in_tile * W;
   const unsigned run = 64u * W;                                     // floats of one frame row of the patch
   // stream-major side: element (s, c) at (s0 + s) * TW + t0 * W + c, c < nt * W
   // full patches of 16-byte-aligned layouts move as float4 (all 4 loads of a thread in flight at once);
   // edge patches and odd wire counts take the scalar path
   const bool vec = nt * W == 64u && ns_here == 64u && nt_here == nt && (TW & 3u) == 0 && ((row_streams * W) & 3u) == 0;
   if (vec) {
      const unsigned q = tid & 15u, r0 = tid >> 4;                  // float4 column, first row
      if (!TO_STREAM_MAJOR) {
         fzr_f4 v[4];
#pragma unroll
         for (int k = 0; k < 4; ++k)
            v[k] = __builtin_nontemporal_load(reinterpret_cast<const fzr_f4*>(src + (s0 + r0 + 16u * k) * TW + (unsigned long long)t0 * W) + q);
#pragma unroll
         for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) patch[r0 + 16u * k][q * 4u + j] = v[k][j];
         __syncthreads();
         // frame rows: nt rows of `run` floats; float4 index over the whole patch output
         for (unsigned e = tid; e < nt * run / 4u; e += 256u) {
            const unsigned t = e / (run / 4u), r = (e - t * (run / 4u)) * 4u;
            fzr_f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
               const unsigned rr = r + j, sl = rr / W, w = rr - sl * W;
               o[j] = patch[sl][t * W + w];
            }
            __builtin_nontemporal_store(o, reinterpret_cast<fzr_f4*>(dst + fbase + (unsigned long long)(t0 + t) * row_streams * W + r));
         }
      } else {
         for (unsigned e = tid; e < nt * run / 4u; e += 256u) {
            const unsigned t = e / (run / 4u), r = (e - t * (run / 4u)) * 4u;
            const fzr_f4 o = __builtin_nontemporal_load(reinterpret_cast<const fzr_f4*>(src + fbase + (unsigned long long)(t0 + t) * row_streams * W + r));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
               const unsigned rr = r + j, sl = rr / W, w = rr - sl * W;
               patch[sl][t * W + w] = o[j];
            }
         }
         __syncthreads();
#pragma unroll
         for (int k = 0; k < 4; ++k) {
            fzr_f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = patch[r0 + 16u * k][q * 4u + j];
            __builtin_nontemporal_store(o, reinterpret_cast<fzr_f4*>(dst + (s0 + r0 + 16u * k) * TW + (unsigned long long)t0 * W) + q);
         }
      }
      return;
   }
   if (!TO_STREAM_MAJOR) {
      for (unsigned e = tid; e < 64u * 64u; e += 256u) {
         const unsigned sl = e >> 6, c = e & 63u;
         if (sl < ns_here && c < nt_here * W) patch[sl][c] = __builtin_nontemporal_load(src + (s0 + sl) * TW + (unsigned long long)t0 * W + c);
      }
      __syncthreads();
      for (unsigned e = tid; e < nt * run; e += 256u) {
         const unsigned t = e / run, r = e - t * run, sl = r / W, w = r - sl * W;
         if (t < nt_here && sl < ns_here)
            __builtin_nontemporal_store(patch[sl][t * W + w], dst + fbase + (unsigned long long)(t0 + t) * row_streams * W + r);
      }
   } else {
      for (unsigned e = tid; e < nt * run; e += 256u) {
         const unsigned t = e / run, r = e - t * run, sl = r / W, w = r - sl * W;
         if (t < nt_here && sl < ns_here)
            patch[sl][t * W + w] = __builtin_nontemporal_load(src + fbase + (unsigned long long)(t0 + t) * row_streams * W + r);
      }
      __syncthreads();
      for (unsigned e = tid; e < 64u * 64u; e += 256u) {
         const unsigned sl = e >> 6, c = e & 63u;
         if (sl < ns_here && c < nt_here * W) __builtin_nontemporal_store(patch[sl][c], dst + (s0 + sl) * TW + (unsigned long long)t0 * W + c);
      }
   }
}

}  // namespace fz

using namespace fz;

extern "C" {

int fz_device_count(void) { return fz::device_count(); }

int fz_synth_fill(float* dst, uint64_t n_streams, uint32_t n_samples, uint32_t n_wires, uint32_t seed,
                  uint64_t stream0, uint64_t t0, uint32_t tile_streams, void* hip_stream)
{
   FZ_GUARD(
      if (!dst || !n_streams || !n_samples || !n_wires) fail(FZ_E_INVALID, "fz_synth_fill: bad arguments");
      require_device();
      const unsigned long long row = n_streams * n_wires;
      if (tile_streams && n_streams % tile_streams) fail(FZ_E_INVALID, "n_streams must be a multiple of tile_streams");
      const unsigned long long tile_floats = (tile_streams && tile_streams < n_streams) ? (unsigned long long)tile_streams * n_wires : row;
      dim3 grid((unsigned)((row + 1023) / 1024), std::min<uint32_t>(n_samples, 64u));
      hipLaunchKernelGGL(fz_synth_fill_kernel, grid, dim3(256), 0, (hipStream_t)hip_stream, dst, row, n_wires, seed,
                         (unsigned long long)stream0, (unsigned long long)t0, n_samples, tile_floats);
      FZ_HIP(hipGetLastError());
      return FZ_OK;)
}

int fz_rbj_lowpass(const float* freq, const float* q, float sample_rate, uint64_t n_streams, float* raw6, float* df1,
                   void* hip_stream)
{
   FZ_GUARD(
      if (!freq || !q || !n_streams || (!raw6 && !df1)) fail(FZ_E_INVALID, "fz_rbj_lowpass: bad arguments");
      require_device();
      hipLaunchKernelGGL(fz_rbj_lowpass_kernel, dim3((unsigned)((n_streams + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                         freq, q, sample_rate, (unsigned long long)n_streams, raw6, df1);
      FZ_HIP(hipGetLastError());
      return FZ_OK;)
}

int fz_copy_probe(const float* src, float* dst, uint64_t n_floats, void* hip_stream)
{
   FZ_GUARD(
      if (!src || !dst || (n_floats & 3)) fail(FZ_E_INVALID, "fz_copy_probe: need non-null pointers and n_floats % 4 == 0");
      require_device();
      const unsigned long long n4 = n_floats / 4;
      if (n4 > 1024ull * 0x7FFFFFFFull) fail(FZ_E_INVALID, "fz_copy_probe: buffer too large");
      hipLaunchKernelGGL(fz_copy_kernel, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, (hipStream_t)hip_stream,
                         (const fzr_f4*)src, (fzr_f4*)dst, (unsigned long long)(n_floats / 4));
      FZ_HIP(hipGetLastError());
      return FZ_OK;)
}

int fz_transpose_frames(const float* src, float* dst, uint64_t n_streams, uint32_t n_samples, uint32_t n_wires,
                        uint32_t tile_streams, int to_stream_major, void* hip_stream)
{
   FZ_GUARD(
      if (!src || !dst || !n_streams || !n_samples || !n_wires) fail(FZ_E_INVALID, "fz_transpose_frames: bad arguments");
      if (n_wires > 64) fail(FZ_E_UNSUPPORTED, "fz_transpose_frames: more than 64 wires per frame");
      if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) fail(FZ_E_INVALID, "device pointers must be 16-byte aligned");
      if (tile_streams >= n_streams) tile_streams = 0;
      if (tile_streams && (tile_streams % 64 || n_streams % tile_streams))
         fail(FZ_E_INVALID, "tile_streams must be a multiple of 64 and divide n_streams");
      require_device();
      const unsigned nt = 64u / n_wires;
      const uint64_t gx = (n_streams + 63) / 64, gy = ((uint64_t)n_samples + nt - 1) / nt;
      const uint64_t blocks = ((gx + 15) / 16) * ((gy + 15) / 16) * 256;
      if (gx > 0xFFFFFFFFull || blocks > 0x7FFFFFFFull) fail(FZ_E_UNSUPPORTED, "fz_transpose_frames: too many patches for one launch: split the block");
      if (to_stream_major)
         hipLaunchKernelGGL(fz_transpose_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, src, dst,
                            (unsigned long long)n_streams, n_samples, n_wires, tile_streams, nt, (unsigned)gx, (unsigned)gy);
      else
         hipLaunchKernelGGL(fz_transpose_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, src, dst,
                            (unsigned long long)n_streams, n_samples, n_wires, tile_streams, nt, (unsigned)gx, (unsigned)gy);
      FZ_HIP(hipGetLastError());
      return FZ_OK;)
}

}  // extern "C"
