// The block API of include/flowz/flowz.hpp on device buffers: every route through the library must give
// the same bits -- time-major frames, stream-tiled frames (through the layout adapter), stream-major
// buffers straight through the kernel, block windows with per-block coefficients, a tuned plan, and host
// rows per stream.  Device memory comes from the HIP runtime (declared here: the header needs no HIP).
#include <cstdio>
#include <cstring>
#include <vector>

#include <flowz/flowz.hpp>

extern "C" {
int hipMalloc(void** p, size_t n);
int hipFree(void* p);
int hipMemcpy(void* dst, const void* src, size_t n, int kind);
int hipDeviceSynchronize(void);
}
enum { H2D = 1, D2H = 2 };

static int failures = 0;
#define CHECK(cond)                                                                  \
   do {                                                                              \
      if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
   } while (0)

struct dbuf {
   float* p = nullptr;
   size_t n;
   explicit dbuf(size_t floats) : n(floats) { if (hipMalloc(reinterpret_cast<void**>(&p), floats * 4)) p = nullptr; }
   ~dbuf() { hipFree(p); }
   void up(const std::vector<float>& h) { hipMemcpy(p, h.data(), h.size() * 4, H2D); }
   std::vector<float> down() const { std::vector<float> h(n); hipDeviceSynchronize(); hipMemcpy(h.data(), p, n * 4, D2H); return h; }
};

static bool same(const std::vector<float>& a, const std::vector<float>& b) { return a.size() == b.size() && !std::memcmp(a.data(), b.data(), a.size() * 4); }

int main()
{
   using namespace flowz;
   const float b0 = 0.05f, b1 = -0.075f, b2 = 0.275f, a1 = 0.2f, a2 = -0.8f;
   auto biquad = compile((b0 * _1 + b1 * _1[_1] + b2 * _1[_2]) |= ~(_2 + a1 * _1[_1] + a2 * _1[_2]));
   const uint32_t ns = 2048, T = 96, tile = 512;
   std::vector<float> x(size_t(T) * ns), xs(x.size());                 // frames [t][s] and rows [s][t]
   for (uint32_t t = 0; t < T; ++t)
      for (uint32_t s = 0; s < ns; ++s) {
         const float v = float(int((s * 2654435761u + t * 40503u) >> 20 & 1023) - 512) / 512.f;
         x[size_t(t) * ns + s] = v;
         xs[size_t(s) * T + t] = v;
      }
   // reference route: host frames through process_host
   std::vector<float> want(x.size());
   {
      auto bank = biquad.bank(ns);
      bank.process_host(x.data(), want.data(), T);
   }
   auto rows_of = [&](const std::vector<float>& frames) {              // [t][s] -> [s][t]
      std::vector<float> r(frames.size());
      for (uint32_t t = 0; t < T; ++t)
         for (uint32_t s = 0; s < ns; ++s) r[size_t(s) * T + t] = frames[size_t(t) * ns + s];
      return r;
   };
   dbuf din(x.size()), dout(x.size()), drows(x.size()), drows_out(x.size()), dtiled(x.size()), dtiled_out(x.size());
   CHECK(din.p && dout.p && drows.p && drows_out.p && dtiled.p && dtiled_out.p);
   din.up(x);
   drows.up(xs);
   {  // device frames, time-major, one block; then two windows with state carried
      auto bank = biquad.bank(ns);
      bank.process(din.p, dout.p, T);
      CHECK(same(dout.down(), want));
      bank.reset();
      bank.process_blocks(din.p, dout.p, T, 40);                         // blocks of 40, 40, 16 samples
      CHECK(same(dout.down(), want));
   }
   {  // stream-major rows -> tiled frames (adapter) -> kernel -> back
      auto bank = biquad.bank(ns);
      frames_from_stream_major(drows.p, dtiled.p, ns, T, 1, tile);
      bank.process_tiled(dtiled.p, dtiled_out.p, T, tile);
      frames_to_stream_major(dtiled_out.p, drows_out.p, ns, T, 1, tile);
      CHECK(same(drows_out.down(), rows_of(want)));
      bank.reset();                                                       // a measured plan computes the same bits
      const fz_variant plan = bank.tune(dtiled.p, dtiled_out.p, T, tile);
      (void)plan;
      bank.reset();
      bank.process_tiled(dtiled.p, dtiled_out.p, T, tile);
      frames_to_stream_major(dtiled_out.p, drows_out.p, ns, T, 1, tile);
      CHECK(same(drows_out.down(), rows_of(want)));
   }
   {  // stream-major rows straight through the kernel, in two windows
      auto bank = biquad.bank(ns);
      bank.process_stream_major(drows.p, drows_out.p, T, 0, 36);
      bank.process_stream_major(drows.p, drows_out.p, T, 36, T - 36);
      CHECK(same(drows_out.down(), rows_of(want)));
   }
   {  // host rows per stream
      auto bank = biquad.bank(ns);
      std::vector<float> got(xs.size());
      bank.process_host_stream_major(xs.data(), got.data(), T);
      CHECK(same(got, rows_of(want)));
   }
   std::printf(failures ? "%d FAILURES\n" : "all block API checks passed\n", failures);
   return failures != 0;
}
