// Example: the reference's benchmark biquad (test/benchmark.cpp:18-33) as a bank of streams on an MI355X.
//
//   g++ -std=c++14 -I../include biquad_bank.cpp -o biquad_bank -L../zignal_amd/lib -lflowz_hip \
//       -Wl,-rpath,$PWD/../zignal_amd/lib -Wl,-rpath,/opt/rocm/lib
#include <cstdio>
#include <vector>

#include <flowz/flowz.hpp>

int main()
{
   using namespace flowz;
   const float b0 = 0.2f * 0.25f, b1 = -0.3f * 0.25f, b2 = 1.1f * 0.25f, a1 = 0.2f, a2 = -0.8f;
   auto fwd = (b0 * _1 + b1 * _1[_1] + b2 * _1[_2]);
   auto bwd = ~(_2 + a1 * _1[_1] + a2 * _1[_2]);
   auto biquad = compile(fwd |= bwd);

   // 1) the reference's call protocol: one sample per call (each call is a GPU launch)
   std::printf("impulse response:");
   for (int n = 0; n < 6; ++n) std::printf(" %a", std::get<0>(biquad(n == 0 ? 1.f : 0.f)));
   std::printf("\n");

   // 2) the block API: 4096 independent streams, 256 samples per launch, host buffers
   const int n_streams = 4096, n_samples = 256;
   auto bank = biquad.bank(n_streams);
   std::vector<float> in(size_t(n_samples) * n_streams, 0.f), out(in.size());
   for (int s = 0; s < n_streams; ++s) in[s] = 1.f + s;          // a scaled impulse per stream at t = 0
   bank.process_host(in.data(), out.data(), n_samples);           // frames are [t][stream]
   std::printf("stream 7, t = 0..3: %g %g %g %g\n", out[7], out[n_streams + 7], out[2 * n_streams + 7], out[3 * n_streams + 7]);
   bank.process_host(in.data(), out.data(), n_samples);           // next block continues from the carried state
   std::printf("recommended frame tile for device-resident data: %u streams\n", bank.recommended_tile_streams());
   return 0;
}
