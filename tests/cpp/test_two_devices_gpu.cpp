// One process, several devices (SURVEY 8e hardened on a single host process): the streams of a batch are cut into contiguous
// shards, one per device, each shard driven by a host thread of its own that selects its device (hipSetDevice), owns its bank
// and its device buffers, and shares nothing with the others but the compiled program -- whose kernels are loaded per device,
// whose measured plans are keyed by device and whose arrival counters are per device.  Every shard must equal the columns of
// the single-device result bit for bit.  With one visible device the same threads share device 0 (two banks, two HIP streams):
// the thread-safety half of the test still runs; the device half is reported as skipped.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <flowz/flowz.hpp>
#include <flowz/shard.hpp>      // shard_range + the statistics reduction over RCCL (brings the HIP runtime and RCCL headers)

static const hipMemcpyKind H2D = hipMemcpyHostToDevice, D2H = hipMemcpyDeviceToHost;

static int failures = 0;
#define CHECK(cond)                                                                  \
   do {                                                                              \
      if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
   } while (0)

int main()
{
   using namespace flowz;
   int n_dev = 0;
   if (hipGetDeviceCount(&n_dev) != 0 || n_dev < 1) {
      std::printf("no device\n");
      return 1;
   }
   const float b0 = 0.05f, b1 = -0.075f, b2 = 0.275f, a1 = 0.2f, a2 = -0.8f;
   auto stage = [&] { return (b0 * _1 + b1 * _1[_1] + b2 * _1[_2]) |= ~(_2 + a1 * _1[_1] + a2 * _1[_2]); };
   auto cascade = compile(stage() |= stage() |= stage());
   const uint32_t ns = 6144, T = 160, shards = 2;
   std::vector<float> x(size_t(T) * ns), want(x.size());
   for (uint32_t t = 0; t < T; ++t)
      for (uint32_t s = 0; s < ns; ++s) x[size_t(t) * ns + s] = float(int((s * 2654435761u + t * 40503u) >> 20 & 1023) - 512) / 512.f;
   {
      hipSetDevice(0);
      auto bank = cascade.bank(ns);
      bank.process_host(x.data(), want.data(), T);                        // the single-device result
   }
   std::vector<std::vector<float>> got(shards);
   std::vector<int> ok(shards, 0);
   auto work = [&](uint32_t k) {
      const int dev = n_dev >= 2 ? int(k) : 0;
      if (hipSetDevice(dev) != 0) return;
      const auto range = shard_range(ns, k, shards);                        // contiguous stream range
      const uint32_t begin = (uint32_t)range.first, n = (uint32_t)(range.second - range.first);
      std::vector<float> xin(size_t(T) * n);
      for (uint32_t t = 0; t < T; ++t) std::memcpy(&xin[size_t(t) * n], &x[size_t(t) * ns + begin], n * sizeof(float));
      float *din = nullptr, *dout = nullptr;
      hipStream_t stream = nullptr;
      if (hipMalloc(reinterpret_cast<void**>(&din), xin.size() * 4) || hipMalloc(reinterpret_cast<void**>(&dout), xin.size() * 4) || hipStreamCreate(&stream)) return;
      hipMemcpy(din, xin.data(), xin.size() * 4, H2D);
      try {
         auto bank = cascade.bank(n);                                      // state on THIS thread's device
         bank.process(din, dout, 64, stream);                              // three blocks chained on the thread's own stream
         bank.process(din + size_t(64) * n, dout + size_t(64) * n, 64, stream);
         bank.process(din + size_t(128) * n, dout + size_t(128) * n, T - 128, stream);
         hipStreamSynchronize(stream);
         got[k].resize(xin.size());
         hipMemcpy(got[k].data(), dout, xin.size() * 4, D2H);
         ok[k] = 1;
      } catch (const std::exception& e) {
         std::printf("shard %u on device %d: %s\n", k, dev, e.what());
      }
      hipStreamDestroy(stream);
      hipFree(din);
      hipFree(dout);
   };
   std::vector<std::thread> ths;
   for (uint32_t k = 0; k < shards; ++k) ths.emplace_back(work, k);
   for (auto& t : ths) t.join();
   for (uint32_t k = 0; k < shards; ++k) {
      CHECK(ok[k]);
      if (!ok[k]) continue;
      const uint32_t begin = k * ns / shards, n = (k + 1) * ns / shards - begin;
      bool same = true;
      for (uint32_t t = 0; t < T && same; ++t) same = !std::memcmp(&got[k][size_t(t) * n], &want[size_t(t) * ns + begin], n * sizeof(float));
      CHECK(same);
   }
   {  // the ONE collective of the sharded path, from C++: max seconds / sum of samples / sum of an integer checksum over RCCL (include/flowz/shard.hpp).
      // One communicator per device in use; with one visible device the two shards' statistics are summed on the host first (world size 1).
      auto checksum_of = [&](const float* row, uint32_t n) {
         unsigned long long c = 0;
         for (uint32_t i = 0; i < n; ++i) { uint32_t u; std::memcpy(&u, row + i, 4); c += u; }
         return c;
      };
      const unsigned long long want_sum = checksum_of(&want[size_t(T - 1) * ns], ns);
      std::vector<run_stats> st(n_dev >= 2 ? shards : 1);
      for (uint32_t k = 0; k < shards; ++k) {
         if (!ok[k]) continue;
         const auto r = shard_range(ns, k, shards);
         const uint32_t n = (uint32_t)(r.second - r.first);
         run_stats& q = st[n_dev >= 2 ? k : 0];
         q.seconds = std::max(q.seconds, 0.001 * (k + 1));
         q.samples += double(n) * T;
         q.checksum += checksum_of(&got[k][size_t(T - 1) * n], n);
      }
      try {
         std::vector<int> devs;
         for (size_t i = 0; i < st.size(); ++i) devs.push_back((int)i);
         stats_reducer red(devs);
         const run_stats all = red.reduce(st);
         CHECK(all.samples == double(ns) * T);
         CHECK(all.checksum == want_sum);
         CHECK(all.seconds == 0.001 * shards);
         std::printf("statistics reduced over RCCL, world size %u: %.0f samples, checksum %llu\n", red.world(), all.samples, all.checksum);
      } catch (const std::exception& e) {
         std::printf("FAILED: %s\n", e.what());
         ++failures;
      }
   }
   if (n_dev < 2) std::printf("one visible device: both shards ran on device 0 (two host threads, two banks, two streams); the two-device run is skipped\n");
   else std::printf("%u shards on %d devices, one host thread each\n", shards, n_dev);
   std::printf(failures ? "%d FAILURES\n" : "all multi-device checks passed\n", failures);
   return failures != 0;
}
