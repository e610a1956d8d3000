// Stream sharding over the GPUs of one node for C++ hosts -- the native counterpart of zignal_amd/dist.py.
//
// Streams are private closures (flowz.hpp:1181-1230): the data path has NO collective.  Device g of G owns the contiguous range
// shard_range(total, g, G) of global stream ids, with a bank and buffers of its own (tests/cpp/test_two_devices_gpu.cpp drives them
// from one host thread per device).  The ONE collective is the reduction of the run statistics at the end -- max seconds, sum of samples,
// sum of an integer checksum -- as RCCL all-reduces over xGMI on a few device words.  This header is the only place of the C++ front end
// that needs RCCL and the HIP runtime headers; include it only where shards are reduced:
//     g++ -std=c++14 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ... -lrccl -lamdhip64
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace flowz {

// contiguous [begin, end) of global stream ids for shard `rank` of `world`; sizes differ by at most one
inline std::pair<uint64_t, uint64_t> shard_range(uint64_t total, unsigned rank, unsigned world)
{
   const uint64_t q = total / world, r = total % world;
   const uint64_t begin = rank * q + (rank < r ? rank : r);
   return {begin, begin + q + (rank < r ? 1 : 0)};
}

struct run_stats {
   double seconds = 0.0;               // reduced with MAX: the slowest shard bounds the job
   double samples = 0.0;               // reduced with SUM
   unsigned long long checksum = 0;    // reduced with SUM in 64 bits: exact, independent of the sharding
};

// One process driving several devices: a communicator per device (ncclCommInitAll), the per-device statistics in device memory, three
// grouped all-reduces.  reduce(per_device) returns what every device then holds.
class stats_reducer {
public:
   explicit stats_reducer(const std::vector<int>& devices) : devs_(devices), comms_(devices.size()), buf_(devices.size(), nullptr)
   {
      if (devs_.empty()) throw std::invalid_argument("stats_reducer: no devices");
      rccl(ncclCommInitAll(comms_.data(), (int)devs_.size(), devs_.data()), "ncclCommInitAll");
      for (size_t i = 0; i < devs_.size(); ++i) {
         hip(hipSetDevice(devs_[i]), "hipSetDevice");
         hip(hipMalloc(&buf_[i], 3 * sizeof(double)), "hipMalloc");
      }
   }
   ~stats_reducer()
   {
      for (size_t i = 0; i < devs_.size(); ++i) {
         if (buf_[i] && hipSetDevice(devs_[i]) == hipSuccess) (void)hipFree(buf_[i]);
         if (comms_[i]) (void)ncclCommDestroy(comms_[i]);
      }
   }
   stats_reducer(const stats_reducer&) = delete;
   stats_reducer& operator=(const stats_reducer&) = delete;
   unsigned world() const { return (unsigned)devs_.size(); }

   run_stats reduce(const std::vector<run_stats>& per_device)
   {
      if (per_device.size() != devs_.size()) throw std::invalid_argument("stats_reducer: one run_stats per device");
      for (size_t i = 0; i < devs_.size(); ++i) {
         hip(hipSetDevice(devs_[i]), "hipSetDevice");
         static_assert(sizeof(unsigned long long) == sizeof(double), "the three statistics travel as 64-bit words");
         unsigned char words[3 * sizeof(double)];
         std::memcpy(words, &per_device[i].seconds, 8);
         std::memcpy(words + 8, &per_device[i].samples, 8);
         std::memcpy(words + 16, &per_device[i].checksum, 8);
         hip(hipMemcpy(buf_[i], words, sizeof words, hipMemcpyHostToDevice), "hipMemcpy");
      }
      rccl(ncclGroupStart(), "ncclGroupStart");
      for (size_t i = 0; i < devs_.size(); ++i) {
         char* b = static_cast<char*>(buf_[i]);
         rccl(ncclAllReduce(b, b, 1, ncclDouble, ncclMax, comms_[i], nullptr), "ncclAllReduce(max seconds)");
         rccl(ncclAllReduce(b + 8, b + 8, 1, ncclDouble, ncclSum, comms_[i], nullptr), "ncclAllReduce(sum samples)");
         rccl(ncclAllReduce(b + 16, b + 16, 1, ncclUint64, ncclSum, comms_[i], nullptr), "ncclAllReduce(sum checksum)");
      }
      rccl(ncclGroupEnd(), "ncclGroupEnd");
      run_stats out;
      for (size_t i = 0; i < devs_.size(); ++i) {              // every device holds the same three words afterwards; device 0's are returned
         hip(hipSetDevice(devs_[i]), "hipSetDevice");
         hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
         if (i == 0) {
            unsigned char words[3 * sizeof(double)];
            hip(hipMemcpy(words, buf_[0], sizeof words, hipMemcpyDeviceToHost), "hipMemcpy");
            std::memcpy(&out.seconds, words, 8);
            std::memcpy(&out.samples, words + 8, 8);
            std::memcpy(&out.checksum, words + 16, 8);
         }
      }
      return out;
   }

private:
   static void hip(hipError_t e, const char* what)
   {
      if (e != hipSuccess) throw std::runtime_error(std::string("flowz::stats_reducer: ") + what + ": " + hipGetErrorString(e));
   }
   static void rccl(ncclResult_t r, const char* what)
   {
      if (r != ncclSuccess) throw std::runtime_error(std::string("flowz::stats_reducer: ") + what + ": " + ncclGetErrorString(r));
   }
   std::vector<int> devs_;
   std::vector<ncclComm_t> comms_;
   std::vector<void*> buf_;
};

}  // namespace flowz
