// Internal to the runtime half of libflowz_hip (the files that talk to HIP / hiprtc):
//   fz_kernel_cache.cpp  hiprtc build, on-disk code-object cache, module loading, register budget
//   fz_plan.cpp          variant resolution: the library's static choice per layout and kernel body
//   fz_tune.cpp          measured plans: tune candidates, fz_program_tune, persistence per board
//   fz_launch.cpp        the launch of the fused block kernel
//   fz_bank.cpp          device-resident closure state (fz_bank) and the host-frames pipelines
//   fz_aot_kernels.hip   the AOT utility kernels (synthetic fill, copy probe, RBJ coefficients, layout adapter)
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>

#include "fz_internal.hpp"

namespace fz {

#define FZ_HIP(call)                                                                            \
   do {                                                                                         \
      hipError_t e_ = (call);                                                                   \
      if (e_ != hipSuccess)                                                                     \
         fail(FZ_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                     \
   } while (0)

#define FZ_GUARD(...)                                                           \
   try { __VA_ARGS__ }                                                                 \
   catch (const fz::Error& er) { fz::set_error(er.msg); return er.code; }       \
   catch (const std::exception& ex) { fz::set_error(ex.what()); return FZ_E_INVALID; }

void require_device();                       // FZ_E_NO_DEVICE: there is no CPU fallback in the product path
std::string cache_dir();                     // where code objects and plans.txt live ("" = nowhere)

inline uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull)
{
   for (unsigned char ch : s) {
      h ^= ch;
      h *= 1099511628211ull;
   }
   return h;
}

}  // namespace fz
