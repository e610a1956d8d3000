// fz_rtc_worker -- the ROCm installation's hiprtc in a process of its own.
//
// libflowz_hip.so builds its fused kernels with hiprtc at run time.  A host process that has loaded ANOTHER libhiprtc.so.7 before
// the library (a PyTorch wheel bundles the hiprtc + comgr of the ROCm release it was built with) binds the library to that copy, and
// the code -- registers, spills, which kernel variant fits -- would then depend on who imported what first.  So when the library
// finds itself bound to a foreign hiprtc it hands every build to this program instead: a fresh process whose only hiprtc is the
// installation's (DT_RPATH, searched before LD_LIBRARY_PATH; the parent also clears LD_LIBRARY_PATH / LD_PRELOAD).
// Code objects are byte-identical to what a torch-free process builds in-process.
//
//   fz_rtc_worker --identify                      prints the hiprtc it is bound to, exits 0
//   fz_rtc_worker <request file> <output file>
//     request:  "FZRTC1 <n>\n" then n sections "<kind> <name> <bytes>\n<bytes of data>\n"
//               kind = source (name: file name) | header (name: include name) | option (name: "-", data: the option)
//     output:   the code object;  <output file>.log: the compiler's log when the build fails
//     stdout:   "hiprtc <real path of the libhiprtc in use>"
//     exit:     0 built, 2 bad request, 3 compile error, 4 hiprtc failure
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <limits.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

struct Section {
   std::string kind, name, data;
};

static bool read_request(const char* path, std::vector<Section>& out)
{
   std::ifstream f(path, std::ios::binary);
   if (!f) return false;
   std::string magic;
   size_t n = 0;
   if (!(f >> magic >> n) || magic != "FZRTC1" || n > 4096) return false;
   f.get();
   for (size_t i = 0; i < n; ++i) {
      Section s;
      size_t bytes = 0;
      if (!(f >> s.kind >> s.name >> bytes) || bytes > (size_t(1) << 28)) return false;
      f.get();
      s.data.resize(bytes);
      if (bytes && !f.read(&s.data[0], (std::streamsize)bytes)) return false;
      f.get();
      out.push_back(std::move(s));
   }
   return true;
}

int main(int argc, char** argv)
{
   const bool identify = argc == 2 && std::strcmp(argv[1], "--identify") == 0;
   if (argc != 3 && !identify) {
      std::fprintf(stderr, "usage: fz_rtc_worker <request file> <output file> | --identify\n");
      return 2;
   }
   Dl_info info;
   char real[PATH_MAX];
   const char* lib = dladdr((const void*)&hiprtcCompileProgram, &info) && info.dli_fname ? info.dli_fname : "?";
   std::printf("hiprtc %s\n", ::realpath(lib, real) ? real : lib);
   std::fflush(stdout);
   if (identify) return 0;
   std::vector<Section> req;
   if (!read_request(argv[1], req)) {
      std::fprintf(stderr, "fz_rtc_worker: cannot read the request %s\n", argv[1]);
      return 2;
   }
   const Section* src = nullptr;
   std::vector<const char*> hdr_text, hdr_name, opts;
   for (const Section& s : req) {
      if (s.kind == "source") src = &s;
      else if (s.kind == "header") {
         hdr_text.push_back(s.data.c_str());
         hdr_name.push_back(s.name.c_str());
      } else if (s.kind == "option") opts.push_back(s.data.c_str());
      else return 2;
   }
   if (!src) return 2;
   hiprtcProgram prog;
   if (hiprtcCreateProgram(&prog, src->data.c_str(), src->name.c_str(), (int)hdr_text.size(), hdr_text.data(), hdr_name.data()) != HIPRTC_SUCCESS) {
      std::fprintf(stderr, "fz_rtc_worker: hiprtcCreateProgram failed\n");
      return 4;
   }
   const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
   if (r != HIPRTC_SUCCESS) {
      size_t n = 0;
      hiprtcGetProgramLogSize(prog, &n);
      std::string log(n, ' ');
      if (n) hiprtcGetProgramLog(prog, &log[0]);
      std::ofstream lf(std::string(argv[2]) + ".log", std::ios::binary);
      lf << "hiprtc: " << hiprtcGetErrorString(r) << "\n" << log;
      hiprtcDestroyProgram(&prog);
      return 3;
   }
   size_t n = 0;
   if (hiprtcGetCodeSize(prog, &n) != HIPRTC_SUCCESS || n == 0) return 4;
   std::vector<char> code(n);
   if (hiprtcGetCode(prog, code.data()) != HIPRTC_SUCCESS) return 4;
   hiprtcDestroyProgram(&prog);
   std::ofstream of(argv[2], std::ios::binary);
   of.write(code.data(), (std::streamsize)code.size());
   of.close();
   return of.good() ? 0 : 4;
}
