/* Dev tool: LD_PRELOAD=tools/_bin/libsegvbt.so -- prints the native backtrace of a SIGSEGV / SIGBUS / SIGABRT to stderr (with the
   /proc/self/maps lines of the frames' modules) before the default action.  gcc -shared -fPIC -O1 -o tools/_bin/libsegvbt.so tools/segv_backtrace.c */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* si, void* ctx)
{
   (void)ctx;
   void* frames[64];
   char line[256];
   int n = backtrace(frames, 64);
   int len = snprintf(line, sizeof line, "\n[segvbt] signal %d at address %p, %d frames\n", sig, si ? si->si_addr : 0, n);
   if (write(2, line, (size_t)len) < 0) {}
   for (int i = 0; i < n; ++i) {
      Dl_info info;
      if (dladdr(frames[i], &info) && info.dli_fname)
         len = snprintf(line, sizeof line, "[segvbt] #%d %p %s + 0x%lx (%s)\n", i, frames[i], info.dli_fname,
                        (unsigned long)((char*)frames[i] - (char*)info.dli_fbase), info.dli_sname ? info.dli_sname : "?");
      else
         len = snprintf(line, sizeof line, "[segvbt] #%d %p ?\n", i, frames[i]);
      if (write(2, line, (size_t)len) < 0) {}
   }
   signal(sig, SIG_DFL);
   raise(sig);
}

__attribute__((constructor)) static void install(void)
{
   struct sigaction sa;
   memset(&sa, 0, sizeof sa);
   sa.sa_sigaction = handler;
   sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_RESETHAND;
   static char stack[1 << 16];
   stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
   sigaltstack(&ss, 0);
   sigaction(SIGSEGV, &sa, 0);
   sigaction(SIGBUS, &sa, 0);
   void* warm[4];
   backtrace(warm, 4);        /* loads libgcc now, not inside the handler */
}
