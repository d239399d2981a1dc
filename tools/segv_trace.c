/* tools/segv_trace.c -- LD_PRELOAD helper for the GPU box (no gdb there): prints a backtrace (module + offset per frame: resolve with addr2line -e <module> <offset> in the
 * build container, where the same binaries live) when the process dies of SIGSEGV / SIGABRT / SIGBUS.   gcc -shared -fPIC -O1 tools/segv_trace.c -o tools/segv_trace.so */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void handler(int sig, siginfo_t* si, void* ctx) {
  void* frames[64];
  char msg[96];
  int n = snprintf(msg, sizeof msg, "\n[segv_trace] signal %d at address %p\n", sig, si ? si->si_addr : 0);
  if (n > 0) (void)!write(2, msg, (size_t)n);
  n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  _exit(128 + sig);
}
__attribute__((constructor)) static void install(void) {
  struct sigaction sa; sa.sa_sigaction = handler; sigemptyset(&sa.sa_mask); sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, 0); sigaction(SIGBUS, &sa, 0); sigaction(SIGABRT, &sa, 0);
}
