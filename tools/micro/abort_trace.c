/* LD_PRELOAD helper: native backtrace on SIGABRT / SIGSEGV (the GPU boxes have no gdb).
 * gcc -O1 -g -fPIC -shared tools/micro/abort_trace.c -o /tmp/abort_trace.so */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_sig(int sig) {
  void *bt[64];
  const char msg[] = "\n[abort_trace] native backtrace:\n";
  if (write(2, msg, sizeof msg - 1) < 0) _exit(1);
  int n = backtrace(bt, 64);
  backtrace_symbols_fd(bt, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void init(void) {
  signal(SIGABRT, on_sig);
  signal(SIGSEGV, on_sig);
}
