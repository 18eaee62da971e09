// LD_PRELOAD helper: native backtrace on SIGSEGV (debugging only)
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void h(int sig) { void *bt[64]; int n = backtrace(bt, 64); dprintf(2, "== native backtrace (signal %d)\n", sig); backtrace_symbols_fd(bt, n, 2); _exit(139); }
__attribute__((constructor)) static void init(void) { signal(SIGSEGV, h); signal(SIGBUS, h); signal(SIGABRT, h); }
