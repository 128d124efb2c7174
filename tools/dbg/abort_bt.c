// LD_PRELOAD helper: print the C backtrace of whoever calls abort() / fails an assert (debugging aid for GPU-box runs).
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void dump(const char* what) {
    void* bt[64];
    int n = backtrace(bt, 64);
    dprintf(2, "=== %s, C backtrace (%d frames)\n", what, n);
    backtrace_symbols_fd(bt, n, 2);
}
void abort(void) { dump("abort()"); signal(SIGABRT, SIG_DFL); raise(SIGABRT); _exit(134); }
void __assert_fail(const char* a, const char* f, unsigned l, const char* fn) { dprintf(2, "assert %s at %s:%u %s\n", a, f, l, fn); dump("assert"); signal(SIGABRT, SIG_DFL); raise(SIGABRT); _exit(134); }
static void handler(int sig) { dump("signal"); signal(sig, SIG_DFL); raise(sig); }
__attribute__((constructor)) static void init(void) { signal(SIGSEGV, handler); signal(SIGBUS, handler); }
