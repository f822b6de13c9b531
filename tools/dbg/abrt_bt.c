// LD_PRELOAD helper: native backtrace on SIGABRT (who calls abort()?).  gcc -shared -fPIC -o abrt_bt.so abrt_bt.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void h(int sig)
{
    void *bt[64];
    int n = backtrace(bt, 64);
    dprintf(2, "\n==== SIGABRT backtrace (%d frames)\n", n);
    backtrace_symbols_fd(bt, n, 2);
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}
__attribute__((constructor)) static void init(void) { signal(SIGABRT, h); }
