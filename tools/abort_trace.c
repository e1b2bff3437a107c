/* Debug aid: LD_PRELOAD this to get a native backtrace when something calls abort() (the HIP runtime does on
 * some internal errors, without a message).  gcc -shared -fPIC -o /tmp/abort_trace.so tools/abort_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void handler(int sig) {
    void *bt[64];
    int n = backtrace(bt, 64);
    const char msg[] = "\n==== native backtrace at SIGABRT ====\n";
    write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) { signal(SIGABRT, handler); }
