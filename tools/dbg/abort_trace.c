/* LD_PRELOAD helper: C-level backtrace when a library calls abort() (also on SIGABRT; run pytest with -p no:faulthandler). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <unistd.h>
static void dump(const char* why) {
    void* b[96];
    int n = backtrace(b, 96);
    write(2, why, __builtin_strlen(why));
    backtrace_symbols_fd(b, n, 2);
}
static void on_abort(int s) { dump("\n[abort_trace] SIGABRT\n"); signal(s, SIG_DFL); raise(s); }
void abort(void) { dump("\n[abort_trace] abort() called\n"); signal(SIGABRT, SIG_DFL); raise(SIGABRT); _exit(134); }
__attribute__((constructor)) static void init(void) { signal(SIGABRT, on_abort); }
