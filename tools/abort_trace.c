/* Debug helper (LD_PRELOAD): print a native backtrace on abort() / SIGSEGV / SIGBUS / SIGILL / SIGFPE. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <fcntl.h>
static int out_fd = 2;
static void on_abort(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    const char msg[] = "\n==== fatal signal: native backtrace ====\n";
    (void)!write(out_fd, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, out_fd);
    _exit(134);
}
__attribute__((constructor)) static void install(void) {
    /* a test runner may redirect fd 2: keep a file of our own (ABORT_TRACE_FILE) */
    const char* f = getenv("ABORT_TRACE_FILE");
    if (f) { int fd = open(f, O_WRONLY | O_CREAT | O_APPEND, 0644); if (fd >= 0) out_fd = fd; }
    void* warm[4]; (void)backtrace(warm, 4);   /* loads libgcc now, not inside the handler */
    signal(SIGABRT, on_abort); signal(SIGSEGV, on_abort); signal(SIGBUS, on_abort); signal(SIGILL, on_abort); signal(SIGFPE, on_abort);
}
