/* fuzz_guard.c -- SIGSEGV / SIGBUS reporter of scripts/gpu_fuzz.py (test tooling, not product code).
 * The fuzz harness keeps every buffer it hands to libdebwt_hip.so in page-granular mappings with PROT_NONE guard pages
 * and retires them behind PROT_NONE; a write that lands there faults in the thread that issued it.  This handler prints
 * the faulting address, the thread and a native backtrace (the culprit, also when it is not a Python thread), then
 * lets the default action end the process.  gcc -O1 -g -shared -fPIC -o fuzz_guard.so fuzz_guard.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t *si, void *uc) {
    (void)uc;
    char line[160];
    int n = snprintf(line, sizeof line, "\nFUZZ_GUARD: signal %d at address %p (si_code %d) in thread %ld\n", sig, si->si_addr,
                     si->si_code, (long)syscall(SYS_gettid));
    if (n > 0) (void)!write(2, line, (size_t)n);
    void *bt[64];
    int d = backtrace(bt, 64);
    backtrace_symbols_fd(bt, d, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int fuzz_guard_install(void) {
    void *bt[4];
    backtrace(bt, 4);                       /* loads libgcc now: not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    if (sigaction(SIGSEGV, &sa, NULL)) return -1;
    if (sigaction(SIGBUS, &sa, NULL)) return -1;
    return 0;
}
