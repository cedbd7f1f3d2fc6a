// LD_PRELOAD helper (debugging aid): on SIGUSR1 print the receiving thread's backtrace to stderr.
//   gcc -O1 -g -shared -fPIC tools/debug/bt_on_usr1.c -o tools/debug/libbt.so
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void handler(int sig) {
    (void)sig;
    void* frames[64];
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    const char nl[] = "---- end of backtrace\n";
    (void)!write(2, nl, sizeof nl - 1);
}
__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    sa.sa_handler = handler;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = 0;
    sigaction(SIGUSR1, &sa, 0);
}
