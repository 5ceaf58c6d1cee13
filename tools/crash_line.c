/* bench.py helper: if the process dies on a fatal signal while an OPTIONAL leg runs (the hipGraph capture of the data-parallel step on N > 1
 * ranks was never run on real xGMI hardware), rank 0 still prints the result line it had already measured.  Async-signal-safe: the
 * handler only write()s a preformatted buffer and _exit()s.   gcc -O2 -shared -fPIC tools/crash_line.c -o tools/libcrash_line.so */
#include <signal.h>
#include <string.h>
#include <unistd.h>

static char g_line[8192];
static size_t g_len;
static int g_code;

static void on_fatal(int sig) {
  (void)sig;
  if (g_len) {
    ssize_t r = write(1, g_line, g_len);
    (void)r;
  }
  _exit(g_code);
}

/* arm: line (may be empty: just leave quietly) is printed on SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL, then _exit(code) */
int crash_line_arm(const char* line, int code) {
  size_t n = line ? strlen(line) : 0;
  if (n + 2 > sizeof g_line) return -1;
  if (n) { memcpy(g_line, line, n); g_line[n++] = '\n'; }
  g_len = n;
  g_code = code;
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_fatal;
  sigemptyset(&sa.sa_mask);
  const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
  for (unsigned i = 0; i < sizeof sigs / sizeof sigs[0]; ++i) sigaction(sigs[i], &sa, 0);
  return 0;
}

int crash_line_disarm(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = SIG_DFL;
  sigemptyset(&sa.sa_mask);
  const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
  for (unsigned i = 0; i < sizeof sigs / sizeof sigs[0]; ++i) sigaction(sigs[i], &sa, 0);
  g_len = 0;
  return 0;
}
