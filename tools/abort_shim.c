/* LD_PRELOAD debugging aid: native backtrace of whoever calls abort() (or raises SIGABRT), written to $ABORT_BT_FILE.
 *   gcc -shared -fPIC -O1 tools/abort_shim.c -o /tmp/abort_shim.so -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void dump(const char* why) {
  const char* path = getenv("ABORT_BT_FILE");
  int fd = open(path ? path : "/tmp/abort_bt.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
  if (fd < 0) return;
  write(fd, why, strlen(why));
  void* bt[64];
  int n = backtrace(bt, 64);
  backtrace_symbols_fd(bt, n, fd);
  /* what was written to (a possibly captured) stderr just before: the tail of whatever file fd 2 points at */
  {
    int f2 = open("/proc/self/fd/2", O_RDONLY);
    if (f2 >= 0) {
      static char buf[262144];
      off_t end = lseek(f2, 0, SEEK_END);
      off_t from = end > (off_t)sizeof(buf) ? end - (off_t)sizeof(buf) : 0;
      lseek(f2, from, SEEK_SET);
      ssize_t r = read(f2, buf, sizeof(buf));
      write(fd, "== tail of fd 2:\n", 17);
      if (r > 0) write(fd, buf, (size_t)r);
      close(f2);
    }
  }
  close(fd);
}

void abort(void) {
  dump("== abort() called\n");
  void (*real)(void) = (void (*)(void))dlsym(RTLD_NEXT, "abort");
  real();
  _exit(134);
}

