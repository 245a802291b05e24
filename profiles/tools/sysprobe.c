// Dev tool (LD_PRELOAD): time spent inside libc's open / openat / ioctl / read / mmap / pthread_create wrappers, with the slowest individual calls, to see
// what a HIP start-up (hipInit = 0.2 s on the GPU box) is made of without strace.   gcc -O2 -shared -fPIC profiles/tools/sysprobe.c -o profiles/tools/bin/sysprobe.so -ldl
//   LD_PRELOAD=profiles/tools/bin/sysprobe.so profiles/tools/bin/ctx_probe bare
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
enum { OPEN, IOCTL, READ, MMAP, THREAD, NK };
static const char* kname[NK] = {"open/openat", "ioctl", "read", "mmap", "pthread_create"};
static double tot[NK]; static long cnt[NK];
#define NSLOW 24
static struct { double t, at; char what[160]; } slow[NSLOW];
static double t_first;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static void note(int k, double t0, const char* what) {
    double t = now() - t0;
    pthread_mutex_lock(&mu);
    if (!t_first) t_first = t0;
    tot[k] += t; cnt[k]++;
    int m = 0; for (int i = 1; i < NSLOW; i++) if (slow[i].t < slow[m].t) m = i;
    if (t > slow[m].t) { slow[m].t = t; slow[m].at = t0 - t_first; snprintf(slow[m].what, sizeof slow[m].what, "%s %s", kname[k], what); }
    pthread_mutex_unlock(&mu);
}
int open(const char* p, int fl, ...) { static int (*f)(const char*, int, ...); if (!f) f = dlsym(RTLD_NEXT, "open"); va_list a; va_start(a, fl); int md = va_arg(a, int); va_end(a); double t0 = now(); int r = f(p, fl, md); note(OPEN, t0, p); return r; }
int open64(const char* p, int fl, ...) { static int (*f)(const char*, int, ...); if (!f) f = dlsym(RTLD_NEXT, "open64"); va_list a; va_start(a, fl); int md = va_arg(a, int); va_end(a); double t0 = now(); int r = f(p, fl, md); note(OPEN, t0, p); return r; }
int openat(int d, const char* p, int fl, ...) { static int (*f)(int, const char*, int, ...); if (!f) f = dlsym(RTLD_NEXT, "openat"); va_list a; va_start(a, fl); int md = va_arg(a, int); va_end(a); double t0 = now(); int r = f(d, p, fl, md); note(OPEN, t0, p); return r; }
FILE* fopen(const char* p, const char* m) { static FILE* (*f)(const char*, const char*); if (!f) f = dlsym(RTLD_NEXT, "fopen"); double t0 = now(); FILE* r = f(p, m); note(OPEN, t0, p); return r; }
FILE* fopen64(const char* p, const char* m) { static FILE* (*f)(const char*, const char*); if (!f) f = dlsym(RTLD_NEXT, "fopen64"); double t0 = now(); FILE* r = f(p, m); note(OPEN, t0, p); return r; }
int ioctl(int fd, unsigned long req, ...) { static int (*f)(int, unsigned long, ...); if (!f) f = dlsym(RTLD_NEXT, "ioctl"); va_list a; va_start(a, req); void* arg = va_arg(a, void*); va_end(a); double t0 = now(); int r = f(fd, req, arg); char b[64]; snprintf(b, sizeof b, "fd %d req 0x%lx (nr 0x%lx)", fd, req, req & 0xff); note(IOCTL, t0, b); return r; }
ssize_t read(int fd, void* buf, size_t n) { static ssize_t (*f)(int, void*, size_t); if (!f) f = dlsym(RTLD_NEXT, "read"); double t0 = now(); ssize_t r = f(fd, buf, n); char b[48]; snprintf(b, sizeof b, "fd %d n %zu", fd, n); note(READ, t0, b); return r; }
void* mmap(void* a, size_t n, int pr, int fl, int fd, off_t o) { static void* (*f)(void*, size_t, int, int, int, off_t); if (!f) f = dlsym(RTLD_NEXT, "mmap"); double t0 = now(); void* r = f(a, n, pr, fl, fd, o); char b[64]; snprintf(b, sizeof b, "n %zu fd %d flags 0x%x", n, fd, fl); note(MMAP, t0, b); return r; }
int pthread_create(pthread_t* t, const pthread_attr_t* at, void* (*fn)(void*), void* arg) { static int (*f)(pthread_t*, const pthread_attr_t*, void* (*)(void*), void*); if (!f) f = dlsym(RTLD_NEXT, "pthread_create"); double t0 = now(); int r = f(t, at, fn, arg); note(THREAD, t0, ""); return r; }
__attribute__((destructor)) static void dump(void) {
    fprintf(stderr, "[sysprobe] pid %d\n", getpid());
    for (int k = 0; k < NK; k++) fprintf(stderr, "[sysprobe] %-16s %6ld calls %8.2f ms\n", kname[k], cnt[k], 1e3 * tot[k]);
    for (int i = 0; i < NSLOW; i++) { int m = -1; for (int j = 0; j < NSLOW; j++) if (slow[j].t > 0 && (m < 0 || slow[j].at < slow[m].at)) m = j; if (m < 0) break;
        fprintf(stderr, "[sysprobe]   at %7.2f ms  %7.2f ms  %s\n", 1e3 * slow[m].at, 1e3 * slow[m].t, slow[m].what); slow[m].t = 0; }
}
