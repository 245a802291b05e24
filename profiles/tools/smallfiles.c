// Dev tool: what creating N small files (the reference's readmapping/<id>.map.txt, one per miRNA locus: 4,002 on config[1], 16,016 on config[2]) costs on a
// file system, by method.   gcc -O2 -pthread profiles/tools/smallfiles.c -o profiles/tools/bin/smallfiles;  smallfiles <dir> [N] [bytes]
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static int N = 16016, B = 1500, NT = 8;
static char base[512]; static char* body;
static void mk(const char* sub) { char p[600]; snprintf(p, sizeof p, "%s/%s", base, sub); mkdir(p, 0755); }
static void rm(const char* sub) { char c[700]; snprintf(c, sizeof c, "rm -rf %s/%s", base, sub); system(c); }
struct job { int t, mode; };
static void* worker(void* a) {
    struct job* j = a; char p[700], q[700];
    int dfd = -1;
    if (j->mode == 2) { snprintf(p, sizeof p, "%s/d", base); dfd = open(p, O_RDONLY | O_DIRECTORY); }
    for (int k = j->t; k < N; k += (j->mode == 0 ? 1 : NT)) {
        int fd;
        if (j->mode == 2) { snprintf(p, sizeof p, "miRNA-precursor_%d.map.txt", k); fd = openat(dfd, p, O_CREAT | O_WRONLY | O_TRUNC, 0644); }
        else if (j->mode == 3) { snprintf(p, sizeof p, "%s/s%d/miRNA-precursor_%d.map.txt", base, j->t, k); fd = open(p, O_CREAT | O_WRONLY | O_TRUNC, 0644); }
        else { snprintf(p, sizeof p, "%s/d/miRNA-precursor_%d.map.txt", base, k); fd = open(p, O_CREAT | O_WRONLY | O_TRUNC, 0644); }
        if (fd < 0) { perror(p); exit(1); }
        if (write(fd, body, B) != B) exit(1);
        close(fd);
        if (j->mode == 3) { snprintf(q, sizeof q, "%s/d/miRNA-precursor_%d.map.txt", base, k); if (rename(p, q)) { perror("rename"); exit(1); } }
    }
    if (dfd >= 0) close(dfd);
    return 0;
}
static void run(const char* label, int mode, int nt) {
    rm("d"); mk("d");
    if (mode == 3) for (int t = 0; t < nt; t++) { char s[16]; snprintf(s, sizeof s, "s%d", t); rm(s); mk(s); }
    pthread_t th[64]; struct job jb[64];
    int save = NT; NT = nt;
    double t0 = now();
    for (int t = 0; t < nt; t++) { jb[t].t = t; jb[t].mode = (nt == 1 && mode == 1) ? 0 : mode; if (nt == 1 && mode != 3) jb[t].mode = mode == 2 ? 2 : 0; pthread_create(&th[t], 0, worker, &jb[t]); }
    for (int t = 0; t < nt; t++) pthread_join(th[t], 0);
    double t1 = now();
    printf("%-56s %8.1f ms  (%.1f us per file)\n", label, 1e3 * (t1 - t0), 1e6 * (t1 - t0) / N);
    NT = save;
    if (mode == 3) for (int t = 0; t < nt; t++) { char s[16]; snprintf(s, sizeof s, "s%d", t); rm(s); }
}
int main(int argc, char** argv) {
    snprintf(base, sizeof base, "%s/smallfiles_%d", argc > 1 ? argv[1] : "/tmp", getpid());
    if (argc > 2) N = atoi(argv[2]);
    if (argc > 3) B = atoi(argv[3]);
    body = malloc(B); memset(body, 'A', B);
    mkdir(base, 0755);
    printf("%d files of %d bytes under %s\n", N, B, base);
    for (int rep = 0; rep < 2; rep++) {
        run("1 thread, open(path)", 1, 1);
        run("1 thread, openat(dirfd)", 2, 1);
        run("4 threads, one directory", 1, 4);
        run("8 threads, one directory", 1, 8);
        run("16 threads, one directory", 1, 16);
        run("8 threads, own directory each, rename into place", 3, 8);
    }
    char c[700]; snprintf(c, sizeof c, "rm -rf %s", base); system(c);
    return 0;
}
