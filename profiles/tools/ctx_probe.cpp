// Dev tool: where the 0.2 s of "library load + device context" and the 0.13 s of process exit of a CLI run go.  Times every step of a bare HIP
// start-up, then the same through libmirprefer.so (dlopen + mirp_create), then a 6 GB allocation; the parent clocks the child's exit by mode.
//   hipcc -O2 --offload-arch=gfx950 profiles/tools/ctx_probe.cpp -o profiles/tools/bin/ctx_probe -ldl
//   profiles/tools/bin/ctx_probe                 (GPU box, from the repo root)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int* p) { p[threadIdx.x] = threadIdx.x; }
#define T(label, ...) do { double t0 = now(); __VA_ARGS__; std::printf("  %-44s %8.2f ms\n", label, 1e3 * (now() - t0)); } while (0)

static int child(const char* mode, const char* lib) {
    double t_start = now();
    if (!std::strcmp(mode, "bare") || !std::strncmp(mode, "exit", 4)) {
        int n = 0; hipStream_t st; int* p = nullptr; void* big = nullptr;
        T("hipInit(0)", hipInit(0));
        T("hipGetDeviceCount", hipGetDeviceCount(&n));
        T("hipSetDevice(0)", hipSetDevice(0));
        hipDeviceProp_t prop;
        T("hipGetDeviceProperties", hipGetDeviceProperties(&prop, 0));
        T("hipStreamCreateWithFlags", hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        T("hipMalloc 4 KB (first)", hipMalloc((void**)&p, 4096));
        T("first kernel launch + sync", { k<<<1, 64, 0, st>>>(p); hipStreamSynchronize(st); });
        T("second kernel launch + sync", { k<<<1, 64, 0, st>>>(p); hipStreamSynchronize(st); });
        T("hipMalloc 6 GB", hipMalloc(&big, 6ull << 30));
        T("hipMemsetAsync 6 GB + sync", { hipMemsetAsync(big, 0, 6ull << 30, st); hipStreamSynchronize(st); });
        T("hipMalloc 64 MB", { void* q; hipMalloc(&q, 64 << 20); });
        if (!std::strcmp(mode, "exit_free")) T("hipFree 6 GB", hipFree(big));
        if (!std::strcmp(mode, "exit_hsa")) {          // release the runtime (and with it /dev/kfd) before the process ends: does the kernel-side teardown move in here?
            typedef int (*shut_t)(void);
            shut_t shut = (shut_t)dlsym(RTLD_DEFAULT, "hsa_shut_down");
            std::printf("  hsa_shut_down %s\n", shut ? "found" : "not found");
            if (shut) { int rc = -1; T("hsa_shut_down #1", rc = shut()); std::printf("  rc %d\n", rc); T("hsa_shut_down #2", rc = shut()); std::printf("  rc %d\n", rc); }
            std::printf("  %-44s %8.2f ms\n", "child total before exit", 1e3 * (now() - t_start));
            std::fflush(stdout);
            _exit(0);
        }
        if (!std::strcmp(mode, "exit_reset")) { T("hipStreamDestroy", hipStreamDestroy(st)); T("hipDeviceReset", hipDeviceReset()); }
        std::printf("  %-44s %8.2f ms\n", "child total before exit", 1e3 * (now() - t_start));
        std::fflush(stdout);
        if (!std::strcmp(mode, "exit__exit")) _exit(0);
        if (!std::strcmp(mode, "exit_quick")) quick_exit(0);
        return 0;
    }
    // through the product library
    void* h = nullptr;
    T("dlopen(libmirprefer.so)", h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL));
    if (!h) { std::printf("dlopen failed: %s\n", dlerror()); return 1; }
    typedef int (*create_t)(int, void**);
    create_t create = (create_t)dlsym(h, "mirp_create");
    void* ctx = nullptr;
    T("mirp_create(0)", create(0, &ctx));
    std::printf("  %-44s %8.2f ms\n", "child total before exit", 1e3 * (now() - t_start));
    std::fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const char* lib = argc > 2 ? argv[2] : "mir-prefer_amd/libmirprefer.so";
    if (argc > 1 && std::strcmp(argv[1], "all")) return child(argv[1], lib);
    const char* modes[] = {"bare", "bare", "lib", "lib", "exit_return", "exit__exit", "exit_quick", "exit_free", "exit_reset", "exit_reset"};
    for (const char* m : modes) {
        std::printf("== %s\n", m); std::fflush(stdout);
        double t0 = now();
        pid_t pid = fork();
        if (pid == 0) { execl(argv[0], argv[0], m, lib, (char*)nullptr); _exit(127); }
        int st = 0; waitpid(pid, &st, 0);
        std::printf("  parent: spawn -> reaped %.1f ms (status %d)\n", 1e3 * (now() - t0), st);
    }
    return 0;
}
