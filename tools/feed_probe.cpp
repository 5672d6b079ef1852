// Developer probe (not part of the library): costs of the host->device feed primitives on this box (pageable vs pinned vs
// registered memory, threaded staging).  hipcc --offload-arch=gfx950 -O2 -o tools/feed_probe tools/feed_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t chunk = 64ull * 12197888ull, total = 4 * chunk;
    char *host = (char *)malloc(total + 4096);
    memset(host, 1, total + 4096);
    char *aligned = (char *)(((uintptr_t)host + 4095) & ~(uintptr_t)4095);
    void *dev; CK(hipMalloc(&dev, total));
    hipStream_t s; CK(hipStreamCreate(&s));
    // pageable async copy
    double t0 = now(); CK(hipMemcpyAsync(dev, aligned, total, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t1 = now();
    printf("pageable H2D %.1f GB/s\n", total / (t1 - t0) / 1e9);
    for (int rep = 0; rep < 2; rep++) {
        t0 = now(); CK(hipHostRegister(aligned, chunk, hipHostRegisterDefault)); t1 = now();
        double treg = t1 - t0;
        t0 = now(); CK(hipMemcpyAsync(dev, aligned, chunk, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t1 = now();
        double tcp = t1 - t0;
        t0 = now(); CK(hipHostUnregister(aligned)); t1 = now();
        printf("chunk %.0f MB: register %.2f ms, H2D %.2f ms (%.1f GB/s), unregister %.2f ms\n", chunk / 1e6, treg * 1e3, tcp * 1e3, chunk / tcp / 1e9, (t1 - t0) * 1e3);
    }
    // register chunk k+1 on a helper thread while chunk k is copied
    {
        t0 = now();
        CK(hipHostRegister(aligned, chunk, hipHostRegisterDefault));
        for (int k = 0; k < 4; k++) {
            std::thread th;
            if (k + 1 < 4) th = std::thread([&, k] { hipHostRegister(aligned + (k + 1) * chunk, chunk, hipHostRegisterDefault); });
            hipMemcpyAsync((char *)dev + k * chunk, aligned + k * chunk, chunk, hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s);
            if (th.joinable()) th.join();
            hipHostUnregister(aligned + k * chunk);
        }
        t1 = now();
        printf("pipelined register+copy of 4 chunks: %.1f GB/s\n", total / (t1 - t0) / 1e9);
    }
    // threaded memcpy into a pinned ring
    void *pin; CK(hipHostMalloc(&pin, chunk, hipHostMallocDefault));
    for (int T : {1, 4, 8, 12, 16}) {
        t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] { size_t a = chunk * t / T, b = chunk * (t + 1) / T; memcpy((char *)pin + a, aligned + a, b - a); });
        for (auto &x : th) x.join();
        t1 = now();
        printf("memcpy pageable->pinned, %2d threads: %.1f GB/s\n", T, chunk / (t1 - t0) / 1e9);
    }
    t0 = now(); CK(hipMemcpyAsync(dev, pin, chunk, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t1 = now();
    printf("pinned H2D %.1f GB/s\n", chunk / (t1 - t0) / 1e9);
    t0 = now(); void *pin2; CK(hipHostMalloc(&pin2, chunk, hipHostMallocDefault)); t1 = now();
    printf("hipHostMalloc %.0f MB: %.1f ms\n", chunk / 1e6, (t1 - t0) * 1e3);
    return 0;
}
