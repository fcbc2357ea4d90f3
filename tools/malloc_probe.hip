// hipMalloc / hipFree wall time against size on the GPU box (sizes in GiB on the command line).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    for (int rep = 0; rep < 2; ++rep)
        for (int a = 1; a < argc; ++a) {
            const double gib = std::atof(argv[a]);
            const size_t bytes = (size_t)(gib * (1ull << 30));
            void* p = nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            const hipError_t e = hipMalloc(&p, bytes);
            const auto t1 = std::chrono::steady_clock::now();
            if (e != hipSuccess) { std::printf("%.2f GiB: %s\n", gib, hipGetErrorString(e)); continue; }
            (void)hipMemset(p, 0, 4096);
            (void)hipDeviceSynchronize();
            const auto t2 = std::chrono::steady_clock::now();
            (void)hipFree(p);
            const auto t3 = std::chrono::steady_clock::now();
            std::printf("%6.2f GiB  malloc %8.3f ms   free %8.3f ms\n", gib, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                        std::chrono::duration<double, std::milli>(t3 - t2).count());
        }
    return 0;
}
