// What a write-only stream reaches on this GPU, by store instruction: the dense kernel's rows leave as 8-byte stores of a
// wave (512 contiguous bytes per instruction).  Variants: 8 / 16 bytes per lane, plain / nontemporal (global_store ... nt).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/hbm_store_probe tools/hbm_store_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NT>
__global__ void fill8(double* __restrict__ p, size_t n, double v)
{
    // grid-stride over 512-byte wave rows: consecutive waves write consecutive rows, as the dense flush does
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
template <int NT>
__global__ void fill16(double2* __restrict__ p, size_t n, double v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) { __builtin_nontemporal_store(v, &p[i].x); __builtin_nontemporal_store(v, &p[i].y); }
        else p[i] = make_double2(v, v);
    }
}
// five 512-byte rows of one wave, 5 x 8 bytes per lane at a stride of 512 bytes: exactly the flush of one dense row
template <int NT>
__global__ void fill_rows(double* __restrict__ p, size_t nrows, double v)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x % 64, nw = (size_t)gridDim.x * blockDim.x / 64;
    for (size_t r = wave; r < nrows; r += nw) {
        double* o = p + r * 320 + lane;
#pragma unroll
        for (int c = 0; c < 5; ++c) { if (NT) __builtin_nontemporal_store(v, o + c * 64); else o[c * 64] = v; }
    }
}

int main(int argc, char** argv)
{
    const size_t bytes = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 57640000000ull;
    const size_t n = bytes / 8 / 320 * 320;
    double* p = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&p), n * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto timed = [&](const char* name, auto launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        std::printf("%-34s %8.2f ms  %7.0f GB/s\n", name, best, n * 8 / best / 1e6);
    };
    for (int blocks : {2048, 8192, 65536}) {
        std::printf("grid %d x 256\n", blocks);
        timed("8 B per lane", [&] { hipLaunchKernelGGL(fill8<0>, dim3(blocks), dim3(256), 0, 0, p, n, 1.5); });
        timed("8 B per lane, nontemporal", [&] { hipLaunchKernelGGL(fill8<1>, dim3(blocks), dim3(256), 0, 0, p, n, 1.5); });
        timed("16 B per lane", [&] { hipLaunchKernelGGL(fill16<0>, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<double2*>(p), n / 2, 1.5); });
        timed("16 B per lane, nontemporal", [&] { hipLaunchKernelGGL(fill16<1>, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<double2*>(p), n / 2, 1.5); });
        timed("dense rows (5 x 512 B per wave)", [&] { hipLaunchKernelGGL(fill_rows<0>, dim3(blocks), dim3(256), 0, 0, p, n / 320, 1.5); });
        timed("dense rows, nontemporal", [&] { hipLaunchKernelGGL(fill_rows<1>, dim3(blocks), dim3(256), 0, 0, p, n / 320, 1.5); });
    }
    CHECK(hipMemset(p, 0, 1 << 20));
    timed("hipMemsetAsync", [&] { hipMemsetAsync(p, 0, n * 8, 0); });
    return 0;
}
