// Debug aid (not part of the library): every intermediate of strict_pow_two_thirds_core on the device, recomputed on
// the host from the device's own seed; prints the first intermediate that differs.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast -I tiger_hlm_gpu_amd/csrc tools/debug/strict_pow_debug.hip -o /tmp/spd && /tmp/spd
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct Tr { double m, x1, w, w2, w3, w3e, e0, e, s, lg, t, b, be, sum, res; int ex, r, k; float L1; };

__global__ void trace(const double* in, Tr* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    Tr T;
    T.ex = __builtin_amdgcn_frexp_exp(x);
    T.m = __builtin_amdgcn_frexp_mant(x);
    const unsigned e3 = (unsigned)(T.ex + 1080);
    const unsigned k3 = (e3 * 43691u) >> 17;
    T.r = (int)(e3 - 3u * k3);
    T.k = (int)k3 - 360;
    T.x1 = __builtin_amdgcn_ldexp(T.m, T.r);
    T.L1 = (float)T.r + __builtin_amdgcn_logf((float)T.m);
    T.w = (double)__builtin_amdgcn_exp2f(T.L1 * (-1.0f / 3.0f));
    T.w2 = T.w * T.w;
    T.w3 = T.w2 * T.w;
    T.w3e = __builtin_fma(T.w2, T.w, -T.w3);
    T.e0 = __builtin_fma(-T.x1, T.w3, 1.0);
    T.e = __builtin_fma(-T.x1, T.w3e, T.e0);
    double s = __builtin_fma(T.e, 14.0 / 81.0, 2.0 / 9.0);
    T.s = __builtin_fma(T.e, s, 1.0 / 3.0);
    T.lg = (double)(T.L1 + (float)(3 * T.k));
    T.t = __builtin_fma(T.e, T.s, -2.5651598643722064e-17 * T.lg);
    T.b = T.x1 * T.w;
    asm volatile("" : "+v"(T.b));   // as HLM_POW_OPAQUE: without it -ffp-contract=fast fuses this product into the final sum
    T.be = __builtin_fma(T.x1, T.w, -T.b);
    T.sum = T.b + __builtin_fma(T.b, T.t, T.be);
    T.res = __builtin_amdgcn_ldexp(T.sum, 2 * T.k);
    out[i] = T;
}

int main()
{
    const int n = 4096;
    std::vector<double> x(n);
    srand(5);
    for (int i = 0; i < n; ++i) x[i] = exp2(3.0 * rand() / RAND_MAX);
    double* d_in; Tr* d_out;
    hipMalloc(&d_in, n * sizeof(double)); hipMalloc(&d_out, n * sizeof(Tr));
    hipMemcpy(d_in, x.data(), n * sizeof(double), hipMemcpyHostToDevice);
    trace<<<(n + 255) / 256, 256>>>(d_in, d_out, n);
    std::vector<Tr> T(n);
    hipMemcpy(T.data(), d_out, n * sizeof(Tr), hipMemcpyDeviceToHost);
    int bad[16] = {0};
    const char* names[] = {"m", "x1", "w2", "w3", "w3e", "e0", "e", "s", "lg", "t", "b", "be", "sum", "res", "w-has-more-than-24-bits"};
    for (int i = 0; i < n; ++i) {
        const Tr& d = T[i];
        int ex; const double m = frexp(x[i], &ex);
        const double x1 = ldexp(m, d.r), w = d.w;
        const double w2 = w * w, w3 = w2 * w, w3e = fma(w2, w, -w3);
        const double e0 = fma(-x1, w3, 1.0), e = fma(-x1, w3e, e0);
        double s = fma(e, 14.0 / 81.0, 2.0 / 9.0); s = fma(e, s, 1.0 / 3.0);
        const double lg = (double)(d.L1 + (float)(3 * d.k));
        const double t = fma(e, s, -2.5651598643722064e-17 * lg);
        const double b = x1 * w, be = fma(x1, w, -b);
        const double sum = b + fma(b, t, be), res = ldexp(sum, 2 * d.k);
        const double hv[] = {m, x1, w2, w3, w3e, e0, e, s, lg, t, b, be, sum, res};
        const double dv[] = {d.m, d.x1, d.w2, d.w3, d.w3e, d.e0, d.e, d.s, d.lg, d.t, d.b, d.be, d.sum, d.res};
        for (int j = 0; j < 14; ++j)
            if (memcmp(&hv[j], &dv[j], 8) != 0) {
                if (bad[j]++ < 3) printf("x=%.17g: %s device %.17g host %.17g\n", x[i], names[j], dv[j], hv[j]);
                break;
            }
        if ((double)(float)w != w) bad[14]++;
    }
    for (int j = 0; j < 15; ++j) printf("%s: %d first differences\n", names[j], bad[j]);
    return 0;
}
