// Host build of csrc/hlm_strict_pow.h for tests/test_strict_pow_cpu.py (g++ -O2 -ffp-contract=off -shared -fPIC):
// the STRICT-mode powers evaluated over arrays, no GPU involved.
#include "hlm_strict_pow.h"

extern "C" {
void strict_pow_two_thirds_array(const double* x, double* out, long n)
{
    for (long i = 0; i < n; ++i) out[i] = hlm::strict_pow_two_thirds(x[i]);
}
void strict_pow_fifth_array(const double* x, double* out, long n)
{
    for (long i = 0; i < n; ++i) out[i] = hlm::strict_pow_fifth(x[i]);
}
void strict_pow_fourth_array(const double* x, double* out, long n)
{
    for (long i = 0; i < n; ++i) out[i] = hlm::strict_pow_fourth(x[i]);
}
}
