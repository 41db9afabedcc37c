// The RK45 loop of hlm_rk45_kernel.h with LANE REFILL: a lane whose hillslope is finished -- reached
// tf, or flagged stiff after a few hundred attempts while its neighbours need 16 000 -- writes its
// results and takes the next hillslope from a queue, so a wave keeps 64 lanes busy as long as there is
// work (north_star: "wavefront ballot/prefix on the step-accept mask to keep divergent
// adaptive steppers converged").
//
// Restates reference src/solver/rk45_kernel.cu:17-176 like rk45_multi_kernel; what changes is only WHICH
// lane integrates WHICH system (rk45_kernel.cu:33: "sys = blockIdx.x * blockDim.x + threadIdx.x" -- one
// fixed system per thread, so a block of the reference idles behind its slowest thread).  Every system is
// integrated by exactly the same sequence of operations as before: results are bit-identical.
//
// When it matters: SURVEY 8d's C3 generator draws daily temperatures from U[-5, 15]; below 0 C the
// reference's ET stub flags a hillslope within hours, 42 % of them over 30 days.  Measured on MI355X
// (1 M hillslopes x 30 days): fixed assignment 160 ms at 40.6 of 64 lanes active per VALU instruction.
// The launch is a PERSISTENT grid: as many one-wave workgroups as the GPU holds at once, each starting on
// 64 consecutive hillslopes and then drawing from a device counter; one atomic per refill event (the
// first finished lane draws for all lanes of the wave that finished in the same iteration, a popcount
// prefix hands out the indices).  No dense output, no resume state: those variants keep the fixed map.
#pragma once

#include "hlm_rk45_kernel.h"

namespace hlm {

template <class Model, int MATH, bool FIXED>
__global__ void __launch_bounds__(WAVE, MATH == MATH_FAST ? HLM_MIN_WAVES_PER_SIMD_LEAN : HLM_MIN_WAVES_PER_SIMD)
    rk45_refill_kernel(const Rk45Args A, unsigned int* __restrict__ queue)  // *queue: next unassigned system (starts at gridDim.x * 64)
{
    using namespace dp;
    constexpr int NEQ = Model::N_EQ;
    const int lane = (int)threadIdx.x;
    int sys = (int)(blockIdx.x * WAVE) + lane;
    bool retired = sys >= A.ns;  // nothing left for this lane

    typename Model::Local P = Model::template load<MATH>(A.sp, retired ? 0 : sys);
    double y[NEQ];
#pragma unroll
    for (int i = 0; i < NEQ; ++i) y[i] = retired ? 0.0 : A.y0[(size_t)sys * NEQ + i];
    HeldForcing<Model, false> held;
    held.init(A, retired ? 0 : sys);

    double t = A.t0;
    double h = FIXED ? A.fixed_step : A.h0;
    const double tf = A.tf;
    int reject_count = 0;
    int status = HLM_SYS_OK;
    uint32_t n_attempt = 0, n_accept = 0;
    bool have_k0 = false;
    double k0[NEQ];
    typename Model::Forc FV = Model::prep_forcing(P, held.F);

    while (__builtin_amdgcn_ballot_w64(!retired) != 0ull) {
        const bool active = !retired && t < tf && status == HLM_SYS_OK && n_attempt < A.cap;
        // ---- a finished hillslope: results out, next one in -------------------------------------
        const bool finished = !retired && !active;
        const unsigned long long fin = __builtin_amdgcn_ballot_w64(finished);
        if (fin != 0ull) {
            unsigned base = 0;
            if (finished) {
                if (status == HLM_SYS_OK && t < tf) status = HLM_SYS_CAPPED;  // stopped by the attempt cap
                if (A.n_attempt) A.n_attempt[sys] = n_attempt;
                if (A.n_accept) A.n_accept[sys] = n_accept;
                const bool flagged = status != HLM_SYS_OK && t < tf;  // rk45_kernel.cu:167-170
                A.stiff[sys] = flagged ? status : HLM_SYS_OK;
#pragma unroll
                for (int i = 0; i < NEQ; ++i) A.y_final[(size_t)sys * NEQ + i] = flagged ? 0.0 : y[i];
                if (A.flag_state && flagged) record_flag_state<NEQ>(A.flag_state, sys, y, t);
                // one draw for all lanes that finished in this iteration
                if (lane == (int)__builtin_ctzll(fin)) base = atomicAdd(queue, (unsigned)__builtin_popcountll(fin));
            }
            base = (unsigned)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(fin));
            if (finished) {
                const unsigned mine = base + (unsigned)__builtin_popcountll(fin & ((1ull << lane) - 1ull));
                if (mine < (unsigned)A.ns) {
                    sys = (int)mine;
                    P = Model::template load<MATH>(A.sp, sys);
#pragma unroll
                    for (int i = 0; i < NEQ; ++i) y[i] = A.y0[(size_t)sys * NEQ + i];
                    held = HeldForcing<Model, false>{};
                    held.init(A, sys);
                    FV = Model::prep_forcing(P, held.F);
                    t = A.t0;
                    h = FIXED ? A.fixed_step : A.h0;
                    reject_count = 0;
                    status = HLM_SYS_OK;
                    n_attempt = 0;
                    n_accept = 0;
                    have_k0 = false;
                } else {
                    retired = true;
                }
            }
            continue;  // (a lane that just took a hillslope starts it in the next iteration, with the others)
        }

        // ---- one attempt (the body of rk45_multi_kernel's loop) -------------------------------------
        if (active) {
            if (t + h > tf) h = tf - t;  // rk45_kernel.cu:54
            if (held.refresh(A, P, t, FV)) have_k0 = false;
            ++n_attempt;
            double k1[NEQ], k2[NEQ], k3[NEQ], k4[NEQ], k5[NEQ], k6[NEQ], yt[NEQ];
            [[maybe_unused]] double err = 0.0, jump = 0.0;
            attempt_on_cheapest_path<Model, MATH, FIXED, false>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
            bool accepted = true;
            [[maybe_unused]] double fac = 1.0;
            if constexpr (!FIXED) {
                fac = A.safety * inv_fifth_root<MATH>(err + 1e-16);  // rk45_kernel.cu:151,156
                if (err <= 1.0) {
                    reject_count = 0;
                    if (jump > SLOPE_JUMP_THRESH) {  // rk45_kernel.cu:132-136
                        h = fmax(h * 0.5, A.h0 * MIN_STEP_FRACTION);
                        accepted = false;
                    }
                } else {  // rk45_kernel.cu:153-163
                    ++reject_count;
                    double f = fmin(fac, 1.0);
                    f = fmin(fmax(f, A.min_scale), A.max_scale);
                    h *= f;
                    if (reject_count > 5 || h < (tf - A.t0) * MIN_STEP_FRACTION) status = HLM_SYS_STIFF;
                    accepted = false;
                }
            }
            if (accepted) {
#pragma unroll
                for (int i = 0; i < NEQ; ++i) {
                    y[i] = yt[i];
                    k0[i] = k6[i];  // = rhs(t1, y_new, F): next attempt's k0 unless the forcing moves
                }
                t += h;
                ++n_accept;
                if constexpr (FIXED) h = A.fixed_step;
                else h *= vclamp_uniform(fac, A.min_scale, A.max_scale);  // rk45_kernel.cu:151-152
            }
        }
    }
}

}  // namespace hlm
