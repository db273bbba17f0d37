// Shared host/device helpers for the gfx950 replay data path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pfrl_amd.h"

extern "C" void pfrl_set_error(const char *msg);

// Output size from which the gather kernels use non-temporal stores (the minibatch is
// read once by the next kernel and would only evict useful lines from L2 / MALL).
// PFRL_NT_MIN_BYTES overrides the default for tuning.
int64_t pfrl_nt_min_bytes();

// bench.py roofline support (defined in replay.hip): when profiling is enabled,
// returns a start/stop event pair to attach to ONE dispatch with
// hipExtLaunchKernelGGL; both are nullptr otherwise.
enum { PFRL_PROFILE_BATCH_EXPERIENCES = 0, PFRL_PROFILE_BATCH_STATES_U8 = 1, PFRL_PROFILE_GAE_SCAN = 2,
       PFRL_PROFILE_ADV_STATS = 3,
       // pfrl_batch_states_u8_raw_nhwc4: 2 bytes moved per frame byte (BATCH_STATES_U8 prices the
       // fp32 form's 5); mirrored in pfrl_amd/ops.py and benchkit/roofline.py
       PFRL_PROFILE_BATCH_STATES_U8_RAW = 4 };
void pfrl_profile_events(int kind, int64_t units, hipEvent_t *start, hipEvent_t *stop);

#define PFRL_CHECK_ARG(cond, msg)      \
    do {                               \
        if (!(cond)) {                 \
            pfrl_set_error(msg);       \
            return PFRL_ERR_ARG;       \
        }                              \
    } while (0)

#define PFRL_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) {                              \
            pfrl_set_error(hipGetErrorString(e__));           \
            return (int)e__;                                  \
        }                                                     \
        return 0;                                             \
    } while (0)

// ---------------------------------------------------------------------------
// NEP-50 typed scalars (see include/pfrl_amd.h).  Every arithmetic step is a
// single correctly-rounded IEEE operation; the file is built with
// -ffp-contract=off so nothing fuses.
// ---------------------------------------------------------------------------
struct TV {
    double v;
    int t;
};

__device__ __forceinline__ TV mk_tv(double v, int t) {
    TV r;
    r.v = v;
    r.t = t;
    return r;
}

// Result type of a binary op between two PRESENT scalars.  The tags are ordered
// PY(1) < F32(2) < F64(3) exactly like the NEP-50 promotion lattice, so the
// result type is the larger tag.
__device__ __forceinline__ int tv_res_type(int a, int b) { return a > b ? a : b; }

// The arithmetic helpers are branch-free on purpose: both the f32 and the f64
// candidate are computed and one is selected.  The tree kernels are latency
// chains executed by a single wave, and tag-dependent branches (scalar
// compare + s_cbranch + waitcnt per typed op) were the dominant cost.
__device__ __forceinline__ TV tv_add(TV a, TV b) {
    const int t = tv_res_type(a.t, b.t);
    const double r32 = (double)__fadd_rn((float)a.v, (float)b.v);
    const double r64 = __dadd_rn(a.v, b.v);
    return mk_tv(t == PFRL_TAG_F32 ? r32 : r64, t);
}

__device__ __forceinline__ TV tv_sub(TV a, TV b) {
    const int t = tv_res_type(a.t, b.t);
    const double r32 = (double)__fsub_rn((float)a.v, (float)b.v);
    const double r64 = __dsub_rn(a.v, b.v);
    return mk_tv(t == PFRL_TAG_F32 ? r32 : r64, t);
}

__device__ __forceinline__ TV tv_div(TV a, TV b) {
    const int t = tv_res_type(a.t, b.t);
    if (t == PFRL_TAG_F32) return mk_tv((double)__fdiv_rn((float)a.v, (float)b.v), t);
    return mk_tv(__ddiv_rn(a.v, b.v), t);
}

__device__ __forceinline__ bool tv_lt(TV a, TV b) {
    const bool l32 = (float)a.v < (float)b.v;
    const bool l64 = a.v < b.v;
    return tv_res_type(a.t, b.t) == PFRL_TAG_F32 ? l32 : l64;
}
