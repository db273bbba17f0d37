// Shared host/device helpers for the gfx950 replay data path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pfrl_amd.h"

extern "C" void pfrl_set_error(const char *msg);

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

__device__ __forceinline__ int tv_res_type(int a, int b) {
    if (a == PFRL_TAG_F64 || b == PFRL_TAG_F64) return PFRL_TAG_F64;
    if (a == PFRL_TAG_F32 || b == PFRL_TAG_F32) return PFRL_TAG_F32;
    return PFRL_TAG_PY;
}

__device__ __forceinline__ TV tv_add(TV a, TV b) {
    int t = tv_res_type(a.t, b.t);
    if (t == PFRL_TAG_F32) return mk_tv((double)__fadd_rn((float)a.v, (float)b.v), t);
    return mk_tv(__dadd_rn(a.v, b.v), t);
}

__device__ __forceinline__ TV tv_sub(TV a, TV b) {
    int t = tv_res_type(a.t, b.t);
    if (t == PFRL_TAG_F32) return mk_tv((double)__fsub_rn((float)a.v, (float)b.v), t);
    return mk_tv(__dsub_rn(a.v, b.v), t);
}

__device__ __forceinline__ TV tv_div(TV a, TV b) {
    int t = tv_res_type(a.t, b.t);
    if (t == PFRL_TAG_F32) return mk_tv((double)__fdiv_rn((float)a.v, (float)b.v), t);
    return mk_tv(__ddiv_rn(a.v, b.v), t);
}

__device__ __forceinline__ bool tv_lt(TV a, TV b) {
    if (tv_res_type(a.t, b.t) == PFRL_TAG_F32) return (float)a.v < (float)b.v;
    return a.v < b.v;
}
