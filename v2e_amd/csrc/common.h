// common.h -- error plumbing shared by the gfx950 translation units of libv2e_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#include "../../include/v2e_amd.h"

void v2e_set_error(const char *fmt, ...);

#define V2E_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            v2e_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return V2E_EHIP;                                                               \
        }                                                                                  \
    } while (0)

#define V2E_REQUIRE(cond, msg)                                       \
    do {                                                             \
        if (!(cond)) {                                               \
            v2e_set_error("%s:%d %s", __FILE__, __LINE__, msg);      \
            return V2E_EINVAL;                                       \
        }                                                            \
    } while (0)

static inline int v2e_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// csdvs.hip: one frame's surround (diffuser) update enqueued on `s` without a host step (emu.hip's per-frame run loop)
int v2e_csdvs_enqueue_frame(const void *p_plane, void *h_plane, void *h_scratch, int H, int W, int f64, double alpha_p, double alpha_h,
                            int num_steps, double thr, unsigned long long *slots, int *steps_taken_dev, hipStream_t s);
