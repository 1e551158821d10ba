// Arguments of one decode launch, shared by every decode kernel of the library and by the
// translation units that hold them.
#pragma once
#include <stdint.h>

enum { GEN_MAX_POINTS = 8, GEN_MAX_BATCH = 8192 };   // sweep points, and frames in all, that one launch may serve

struct DecodeArgs {
    const void *llr;       // [frames][n] float32, or float64 if llr_f64
    int llr_f64;
    float llr_scale;
    int64_t n_frames;
    int max_iters;
    int sign_mode;
    uint8_t *hard;         // [frames][ceil(n/8)] or null
    int32_t *iters;        // [frames] or null
    uint8_t *conv;         // [frames] or null
    int32_t *ones;         // [frames] or null
    void *soft;            // [frames][n] T or null
    unsigned char *chain;  // null, or the persistent check-node state of a chained decoder:
                           // frames are decoded one after another (grid of 1 workgroup) and the
                           // check rows' last outgoing messages carry over from frame to frame
    float ms_alpha, ms_beta;   // min-sum magnitude correction: |c2v| = max(alpha*min - beta, 0); 1, 0 = the reference's plain rule
    int precheck;          // rows left unsatisfied up to which an iteration starts with a syndrome-only pass;
                           // -1 = each kernel's own default, 0 = never (LDPC_HIP_PRECHECK_T)
    int *work;             // null, or a zeroed counter: workgroups claim frames gridDim.x + atomicAdd(work, 1)
                           // one at a time instead of walking a fixed stride (message-plane kernel)
    // Kernels that make their own channel values (the GEN instantiations; llr is then unused):
    // frame f of the launch is frame gen_first_frame + f of noise stream (gen_seed, gen_stream),
    // y = -1 + gen_sigma * z (awgn.h), and the per-frame channel statistics go to
    unsigned long long gen_seed = 0, gen_first_frame = 0;
    uint32_t gen_stream = 0;
    float gen_sigma = 0.0f;
    // Several noise streams in one launch (a small sweep: the points' frames decode side by side instead of
    // one launch after the other).  gen_per_point > 0: frame f of the launch belongs to point p = f / gen_per_point
    // and is frame gen_first_frame + f % gen_per_point of stream gen_stream + p, made with gen_point_sigma[p]
    // and scaled by gen_point_scale[p] (gen_sigma and llr_scale are then unused).
    uint32_t gen_per_point = 0;
    float gen_point_sigma[GEN_MAX_POINTS] = {}, gen_point_scale[GEN_MAX_POINTS] = {};
    int32_t *gen_uncoded = nullptr;              // [frames] channel hard-decision errors, or null
    unsigned long long *gen_sumsq = nullptr;     // [frames] sum of z^2 in 2^-19 fixed point (awgn.h), or null
    // null, or [2] summed over the launch: full check passes, syndrome-only passes (measurement only)
    unsigned long long *pass_counts = nullptr;
};
