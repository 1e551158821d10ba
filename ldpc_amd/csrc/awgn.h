// BPSK/AWGN channel of the all-zero codeword, generated on the device (modulate + addAWGN,
// ldpc.py:51-66): counter-based noise, so a sample depends only on (seed, SNR-point stream,
// global frame number, position) and never on how frames are split over launches, workgroups
// or GPUs.  Included by ldpc_hip.hip; used by the stand-alone channel kernel and by the decode
// kernels that generate their own input.
#pragma once

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 64-bit product per multiplier: v_mad_u64_u32 returns both halves (220 against 286 clocks
        // per wave-call for a high and a low multiply each)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// (x >> 8) + 0.5 scaled by 2^-24, as one fused multiply-add: scaling by a power of two commutes with
// rounding, so this is bit for bit ((float)(x >> 8) + 0.5f) * 2^-24 -- in (0, 1], never 0
__device__ __forceinline__ float u01(uint32_t x)
{
    return __builtin_fmaf((float)(x >> 8), 1.0f / 16777216.0f, 0.5f / 16777216.0f);
}

// Box-Muller on the hardware transcendental units (log2, sqrt, and sin/cos that take their
// argument in revolutions): absolute error ~1e-6, far below the Monte-Carlo resolution,
// at a third of the instructions of the library routines
__device__ __forceinline__ float4 box_muller4(uint4 r)
{
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r.x)));   // -2 ln u = -2 ln2 log2 u
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r.z)));
    const float a0 = u01(r.y), a1 = u01(r.w);
    return make_float4(r0 * __builtin_amdgcn_cosf(a0), r0 * __builtin_amdgcn_sinf(a0),
                       r1 * __builtin_amdgcn_cosf(a1), r1 * __builtin_amdgcn_sinf(a1));
}

// Four standard normals for positions 4*q .. 4*q+3 of frame `frame`.
__device__ __forceinline__ float4 normal4(uint64_t seed, uint64_t frame, uint32_t q, uint32_t stream)
{
    return box_muller4(philox4x32_10(make_uint4(q, (uint32_t)frame, (uint32_t)(frame >> 32), stream),
                                     make_uint2((uint32_t)seed, (uint32_t)(seed >> 32))));
}

// The same for NC values of q at once, round by round: the calls' multiplier chains are independent
// and every round key is used NC times where it is made.
template <int NC>
__device__ __forceinline__ void normal4_calls(uint32_t seed_lo, uint32_t seed_hi, uint64_t frame,
                                              const uint32_t (&q)[NC], uint32_t stream, float4 (&z)[NC])
{
    uint4 c[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) c[i] = make_uint4(q[i], (uint32_t)frame, (uint32_t)(frame >> 32), stream);
    uint2 k = make_uint2(seed_lo, seed_hi);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c[i].x, p1 = (uint64_t)0xCD9E8D57u * c[i].z;
            c[i] = make_uint4((uint32_t)(p1 >> 32) ^ c[i].y ^ k.x, (uint32_t)p1,
                              (uint32_t)(p0 >> 32) ^ c[i].w ^ k.y, (uint32_t)p0);
        }
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) z[i] = box_muller4(c[i]);
}


// sigma_hat (the sample RMS of a frame's noise, ldpc.py:58) is sigma * sqrt(sum z^2 / n).  The sum is
// kept in 2^-19 fixed point (z^2 < 36 for every z this generator can produce): integer addition is
// associative, so the result does not depend on which thread, wave or kernel added which sample, and
// eight samples fit a 32-bit partial sum.  Rounding to nearest leaves no bias; the resolution is far
// below the float32 result's own.
__device__ __forceinline__ uint32_t awgn_sq_fixed(float z)
{
    return (uint32_t)__float2uint_rn((z * z) * 524288.0f);
}
__device__ __forceinline__ float awgn_sigma_hat(unsigned long long sumsq_fixed, float sigma, int n)
{
    return (float)((double)sigma * sqrt((double)sumsq_fixed * (1.0 / 524288.0) / (double)n));
}

// Sum over each row of 16 lanes, left in the row's last lane (lanes 15, 31, 47, 63).
__device__ __forceinline__ uint32_t awgn_row_sum16(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   // row_shr:8
    return x;
}
