// Cost of counter-based generators and of Box-Muller on the GPU this runs on: SIMD clocks per
// wave-call (one call = 4 outputs per lane), 16 waves per CU as in the decode kernels.
//   hipcc --offload-arch=gfx950 -O3 -o rng rng.hip && ./rng
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int THREADS = 1024;
constexpr int CALLS = 512;

template <int ROUNDS>
__device__ __forceinline__ uint4 philox(uint4 c, uint2 k)
{
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// The same rounds with each product taken as one 64-bit multiply (v_mad_u64_u32 gives both halves).
template <int ROUNDS>
__device__ __forceinline__ uint4 philox_wide(uint4 c, uint2 k)
{
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        c = make_uint4((uint32_t)(p1 >> 32) ^ c.y ^ k.x, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k.y, (uint32_t)p0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

__device__ __forceinline__ uint32_t rotl(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }

template <int ROUNDS>
__device__ __forceinline__ uint4 threefry(uint4 c, uint4 k)
{
    const int R0[8] = {10, 11, 13, 23, 6, 17, 25, 18}, R1[8] = {26, 21, 27, 5, 20, 11, 10, 20};
    uint32_t ks[5] = {k.x, k.y, k.z, k.w, 0x1BD11BDAu ^ k.x ^ k.y ^ k.z ^ k.w};
    uint32_t x0 = c.x + ks[0], x1 = c.y + ks[1], x2 = c.z + ks[2], x3 = c.w + ks[3];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r % 2 == 0) {
            x0 += x1; x1 = rotl(x1, R0[r % 8]); x1 ^= x0;
            x2 += x3; x3 = rotl(x3, R1[r % 8]); x3 ^= x2;
        } else {
            x0 += x3; x3 = rotl(x3, R0[r % 8]); x3 ^= x0;
            x2 += x1; x1 = rotl(x1, R1[r % 8]); x1 ^= x2;
        }
        if (r % 4 == 3) {
            const int s = r / 4 + 1;
            x0 += ks[s % 5]; x1 += ks[(s + 1) % 5]; x2 += ks[(s + 2) % 5]; x3 += ks[(s + 3) % 5] + (uint32_t)s;
        }
    }
    return make_uint4(x0, x1, x2, x3);
}

__device__ __forceinline__ float u01(uint32_t x) { return __builtin_fmaf((float)(x >> 8), 1.0f / 16777216.0f, 0.5f / 16777216.0f); }
__device__ __forceinline__ float4 box_muller(uint4 r)
{
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r.x)));
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r.z)));
    const float a0 = u01(r.y), a1 = u01(r.w);
    return make_float4(r0 * __builtin_amdgcn_cosf(a0), r0 * __builtin_amdgcn_sinf(a0),
                       r1 * __builtin_amdgcn_cosf(a1), r1 * __builtin_amdgcn_sinf(a1));
}

// MODE 0 philox-10, 1 philox-7, 2 threefry-20, 3 threefry-12, 4 box-muller only, 5 philox-10 + box-muller, 6 threefry-20 + box-muller,
// 7 philox-10 on 64-bit products, 8 the same + box-muller
template <int MODE>
__global__ void __launch_bounds__(THREADS) rng_kernel(float *out, unsigned long long *cycles, uint32_t seed)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    float facc = 0.f;
    __syncthreads();
    const unsigned long long t0 = clock64();
#pragma unroll 2
    for (int q = 0; q < CALLS; ++q) {
        const uint4 c = make_uint4(threadIdx.x + q * THREADS, blockIdx.x, 0, 3);
        uint4 r;
        if (MODE == 0 || MODE == 5) r = philox<10>(c, make_uint2(seed, 7));
        if (MODE == 1) r = philox<7>(c, make_uint2(seed, 7));
        if (MODE == 2 || MODE == 6) r = threefry<20>(c, make_uint4(seed, 7, 0, 0));
        if (MODE == 3) r = threefry<12>(c, make_uint4(seed, 7, 0, 0));
        if (MODE == 7 || MODE == 8) r = philox_wide<10>(c, make_uint2(seed, 7));
        if (MODE == 4) r = make_uint4(c.x * 2654435761u, c.x ^ 0x9E3779B9u, c.x + 0x12345u, c.x * 3u);
        if (MODE >= 4 && MODE != 7) {
            const float4 z = box_muller(r);
            facc += (z.x + z.y) + (z.z + z.w);
        } else {
            acc.x ^= r.x; acc.y ^= r.y; acc.z ^= r.z; acc.w ^= r.w;
        }
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * THREADS + threadIdx.x] = facc + (float)(acc.x ^ acc.y ^ acc.z ^ acc.w);
}

int main()
{
    int dev = 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int blocks = prop.multiProcessorCount;
    float *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, sizeof(float) * THREADS * blocks));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks));
    const char *names[9] = {"philox4x32_10", "philox4x32_7", "threefry4x32_20", "threefry4x32_12", "box_muller_4",
                            "philox4x32_10+box_muller", "threefry4x32_20+box_muller", "philox4x32_10_mad64",
                            "philox4x32_10_mad64+box_muller"};
    printf("{");
    for (int mode = 0; mode < 9; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
            case 0: hipLaunchKernelGGL(rng_kernel<0>, dim3(blocks), dim3(THREADS), 0, 0, out, cyc, 1u); break;
            case 1: hipLaunchKernelGGL(rng_kernel<1>, dim3(blocks), dim3(THREADS), 0, 0, out, cyc, 1u); break;
            case 2: hipLaunchKernelGGL(rng_kernel<2>, dim3(blocks), dim3(THREADS), 0, 0, out, cyc, 1u); break;
            case 3: hipLaunchKernelGGL(rng_kernel<3>, dim3(blocks), dim3(THREADS), 0, 0, out, cyc, 1u); break;
            case 4: hipLaunchKernelGGL(rng_kernel<4>, dim3(blocks), dim3(THREADS), 0, 0, out, cyc, 1u); break;
            case 5: hipLaunchKernelGGL(rng_kernel<5>, dim3(blocks), dim3(THREADS), 0, 0, out, cyc, 1u); break;
            case 6: hipLaunchKernelGGL(rng_kernel<6>, dim3(blocks), dim3(THREADS), 0, 0, out, cyc, 1u); break;
            case 7: hipLaunchKernelGGL(rng_kernel<7>, dim3(blocks), dim3(THREADS), 0, 0, out, cyc, 1u); break;
            default: hipLaunchKernelGGL(rng_kernel<8>, dim3(blocks), dim3(THREADS), 0, 0, out, cyc, 1u); break;
            }
            CHECK(hipDeviceSynchronize());
        }
        unsigned long long h[1024];
        CHECK(hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
        double s = 0;
        for (int b = 0; b < blocks; ++b) s += (double)h[b];
        // 16 waves share 4 SIMDs: clocks of one SIMD per wave-call
        printf("%s\"%s\": %.1f", mode ? ", " : "", names[mode], s / blocks / CALLS / 4.0);
    }
    printf(", \"unit\": \"SIMD clocks per wave-call (4 outputs per lane), 16 waves per CU\"}\n");
    return 0;
}
