// On-chip peaks of the GPU this runs on, for the decoder's roofline: LDS bytes per clock per CU for
// the access shapes the kernels use, and vector-instruction issue per SIMD.  Prints one JSON object.
//   hipcc --offload-arch=gfx950 -O3 -o peaks peaks.hip && ./peaks
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int THREADS = 1024;       // 16 waves per CU, as in the decode kernels
constexpr int ITERS = 2048;

// MODE 0: ds_read_b32, 1: ds_read2_b32 (two dwords 16 bytes apart), 2: ds_read_b64, 3: ds_read_b128, 4: ds_write_b32,
// 5: ds_write_b64, 6: ds_write_b128, 7: ds_write2_b32 (two dwords 4 bytes apart)
// stride: bytes between consecutive lanes (20 = the message-plane record, 4/8/16 = dense)
template <int MODE>
__global__ void __launch_bounds__(THREADS) lds_kernel(float *out, int stride, unsigned long long *cycles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int x = threadIdx.x; x < 40960; x += THREADS) reinterpret_cast<float *>(smem)[x] = (float)x;
    __syncthreads();
    // eight fixed addresses per lane, computed once: the loop is LDS instructions and one add each
    uint32_t ad[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        uint32_t a = ((uint32_t)threadIdx.x * (uint32_t)stride + (uint32_t)u * 20480u) % 131072u;
        if (MODE == 2 || MODE == 5) a &= ~7u;
        if (MODE == 3 || MODE == 6) a &= ~15u;
        ad[u] = a + (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)smem;
    }
    float acc = 0.0f;
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    v2f_ w2 = {1.0f, 2.0f};
    v4f_ w4 = {1.0f, 2.0f, 3.0f, 4.0f};
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < ITERS; it += 8) {
        float v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(v[u][0]) : "v"(ad[u]));
            if (MODE == 1) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:4" : "=v"(*reinterpret_cast<float2 *>(v[u])) : "v"(ad[u]));
            if (MODE == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(*reinterpret_cast<float2 *>(v[u])) : "v"(ad[u]));
            if (MODE == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(*reinterpret_cast<float4 *>(v[u])) : "v"(ad[u]));
            if (MODE == 4) asm volatile("ds_write_b32 %0, %1" : : "v"(ad[u]), "v"(acc) : "memory");
            if (MODE == 5) asm volatile("ds_write_b64 %0, %1" : : "v"(ad[u]), "v"(w2) : "memory");
            if (MODE == 6) asm volatile("ds_write_b128 %0, %1" : : "v"(ad[u]), "v"(w4) : "memory");
            if (MODE == 7) asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:1" : : "v"(ad[u]), "v"(acc), "v"(acc) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE < 4) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float x = v[u][0];
                asm volatile("" : "+v"(x));
                acc += x;
            }
        } else {
            acc += 1.0f;
        }
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * THREADS + threadIdx.x] = acc + reinterpret_cast<float *>(smem)[threadIdx.x];
}

// Independent chains of float adds written as bare instructions (the compiler would pair two chains
// into one packed add): vector-instruction issue rate.  PACKED: v_pk_add_f32, two adds per instruction.
template <int CHAINS, bool PACKED>
__global__ void __launch_bounds__(THREADS) valu_kernel(float *out, float x, unsigned long long *cycles)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[CHAINS];
    f2 xx = {x, x};
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) { a[k].x = (float)(threadIdx.x + k); a[k].y = 1.0f; }
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < CHAINS; ++k) {
                if (PACKED) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(xx));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k].x) : "v"(x));
            }
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) s += a[k].x + a[k].y;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

// Issue cost of single instructions: 4 independent chains per lane, 16 waves per CU.
#define INST_KERNEL(NAME, ASM)                                                                              \
    __global__ void __launch_bounds__(THREADS) NAME(float *out, float x, unsigned long long *cycles)       \
    {                                                                                                       \
        float a[4], b[4];                                                                                   \
        for (int k = 0; k < 4; ++k) { a[k] = (float)(threadIdx.x + k) + 0.5f; b[k] = x + k; }               \
        unsigned long long m = blockIdx.x;                                                                  \
        double da[2] = {1.0 + threadIdx.x, 2.0}, db[2] = {x, x + 1.0};                                      \
        const unsigned long long t0 = clock64();                                                            \
        _Pragma("unroll 1") for (int it = 0; it < ITERS; ++it) {                                            \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                             \
                    asm volatile(ASM : "+v"(a[k]), "+s"(m), "+v"(da[k & 1]) : "v"(b[k]), "v"(x), "v"(db[k & 1]) : "vcc"); \
                }                                                                                           \
        }                                                                                                   \
        __syncthreads();                                                                                    \
        const unsigned long long t1 = clock64();                                                            \
        if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;                                                 \
        out[blockIdx.x * THREADS + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + (float)(m & 1) + (float)(da[0] + da[1]); \
    }
// operands: %0 float chain (in/out), %1 scalar pair (in/out), %2 double chain (in/out), %3 float, %4 float, %5 double
#define INST_LIST(X)                                                             \
    X(i_sub_f32, "v_sub_f32 %0, %0, %3")                                         \
    X(i_mul_f32, "v_mul_f32 %0, %0, %3")                                         \
    X(i_fma_f32, "v_fma_f32 %0, %0, %3, %4")                                     \
    X(i_min_f32, "v_min_f32 %0, %0, %3")                                         \
    X(i_max_f32, "v_max_f32 %0, %0, %3")                                         \
    X(i_min_f32_abs, "v_min_f32 %0, %0, |%3|")                                   \
    X(i_med3_f32, "v_med3_f32 %0, %0, %3, |%4|")                                 \
    X(i_min3_f32, "v_min3_f32 %0, %0, %3, %4")                                   \
    X(i_and_b32, "v_and_b32 %0, %0, %3")                                         \
    X(i_or_b32, "v_or_b32 %0, %0, %3")                                           \
    X(i_xor_b32, "v_xor_b32 %0, %0, %3")                                         \
    X(i_add_u32, "v_add_u32 %0, %0, %3")                                         \
    X(i_sub_u32, "v_sub_u32 %0, %0, %3")                                         \
    X(i_min_u32, "v_min_u32 %0, %0, %3")                                         \
    X(i_max_u32, "v_max_u32 %0, %0, %3")                                         \
    X(i_min_i32, "v_min_i32 %0, %0, %3")                                         \
    X(i_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")                                  \
    X(i_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")                                  \
    X(i_ashrrev_i32, "v_ashrrev_i32 %0, 3, %0")                                  \
    X(i_mov_b32, "v_mov_b32 %0, %3")                                             \
    X(i_alignbit_b32, "v_alignbit_b32 %0, %0, %3, 31")                           \
    X(i_and_or_b32, "v_and_or_b32 %0, %0, %3, %4")                               \
    X(i_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %4")                              \
    X(i_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %4")                            \
    X(i_add3_u32, "v_add3_u32 %0, %0, %3, %4")                                   \
    X(i_bfi_b32, "v_bfi_b32 %0, %3, %0, %4")                                     \
    X(i_bfe_u32, "v_bfe_u32 %0, %0, 3, 8")                                       \
    X(i_xad_u32, "v_xad_u32 %0, %0, %3, %4")                                     \
    X(i_mad_u32_u24, "v_mad_u32_u24 %0, %0, %3, %4")                             \
    X(i_mul_lo_u32, "v_mul_lo_u32 %0, %0, %3")                                   \
    X(i_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %3")                                     \
    X(i_cmp_sgpr, "v_cmp_lt_f32 %1, %0, %3")                                     \
    X(i_cmp_u32_vcc, "v_cmp_lt_u32 vcc, %0, %3")                                 \
    X(i_cndmask_vcc, "v_cndmask_b32 %0, %0, %3, vcc")                            \
    X(i_cndmask_sgpr, "v_cndmask_b32 %0, %0, %3, %1")                            \
    X(i_add_f64, "v_add_f64 %2, %2, %5")                                         \
    X(i_min_f64, "v_min_f64 %2, %2, %5")                                         \
    X(i_fma_f64, "v_fma_f64 %2, %2, %5, %5")                                     \
    X(i_cmp_f64, "v_cmp_lt_f64 vcc, %2, %5")                                     \
    X(i_exp_f32, "v_exp_f32 %0, %0")                                             \
    X(i_log_f32, "v_log_f32 %0, %0")                                             \
    X(i_rcp_f32, "v_rcp_f32 %0, %0")                                             \
    X(i_sdwa_mov, "v_mov_b32_sdwa %0, %3 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3") \
    X(i_pk_add_f32, "v_pk_add_f32 %2, %2, %5")                                   \
    X(i_pk_mul_f32, "v_pk_mul_f32 %2, %2, %5")                                   \
    X(i_pk_fma_f32, "v_pk_fma_f32 %2, %2, %5, %5")                               \
    X(i_mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0")                                  \
    X(i_bcnt, "v_bcnt_u32_b32 %0, %0, %3")                                       \
    X(i_readlane_like_perm, "v_perm_b32 %0, %0, %3, %4")

#define X(NAME, ASM) INST_KERNEL(NAME, ASM)
INST_LIST(X)
#undef X

typedef void (*inst_fn)(float *, float, unsigned long long *);
static double inst_cost(inst_fn fn, int cus, float *out, unsigned long long *cyc, unsigned long long *h)
{
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(fn, dim3(cus), dim3(THREADS), 0, 0, out, 1.5f, cyc);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h, cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost));
    double s = 0;
    for (int k = 0; k < cus; ++k) s += (double)h[k];
    return s / cus / (4.0 * ITERS * 16.0);
}

static double mean_cycles(const unsigned long long *h, int n)
{
    double s = 0;
    for (int k = 0; k < n; ++k) s += (double)h[k];
    return s / n;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out;
    unsigned long long *cyc, *h = (unsigned long long *)malloc(sizeof(unsigned long long) * cus);
    CHECK(hipMalloc(&out, sizeof(float) * THREADS * cus));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * cus));
    const size_t lds = 160 * 1024;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"lds\": [", prop.gcnArchName, cus, prop.clockRate / 1000);
    struct Case { int mode; int stride; const char *name; int bytes; } cases[] = {
        {0, 4, "ds_read_b32 dense", 4}, {0, 20, "ds_read_b32 stride 20", 4}, {1, 20, "ds_read2_b32 stride 20", 8},
        {2, 8, "ds_read_b64 dense", 8}, {2, 24, "ds_read_b64 stride 24", 8}, {3, 16, "ds_read_b128 dense", 16},
        {3, 32, "ds_read_b128 stride 32", 16}, {4, 4, "ds_write_b32 dense", 4}, {4, 20, "ds_write_b32 stride 20", 4},
        {5, 8, "ds_write_b64 dense", 8}, {6, 16, "ds_write_b128 dense", 16}, {7, 20, "ds_write2_b32 stride 20", 8},
        {0, 16, "ds_read_b32 stride 16", 4}, {4, 16, "ds_write_b32 stride 16", 4}};
    bool first = true;
    for (const Case &c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (c.mode) {
            case 0: hipLaunchKernelGGL(lds_kernel<0>, dim3(cus), dim3(THREADS), lds, 0, out, c.stride, cyc); break;
            case 1: hipLaunchKernelGGL(lds_kernel<1>, dim3(cus), dim3(THREADS), lds, 0, out, c.stride, cyc); break;
            case 2: hipLaunchKernelGGL(lds_kernel<2>, dim3(cus), dim3(THREADS), lds, 0, out, c.stride, cyc); break;
            case 3: hipLaunchKernelGGL(lds_kernel<3>, dim3(cus), dim3(THREADS), lds, 0, out, c.stride, cyc); break;
            case 4: hipLaunchKernelGGL(lds_kernel<4>, dim3(cus), dim3(THREADS), lds, 0, out, c.stride, cyc); break;
            case 5: hipLaunchKernelGGL(lds_kernel<5>, dim3(cus), dim3(THREADS), lds, 0, out, c.stride, cyc); break;
            case 6: hipLaunchKernelGGL(lds_kernel<6>, dim3(cus), dim3(THREADS), lds, 0, out, c.stride, cyc); break;
            default: hipLaunchKernelGGL(lds_kernel<7>, dim3(cus), dim3(THREADS), lds, 0, out, c.stride, cyc); break;
            }
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h, cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost));
        const double clk = mean_cycles(h, cus);
        const double bytes = (double)THREADS * ITERS * c.bytes;
        printf("%s{\"access\": \"%s\", \"bytes_per_clk_per_cu\": %.1f}", first ? "" : ", ", c.name, bytes / clk);
        first = false;
    }
    printf("], \"valu\": [");
    first = true;
    for (int cfg = 0; cfg < 6; ++cfg) {
        const int chains = cfg < 4 ? (1 << cfg) : (cfg == 4 ? 2 : 8);
        const bool packed = cfg >= 4;
        for (int rep = 0; rep < 2; ++rep) {
            switch (cfg) {
            case 0: hipLaunchKernelGGL((valu_kernel<1, false>), dim3(cus), dim3(THREADS), 0, 0, out, 1.5f, cyc); break;
            case 1: hipLaunchKernelGGL((valu_kernel<2, false>), dim3(cus), dim3(THREADS), 0, 0, out, 1.5f, cyc); break;
            case 2: hipLaunchKernelGGL((valu_kernel<4, false>), dim3(cus), dim3(THREADS), 0, 0, out, 1.5f, cyc); break;
            case 3: hipLaunchKernelGGL((valu_kernel<8, false>), dim3(cus), dim3(THREADS), 0, 0, out, 1.5f, cyc); break;
            case 4: hipLaunchKernelGGL((valu_kernel<2, true>), dim3(cus), dim3(THREADS), 0, 0, out, 1.5f, cyc); break;
            default: hipLaunchKernelGGL((valu_kernel<8, true>), dim3(cus), dim3(THREADS), 0, 0, out, 1.5f, cyc); break;
            }
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h, cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost));
        const double clk = mean_cycles(h, cus);
        // wave-instructions issued per SIMD: 16 waves / 4 SIMDs, ITERS * 4 * chains instructions each
        const double per_simd = 4.0 * ITERS * 4.0 * chains;
        printf("%s{\"instruction\": \"%s\", \"independent_chains\": %d, \"clocks_per_wave_instruction_per_simd\": %.2f}",
               first ? "" : ", ", packed ? "v_pk_add_f32" : "v_add_f32", chains, clk / per_simd);
        first = false;
    }
    printf("], \"instruction_clocks_per_simd\": {");
    {
        bool f1 = true;
#define X(NAME, ASM) printf("%s\"%s\": %.2f", f1 ? "" : ", ", ASM, inst_cost(NAME, cus, out, cyc, h)); f1 = false;
        INST_LIST(X)
#undef X
    }
    printf("}}\n");
    return 0;
}
