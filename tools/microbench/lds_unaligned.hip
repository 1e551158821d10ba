// Wide LDS accesses at addresses that are only 4-byte aligned (the message-plane kernel's records are 20 bytes:
// a column's four messages are 16 contiguous bytes at offset 4): do they work, what do they return, how fast?
// Prints one JSON object.   hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int THREADS = 1024, ITERS = 2048;

// MODE 0: two ds_read2_b32 (words 1,2 and 3,4 of a 20-byte record: today's variable pass)
//      1: one ds_read_b128 at record + 4        2: two ds_read_b64 at record + 4 and record + 12
//      3: ds_read_b96 at record + 4 plus ds_read_b32 at record + 16
//      4: ds_write_b64 at record + 4 (two messages of one record)    5: two ds_write_b32 (the same words)
template <int MODE>
__global__ void __launch_bounds__(THREADS) k(float *out, unsigned long long *cycles, int *wrong)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *w = reinterpret_cast<float *>(smem);
    for (int x = threadIdx.x; x < 40960; x += THREADS) w[x] = (float)x;
    __syncthreads();
    uint32_t ad[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ad[u] = ((uint32_t)threadIdx.x * 20u + (uint32_t)u * 20480u) % 131072u / 20u * 20u + 4u;
    typedef float v2 __attribute__((ext_vector_type(2)));
    typedef float v3 __attribute__((ext_vector_type(3)));
    typedef float v4 __attribute__((ext_vector_type(4)));
    float acc = 0.0f;
    int bad = 0;
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < ITERS; it += 8) {
        float v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v2 a, b; v4 q; v3 t; float s;
            if (MODE == 0) {
                asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(a) : "v"(ad[u]));
                asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(b) : "v"(ad[u]));
                v[u][0] = a.x; v[u][1] = a.y; v[u][2] = b.x; v[u][3] = b.y;
            } else if (MODE == 1) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(ad[u]));
                v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
            } else if (MODE == 2) {
                asm volatile("ds_read_b64 %0, %1" : "=v"(a) : "v"(ad[u]));
                asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(b) : "v"(ad[u]));
                v[u][0] = a.x; v[u][1] = a.y; v[u][2] = b.x; v[u][3] = b.y;
            } else if (MODE == 3) {
                asm volatile("ds_read_b96 %0, %1" : "=v"(t) : "v"(ad[u]));
                asm volatile("ds_read_b32 %0, %1 offset:12" : "=v"(s) : "v"(ad[u]));
                v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = s;
            } else if (MODE == 4) {
                const v2 d = {acc, acc + 1.0f};
                asm volatile("ds_write_b64 %0, %1" : : "v"(ad[u]), "v"(d) : "memory");
            } else {
                asm volatile("ds_write_b32 %0, %1" : : "v"(ad[u]), "v"(acc) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:4" : : "v"(ad[u]), "v"(acc) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE < 4) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (it == 0)
                    for (int c = 0; c < 4; ++c) bad += v[u][c] != (float)(ad[u] / 4 + c);
                float x = v[u][0] + v[u][3];
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
    if (bad) atomicAdd(wrong, bad);
    if (MODE == 4 && threadIdx.x == 0) {                 // what an unaligned 8-byte write left behind
        const float a0 = w[ad[0] / 4], a1 = w[ad[0] / 4 + 1];
        if (a1 != a0 + 1.0f) atomicAdd(wrong, 1);
    }
    out[blockIdx.x * THREADS + threadIdx.x] = acc + w[threadIdx.x];
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out; unsigned long long *cyc; int *wrong;
    unsigned long long *h = (unsigned long long *)malloc(sizeof(unsigned long long) * cus);
    CHECK(hipMalloc(&out, sizeof(float) * THREADS * cus));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * cus));
    CHECK(hipMalloc(&wrong, sizeof(int)));
    const size_t lds = 160 * 1024;
    const char *names[] = {"2 x ds_read2_b32 (16 bytes at record + 4)", "ds_read_b128 at record + 4 (4-byte aligned)",
                           "2 x ds_read_b64 at record + 4, + 12", "ds_read_b96 + ds_read_b32", "ds_write_b64 at record + 4",
                           "2 x ds_write_b32"};
    const int bytes[] = {16, 16, 16, 16, 8, 8};
    printf("{\"device\": \"%s\", \"record_stride\": 20, \"cases\": [", prop.gcnArchName);
    for (int m = 0; m < 6; ++m) {
        int w = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(wrong, 0, sizeof(int)));
            switch (m) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(cus), dim3(THREADS), lds, 0, out, cyc, wrong); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(cus), dim3(THREADS), lds, 0, out, cyc, wrong); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(cus), dim3(THREADS), lds, 0, out, cyc, wrong); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(cus), dim3(THREADS), lds, 0, out, cyc, wrong); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(cus), dim3(THREADS), lds, 0, out, cyc, wrong); break;
            default: hipLaunchKernelGGL(k<5>, dim3(cus), dim3(THREADS), lds, 0, out, cyc, wrong); break;
            }
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("%s{\"access\": \"%s\", \"error\": \"%s\"}", m ? ", " : "", names[m], hipGetErrorString(e)); w = -1; break; }
        }
        if (w < 0) break;
        CHECK(hipMemcpy(h, cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&w, wrong, sizeof(int), hipMemcpyDeviceToHost));
        double clk = 0;
        for (int c = 0; c < cus; ++c) clk += (double)h[c];
        clk /= cus;
        printf("%s{\"access\": \"%s\", \"bytes_per_clk_per_cu\": %.1f, \"wrong_values\": %d}", m ? ", " : "", names[m],
               (double)THREADS * ITERS * bytes[m] / clk, w);
    }
    printf("]}\n");
    return 0;
}
