// libldpc_hip.so -- LDPC code evaluation for AMD MI355X (gfx950, CDNA4).
//
// One workgroup decodes one frame (or a few short frames) entirely on chip:
//   * posteriors of all n variable nodes live in LDS,
//   * every check row keeps its outgoing messages in compressed form
//     {min1, min2, argmin, sign bits} (plain min-sum sends only two distinct
//     magnitudes per row, ldpc.py:190-191) -- in registers of the thread that
//     owns the row and, for the variable update, in a small LDS array,
//   * the parity syndrome of the previous posterior is a by-product of the
//     check pass (the hard decisions of the values it gathers anyway).
// Quasi-cyclic structure is used for addressing: thread i of block-row b reads
// posterior (i + s) mod Z of block-column j, so a wavefront touches 64
// consecutive LDS words per circulant shift.
//
// Floating-point evaluation order follows the reference decoder exactly
// (ldpc.py:150-163, 174-202, 295-324); build with -ffp-contract=off.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <new>
#include <thread>
#include <mutex>
#include <condition_variable>

#include "../../include/ldpc_hip.h"

// Register budget of the table-driven float32 kernel: 8 waves per SIMD (64 registers, a few spilled)
// lets two 1024-thread workgroups share a CU, which hides the latency of its scalar table loads
// (measured on a weight-3 mutation of NearEarth: 15.8 -> 20.2 Gbit/s).  0 = the compiler's choice.
#ifndef QC_GENERIC_WAVES
#define QC_GENERIC_WAVES 8
#endif

// ---------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

// Environment switches that force a slower kernel or a different schedule exist for the parity
// tests and for A/B timing only.  They are honoured only when LDPC_HIP_TEST_HOOKS=1 was set
// when the library was loaded; otherwise no decode call reads the environment.
static const bool g_test_hooks = [] { const char *e = getenv("LDPC_HIP_TEST_HOOKS"); return e && e[0] == '1'; }();
static const char *hook(const char *name) { return g_test_hooks ? getenv(name) : nullptr; }

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// for the library's other source file (comm.hip); not part of the C ABI
extern "C" __attribute__((visibility("hidden"))) void ldpc_set_error_(const char *text)
{
    snprintf(g_err, sizeof(g_err), "%s", text);
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess)                                                      \
            return fail(LDPC_E_DEVICE, "%s failed: %s (%s:%d)", #expr,            \
                        hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

// ---------------------------------------------------------------- device tables
struct QcTables {
    int Z, Zp, mb, nb, m, n;
    int n_ent, n_grp;        // entries of row_ent (= of col_ent) and of col_grp
    const int32_t *row_ptr;  // [mb+1] entry ranges of each block-row
    const int2 *row_ent;     // {j*Z, s}: row i has an edge to column j*Z + (i+s) mod Z
    const int32_t *col_ptr;  // [nb+1] group ranges of each block-column
    const int4 *col_grp;     // {b, w, first entry, -}: one non-empty block (b, j), ascending b
    const int2 *col_ent;     // {s, e}: shift and edge number inside row list of b; s descending
};

// LDS copy of the tables: col_grp | row_ent | col_ent | row_ptr | col_ptr (16-byte items first)
__host__ __device__ inline size_t qc_table_bytes(const QcTables &tb)
{
    return (size_t)tb.n_grp * 16 + (size_t)tb.n_ent * 16 + (((size_t)(tb.mb + 1 + tb.nb + 1) * 4 + 15) & ~(size_t)15);
}

struct CsrTables {
    int m, n, E;
    const int32_t *row_ptr;  // [m+1]
    const int32_t *col_idx;  // [E]
    const int32_t *col_ptr;  // [n+1]
    const int32_t *col_row;  // [E] rows of each column, ascending
    const int32_t *col_e;    // [E] edge number of that entry inside its row
};

#include "decode_args.h"
#include "qc_heavy.h"

// ---------------------------------------------------------------- small helpers
template <typename T> struct Vec2;
template <> struct Vec2<float> { typedef float2 type; };
template <> struct Vec2<double> { typedef double2 type; };

// One sign bit per edge of a check row, in a word of SB = 32, 64 or 128 bits (the widest: the state of a long gym
// episode, where every action leaves a circulant of up to 7 hot bits, ldpc_env.py:293-317: 16 x 7 = 112 edges).
template <int SB> struct SignBits;
template <> struct SignBits<32> {
    typedef uint32_t type;
    struct __attribute__((aligned(8))) Meta { uint32_t signs; uint32_t idx; };
};
template <> struct SignBits<64> {
    typedef uint64_t type;
    struct __attribute__((aligned(16))) Meta { uint64_t signs; uint32_t idx; uint32_t pad; };
};
template <> struct SignBits<128> {
    typedef unsigned __int128 type;
    struct __attribute__((aligned(16))) Meta { unsigned __int128 signs; uint32_t idx; uint32_t pad[3]; };
};

__device__ __forceinline__ float with_sign(float mag, uint32_t neg)
{
    return __uint_as_float(__float_as_uint(mag) | (neg << 31));
}
__device__ __forceinline__ double with_sign(double mag, uint32_t neg)
{
    return __hiloint2double(__double2hiint(mag) | (int)(neg << 31), __double2loint(mag));
}
// the same with the sign already in bit 31
__device__ __forceinline__ float with_sign_bit(float mag, uint32_t bit31)
{
    return __uint_as_float(__float_as_uint(mag) | bit31);
}
__device__ __forceinline__ double with_sign_bit(double mag, uint32_t bit31)
{
    return __hiloint2double(__double2hiint(mag) | (int)bit31, __double2loint(mag));
}
__device__ __forceinline__ uint32_t sign_word(float v) { return __float_as_uint(v); }
__device__ __forceinline__ uint32_t sign_word(double v) { return (uint32_t)__double2hiint(v); }
// channel value (frame-major) as the arithmetic type T
template <typename T>
__device__ __forceinline__ T load_llr(const void *llr, int is_f64, int64_t idx)
{
    return is_f64 ? T(reinterpret_cast<const double *>(llr)[idx])
                  : T(reinterpret_cast<const float *>(llr)[idx]);
}
// CPT channel values of one frame, element base + q*stride (q = 0..CPT-1).  The float32 /
// float64 decision is taken ONCE around the unrolled loop, so the loads are issued back to
// back; a per-element decision makes the compiler branch and wait for every single load.
template <typename T, int CPT>
__device__ __forceinline__ void load_llr_strided(const void *llr, int is_f64, int64_t base, int stride, T (&out)[CPT])
{
    if (is_f64) {
        const double *p = reinterpret_cast<const double *>(llr) + base;
#pragma unroll
        for (int q = 0; q < CPT; ++q) out[q] = T(p[(int64_t)q * stride]);
    } else {
        const float *p = reinterpret_cast<const float *>(llr) + base;
#pragma unroll
        for (int q = 0; q < CPT; ++q) out[q] = T(p[(int64_t)q * stride]);
    }
}
__device__ __forceinline__ float t_inf(float) { return __uint_as_float(0x7f800000u); }
__device__ __forceinline__ double t_inf(double) { return __hiloint2double(0x7ff00000, 0); }
__device__ __forceinline__ float t_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double t_abs(double x) { return fabs(x); }

// normalised / offset min-sum magnitude (applied once per check row to min1 and min2);
// alpha = 1, beta = 0 leaves every value unchanged bit for bit
__device__ __forceinline__ float ms_correct(float m, float alpha, float beta) { return fmaxf(alpha * m - beta, 0.0f); }
__device__ __forceinline__ double ms_correct(double m, float alpha, float beta) { return fmax((double)alpha * m - (double)beta, 0.0); }

// second-smallest update: new min2 given min1 <= min2 and a candidate a
__device__ __forceinline__ float t_mid(float min1, float min2, float a)
{
    return __builtin_amdgcn_fmed3f(min1, min2, a);
}
__device__ __forceinline__ double t_mid(double min1, double min2, double a)
{
    return fmin(min2, fmax(min1, a));
}
__device__ __forceinline__ float t_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double t_min(double a, double b) { return fmin(a, b); }

template <typename ST> __device__ __forceinline__ uint32_t parity_of(ST x);
template <> __device__ __forceinline__ uint32_t parity_of<uint32_t>(uint32_t x) { return __popc(x) & 1u; }
template <> __device__ __forceinline__ uint32_t parity_of<uint64_t>(uint64_t x) { return __popcll(x) & 1u; }
template <> __device__ __forceinline__ uint32_t parity_of<unsigned __int128>(unsigned __int128 x)
{
    return (__popcll((uint64_t)x) + __popcll((uint64_t)(x >> 64))) & 1u;
}

// message a check row sends along its edge e, rebuilt from the row's compressed state
template <typename T, int S64>
__device__ __forceinline__ T rebuild(const typename Vec2<T>::type mm,
                                     const typename SignBits<S64>::Meta mt, int e)
{
    const T mag = (mt.idx == (uint32_t)e) ? mm.y : mm.x;
    return with_sign(mag, (uint32_t)((mt.signs >> e) & 1u));
}

// Compressed state of a check row in the table-driven kernel below: { min1, min2 } and, next to it, NW = SB / 32
// words of sign bits (bit 31 - k of word w: the message on edge 32 w + k is negative; first edge highest, so that a
// shift by k puts an edge's bit where a float keeps its sign) followed by the arg-min edge.
// The variable pass reads, per edge, the one sign word that holds its bit, the arg-min and the two magnitudes:
// 16 bytes, whatever the row's degree.  MW = NW + 1 words per row: odd for SB = 64 and 128, so that the rows
// neighbouring lanes read lie on different LDS banks (SB = 32: the two words are read as one 8-byte value).
template <int SB> struct RowMeta { enum { NW = SB / 32, MW = NW + 1 }; };

template <typename T, int SB>
__device__ __forceinline__ T rebuild_row(const typename Vec2<T>::type *mins, const uint32_t *meta, int slot, int e)
{
    enum { NW = RowMeta<SB>::NW, MW = RowMeta<SB>::MW };
    uint32_t word, idx;
    if (SB == 32) {
        const uint2 wi = *reinterpret_cast<const uint2 *>(meta + (size_t)slot * MW);
        word = wi.x; idx = wi.y;
    } else {
        word = meta[slot * MW + (e >> 5)];
        idx = meta[slot * MW + NW];
    }
    const typename Vec2<T>::type mm = mins[slot];
    const T mag = (idx == (uint32_t)e) ? mm.y : mm.x;
    return with_sign_bit(mag, (word << (e & 31)) & 0x80000000u);
}

// ---------------------------------------------------------------- QC min-sum kernel
//
// Thread layout: blockDim = FPW * mb * Zp, Zp = roundup(Z, 64).  Thread
// t = ((f*mb + g)*Zp + i) serves, for frame slot f,
//   check pass:    row (block-row g, i)                              (i < Z)
//   variable pass: columns (block-col j, i) for j = g, g+mb, g+2mb... (< nb)
// g and f are wavefront-uniform, so all table look-ups are scalar loads.
//
// Restates ldpcDecoder.decoderMainLoop (ldpc.py:326-337): syndrome of the
// channel hard decision first; then per iteration variableNodesToCheckNodes +
// calcOutgoingValues (check pass) and checkNodesToVariableNodes + channel add
// (variable pass); the syndrome of iteration k's posterior is evaluated by the
// check pass of iteration k+1, whose other results are dropped if it is zero.
template <typename T, int S64, int CPT, int FPW>
__global__ void __launch_bounds__(1024)
#if QC_GENERIC_WAVES
__attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 && CPT <= 8 ? QC_GENERIC_WAVES : 4)))
#endif
qc_minsum_kernel(const QcTables tb, const DecodeArgs a)
{
    typedef typename Vec2<T>::type T2;
    enum { NW = RowMeta<S64>::NW, MW = RowMeta<S64>::MW };

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Z = tb.Z, Zp = tb.Zp, mb = tb.mb, nb = tb.nb, n = tb.n;
    const int slots = FPW * mb * Zp;
    // carve (all offsets multiples of 16)
    size_t off = 0;
    T *post = reinterpret_cast<T *>(smem + off);
    off += (((size_t)FPW * n * sizeof(T)) + 15) & ~(size_t)15;
    T2 *mins = reinterpret_cast<T2 *>(smem + off);
    off += (size_t)slots * sizeof(T2);
    uint32_t *meta = reinterpret_cast<uint32_t *>(smem + off);     // [slots][MW]: sign words, arg-min
    off += ((size_t)slots * MW * 4 + 15) & ~(size_t)15;
    int *flags = reinterpret_cast<int *>(smem + off);   // [2][FPW] "syndrome non-zero"
    int *ones_cnt = flags + 2 * FPW;                    // [FPW]
    off = (off + 3 * FPW * sizeof(int) + 15) & ~(size_t)15;
    // The tables live in LDS for the whole launch.  Read from global memory they cost a full
    // (vector) load latency per edge: their addresses are uniform, but the compiler cannot prove it.
    int4 *l_col_grp = reinterpret_cast<int4 *>(smem + off);
    int2 *l_row_ent = reinterpret_cast<int2 *>(l_col_grp + tb.n_grp);
    int2 *l_col_ent = l_row_ent + tb.n_ent;
    int *l_row_ptr = reinterpret_cast<int *>(l_col_ent + tb.n_ent);
    int *l_col_ptr = l_row_ptr + (mb + 1);
    for (int x = threadIdx.x; x < tb.n_grp; x += blockDim.x) l_col_grp[x] = tb.col_grp[x];
    for (int x = threadIdx.x; x < tb.n_ent; x += blockDim.x) { l_row_ent[x] = tb.row_ent[x]; l_col_ent[x] = tb.col_ent[x]; }
    for (int x = threadIdx.x; x <= mb; x += blockDim.x) l_row_ptr[x] = tb.row_ptr[x];
    for (int x = threadIdx.x; x <= nb; x += blockDim.x) l_col_ptr[x] = tb.col_ptr[x];
    __syncthreads();

    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid / Zp);
    const int f = grp / mb;          // frame slot inside the workgroup
    const int g = grp - f * mb;      // block-row served / first block-column served
    const int i = tid - grp * Zp;
    const bool live = i < Z;
    const int fbase = f * n;         // start of this frame's posteriors in LDS
    const int slot = grp * Zp + i;   // this row's slot in mins/meta
    const int re0 = __builtin_amdgcn_readfirstlane(l_row_ptr[g]);
    const int deg = __builtin_amdgcn_readfirstlane(l_row_ptr[g + 1]) - re0;
    const int nbytes = (n + 7) >> 3;
    const T INF = t_inf(T(0));

    // outgoing messages of this thread's check row, compressed
    T omin1 = T(0), omin2 = T(0);
    uint32_t osw[NW];
    uint32_t oidx = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) osw[w] = 0;
    T2 *chain_mins = reinterpret_cast<T2 *>(a.chain);
    uint32_t *chain_meta = reinterpret_cast<uint32_t *>(a.chain + (size_t)slots * sizeof(T2));
    if (a.chain && live) {
        const T2 mm = chain_mins[slot];
        omin1 = mm.x; omin2 = mm.y;
#pragma unroll
        for (int w = 0; w < NW; ++w) osw[w] = chain_meta[slot * MW + w];
        oidx = chain_meta[slot * MW + NW];
    }

    const int64_t n_fblocks = (a.n_frames + FPW - 1) / FPW;
    for (int64_t fb = blockIdx.x; fb < n_fblocks; fb += gridDim.x) {
        const int64_t frame = fb * FPW + f;
        const bool valid = frame < a.n_frames;

        // ---- load channel values; posterior := channel (decoderSet, ldpc.py:270-274)
        // All loads of the frame are issued before the first one is waited for (one branch on the
        // input type around the whole batch, not one per value: that serialised them).
        T y[CPT];
        {
            const bool take = live && valid;
            const int64_t base = take ? frame * n + i : 0;
            if (a.llr_f64) {
                const double *src = reinterpret_cast<const double *>(a.llr) + base;
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    const int j = g + q * mb;
                    y[q] = (take && j < nb) ? T(src[(int64_t)j * Z]) : T(0);
                }
            } else {
                const float *src = reinterpret_cast<const float *>(a.llr) + base;
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    const int j = g + q * mb;
                    y[q] = (take && j < nb) ? T(src[(int64_t)j * Z]) : T(0);
                }
            }
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int j = g + q * mb;
                // "+ 0" turns a -0.0 input into +0.0: no effect on any comparison or sum
                // below, and it keeps "sign bit set" equivalent to "< 0" for every message.
                y[q] = y[q] * T(a.llr_scale) + T(0);
                if (take && j < nb) post[fbase + j * Z + i] = y[q];
            }
        }
        if (tid < 3 * FPW) flags[tid] = 0;
        if (!a.chain) {
            // fresh check nodes: every outgoing message is +0 (checkNode.__init__, ldpc.py:121).
            // A chained decoder keeps them instead: decoderSet resets only the variable
            // nodes (ldpc.py:270-274).
            omin1 = T(0); omin2 = T(0); oidx = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) osw[w] = 0;
        }
        uint32_t act_mask = 0;       // frame slots still iterating (same value in every thread)
#pragma unroll
        for (int ff = 0; ff < FPW; ++ff)
            if (fb * FPW + ff < a.n_frames) act_mask |= 1u << ff;
        int my_iters = 0;
        int my_conv = 0;
        __syncthreads();

        for (int it = 0;; ++it) {
            const bool active = (act_mask >> f) & 1u;
            const int par = it & 1;
            // ================= check pass (+ syndrome of the current posterior)
            T min1 = INF, min2 = INF;
            uint32_t idx = 0;
            uint32_t nsw[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) nsw[w] = 0;
            if (active && live) {
                unsigned long long synd = 0;
                uint32_t S = 0;
                // 32 edges per sign word; the word number is a constant in each copy of the loop
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const int lim = deg - 32 * w < 32 ? deg - 32 * w : 32;       // uniform
                    if (lim > 0) {
                        const uint32_t ow = osw[w];
                        uint32_t nw = 0;
#pragma unroll 4
                        for (int k = 0; k < lim; ++k) {
                            const int e = 32 * w + k;
                            const int2 ent = l_row_ent[re0 + e];
                            uint32_t c = (uint32_t)(i + ent.y);
                            c = min(c, c - (uint32_t)Z);                 // (i + s) mod Z
                            const T p = post[fbase + ent.x + (int)c];
                            synd ^= __ballot(p > T(0));                  // slicer, ldpc.py:47
                            const T mag_old = (oidx == (uint32_t)e) ? omin2 : omin1;
                            const T c_old = with_sign_bit(mag_old, (ow << k) & 0x80000000u);
                            const T v = p - c_old;                       // receive, ldpc.py:161
                            // setSign, ldpc.py:139: v is never -0 (x - x is +0, and no posterior is -0), so its sign
                            // bit says "negative"; collected first edge highest
                            nw = __builtin_amdgcn_alignbit(nw, sign_word(v), 31);
                            const T av = t_abs(v);                       // ldpc.py:178
                            const bool lt = av < min1;
                            min2 = t_mid(min1, min2, av);
                            idx = lt ? (uint32_t)e : idx;
                            min1 = t_min(min1, av);
                        }
                        if (lim < 32) nw <<= (32 - lim);
                        nsw[w] = nw;
                        S ^= __popc(nw) & 1u;
                    }
                }
                // sign of every outgoing message = (product of all signs) * own sign, ldpc.py:190
                if (a.sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)deg & 1u;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const int lim = deg - 32 * w;
                    const uint32_t full = lim >= 32 ? 0xffffffffu : lim > 0 ? 0xffffffffu << (32 - lim) : 0u;
                    nsw[w] ^= S ? full : 0u;
                }
                min1 = ms_correct(min1, a.ms_alpha, a.ms_beta);
                min2 = ms_correct(min2, a.ms_alpha, a.ms_beta);
                T2 mm; mm.x = min1; mm.y = min2;
                mins[slot] = mm;
#pragma unroll
                for (int w = 0; w < NW; ++w) meta[slot * MW + w] = nsw[w];
                meta[slot * MW + NW] = idx;
                if (synd != 0ull) flags[par * FPW + f] = 1;
            }
            __syncthreads();
            // ================= stop test (ldpc.py:332-335), identical in every thread
            uint32_t next_mask = 0;
#pragma unroll
            for (int ff = 0; ff < FPW; ++ff) {
                if ((act_mask >> ff) & 1u) {
                    const int bad = flags[par * FPW + ff];
                    if (bad != 0 && it < a.max_iters) {
                        next_mask |= 1u << ff;
                    } else if (ff == f) {
                        my_iters = it;
                        my_conv = (bad == 0);
                    }
                }
            }
            act_mask = next_mask;
            if (act_mask == 0) break;
            // the row's messages of this iteration become its "previous" messages only now:
            // a frame that stops here keeps the ones its last completed iteration sent
            if (((act_mask >> f) & 1u) && live) {
                omin1 = min1; omin2 = min2; oidx = idx;
#pragma unroll
                for (int w = 0; w < NW; ++w) osw[w] = nsw[w];
            }
            if (tid < FPW) flags[(par ^ 1) * FPW + tid] = 0;
            // ================= variable pass (ldpc.py:295-311, 321-323)
            if (((act_mask >> f) & 1u) && live) {
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    const int j = g + q * mb;
                    if (j < nb) {
                        T acc = T(0);
                        const int g0 = __builtin_amdgcn_readfirstlane(l_col_ptr[j]), g1 = __builtin_amdgcn_readfirstlane(l_col_ptr[j + 1]);
                        for (int gi = g0; gi < g1; ++gi) {
                            const int4 cg = l_col_grp[gi];
                            const int rbase = (f * mb + cg.x) * Zp;
                            const int2 *ent = l_col_ent + cg.z;
                            if (cg.y == 1) {
                                const int2 en = ent[0];
                                uint32_t r = (uint32_t)(i - en.x);
                                r = min(r, r + (uint32_t)Z);
                                acc = acc + rebuild_row<T, S64>(mins, meta, rbase + (int)r, en.y);
                            } else if (cg.y == 2) {
                                const int2 hi = ent[0], lo = ent[1];     // hi.x > lo.x
                                uint32_t rh = (uint32_t)(i - hi.x); rh = min(rh, rh + (uint32_t)Z);
                                uint32_t rl = (uint32_t)(i - lo.x); rl = min(rl, rl + (uint32_t)Z);
                                const T vh = rebuild_row<T, S64>(mins, meta, rbase + (int)rh, hi.y);
                                const T vl = rebuild_row<T, S64>(mins, meta, rbase + (int)rl, lo.y);
                                // smaller row index first: the low shift's row comes first
                                // exactly when lo.x <= i < hi.x
                                const bool lo_first = (i >= lo.x) && (i < hi.x);
                                acc = acc + (lo_first ? vl : vh);
                                acc = acc + (lo_first ? vh : vl);
                            } else if (cg.y <= 8) {
                                // Shifts descending: the rows i - s of the shifts s <= i ascend, then the rows i - s + Z
                                // of the shifts s > i, which are the first `wrapped` entries.  All messages are fetched
                                // first (their LDS latencies overlap), then added in that order; an entry that is not
                                // its pass's turn adds +0, which changes no sum (x + 0 == x; a partial sum of -0 that
                                // becomes +0 meets a channel value that is never -0 in the end).
                                T v[8];
                                int wrapped = 0;
#pragma unroll
                                for (int k = 0; k < 8; ++k) {
                                    v[k] = T(0);
                                    if (k < cg.y) {                      // uniform
                                        const int2 en = ent[k];
                                        int r = i - en.x;
                                        const bool wr = r < 0;
                                        r += wr ? Z : 0;
                                        wrapped += wr ? 1 : 0;
                                        v[k] = rebuild_row<T, S64>(mins, meta, rbase + r, en.y);
                                    }
                                }
#pragma unroll
                                for (int k = 0; k < 8; ++k) acc = acc + (k >= wrapped ? v[k] : T(0));
#pragma unroll
                                for (int k = 0; k < 8; ++k) acc = acc + (k < wrapped ? v[k] : T(0));
                            } else {
                                // shifts descending: rows i - s (s <= i) ascend, then rows i - s + Z (s > i) ascend
                                for (int k = 0; k < cg.y; ++k) {
                                    const int2 en = ent[k];
                                    if (en.x <= i)
                                        acc = acc + rebuild_row<T, S64>(mins, meta, rbase + i - en.x, en.y);
                                }
                                for (int k = 0; k < cg.y; ++k) {
                                    const int2 en = ent[k];
                                    if (en.x > i)
                                        acc = acc + rebuild_row<T, S64>(mins, meta, rbase + i - en.x + Z, en.y);
                                }
                            }
                        }
                        post[fbase + j * Z + i] = acc + y[q];            // channel last, ldpc.py:322
                    }
                }
            }
            __syncthreads();
        }

        // ================= outputs
        for (int x = tid; x < FPW * nbytes; x += blockDim.x) {
            const int ff = x / nbytes;
            const int qb = x - ff * nbytes;
            const int64_t fr = fb * FPW + ff;
            if (fr < a.n_frames) {
                uint32_t byte = 0;
                const int c0 = qb * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c0 + k < n) byte |= (uint32_t)(post[ff * n + c0 + k] > T(0)) << k;
                if (a.hard) a.hard[fr * nbytes + qb] = (uint8_t)byte;
                if (byte) atomicAdd(&ones_cnt[ff], (int)__popc(byte));
            }
        }
        if (a.soft && valid && live) {
            T *so = reinterpret_cast<T *>(a.soft) + frame * n;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int j = g + q * mb;
                if (j < nb) so[j * Z + i] = post[fbase + j * Z + i];
            }
        }
        __syncthreads();
        if (valid && g == 0 && i == 0) {
            if (a.iters) a.iters[frame] = my_iters;
            if (a.conv) a.conv[frame] = (uint8_t)my_conv;
            if (a.ones) a.ones[frame] = ones_cnt[f];
        }
        __syncthreads();
    }
    if (a.chain && live) {
        T2 mm; mm.x = omin1; mm.y = omin2;
        chain_mins[slot] = mm;
#pragma unroll
        for (int w = 0; w < NW; ++w) chain_meta[slot * MW + w] = osw[w];
        chain_meta[slot * MW + NW] = oidx;
    }
}

#include "awgn.h"
#include "qc_regular.h"
#include "qc_planes.h"
#include "edge_slots.h"
#include "edge_packed.h"

// ---------------------------------------------------------------- CSR min-sum kernel
//
// Arbitrary sparse H.  One 256-thread workgroup per frame; posteriors and the
// compressed row states live in a per-workgroup slice of a global workspace
// (they stay in L2 for the sizes this path is used for).  Same arithmetic as
// the QC kernel; the syndrome is a pass of its own here.
template <typename T>
__global__ void __launch_bounds__(256)
csr_minsum_kernel(const CsrTables tb, const DecodeArgs a, unsigned char *workspace, size_t ws_stride)
{
    typedef typename Vec2<T>::type T2;
    typedef typename SignBits<64>::Meta Meta;
    __shared__ int flags[2];
    __shared__ int ones_cnt;
    __shared__ int result[2];
    const int m = tb.m, n = tb.n;
    unsigned char *ws = workspace + (size_t)blockIdx.x * ws_stride;
    T *post = reinterpret_cast<T *>(ws);
    T *ych = post + n;
    size_t off = (((size_t)2 * n * sizeof(T)) + 15) & ~(size_t)15;
    T2 *mins = reinterpret_cast<T2 *>(ws + off);
    off += (size_t)m * sizeof(T2);
    Meta *meta = reinterpret_cast<Meta *>(ws + off);
    const int tid = threadIdx.x;
    const int nbytes = (n + 7) >> 3;
    const T INF = t_inf(T(0));

    for (int64_t frame = blockIdx.x; frame < a.n_frames; frame += gridDim.x) {
        for (int c = tid; c < n; c += blockDim.x) {
            const T yv = load_llr<T>(a.llr, a.llr_f64, frame * n + c) * T(a.llr_scale) + T(0);
            ych[c] = yv;
            post[c] = yv;
        }
        if (!a.chain) {
            for (int r = tid; r < m; r += blockDim.x) {
                T2 z; z.x = T(0); z.y = T(0);
                mins[r] = z;
                Meta mt; mt.signs = 0; mt.idx = 0; mt.pad = 0;
                meta[r] = mt;
            }
        }
        if (tid < 2) flags[tid] = 0;
        if (tid == 0) ones_cnt = 0;
        __syncthreads();
        int it = 0;
        for (;; ++it) {
            const int par = it & 1;
            // syndrome of the current posterior (isCodeword, ldpc.py:249-268)
            bool bad_row = false;
            for (int r = tid; r < m; r += blockDim.x) {
                uint32_t sy = 0;
                for (int e = tb.row_ptr[r]; e < tb.row_ptr[r + 1]; ++e)
                    sy ^= (uint32_t)(post[tb.col_idx[e]] > T(0));
                bad_row |= (sy != 0);
            }
            if (bad_row) flags[par] = 1;
            __syncthreads();
            const int bad = flags[par];
            if (bad == 0 || it >= a.max_iters) {
                if (tid == 0) { result[0] = it; result[1] = (bad == 0); }
                break;
            }
            if (tid == 0) flags[par ^ 1] = 0;
            // check pass
            for (int r = tid; r < m; r += blockDim.x) {
                const int e0 = tb.row_ptr[r], e1 = tb.row_ptr[r + 1];
                const T2 om = mins[r];
                const Meta ot = meta[r];
                T min1 = INF, min2 = INF;
                uint32_t idx = 0;
                uint64_t ns = 0;
                for (int e = e0; e < e1; ++e) {
                    const int k = e - e0;
                    const T v = post[tb.col_idx[e]] - rebuild<T, 64>(om, ot, k);
                    ns |= (uint64_t)(v < T(0)) << k;
                    const T av = t_abs(v);
                    const bool lt = av < min1;
                    min2 = t_mid(min1, min2, av);
                    idx = lt ? (uint32_t)k : idx;
                    min1 = t_min(min1, av);
                }
                const int deg = e1 - e0;
                uint32_t S = parity_of<uint64_t>(ns);
                if (a.sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)deg & 1u;
                const uint64_t full = (deg >= 64) ? ~0ull : ((1ull << deg) - 1);
                ns ^= S ? full : 0ull;
                min1 = ms_correct(min1, a.ms_alpha, a.ms_beta);
                min2 = ms_correct(min2, a.ms_alpha, a.ms_beta);
                T2 mm; mm.x = min1; mm.y = min2;
                mins[r] = mm;
                Meta mt; mt.signs = ns; mt.idx = idx; mt.pad = 0;
                meta[r] = mt;
            }
            __syncthreads();
            // variable pass
            for (int c = tid; c < n; c += blockDim.x) {
                T acc = T(0);
                for (int k = tb.col_ptr[c]; k < tb.col_ptr[c + 1]; ++k) {
                    const int r = tb.col_row[k];
                    acc = acc + rebuild<T, 64>(mins[r], meta[r], tb.col_e[k]);
                }
                post[c] = acc + ych[c];
            }
            __syncthreads();
        }
        for (int qb = tid; qb < nbytes; qb += blockDim.x) {
            uint32_t byte = 0;
            for (int k = 0; k < 8; ++k)
                if (qb * 8 + k < n) byte |= (uint32_t)(post[qb * 8 + k] > T(0)) << k;
            if (a.hard) a.hard[frame * nbytes + qb] = (uint8_t)byte;
            if (byte) atomicAdd(&ones_cnt, (int)__popc(byte));
        }
        if (a.soft) {
            T *so = reinterpret_cast<T *>(a.soft) + frame * n;
            for (int c = tid; c < n; c += blockDim.x) so[c] = post[c];
        }
        __syncthreads();
        if (tid == 0) {
            if (a.iters) a.iters[frame] = result[0];
            if (a.conv) a.conv[frame] = (uint8_t)result[1];
            if (a.ones) a.ones[frame] = ones_cnt;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- min-sum on an edge list, any check degree
//
// The state of a long gym episode: every action installs a circulant of up to 7 hot bits and the
// environment keeps them all (ldpc_env.py:293-317), so check degrees grow past what the other kernels'
// sign masks and unrolled rows hold (64).  One explicit message per edge in a per-workgroup slice of a
// global workspace, rows and columns walked by loops: no degree enters a data structure.  Order of the
// passes and arithmetic are csr_minsum_kernel's, so results are bit-identical; this is the slow, general
// fallback.  Chained mode keeps the message array between frames.
template <typename T>
__global__ void __launch_bounds__(256)
csr_wide_minsum_kernel(const CsrTables tb, const DecodeArgs a, unsigned char *workspace, size_t ws_stride)
{
    __shared__ int flags[2];
    __shared__ int ones_cnt;
    __shared__ int result[2];
    const int m = tb.m, n = tb.n;
    unsigned char *ws = workspace + (size_t)blockIdx.x * ws_stride;
    T *post = reinterpret_cast<T *>(ws);
    T *ych = post + n;
    T *msg = ych + n;                                    // [E]: check-to-variable message of every edge (in a check pass: v)
    const int tid = threadIdx.x;
    const int nbytes = (n + 7) >> 3;
    const T INF = t_inf(T(0));

    for (int64_t frame = blockIdx.x; frame < a.n_frames; frame += gridDim.x) {
        for (int c = tid; c < n; c += blockDim.x) {
            const T yv = load_llr<T>(a.llr, a.llr_f64, frame * n + c) * T(a.llr_scale) + T(0);
            ych[c] = yv;
            post[c] = yv;
        }
        if (!a.chain)
            for (int e = tid; e < tb.E; e += blockDim.x) msg[e] = T(0);      // fresh check nodes, ldpc.py:121
        if (tid < 2) flags[tid] = 0;
        if (tid == 0) ones_cnt = 0;
        __syncthreads();
        int it = 0;
        for (;; ++it) {
            const int par = it & 1;
            // syndrome of the current posterior (isCodeword, ldpc.py:249-268)
            bool bad_row = false;
            for (int r = tid; r < m; r += blockDim.x) {
                uint32_t sy = 0;
                for (int e = tb.row_ptr[r]; e < tb.row_ptr[r + 1]; ++e)
                    sy ^= (uint32_t)(post[tb.col_idx[e]] > T(0));
                bad_row |= (sy != 0);
            }
            if (bad_row) flags[par] = 1;
            __syncthreads();
            const int bad = flags[par];
            if (bad == 0 || it >= a.max_iters) {                             // ldpc.py:332-335
                if (tid == 0) { result[0] = it; result[1] = (bad == 0); }
                break;
            }
            if (tid == 0) flags[par ^ 1] = 0;
            // check pass (ldpc.py:150-163, 174-202)
            for (int r = tid; r < m; r += blockDim.x) {
                const int e0 = tb.row_ptr[r], e1 = tb.row_ptr[r + 1];
                T min1 = INF, min2 = INF;
                int idx = e0;
                uint32_t neg = 0;
                for (int e = e0; e < e1; ++e) {
                    const T v = post[tb.col_idx[e]] - msg[e];
                    msg[e] = v;
                    neg ^= (uint32_t)(v < T(0));                             // sign(0) = +1, ldpc.py:139
                    const T av = t_abs(v);
                    const bool lt = av < min1;
                    min2 = t_mid(min1, min2, av);
                    idx = lt ? e : idx;
                    min1 = t_min(min1, av);
                }
                uint32_t S = neg;
                if (a.sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)(e1 - e0) & 1u;
                min1 = ms_correct(min1, a.ms_alpha, a.ms_beta);
                min2 = ms_correct(min2, a.ms_alpha, a.ms_beta);
                for (int e = e0; e < e1; ++e) {
                    const T mag = e == idx ? min2 : min1;                    // ldpc.py:190-191
                    const uint32_t ng = (uint32_t)(msg[e] < T(0)) ^ S;
                    msg[e] = ng ? -mag : mag;
                }
            }
            __syncthreads();
            // variable pass: ordered sum over ascending check index, channel last (ldpc.py:298-303, 322)
            for (int c = tid; c < n; c += blockDim.x) {
                T acc = T(0);
                for (int k = tb.col_ptr[c]; k < tb.col_ptr[c + 1]; ++k)
                    acc = acc + msg[tb.row_ptr[tb.col_row[k]] + tb.col_e[k]];
                post[c] = acc + ych[c];
            }
            __syncthreads();
        }
        for (int qb = tid; qb < nbytes; qb += blockDim.x) {
            uint32_t byte = 0;
            for (int k = 0; k < 8; ++k)
                if (qb * 8 + k < n) byte |= (uint32_t)(post[qb * 8 + k] > T(0)) << k;
            if (a.hard) a.hard[frame * nbytes + qb] = (uint8_t)byte;
            if (byte) atomicAdd(&ones_cnt, (int)__popc(byte));
        }
        if (a.soft) {
            T *so = reinterpret_cast<T *>(a.soft) + frame * n;
            for (int c = tid; c < n; c += blockDim.x) so[c] = post[c];
        }
        __syncthreads();
        if (tid == 0) {
            if (a.iters) a.iters[frame] = result[0];
            if (a.conv) a.conv[frame] = (uint8_t)result[1];
            if (a.ones) a.ones[frame] = ones_cnt;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- sum-product kernel
// phi(x) = -ln tanh(x/2) = ln((e^x + 1)/(e^x - 1)), its own inverse on x > 0
__device__ __forceinline__ float spa_phi(float x) { return log1pf(2.0f / expm1f(x)); }
__device__ __forceinline__ double spa_phi(double x) { return log1p(2.0 / expm1(x)); }
template <typename T> __device__ __forceinline__ T spa_clamp(T x)
{
    // magnitudes entering phi are kept in [1e-12, 80]: phi(1e-12) = 28.3, phi(80) = 3.6e-35 (a normal float)
    return x < T(1e-12) ? T(1e-12) : (x > T(80) ? T(80) : x);
}

// Flooding sum-product (tanh rule in the log domain) on an edge list; one frame per
// workgroup pass.  No counterpart in the reference, whose only check rule is min-sum
// (ldpc.py:174-202); schedule, polarity, stop test and variable update are the reference's
// (ldpc.py:282-337).  Per check row, with F_e = phi(clamp|v_e|) accumulated in edge order:
//     F1 = largest F so far (first wins ties), rest = ordered sum of all the others
//     exclusive sum of edge e = rest if e holds F1, else (rest - F_e) + F1
//     |message_e| = phi(clamp(exclusive sum));  signs as in the min-sum kernels.
// Splitting off the largest term keeps every subtraction well conditioned.
// post/msg live in LDS when use_lds (n + E values fit), else in this workgroup's slice of
// `workspace`.  The message slot of an edge holds F_e between the two halves of a check pass.
template <typename T>
__global__ void __launch_bounds__(1024)
csr_spa_kernel(const CsrTables tb, const DecodeArgs a, unsigned char *workspace, size_t ws_stride, int use_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int flags[2];
    __shared__ int ones_cnt;
    const int m = tb.m, n = tb.n;
    T *post = use_lds ? reinterpret_cast<T *>(smem) : reinterpret_cast<T *>(workspace + (size_t)blockIdx.x * ws_stride);
    T *msg = post + n;
    const int tid = threadIdx.x;
    const int nbytes = (n + 7) >> 3;

    for (int64_t frame = blockIdx.x; frame < a.n_frames; frame += gridDim.x) {
        for (int c = tid; c < n; c += blockDim.x)
            post[c] = load_llr<T>(a.llr, a.llr_f64, frame * n + c) * T(a.llr_scale) + T(0);
        for (int e = tid; e < tb.E; e += blockDim.x) msg[e] = T(0);
        if (tid < 2) flags[tid] = 0;
        if (tid == 0) ones_cnt = 0;
        __syncthreads();
        int it = 0, conv = 0;
        for (;; ++it) {
            const int par = it & 1;
            bool bad_row = false;
            for (int r = tid; r < m; r += blockDim.x) {
                const int e0 = tb.row_ptr[r], e1 = tb.row_ptr[r + 1];
                uint32_t sy = 0, neg = 0;
                T F1 = T(-1), rest = T(0);
                int arg = -1;
                for (int e = e0; e < e1; ++e) {
                    const T p = post[tb.col_idx[e]];
                    sy ^= (uint32_t)(p > T(0));                      // slicer, ldpc.py:47
                    const T v = p - msg[e];                          // receive, ldpc.py:161
                    const bool ng = v < T(0);                        // sign(0) = +1, ldpc.py:139
                    neg ^= (uint32_t)ng;
                    const T F = spa_phi(spa_clamp(t_abs(v)));
                    if (F > F1) {
                        if (arg >= 0) rest = rest + F1;
                        F1 = F;
                        arg = e;
                    } else {
                        rest = rest + F;
                    }
                    msg[e] = ng ? -F : F;                            // F > 0 (the clamp): its sign carries the edge's (any degree)
                }
                bad_row |= (sy != 0);
                const int deg = e1 - e0;
                uint32_t S = neg;
                if (a.sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)deg & 1u;
                for (int e = e0; e < e1; ++e) {
                    const T sf = msg[e];
                    const T x = (e == arg) ? rest : (rest - t_abs(sf)) + F1;
                    const T mag = spa_phi(spa_clamp(x));
                    const uint32_t ng = (uint32_t)(sf < T(0)) ^ S;
                    msg[e] = ng ? -mag : mag;
                }
            }
            if (bad_row) flags[par] = 1;
            __syncthreads();
            const int bad = flags[par];                              // stop test, ldpc.py:332-335
            if (bad == 0 || it >= a.max_iters) {
                conv = (bad == 0);
                break;
            }
            if (tid == 0) flags[par ^ 1] = 0;
            // variable pass: ordered sum over ascending check index, channel last (ldpc.py:298-303, 322)
            for (int c = tid; c < n; c += blockDim.x) {
                T acc = T(0);
                for (int k = tb.col_ptr[c]; k < tb.col_ptr[c + 1]; ++k)
                    acc = acc + msg[tb.row_ptr[tb.col_row[k]] + tb.col_e[k]];
                post[c] = acc + (load_llr<T>(a.llr, a.llr_f64, frame * n + c) * T(a.llr_scale) + T(0));
            }
            __syncthreads();
        }
        for (int qb = tid; qb < nbytes; qb += blockDim.x) {
            uint32_t byte = 0;
            for (int k = 0; k < 8; ++k)
                if (qb * 8 + k < n) byte |= (uint32_t)(post[qb * 8 + k] > T(0)) << k;
            if (a.hard) a.hard[frame * nbytes + qb] = (uint8_t)byte;
            if (byte) atomicAdd(&ones_cnt, (int)__popc(byte));
        }
        if (a.soft) {
            T *so = reinterpret_cast<T *>(a.soft) + frame * n;
            for (int c = tid; c < n; c += blockDim.x) so[c] = post[c];
        }
        __syncthreads();
        if (tid == 0) {
            if (a.iters) a.iters[frame] = it;
            if (a.conv) a.conv[frame] = (uint8_t)conv;
            if (a.ones) a.ones[frame] = ones_cnt;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- channel kernel
// y = -1 + sigma*z for the all-zero codeword (modulate: 0 -> -1, ldpc.py:62-66;
// addAWGN, ldpc.py:51-60).  One workgroup per frame; per-frame sigma_hat
// (ldpc.py:58) and channel hard-decision errors (ldpcCUDA.py:586-587).
__global__ void __launch_bounds__(256)
channel_awgn_kernel(uint64_t seed, uint32_t stream, uint64_t first_frame, int64_t n_frames, int n,
                    float sigma, float *llr, float *sigma_hat, int32_t *uncoded)
{
    __shared__ unsigned long long s_sum[4];
    __shared__ int s_err[4];
    const int tid = threadIdx.x;
    const int nq = (n + 3) >> 2;
    for (int64_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        unsigned long long sum = 0;
        int err = 0;
        float *out = llr + f * n;
        for (int q = tid; q < nq; q += blockDim.x) {
            const float4 z = normal4(seed, first_frame + (uint64_t)f, (uint32_t)q, stream);
            const float nz[4] = {sigma * z.x, sigma * z.y, sigma * z.z, sigma * z.w};
            float yv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                yv[k] = -1.0f + nz[k];
                if (4 * q + k < n) {
                    sum += awgn_sq_fixed(k == 0 ? z.x : k == 1 ? z.y : k == 2 ? z.z : z.w);
                    err += yv[k] > 0.f;
                }
            }
            if (4 * q + 3 < n && ((((uintptr_t)(out + 4 * q)) & 15) == 0)) {
                *reinterpret_cast<float4 *>(out + 4 * q) = make_float4(yv[0], yv[1], yv[2], yv[3]);
            } else {
                for (int k = 0; k < 4; ++k)
                    if (4 * q + k < n) out[4 * q + k] = yv[k];
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_down(sum, o);
            err += __shfl_down(err, o);
        }
        if ((tid & 63) == 0) { s_sum[tid >> 6] = sum; s_err[tid >> 6] = err; }
        __syncthreads();
        if (tid == 0) {
            if (sigma_hat) sigma_hat[f] = awgn_sigma_hat(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3], sigma, n);
            if (uncoded) uncoded[f] = s_err[0] + s_err[1] + s_err[2] + s_err[3];
        }
        __syncthreads();
    }
}

struct PointSigmas { float sigma[GEN_MAX_POINTS]; };

// counters[6]: frames, frame errors, decoded bit errors, iterations, uncoded bit errors, not-a-codeword
__global__ void __launch_bounds__(256)
accumulate_kernel(int64_t n_frames, int n, float sigma, const int32_t *iters, const uint8_t *conv, const int32_t *ones,
                  const int32_t *uncoded, const float *sigma_hat, const unsigned long long *sumsq,
                  unsigned long long *counters, float *per_frame, uint32_t per_point, PointSigmas points)
{
    if (per_point) {
        // a launch that served several sweep points (at most GEN_MAX_BATCH frames): frame f counts for point f / per_point
        for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n_frames;
             f += (int64_t)gridDim.x * blockDim.x) {
            const uint32_t pt = (uint32_t)f / per_point;
            unsigned long long *c = counters + 6 * pt;
            atomicAdd(&c[0], 1ull);
            if (ones[f]) { atomicAdd(&c[1], 1ull); atomicAdd(&c[2], (unsigned long long)ones[f]); }
            if (iters[f]) atomicAdd(&c[3], (unsigned long long)iters[f]);
            if (uncoded[f]) atomicAdd(&c[4], (unsigned long long)uncoded[f]);
            if (conv[f] == 0) atomicAdd(&c[5], 1ull);
            if (per_frame) {
                float *p = per_frame + 4 * f;
                p[0] = sumsq ? awgn_sigma_hat(sumsq[f], points.sigma[pt], n) : sigma_hat[f];
                p[1] = (float)uncoded[f];
                p[2] = (float)ones[f];
                p[3] = (float)iters[f];
            }
        }
        return;
    }
    unsigned long long c[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n_frames;
         f += (int64_t)gridDim.x * blockDim.x) {
        c[0] += 1;
        c[1] += ones[f] != 0;
        c[2] += (unsigned long long)ones[f];
        c[3] += (unsigned long long)iters[f];
        c[4] += (unsigned long long)uncoded[f];
        c[5] += conv[f] == 0;
        if (per_frame) {
            float *p = per_frame + 4 * f;
            p[0] = sumsq ? awgn_sigma_hat(sumsq[f], sigma, n) : sigma_hat[f];      // a decode kernel that made its own noise leaves the sum
            p[1] = (float)uncoded[f];
            p[2] = (float)ones[f];
            p[3] = (float)iters[f];
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        unsigned long long v = c[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&counters[k], v);
    }
}

// LDPC_HIP_PRECHECK_T: threshold of the syndrome-only pass for every kernel that has one (tuning, and the
// tests that run it always / never); unset = the kernels' defaults
static int precheck_override()
{
    const char *e = hook("LDPC_HIP_PRECHECK_T");
    return (e && e[0]) ? std::max(0, std::min(1 << 19, atoi(e))) : -1;
}

// ---------------------------------------------------------------- handle
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { p = nullptr; return fail(LDPC_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Page-locked host memory (staging of the host entry points: a copy engine reads and writes it
// at PCIe speed and asynchronously, which it cannot do with the caller's pageable arrays)
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return fail(LDPC_E_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// One of the two slots of the pipelined host entry point: chunk k uses slot k & 1
struct HostSlot {
    PinBuf in, out;
    DevBuf llr, hard, iters, conv, ones, soft;
    hipEvent_t up = nullptr, done = nullptr;
};

// memcpy over several host threads (one thread moves 8-12 GB/s, a PCIe 5 x16 link 50+).
class CopyPool {
  public:
    explicit CopyPool(int workers)
    {
        for (int t = 0; t < workers; ++t) threads_.emplace_back([this, t] { run(t); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            ++generation_;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void copy(void *dst, const void *src, size_t bytes)
    {
        const size_t parts = threads_.size() + 1;
        if (bytes < (1u << 20) || parts == 1) { memcpy(dst, src, bytes); return; }
        {
            std::lock_guard<std::mutex> g(m_);
            dst_ = static_cast<unsigned char *>(dst);
            src_ = static_cast<const unsigned char *>(src);
            bytes_ = bytes;
            pending_ = (int)threads_.size();
            ++generation_;
        }
        cv_.notify_all();
        slice(parts - 1);                       // the caller moves the last part itself
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
    }

  private:
    void slice(size_t part)
    {
        const size_t parts = threads_.size() + 1;
        const size_t step = ((bytes_ + parts - 1) / parts + 4095) & ~(size_t)4095;
        const size_t a = std::min(bytes_, part * step), b = std::min(bytes_, a + step);
        if (b > a) memcpy(dst_ + a, src_ + a, b - a);
    }
    void run(int t)
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
            }
            slice((size_t)t);
            {
                std::lock_guard<std::mutex> g(m_);
                --pending_;
            }
            done_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    unsigned char *dst_ = nullptr;
    const unsigned char *src_ = nullptr;
    size_t bytes_ = 0;
    int pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

struct Launch {
    const void *fn = nullptr;
    const void *fn_gen = nullptr;   // the same kernel generating its own channel values (sweep driver), if it exists
    int grid = 0, block = 0, fpw = 1;
    size_t lds = 0;
    bool spa_lds = false;     // sum-product: posteriors and messages in LDS (else in the workspace)
    bool spa_planes = false;  // sum-product on the message-plane kernel
    bool slots = false;       // short-code kernel on the edge list (edge_slots.h, edge_packed.h)
    bool packed = false;      // ... the one with records of the columns' own length (edge_packed.h)
    bool planes = false;      // message-plane kernel (qc_planes.h)
    bool csr = false;         // csr_minsum_kernel (state in a global workspace)
    bool heavyqc = false;     // qc_heavy_kernel (qc_heavy.hip)
    int max_grid = 1;         // workgroups that are resident at once
    size_t ws_stride = 0;     // bytes of global workspace per workgroup (0: none)
};

struct ldpc_handle {
    int device = 0;
    int n_cu = 0;
    size_t lds_max = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool is_qc = false;
    int m = 0, n = 0, E = 0, max_row_deg = 0;
    bool wide = false;        // some check has more than 64 edges: of the edge-list min-sum kernels the any-degree one applies
    // QC
    QcTables qc{};
    int cols_per_thread = 0;
    // regular fast path (every non-empty block has the same number of shifts)
    bool is_regular = false;
    bool reg_holes = false;
    int reg_w = 0;
    RegTables reg{};
    // message-plane kernel (2 x 16 circulants of size 511, every weight <= 2)
    float ms_alpha = 1.0f, ms_beta = 0.0f;     // min-sum magnitude correction (ldpc_set_minsum_scaling)
    bool planes_ok = false;
    bool planes_heavy = false;     // some circulant has weight > 2 (extra edges through a global scratch row)
    int planes_max_w = 0;          // largest circulant weight
    PlanesTables planes{};
    // heavy-circulant kernel (float32 min-sum; qc_heavy.h)
    bool heavyqc_ok = false;
    HeavyTables heavyqc{};
    size_t heavyqc_lds = 0;
    // CSR
    CsrTables csr{};
    const int32_t *edge_slot = nullptr;   // [E] rank of each edge's row inside its column
    const int32_t *col_rec = nullptr;     // [n] first word of each column's record in a frame of edge_packed.h
    int rec_words = 0;                    // words of all the records of a frame
    int max_col_deg = 0, min_row_deg = 0;
    // the code's tables: staged on the host while the handle is built, then ONE device allocation
    // (taken from the per-device pool when a destroyed handle left one that is large enough)
    std::vector<unsigned char> stage;
    std::vector<std::pair<size_t, const void **>> fixups;    // (offset in stage, pointer to set)
    void *arena = nullptr;
    size_t arena_cap = 0;
    // The edge list of a block-circulant code is built when a kernel that walks it is first chosen: a code
    // the message-plane kernels decode never needs it, and the gym makes a new handle per step
    // (ldpc_env.py:313, 370), where sorting 32,704 edges and copying 0.8 MB of tables was most of the cost.
    bool edges_ready = false;
    std::vector<int32_t> qc_blk_ptr, qc_shifts;    // the code as ldpc_create_qc received it, until then
    void *edge_arena = nullptr;
    size_t edge_arena_cap = 0;
    // staging for the host entry points
    DevBuf b_llr, b_hard, b_iters, b_conv, b_ones, b_soft, b_sig, b_sq, b_unc, b_cnt, b_pf, b_ws, b_work;
    // check-node state of the chained (stateful) decoder, one per arithmetic width
    DevBuf b_chain32, b_chain64;
    // pipelined host entry point (large calls only): two slots and a stream for the uploads
    HostSlot slot[2];
    hipStream_t copy_stream = nullptr;
    unsigned long long *pass_counts = nullptr;   // caller's device counters (ldpc_set_pass_counters), or null
    unsigned char *scratch = nullptr;   // this device's copy of the static scratch array
    Launch launch_cache[2][2][2];   // [algo][prec == 64][chained]; fn == nullptr: not chosen yet
};


// posteriors [n] + one message per edge, of the arithmetic type
static size_t spa_state_bytes(const ldpc_handle *h, int prec)
{
    return ((size_t)h->n + (size_t)h->E) * (prec == 64 ? 8 : 4);
}

template <typename H> static int upload(ldpc_handle *h, const std::vector<H> &v, const H **out)
{
    const size_t off = (h->stage.size() + 255) & ~(size_t)255;
    h->stage.resize(off + std::max<size_t>(v.size(), 1) * sizeof(H));
    if (!v.empty()) memcpy(h->stage.data() + off, v.data(), v.size() * sizeof(H));
    h->fixups.emplace_back(off, reinterpret_cast<const void **>(out));
    *out = nullptr;
    return 0;
}

// What a destroyed handle leaves behind for the next one on the same device: its stream and its
// table allocation.  The gym hands over a new matrix at every step (ldpc_env.py:313, 370), so
// handles come and go at that rate; a stream creation and a dozen hipMalloc calls per step are
// most of what a new handle costs.
// Its small staging and result buffers travel along (a sweep of 3 x 10 frames on a fresh handle spent
// more time in eight hipMalloc calls than in its kernels); large ones are freed as before.
enum { POOL_KEEP = 8, POOL_BUFS = 13, POOL_BUF_MAX = 4 << 20 };
struct PooledParts { hipStream_t stream; void *arena; size_t cap; DevBuf bufs[POOL_BUFS]; };
static std::mutex g_pool_mutex;
static std::vector<PooledParts> g_pool[64];

// The buffers of a handle that hold nothing between calls, in the order they are pooled in.
static void scratch_buffers(ldpc_handle *h, DevBuf *(&out)[POOL_BUFS])
{
    DevBuf *all[POOL_BUFS] = {&h->b_llr, &h->b_hard, &h->b_iters, &h->b_conv, &h->b_ones, &h->b_soft, &h->b_sig,
                              &h->b_sq, &h->b_unc, &h->b_cnt, &h->b_pf, &h->b_ws, &h->b_work};
    for (int k = 0; k < POOL_BUFS; ++k) out[k] = all[k];
}

// All staged tables to the device in one allocation and one copy; sets the table pointers.
static int commit_tables(ldpc_handle *h, void **arena, size_t *arena_cap)
{
    const size_t need = std::max<size_t>(h->stage.size(), 256);
    if (*arena_cap < need) {
        if (*arena) { (void)hipFree(*arena); *arena = nullptr; *arena_cap = 0; }
        const size_t cap = (need + 65535) & ~(size_t)65535;
        HIP_TRY(hipMalloc(arena, cap));
        *arena_cap = cap;
    }
    HIP_TRY(hipMemcpyAsync(*arena, h->stage.data(), h->stage.size(), hipMemcpyHostToDevice, h->own_stream));
    HIP_TRY(hipStreamSynchronize(h->own_stream));         // the staging vector goes away below
    for (auto &f : h->fixups) *f.second = static_cast<const unsigned char *>(*arena) + f.first;
    h->stage.clear();
    h->stage.shrink_to_fit();
    h->fixups.clear();
    return 0;
}

// ---------------------------------------------------------------- scratch of the allocation-free entry point
// ldpc_decode_device_noalloc keeps the frame-claim counter (first NOALLOC_COUNTER_BYTES) and the
// per-workgroup workspaces of the edge-list kernels in this array instead of in buffers the
// handle allocates.  Every GPU the library runs on holds its own copy, at its own address,
// until the process ends.  In host code the array's name is only a stand-in: device addresses
// come from hipGetSymbolAddress, once per device, and are kept in g_scratch_ptr.
static __device__ __align__(256) unsigned char scratch[LDPC_NOALLOC_SCRATCH_BYTES];

enum { MAX_DEVICES = 64 };
static unsigned char *g_scratch_ptr[MAX_DEVICES];
static std::mutex g_scratch_mutex;

// Address of `scratch` on `device`, which must be the current device.
static int scratch_address(int device, unsigned char **out)
{
    if (device < 0 || device >= MAX_DEVICES) return fail(LDPC_E_ARG, "device %d beyond the %d this library tracks", device, (int)MAX_DEVICES);
    std::lock_guard<std::mutex> g(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return fail(LDPC_E_DEVICE, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e));
        g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
    }
    *out = g_scratch_ptr[device];
    return 0;
}

static int open_device(int device, ldpc_handle *h)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
        return fail(LDPC_E_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= cnt) return fail(LDPC_E_ARG, "device %d out of range (%d visible)", device, cnt);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    h->device = device;
    h->n_cu = prop.multiProcessorCount;
    h->lds_max = prop.sharedMemPerBlock;
    {
        std::lock_guard<std::mutex> g(g_pool_mutex);
        if (device < 64 && !g_pool[device].empty()) {
            const PooledParts p = g_pool[device].back();
            g_pool[device].pop_back();
            h->own_stream = p.stream;
            h->arena = p.arena;
            h->arena_cap = p.cap;
            DevBuf *mine[POOL_BUFS];
            scratch_buffers(h, mine);
            for (int k = 0; k < POOL_BUFS; ++k) *mine[k] = p.bufs[k];
        }
    }
    if (!h->own_stream) HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    return scratch_address(device, &h->scratch);     // with `device` current
}

extern "C" const char *ldpc_last_error(void) { return g_err; }

extern "C" int ldpc_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

extern "C" int ldpc_release_pool(void)
{
    // what destroyed handles left behind for their successors: streams, table allocations, small buffers
    std::vector<PooledParts> all;
    std::vector<int> dev;
    {
        std::lock_guard<std::mutex> g(g_pool_mutex);
        for (int d = 0; d < 64; ++d) {
            for (PooledParts &p : g_pool[d]) { all.push_back(p); dev.push_back(d); }
            g_pool[d].clear();
        }
    }
    for (size_t k = 0; k < all.size(); ++k) {
        if (hipSetDevice(dev[k]) != hipSuccess) continue;
        PooledParts &p = all[k];
        if (p.stream) { (void)hipStreamSynchronize(p.stream); (void)hipStreamDestroy(p.stream); }
        if (p.arena) (void)hipFree(p.arena);
        for (DevBuf &b : p.bufs) b.release();
    }
    return (int)all.size();
}

extern "C" void ldpc_destroy(ldpc_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->own_stream) { (void)hipStreamSynchronize(h->own_stream); }
    if (h->stream != h->own_stream) (void)hipStreamSynchronize(h->stream);     // nothing may still read the tables
    bool pooled = false;
    if (h->own_stream && h->device < 64) {
        std::lock_guard<std::mutex> g(g_pool_mutex);
        if (g_pool[h->device].size() < POOL_KEEP) {
            PooledParts p{h->own_stream, h->arena, h->arena_cap, {}};
            DevBuf *mine[POOL_BUFS];
            scratch_buffers(h, mine);
            for (int k = 0; k < POOL_BUFS; ++k)
                if (mine[k]->cap <= (size_t)POOL_BUF_MAX) { p.bufs[k] = *mine[k]; *mine[k] = DevBuf(); }
            g_pool[h->device].push_back(p);
            pooled = true;
        }
    }
    if (pooled) { h->own_stream = nullptr; h->arena = nullptr; }
    if (h->arena) (void)hipFree(h->arena);
    if (h->edge_arena) (void)hipFree(h->edge_arena);
    DevBuf *bufs[POOL_BUFS];
    scratch_buffers(h, bufs);
    for (DevBuf *b : bufs) b->release();
    h->b_chain32.release();
    h->b_chain64.release();
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    for (HostSlot &sl : h->slot) {
        DevBuf *sb[] = {&sl.llr, &sl.hard, &sl.iters, &sl.conv, &sl.ones, &sl.soft};
        for (DevBuf *b : sb) b->release();
        sl.in.release();
        sl.out.release();
        if (sl.up) (void)hipEventDestroy(sl.up);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

// Column-side tables of a row-major edge list (rows ascending => rows ascending inside
// each column, the order the reference's variable nodes add in, ldpc.py:240-244) and upload.
static int build_csr(ldpc_handle *h, const std::vector<int32_t> &rp, const std::vector<int32_t> &ci, int m, int n)
{
    const int E = rp[m];
    std::vector<int32_t> cp(n + 1, 0), crow(E), ce(E);
    for (int e = 0; e < E; ++e) cp[ci[e] + 1]++;
    for (int c = 0; c < n; ++c) cp[c + 1] += cp[c];
    std::vector<int32_t> fill(cp.begin(), cp.end() - 1), eslot(E);
    for (int r = 0; r < m; ++r)
        for (int e = rp[r]; e < rp[r + 1]; ++e) {
            const int pos = fill[ci[e]]++;
            crow[pos] = r;
            ce[pos] = e - rp[r];
            eslot[e] = pos - cp[ci[e]];
        }
    h->csr.m = m; h->csr.n = n; h->csr.E = E;
    h->max_col_deg = 0;
    for (int c = 0; c < n; ++c) h->max_col_deg = std::max(h->max_col_deg, cp[c + 1] - cp[c]);
    h->min_row_deg = m > 0 ? rp[1] - rp[0] : 0;
    for (int r = 0; r < m; ++r) h->min_row_deg = std::min(h->min_row_deg, rp[r + 1] - rp[r]);
    // records of edge_packed.h: posterior + one word per check, padded to an odd number of words (columns
    // of a circulant have equal degrees, so neighbouring lanes are one record apart: an even stride would
    // put them on half or a quarter of the LDS banks)
    std::vector<int32_t> crec(n);
    int words = 0;
    for (int c = 0; c < n; ++c) {
        crec[c] = words;
        words += (1 + (cp[c + 1] - cp[c])) | 1;
    }
    h->rec_words = words;
    int rc;
    if ((rc = upload(h, crec, &h->col_rec))) return rc;
    if ((rc = upload(h, eslot, &h->edge_slot))) return rc;
    if ((rc = upload(h, rp, &h->csr.row_ptr)) || (rc = upload(h, ci, &h->csr.col_idx)) ||
        (rc = upload(h, cp, &h->csr.col_ptr)) || (rc = upload(h, crow, &h->csr.col_row)) ||
        (rc = upload(h, ce, &h->csr.col_e)))
        return rc;
    h->edges_ready = true;
    return 0;
}

// A block-circulant code as an edge list, columns ascending inside a row: staged, not yet on the device.
static int stage_qc_edges(ldpc_handle *h, const int32_t *blk_ptr, const int32_t *shifts)
{
    const int Z = h->qc.Z, mb = h->qc.mb, nb = h->qc.nb;
    std::vector<int32_t> rp((size_t)h->m + 1, 0), ci;
    ci.reserve((size_t)h->E);
    std::vector<int32_t> cols;
    for (int b = 0; b < mb; ++b)
        for (int i = 0; i < Z; ++i) {
            cols.clear();
            for (int j = 0; j < nb; ++j)
                for (int k = blk_ptr[b * nb + j]; k < blk_ptr[b * nb + j + 1]; ++k)
                    cols.push_back(j * Z + (i + shifts[k]) % Z);
            std::sort(cols.begin(), cols.end());
            ci.insert(ci.end(), cols.begin(), cols.end());
            rp[(size_t)b * Z + i + 1] = (int32_t)ci.size();
        }
    return build_csr(h, rp, ci, h->m, h->n);
}

// The edge list on the device, for the kernels that walk it (table-driven min-sum, sum-product outside
// the message planes, the short-code kernel).  may_allocate false: the caller promised not to allocate.
static int edge_tables(ldpc_handle *h, bool may_allocate)
{
    if (h->edges_ready) return 0;
    if (!may_allocate)
        return fail(LDPC_E_UNSUPPORTED, "this handle's edge tables are not on the device yet and building them allocates: "
                                        "decode once through ldpc_decode_device with the same algorithm and precision first");
    HIP_TRY(hipSetDevice(h->device));
    int rc = stage_qc_edges(h, h->qc_blk_ptr.data(), h->qc_shifts.data());
    if (rc || (rc = commit_tables(h, &h->edge_arena, &h->edge_arena_cap))) {
        h->edges_ready = false;
        h->stage.clear();
        h->fixups.clear();
        return rc;
    }
    h->qc_blk_ptr.clear(); h->qc_blk_ptr.shrink_to_fit();
    h->qc_shifts.clear(); h->qc_shifts.shrink_to_fit();
    return 0;
}

extern "C" int ldpc_create_qc(int device, int Z, int mb, int nb, const int32_t *blk_ptr,
                              const int32_t *shifts, ldpc_handle **out)
{
    if (!out) return fail(LDPC_E_ARG, "out is NULL");
    *out = nullptr;
    if (Z <= 0 || mb <= 0 || nb <= 0 || !blk_ptr) return fail(LDPC_E_ARG, "bad QC dimensions");
    const int nblk = mb * nb;
    if (blk_ptr[0] != 0) return fail(LDPC_E_ARG, "blk_ptr[0] must be 0");
    for (int k = 0; k < nblk; ++k)
        if (blk_ptr[k + 1] < blk_ptr[k]) return fail(LDPC_E_ARG, "blk_ptr must be non-decreasing");
    if (blk_ptr[nblk] > 0 && !shifts) return fail(LDPC_E_ARG, "shifts is NULL");
    const int Zp = (Z + 63) / 64 * 64;
    if ((long long)mb * Zp > 1024)
        return fail(LDPC_E_UNSUPPORTED, "mb*roundup(Z,64) = %lld exceeds 1024 threads", (long long)mb * Zp);
    if ((nb + mb - 1) / mb > 32)
        return fail(LDPC_E_UNSUPPORTED, "more than 32 block-columns per block-row group");

    std::vector<int32_t> row_ptr(mb + 1, 0), col_ptr(nb + 1, 0);
    std::vector<int2> row_ent, col_ent;
    std::vector<int4> col_grp;
    // edge number of shift k of block (b, j) inside the row list of block-row b
    std::vector<int> edge_no(blk_ptr[nblk] > 0 ? blk_ptr[nblk] : 1, 0);
    int max_deg = 0;
    for (int b = 0; b < mb; ++b) {
        int cnt = 0;
        for (int j = 0; j < nb; ++j) {
            const int k0 = blk_ptr[b * nb + j], k1 = blk_ptr[b * nb + j + 1];
            for (int k = k0; k < k1; ++k) {
                const int s = shifts[k];
                if (s < 0 || s >= Z) return fail(LDPC_E_ARG, "shift %d out of range [0,%d)", s, Z);
                for (int k2 = k0; k2 < k; ++k2)
                    if (shifts[k2] == s) return fail(LDPC_E_ARG, "duplicate shift %d in block (%d,%d)", s, b, j);
                edge_no[k] = cnt++;
                row_ent.push_back(make_int2(j * Z, s));
            }
        }
        row_ptr[b + 1] = (int)row_ent.size();
        max_deg = std::max(max_deg, cnt);
    }
    const bool wide = max_deg > 128;     // beyond the block-circulant kernels' rows (128 sign bits): decoded from the edge list only
    for (int j = 0; j < nb; ++j) {
        for (int b = 0; b < mb; ++b) {
            const int k0 = blk_ptr[b * nb + j], k1 = blk_ptr[b * nb + j + 1];
            if (k1 == k0) continue;
            std::vector<int2> ent;
            for (int k = k0; k < k1; ++k) ent.push_back(make_int2(shifts[k], edge_no[k]));
            std::sort(ent.begin(), ent.end(), [](const int2 &x, const int2 &y) { return x.x > y.x; });
            col_grp.push_back(make_int4(b, k1 - k0, (int)col_ent.size(), 0));
            col_ent.insert(col_ent.end(), ent.begin(), ent.end());
        }
        col_ptr[j + 1] = (int)col_grp.size();
    }

    ldpc_handle *h = new (std::nothrow) ldpc_handle();
    if (!h) return fail(LDPC_E_NOMEM, "out of host memory");
    int rc = open_device(device, h);
    if (rc) { ldpc_destroy(h); return rc; }      // (returns what open_device took from the pool)
    h->is_qc = !wide;
    h->wide = max_deg > 64;
    h->m = mb * Z; h->n = nb * Z; h->E = blk_ptr[nblk] * Z; h->max_row_deg = max_deg;
    h->cols_per_thread = (nb + mb - 1) / mb;
    h->qc.Z = Z; h->qc.Zp = Zp; h->qc.mb = mb; h->qc.nb = nb; h->qc.m = h->m; h->qc.n = h->n;
    h->qc.n_ent = (int)row_ent.size(); h->qc.n_grp = (int)col_grp.size();
    if (wide) {
        if ((rc = stage_qc_edges(h, blk_ptr, shifts)) || (rc = commit_tables(h, &h->arena, &h->arena_cap))) {
            ldpc_destroy(h);
            return rc;
        }
        *out = h;
        return 0;
    }
    // regular structure?  (same weight W in {1,2} for every non-empty block, and a block
    // geometry one of the unrolled instantiations covers)
    {
        int w = 0;
        bool regular = true, holes = false;
        for (int k = 0; k < nblk && regular; ++k) {
            const int wk = blk_ptr[k + 1] - blk_ptr[k];
            if (wk == 0) { holes = true; continue; }
            if (w == 0) w = wk;
            if (wk != w) regular = false;
        }
        const bool geom = (mb == 2 && nb == 16 && w == 2) || (mb == 4 && nb == 24 && w == 1);
        const char *force = hook("LDPC_HIP_FORCE_GENERIC");
        if (regular && geom && !(force && force[0] == '1')) {
            std::vector<int32_t> sh((size_t)mb * nb * w, 0);
            for (int b = 0; b < mb; ++b)
                for (int j = 0; j < nb; ++j) {
                    const int k0 = blk_ptr[b * nb + j], k1 = blk_ptr[b * nb + j + 1];
                    if (k1 == k0) continue;
                    std::vector<int32_t> v(shifts + k0, shifts + k1);
                    std::sort(v.begin(), v.end());
                    for (int k = 0; k < w; ++k) sh[((size_t)b * nb + j) * w + k] = v[k];
                    h->reg.row_mask[b] |= 1u << j;
                }
            h->is_regular = true;
            h->reg_holes = holes;
            h->reg_w = w;
            h->reg.Z = Z; h->reg.Zp = Zp; h->reg.nb = nb; h->reg.n = h->n;
            if ((rc = upload(h, sh, &h->reg.shifts))) { ldpc_destroy(h); return rc; }
        }
    }
    if ((rc = upload(h, row_ptr, &h->qc.row_ptr)) || (rc = upload(h, row_ent, &h->qc.row_ent)) ||
        (rc = upload(h, col_ptr, &h->qc.col_ptr)) || (rc = upload(h, col_grp, &h->qc.col_grp)) ||
        (rc = upload(h, col_ent, &h->qc.col_ent))) {
        ldpc_destroy(h);
        return rc;
    }
    if (mb == 2 && nb == 16 && Z == 511) {
        // message-plane kernel: weights 0..2 directly; up to PLANES_MAXW with the shifts beyond the two
        // lowest as "extra edges", at most PLANES_MAXX of them per block-row
        bool fits = true;
        int n_extra[2] = {0, 0}, h_max_w = 0;
        for (int b = 0; b < mb; ++b)
            for (int j = 0; j < nb; ++j) {
                const int w = blk_ptr[b * nb + j + 1] - blk_ptr[b * nb + j];
                fits = fits && w <= PLANES_MAXW;
                h_max_w = std::max(h_max_w, w);
                n_extra[b] += std::max(0, w - 2);
            }
        fits = fits && n_extra[0] <= PLANES_MAXX && n_extra[1] <= PLANES_MAXX;
        for (int g = 0; g < 2; ++g) {               // heavy block-columns per column group
            int cnt = 0;
            for (int j = g; j < nb; j += 2)
                cnt += (blk_ptr[j + 1] - blk_ptr[j] > 2) || (blk_ptr[nb + j + 1] - blk_ptr[nb + j] > 2);
            fits = fits && cnt <= PLANES_MAXH;
        }
        if (fits) {
            std::vector<int32_t> sh((size_t)mb * nb * 2, 0);
            int32_t ent[16][2][8];                   // [block-column][block-row]: the 8 table words (qc_planes.h)
            uint32_t heavy_cols = 0;
            int nx[2] = {0, 0};
            for (int b = 0; b < mb; ++b)
                for (int j = 0; j < nb; ++j) {
                    std::vector<int32_t> v(shifts + blk_ptr[b * nb + j], shifts + blk_ptr[b * nb + j + 1]);
                    std::sort(v.begin(), v.end());
                    const int w = (int)v.size();
                    const int row0 = (b == 0 ? 0 : n_extra[0]) + nx[b];       // scratch row of this block's first extra edge
                    ent[j][b][0] = w | ((w > 0 ? v[0] : 511) << 8) | ((w > 1 ? v[1] : 511) << 20) | (b == 0 ? (j >> 1) << 29 : 0);
                    for (int k = 2; k < PLANES_MAXW; ++k)
                        ent[j][b][k - 1] = (k < w ? v[k] : 511) | ((k < w ? row0 + k - 2 : (w > 2 ? row0 : 0)) << 16);
                    ent[j][b][7] = 0;
                    for (int k = 0; k < w; ++k) {
                        if (k < 2) {
                            sh[((size_t)b * nb + j) * 2 + k] = v[k];
                            h->planes.edge_mask[b] |= 1u << (2 * j + k);
                        } else {
                            h->planes.extra[b][nx[b]++] = j | (v[k] << 8);
                            heavy_cols |= 1u << j;
                        }
                    }
                }
            h->planes.n_extra[0] = n_extra[0];
            h->planes.n_extra[1] = n_extra[1];
            h->planes.xstride = (uint32_t)(n_extra[0] + n_extra[1]) * PLANES_XROW;
            for (int j = 0; j < nb; ++j)
                if ((heavy_cols >> j) & 1u) {
                    const int g = j & 1, hh = h->planes.n_heavy[g]++;
                    memcpy(h->planes.hent[g][hh], ent[j], sizeof(int32_t) * 16);
                }
            if ((rc = upload(h, sh, &h->planes.shifts))) { ldpc_destroy(h); return rc; }
            h->planes_ok = true;
            h->planes_heavy = heavy_cols != 0;
            h->planes_max_w = h_max_w;
        }
    }
    // heavy-circulant kernel (qc_heavy.h): float32 min-sum on what the message-plane and the regular kernels
    // do not take -- checks of up to 128 edges, circulants of weight <= HQ_WMAX
    {
        const char *force = hook("LDPC_HIP_FORCE_HEAVYQC"), *off = hook("LDPC_HIP_NO_HEAVYQC");
        const bool forced = force && force[0] == '1';
        int max_w = 0, min_deg = max_deg;
        for (int b = 0; b < mb; ++b) {
            min_deg = std::min(min_deg, blk_ptr[(b + 1) * nb] - blk_ptr[b * nb]);
            for (int j = 0; j < nb; ++j) max_w = std::max(max_w, blk_ptr[b * nb + j + 1] - blk_ptr[b * nb + j]);
        }
        const int dpad_max = (max_deg + HQ_PAD - 1) / HQ_PAD * HQ_PAD;
        const int nwu = (dpad_max + 31) / 32;
        const HeavyGeom geo(Z, Zp, mb, nb, nwu);
        // one frame per workgroup: worth it where the table-driven kernel holds one frame per workgroup too
        // (and where the short-code kernels do not apply: they take checks of up to 24 edges)
        const bool wanted = forced || (!h->planes_ok && !h->is_regular && mb * Zp > 512 && max_deg > 24);
        if (wanted && !(off && off[0] == '1') && mb <= HQ_MAXMB && h->cols_per_thread <= HQ_CPT && max_w <= HQ_WMAX &&
            min_deg >= 2 && max_deg <= 32 * HQ_NW && Z <= 1023 && (size_t)geo.bytes <= h->lds_max) {
            HeavyTables &t = h->heavyqc;
            t.Z = Z; t.Zp = Zp; t.mb = mb; t.nb = nb; t.n = h->n; t.nwu = nwu;
            std::vector<uint32_t> rowtab((size_t)2 * mb * geo.tabw, geo.dummy_off);
            auto table_word = [](uint32_t s, uint32_t e) { return (s << 18) | (e << 5) | (31u - (e & 31u)); };
            for (int b = 0; b < mb; ++b) {
                t.deg[b] = row_ptr[b + 1] - row_ptr[b];
                t.dpad[b] = (t.deg[b] + HQ_PAD - 1) / HQ_PAD * HQ_PAD;
                for (int e = 0; e < t.deg[b]; ++e) {
                    const int2 en = row_ent[(size_t)row_ptr[b] + e];           // { j * Z, s }
                    rowtab[(size_t)b * geo.tabw + e] = (uint32_t)((en.x / Z) * (int)geo.Z2 + en.y) * 4u;
                    rowtab[(size_t)(mb + b) * geo.tabw + e] = table_word((uint32_t)en.y, (uint32_t)e);
                }
            }
            const size_t blkw = geo.split ? 2 * HQ_BLK : HQ_BLK;
            const size_t gstride = (size_t)(HQ_CPT * mb + 2) * blkw;           // words per thread group: its blocks and two spare ones
            std::vector<uint32_t> coltab((size_t)mb * gstride, 0u);            // (uploads start on 256-byte boundaries)
            for (int g = 0; g < mb; ++g) {
                t.ncol[g] = 0;
                for (int q = 0; q < HQ_CPT; ++q) {
                    const int j = g + q * mb;
                    if (j >= nb) continue;
                    t.ncol[g] = q + 1;
                    for (int b = 0; b < mb; ++b) {
                        const int k0 = blk_ptr[b * nb + j], k1 = blk_ptr[b * nb + j + 1];
                        std::vector<int2> ent;
                        for (int k = k0; k < k1; ++k) ent.push_back(make_int2(shifts[k], edge_no[k]));
                        std::sort(ent.begin(), ent.end(), [](const int2 &x, const int2 &y) { return x.x > y.x; });
                        uint32_t *blk = &coltab[(size_t)g * gstride + ((size_t)q * mb + b) * blkw];
                        blk[14] = (uint32_t)ent.size();
                        for (size_t k = 0; k < ent.size(); ++k) {
                            const uint32_t s = (uint32_t)ent[k].x, e = (uint32_t)ent[k].y, wd = e >> 5;
                            const uint32_t row = (uint32_t)b * geo.Z2 + (uint32_t)Z - s;
                            blk[2 * k] = geo.rec_off + (geo.split ? 0u : wd * geo.rec_wstride) + row * 16u;
                            blk[2 * k + 1] = table_word(s, e);
                            if (geo.split) blk[16 + k] = geo.sw_off + wd * geo.sw_wstride + row * 4u;
                        }
                    }
                }
            }
            if ((rc = upload(h, rowtab, &t.rowtab)) || (rc = upload(h, coltab, &t.coltab))) { ldpc_destroy(h); return rc; }
            h->heavyqc_ok = true;
            h->heavyqc_lds = geo.bytes;
        }
    }
    // column and row degrees, which kernel selection looks at
    h->max_col_deg = 0;
    for (int j = 0; j < nb; ++j) {
        int d = 0;
        for (int b = 0; b < mb; ++b) d += blk_ptr[b * nb + j + 1] - blk_ptr[b * nb + j];
        h->max_col_deg = std::max(h->max_col_deg, d);
    }
    h->min_row_deg = max_deg;
    for (int b = 0; b < mb; ++b) h->min_row_deg = std::min(h->min_row_deg, blk_ptr[(b + 1) * nb] - blk_ptr[b * nb]);
    // the same graph as an edge list: now, unless the message-plane kernels take the code (then on first need)
    {
        const char *eager = hook("LDPC_HIP_EAGER_EDGES");
        const char *force_hq = hook("LDPC_HIP_FORCE_HEAVYQC");
        const bool heavy_by_choice = h->heavyqc_ok && !(force_hq && force_hq[0] == '1');
        if ((h->planes_ok || heavy_by_choice) && !(eager && eager[0] == '1')) {
            h->qc_blk_ptr.assign(blk_ptr, blk_ptr + nblk + 1);
            h->qc_shifts.assign(shifts, shifts + blk_ptr[nblk]);
        } else if ((rc = stage_qc_edges(h, blk_ptr, shifts))) {
            ldpc_destroy(h);
            return rc;
        }
    }
    if ((rc = commit_tables(h, &h->arena, &h->arena_cap))) { ldpc_destroy(h); return rc; }
    *out = h;
    return 0;
}

extern "C" int ldpc_create_csr(int device, int m, int n, const int32_t *row_ptr,
                               const int32_t *col_idx, ldpc_handle **out)
{
    if (!out) return fail(LDPC_E_ARG, "out is NULL");
    *out = nullptr;
    if (m <= 0 || n <= 0 || !row_ptr) return fail(LDPC_E_ARG, "bad CSR dimensions");
    if (row_ptr[0] != 0) return fail(LDPC_E_ARG, "row_ptr[0] must be 0");
    int max_deg = 0;
    for (int r = 0; r < m; ++r) {
        if (row_ptr[r + 1] < row_ptr[r]) return fail(LDPC_E_ARG, "row_ptr must be non-decreasing");
        max_deg = std::max(max_deg, row_ptr[r + 1] - row_ptr[r]);
    }
    const int E = row_ptr[m];
    if (E > 0 && !col_idx) return fail(LDPC_E_ARG, "col_idx is NULL");
    for (int r = 0; r < m; ++r)
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
            if (col_idx[e] < 0 || col_idx[e] >= n) return fail(LDPC_E_ARG, "column index out of range");
            if (e > row_ptr[r] && col_idx[e] <= col_idx[e - 1])
                return fail(LDPC_E_ARG, "col_idx must be strictly ascending inside a row");
        }
    std::vector<int32_t> rp(row_ptr, row_ptr + m + 1), ci(col_idx, col_idx + E);
    ldpc_handle *h = new (std::nothrow) ldpc_handle();
    if (!h) return fail(LDPC_E_NOMEM, "out of host memory");
    int rc = open_device(device, h);
    if (rc) { ldpc_destroy(h); return rc; }      // (returns what open_device took from the pool)
    h->is_qc = false;
    h->wide = max_deg > 64;
    h->m = m; h->n = n; h->E = E; h->max_row_deg = max_deg;
    if ((rc = build_csr(h, rp, ci, m, n)) || (rc = commit_tables(h, &h->arena, &h->arena_cap))) {
        ldpc_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

extern "C" int ldpc_get_dims(const ldpc_handle *h, int *m, int *n, int *edges)
{
    if (!h) return fail(LDPC_E_ARG, "handle is NULL");
    if (m) *m = h->m;
    if (n) *n = h->n;
    if (edges) *edges = h->E;
    return 0;
}

extern "C" int ldpc_set_minsum_scaling(ldpc_handle *h, float alpha, float beta)
{
    if (!h) return fail(LDPC_E_ARG, "handle is NULL");
    if (!(alpha > 0.0f) || !(beta >= 0.0f) || !(alpha <= 1e6f) || !(beta <= 1e6f))
        return fail(LDPC_E_ARG, "need alpha > 0 and beta >= 0");
    h->ms_alpha = alpha;
    h->ms_beta = beta;
    return 0;
}

extern "C" int ldpc_set_pass_counters(ldpc_handle *h, uint64_t *d_counts)
{
    if (!h) return fail(LDPC_E_ARG, "handle is NULL");
    h->pass_counts = reinterpret_cast<unsigned long long *>(d_counts);
    return 0;
}

extern "C" int ldpc_set_stream(ldpc_handle *h, void *hip_stream, int use_own)
{
    if (!h) return fail(LDPC_E_ARG, "handle is NULL");
    hipStream_t next = use_own ? h->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
    if (next != h->stream) {
        // the frame-claim counter and the workspaces are per handle: what is queued on the old stream
        // must have finished with them before a launch on the new one resets them
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->stream = next;
    return 0;
}

extern "C" int ldpc_sync(ldpc_handle *h)
{
    if (!h) return fail(LDPC_E_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// ---------------------------------------------------------------- kernel selection
template <typename T, int S64, int CPT, int FPW>
static size_t qc_lds_bytes(const QcTables &tb)
{
    const size_t slots = (size_t)FPW * tb.mb * tb.Zp;
    size_t b = (((size_t)FPW * tb.n * sizeof(T)) + 15) & ~(size_t)15;
    b += slots * sizeof(typename Vec2<T>::type);
    b += (slots * (size_t)RowMeta<S64>::MW * 4 + 15) & ~(size_t)15;
    b += 3 * FPW * sizeof(int);
    b = (b + 15) & ~(size_t)15;
    b += qc_table_bytes(tb);
    return (b + 15) & ~(size_t)15;
}

template <typename T, int S64, int CPT, int FPW>
static int qc_pick(const ldpc_handle *h, Launch *L)
{
    L->fn = reinterpret_cast<const void *>(&qc_minsum_kernel<T, S64, CPT, FPW>);
    L->block = FPW * h->qc.mb * h->qc.Zp;
    L->fpw = FPW;
    L->lds = qc_lds_bytes<T, S64, CPT, FPW>(h->qc);
    return 0;
}

template <typename T, int S64>
static int qc_select(const ldpc_handle *h, bool chained, Launch *L)
{
    const int cpt = h->cols_per_thread;
    const int thr1 = h->qc.mb * h->qc.Zp;
    if (cpt <= 8 && chained) return qc_pick<T, S64, 8, 1>(h, L);
    if (cpt <= 8) {
        // pack several short frames into one workgroup while threads and LDS allow
        if (thr1 * 4 <= 1024 && qc_lds_bytes<T, S64, 8, 4>(h->qc) <= h->lds_max / 2) return qc_pick<T, S64, 8, 4>(h, L);
        if (thr1 * 2 <= 1024 && qc_lds_bytes<T, S64, 8, 2>(h->qc) <= h->lds_max / 2) return qc_pick<T, S64, 8, 2>(h, L);
        return qc_pick<T, S64, 8, 1>(h, L);
    }
    if (cpt <= 16) return qc_pick<T, S64, 16, 1>(h, L);
    return qc_pick<T, S64, 32, 1>(h, L);
}

template <typename T, int MB, int W, int CPT, int FPW>
static int reg_pick(const ldpc_handle *h, Launch *L)
{
    L->fn = h->reg_holes ? reinterpret_cast<const void *>(&qc_regular_kernel<T, MB, W, CPT, FPW, true>)
                         : reinterpret_cast<const void *>(&qc_regular_kernel<T, MB, W, CPT, FPW, false>);
    L->block = FPW * MB * h->reg.Zp;
    L->fpw = FPW;
    const size_t slots = (size_t)FPW * MB * h->reg.Zp;
    size_t b = (((size_t)FPW * h->reg.n * sizeof(T)) + 15) & ~(size_t)15;
    b += slots * RegState<T>::BYTES + (3 * FPW + 1) * sizeof(int);
    L->lds = (b + 15) & ~(size_t)15;
    return 0;
}

// float32 NearEarth-shaped codes: explicit messages in LDS (qc_planes.h)
static bool planes_applies(const ldpc_handle *h, int prec, bool chained, bool spa = false)
{
    const char *off = hook("LDPC_HIP_NO_PLANES");
    (void)spa;
    return h->planes_ok && prec == 32 && !chained &&
           (size_t)PlanesGeom<511, 8>::LDS_BYTES <= h->lds_max && !(off && off[0] == '1');
}

static int planes_pick(const ldpc_handle *h, bool spa, Launch *L)
{
    // LDPC_HIP_FORCE_HEAVY=1 (test hook): the heavy-circulant kernel for a code without heavy circulants
    const char *force = hook("LDPC_HIP_FORCE_HEAVY");
    if (h->planes_heavy || (force && force[0] == '1')) {
        const bool small = h->planes.n_extra[0] <= PlanesHeavy<1>::MX && h->planes.n_extra[1] <= PlanesHeavy<1>::MX &&
                           h->planes_max_w <= PlanesHeavy<1>::MW;
        if (spa) {
            L->fn = small ? reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, true, false, 1>)
                          : reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, true, false, 2>);
            L->fn_gen = small ? reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, true, true, 1>)
                              : reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, true, true, 2>);
        } else {
            L->fn = small ? reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, false, false, 1>)
                          : reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, false, false, 2>);
            L->fn_gen = small ? reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, false, true, 1>)
                              : reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, false, true, 2>);
        }
        L->ws_stride = std::max<size_t>((size_t)h->planes.xstride * sizeof(float), 256);
    } else {
        L->fn = spa ? reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, true, false, 0>)
                    : reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, false, false, 0>);
        L->fn_gen = spa ? reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, true, true, 0>)
                        : reinterpret_cast<const void *>(&qc_planes_kernel<511, 8, false, true, 0>);
    }
    L->planes = true;
    L->block = 2 * PlanesGeom<511, 8>::ZP;
    L->fpw = 1;
    L->lds = PlanesGeom<511, 8>::LDS_BYTES;
    return 0;
}

template <typename T>
static int reg_select(const ldpc_handle *h, bool chained, Launch *L)
{
    if (h->qc.mb == 2) return reg_pick<T, 2, 2, 8, 1>(h, L);
    if (chained) return reg_pick<T, 4, 1, 6, 1>(h, L);
    return reg_pick<T, 4, 1, 6, 2>(h, L);
}

static size_t csr_ws_stride(const ldpc_handle *h, int prec)
{
    const size_t ts = prec == 64 ? 8 : 4;
    if (h->wide) return ((((size_t)2 * h->n + (size_t)h->E) * ts) + 255) & ~(size_t)255;     // posteriors, channel values, one message per edge
    size_t b = (((size_t)2 * h->n * ts) + 15) & ~(size_t)15;
    b += (size_t)h->m * 2 * ts;
    b += (size_t)h->m * 16;
    return (b + 255) & ~(size_t)255;
}

// Short codes, float32: several frames per workgroup with explicit messages (edge_slots.h).
// Message slots per column record: the instantiated class that covers the largest column degree.
static int slots_dv(const ldpc_handle *h)
{
    return h->max_col_deg <= 4 ? 4 : h->max_col_deg <= 8 ? 8 : h->max_col_deg <= 12 ? 12 : 0;
}

static int slots_frames(const ldpc_handle *h)
{
    const int dv = slots_dv(h);
    if (h->m <= 0 || h->m > 1024 || dv == 0) return 0;
    int F = std::min(1024 / h->m, 32);
    while (F > 0 && SlotGeom(h->n, F, dv).bytes > h->lds_max) --F;
    return F;
}

template <bool SPA>
static const void *slots_fn(int d, int dv)
{
    if (dv == 4)
        return d <= 8    ? reinterpret_cast<const void *>(&slot_minsum_kernel<8, 4, SPA>)
               : d <= 12 ? reinterpret_cast<const void *>(&slot_minsum_kernel<12, 4, SPA>)
               : d <= 16 ? reinterpret_cast<const void *>(&slot_minsum_kernel<16, 4, SPA>)
               : d <= 20 ? reinterpret_cast<const void *>(&slot_minsum_kernel<20, 4, SPA>)
               : d <= 24 ? reinterpret_cast<const void *>(&slot_minsum_kernel<24, 4, SPA>) : nullptr;
    if (dv == 8)
        return d <= 12   ? reinterpret_cast<const void *>(&slot_minsum_kernel<12, 8, SPA>)
               : d <= 16 ? reinterpret_cast<const void *>(&slot_minsum_kernel<16, 8, SPA>)
               : d <= 24 ? reinterpret_cast<const void *>(&slot_minsum_kernel<24, 8, SPA>) : nullptr;
    if (dv == 12)
        return d <= 8    ? reinterpret_cast<const void *>(&slot_minsum_kernel<8, 12, SPA>)
               : d <= 16 ? reinterpret_cast<const void *>(&slot_minsum_kernel<16, 12, SPA>) : nullptr;
    return nullptr;
}

static bool slots_applies(const ldpc_handle *h, int prec, bool chained)
{
    const char *off = hook("LDPC_HIP_NO_SLOTS");
    if (prec != 32 || chained || (off && off[0] == '1')) return false;
    if (h->min_row_deg < 2 || !slots_fn<false>(h->max_row_deg, slots_dv(h))) return false;
    const int F = slots_frames(h);
    if (F < 1) return false;
    const int nt = (F * h->m + 63) / 64 * 64;
    return (long long)F * h->n <= (long long)SlotGeom::YMAX * nt;
}

static int slots_pick(const ldpc_handle *h, bool spa, Launch *L)
{
    const int dv = slots_dv(h);
    L->fn = spa ? slots_fn<true>(h->max_row_deg, dv) : slots_fn<false>(h->max_row_deg, dv);
    L->fpw = slots_frames(h);
    L->block = (L->fpw * h->m + 63) / 64 * 64;
    L->lds = SlotGeom(h->n, L->fpw, dv).bytes;
    L->slots = true;
    return 0;
}

// Short codes of low row degree, float32 min-sum: records of the columns' own length, several frames per
// thread (edge_packed.h).  The plan: as many frames in LDS as fit, G groups of threads x RPT frames each.
struct PackedPlan { int F = 0, rpt = 0, block = 0; const void *fn = nullptr; size_t lds = 0; };

static const void *packed_fn(int d, int rpt)
{
#define PACKED_ROW(DCV)                                                                                   \
    (rpt == 2   ? reinterpret_cast<const void *>(&packed_minsum_kernel<DCV, 2>)                          \
     : rpt == 4 ? reinterpret_cast<const void *>(&packed_minsum_kernel<DCV, 4>) : nullptr)
    return d <= 8 ? PACKED_ROW(8) : d <= 12 ? PACKED_ROW(12) : d <= 16 ? PACKED_ROW(16) : d <= 20 ? PACKED_ROW(20)
           : d <= 24 ? PACKED_ROW(24) : nullptr;
#undef PACKED_ROW
}

static PackedPlan packed_plan(const ldpc_handle *h)
{
    PackedPlan best;
    if (!h->edges_ready || h->m <= 0 || h->m > 1024 || h->max_col_deg > 12 || h->min_row_deg < 2 || h->max_row_deg > 24) return best;
    const char *force = hook("LDPC_HIP_PACKED_RPT");
    for (int rpt : {2, 4}) {
        if (force && force[0] && atoi(force) != rpt) continue;
        for (int g = 1024 / h->m; g >= 1; --g) {
            const int F = g * rpt;
            if (F > 32) continue;
            const PackedGeom geo(h->rec_words, F, rpt);
            if (geo.bytes > h->lds_max) continue;
            const int per_thread = PackedGeom::YTOT / rpt;             // columns a thread can hold
            const int need_cols = (g * h->n + per_thread - 1) / per_thread;
            const int block = (std::max(g * h->m, need_cols) + 63) / 64 * 64;
            if (block > 1024) continue;
            if (F > best.F) {                      // most frames in flight; of equal plans the one with more row threads (the narrower groups)
                best.F = F; best.rpt = rpt; best.block = block; best.lds = geo.bytes;
                best.fn = packed_fn(h->max_row_deg, rpt);
            }
            break;                                 // smaller g only lowers F for this rpt
        }
    }
    if (!best.fn) best = PackedPlan();
    return best;
}

static bool packed_applies(const ldpc_handle *h, int prec, bool chained)
{
    const char *off = hook("LDPC_HIP_NO_PACKED"), *off2 = hook("LDPC_HIP_NO_SLOTS");
    if (prec != 32 || chained || (off && off[0] == '1') || (off2 && off2[0] == '1')) return false;
    // worth it where the other short-code kernel cannot keep a compute unit busy: fewer than half the frames per workgroup
    const int F = packed_plan(h).F;
    const char *force = hook("LDPC_HIP_PACKED_RPT");
    return F >= 2 && ((force && force[0]) || !slots_applies(h, prec, chained) || F >= 2 * slots_frames(h));
}

// Workgroups of `fn` that fit a CU: one attribute call and one occupancy query per
// (device, kernel, geometry) and process, not per decode call.
static int blocks_per_cu(int device, const void *fn, int block, size_t lds, int *out, bool needs_lds_at_zero = false)
{
    struct Key { int device; const void *fn; int block; size_t lds; int per_cu; };
    struct Attr { int device; const void *fn; size_t lds; };      // largest dynamic LDS size a kernel was set up for
    static std::mutex mu;
    static std::vector<Key> seen;
    static std::vector<Attr> attrs;
    std::lock_guard<std::mutex> g(mu);
    for (const Key &k : seen)
        if (k.device == device && k.fn == fn && k.block == block && k.lds == lds) { *out = k.per_cu; return 0; }
    if (needs_lds_at_zero) {
        // the message-plane kernels address their dynamic LDS from 0: true as long as they have no static LDS
        hipFuncAttributes at;
        HIP_TRY(hipFuncGetAttributes(&at, fn));
        if (at.sharedSizeBytes != 0)
            return fail(LDPC_E_DEVICE, "kernel built with %zu bytes of static LDS; its dynamic LDS would not start at 0", (size_t)at.sharedSizeBytes);
    }
    // The attribute belongs to the kernel, not to one launch geometry: codes that share a template instance
    // with different LDS sizes must leave it at the largest of them, so it is only ever raised.
    if (lds > 0) {
        Attr *at = nullptr;
        for (Attr &a : attrs)
            if (a.device == device && a.fn == fn) at = &a;
        if (!at || at->lds < lds) {
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (at) at->lds = lds;
            else attrs.push_back(Attr{device, fn, lds});
        }
    }
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, lds));
    if (per_cu < 1) per_cu = 1;
    seen.push_back(Key{device, fn, block, lds, per_cu});
    *out = per_cu;
    return 0;
}

// Kernel and geometry for (algo, prec, chained) on this handle's code; everything but the grid.
static int select_kernel(ldpc_handle *h, int algo, int prec, bool chained, Launch *L)
{
    int per_cu = 1;
    if (algo == LDPC_ALGO_SUMPRODUCT) {
        if (chained)
            return fail(LDPC_E_UNSUPPORTED, "the re-used-decoder mode exists for min-sum only");
        const char *edge_only = hook("LDPC_HIP_SPA_EDGE_LIST");
        const bool fast_ok = !(edge_only && edge_only[0] == '1');
        if (fast_ok && !(h->is_qc && planes_applies(h, prec, false, true)) && slots_applies(h, prec, false)) {
            // short codes, float32: several frames per workgroup (edge_slots.h)
            int rc = slots_pick(h, true, L);
            if (rc) return rc;
            if ((rc = blocks_per_cu(h->device, L->fn, L->block, L->lds, &per_cu, true))) return rc;
            L->max_grid = h->n_cu * per_cu;
            return 0;
        }
        if (h->is_qc && planes_applies(h, prec, false, true) && fast_ok) {
            // float32 NearEarth-shaped codes: message planes in LDS, hardware transcendentals
            int rc = planes_pick(h, true, L);
            if (rc) return rc;
            L->spa_planes = true;
            if ((rc = blocks_per_cu(h->device, L->fn, L->block, L->lds, &per_cu, true)) ||
                (rc = blocks_per_cu(h->device, L->fn_gen, L->block, L->lds, &per_cu, true)))
                return rc;
            L->max_grid = h->n_cu;
            return 0;
        }
        L->fn = prec == 32 ? reinterpret_cast<const void *>(&csr_spa_kernel<float>)
                           : reinterpret_cast<const void *>(&csr_spa_kernel<double>);
        L->block = 1024;
        L->fpw = 1;
        const size_t need = spa_state_bytes(h, prec);
        L->spa_lds = need + 64 <= h->lds_max;
        L->lds = L->spa_lds ? need : 0;
        int rc = blocks_per_cu(h->device, L->fn, L->block, L->lds, &per_cu);
        if (rc) return rc;
        L->max_grid = h->n_cu * per_cu;
        L->ws_stride = L->spa_lds ? 0 : ((spa_state_bytes(h, prec) + 255) & ~(size_t)255);
        return 0;
    }
    // LDPC_HIP_FORCE_HEAVYQC=1 (test hook): the heavy-circulant kernel wherever its tables were built
    const char *force_hq = hook("LDPC_HIP_FORCE_HEAVYQC");
    const bool heavy_ok = h->is_qc && h->heavyqc_ok && prec == 32 && !chained;
    const bool heavy_first = heavy_ok && force_hq && force_hq[0] == '1';
    if (!heavy_first && !(h->is_qc && planes_applies(h, prec, chained)) && packed_applies(h, prec, chained)) {
        const PackedPlan pl = packed_plan(h);
        L->fn = pl.fn; L->fpw = pl.F; L->block = pl.block; L->lds = pl.lds;
        L->slots = true; L->packed = true;
        int rc = blocks_per_cu(h->device, L->fn, L->block, L->lds, &per_cu, true);
        if (rc) return rc;
        L->max_grid = h->n_cu * per_cu;
        return 0;
    }
    if (!heavy_first && !(h->is_qc && planes_applies(h, prec, chained)) && slots_applies(h, prec, chained)) {
        int rc = slots_pick(h, false, L);
        if (rc) return rc;
        if ((rc = blocks_per_cu(h->device, L->fn, L->block, L->lds, &per_cu, true))) return rc;
        L->max_grid = h->n_cu * per_cu;
        return 0;
    }
    if (h->is_qc) {
        const int sb = h->max_row_deg > 64 ? 128 : h->max_row_deg > 32 ? 64 : 32;
        int rc;
        auto heavy_pick = [&]() {
            L->fn = heavyqc_kernel(h->heavyqc.nwu);
            L->block = h->qc.mb * h->qc.Zp; L->fpw = 1; L->lds = h->heavyqc_lds; L->heavyqc = true;
            return 0;
        };
        if (heavy_first) rc = heavy_pick();
        else if (planes_applies(h, prec, chained)) rc = planes_pick(h, false, L);
        else if (h->is_regular) rc = prec == 32 ? reg_select<float>(h, chained, L) : reg_select<double>(h, chained, L);
        else if (heavy_ok) rc = heavy_pick();
        else if (prec == 32) rc = sb == 128 ? qc_select<float, 128>(h, chained, L) : sb == 64 ? qc_select<float, 64>(h, chained, L) : qc_select<float, 32>(h, chained, L);
        else            rc = sb == 128 ? qc_select<double, 128>(h, chained, L) : sb == 64 ? qc_select<double, 64>(h, chained, L) : qc_select<double, 32>(h, chained, L);
        if (rc) return rc;
        if (L->lds <= h->lds_max) {
            if ((rc = blocks_per_cu(h->device, L->fn, L->block, L->lds, &per_cu, L->planes || L->heavyqc))) return rc;
            if (L->fn_gen && (rc = blocks_per_cu(h->device, L->fn_gen, L->block, L->lds, &per_cu, L->planes))) return rc;
            L->max_grid = h->n_cu * per_cu;
            return 0;
        }
        // the frame state does not fit a CU's LDS in this arithmetic: decode from the edge list
        // with the state in a global workspace (every QC handle carries its edge list too)
        *L = Launch();
    }
    if (h->wide)
        L->fn = prec == 32 ? reinterpret_cast<const void *>(&csr_wide_minsum_kernel<float>)
                           : reinterpret_cast<const void *>(&csr_wide_minsum_kernel<double>);
    else
        L->fn = prec == 32 ? reinterpret_cast<const void *>(&csr_minsum_kernel<float>)
                           : reinterpret_cast<const void *>(&csr_minsum_kernel<double>);
    L->csr = true;
    L->block = 256;
    L->fpw = 1;
    L->lds = 0;
    L->max_grid = h->n_cu * 4;
    L->ws_stride = csr_ws_stride(h, prec);
    return 0;
}

static int select_launch(ldpc_handle *h, int algo, int prec, int64_t n_frames, bool chained, Launch *L)
{
    if (algo != LDPC_ALGO_MINSUM && algo != LDPC_ALGO_SUMPRODUCT)
        return fail(LDPC_E_ARG, "algo must be 0 (min-sum) or 1 (sum-product)");
    if (prec != 32 && prec != 64) return fail(LDPC_E_ARG, "prec must be 32 or 64");
    HIP_TRY(hipSetDevice(h->device));
    Launch &c = h->launch_cache[algo][prec == 64][chained ? 1 : 0];
    if (!c.fn || g_test_hooks) {          // (a test hook may change the choice from call to call)
        Launch fresh;
        int rc = select_kernel(h, algo, prec, chained, &fresh);
        if (rc) return rc;
        c = fresh;
    }
    *L = c;
    const int64_t fblocks = (n_frames + L->fpw - 1) / L->fpw;
    L->grid = chained ? 1 : (int)std::max<int64_t>(1, std::min<int64_t>(fblocks, (int64_t)L->max_grid));
    return 0;
}

// Bytes of persistent check-node state of a chained decoder.
static size_t chain_bytes(const ldpc_handle *h, int prec)
{
    // edge-list kernel: its whole workspace slice persists; QC kernels: 48 bytes per row slot
    // cover the state layout of either of them.  A QC code whose frame state does not fit the
    // LDS runs on the edge-list kernel, so the buffer is sized for whichever is larger.
    const size_t qc = h->is_qc ? (size_t)h->qc.mb * h->qc.Zp * 48 : 0;
    return std::max(qc, csr_ws_stride(h, prec));
}

// Frames (or blocks of frames) are claimed from a zeroed device counter: they differ in
// iteration count, and a fixed share per workgroup leaves most CUs idle while the unluckiest
// one finishes.  LDPC_HIP_STATIC_FRAMES=1 (test hook) keeps the fixed stride.
// `fixed`: the counter's home when the caller may not allocate (the library's static array).
static int claim_counter(ldpc_handle *h, const Launch &L, DecodeArgs *a, unsigned char *fixed)
{
    const char *stat = hook("LDPC_HIP_STATIC_FRAMES");
    a->work = nullptr;
    if (a->n_frames >= (int64_t)0x7fffffff - 4 * (int64_t)L.grid) {
        if (fixed) return fail(LDPC_E_UNSUPPORTED, "n_frames %lld exceeds what the allocation-free entry point takes (2^31 - 1 - 4 * %d)",
                               (long long)a->n_frames, L.grid);
        return 0;
    }
    if (stat && stat[0] == '1') return 0;
    void *ctr = fixed;
    if (!ctr) {
        int rc = h->b_work.ensure(256);
        if (rc) return rc;
        ctr = h->b_work.p;
    }
    HIP_TRY(hipMemsetAsync(ctr, 0, 4, h->stream));
    a->work = reinterpret_cast<int *>(ctr);
    return 0;
}

// Bytes at the start of the static scratch array that belong to the frame-claim counter; the
// per-workgroup workspaces of the edge-list kernels follow.
enum { NOALLOC_COUNTER_BYTES = 256 };

// `fixed` == nullptr: scratch buffers come from the handle (allocated and grown on demand).
// Otherwise `fixed` points to `fixed_bytes` of device memory that this call may use as its
// scratch, and nothing is allocated: a kernel that needs a workspace per workgroup gets as
// many workgroups as fit (results do not depend on the grid), or the call fails.
static int launch_decode(ldpc_handle *h, int algo, int prec, const DecodeArgs &args,
                         unsigned char *fixed = nullptr, size_t fixed_bytes = 0)
{
    Launch L;
    int rc = select_launch(h, algo, prec, args.n_frames, args.chain != nullptr, &L);
    if (rc) return rc;
    if (!args.llr && !L.fn_gen) return fail(LDPC_E_UNSUPPORTED, "this kernel does not generate its own channel values");
    unsigned char *ws = nullptr;
    if (L.ws_stride && !(L.csr && args.chain)) {
        if (fixed) {
            const size_t room = fixed_bytes > NOALLOC_COUNTER_BYTES ? fixed_bytes - NOALLOC_COUNTER_BYTES : 0;
            const int64_t fit = (int64_t)(room / L.ws_stride);
            if (fit < 1)
                return fail(LDPC_E_UNSUPPORTED, "this code needs %zu bytes of workspace per workgroup; the allocation-free "
                            "entry point has %zu", L.ws_stride, room);
            L.grid = (int)std::min<int64_t>(L.grid, fit);
            ws = fixed + NOALLOC_COUNTER_BYTES;
        } else {
            if ((rc = h->b_ws.ensure(L.ws_stride * (size_t)L.grid))) return rc;
            ws = reinterpret_cast<unsigned char *>(h->b_ws.p);
        }
    }
    DecodeArgs a = args;
    a.pass_counts = h->pass_counts;
    if ((L.slots || L.csr || (algo == LDPC_ALGO_SUMPRODUCT && !L.planes)) && (rc = edge_tables(h, fixed == nullptr))) return rc;
    if (L.slots) {
        SlotTables tb;
        tb.m = h->m; tb.n = h->n; tb.E = h->E; tb.F = L.fpw;
        tb.row_ptr = h->csr.row_ptr; tb.col_idx = h->csr.col_idx; tb.edge_slot = h->edge_slot; tb.col_ptr = h->csr.col_ptr;
        tb.col_rec = h->col_rec; tb.rec_words = h->rec_words;
        if ((rc = claim_counter(h, L, &a, fixed))) return rc;
        void *kargs[] = {&tb, &a};
        HIP_TRY(hipLaunchKernel(L.fn, dim3(L.grid), dim3(L.block), kargs, L.lds, h->stream));
    } else if (L.planes) {
        PlanesTables tb = h->planes;
        tb.xmsg = reinterpret_cast<float *>(ws);          // heavy circulants: the workgroups' scratch rows
        if ((rc = claim_counter(h, L, &a, fixed))) return rc;
        void *kargs[] = {&tb, &a};
        HIP_TRY(hipLaunchKernel(a.llr ? L.fn : L.fn_gen, dim3(L.grid), dim3(L.block), kargs, L.lds, h->stream));
    } else if (algo == LDPC_ALGO_SUMPRODUCT) {
        size_t stride = L.ws_stride;
        CsrTables tb = h->csr;
        int use_lds = L.spa_lds ? 1 : 0;
        void *kargs[] = {&tb, &a, &ws, &stride, &use_lds};
        HIP_TRY(hipLaunchKernel(L.fn, dim3(L.grid), dim3(L.block), kargs, L.lds, h->stream));
    } else if (L.csr) {
        size_t stride = L.ws_stride;
        CsrTables tb = h->csr;
        // chained: the persistent buffer IS the workspace of the single workgroup
        if (args.chain) ws = args.chain;
        void *kargs[] = {&tb, &a, &ws, &stride};
        HIP_TRY(hipLaunchKernel(L.fn, dim3(L.grid), dim3(L.block), kargs, 0, h->stream));
    } else if (L.heavyqc) {
        if ((rc = claim_counter(h, L, &a, fixed))) return rc;
        HIP_TRY((hipError_t)heavyqc_launch(L.fn, h->heavyqc, a, L.grid, L.block, L.lds, h->stream));
    } else if (h->is_regular) {
        RegTables tb = h->reg;
        if (!a.chain && (rc = claim_counter(h, L, &a, fixed))) return rc;      // a re-used decoder takes its frames in order
        void *kargs[] = {&tb, &a};
        HIP_TRY(hipLaunchKernel(L.fn, dim3(L.grid), dim3(L.block), kargs, L.lds, h->stream));
    } else {
        QcTables tb = h->qc;
        void *kargs[] = {&tb, &a};
        HIP_TRY(hipLaunchKernel(L.fn, dim3(L.grid), dim3(L.block), kargs, L.lds, h->stream));
    }
    return 0;
}

static int check_common(ldpc_handle *h, int max_iters, int64_t n_frames, int sign_mode)
{
    if (!h) return fail(LDPC_E_ARG, "handle is NULL");
    if (max_iters < 0) return fail(LDPC_E_ARG, "max_iters must be >= 0");
    if (n_frames < 0) return fail(LDPC_E_ARG, "n_frames must be >= 0");
    if (sign_mode != LDPC_SIGN_REFERENCE && sign_mode != LDPC_SIGN_CORRECTED)
        return fail(LDPC_E_ARG, "sign_mode must be 0 or 1");
    return 0;
}

extern "C" int ldpc_get_launch_info(ldpc_handle *h, int algo, int prec, int *grid, int *block,
                                    int *lds_bytes, int *frames_per_wg)
{
    if (!h) return fail(LDPC_E_ARG, "handle is NULL");
    Launch L;
    int rc = select_launch(h, algo, prec, INT64_MAX / 2, false, &L);
    if (rc) return rc;
    if (grid) *grid = L.grid;
    if (block) *block = L.block;
    if (lds_bytes) *lds_bytes = (int)L.lds;
    if (frames_per_wg) *frames_per_wg = L.fpw;
    return 0;
}

extern "C" int ldpc_decode_device(ldpc_handle *h, int algo, int prec, int sign_mode, int max_iters,
                                  int64_t n_frames, const void *d_llr, int llr_bits, float llr_scale,
                                  uint8_t *d_hard_bits, int32_t *d_iters, uint8_t *d_converged,
                                  int32_t *d_ones, void *d_soft_out)
{
    int rc = check_common(h, max_iters, n_frames, sign_mode);
    if (rc) return rc;
    if (n_frames == 0) return 0;
    if (!d_llr) return fail(LDPC_E_ARG, "d_llr is NULL");
    if (llr_bits != 32 && llr_bits != 64) return fail(LDPC_E_ARG, "llr_bits must be 32 or 64");
    DecodeArgs a;
    a.llr = d_llr; a.llr_f64 = llr_bits == 64; a.llr_scale = llr_scale; a.n_frames = n_frames; a.max_iters = max_iters;
    a.sign_mode = sign_mode; a.hard = d_hard_bits; a.iters = d_iters; a.conv = d_converged;
    a.ones = d_ones; a.soft = d_soft_out; a.chain = nullptr; a.work = nullptr;
    a.ms_alpha = h->ms_alpha; a.ms_beta = h->ms_beta; a.precheck = precheck_override();
    return launch_decode(h, algo, prec, a);
}

extern "C" int ldpc_decode_device_noalloc(ldpc_handle *h, int algo, int prec, int sign_mode, int max_iters,
                                          int64_t n_frames, const void *d_llr, int llr_bits, float llr_scale,
                                          uint8_t *d_hard_bits, int32_t *d_iters, uint8_t *d_converged,
                                          int32_t *d_ones, void *d_soft_out)
{
    int rc = check_common(h, max_iters, n_frames, sign_mode);
    if (rc) return rc;
    if (n_frames == 0) return 0;
    if (!d_llr) return fail(LDPC_E_ARG, "d_llr is NULL");
    if (llr_bits != 32 && llr_bits != 64) return fail(LDPC_E_ARG, "llr_bits must be 32 or 64");
    if (!h->scratch) return fail(LDPC_E_DEVICE, "the static scratch array has no address on device %d", h->device);
    DecodeArgs a;
    a.llr = d_llr; a.llr_f64 = llr_bits == 64; a.llr_scale = llr_scale; a.n_frames = n_frames; a.max_iters = max_iters;
    a.sign_mode = sign_mode; a.hard = d_hard_bits; a.iters = d_iters; a.conv = d_converged;
    a.ones = d_ones; a.soft = d_soft_out; a.chain = nullptr; a.work = nullptr;
    a.ms_alpha = h->ms_alpha; a.ms_beta = h->ms_beta; a.precheck = precheck_override();
    return launch_decode(h, algo, prec, a, h->scratch, (size_t)LDPC_NOALLOC_SCRATCH_BYTES);
}

static int chain_state(ldpc_handle *h, int prec, unsigned char **out)
{
    DevBuf &b = prec == 64 ? h->b_chain64 : h->b_chain32;
    if (!b.p) {
        const size_t bytes = chain_bytes(h, prec);
        int rc = b.ensure(bytes);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(b.p, 0, bytes, h->stream));   // all-zero = fresh check nodes
    }
    *out = reinterpret_cast<unsigned char *>(b.p);
    return 0;
}

extern "C" int ldpc_reset_state(ldpc_handle *h)
{
    if (!h) return fail(LDPC_E_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    for (int prec : {32, 64}) {
        DevBuf &b = prec == 64 ? h->b_chain64 : h->b_chain32;
        if (b.p) HIP_TRY(hipMemsetAsync(b.p, 0, b.cap, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// Large host calls: the caller's (pageable) arrays are moved through two page-locked slots by
// several host threads; uploads run on their own stream, so that the host copy of chunk k+1, the
// DMA of chunk k and the decode of chunk k-1's successor overlap.  Decodes stay in order on the
// handle's stream (a re-used decoder's state passes from chunk to chunk).
static int decode_host_pipelined(ldpc_handle *h, unsigned char *chain, int algo, int prec, int sign_mode, int max_iters,
                                 int64_t n_frames, const void *llr, int llr_bits, uint8_t *hard_bits,
                                 int32_t *iters, uint8_t *converged, int32_t *ones, void *soft_out)
{
    const size_t n = (size_t)h->n, nbytes = (n + 7) / 8, ts = prec == 64 ? 8 : 4, ls = llr_bits / 8;
    const int64_t chunk = std::max<int64_t>(1, (int64_t)((64u << 20) / (n * ls)));
    auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_hard = 0, o_iters = up256(o_hard + (size_t)chunk * nbytes), o_conv = up256(o_iters + (size_t)chunk * 4),
                 o_ones = up256(o_conv + (size_t)chunk), o_soft = up256(o_ones + (size_t)chunk * 4),
                 out_bytes = o_soft + (soft_out ? (size_t)chunk * n * ts : 0);
    int rc;
    if (!h->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (HostSlot &sl : h->slot) {
        if ((rc = sl.in.ensure((size_t)chunk * n * ls)) || (rc = sl.out.ensure(out_bytes)) ||
            (rc = sl.llr.ensure((size_t)chunk * n * ls)) || (rc = sl.hard.ensure((size_t)chunk * nbytes)) ||
            (rc = sl.iters.ensure((size_t)chunk * 4)) || (rc = sl.conv.ensure((size_t)chunk)) ||
            (rc = sl.ones.ensure((size_t)chunk * 4)))
            return rc;
        if (soft_out && (rc = sl.soft.ensure((size_t)chunk * n * ts))) return rc;
        if (!sl.up) HIP_TRY(hipEventCreateWithFlags(&sl.up, hipEventDisableTiming));
        if (!sl.done) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    int workers = 7;                         // + the calling thread; measured: 4 threads 52, 8 threads 55.5 GB/s
    if (const char *e = hook("LDPC_HIP_COPY_THREADS")) workers = std::max(0, std::min(31, atoi(e) - 1));
    CopyPool pool(workers);
    const int64_t n_chunks = (n_frames + chunk - 1) / chunk;
    // results of chunk k: page-locked slot -> caller's arrays
    auto collect = [&](int64_t k) -> int {
        HostSlot &sl = h->slot[k & 1];
        const int64_t f0 = k * chunk, nf = std::min<int64_t>(chunk, n_frames - f0);
        HIP_TRY(hipEventSynchronize(sl.done));
        const unsigned char *o = static_cast<const unsigned char *>(sl.out.p);
        if (hard_bits) pool.copy(hard_bits + (size_t)f0 * nbytes, o + o_hard, (size_t)nf * nbytes);
        if (iters) memcpy(iters + f0, o + o_iters, (size_t)nf * 4);
        if (converged) memcpy(converged + f0, o + o_conv, (size_t)nf);
        if (ones) memcpy(ones + f0, o + o_ones, (size_t)nf * 4);
        if (soft_out) pool.copy(static_cast<unsigned char *>(soft_out) + (size_t)f0 * n * ts, o + o_soft, (size_t)nf * n * ts);
        return 0;
    };
    for (int64_t k = 0; k < n_chunks; ++k) {
        HostSlot &sl = h->slot[k & 1];
        const int64_t f0 = k * chunk, nf = std::min<int64_t>(chunk, n_frames - f0);
        if (k >= 2 && (rc = collect(k - 2))) return rc;          // also: the slot's upload and decode are over
        pool.copy(sl.in.p, static_cast<const unsigned char *>(llr) + (size_t)f0 * n * ls, (size_t)nf * n * ls);
        HIP_TRY(hipMemcpyAsync(sl.llr.p, sl.in.p, (size_t)nf * n * ls, hipMemcpyHostToDevice, h->copy_stream));
        HIP_TRY(hipEventRecord(sl.up, h->copy_stream));
        HIP_TRY(hipStreamWaitEvent(h->stream, sl.up, 0));
        DecodeArgs a;
        a.llr = sl.llr.p; a.llr_f64 = llr_bits == 64; a.llr_scale = 1.0f; a.n_frames = nf;
        a.max_iters = max_iters; a.sign_mode = sign_mode;
        a.hard = reinterpret_cast<uint8_t *>(sl.hard.p);
        a.iters = reinterpret_cast<int32_t *>(sl.iters.p);
        a.conv = reinterpret_cast<uint8_t *>(sl.conv.p);
        a.ones = reinterpret_cast<int32_t *>(sl.ones.p);
        a.soft = soft_out ? sl.soft.p : nullptr;
        a.chain = chain; a.work = nullptr; a.ms_alpha = h->ms_alpha; a.ms_beta = h->ms_beta; a.precheck = precheck_override();
        if ((rc = launch_decode(h, algo, prec, a))) return rc;
        unsigned char *o = static_cast<unsigned char *>(sl.out.p);
        HIP_TRY(hipMemcpyAsync(o + o_hard, sl.hard.p, (size_t)nf * nbytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(o + o_iters, sl.iters.p, (size_t)nf * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(o + o_conv, sl.conv.p, (size_t)nf, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(o + o_ones, sl.ones.p, (size_t)nf * 4, hipMemcpyDeviceToHost, h->stream));
        if (soft_out) HIP_TRY(hipMemcpyAsync(o + o_soft, sl.soft.p, (size_t)nf * n * ts, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipEventRecord(sl.done, h->stream));
    }
    for (int64_t k = std::max<int64_t>(0, n_chunks - 2); k < n_chunks; ++k)
        if ((rc = collect(k))) return rc;
    return 0;
}

static int decode_host(ldpc_handle *h, bool chained, int algo, int prec, int sign_mode, int max_iters,
                       int64_t n_frames, const void *llr, int llr_bits, uint8_t *hard_bits,
                       int32_t *iters, uint8_t *converged, int32_t *ones, void *soft_out)
{
    int rc = check_common(h, max_iters, n_frames, sign_mode);
    if (rc) return rc;
    if (n_frames == 0) return 0;
    if (!llr) return fail(LDPC_E_ARG, "llr is NULL");
    if (prec != 32 && prec != 64) return fail(LDPC_E_ARG, "prec must be 32 or 64");
    if (llr_bits != 32 && llr_bits != 64) return fail(LDPC_E_ARG, "llr_bits must be 32 or 64");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->n, nbytes = (n + 7) / 8, ts = prec == 64 ? 8 : 4, ls = llr_bits / 8;
    // from 128 MB of channel values on, the two-slot pipeline pays for its page-locked staging
    const char *nopipe = hook("LDPC_HIP_NO_PIPELINE");
    if ((size_t)n_frames * n * ls >= ((size_t)128 << 20) && !(nopipe && nopipe[0] == '1')) {
        unsigned char *chain0 = nullptr;
        if (chained && (rc = chain_state(h, prec, &chain0))) return rc;
        rc = decode_host_pipelined(h, chain0, algo, prec, sign_mode, max_iters, n_frames, llr, llr_bits, hard_bits,
                                   iters, converged, ones, soft_out);
        if (rc) {                            // leave nothing in flight that still reads the slots
            (void)hipStreamSynchronize(h->copy_stream);
            (void)hipStreamSynchronize(h->stream);
        }
        return rc;
    }
    // stage at most ~256 MB of channel values at a time
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n_frames, (int64_t)((256u << 20) / (n * ls)) + 1));
    if ((rc = h->b_llr.ensure((size_t)chunk * n * ls)) || (rc = h->b_hard.ensure((size_t)chunk * nbytes)) ||
        (rc = h->b_iters.ensure((size_t)chunk * 4)) || (rc = h->b_conv.ensure((size_t)chunk)) ||
        (rc = h->b_ones.ensure((size_t)chunk * 4)))
        return rc;
    if (soft_out && (rc = h->b_soft.ensure((size_t)chunk * n * ts))) return rc;
    unsigned char *chain = nullptr;
    if (chained && (rc = chain_state(h, prec, &chain))) return rc;
    for (int64_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const int64_t nf = std::min<int64_t>(chunk, n_frames - f0);
        HIP_TRY(hipMemcpyAsync(h->b_llr.p, reinterpret_cast<const unsigned char *>(llr) + (size_t)f0 * n * ls,
                               (size_t)nf * n * ls, hipMemcpyHostToDevice, h->stream));
        DecodeArgs a;
        a.llr = h->b_llr.p; a.llr_f64 = llr_bits == 64; a.llr_scale = 1.0f; a.n_frames = nf;
        a.max_iters = max_iters; a.sign_mode = sign_mode;
        a.hard = reinterpret_cast<uint8_t *>(h->b_hard.p);
        a.iters = reinterpret_cast<int32_t *>(h->b_iters.p);
        a.conv = reinterpret_cast<uint8_t *>(h->b_conv.p);
        a.ones = reinterpret_cast<int32_t *>(h->b_ones.p);
        a.soft = soft_out ? h->b_soft.p : nullptr;
        a.chain = chain; a.work = nullptr; a.ms_alpha = h->ms_alpha; a.ms_beta = h->ms_beta; a.precheck = precheck_override();
        if ((rc = launch_decode(h, algo, prec, a))) return rc;
        if (hard_bits) HIP_TRY(hipMemcpyAsync(hard_bits + (size_t)f0 * nbytes, h->b_hard.p, (size_t)nf * nbytes, hipMemcpyDeviceToHost, h->stream));
        if (iters) HIP_TRY(hipMemcpyAsync(iters + f0, h->b_iters.p, (size_t)nf * 4, hipMemcpyDeviceToHost, h->stream));
        if (converged) HIP_TRY(hipMemcpyAsync(converged + f0, h->b_conv.p, (size_t)nf, hipMemcpyDeviceToHost, h->stream));
        if (ones) HIP_TRY(hipMemcpyAsync(ones + f0, h->b_ones.p, (size_t)nf * 4, hipMemcpyDeviceToHost, h->stream));
        if (soft_out) HIP_TRY(hipMemcpyAsync(reinterpret_cast<unsigned char *>(soft_out) + (size_t)f0 * n * ts, h->b_soft.p, (size_t)nf * n * ts, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" int ldpc_decode(ldpc_handle *h, int algo, int prec, int sign_mode, int max_iters,
                           int64_t n_frames, const void *llr, int llr_bits, uint8_t *hard_bits,
                           int32_t *iters, uint8_t *converged, int32_t *ones, void *soft_out)
{
    return decode_host(h, false, algo, prec, sign_mode, max_iters, n_frames, llr, llr_bits, hard_bits,
                       iters, converged, ones, soft_out);
}

extern "C" int ldpc_decode_chained(ldpc_handle *h, int algo, int prec, int sign_mode, int max_iters,
                                   int64_t n_frames, const void *llr, int llr_bits, uint8_t *hard_bits,
                                   int32_t *iters, uint8_t *converged, int32_t *ones, void *soft_out)
{
    return decode_host(h, true, algo, prec, sign_mode, max_iters, n_frames, llr, llr_bits, hard_bits,
                       iters, converged, ones, soft_out);
}

static int launch_channel(ldpc_handle *h, double snr_db, uint64_t seed, uint32_t stream_id,
                          uint64_t first_frame, int64_t n_frames, float *d_llr, float *d_sigma_hat,
                          int32_t *d_uncoded)
{
    const double snr = pow(10.0, snr_db / 10.0);         // ldpc.py:53
    const float sigma = (float)sqrt(0.5 / snr);          // ldpc.py:55
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_frames, (int64_t)h->n_cu * 8));
    hipLaunchKernelGGL(channel_awgn_kernel, dim3(grid), dim3(256), 0, h->stream, seed, stream_id,
                       first_frame, n_frames, h->n, sigma, d_llr, d_sigma_hat, d_uncoded);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int ldpc_channel_awgn_device(ldpc_handle *h, double snr_db, uint64_t seed, uint64_t first_frame,
                                        int64_t n_frames, float *d_llr, float *d_sigma_hat,
                                        int32_t *d_uncoded)
{
    if (!h) return fail(LDPC_E_ARG, "handle is NULL");
    if (n_frames < 0) return fail(LDPC_E_ARG, "n_frames must be >= 0");
    if (n_frames == 0) return 0;
    if (!d_llr) return fail(LDPC_E_ARG, "d_llr is NULL");
    HIP_TRY(hipSetDevice(h->device));
    return launch_channel(h, snr_db, seed, 0u, first_frame, n_frames, d_llr, d_sigma_hat, d_uncoded);
}

extern "C" int ldpc_eval_awgn(ldpc_handle *h, int algo, int prec, int sign_mode, int max_iters,
                              int n_snr, const double *snr_db, int64_t frames_per_snr,
                              uint64_t seed, uint64_t first_frame, uint64_t *counters,
                              float *per_frame)
{
    int rc = check_common(h, max_iters, frames_per_snr, sign_mode);
    if (rc) return rc;
    if (n_snr < 0 || (n_snr > 0 && !snr_db)) return fail(LDPC_E_ARG, "bad SNR list");
    if (!counters) return fail(LDPC_E_ARG, "counters is NULL");
    memset(counters, 0, sizeof(uint64_t) * 6 * (size_t)n_snr);
    if (n_snr == 0 || frames_per_snr == 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->n;
    // Where the decode kernel can make its own channel values (message-plane kernels) a sweep point is
    // one launch that reads nothing from HBM; elsewhere the channel kernel fills a buffer of at most
    // 8 GiB that the decode kernel then reads.  Same noise either way (LDPC_HIP_NO_FUSED_CHANNEL=1, a
    // test hook, forces the second form).
    Launch L;
    if ((rc = select_launch(h, algo, prec, frames_per_snr, false, &L))) return rc;
    const char *unfused = hook("LDPC_HIP_NO_FUSED_CHANNEL");
    const bool fused = L.fn_gen != nullptr && !(unfused && unfused[0] == '1');
    // A small sweep (the gym's 3 points x a few frames, ldpc_env.py:353-377) goes out as ONE launch: a point's
    // time is that of its slowest frame, and with most of the card idle the points can run side by side.
    // Same noise per (point, frame) as one launch per point (LDPC_HIP_NO_POINT_BATCH=1, a test hook).
    const char *unbatched = hook("LDPC_HIP_NO_POINT_BATCH");
    // (Kernels that read their channel values take part when the points share one LLR scale -- min-sum: the
    // channel kernel then fills one buffer point after point, and one decode launch reads it.)
    const bool batched = (fused || algo == LDPC_ALGO_MINSUM) && n_snr > 1 && n_snr <= GEN_MAX_POINTS &&
                         frames_per_snr * n_snr <= GEN_MAX_BATCH && !(unbatched && unbatched[0] == '1');
    const int64_t cap = fused ? ((int64_t)1 << 24) : (int64_t)(((size_t)8192 << 20) / (n * 4)) + 1;
    const int64_t chunk = batched ? frames_per_snr * n_snr : std::max<int64_t>(1, std::min<int64_t>(frames_per_snr, cap));
    if (batched && (rc = select_launch(h, algo, prec, chunk, false, &L))) return rc;      // grid for all the points' frames
    if ((rc = h->b_iters.ensure((size_t)chunk * 4)) ||
        (rc = h->b_conv.ensure((size_t)chunk)) || (rc = h->b_ones.ensure((size_t)chunk * 4)) ||
        (rc = h->b_unc.ensure((size_t)chunk * 4)) ||
        (rc = h->b_cnt.ensure(sizeof(uint64_t) * 6 * (size_t)n_snr)))
        return rc;
    if (fused ? (rc = h->b_sq.ensure((size_t)chunk * 8))
              : ((rc = h->b_llr.ensure((size_t)chunk * n * 4)) || (rc = h->b_sig.ensure((size_t)chunk * 4))))
        return rc;
    if (per_frame && (rc = h->b_pf.ensure((size_t)chunk * 16))) return rc;
    HIP_TRY(hipMemsetAsync(h->b_cnt.p, 0, sizeof(uint64_t) * 6 * (size_t)n_snr, h->stream));
    const int n_launch_points = batched ? 1 : n_snr;
    for (int s = 0; s < n_launch_points; ++s) {
        for (int64_t f0 = 0; f0 < (batched ? chunk : frames_per_snr); f0 += chunk) {
            const int64_t nf = batched ? chunk : std::min<int64_t>(chunk, frames_per_snr - f0);
            DecodeArgs a;
            a.llr_f64 = 0; a.n_frames = nf;
            // min-sum works on the raw channel values like the reference (ldpc.py:371-377: no scaling);
            // sum-product needs true log-likelihood ratios, 2y/sigma^2 for BPSK on AWGN
            a.llr_scale = algo == LDPC_ALGO_SUMPRODUCT ? (float)(4.0 * pow(10.0, snr_db[s] / 10.0)) : 1.0f;
            a.max_iters = max_iters; a.sign_mode = sign_mode; a.hard = nullptr;
            a.iters = reinterpret_cast<int32_t *>(h->b_iters.p);
            a.conv = reinterpret_cast<uint8_t *>(h->b_conv.p);
            a.ones = reinterpret_cast<int32_t *>(h->b_ones.p);
            a.soft = nullptr; a.chain = nullptr; a.work = nullptr; a.ms_alpha = h->ms_alpha; a.ms_beta = h->ms_beta; a.precheck = precheck_override();
            PointSigmas ps{};
            if (fused) {
                a.llr = nullptr;
                a.gen_seed = seed; a.gen_stream = (uint32_t)s; a.gen_first_frame = first_frame + (uint64_t)f0;
                a.gen_sigma = (float)sqrt(0.5 / pow(10.0, snr_db[s] / 10.0));      // ldpc.py:53-55
                a.gen_uncoded = reinterpret_cast<int32_t *>(h->b_unc.p);
                a.gen_sumsq = reinterpret_cast<unsigned long long *>(h->b_sq.p);
                if (batched) {
                    a.gen_per_point = (uint32_t)frames_per_snr;
                    for (int q = 0; q < n_snr; ++q) {
                        a.gen_point_sigma[q] = ps.sigma[q] = (float)sqrt(0.5 / pow(10.0, snr_db[q] / 10.0));
                        a.gen_point_scale[q] = algo == LDPC_ALGO_SUMPRODUCT ? (float)(4.0 * pow(10.0, snr_db[q] / 10.0)) : 1.0f;
                    }
                }
            } else if (batched) {
                for (int q = 0; q < n_snr; ++q) {
                    const size_t at = (size_t)q * (size_t)frames_per_snr;
                    if ((rc = launch_channel(h, snr_db[q], seed, (uint32_t)q, first_frame, frames_per_snr,
                                             reinterpret_cast<float *>(h->b_llr.p) + at * n,
                                             reinterpret_cast<float *>(h->b_sig.p) + at,
                                             reinterpret_cast<int32_t *>(h->b_unc.p) + at)))
                        return rc;
                }
                a.llr = h->b_llr.p;
            } else {
                if ((rc = launch_channel(h, snr_db[s], seed, (uint32_t)s, first_frame + (uint64_t)f0, nf,
                                         reinterpret_cast<float *>(h->b_llr.p),
                                         reinterpret_cast<float *>(h->b_sig.p),
                                         reinterpret_cast<int32_t *>(h->b_unc.p))))
                    return rc;
                a.llr = h->b_llr.p;
            }
            if ((rc = launch_decode(h, algo, prec, a))) return rc;
            const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((nf + 255) / 256, 1024));
            hipLaunchKernelGGL(accumulate_kernel, dim3(grid), dim3(256), 0, h->stream, nf, h->n,
                               (float)sqrt(0.5 / pow(10.0, snr_db[s] / 10.0)),
                               reinterpret_cast<const int32_t *>(h->b_iters.p),
                               reinterpret_cast<const uint8_t *>(h->b_conv.p),
                               reinterpret_cast<const int32_t *>(h->b_ones.p),
                               reinterpret_cast<const int32_t *>(h->b_unc.p),
                               fused ? nullptr : reinterpret_cast<const float *>(h->b_sig.p),
                               fused ? reinterpret_cast<const unsigned long long *>(h->b_sq.p) : nullptr,
                               reinterpret_cast<unsigned long long *>(h->b_cnt.p) + 6 * s,
                               per_frame ? reinterpret_cast<float *>(h->b_pf.p) : nullptr,
                               batched ? (uint32_t)frames_per_snr : 0u, ps);
            HIP_TRY(hipGetLastError());
            if (per_frame)
                HIP_TRY(hipMemcpyAsync(per_frame + ((size_t)s * frames_per_snr + f0) * 4, h->b_pf.p,
                                       (size_t)nf * 16, hipMemcpyDeviceToHost, h->stream));
        }
    }
    HIP_TRY(hipMemcpyAsync(counters, h->b_cnt.p, sizeof(uint64_t) * 6 * (size_t)n_snr, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}
