// Float32 min-sum for block-circulant codes with heavy circulants (see qc_heavy.h): one frame per
// workgroup, thread = one check row in the check pass and up to HQ_CPT columns in the variable pass.
//
// Restates ldpcDecoder.decoderMainLoop (ldpc.py:326-337) like the library's other kernels: syndrome of
// the channel hard decision first; per iteration variableNodesToCheckNodes + calcOutgoingValues (check
// pass: v = posterior - previous message, ldpc.py:161; sign(0) = +1, ldpc.py:139; two smallest
// magnitudes, ldpc.py:174-191) and checkNodesToVariableNodes + channel add (variable pass: messages
// added in ascending check index, channel value last, ldpc.py:298-303, 322); the syndrome of iteration
// k's posterior is a by-product of the check pass of iteration k + 1.  Same operations in the same
// order, no contraction: results are bit-identical to the other kernels and to the oracle.
//
// What makes it fast where the table-driven kernel is not (52 vector instructions per edge and
// iteration there, about 22 here):
//   * no reduction modulo Z anywhere: posteriors and row records are stored twice (qc_heavy.h), so an
//     address is one vector add of a lane constant and a table entry;
//   * the tables are read with scalar loads (their addresses are uniform in a wave and made so
//     explicitly), entries are operands of the vector instructions directly;
//   * the bit position of an edge inside its sign word is a compile-time constant in the check pass
//     (rows are walked in unrolled steps of HQ_CH edges, with a uniform exit);
//   * magnitudes are stored doubled, so that sign and magnitude of a message join in one funnel shift;
//   * arg-min as a bit mask ("this edge lowered the minimum": the sign of new minimum - old minimum),
//     second minimum and minimum as bare v_med3_f32 / v_min_f32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ldpc_hip.h"
#include "decode_args.h"
#include "qc_heavy.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) u32x4 lds_u128_t;
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) u32x16 *tab512_t;     // constant address space: scalar loads

// LDS by byte address: the kernel has no static LDS, so its dynamic LDS starts at address 0 (the host
// checks that) and a layout offset IS the address.
__device__ __forceinline__ float ld_f32(uint32_t adr) { return *reinterpret_cast<const lds_f32_t *>((uintptr_t)adr); }
__device__ __forceinline__ uint32_t ld_u32(uint32_t adr) { return *reinterpret_cast<const lds_u32_t *>((uintptr_t)adr); }
__device__ __forceinline__ u32x4 ld_u128(uint32_t adr) { return *reinterpret_cast<const lds_u128_t *>((uintptr_t)adr); }
__device__ __forceinline__ void st_f32(uint32_t adr, float v) { *reinterpret_cast<lds_f32_t *>((uintptr_t)adr) = v; }
__device__ __forceinline__ void st_u32(uint32_t adr, uint32_t v) { *reinterpret_cast<lds_u32_t *>((uintptr_t)adr) = v; }
__device__ __forceinline__ void st_u128(uint32_t adr, u32x4 v) { *reinterpret_cast<lds_u128_t *>((uintptr_t)adr) = v; }

#ifndef HQ_PRIO
#define HQ_PRIO 1
#endif
__device__ __forceinline__ void hq_setprio(int p)
{
    if (p >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ void hq_barrier()
{
    // LDS traffic only: global loads and stores in flight do not hold the barrier up
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// What a row sent along its edges in the previous iteration: { 2 min1, 2 min2 } as bit patterns, the
// edge that got min2, and one sign bit per edge (first edge of a word highest)
template <int NW> struct HqRow { uint32_t m1x, m2x, idx, sw[NW]; };

// Check pass of one row.  rt: this block-row's edge table.  Returns the lanes whose parity check on
// the current hard decisions fails.  FIRST: a fresh decoder's previous messages are all +0
// (checkNode.__init__, ldpc.py:121): v = posterior.
template <int NW, bool FIRST, bool SPLIT>
__device__ __forceinline__ unsigned long long
hq_check_pass(uint32_t rt, uint32_t cw, int dpad, int deg, uint32_t i4, uint32_t my_rec, uint32_t my_sw, const HeavyGeom &G,
              HqRow<NW> &st, int sign_mode, float alpha, float beta, unsigned long long synd0)
{
    enum { CH = HQ_CH, NCH = NW * 32 / CH };
    const float INF = __uint_as_float(0x7f800000u);
    float min1 = INF, min2 = INF;
    uint32_t nsw[NW], LT[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) nsw[w] = LT[w] = 0;
    unsigned long long synd = synd0;

    // The row's edge table sits in LDS (every wave of a block-row walks the same one): table entries of
    // step c + 2 and posteriors of step c + 1 are in flight while step c is reduced.  (Scalar loads would
    // share the LDS counter and, returning out of order, force every wait in here down to zero.)
    asm volatile("" : "+s"(dpad));                       // the steps' exit tests stay where they are used
    float pb[2][CH];
    u32x4 tv[2][2];
    auto issue_tab = [&](int cc, u32x4 (&dst)[2]) {
        dst[0] = ld_u128(rt + 32u * cc);
        dst[1] = ld_u128(rt + 32u * cc + 16u);
    };
    auto issue = [&](const u32x4 (&t)[2], float (&dst)[CH]) {
        dst[0] = ld_f32(i4 + t[0].x); dst[1] = ld_f32(i4 + t[0].y); dst[2] = ld_f32(i4 + t[0].z); dst[3] = ld_f32(i4 + t[0].w);
        dst[4] = ld_f32(i4 + t[1].x); dst[5] = ld_f32(i4 + t[1].y); dst[6] = ld_f32(i4 + t[1].z); dst[7] = ld_f32(i4 + t[1].w);
    };
    issue_tab(0, tv[0]);
    issue(tv[0], pb[0]);
    issue_tab(1, tv[1]);
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
        if (cc * CH >= dpad) break;                                      // uniform
        // the next step's posteriors and the table entries of the one after it: asked for whether the row has
        // them or not (past its end the table names the dummy row), so that no value is defined under a branch
        if (cc + 1 < NCH) issue(tv[(cc + 1) & 1], pb[(cc + 1) & 1]);
        if (cc + 2 < NCH) issue_tab(cc + 2, tv[cc & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const int w = cc * CH / 32;
        // waves lower their own priority as they advance, so that the four of a SIMD finish the pass together
        // instead of one after the other (the issue arbiter favours the oldest wave)
        if (HQ_PRIO) hq_setprio(dpad > 48 ? 3 - cc / 2 : 3 - cc);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (u == HQ_PAD && CH > HQ_PAD && cc * CH + u >= dpad) break;    // rows are padded to multiples of HQ_PAD (uniform)
            const int k = cc * CH % 32 + u;                              // bit 31 - k of word w
            const uint32_t e = (uint32_t)(cc * CH + u);
            const float p = pb[cc & 1][u];
            synd ^= __ballot(p > 0.0f);                                  // slicer, ldpc.py:47
            float v = p;
            if (!FIRST) {
                const uint32_t magx = st.idx == e ? st.m2x : st.m1x;
                const uint32_t ts = k == 31 ? st.sw[w] : st.sw[w] >> (31 - k);
                uint32_t c_old;
                asm("v_alignbit_b32 %0, %1, %2, 1" : "=v"(c_old) : "v"(ts), "v"(magx));
                v = p - __uint_as_float(c_old);                          // receive, ldpc.py:161
            }
            nsw[w] = __builtin_amdgcn_alignbit(nsw[w], __float_as_uint(v), 31);      // setSign, ldpc.py:139
            float n2, n1, dlt;
            asm("v_med3_f32 %0, %1, %2, |%3|" : "=v"(n2) : "v"(min1), "v"(min2), "v"(v));    // ldpc.py:178-183
            asm("v_min_f32 %0, %1, |%2|" : "=v"(n1) : "v"(min1), "v"(v));
            // new minimum - old minimum is negative exactly when this edge lowered it
            asm("v_sub_f32 %0, %1, %2" : "=v"(dlt) : "v"(n1), "v"(min1));
            LT[w] = __builtin_amdgcn_alignbit(LT[w], __float_as_uint(dlt), 31);
            min2 = n2;
            min1 = n1;
            if ((u & 3) == 3) {
                // keeps the running state where it is: otherwise the chains are deferred to the end of the
                // pass and every intermediate stays alive until then
                asm volatile("" : "+v"(min1), "+v"(min2), "+v"(LT[w]), "+v"(nsw[w]), "+s"(synd));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (HQ_PRIO) __builtin_amdgcn_s_setprio(0);
    // left-align the words (first edge highest); the last edge that lowered the minimum is the arg-min
    uint32_t idx = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int cnt = dpad - 32 * w;                                   // uniform
        if (cnt > 0 && cnt < 32) { nsw[w] <<= (32 - cnt); LT[w] <<= (32 - cnt); }
        if (cnt > 0) idx = LT[w] != 0u ? (uint32_t)(32 * w + 31) - (uint32_t)__builtin_ctz(LT[w]) : idx;
    }
    // sign of every outgoing message = (product of all signs) * own sign, ldpc.py:190
    uint32_t S = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) S ^= __popc(nsw[w]);
    S &= 1u;
    if (sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)deg & 1u;
    const uint32_t flip = 0u - S;
    if (alpha != 1.0f || beta != 0.0f) {                                 // normalised / offset min-sum (uniform)
        min1 = fmaxf(alpha * min1 - beta, 0.0f);
        min2 = fmaxf(alpha * min2 - beta, 0.0f);
    }
    st.m1x = __float_as_uint(min1) << 1;                                 // magnitudes are >= 0: nothing is shifted out
    st.m2x = __float_as_uint(min2) << 1;
    st.idx = idx;
#pragma unroll
    for (int w = 0; w < NW; ++w) st.sw[w] = nsw[w] ^ flip;
    // the row's record, at index i and at i + Z
    // what a column compares its own edge's table word with: the table word of the arg-min edge
    const uint32_t amw = ld_u32(cw + 4u * idx);
    if (SPLIT) {
        const u32x4 r = {st.m1x, st.m2x, amw, 0u};
        st_u128(my_rec, r);
        st_u128(my_rec + G.Z * 16, r);
#pragma unroll
        for (int w = 0; w < NW; ++w)
            if (32 * w < dpad) {
                st_u32(my_sw + w * G.sw_wstride, st.sw[w]);
                st_u32(my_sw + w * G.sw_wstride + G.Z * 4, st.sw[w]);
            }
    } else {
#pragma unroll
        for (int w = 0; w < NW; ++w)
            if (32 * w < dpad) {
                const u32x4 r = {st.m1x, st.m2x, amw, st.sw[w]};
                st_u128(my_rec + w * G.rec_wstride, r);
                st_u128(my_rec + w * G.rec_wstride + G.Z * 16, r);
            }
    }
    return synd;
}

// Messages of one block (weight W) to the column at position c of its block-column, added to acc in
// ascending check index (ldpc.py:298-303): with the block's shifts descending, the rows c - s of the
// shifts s <= c ascend, then come the wrapped rows c - s + Z of the shifts s > c, ascending as well.
// An entry that is not its pass's turn adds +0, which changes no sum that started from +0.
// E: the block's 16 table words (qc_heavy.h) in scalar registers, used as they are: word 2k is an address,
// word 2k + 1 = s << 18 | edge << 5 | (31 - edge mod 32) is at once the shift amount that brings the edge's sign
// to bit 0 (a shift looks at 5 bits), what the row record's third word equals if this edge got the second
// minimum, and -- against cw = c << 18 | 0x3ffff -- the test "s > c".  SA: split layout, sign word addresses.
template <int W, bool SPLIT>
__device__ __forceinline__ float hq_block(const u32x16 &E, const u32x8 &SA, uint32_t c16, uint32_t c4, uint32_t cw, float acc)
{
    u32x4 R[W];
    uint32_t SW[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        R[k] = ld_u128(c16 + E[2 * k]);
        if (SPLIT) SW[k] = ld_u32(c4 + SA[k]);
    }
    float m[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t magx = R[k].z == E[2 * k + 1] ? R[k].y : R[k].x;  // ldpc.py:190-191
        uint32_t ts, mk;
        asm("v_lshrrev_b32 %0, %1, %2" : "=v"(ts) : "s"(E[2 * k + 1]), "v"(SPLIT ? SW[k] : R[k].w));
        asm("v_alignbit_b32 %0, %1, %2, 1" : "=v"(mk) : "v"(ts), "v"(magx));
        m[k] = __uint_as_float(mk);
    }
    if (W == 1) return acc + m[0];
#pragma unroll
    for (int k = 0; k < W; ++k) acc = acc + (E[2 * k + 1] > cw ? 0.0f : m[k]);
#pragma unroll
    for (int k = 0; k < W; ++k) acc = acc + (E[2 * k + 1] > cw ? m[k] : 0.0f);
    return acc;
}

template <bool SPLIT>
__device__ __forceinline__ float hq_block_any(const u32x16 &E, const u32x8 &SA, uint32_t c16, uint32_t c4, uint32_t cw, float acc)
{
    switch (E[14]) {                                                     // the block's weight (uniform)
    case 0: return acc;
    case 1: return hq_block<1, SPLIT>(E, SA, c16, c4, cw, acc);
    case 2: return hq_block<2, SPLIT>(E, SA, c16, c4, cw, acc);
    case 3: return hq_block<3, SPLIT>(E, SA, c16, c4, cw, acc);
    case 4: return hq_block<4, SPLIT>(E, SA, c16, c4, cw, acc);
    case 5: return hq_block<5, SPLIT>(E, SA, c16, c4, cw, acc);
    case 6: return hq_block<6, SPLIT>(E, SA, c16, c4, cw, acc);
    default: return hq_block<7, SPLIT>(E, SA, c16, c4, cw, acc);
    }
}

// One block's table words by scalar load; the matching wait names the registers, so that nothing reads them earlier.
template <bool SPLIT>
__device__ __forceinline__ void hq_tab_load(u32x16 &E, u32x8 &SA, tab512_t p)
{
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(E) : "s"(p));
    if (SPLIT) asm volatile("s_load_dwordx8 %0, %1, 0x40" : "=s"(SA) : "s"(p));
}
template <bool SPLIT>
__device__ __forceinline__ void hq_tab_wait(u32x16 &E, u32x8 &SA)
{
    if (SPLIT) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(E), "+s"(SA));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(E));
}

#ifdef HQ_TRACE
// phase timing (a build for tools/trace_heavy.py only): clock samples summed per wave, written once at the end
#define HQ_T(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tr_acc[k] += now_ - tr_t; tr_t = now_; }
#else
#define HQ_T(k)
#endif

template <int NW, bool SPLIT>
__global__ void __launch_bounds__(1024)
qc_heavy_kernel(const HeavyTables tb, const DecodeArgs a)
{
    const int Z = tb.Z, Zp = tb.Zp, mb = tb.mb, nb = tb.nb, n = tb.n;
    const HeavyGeom G(Z, Zp, mb, nb, tb.nwu);
    const int tid = threadIdx.x;
    const int NT = mb * Zp;
    const int g = __builtin_amdgcn_readfirstlane(tid / Zp);      // block-row served / first block-column served
    const int i = tid - g * Zp;
    const bool live = i < Z;
    const uint32_t ic = live ? (uint32_t)i : 0u;
    const int lane = tid & 63;
    const int nbytes = (n + 7) >> 3;
    const uint32_t i4 = ic * 4, i16 = ic * 16;
    const int deg = tb.deg[g], dpad = tb.dpad[g];
    const uint32_t rt = G.rowtab_off + (uint32_t)g * 4u * G.tabw;
    const uint32_t cwt = G.rowtab_off + ((uint32_t)mb + (uint32_t)g) * 4u * G.tabw;      // the edges' column-table words
    enum { BLKW = SPLIT ? 2 * HQ_BLK : HQ_BLK };                  // table words per block
    const tab512_t ct = (tab512_t)(uintptr_t)(tb.coltab + (size_t)g * (HQ_CPT * mb + 2) * BLKW);
    const int ncol = tb.ncol[g], nblk = ncol * mb;               // my block-columns; their blocks
    const uint32_t cwrap = (ic << 18) | 0x3ffffu;
    const uint32_t col0 = ((uint32_t)g * G.Z2 + ic) * 4, colstep = (uint32_t)mb * G.Z2 * 4;      // my first column's posterior
    const uint32_t my_rec = G.rec_off + ((uint32_t)g * G.Z2 + ic) * 16;
    const uint32_t my_sw = G.sw_off + ((uint32_t)g * G.Z2 + ic) * 4;
    // padding edges read +inf: hard decision 1 each
    const unsigned long long synd0 = ((dpad - deg) & 1) ? __ballot(live) : 0ull;
    const uint32_t flags = G.flag_off;                            // [2] "syndrome non-zero", ones counter, claimed frame

#ifdef HQ_TRACE
    unsigned long long tr_acc[4] = {0, 0, 0, 0}, tr_t = 0;
#endif
    for (int x = tid; x < Zp; x += NT) st_u32(G.dummy_off + 4u * x, 0x7f800000u);
    for (int x = tid; x < 2 * mb * (int)G.tabw; x += NT) st_u32(G.rowtab_off + 4u * x, tb.rowtab[x]);

    for (int64_t frame = blockIdx.x, next_frame = 0; frame < a.n_frames; frame = next_frame) {
        // ---- channel values; posterior := channel (decoderSet, ldpc.py:270-274)
        float y[HQ_CPT];
        {
            const int64_t base = frame * n + (int64_t)ic;
            if (a.llr_f64) {
                const double *src = reinterpret_cast<const double *>(a.llr) + base;
#pragma unroll
                for (int q = 0; q < HQ_CPT; ++q) {
                    const int j = g + q * mb;
                    y[q] = (live && j < nb) ? (float)src[(int64_t)j * Z] : 0.0f;
                }
            } else {
                const float *src = reinterpret_cast<const float *>(a.llr) + base;
#pragma unroll
                for (int q = 0; q < HQ_CPT; ++q) {
                    const int j = g + q * mb;
                    y[q] = (live && j < nb) ? src[(int64_t)j * Z] : 0.0f;
                }
            }
#pragma unroll
            for (int q = 0; q < HQ_CPT; ++q) {
                const int j = g + q * mb;
                // "+ 0": -0.0 becomes +0.0 (invisible to every comparison and sum below; keeps "sign bit set"
                // equivalent to "< 0" for every message)
                y[q] = y[q] * a.llr_scale + 0.0f;
                if (live && j < nb) {
                    st_f32(((uint32_t)j * G.Z2 + ic) * 4, y[q]);
                    st_f32(((uint32_t)j * G.Z2 + ic + G.Z) * 4, y[q]);
                }
            }
        }
        if (tid < 3) st_u32(flags + 4u * tid, 0u);
        if (tid == 0) st_u32(flags + 12, a.work ? (uint32_t)atomicAdd(a.work, 1) : 0u);
        hq_barrier();
        next_frame = a.work ? (int64_t)gridDim.x + (int64_t)__builtin_amdgcn_readfirstlane((int)ld_u32(flags + 12))
                            : frame + (int64_t)gridDim.x;

        HqRow<NW> st;
        st.m1x = st.m2x = st.idx = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) st.sw[w] = 0;
        int my_iters = 0, my_conv = 0;
#ifdef HQ_TRACE
        tr_t = __builtin_readcyclecounter();
#endif
        for (int it = 0;; ++it) {
            const int par = it & 1;
            // ================= check pass (+ syndrome of the current posterior)
            if (live) {
                const unsigned long long synd =
                    it == 0 ? hq_check_pass<NW, true, SPLIT>(rt, cwt, dpad, deg, i4, my_rec, my_sw, G, st, a.sign_mode, a.ms_alpha, a.ms_beta, synd0)
                            : hq_check_pass<NW, false, SPLIT>(rt, cwt, dpad, deg, i4, my_rec, my_sw, G, st, a.sign_mode, a.ms_alpha, a.ms_beta, synd0);
                if (synd != 0ull && lane == 0) st_u32(flags + 4u * par, 1u);
            }
            HQ_T(0);
            hq_barrier();
            HQ_T(1);
            // ================= stop test (ldpc.py:332-335)
            const int bad = __builtin_amdgcn_readfirstlane((int)ld_u32(flags + 4u * par));
            if (bad == 0 || it >= a.max_iters) {
                my_iters = it;
                my_conv = bad == 0;
                break;
            }
            if (tid == 0) st_u32(flags + 4u * (par ^ 1), 0u);
            // ================= variable pass (ldpc.py:295-311, 321-323)
            if (live) {
                // Blocks in pairs: while one block's messages are fetched and summed, the next block's table words
                // are on their way (asked for before this block's LDS traffic, waited for after it).
                float acc = 0.0f;
                int b = 0, done = 0;
                uint32_t col = col0;
                auto end_of_block = [&]() {
                    if (++b == mb) {
                        // the column's last block-row: add the channel value (last, ldpc.py:322) and store; the
                        // channel values rotate, so that y[0] is always the current column's
                        const float yq = y[0];
#pragma unroll
                        for (int qq = 0; qq + 1 < HQ_CPT; ++qq) y[qq] = y[qq + 1];
                        y[HQ_CPT - 1] = yq;
                        if (done < ncol) {
                            const float post = acc + yq;
                            st_f32(col, post);
                            st_f32(col + G.Z * 4, post);
                        }
                        col += colstep;
                        acc = 0.0f;
                        b = 0;
                        ++done;
                    }
                };
                u32x16 EA, EB;
                u32x8 SA, SB;
                hq_tab_load<SPLIT>(EA, SA, ct);
                hq_tab_wait<SPLIT>(EA, SA);
#pragma unroll 1
                for (int t = 0; t < nblk; t += 2) {                      // (the table ends with two spare blocks)
                    hq_tab_load<SPLIT>(EB, SB, (tab512_t)((const __attribute__((address_space(4))) uint32_t *)ct + (size_t)(t + 1) * BLKW));
                    acc = hq_block_any<SPLIT>(EA, SA, i16, i4, cwrap, acc);
                    end_of_block();
                    hq_tab_wait<SPLIT>(EB, SB);
                    hq_tab_load<SPLIT>(EA, SA, (tab512_t)((const __attribute__((address_space(4))) uint32_t *)ct + (size_t)(t + 2) * BLKW));
                    acc = hq_block_any<SPLIT>(EB, SB, i16, i4, cwrap, acc);
                    end_of_block();
                    hq_tab_wait<SPLIT>(EA, SA);
                }
#pragma unroll 1
                for (; done < HQ_CPT; ++done) {                          // the rotation comes full circle
                    const float yq = y[0];
#pragma unroll
                    for (int qq = 0; qq + 1 < HQ_CPT; ++qq) y[qq] = y[qq + 1];
                    y[HQ_CPT - 1] = yq;
                }
            }
            HQ_T(2);
            hq_barrier();
            HQ_T(3);
        }

        // ================= outputs
        for (int x = tid; x < nbytes; x += NT) {
            const int c0 = x * 8;
            int j = c0 / Z, c = c0 - j * Z;
            uint32_t byte = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (c0 + k < n) byte |= (uint32_t)(ld_f32(((uint32_t)j * G.Z2 + (uint32_t)c) * 4) > 0.0f) << k;
                if (++c == Z) { c = 0; ++j; }
            }
            if (a.hard) a.hard[frame * nbytes + x] = (uint8_t)byte;
            if (byte) __hip_atomic_fetch_add(reinterpret_cast<lds_u32_t *>((uintptr_t)(flags + 8)), (uint32_t)__popc(byte), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#ifndef HQ_TRACE
        if (a.soft && live) {
            float *so = reinterpret_cast<float *>(a.soft) + frame * n;
#pragma unroll
            for (int q = 0; q < HQ_CPT; ++q) {
                const int j = g + q * mb;
                if (j < nb) so[j * Z + (int)ic] = ld_f32(((uint32_t)j * G.Z2 + ic) * 4);
            }
        }
#endif
        hq_barrier();
        if (tid == 0) {
            if (a.iters) a.iters[frame] = my_iters;
            if (a.conv) a.conv[frame] = (uint8_t)my_conv;
            if (a.ones) a.ones[frame] = (int32_t)ld_u32(flags + 8);
        }
        hq_barrier();
    }
#ifdef HQ_TRACE
    if (lane == 0 && a.soft)
        for (int k = 0; k < 4; ++k)
            reinterpret_cast<unsigned long long *>(a.soft)[((size_t)blockIdx.x * 16 + (tid >> 6)) * 4 + k] = tr_acc[k];
#endif
}

}  // namespace

const void *heavyqc_kernel(int nwu)
{
    return nwu <= 2 ? reinterpret_cast<const void *>(&qc_heavy_kernel<2, false>)
                    : reinterpret_cast<const void *>(&qc_heavy_kernel<4, true>);
}

int heavyqc_launch(const void *fn, const HeavyTables &tb, const DecodeArgs &a, int grid, int block, size_t lds,
                   hipStream_t stream)
{
    HeavyTables t = tb;
    DecodeArgs d = a;
    void *kargs[] = {&t, &d};
    return (int)hipLaunchKernel(fn, dim3(grid), dim3(block), kargs, lds, stream);
}
