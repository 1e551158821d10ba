// Throughput kernel for float32 min-sum on regular quasi-cyclic codes with two
// block-rows of weight-2 circulants (NearEarth (8176,7154) and its variants with
// zeroed circulants).  Included by ldpc_hip.hip.
//
// Measured on MI355X the decode is bound by vector-instruction issue (one
// wave64 instruction per 4 clocks per SIMD), with the LDS pipe about one third
// busy, so this kernel trades VALU work for LDS traffic: check-to-variable
// messages are stored EXPLICITLY in LDS, one record of five dwords per variable
// node
//     { posterior, msg(block-row 0, shift 0), msg(0, 1), msg(1, 0), msg(1, 1) }
// (stride 5 dwords: consecutive columns fall into distinct banks).  The whole
// frame state is 8176 * 20 B = 163,520 B of the CU's 160 KiB.
//   check pass, per edge:  2 LDS reads (posterior, previous message) and
//       7 vector instructions (hard decision, subtract, sign bit, two-minimum
//       update, argmin), then 1 LDS write and 2 instructions to send
//       "min1 with sign"; the single edge that must carry min2 instead is
//       patched per row afterwards;
//   variable pass, per column: 4 LDS reads, 5 additions, 1 LDS write.
// Same arithmetic in the same order as the other kernels (ldpc.py:150-163,
// 174-202, 295-324): results are bit-identical.
//
// Circulants of weight 0 or 1 (the zeroed variants of testMatricesGeneratorScript.py:23-34,
// light circulants a gym action installs, ldpc_env.py:206-210) run the same code: an edge a
// row lacks points to one dummy record whose posterior is +inf.  Such an edge has |v| = inf
// (never a minimum), sign + and hard decision 1; a row with an odd number of them starts its
// parity from 1.  What the row writes back lands in the dummy's message slots.  The message
// slots no check owns are zeroed once per launch and add +0.
#pragma once

// LDS accesses by byte address.  The kernels of this file have no static LDS, so their dynamic LDS
// starts at address 0 (the host checks that before it selects them) and a record address IS the
// LDS address: no base is added, which otherwise costs a vector add per access wherever the
// compiler does not keep base + address in a register of its own.
typedef __attribute__((address_space(3))) float lds_float_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
__device__ __forceinline__ float lds_f32(const unsigned char *, uint32_t addr)
{
    return *reinterpret_cast<const lds_float_t *>((uintptr_t)addr);
}
__device__ __forceinline__ void lds_put_f32(unsigned char *, uint32_t addr, float v)
{
    *reinterpret_cast<lds_float_t *>((uintptr_t)addr) = v;
}
__device__ __forceinline__ void lds_put_u32(unsigned char *, uint32_t addr, uint32_t v)
{
    *reinterpret_cast<lds_u32_t *>((uintptr_t)addr) = v;
}

// Channel values of one frame for this thread's CPT columns.  `frame_base` is uniform (scalar
// registers) and the per-lane part is a 32-bit element offset, laundered so that it is
// recomputed at each use instead of being kept (as CPT 64-bit addresses) across the decode loop.
template <int CPT, int Q0 = 0, int Q1 = CPT>
__device__ __forceinline__ void
planes_load_frame(const void *llr, int is_f64, int64_t frame_base, uint32_t first, uint32_t stride, float (&out)[CPT])
{
    first = opaque(first);
    if (is_f64) {
        const double *p = reinterpret_cast<const double *>(llr) + frame_base;
#pragma unroll
        for (int q = Q0; q < Q1; ++q) out[q] = (float)p[first + (uint32_t)q * stride];
    } else {
        const float *p = reinterpret_cast<const float *>(llr) + frame_base;
#pragma unroll
        for (int q = Q0; q < Q1; ++q) out[q] = p[first + (uint32_t)q * stride];
    }
}

#ifndef PLANES_CH
#define PLANES_CH 4                 // edges per pipeline stage of the check pass
#endif
#ifndef PLANES_PRIO
#define PLANES_PRIO 1               // 0: no priority staging in the check pass (5 % slower; kept for comparison)
#endif
#ifndef PLANES_ARGMIN_FSUB
#define PLANES_ARGMIN_FSUB 1        // 0: arg-min bits through the carry (v_sub_co / v_addc_co), 1.9 % slower; kept for comparison
#endif
#ifndef PLANES_OUT_ALIGNBIT
#define PLANES_OUT_ALIGNBIT 1       // 0: outgoing sign by left shift and and-or, 1.4 % slower; kept for comparison
#endif
#ifndef PLANES_CANON
#define PLANES_CANON 1              // zeros are kept as -0 in LDS (posteriors and messages), so that a posterior's sign bit is its hard decision; 0: compare per edge, 0.95 % slower, kept for comparison
#endif
#ifndef PLANES_VB
#define PLANES_VB 2                 // columns per batch of the variable pass (measured: 2 < 4 < 1 < 8)
#endif

// Heavy circulants (weight 3 .. PLANES_MAXW: what a gym action with up to 7 hot bits installs,
// ldpc_env.py:293-317): the two lowest shifts of such a block use its two message slots like any
// other block; its further shifts are "extra edges".  A row of block-row b has n_extra[b] of them
// (at most PLANES_MAXX).  The row thread keeps their record addresses in registers and rebuilds the
// messages it sent along them last time from the row's compressed state {min1, min2, argmin,
// signs}; what it sends goes to a per-workgroup scratch row in global memory (the LDS is full),
// from where the column's owner adds it in the variable pass.
enum { PLANES_MAXX = 8, PLANES_MAXW = 8, PLANES_MAXH = 8, PLANES_XROW = 512 };

// Kernel variants by how heavy the code is (register and code-size budget: the unrolled loops over a
// row's extra edges and over a heavy block's shifts are sized by these limits, and measured on
// MI355X the variant with the large limits decodes a weight-3 mutation at 51 Gbit/s, the one with
// the small limits at 62): HV 0 no heavy circulant, 1 at most 2 extra edges per block-row and
// weight <= 4, 2 at most PLANES_MAXX extra edges and weight <= PLANES_MAXW.
template <int HV> struct PlanesHeavy { enum { MX = 1, MW = 2 }; };           // HV 0: unused (array bounds only)
template <> struct PlanesHeavy<1> { enum { MX = 2, MW = 4 }; };
template <> struct PlanesHeavy<2> { enum { MX = PLANES_MAXX, MW = PLANES_MAXW }; };

struct PlanesTables {
    uint32_t edge_mask[2];   // [block-row] bit 2j+k set: shift k of circulant (b, j) exists (weight 0, 1 or 2)
    const int32_t *shifts;   // [2][NB][2] ascending inside a circulant; entries of missing shifts unused
    // ---- heavy circulants only (HEAVY kernels)
    int n_extra[2];                  // extra edges per row of block-row b
    int32_t extra[2][PLANES_MAXX];   // block-column | shift << 8
    // Block-columns that hold a heavy circulant, by column group g (thread group g owns block-columns
    // j = g + 2q): how many, and per column 16 words (by value: kernel arguments are read with scalar
    // loads, one wide load per column).  Per block-row b, 8 words for block (b, j), whose shifts are
    // s_0 < s_1 < ... < s_(w-1), s_0 and s_1 on the message slots and s_k, k >= 2, on extra edges:
    //   word 0: w | s_0 << 8 | s_1 << 20 | q << 29 (q in block-row 0's word only)
    //   word k-1, k = 2..7: s_k | scratch row of that extra edge << 16
    // with 511 in place of every shift that does not exist (beyond any position, so it never counts
    // as "not wrapped") and, for k >= w, the scratch row of some existing extra edge (a valid address;
    // what is read there is discarded).
    int n_heavy[2];
    int32_t hent[2][PLANES_MAXH][16];
    float *xmsg;                     // scratch [workgroup][n_extra[0] + n_extra[1]][PLANES_XROW] extra-edge messages
    uint32_t xstride;                // floats per workgroup in xmsg
};

// A row's extra edges (HEAVY kernels): record addresses, and the compressed form of what the row
// sent along them in the previous iteration
struct PlanesXState {
    uint32_t xadr2[PLANES_MAXX / 2]; // record addresses / 4, two to a register (they are below 2^18 and multiples of 4)
    __device__ __forceinline__ uint32_t xadr(int x) const
    {
        return (x & 1) ? (xadr2[x >> 1] >> 14) & 0x3fffcu : (xadr2[x >> 1] << 2) & 0x3fffcu;
    }
    float cprev[PLANES_MAXX];        // sum-product: the message sent along each extra edge (they all differ)
    float pm1, pm2;                  // min-sum: the two magnitudes sent
    uint32_t pidx;                   // edge that got pm2: 0..31 slot edges, 32 + x extra x
    uint32_t psign;                  // bit x: the message on extra x was negative
    float *xbase;                    // this workgroup's scratch rows (uniform)
    uint32_t xoff;                   // my row's element in them: extra x at xbase[xoff + x * PLANES_XROW]
    int X;                           // number of extra edges (uniform)
};

template <int Z, int CPT>
struct PlanesGeom {
    enum { MB = 2, W = 2, NB = CPT * MB, NS = NB * W, N = NB * Z, ZP = (Z + 63) / 64 * 64,
           CB = 20,                                  // bytes per column record
           REC_BYTES = N * CB, SHTAB_OFF = REC_BYTES, SHTAB_BYTES = MB * NB * W * 4,
           FLAG_OFF = SHTAB_OFF + SHTAB_BYTES,
           DUMMY_OFF = FLAG_OFF + 16,                // record every edge of a zeroed circulant points to
           GEN_OFF = DUMMY_OFF + CB,                 // kernels that make their own noise: uncoded errors (4 bytes), then
                                                     // the frame's sum of squared noise (8 bytes, 8-byte aligned)
           LDS_BYTES = (GEN_OFF + 12 + 15) / 16 * 16 };
    static_assert((GEN_OFF + 4) % 8 == 0, "the 64-bit noise sum must be 8-byte aligned");
};

#ifndef PLANES_PRECHECK_T_SPA
#define PLANES_PRECHECK_T_SPA 64    // the same for the sum-product kernel, whose check pass costs four times as much
#endif
#ifndef PLANES_PRECHECK_T
#define PLANES_PRECHECK_T 32        // see planes_syndrome_only; 0 = never
#endif

// Only the parity of a row's hard decisions (slicer + isCodeword, ldpc.py:47, 249-268).  A frame's
// last check pass finds every parity satisfied and all its other work is thrown away; when the pass
// before left no more than PLANES_PRECHECK_T rows unsatisfied, this much cheaper pass is run first
// and the full one only if it still finds a failing row.  Same stop decision, same iteration count.
template <int Z, int CPT, int HV, bool CANON = false>
__device__ __forceinline__ unsigned long long
planes_syndrome_only(const unsigned char *smem, const uint32_t (&radr)[CPT * 4], unsigned long long synd0,
                     const PlanesXState &xs)
{
    enum { NS = CPT * 4, CHS = 8, HEAVY = HV > 0, MX = PlanesHeavy<HV>::MX };
    unsigned long long synd = synd0;
    if (CANON) {
        // a stored posterior is never +0 (a zero is kept as -0): sign bit clear <=> hard decision 1, and over an
        // even number of slots the XOR of the sign bits is the XOR of the hard decisions
        static_assert(!CANON || NS % 2 == 0, "even number of slots");
        uint32_t H = 0;
#pragma unroll
        for (int s0 = 0; s0 < NS; s0 += CHS) {
            float p[CHS];
#pragma unroll
            for (int u = 0; u < CHS; ++u) p[u] = lds_f32(smem, radr[s0 + u]);
#pragma unroll
            for (int u = 0; u < CHS; ++u) H ^= __float_as_uint(p[u]);
        }
        return synd ^ __ballot((int)H < 0);
    }
    if (HEAVY) {
#pragma unroll
        for (int x = 0; x < MX; ++x)
            if (x < xs.X) synd ^= __ballot(lds_f32(smem, xs.xadr(x)) > 0.0f);
    }
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += CHS) {
        float p[CHS];
#pragma unroll
        for (int u = 0; u < CHS; ++u) p[u] = lds_f32(smem, radr[s0 + u]);
#pragma unroll
        for (int u = 0; u < CHS; ++u) synd ^= __ballot(p[u] > 0.0f);
    }
    return synd;
}

// Check pass of one row of block-row B.  FIRST: the row's previous messages are all +0
// (fresh decoder, checkNode.__init__ ldpc.py:121) and are not read.  Returns the lanes
// whose parity check on the current hard decisions fails.
template <int Z, int CPT, int B, bool FIRST, int HV, bool CANON = false>
__device__ __forceinline__ unsigned long long
planes_check_pass(unsigned char *smem, const uint32_t (&radr)[CPT * 4], int D, int sign_mode, int ic,
                  unsigned long long synd0, float alpha, float beta, PlanesXState &xs)
{
    typedef PlanesGeom<Z, CPT> G;
    enum { NS = G::NS, CH = PLANES_CH, NCH = NS / CH, HEAVY = HV > 0, MX = PlanesHeavy<HV>::MX };
    const float INF = __uint_as_float(0x7f800000u);
    float min1 = INF, min2 = INF;
    uint32_t idx = 0, R = 0;
    unsigned long long synd = synd0;     // set for every row if rows have an odd number of missing edges (each reads as hard decision 1)
    uint32_t RX = 0;                     // bit x: v < 0 on extra edge x
    uint32_t LT = 0;                     // one bit per slot edge: it lowered the running minimum
    // CANON: posteriors and messages are stored with -0 in place of either zero.  A posterior's sign bit is then
    // clear exactly when its hard decision is 1, so the row parity is the top bit of the XOR of the words it
    // reads (one instruction of the fast class per edge instead of a compare and a scalar XOR); p - c stays
    // exact: (-0) - (-0) = +0 as (+0) - (+-0), x - (-0) = x, (-0) - x = -x.  The first pass of a frame reads
    // channel values, which are not stored that way, and compares.
    enum { XORP = CANON && !FIRST };
    uint32_t H = 0;
    if (HEAVY) {
        // The extra edges first (minimum, second minimum and sign product do not depend on the order
        // edges are visited in; equal magnitudes make min2 == min1, so neither does the result of argmin).
        float px[MX > 0 ? MX : 1];
#pragma unroll
        for (int x = 0; x < MX; ++x) px[x] = x < xs.X ? lds_f32(smem, xs.xadr(x)) : 0.0f;
#pragma unroll
        for (int x = 0; x < MX; ++x) {
            if (x < xs.X) {
                synd ^= __ballot(px[x] > 0.0f);
                float v = px[x];
                if (!FIRST) {
                    const uint32_t mag = xs.pidx == (uint32_t)(NS + x) ? __float_as_uint(xs.pm2) : __float_as_uint(xs.pm1);
                    v = px[x] - __uint_as_float(mag | (((xs.psign >> x) & 1u) << 31));     // receive, ldpc.py:161
                }
                RX |= (__float_as_uint(v) >> 31) << x;
                const float av = fabsf(v);
                idx = av < min1 ? (uint32_t)(NS + x) : idx;
                min2 = __builtin_amdgcn_fmed3f(min1, min2, av);
                min1 = fminf(min1, av);
            }
        }
    }
    // Software pipeline over chunks of CH edges: the LDS reads of chunk k+1 are in flight
    // while chunk k is reduced, so a wave's own LDS latency is covered even when its three
    // SIMD neighbours are waiting too (right after a barrier all of them are).
    float p[2][CH], c[2][CH];
#pragma unroll
    for (int k = -1; k < NCH; ++k) {
        if (k + 1 < NCH) {
            const int nb = (k + 1) & 1;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int s = (k + 1) * CH + u;
                c[nb][u] = 0.0f;
                {
                    p[nb][u] = lds_f32(smem, radr[s]);
                    if (!FIRST) c[nb][u] = lds_f32(smem, radr[s] + 4u * (1 + 2 * B + (s & 1)));
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (k < 0) continue;
        // The issue arbiter favours the oldest wave, which would let each SIMD's four waves
        // finish the pass one after the other, the last one alone with its LDS latency
        // exposed.  Dropping priority as the pass advances keeps the waves level
        // (measured: 11 % off the iteration time; finer steps than one per 8 edges were slower).
        if (PLANES_PRIO && k * CH % 8 == 0) set_wave_prio(3 - (k * CH / 8) * 3 / (NS / 8 - 1));
        const int cb = k & 1;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            {
                if (XORP) H ^= __float_as_uint(p[cb][u]);
                else synd ^= __ballot(p[cb][u] > 0.0f);                     // slicer, ldpc.py:47
                const float v = FIRST ? p[cb][u] : p[cb][u] - c[cb][u];     // receive, ldpc.py:161 (x - 0 == x)
                R = __builtin_amdgcn_alignbit(R, __float_as_uint(v), 31);   // (R << 1) | sign(v), ldpc.py:139
                // second minimum and minimum as bare instructions: behind the register pins below the
                // compiler no longer knows that min1/min2 are canonical and would put a v_max x, x
                // in front of its own med3 / min once per chunk  (|v|: ldpc.py:178)
                float n2, n1;
                asm("v_med3_f32 %0, %1, %2, |%3|" : "=v"(n2) : "v"(min1), "v"(min2), "v"(v));
                asm("v_min_f32 %0, %1, |%2|" : "=v"(n1) : "v"(min1), "v"(v));
                // arg-min: one bit per edge, "this edge lowered the minimum", shifted into LT through the
                // carry -- the integer difference of the new and the old minimum borrows exactly when
                // |v| < min1 (non-negative floats order like their bit patterns).  Two full-rate
                // instructions where a compare and a select are two half-rate ones; the edge itself is
                // read off the lowest set bit after the loop.
#if PLANES_ARGMIN_FSUB
                // the same bit from the float difference: new minimum - old minimum is negative exactly when
                // the edge lowered it (x - x = +0; distinct floats never subtract to zero), and the
                // subtraction is in the instruction class that issues twice as fast as the carry forms
                float dlt;
                asm("v_sub_f32 %0, %1, %2" : "=v"(dlt) : "v"(n1), "v"(min1));
                LT = __builtin_amdgcn_alignbit(LT, __float_as_uint(dlt), 31);
#else
                uint32_t borrow_tmp;
                asm("v_sub_co_u32 %1, vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                    : "+v"(LT), "=&v"(borrow_tmp) : "v"(n1), "v"(min1) : "vcc");
#endif
                min2 = n2;
                min1 = n1;
            }
            // Pin the running state here.  Otherwise instruction selection defers the whole
            // second-minimum / argmin / syndrome chains to the end of the pass and keeps every
            // |v|, every intermediate minimum and every compare mask alive until then
            // (60+ vector and 60+ scalar registers, spills).
            if ((u & 3) == 3) {
                if (XORP) asm volatile("" : "+v"(min1), "+v"(min2), "+v"(LT), "+v"(R), "+v"(H));
                else asm volatile("" : "+v"(min1), "+v"(min2), "+v"(LT), "+v"(R), "+s"(synd));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (XORP) synd ^= __ballot((int)H < 0);
    // the last slot edge that lowered the minimum is the arg-min (bit NS-1-s of LT belongs to edge s),
    // unless no slot edge went below what the extra edges left
    if (LT != 0u) idx = (uint32_t)(NS - 1) - (uint32_t)__builtin_ctz(LT);
    // R: D sign bits, first edge highest.  Left-align and fold in the product of all signs:
    // a bit of T is then the sign of the outgoing message itself (ldpc.py:190)
    if (NS < 32) R <<= (32 - NS);
    uint32_t S = __popc(R) & 1u;
    if (HEAVY) S ^= __popc(RX) & 1u;
    if (sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)D & 1u;       // D: the row's real degree
    const uint32_t full = 0xffffffffu << (32 - NS);
    uint32_t T = R ^ (S ? full : 0u);
    if (alpha != 1.0f || beta != 0.0f) {                    // normalised / offset min-sum (uniform branch)
        min1 = ms_correct(min1, alpha, beta);
        min2 = ms_correct(min2, alpha, beta);
    }
    const uint32_t m1 = __float_as_uint(min1), m2 = __float_as_uint(min2);
    const uint32_t T0 = T;                                  // the signs as they are, for the arg-min edge
    if (CANON) T |= m1 == 0u ? full : 0u;                   // a zero message is stored as -0
    if (HEAVY) {
        // extra edges: +-min1, or +-min2 on the arg-min edge (ldpc.py:190-191), to the scratch row
        const uint32_t SX = RX ^ (S ? 0xffffffffu : 0u);
#pragma unroll
        for (int x = 0; x < MX; ++x)
            if (x < xs.X)
                xs.xbase[opaque(xs.xoff) + (uint32_t)(x * PLANES_XROW)] =
                    __uint_as_float((idx == (uint32_t)(NS + x) ? m2 : m1) | (((SX >> x) & 1u) << 31));
        xs.pm1 = min1; xs.pm2 = min2; xs.pidx = idx; xs.psign = SX;
    }
    // shift of the arg-min edge, needed after the loop below (idx differs from lane to lane)
    const uint32_t ej = (HEAVY ? idx & (uint32_t)(NS - 1) : idx) >> 1, ek = idx & 1u;
    const int esh = *reinterpret_cast<const int *>(smem + G::SHTAB_OFF + ((B * G::NB + ej) * 2 + ek) * 4);
    // every edge gets +-min1 ...
#if PLANES_OUT_ALIGNBIT
    // Sign and magnitude joined by one funnel shift: { T >> (31 - s), 2 * min1 } >> 1 has the edge's bit of T on
    // top of the 31 magnitude bits.  A right shift issues twice as fast as the left shift and the and-or
    // it replaces, and the funnel shift stands for the second of them.
    const uint32_t m1x = m1 << 1;          // min1 >= 0: nothing is shifted out
#endif
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        {
#if PLANES_OUT_ALIGNBIT
            const uint32_t ts = s == 31 ? T : T >> (31 - s);
            uint32_t o;
            asm("v_alignbit_b32 %0, %1, %2, 1" : "=v"(o) : "v"(ts), "v"(m1x));
            lds_put_u32(smem, radr[s] + 4u * (1 + 2 * B + (s & 1)), o);
#else
            const uint32_t rank = (uint32_t)s;
            lds_put_u32(smem, radr[s] + 4u * (1 + 2 * B + (s & 1)), m1 | ((T << rank) & 0x80000000u));
#endif
        }
        if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // ... except the edge of the smallest magnitude, which gets +-min2 (ldpc.py:191)
    {
        int ci = ic + esh;
        ci = ci >= Z ? ci - Z : ci;
        const uint32_t adr = (uint32_t)(ej * Z + ci) * G::CB + 4u * (1 + 2 * B) + 4u * ek;
        const uint32_t rank = idx & 31u;
        if (D > 0 && (!HEAVY || idx < (uint32_t)NS))          // D == 0: every edge is a dummy; HEAVY: the arg-min may be an extra edge
            lds_put_u32(smem, adr, m2 | ((T0 << rank) & 0x80000000u) | (CANON && m2 == 0u ? 0x80000000u : 0u));
    }
    return synd;
}

// phi(x) = -ln tanh(x/2) for x in [1e-12, 80] from the hardware exp2/log2/rcp units, relative
// error a few 1e-7 (each branch is used where it is well conditioned):
//   x >= 2        2 atanh(t) as a series in t = e^-x            (log of a number near 1 would lose digits)
//   x <  2        ln((1 + t) / (1 - t)), with 1 - t from its series below 1/16 (cancellation)
__device__ __forceinline__ float planes_phi(float x)
{
    // explicit fused multiply-adds: the file is compiled without contraction (the min-sum kernels'
    // rounding order is part of their contract), which would otherwise split every polynomial step in two
    // Bare v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp each): __expf, __logf and __frcp_rn expand to range
    // checks, rescaling for denormals and a full division sequence, none of which this range needs
    // (e^-80 = 1.8e-35 is a normal number; the logarithm's argument is >= 1).
    const float t = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    const float t2 = t * t;
    const float series = (t + t) * __fmaf_rn(t2, __fmaf_rn(t2, __fmaf_rn(t2, 0.14285715f, 0.2f), 0.33333334f), 1.0f);
    const float om_small = x * __fmaf_rn(-x, __fmaf_rn(-x, __fmaf_rn(-x, __fmaf_rn(-x, 0.008333334f, 0.041666668f),
                                                                     0.16666667f), 0.5f), 1.0f);
    const float om = x < 0.0625f ? om_small : 1.0f - t;
    const float lg = __builtin_amdgcn_logf((1.0f + t) * __builtin_amdgcn_rcpf(om)) * 0.6931471805599453f;
    return x >= 2.0f ? series : lg;
}

// Sum-product check pass (the rule stated with csr_spa_kernel): same structure as the min-sum
// pass; the message slot of an edge holds F_e = phi|v_e| between the two halves.
// HOLES: some edge of the row may be missing (its record is the dummy, |v| = inf): such an edge adds no term.
// HV > 0: the row's extra edges (heavy circulants) come first; the F of each stays in a register between the
// two halves, its outgoing message goes to the scratch row (and stays in xs.cprev for the next iteration).
template <int Z, int CPT, int B, bool FIRST, bool HOLES, int HV = 0>
__device__ __forceinline__ unsigned long long
planes_check_pass_spa(unsigned char *smem, const uint32_t (&radr)[CPT * 4], int D, int sign_mode,
                      unsigned long long synd0, PlanesXState &xs)
{
    typedef PlanesGeom<Z, CPT> G;
    enum { NS = G::NS, CH = 4, NCH = NS / CH, HEAVY = HV > 0, MX = PlanesHeavy<HV>::MX };
    float F1 = 0.0f, rest = 0.0f;          // F > 0 always, so the first edge displaces the initial 0 (rest += 0)
    uint32_t GT = 0, R = 0;
    unsigned long long synd = synd0;     // set for every row if rows have an odd number of missing edges (each reads as hard decision 1)
    uint32_t RX = 0, idxx = 0;           // extra edges: bit x = v < 0; the one that holds the largest term so far (NS + x)
    float fx[MX > 0 ? MX : 1];
    if (HEAVY) {
        float px[MX > 0 ? MX : 1];
#pragma unroll
        for (int x = 0; x < MX; ++x) px[x] = x < xs.X ? lds_f32(smem, xs.xadr(x)) : 0.0f;
#pragma unroll
        for (int x = 0; x < MX; ++x) {
            fx[x] = 0.0f;
            if (x < xs.X) {
                synd ^= __ballot(px[x] > 0.0f);
                const float v = FIRST ? px[x] : px[x] - xs.cprev[x];
                RX |= (__float_as_uint(v) >> 31) << x;
                const float F = planes_phi(__builtin_amdgcn_fmed3f(fabsf(v), 1e-12f, 80.0f));
                fx[x] = F;
                idxx = F > F1 ? (uint32_t)(NS + x) : idxx;
                const float hi = fmaxf(F1, F), lo = fminf(F1, F);
                rest = rest + lo;
                F1 = hi;
            }
        }
    }
    float p[2][CH], c[2][CH];
#pragma unroll
    for (int k = -1; k < NCH; ++k) {
        if (k + 1 < NCH) {
            const int nb = (k + 1) & 1;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int s = (k + 1) * CH + u;
                c[nb][u] = 0.0f;
                {
                    p[nb][u] = lds_f32(smem, radr[s]);
                    if (!FIRST) c[nb][u] = lds_f32(smem, radr[s] + 4u * (1 + 2 * B + (s & 1)));
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (k < 0) continue;
        if (PLANES_PRIO && k * CH % 8 == 0) set_wave_prio(3 - (k * CH / 8) * 3 / (NS / 8 - 1));
        const int cb = k & 1;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int s = k * CH + u;
            {
                synd ^= __ballot(p[cb][u] > 0.0f);
                const float v = FIRST ? p[cb][u] : p[cb][u] - c[cb][u];
                R = __builtin_amdgcn_alignbit(R, __float_as_uint(v), 31);
                const float av = fabsf(v);
                float F = planes_phi(__builtin_amdgcn_fmed3f(av, 1e-12f, 80.0f));
                if (HOLES) F = av == __uint_as_float(0x7f800000u) ? 0.0f : F;       // edge of a zeroed circulant: no term
                lds_put_f32(smem, radr[s] + 4u * (1 + 2 * B + (s & 1)), F);
                // largest term so far and the sum of the others: the smaller of { F1, F } joins the sum, the larger
                // stays apart; the edge that holds it is the last one that raised it strictly (one bit per edge,
                // the sign of old maximum - new maximum, first edge highest)
                float hi, lo, dgt;
                asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(F1), "v"(F));
                asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(F1), "v"(F));
                asm("v_sub_f32 %0, %1, %2" : "=v"(dgt) : "v"(F1), "v"(hi));
                GT = __builtin_amdgcn_alignbit(GT, __float_as_uint(dgt), 31);
                rest = rest + lo;
                F1 = hi;
            }
        }
        asm volatile("" : "+v"(F1), "+v"(rest), "+v"(GT), "+v"(R), "+s"(synd));
        __builtin_amdgcn_sched_barrier(0);
    }
    if (NS < 32) GT <<= (32 - NS);
    // the last edge that raised the largest term strictly holds it: a slot edge, or else an extra edge
    const uint32_t idx = GT != 0u ? 31u - (uint32_t)__builtin_ctz(GT) : (HEAVY ? idxx : 0u);
    if (NS < 32) R <<= (32 - NS);
    uint32_t S = __popc(R) & 1u;
    if (HEAVY) S ^= __popc(RX) & 1u;
    if (sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)D & 1u;       // D: the row's real degree
    const uint32_t full = 0xffffffffu << (32 - NS);
    const uint32_t T = R ^ (S ? full : 0u);
    set_wave_prio(0);
    float f[2][CH];
#pragma unroll
    for (int k = -1; k < NCH; ++k) {
        if (k + 1 < NCH) {
            const int nb = (k + 1) & 1;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int s = (k + 1) * CH + u;
                f[nb][u] = 0.0f;
                f[nb][u] = lds_f32(smem, radr[s] + 4u * (1 + 2 * B + (s & 1)));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (k < 0) continue;
        const int cb = k & 1;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int s = k * CH + u;
            {
                const uint32_t rank = (uint32_t)s;
                const float x = idx == (uint32_t)s ? rest : (rest - f[cb][u]) + F1;
                const float mag = planes_phi(__builtin_amdgcn_fmed3f(x, 1e-12f, 80.0f));
                lds_put_u32(smem, radr[s] + 4u * (1 + 2 * B + (s & 1)), __float_as_uint(mag) | ((T << rank) & 0x80000000u));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (HEAVY) {
        const uint32_t SX = RX ^ (S ? 0xffffffffu : 0u);
#pragma unroll
        for (int x = 0; x < MX; ++x)
            if (x < xs.X) {
                const float xe = idx == (uint32_t)(NS + x) ? rest : (rest - fx[x]) + F1;
                const float mag = planes_phi(__builtin_amdgcn_fmed3f(xe, 1e-12f, 80.0f));
                const float msg = __uint_as_float(__float_as_uint(mag) | (((SX >> x) & 1u) << 31));
                xs.cprev[x] = msg;
                xs.xbase[opaque(xs.xoff) + (uint32_t)(x * PLANES_XROW)] = msg;
            }
    }
    return synd;
}

// The whole decode loop for the threads of block-row / column-group B.  It is instantiated
// once per B and the kernel branches ONCE (wave-uniformly) into its copy: with one shared
// loop and a branch per pass the optimiser merges the common tails of the copies and
// carries every per-edge intermediate across the merge in registers.
// GEN: the kernel makes its own channel values (BPSK/AWGN of the all-zero codeword, awgn.h) instead
// of reading them: the sweep driver then touches HBM for its per-frame results only.
template <int Z, int CPT, int B, bool SPA, bool GEN, int HV>
__device__ __forceinline__ void
planes_role(unsigned char *smem, const PlanesTables &tb, const DecodeArgs &a)
{
    enum { HEAVY = HV > 0, MX = PlanesHeavy<HV>::MX, MW = PlanesHeavy<HV>::MW, CANON = PLANES_CANON && !SPA && !HEAVY };
    typedef PlanesGeom<Z, CPT> G;
    enum { MB = G::MB, W = G::W, NB = G::NB, NS = G::NS, N = G::N, ZP = G::ZP, CB = G::CB,
           NT = MB * G::ZP };                    // threads per workgroup
    static_assert(MB * ZP <= 1024 && NS <= 32, "geometry");
    static_assert(3 * MB * Z * CB + 4 * 4 < 65536, "column record offsets must fit the DS immediate");
    int *shtab = reinterpret_cast<int *>(smem + G::SHTAB_OFF);
    int *flags = reinterpret_cast<int *>(smem + G::FLAG_OFF);    // [2] "syndrome non-zero", [2] ones counters

    const int tid = threadIdx.x;
    const int g = B;                                             // block-row served / first block-column served
    const int i = tid - g * ZP;
    const bool live = i < Z;
    const int ic = live ? i : 0;
    const int lane = tid & 63;
    const int nbytes = (N + 7) >> 3;

    // ---- per-launch set-up
    for (int x = tid; x < MB * NB * W; x += NT) shtab[x] = tb.shifts[x];
    const uint32_t emask = g == 0 ? tb.edge_mask[0] : tb.edge_mask[1];   // existing edges of a row of my block-row
    PlanesXState xs;
    xs.X = HEAVY ? tb.n_extra[B] : 0;
    int n_heavy = HEAVY ? tb.n_heavy[B] : 0;
    if (HEAVY) asm volatile("" : "+s"(xs.X), "+s"(n_heavy));          // kept in scalar registers, not reloaded from the arguments
    xs.pm1 = xs.pm2 = 0.0f; xs.pidx = 0; xs.psign = 0; xs.xbase = nullptr; xs.xoff = 0;
    if (HEAVY) {
        xs.xbase = tb.xmsg + (size_t)blockIdx.x * tb.xstride;
        xs.xoff = (uint32_t)(B == 0 ? 0 : tb.n_extra[0]) * PLANES_XROW + (uint32_t)ic;
#pragma unroll
        for (int x = 0; x < MX; ++x) {
            const int e = tb.extra[B][x];
            int c = ic + (e >> 8);
            c = c >= Z ? c - Z : c;
            const uint32_t adr = x < xs.X ? (uint32_t)((e & 0xff) * Z + c) * CB : (uint32_t)G::DUMMY_OFF;
            if (x & 1) xs.xadr2[x >> 1] |= (adr >> 2) << 16;
            else xs.xadr2[x >> 1] = adr >> 2;
        }
        static_assert(G::LDS_BYTES < (1 << 18) && G::DUMMY_OFF % 4 == 0 && CB % 4 == 0, "packed record addresses");
    }
    const int D = __popc(emask) + xs.X;                                 // the row's real degree
    const unsigned long long synd0 = ((NS - __popc(emask)) & 1) ? __ballot(live) : 0ull;   // only lanes that own a row
    // once per launch: message slots (those of zeroed circulants are never written again) and the dummy record
    for (int x = tid; x < N; x += NT)
#pragma unroll
        for (int k = 1; k <= 4; ++k) lds_put_f32(smem, (uint32_t)x * CB + 4u * k, 0.0f);
    if (tid < 5) lds_put_u32(smem, G::DUMMY_OFF + 4u * tid, tid == 0 ? 0x7f800000u : 0u);
    uint32_t radr[NS];                       // record address of the column each of my row's edges goes to
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int sh = tb.shifts[(g * NB + (s >> 1)) * W + (s & 1)];
        int c = ic + sh;
        c = c >= Z ? c - Z : c;
        // laundered: otherwise the per-block constant is split off and re-added at every use
        radr[s] = opaque(((emask >> s) & 1u) ? (uint32_t)((s >> 1) * Z + c) * CB : (uint32_t)G::DUMMY_OFF);
    }
    // my columns: block-column g + q*MB, position ic; records q and q+4 share a base register
    const uint32_t cadr0 = (uint32_t)(g * Z + ic) * CB;
    const uint32_t cadr1 = cadr0 + 4u * (MB * Z * CB);
    // bit q: in block (1, j_q) the row of the lower shift has the lower index
    uint32_t lomask = 0;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int j = g + q * MB;
        const int s0 = tb.shifts[(1 * NB + j) * W + 0], s1 = tb.shifts[(1 * NB + j) * W + 1];
        if (ic >= s0 && ic < s1) lomask |= 1u << q;
    }

    // HEAVY: columns of a block-column that holds a heavy circulant are summed in full by these two
    // steps.  Per block-row the column's messages are added in ascending check index
    // (ldpc.py:298-303): with the block's shifts ascending and t of them <= c (the position of the
    // column in its block), the rows c - s of the first t do not wrap and ascend as s descends, then
    // come the wrapped rows c - s + Z of the others, again as s descends: s_(t-1) .. s_0, s_(w-1) .. s_t.
    // heavy_fetch reads the messages (slots from LDS, extra edges from the scratch rows in global
    // memory), heavy_sum adds them -- absent ones as +0, which leaves a sum that started from +0
    // unchanged bit for bit -- and writes the posterior.  No data-dependent branch, one table load.
    const float *xw = HEAVY ? tb.xmsg + (size_t)blockIdx.x * tb.xstride : nullptr;
    auto heavy_fetch = [&](int h, int (&t)[16], float (&m)[2][MW]) {
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = tb.hent[B][h][k];
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+s"(t[k]));       // scalar registers from here on, not reloaded
        const int icl = (int)opaque((uint32_t)ic);
        const uint32_t adr = (uint32_t)(g * Z + icl) * CB + (uint32_t)((uint32_t)t[0] >> 29) * (MB * Z * CB);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            m[b][0] = lds_f32(smem, adr + 4u * (uint32_t)(1 + 2 * b));
            m[b][1] = lds_f32(smem, adr + 4u * (uint32_t)(2 + 2 * b));
            const int w = t[8 * b] & 0xff;
#pragma unroll
            for (int k = 2; k < MW; ++k) m[b][k] = 0.0f;
            if (w > 2) {
#pragma unroll
                for (int k = 2; k < MW; ++k) {
                    const int e = t[8 * b + k - 1];
                    int r = icl - (e & 0xffff);
                    r = r < 0 ? r + Z : r;
                    const float v = xw[(uint32_t)((e >> 16) * PLANES_XROW + r)];
                    m[b][k] = k < w ? v : 0.0f;
                }
            }
        }
    };
    auto heavy_sum = [&](const int (&t)[16], const float (&m)[2][MW], const float (&y)[CPT]) {
        const int icl = (int)opaque((uint32_t)ic);
        const int q = (int)((uint32_t)t[0] >> 29);
        float acc = 0.0f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int w0 = t[8 * b];
            int nw = (((w0 >> 8) & 0x1ff) <= icl) + (((w0 >> 20) & 0x1ff) <= icl);     // shifts that do not wrap at my position
#pragma unroll
            for (int k = 2; k < MW; ++k) nw += (t[8 * b + k - 1] & 0xffff) <= icl;
#pragma unroll
            for (int k = MW - 1; k >= 0; --k) acc = acc + (k < nw ? m[b][k] : 0.0f);
#pragma unroll
            for (int k = MW - 1; k >= 0; --k) acc = acc + (k >= nw ? m[b][k] : 0.0f);
        }
        float yq = y[0];
#pragma unroll
        for (int qq = 1; qq < CPT; ++qq) {
            yq = q == qq ? y[qq] : yq;
            asm volatile("" : "+v"(yq));          // a chain of selects, not a table in scratch memory
        }
        lds_put_f32(smem, (uint32_t)(g * Z + icl) * CB + (uint32_t)q * (MB * Z * CB), acc + yq);     // channel last, ldpc.py:322
    };

    // variable pass (ldpc.py:295-311, 321-323): columns (g + q*MB, ic)
    auto regular_columns = [&](const float (&y)[CPT]) {
        enum { VB = PLANES_VB };                               // columns per batch: loads first, then sums
#pragma unroll
        for (int q0 = 0; q0 < CPT; q0 += VB) {
            set_wave_prio(VB < CPT ? 3 - q0 * 3 / (CPT - VB + (VB == CPT)) : 3);
            float m[VB][4];
#pragma unroll
            for (int u = 0; u < VB; ++u) {
                const int q = q0 + u;
                const uint32_t adr = (q < 4 ? cadr0 : cadr1) + (uint32_t)(q & 3) * (MB * Z * CB);
#pragma unroll
                for (int k = 0; k < 4; ++k) m[u][k] = lds_f32(smem, adr + 4u * (k + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < VB; ++u) {
                const int q = q0 + u;
                const uint32_t adr = (q < 4 ? cadr0 : cadr1) + (uint32_t)(q & 3) * (MB * Z * CB);
                // first two terms of the column sum: 0 + x + y == x + y == y + x
                float acc = m[u][0] + m[u][1];
                const bool lo = (lomask >> q) & 1u;
                acc = acc + (lo ? m[u][2] : m[u][3]);             // lower check index first
                acc = acc + (lo ? m[u][3] : m[u][2]);
                float post = acc + y[q];                          // channel last, ldpc.py:322
                if (CANON) {
                    // -(0 - post): post itself, except that +0 becomes -0 (as bare instructions: the compiler
                    // would fold 0 - x into a negation and the two negations away)
                    float t;
                    asm("v_sub_f32 %0, 0, %1" : "=v"(t) : "v"(post));
                    post = __uint_as_float(__float_as_uint(t) ^ 0x80000000u);
                }
                lds_put_f32(smem, adr, post);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto variable_pass = [&](const float (&y)[CPT]) {
        if (!HEAVY) {
            regular_columns(y);
        } else {
            // The first heavy block-column's messages are requested before the regular columns, whose
            // work covers the latency of the scratch rows, and added after them (what the regular pass
            // wrote for a heavy column is overwritten here, by the same thread).
            float hm[2][MW];
            int ht[16];
#pragma unroll 1
            for (int h = 0; h == 0 || h < n_heavy; ++h) {
                if (h < n_heavy) heavy_fetch(h, ht, hm);
                if (h == 0) regular_columns(y);
                if (h < n_heavy) heavy_sum(ht, hm, y);
            }
        }
    };

    const int64_t n_frames = a.n_frames;
    // Channel values: y of the frame being decoded, ynext of the frame after it.
    float y[CPT], ynext[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) ynext[q] = 0.0f;
    if (!GEN && live && (int64_t)blockIdx.x < n_frames)
        planes_load_frame<CPT>(a.llr, a.llr_f64, (int64_t)blockIdx.x * N, (uint32_t)(g * Z + ic), MB * Z, ynext);
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        // "+ 0": -0.0 becomes +0.0 (invisible to every comparison and sum below; keeps
        // "sign bit set" equivalent to "< 0" for every message)
        y[q] = ynext[q] * a.llr_scale + 0.0f;
    }
    if (tid < 4) flags[tid] = 0;
    if (GEN && tid == 0) {
        lds_put_u32(smem, G::GEN_OFF, 0u);
        *reinterpret_cast<unsigned long long *>(smem + G::GEN_OFF + 4) = 0ull;
    }
    if (GEN) lds_barrier();
    // GEN: the frame's channel values go straight into the posterior slots, position by position
    // (one generator call makes 4 consecutive positions; thread t takes calls t and t + NT), with
    // the same keying and arithmetic as the stand-alone channel kernel, so both give the same frame.
    static_assert(!GEN || N % 4 == 0, "generator calls cover 4 positions each");
    auto generate_frame = [&](int64_t frame) {
        uint32_t sq = 0;                 // this thread's 8 samples: < 2^28 in 2^-19 fixed point
        int err = 0;                     // wave-uniform: channel hard-decision errors of the wave
        unsigned long long noise_frame = a.gen_first_frame + (uint64_t)frame;
        uint32_t noise_stream = a.gen_stream;
        float sigma = a.gen_sigma, scale = a.llr_scale;
        if (a.gen_per_point) {           // several sweep points in this launch (at most GEN_MAX_BATCH frames in all)
            const uint32_t pt = (uint32_t)frame / a.gen_per_point;
            noise_frame = a.gen_first_frame + (uint64_t)((uint32_t)frame - pt * a.gen_per_point);
            noise_stream += pt;
            sigma = a.gen_point_sigma[pt];
            scale = a.gen_point_scale[pt];
        }
        // Round keys come from the seed by scalar additions where they are used.  Left loop-invariant they
        // are 20 values held across the whole decode, which the register allocator parks in lanes of a
        // vector register and fetches back one by one (a fifth of this function's vector instructions).
        uint32_t seed_lo = (uint32_t)a.gen_seed, seed_hi = (uint32_t)(a.gen_seed >> 32);
        asm volatile("" : "+s"(seed_lo), "+s"(seed_hi));
        enum { NCALL = (N / 4 + NT - 1) / NT };
        uint32_t qb[NCALL];
        float4 zc[NCALL];
#pragma unroll
        for (int hh = 0; hh < NCALL; ++hh) qb[hh] = (uint32_t)tid + (uint32_t)hh * NT;
        normal4_calls<NCALL>(seed_lo, seed_hi, noise_frame, qb, noise_stream, zc);   // calls beyond the frame's end: made, not used
#pragma unroll
        for (int hh = 0; hh < NCALL; ++hh) {
            if (qb[hh] < (uint32_t)(N / 4)) {
                const float zz[4] = {zc[hh].x, zc[hh].y, zc[hh].z, zc[hh].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float yv = -1.0f + sigma * zz[k];
                    sq += awgn_sq_fixed(zz[k]);
                    err += (int)__popcll(__ballot(yv > 0.f));
                    // no "+ 0" as for values that come from memory: yv is never -0.0 (a sum is -0.0 only if
                    // both terms are) and |yv| >= 2^-24 or 0, so the product cannot be -0.0 either
                    lds_put_f32(smem, (4u * qb[hh] + (uint32_t)k) * CB, yv * scale);
                }
            }
        }
        sq = awgn_row_sum16(sq);         // < 2^32 per row of 16 lanes
        const unsigned long long wsq = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)sq, 15) +
                                       (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)sq, 31) +
                                       (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)sq, 47) +
                                       (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)sq, 63);
        if (lane == 0) {
            atomicAdd(reinterpret_cast<int *>(smem + G::GEN_OFF), err);
            atomicAdd(reinterpret_cast<unsigned long long *>(smem + G::GEN_OFF + 4), wsq);
        }
    };
    int cur_unc = 0, out_unc = 0;
    unsigned long long cur_sq = 0, out_sq = 0;
    uint32_t n_full = 0, n_synd = 0;             // check passes of this workgroup (a.pass_counts)

    // Results of the previous frame, held in registers until the next frame's first check pass
    // is under way.  A CU takes thousands of cycles to queue one frame's worth of global
    // loads and stores (measured: a workgroup that issues them between two barriers idles
    // for 2-3 us per frame); issued here they queue while the other waves do arithmetic.
    unsigned long long out_bits = 0;             // lane t < 8: hard decisions of columns 64*wave + NT*t ...
    int64_t out_frame = -1;
    int out_iters = 0, out_conv = 0, out_ones = 0;
    auto flush_outputs = [&]() {
        if (out_frame < 0) return;
        if (a.hard && lane < 8) {
            const int b0 = (int)(opaque((uint32_t)(tid & ~63)) + (uint32_t)lane * NT) >> 3;
            uint8_t *dst = a.hard + out_frame * nbytes + b0;
            if (b0 + 8 <= nbytes) {
                __builtin_memcpy(dst, &out_bits, 8);
            } else {
                // the frame's last bytes: at most three stores (4 + 2 + 1 bytes), no loop -- the wave that
                // owns them is the one every other wave of the workgroup waits for at the next barrier
                const int r = nbytes - b0;
                int k = 0;
                if (r & 4) { const uint32_t w = (uint32_t)out_bits; __builtin_memcpy(dst, &w, 4); k = 4; }
                if (r & 2) { const uint16_t w = (uint16_t)(out_bits >> (8 * k)); __builtin_memcpy(dst + k, &w, 2); k += 2; }
                if (r & 1) dst[k] = (uint8_t)(out_bits >> (8 * k));
            }
        }
        if (tid == 0) {
            if (a.iters) a.iters[out_frame] = out_iters;
            if (a.conv) a.conv[out_frame] = (uint8_t)out_conv;
            if (a.ones) a.ones[out_frame] = out_ones;
            if (GEN && a.gen_uncoded) a.gen_uncoded[out_frame] = out_unc;
            if (GEN && a.gen_sumsq) a.gen_sumsq[out_frame] = out_sq;
        }
    };

#ifdef LDPC_TRACE
    // phase timing without perturbing the steady state: clock samples are summed in
    // (scalar) registers and written once when the workgroup has finished
    unsigned long long tr[11] = {0}, tr_acc[12] = {0}, tr_last = 0;
    unsigned long long it_acc[4] = {0}, it_t = 0;           // steady-state loop: check, barrier, variable, barrier
#define TRACE(k) tr[k] = clock64()
#define TRACE_IT(k) { const unsigned long long now_ = clock64(); if (k >= 0) it_acc[k < 0 ? 0 : k] += now_ - it_t; it_t = now_; }
#else
#define TRACE(k)
#define TRACE_IT(k)
#endif
    for (int64_t frame = blockIdx.x, next_frame = 0; frame < n_frames; frame = next_frame) {
        TRACE(0);
        if (GEN) {
            generate_frame(frame);
        } else {
#pragma unroll
            for (int q = 0; q < CPT; ++q)
                if (live) lds_put_f32(smem, (q < 4 ? cadr0 : cadr1) + (uint32_t)(q & 3) * (MB * Z * CB), y[q]);
        }
        if (tid < 3) flags[tid] = 0;                      // two stop flags, ones counter
        int my_iters = 0, my_conv = 0;
        TRACE(1);
        lds_barrier();
        TRACE(2);
        if (GEN) {
            // my columns' channel values, for the variable passes; the frame's channel statistics
#pragma unroll
            for (int q = 0; q < CPT; ++q)
                y[q] = lds_f32(smem, (q < 4 ? cadr0 : cadr1) + (uint32_t)(q & 3) * (MB * Z * CB));
            cur_unc = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(smem + G::GEN_OFF));
            const unsigned long long sqv = *reinterpret_cast<const unsigned long long *>(smem + G::GEN_OFF + 4);
            cur_sq = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(sqv >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sqv);
        }
        // claim the frame after this one; the answer is needed only after the check pass below
        int claimed = 0;
        if (a.work && tid == 0) claimed = atomicAdd(a.work, 1);
        flush_outputs();
        TRACE(3);

        // ---- iteration 0 is peeled: its check pass has no previous messages to read
        {
            if (live) {
                const unsigned long long synd = SPA
                    ? (D - xs.X == NS ? planes_check_pass_spa<Z, CPT, B, true, false, HV>(smem, radr, D, a.sign_mode, synd0, xs)
                                      : planes_check_pass_spa<Z, CPT, B, true, true, HV>(smem, radr, D, a.sign_mode, synd0, xs))
                    : planes_check_pass<Z, CPT, B, true, HV, CANON>(smem, radr, D, a.sign_mode, ic, synd0, a.ms_alpha, a.ms_beta, xs);
                // flags[parity]: number of rows whose check fails (bit 20: raised by the syndrome-only pass)
                if (synd != 0ull && lane == 0) atomicAdd(&flags[0], (int)__popcll(synd));
            }
            if (a.work && tid == 0) flags[3] = claimed;
            ++n_full;
            TRACE(4);
            if (HEAVY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the extra edges' messages have left for the scratch rows
            lds_barrier();
            TRACE(5);
            next_frame = frame + gridDim.x;
            if (a.work) next_frame = (int64_t)gridDim.x + __builtin_amdgcn_readfirstlane(flags[3]);
            // the next frame's channel values: requested now, used when this frame is done
            if (!GEN && live && next_frame < n_frames)
                planes_load_frame<CPT>(a.llr, a.llr_f64, next_frame * N, (uint32_t)(g * Z + ic), MB * Z, ynext);
            const int bad = __builtin_amdgcn_readfirstlane(flags[0]);
            if (bad == 0 || a.max_iters <= 0) {
                my_conv = (bad == 0);
            } else {
                // HEAVY: the variable pass opens the loop body instead of closing it, so that the kernel
                // holds one copy of it (code size: two CUs share an instruction cache)
                enum { ROT = HEAVY };
                if (!ROT) {
                    if (tid == 0) flags[1] = 0;
                    if (live) variable_pass(y);
                    lds_barrier();
                }
                TRACE_IT(-1);
                int unsat = bad;                              // rows left unsatisfied by the check pass before
                const int precheck_t = a.precheck >= 0 ? a.precheck : (SPA ? PLANES_PRECHECK_T_SPA : PLANES_PRECHECK_T);   // 0: never (unsat > 0 here)
                for (int it = 1;; ++it) {
                    const int par = it & 1;
                    if (ROT) {
                        if (tid == 0) flags[par] = 0;
                        if (live) variable_pass(y);
                        TRACE_IT(2);
                        lds_barrier();
                        TRACE_IT(3);
                    }
                    if (unsat <= precheck_t) {
                        ++n_synd;
                        if (live) {
                            const unsigned long long synd = planes_syndrome_only<Z, CPT, HV, CANON>(smem, radr, synd0, xs);
                            if (synd != 0ull && lane == 0) flags[par] = 1 << 20;
                        }
                        lds_barrier();
                        if (__builtin_amdgcn_readfirstlane(flags[par]) == 0) {     // stop test, ldpc.py:332-335
                            my_iters = it;
                            my_conv = 1;
                            break;
                        }
                    }
                    if (live) {
                        const unsigned long long synd = SPA
                            ? (D - xs.X == NS ? planes_check_pass_spa<Z, CPT, B, false, false, HV>(smem, radr, D, a.sign_mode, synd0, xs)
                                              : planes_check_pass_spa<Z, CPT, B, false, true, HV>(smem, radr, D, a.sign_mode, synd0, xs))
                            : planes_check_pass<Z, CPT, B, false, HV, CANON>(smem, radr, D, a.sign_mode, ic, synd0, a.ms_alpha, a.ms_beta, xs);
                        if (synd != 0ull && lane == 0) atomicAdd(&flags[par], (int)__popcll(synd));
                    }
                    ++n_full;
                    TRACE_IT(0);
                    if (HEAVY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lds_barrier();
                    TRACE_IT(1);
                    const int bad2 = __builtin_amdgcn_readfirstlane(flags[par]);   // stop test, ldpc.py:332-335
                    unsat = bad2 & 0xfffff;
                    if (bad2 == 0 || it >= a.max_iters) {
                        my_iters = it;
                        my_conv = (bad2 == 0);
                        break;
                    }
                    if (!ROT) {
                        if (tid == 0) flags[par ^ 1] = 0;
                        if (live) variable_pass(y);
                        TRACE_IT(2);
                        lds_barrier();
                        TRACE_IT(3);
                    }
                }
            }
        }

        // ---- hard decisions: 64 consecutive columns per wave and step; lane t keeps step t
        TRACE(6);
        unsigned long long mine = 0;
        {
            enum { STEPS = (N + NT - 1) / NT };
            float pv[STEPS];
#pragma unroll
            for (int t = 0; t < STEPS; ++t) {
                const int c = (tid & ~63) + t * NT + lane;
                pv[t] = c < N ? lds_f32(smem, (uint32_t)c * CB) : 0.0f;
            }
            int wave_ones = 0;
#pragma unroll
            for (int t = 0; t < STEPS; ++t) {
                const unsigned long long bits = __ballot(pv[t] > 0.0f);
                wave_ones += __popcll(bits);
                mine = lane == t ? bits : mine;
            }
            if (wave_ones && lane == 0) atomicAdd(&flags[2], wave_ones);
#ifndef LDPC_TRACE
            if (a.soft) {
                const int w0 = (int)opaque((uint32_t)(tid & ~63));
#pragma unroll
                for (int t = 0; t < STEPS; ++t) {
                    const int c = w0 + t * NT + lane;
                    if (c < N) reinterpret_cast<float *>(a.soft)[frame * N + c] = CANON ? pv[t] + 0.0f : pv[t];   // (-0) + 0 = +0
                }
            }
#endif
        }
        TRACE(7);
        if (!GEN) {
#pragma unroll
            for (int q = 0; q < CPT; ++q) y[q] = ynext[q] * a.llr_scale + 0.0f;
        } else if (tid == 0) {
            lds_put_u32(smem, G::GEN_OFF, 0u);
            *reinterpret_cast<unsigned long long *>(smem + G::GEN_OFF + 4) = 0ull;
        }
#ifdef LDPC_TRACE
        asm volatile("" : "+v"(y[0]), "+v"(y[7]));
#endif
        TRACE(8);
        lds_barrier();
        TRACE(9);
        out_unc = cur_unc;
        out_sq = cur_sq;
        out_bits = mine;
        out_frame = frame;
        out_iters = my_iters;
        out_conv = my_conv;
        out_ones = __builtin_amdgcn_readfirstlane(flags[2]);
        TRACE(10);
#ifdef LDPC_TRACE
#pragma unroll
        for (int k = 0; k < 10; ++k) tr_acc[k] += tr[k + 1] - tr[k];
        if (tr_last) tr_acc[10] += tr[0] - tr_last;
        tr_last = tr[10];
        tr_acc[11] += 1;
#endif
    }
    flush_outputs();
    if (a.pass_counts && tid == 0) {
        atomicAdd(&a.pass_counts[0], (unsigned long long)n_full);
        atomicAdd(&a.pass_counts[1], (unsigned long long)n_synd);
    }
#ifdef LDPC_TRACE
    // phase sums of this wave over all its frames -> [workgroup][wave][12] in the soft-output buffer
    if (lane == 0 && a.soft)
        for (int k = 0; k < 12; ++k)
            reinterpret_cast<unsigned long long *>(a.soft)[((size_t)blockIdx.x * 16 + (tid >> 6)) * 12 + k] = tr_acc[k];
    if (lane == 0 && a.soft)
        for (int k = 0; k < 4; ++k)
            reinterpret_cast<unsigned long long *>(a.soft)[(size_t)gridDim.x * 16 * 12 + ((size_t)blockIdx.x * 16 + (tid >> 6)) * 4 + k] = it_acc[k];
#endif
}

template <int Z, int CPT, bool SPA, bool GEN, int HV>
__global__ void __launch_bounds__(1024)
qc_planes_kernel(const PlanesTables tb, const DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x / PlanesGeom<Z, CPT>::ZP) == 0)
        planes_role<Z, CPT, 0, SPA, GEN, HV>(smem, tb, a);
    else
        planes_role<Z, CPT, 1, SPA, GEN, HV>(smem, tb, a);
}
