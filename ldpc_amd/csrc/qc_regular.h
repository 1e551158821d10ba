// Fast min-sum kernel for REGULAR quasi-cyclic codes: MB x NB circulant blocks
// in which every non-empty block is a sum of exactly W cyclic shifts
// (NearEarth: MB=2, NB=16, W=2; its circulant-zeroed variants: the same with
// holes; 802.11n: W=1 with holes).  Included by ldpc_hip.hip.
//
// What makes it fast compared with the table-driven kernel:
//   * every LDS address a thread needs -- the NB*W posteriors of its check row
//     and the MB*W row states of each of its CPT columns -- is computed ONCE per
//     launch and kept in registers, so an iteration issues no address
//     arithmetic, no scalar table loads and no divisions;
//   * the loops over a row's edges and a column's checks are fully unrolled
//     straight-line code, so the LDS reads of a pass are issued back to back;
//   * a row's sign bits live in one register that is rotated by one bit per
//     edge: the bit shifted out at the top is the edge's previous sign, the bit
//     shifted in at the bottom its new one.
// Arithmetic and its order are those of the table-driven kernel (ldpc.py:150-163,
// 174-202, 295-324); results are bit-identical.
#pragma once

#ifndef REG_PRECHECK_T
#define REG_PRECHECK_T 32            // syndrome-only pass when the pass before left at most this many rows unsatisfied; 0 = never
#endif

struct RegTables {
    int Z, Zp, nb, n;
    uint32_t row_mask[8];    // [MB] bit j set: block (b, j) is present
    const int32_t *shifts;   // [MB][NB][W] shifts of block (b, j), ascending; unused if absent
};

template <typename T> struct RegState;
// float: {min1, min2, sign bits (left aligned, pre-multiplied by the row's sign product), argmin slot}
template <> struct RegState<float> { enum { BYTES = 16, STRIDE = 16 }; };
// double: {sign bits, arg-min edge} {min1} {min2}, 8 bytes each: the variable pass reads a row's first 8 bytes, then
// only the magnitude it needs (16 bytes per edge instead of 24; the pass is bound by LDS traffic).  The three parts
// of a row sit side by side (stride 24 bytes) or, with REG_F64_PLANES, in three planes of `plane` bytes each.
#ifndef REG_F64_PLANES
#define REG_F64_PLANES 0             // 1: three planes of 8 bytes per row instead of the interleaved 24 bytes -- LDS bank conflicts 19.9 % -> 9.1 % of the LDS cycles, yet 1.3 % slower (the kernel is bound by vector issue); kept for comparison
#endif
template <> struct RegState<double> { enum { BYTES = 24, STRIDE = REG_F64_PLANES ? 8 : 24 }; };

__device__ __forceinline__ uint32_t hi_bits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ uint32_t hi_bits(double v) { return (uint32_t)__double2hiint(v); }

template <typename T> struct RowStateRegs { T min1, min2; uint32_t signs, idx; };

// LDS image of a row state as it is read by the variable pass
template <typename T> struct StateRaw;
template <> struct StateRaw<float> { uint4 v; };
template <> struct StateRaw<double> { uint2 mt; double mag; };

// LDS byte address of a pointer into the workgroup's shared memory, and typed access by such an
// address (lets a base offset be folded into an address computed by hand)
__device__ __forceinline__ uint32_t lds_address(const unsigned char *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)p;
}
typedef uint32_t lds_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t lds_u32x2 __attribute__((ext_vector_type(2)));
typedef double lds_f64x2 __attribute__((ext_vector_type(2)));
template <typename V>          // V: a scalar or one of the native vector types above
__device__ __forceinline__ V lds_load(uint32_t abs_addr)
{
    return *(const __attribute__((address_space(3))) V *)(uintptr_t)abs_addr;
}

__device__ __forceinline__ void lds_store_state(unsigned char *smem, uint32_t addr, uint32_t, const RowStateRegs<float> &s)
{
    *reinterpret_cast<uint4 *>(smem + addr) =
        make_uint4(__float_as_uint(s.min1), __float_as_uint(s.min2), s.signs, s.idx);
}
__device__ __forceinline__ void lds_store_state(unsigned char *smem, uint32_t addr, uint32_t plane, const RowStateRegs<double> &s)
{
    *reinterpret_cast<uint2 *>(smem + addr) = make_uint2(s.signs, s.idx);
    *reinterpret_cast<double *>(smem + addr + plane) = s.min1;
    *reinterpret_cast<double *>(smem + addr + 2 * plane) = s.min2;
}
__device__ __forceinline__ void lds_load_state_abs(uint32_t abs_addr, StateRaw<float> &r)
{
    const lds_u32x4 v = lds_load<lds_u32x4>(abs_addr);
    r.v = make_uint4(v.x, v.y, v.z, v.w);
}
// double, first step: sign bits and argmin slot of the row
__device__ __forceinline__ void lds_load_state_abs(uint32_t abs_addr, StateRaw<double> &r)
{
    const lds_u32x2 mt = lds_load<lds_u32x2>(abs_addr);
    r.mt = make_uint2(mt.x, mt.y);
}
// second step: the magnitude the row sent along the edge with slot number `slot` (min2 to its argmin edge)
__device__ __forceinline__ void lds_load_magnitude(uint32_t, uint32_t, StateRaw<float> &, uint32_t) {}
__device__ __forceinline__ void lds_load_magnitude(uint32_t abs_addr, uint32_t plane, StateRaw<double> &r, uint32_t slot)
{
    r.mag = lds_load<double>(abs_addr + (r.mt.y == slot ? 2 * plane : plane));
}
// message of a row along the edge with slot number `slot` whose sign sits `rank` bits below the top.
// Magnitudes are never negative, so OR-ing the sign bit in is the multiplication by +-1.
__device__ __forceinline__ float rebuild_msg(const StateRaw<float> &r, uint32_t slot, uint32_t rank)
{
    const uint32_t mag = (r.v.w == slot) ? r.v.y : r.v.x;
    return __uint_as_float(mag | ((r.v.z << rank) & 0x80000000u));
}
__device__ __forceinline__ double rebuild_msg(const StateRaw<double> &r, uint32_t, uint32_t rank)
{
    return __hiloint2double(__double2hiint(r.mag) | (int)((r.mt.x << rank) & 0x80000000u), __double2loint(r.mag));
}
// previous message of my own row: non-negative magnitude, sign = bit 31 of signword
__device__ __forceinline__ float with_top_sign(float mag, uint32_t signword)
{
    return __uint_as_float(__float_as_uint(mag) | (signword & 0x80000000u));
}
__device__ __forceinline__ double with_top_sign(double mag, uint32_t signword)
{
    return __hiloint2double(__double2hiint(mag) | (int)(signword & 0x80000000u), __double2loint(mag));
}

// Identity the optimiser cannot see through: keeps the unpacking of a packed address pair
// next to its use instead of being hoisted out of the iteration loop into 2x the registers.
__device__ __forceinline__ uint32_t opaque(uint32_t x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// Half `hi` of a packed address pair plus `rel`, as ONE instruction that the optimiser can neither
// split nor hoist out of the iteration loop (hoisted, the unpacked addresses would take twice the
// registers the packing saves; laundering the pair instead costs a register copy per use).
template <int HI>
__device__ __forceinline__ uint32_t unpack_addr(uint32_t pk, uint32_t rel)
{
    uint32_t r;
    if (HI)
        asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                     : "=v"(r) : "s"(rel), "v"(pk));
    else
        asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
                     : "=v"(r) : "s"(rel), "v"(pk));
    return r;
}

// s_setprio with a value that is constant only after unrolling
__device__ __forceinline__ void set_wave_prio(int p)
{
    switch (p) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// LDS-only barrier: waits for this wave's LDS traffic, not for global loads in flight
// (the next frame's channel values are prefetched across the whole decode of the current one).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Thread layout as in the table-driven kernel: t = ((f*MB + g)*Zp + i).
// LDS: row states | posteriors | flags.  Addresses are kept two to a register (16 bits
// each); float uses absolute byte addresses, double keeps posterior addresses relative to
// the start of the posterior region (the two regions together exceed 64 KB).
template <typename T, int MB, int W, int CPT, int FPW, bool HOLES>
__global__ void __launch_bounds__(1024)
qc_regular_kernel(const RegTables tb, const DecodeArgs a)
{
    // chunk of the check pass / columns per batch of the variable pass: halved for double, whose
    // values take two registers each (measured, float64: two columns per batch +3.4 % on the 2-row
    // NearEarth shape in spite of 100 bytes of spills, -12 % on the 4-row 802.11n shape)
    enum { NB = CPT * MB, NS = NB * W, SB = RegState<T>::BYTES, SS = RegState<T>::STRIDE, NC = CPT * MB * W,
           CH = sizeof(T) == 8 ? 8 : 16, VQ = sizeof(T) == 8 ? (MB == 2 ? 2 : 1) : 2 };
    static_assert(NS <= 32, "sign bits of a row must fit one register");
    static_assert(NS % 2 == 0 && NC % 2 == 0, "addresses are packed in pairs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Z = tb.Z, Zp = tb.Zp, n = tb.n;
    const int slots = FPW * MB * Zp;
    const uint32_t state_off = 0;
    const uint32_t post_off = (uint32_t)slots * SB;
    const uint32_t plane = REG_F64_PLANES ? (uint32_t)slots * SS : 8u;      // float64: bytes from one part of a row state to the next
    const uint32_t flag_off = post_off + (uint32_t)((((size_t)FPW * n * sizeof(T)) + 15) & ~(size_t)15);
    const uint32_t post_rel = sizeof(T) == 8 ? post_off : 0u;   // added when a packed posterior address is used
    const uint32_t lds0 = lds_address(smem);                    // LDS address of the workgroup's dynamic shared memory
    const uint32_t post_abs = post_rel + lds0;
    int *flags = reinterpret_cast<int *>(smem + flag_off);   // [2][FPW] "syndrome non-zero"
    int *ones_cnt = flags + 2 * FPW;                          // [FPW]
    int *claim_word = flags + 3 * FPW;                        // [1] block of frames decoded after the next one

    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid / Zp);
    const int f = grp / MB;
    const int g = grp - f * MB;
    const int i = tid - grp * Zp;
    const bool live = i < Z;
    const int ic = live ? i : 0;     // idle lanes compute harmless in-range addresses
    const int fbase = f * n;
    const int nbytes = (n + 7) >> 3;
    const T INF = t_inf(T(0));
    const uint32_t my_state = state_off + (uint32_t)(grp * Zp + ic) * SS;

    // ---- per-launch set-up: every LDS address this thread will ever use
    uint32_t rmask = 0;                 // present blocks of my block-row
#pragma unroll
    for (int b = 0; b < MB; ++b) rmask = (b == g) ? tb.row_mask[b] : rmask;
    if (!HOLES) rmask = (NB >= 32) ? 0xffffffffu : ((1u << NB) - 1u);
    const int D = __popc(rmask) * W;    // check degree of my row
    uint32_t rpk[NS / 2];               // posterior addresses of my row's edges
#pragma unroll
    for (int t = 0; t < NS / 2; ++t) {
        uint32_t two[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int sl = 2 * t + h, j = sl / W, k = sl % W;
            const int s = tb.shifts[(g * NB + j) * W + k];
            int c = ic + s;
            c = c >= Z ? c - Z : c;
            two[h] = post_off - post_rel + (uint32_t)(fbase + j * Z + c) * (uint32_t)sizeof(T);
        }
        rpk[t] = two[0] | (two[1] << 16);
    }
    uint32_t cpk[NC / 2];               // row-state addresses of my columns' checks, flat ((q*MB + b)*W + k)
    uint32_t lo_first = 0;              // bit q*MB+b: the row of block (b, j_q)'s lower shift has the lower index
    uint32_t cmask[MB];                 // present blocks of every block-row (uniform)
#pragma unroll
    for (int b = 0; b < MB; ++b) cmask[b] = HOLES ? tb.row_mask[b] : rmask;
#pragma unroll
    for (int t = 0; t < NC / 2; ++t) {
        uint32_t two[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int fl = 2 * t + h, q = fl / (MB * W), b = (fl / W) % MB, k = fl % W;
            const int j = g + q * MB;
            const int s = tb.shifts[(b * NB + j) * W + k];
            int r = ic - s;
            r = r < 0 ? r + Z : r;
            two[h] = state_off + (uint32_t)((f * MB + b) * Zp + r) * SS;
            if (W == 2 && k == 0) {
                const int s1 = tb.shifts[(b * NB + j) * W + (W - 1)];
                if (ic >= s && ic < s1) lo_first |= 1u << (q * MB + b);
            }
        }
        cpk[t] = two[0] | (two[1] << 16);
    }
    uint32_t ypost[1];                  // byte address of my first column's posterior; the others follow at MB*Z elements
    ypost[0] = post_off + (uint32_t)(fbase + g * Z + ic) * (uint32_t)sizeof(T);
    const uint32_t col_stride = (uint32_t)(MB * Z) * (uint32_t)sizeof(T);

    RowStateRegs<T> o;                  // messages my row sent last (compressed)
    o.min1 = T(0); o.min2 = T(0); o.signs = 0; o.idx = 0;
    unsigned char *chain = a.chain;
    if (chain && live) {
        const unsigned char *src = chain + (size_t)(grp * Zp + i) * 32;
        o.min1 = *reinterpret_cast<const T *>(src);
        o.min2 = *reinterpret_cast<const T *>(src + 8);
        o.signs = *reinterpret_cast<const uint32_t *>(src + 16);
        o.idx = *reinterpret_cast<const uint32_t *>(src + 20);
    }

    const int64_t n_fblocks = (a.n_frames + FPW - 1) / FPW;
    // float32 channel values are prefetched one frame ahead (8 registers); float64 inputs are
    // fetched when the frame starts: a second set of doubles would not fit the register budget
    float ypf[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) ypf[q] = 0.0f;
    {
        const int64_t frame0 = (int64_t)blockIdx.x * FPW + f;
        if (!a.llr_f64 && live && blockIdx.x < n_fblocks && frame0 < a.n_frames) {
            const float *src = reinterpret_cast<const float *>(a.llr) + frame0 * n + g * Z + i;
#pragma unroll
            for (int q = 0; q < CPT; ++q) ypf[q] = src[(int64_t)q * (MB * Z)];
        }
    }
    // With a.work set, blocks of frames are claimed from a device counter two blocks ahead
    // (frames differ in iteration count; a fixed stride leaves CUs idle at the end of a launch).
    int64_t fb_next = (int64_t)blockIdx.x + gridDim.x;
    for (int64_t fb = blockIdx.x; fb < n_fblocks;) {
        const int64_t frame = fb * FPW + f;
        const bool valid = frame < a.n_frames;
        T y[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) y[q] = T(ypf[q]);
        if (a.llr_f64 && live && valid) {
            const double *src = reinterpret_cast<const double *>(a.llr) + frame * n + g * Z + i;
#pragma unroll
            for (int q = 0; q < CPT; ++q) y[q] = T(src[(int64_t)q * (MB * Z)]);
        }
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            // "+ 0": a -0.0 input becomes +0.0, which no comparison or sum below can tell
            // apart, and which keeps "sign bit set" equivalent to "< 0" for every message
            y[q] = y[q] * T(a.llr_scale) + T(0);
            if (live && valid)
                *reinterpret_cast<T *>(smem + ypost[0] + (uint32_t)q * col_stride) = y[q];
        }
        if (tid < 3 * FPW) flags[tid] = 0;
        if (!chain) { o.min1 = T(0); o.min2 = T(0); o.signs = 0; o.idx = 0; }
        uint32_t act_mask = 0;
#pragma unroll
        for (int ff = 0; ff < FPW; ++ff)
            if (fb * FPW + ff < a.n_frames) act_mask |= 1u << ff;
        int my_iters = 0, my_conv = 0;
        lds_barrier();
        // Next frame's channel values: requested here, not before the barrier.  A CU needs
        // thousands of cycles to queue a frame's worth of loads; after the barrier that
        // queueing overlaps the other waves' first check pass instead of idling the CU.
        int claimed = 0;
        if (a.work && tid == 0) claimed = atomicAdd(a.work, 1);
        {
            const int64_t fbn = fb_next;
            const int64_t framen = fbn * FPW + f;
            if (!a.llr_f64 && live && fbn < n_fblocks && framen < a.n_frames) {
                const float *src = reinterpret_cast<const float *>(a.llr) + framen * n + g * Z + i;
#pragma unroll
                for (int q = 0; q < CPT; ++q) ypf[q] = src[(int64_t)q * (MB * Z)];
            }
        }

        int unsat = 0x7fffffff;                   // rows the check pass before left unsatisfied (one frame per workgroup)
        for (int it = 0;; ++it) {
            const bool active = (act_mask >> f) & 1u;
            const int par = it & 1;
            // ================= syndrome only, ahead of a check pass that is likely to be a frame's last
            // (all of whose other work would be thrown away): same stop decision, same iteration count
            if (FPW == 1 && unsat <= (a.precheck >= 0 ? a.precheck : REG_PRECHECK_T)) {     // unsat > 0 here, so 0 means never
                if (active && live) {
                    unsigned long long synd = 0;
#pragma unroll
                    for (int c0 = 0; c0 < NS; c0 += 8) {
                        T p[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int s = c0 + u;
                            if (s < NS && (!HOLES || ((rmask >> (s / W)) & 1u))) {
                                const uint32_t ad = (s & 1) ? unpack_addr<1>(rpk[s / 2], post_abs) : unpack_addr<0>(rpk[s / 2], post_abs);
                                p[u] = lds_load<T>(ad);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int s = c0 + u;
                            if (s < NS && (!HOLES || ((rmask >> (s / W)) & 1u))) synd ^= __ballot(p[u] > T(0));   // slicer, ldpc.py:47
                        }
                    }
                    if (synd != 0ull && (tid & 63) == 0) flags[par] = 1 << 20;
                }
                lds_barrier();
                if (flags[par] == 0) {
                    my_iters = it;
                    my_conv = 1;
                    break;
                }
            }
            // ================= check pass (+ syndrome of the current posterior)
            RowStateRegs<T> nw;
            nw.min1 = INF; nw.min2 = INF; nw.idx = 0; nw.signs = 0;
            if (active && live) {
                unsigned long long synd = 0;
                uint32_t R = o.signs;                     // top bit = previous sign of the next edge
#pragma unroll
                for (int c0 = 0; c0 < NS; c0 += CH) {
                    T p[CH];
                    // waves that are behind get issue priority, so that a SIMD's waves finish together
                    set_wave_prio(c0 * 4 / NS >= 3 ? 0 : 3 - c0 * 4 / NS);
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        const int s = c0 + u;
                        if (s < NS && (!HOLES || ((rmask >> (s / W)) & 1u))) {
                            const uint32_t ad = (s & 1) ? unpack_addr<1>(rpk[s / 2], post_abs) : unpack_addr<0>(rpk[s / 2], post_abs);
                            p[u] = lds_load<T>(ad);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        const int s = c0 + u;
                        if (s < NS && (!HOLES || ((rmask >> (s / W)) & 1u))) {
                            synd ^= __ballot(p[u] > T(0));                       // slicer, ldpc.py:47
                            const T mag_old = (o.idx == (uint32_t)s) ? o.min2 : o.min1;
                            const T v = p[u] - with_top_sign(mag_old, R);        // receive, ldpc.py:161
                            R = __builtin_amdgcn_alignbit(R, hi_bits(v), 31);    // (R << 1) | sign(v), ldpc.py:139
                            const T av = t_abs(v);                               // ldpc.py:178
                            const bool lt = av < nw.min1;
                            nw.min2 = t_mid(nw.min1, nw.min2, av);
                            nw.idx = lt ? (uint32_t)s : nw.idx;
                            nw.min1 = t_min(nw.min1, av);
                        }
                    }
                }
                // R: D new sign bits, first edge highest.  Left-align; fold in the product of all
                // signs so that a bit is directly the sign of the outgoing message (ldpc.py:190)
                if (D < 32) R = D > 0 ? (R << (32 - D)) : 0u;
                uint32_t S = __popc(R) & 1u;
                if (a.sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)D & 1u;
                const uint32_t full = D > 0 ? (0xffffffffu << (32 - D)) : 0u;
                nw.signs = R ^ (S ? full : 0u);
                // normalised / offset min-sum; unconditional on purpose (a branch here costs the
                // register allocator hundreds of bytes of spills); exact identity for alpha = 1, beta = 0
                nw.min1 = ms_correct(nw.min1, a.ms_alpha, a.ms_beta);
                nw.min2 = ms_correct(nw.min2, a.ms_alpha, a.ms_beta);
                set_wave_prio(0);
                lds_store_state(smem, my_state, plane, nw);
                if (FPW == 1) {             // count of failing rows (bit 20: raised by the syndrome-only pass)
                    if (synd != 0ull && (tid & 63) == 0) atomicAdd(&flags[par], (int)__popcll(synd));
                } else if (synd != 0ull) {
                    flags[par * FPW + f] = 1;
                }
            }
            lds_barrier();
            // ================= stop test (ldpc.py:332-335), same outcome in every thread
            uint32_t next_mask = 0;
#pragma unroll
            for (int ff = 0; ff < FPW; ++ff) {
                if ((act_mask >> ff) & 1u) {
                    const int bad = flags[par * FPW + ff];
                    if (FPW == 1) unsat = bad & 0xfffff;
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
            // a row's new messages replace its previous ones only if its frame goes on
            if (((act_mask >> f) & 1u) && live) o = nw;
            if (tid < FPW) flags[(par ^ 1) * FPW + tid] = 0;
            // ================= variable pass (ldpc.py:295-311, 321-323)
            if (((act_mask >> f) & 1u) && live) {
#pragma unroll
                for (int q0 = 0; q0 < CPT; q0 += VQ) {
                    set_wave_prio(3 - q0 * 3 / (CPT > 2 ? CPT - 2 : 1));
                    StateRaw<T> st[VQ][MB][W];
                    uint32_t sad[VQ][MB][W];
#pragma unroll
                    for (int dq = 0; dq < VQ; ++dq)
#pragma unroll
                        for (int b = 0; b < MB; ++b)
#pragma unroll
                            for (int k = 0; k < W; ++k) {
                                const int q = q0 + dq, fl = (q * MB + b) * W + k;
                                const int j = g + q * MB;
                                if (q < CPT && (!HOLES || ((cmask[b] >> j) & 1u))) {
                                    const uint32_t ad = (fl & 1) ? unpack_addr<1>(cpk[fl / 2], lds0) : unpack_addr<0>(cpk[fl / 2], lds0);
                                    sad[dq][b][k] = ad;
                                    lds_load_state_abs(ad, st[dq][b][k]);
                                }
                            }
                    if (sizeof(T) == 8) {
#pragma unroll
                        for (int dq = 0; dq < VQ; ++dq)
#pragma unroll
                            for (int b = 0; b < MB; ++b)
#pragma unroll
                                for (int k = 0; k < W; ++k) {
                                    const int q = q0 + dq, j = g + q * MB;
                                    if (q < CPT && (!HOLES || ((cmask[b] >> j) & 1u)))
                                        lds_load_magnitude(sad[dq][b][k], plane, st[dq][b][k], (uint32_t)(j * W + k));
                                }
                    }
#pragma unroll
                    for (int dq = 0; dq < VQ; ++dq) {
                        const int q = q0 + dq;
                        if (q < CPT) {
                            const int j = g + q * MB;
                            T acc = T(0);
                            bool first_group = true;
#pragma unroll
                            for (int b = 0; b < MB; ++b) {
                                if (!HOLES || ((cmask[b] >> j) & 1u)) {
                                    // rank of the block's first edge among the present edges of block-row b
                                    const uint32_t rank = HOLES ? (uint32_t)(__popc(cmask[b] & ((1u << j) - 1u)) * W)
                                                                : (uint32_t)(j * W);
                                    if (W == 1) {
                                        acc = acc + rebuild_msg(st[dq][b][0], (uint32_t)(j * W), rank);
                                    } else {
                                        const T v0 = rebuild_msg(st[dq][b][0], (uint32_t)(j * W), rank);
                                        const T v1 = rebuild_msg(st[dq][b][W - 1], (uint32_t)(j * W + 1), rank + 1);
                                        if (first_group && !HOLES) {
                                            // first two terms of the column sum: 0 + x + y == x + y == y + x
                                            acc = v0 + v1;
                                        } else {
                                            const bool lf = (lo_first >> (q * MB + b)) & 1u;
                                            acc = acc + (lf ? v0 : v1);     // lower check index first
                                            acc = acc + (lf ? v1 : v0);
                                        }
                                    }
                                    first_group = false;
                                }
                            }
                            *reinterpret_cast<T *>(smem + ypost[0] + (uint32_t)q * col_stride) = acc + y[q];   // channel last, ldpc.py:322
                        }
                    }
                }
            }
            lds_barrier();
        }

        // ================= outputs
        const T *post = reinterpret_cast<const T *>(smem + post_off);
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
                const int j = g + q * MB;
                so[j * Z + i] = post[fbase + j * Z + i];
            }
        }
        if (a.work && tid == 0) *claim_word = claimed;
        lds_barrier();
        if (valid && g == 0 && i == 0) {
            if (a.iters) a.iters[frame] = my_iters;
            if (a.conv) a.conv[frame] = (uint8_t)my_conv;
            if (a.ones) a.ones[frame] = ones_cnt[f];
        }
        const int64_t after = a.work ? 2 * (int64_t)gridDim.x + __builtin_amdgcn_readfirstlane(*claim_word)
                                     : fb_next + gridDim.x;
        lds_barrier();
        fb = fb_next;
        fb_next = after;
    }
    if (chain && live) {
        unsigned char *dst = chain + (size_t)(grp * Zp + i) * 32;
        *reinterpret_cast<T *>(dst) = o.min1;
        *reinterpret_cast<T *>(dst + 8) = o.min2;
        *reinterpret_cast<uint32_t *>(dst + 16) = o.signs;
        *reinterpret_cast<uint32_t *>(dst + 20) = o.idx;
    }
}
