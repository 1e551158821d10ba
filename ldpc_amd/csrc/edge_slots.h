// float32 min-sum (and sum-product) for SHORT codes given as an edge list (802.11n n = 1944, toy codes ...):
// several frames per workgroup, every check-to-variable message explicit in LDS.
// Included by ldpc_hip.hip.
//
// Works for any parity-check matrix with column degree <= DV (4, 8 or 12), row degree <= DC and at
// most 1024 rows per workgroup (frames per workgroup F = as many as threads and LDS allow).
//   thread t  -> check row  t mod m  of frame  t / m  of the workgroup's block of F frames
//               (its edge addresses live in registers for the whole launch), and
//            -> columns t, t + NT, t + 2 NT, ... of the F*n columns of the block.
//   LDS       -> one record of 1 + DV words per column { posterior, message slot 0..DV-1 }, slot k holding
//               the message of the column's k-th check in ascending check index, so the variable
//               update is the plain ordered sum of ldpc.py:298-303 (unused slots stay +0);
//               per-frame stop flags and counters; one dummy record.  The channel values of a
//               thread's columns stay in its registers.
// Rows shorter than DC point their spare edges at the dummy record (posterior +inf: never a
// minimum, sign +, hard decision 1 -- the parity of the spare count is folded into the syndrome).
// Arithmetic and order are those of the other min-sum kernels (ldpc.py:150-163, 174-202,
// 295-324, 326-337): results are bit-identical.
#pragma once
// Rows left unsatisfied up to which a frame takes the syndrome-only pass (0 = never).  Measured on
// 802.11n 1944 r5/6 at 4 dB: sum-product 30.8 -> 36.6 Gbit/s; min-sum loses 3-6 % (its check pass is
// cheap, and counting the failing rows alone costs 3 %), so it stays off there.
#ifndef SLOT_PRECHECK_T_SPA
#define SLOT_PRECHECK_T_SPA 32
#endif
#ifndef SLOT_PRECHECK_T_MINSUM
#define SLOT_PRECHECK_T_MINSUM 0
#endif

struct SlotTables {
    int m, n, E;
    int F;                     // frames per workgroup
    const int32_t *row_ptr;    // [m+1]
    const int32_t *col_idx;    // [E] ascending inside a row
    const int32_t *edge_slot;  // [E] rank of the edge's row among the rows of its column
    const int32_t *col_ptr;    // [n+1] (column degrees)
    const int32_t *col_rec;    // [n] first word of a column's record, edge_packed.h
    int rec_words;             // words of a frame's records, edge_packed.h
};

struct SlotGeom {
    enum { YMAX = 8 };
    uint32_t CB;               // bytes per column record: 4 * (1 + DV)
    uint32_t y_off, flag_off, cont_off, claim_off, ones_off, wcnt_off, want_off, pre_off, plan_off, dummy_off, bytes;
    __host__ __device__ SlotGeom(int n, int F, int DV)
    {
        CB = 4u * (1u + (uint32_t)DV);
        const uint32_t cols = (uint32_t)n * (uint32_t)F;
        y_off = cols * CB;                    // (end of the records)
        flag_off = y_off;                     // [2][F] "syndrome of frame non-zero"
        cont_off = flag_off + 2u * F * 4;     // [2]    "some frame goes on"
        claim_off = cont_off + 8;             // [1]    block of frames this workgroup decodes after the next one
        ones_off = claim_off + 8;             // [F]
        wcnt_off = ones_off + (uint32_t)F * 4;   // [2][F] rows a frame's check pass left unsatisfied (estimate)
        want_off = wcnt_off + 2u * F * 4;        // [F]    frame takes the syndrome-only pass next
        pre_off = want_off + (uint32_t)F * 4;    // [F]    ... and that pass found a failing row
        plan_off = pre_off + (uint32_t)F * 4;    // [2] frames that want it, [2] live frames that do not, [2] one of the wanting goes on
        dummy_off = plan_off + 24;
        bytes = (dummy_off + CB + 15) / 16 * 16;
    }
};

// Adds the number of failing rows of this wave (lane mask `synd`, active lanes only) to their frames'
// counters: one LDS atomic per frame the wave touches.  The counts only steer the syndrome-only pass.
__device__ __forceinline__ void slot_count_failures(unsigned long long synd, int f, int lane, int *counters)
{
    unsigned long long rem = synd & __ballot(true);
    while (rem != 0ull) {                                   // uniform: at most once per frame in the wave
        const int fr = __builtin_amdgcn_readlane(f, __builtin_ctzll(rem));
        const unsigned long long of_frame = __ballot(f == fr);
        const int cnt = __popcll(rem & of_frame);
        if (lane == __builtin_ctzll(rem)) atomicAdd(&counters[fr], cnt);
        rem &= ~of_frame;
    }
}

// SPA = true: sum-product check rule (stated with csr_spa_kernel; phi from planes_phi) in
// place of min-sum; everything else is shared.
template <int DC, int DV, bool SPA>
__global__ void __launch_bounds__(1024)
slot_minsum_kernel(const SlotTables tb, const DecodeArgs a)
{
    enum { CB = 4 * (1 + DV), YMAX = SlotGeom::YMAX, CH = 4, NCH = DC / CH,
           SLOT_PRECHECK_T = SPA ? SLOT_PRECHECK_T_SPA : SLOT_PRECHECK_T_MINSUM };
    static_assert(DC % CH == 0 && DC <= 32 && DV <= 15, "degree bounds");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = tb.m, n = tb.n, F = tb.F;
    const SlotGeom G(n, F, DV);
    const int NT = blockDim.x;
    const int tid = threadIdx.x;
    const int cols = F * n;
    const int nbytes = (n + 7) >> 3;
    int *flags = reinterpret_cast<int *>(smem + G.flag_off);
    int *cont = reinterpret_cast<int *>(smem + G.cont_off);
    int *ones_cnt = reinterpret_cast<int *>(smem + G.ones_off);
    int *wcnt = reinterpret_cast<int *>(smem + G.wcnt_off);
    int *want = reinterpret_cast<int *>(smem + G.want_off);
    int *preflag = reinterpret_cast<int *>(smem + G.pre_off);
    int *n_want = reinterpret_cast<int *>(smem + G.plan_off), *n_not = n_want + 2, *pre_cont = n_want + 4;
    const float INF = __uint_as_float(0x7f800000u);

    // ---- per-launch set-up: my row, the LDS addresses of its edges
    const bool row_valid = tid < F * m;
    const int f = row_valid ? tid / m : 0;
    const int r = row_valid ? tid - f * m : 0;
    const int e0 = row_valid ? tb.row_ptr[r] : 0;
    const int deg = row_valid ? tb.row_ptr[r + 1] - e0 : 0;
    uint32_t radr[DC], madr[DC];
#pragma unroll
    for (int e = 0; e < DC; ++e) {
        uint32_t pa = G.dummy_off, ma = G.dummy_off + 4;
        if (e < deg) {
            pa = (uint32_t)(f * n + tb.col_idx[e0 + e]) * CB;
            ma = pa + 4u + 4u * (uint32_t)tb.edge_slot[e0 + e];
        }
        radr[e] = opaque(pa);
        madr[e] = opaque(ma);
    }
    const unsigned long long spare_odd = __ballot(((DC - deg) & 1) != 0);
    // frame of each of my columns, one byte per column
    unsigned long long colf = 0;
#pragma unroll
    for (int k = 0; k < YMAX; ++k) {
        const int cg = k * NT + tid;
        if (cg < cols) colf |= (unsigned long long)(cg / n) << (8 * k);
    }
    // degree of each of my columns, four bits per column (DV > 4: the variable pass reads only the slots in use)
    uint32_t cdeg = 0;
    if (DV > 4) {
#pragma unroll
        for (int k = 0; k < YMAX; ++k) {
            const int cg = k * NT + tid;
            if (cg < cols) {
                const int c = cg % n;
                cdeg |= (uint32_t)(tb.col_ptr[c + 1] - tb.col_ptr[c]) << (4 * k);
            }
        }
    }
    if (tid < 1 + DV) lds_put_u32(smem, G.dummy_off + 4u * tid, tid == 0 ? 0x7f800000u : 0u);

    const int64_t n_fblocks = (a.n_frames + F - 1) / F;
    // channel values of my columns, prefetched one block of frames ahead (float32 inputs)
    float ypf[YMAX];
#pragma unroll
    for (int k = 0; k < YMAX; ++k) ypf[k] = 0.0f;
    auto fetch = [&](int64_t fb, float (&dst)[YMAX]) {
        const int64_t base = fb * F * (int64_t)n;
        const int64_t total = a.n_frames * (int64_t)n;
        const int t = (int)opaque((uint32_t)tid);             // addresses recomputed here, not kept across the decode
#pragma unroll
        for (int k = 0; k < YMAX; ++k) {
            const int cg = k * NT + t;
            if (cg < cols && base + cg < total)
                dst[k] = a.llr_f64 ? (float)reinterpret_cast<const double *>(a.llr)[base + cg]
                                   : reinterpret_cast<const float *>(a.llr)[base + cg];
        }
    };
    if ((int64_t)blockIdx.x < n_fblocks && !a.llr_f64) fetch(blockIdx.x, ypf);

    // Blocks of frames are claimed from a device counter when a.work is set (frames differ in
    // iteration count; a fixed stride leaves CUs idle at the end of a launch).  The claim is made
    // two blocks ahead, so its latency and the prefetch of the next block never wait for it.
    int *claim_word = reinterpret_cast<int *>(smem + G.claim_off);
    int64_t fb_next = (int64_t)blockIdx.x + gridDim.x;
    for (int64_t fb = blockIdx.x; fb < n_fblocks;) {
        if (a.llr_f64) fetch(fb, ypf);
        const int64_t my_frame = fb * F + f;
        bool my_act = row_valid && my_frame < a.n_frames;
        // ---- load the block: posterior = channel value, messages = +0 (fresh decoder, ldpc.py:121)
        float ycur[YMAX];
#pragma unroll
        for (int k = 0; k < YMAX; ++k) {
            const int cg = k * NT + tid;
            ycur[k] = ypf[k] * a.llr_scale + 0.0f;                 // "+ 0": -0.0 -> +0.0
            if (cg < cols) {
                const uint32_t rec = (uint32_t)cg * CB;
                lds_put_f32(smem, rec, ycur[k]);
#pragma unroll
                for (int sl = 1; sl <= DV; ++sl) lds_put_f32(smem, rec + 4u * sl, 0.0f);
            }
        }
        if (tid < 2 * F) { flags[tid] = 0; wcnt[tid] = 0; }
        if (tid < 2) cont[tid] = 0;
        if (tid < F) { ones_cnt[tid] = 0; want[tid] = 0; preflag[tid] = 0; }
        if (tid < 6) n_want[tid] = 0;
        int my_iters = 0, my_conv = 0;
        lds_barrier();
        int claimed = 0;
        if (a.work && tid == 0) claimed = atomicAdd(a.work, 1);
        if (fb_next < n_fblocks && !a.llr_f64) fetch(fb_next, ypf);

        for (int it = 0;; ++it) {
            const int par = it & 1;
            // ================= syndrome only, for frames likely to stop now.  A frame's last check pass
            // finds every parity satisfied and all its other work is thrown away; a frame whose pass
            // before left at most SLOT_PRECHECK_T rows unsatisfied (the plan was laid after that pass's
            // stop test) first takes this much cheaper pass.  Same stop decision, same iteration count.
            if (SLOT_PRECHECK_T > 0 && it > 0) {
                if (tid == 0) { n_want[par ^ 1] = 0; n_not[par ^ 1] = 0; pre_cont[par ^ 1] = 0; }
                if (__builtin_amdgcn_readfirstlane(n_want[par]) != 0) {
                    const bool my_pre = my_act && want[f] != 0;
                    if (my_pre) {
                        unsigned long long synd = spare_odd;
#pragma unroll
                        for (int s0 = 0; s0 < DC; s0 += CH) {
                            float pp[CH];
#pragma unroll
                            for (int u = 0; u < CH; ++u) pp[u] = lds_f32(smem, radr[s0 + u]);
#pragma unroll
                            for (int u = 0; u < CH; ++u) synd ^= __ballot(pp[u] > 0.0f);         // slicer, ldpc.py:47
                        }
                        if ((synd >> (tid & 63)) & 1ull) {
                            preflag[f] = 1;
                            pre_cont[par] = 1;
                        }
                    }
                    lds_barrier();
                    if (my_pre && preflag[f] == 0) {
                        my_iters = it;
                        my_conv = 1;
                        my_act = false;
                    }
                    // does any frame of the block go on?  (one that did not want the pass, or one it failed)
                    if (__builtin_amdgcn_readfirstlane(n_not[par]) == 0 && __builtin_amdgcn_readfirstlane(pre_cont[par]) == 0) break;
                }
            }
            // ================= check pass (+ syndrome of the current posterior)
            if (my_act && !SPA) {
                float min1 = INF, min2 = INF;
                uint32_t idx = 0, R = 0;
                unsigned long long synd = spare_odd;
                float p[2][CH], c[2][CH];
#pragma unroll
                for (int k = -1; k < NCH; ++k) {
                    if (k + 1 < NCH) {
                        const int nb = (k + 1) & 1;
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            p[nb][u] = lds_f32(smem, radr[(k + 1) * CH + u]);
                            c[nb][u] = lds_f32(smem, madr[(k + 1) * CH + u]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (k < 0) continue;
                    {
                        enum { PQ = NCH >= 4 ? NCH / 4 : 1 };            // chunks per priority step
                        if (NCH >= 4 && k % PQ == 0) set_wave_prio(3 - k / PQ > 0 ? 3 - k / PQ : 0);
                    }
                    const int cb = k & 1;
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        const int s = k * CH + u;
                        synd ^= __ballot(p[cb][u] > 0.0f);                          // slicer, ldpc.py:47
                        const float v = p[cb][u] - c[cb][u];                        // receive, ldpc.py:161
                        R = __builtin_amdgcn_alignbit(R, __float_as_uint(v), 31);   // (R << 1) | sign(v), ldpc.py:139
                        const float av = fabsf(v);                                  // ldpc.py:178
                        const bool lt = av < min1;
                        idx = lt ? (uint32_t)s : idx;
                        // bare instructions: behind the register pin below the compiler would
                        // re-canonicalise min1 (v_max x, x) in front of its own med3 / min
                        float n2, n1;
                        asm("v_med3_f32 %0, %1, %2, |%3|" : "=v"(n2) : "v"(min1), "v"(min2), "v"(v));
                        asm("v_min_f32 %0, %1, |%2|" : "=v"(n1) : "v"(min1), "v"(v));
                        min2 = n2;
                        min1 = n1;
                    }
                    asm volatile("" : "+v"(min1), "+v"(min2), "+v"(idx), "+v"(R), "+s"(synd));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (DC < 32) R <<= (32 - DC);
                uint32_t S = __popc(R) & 1u;                                        // spare edges are positive
                if (a.sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)deg & 1u;
                const uint32_t T = R ^ (S ? (0xffffffffu << (32 - DC)) : 0u);
                if (a.ms_alpha != 1.0f || a.ms_beta != 0.0f) {
                    min1 = ms_correct(min1, a.ms_alpha, a.ms_beta);
                    min2 = ms_correct(min2, a.ms_alpha, a.ms_beta);
                }
                const uint32_t m1 = __float_as_uint(min1), m2 = __float_as_uint(min2);
                set_wave_prio(0);
#pragma unroll
                for (int s = 0; s < DC; ++s) {
                    const uint32_t mag = idx == (uint32_t)s ? m2 : m1;              // ldpc.py:190-191
                    lds_put_u32(smem, madr[s], mag | ((T << s) & 0x80000000u));
                    if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
                if ((synd >> (tid & 63)) & 1ull) {
                    flags[par * F + f] = 1;
                    cont[par] = 1;
                }
                if (SLOT_PRECHECK_T > 0) slot_count_failures(synd, f, tid & 63, wcnt + par * F);
            }
            if (my_act && SPA) {
                float F1 = 0.0f, rest = 0.0f;      // every real F is > 0: the first edge displaces the initial 0
                uint32_t idx = 0, R = 0;
                unsigned long long synd = spare_odd;
#pragma unroll
                for (int s = 0; s < DC; ++s) {
                    const float pv = lds_f32(smem, radr[s]);
                    const float v = pv - lds_f32(smem, madr[s]);
                    synd ^= __ballot(pv > 0.0f);
                    R = __builtin_amdgcn_alignbit(R, __float_as_uint(v), 31);
                    const float av = fabsf(v);
                    float Fe = planes_phi(__builtin_amdgcn_fmed3f(av, 1e-12f, 80.0f));
                    Fe = av == INF ? 0.0f : Fe;                                     // spare edge: no term
                    if (s < deg) lds_put_f32(smem, madr[s], Fe);                    // spare edges share one slot: leave it alone
                    const bool gt = Fe > F1;
                    rest = rest + (gt ? F1 : Fe);
                    idx = gt ? (uint32_t)s : idx;
                    F1 = gt ? Fe : F1;
                    if ((s & 3) == 3) {
                        asm volatile("" : "+v"(F1), "+v"(rest), "+v"(idx), "+v"(R));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (DC < 32) R <<= (32 - DC);
                uint32_t S = __popc(R) & 1u;
                if (a.sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)deg & 1u;
                const uint32_t T = R ^ (S ? (0xffffffffu << (32 - DC)) : 0u);
#pragma unroll
                for (int s = 0; s < DC; ++s) {
                    if (s < deg) {
                        const float Fe = lds_f32(smem, madr[s]);
                        const float x = idx == (uint32_t)s ? rest : (rest - Fe) + F1;
                        const float mag = planes_phi(__builtin_amdgcn_fmed3f(x, 1e-12f, 80.0f));
                        lds_put_u32(smem, madr[s], __float_as_uint(mag) | ((T << s) & 0x80000000u));
                    }
                }
                if ((synd >> (tid & 63)) & 1ull) {
                    flags[par * F + f] = 1;
                    cont[par] = 1;
                }
                if (SLOT_PRECHECK_T > 0) slot_count_failures(synd, f, tid & 63, wcnt + par * F);
            }
            lds_barrier();
            // ================= stop test (ldpc.py:332-335): my frame, and whether any frame goes on
            const int my_bad = flags[par * F + f];
            const int any_bad = __builtin_amdgcn_readfirstlane(cont[par]);
            if (my_act && my_bad == 0) {
                my_iters = it;
                my_conv = 1;
                my_act = false;
            }
            if (any_bad == 0 || it >= a.max_iters) {
                if (my_act) { my_iters = it; my_conv = 0; }
                break;
            }
            if (tid < F) { flags[(par ^ 1) * F + tid] = 0; wcnt[(par ^ 1) * F + tid] = 0; }
            if (tid == 0) cont[par ^ 1] = 0;
            // plan of the next iteration's syndrome-only pass, laid by the thread of each live frame's row 0
            if (SLOT_PRECHECK_T > 0 && my_act && r == 0) {
                const int w = wcnt[par * F + f] <= (a.precheck >= 0 ? a.precheck : SLOT_PRECHECK_T);
                want[f] = w;
                preflag[f] = 0;
                atomicAdd(w ? &n_want[par ^ 1] : &n_not[par ^ 1], 1);
            }
            // ================= variable pass of the frames that go on (ldpc.py:295-311, 321-323)
            set_wave_prio(3);
#pragma unroll
            for (int k = 0; k < YMAX; ++k) {
                const int cg = k * NT + tid;
                if (cg < cols) {
                    const int ff = (int)((colf >> (8 * k)) & 0xffull);
                    if (flags[par * F + ff] != 0) {
                        const uint32_t rec = (uint32_t)cg * CB;
                        float acc = lds_f32(smem, rec + 4) + lds_f32(smem, rec + 8);   // 0 + x + y == x + y
                        acc = acc + lds_f32(smem, rec + 12);
                        acc = acc + lds_f32(smem, rec + 16);
                        if (DV > 4) {
                            const int dg = (int)((cdeg >> (4 * k)) & 15u);
#pragma unroll
                            for (int sl = 4; sl < DV; ++sl)
                                if (sl < dg) acc = acc + lds_f32(smem, rec + 4u + 4u * sl);
                        }
                        lds_put_f32(smem, rec, acc + ycur[k]);                        // channel last, ldpc.py:322
                    }
                }
            }
            set_wave_prio(0);
            lds_barrier();
        }

        // ================= outputs
        if ((n & 7) == 0) {
            // frames are whole bytes long, so the packed decisions of the block are one contiguous
            // bit string: 64 columns per wave and step, lane j < 8 stores byte j of the wave's word
            const int64_t left = a.n_frames - fb * F;
            const int vbits = (int)(left < F ? left : F) * n;
            const int lane = tid & 63;
            const int w0 = (int)opaque((uint32_t)(tid & ~63));
#pragma unroll
            for (int k = 0; k < YMAX; ++k) {
                const int c0 = k * NT + w0;
                if (c0 < vbits) {                                   // uniform per wave
                    const int cg = c0 + lane;
                    const bool one = cg < vbits && lds_f32(smem, (uint32_t)cg * CB) > 0.0f;
                    const unsigned long long bits = __ballot(one);
                    if (a.hard && lane < 8 && c0 + 8 * lane < vbits)
                        a.hard[fb * F * (int64_t)nbytes + (c0 >> 3) + lane] = (uint8_t)(bits >> (8 * lane));
                    if (bits) {
                        const int ff = (int)((colf >> (8 * k)) & 0xffull);
                        const int f_lo = __builtin_amdgcn_readfirstlane(ff);
                        const int f_hi = (c0 + 63 < vbits ? c0 + 63 : vbits - 1) / n;
                        for (int fr = f_lo; fr <= f_hi; ++fr) {
                            const int cnt = __popcll(bits & __ballot(ff == fr));
                            if (cnt && lane == 0) atomicAdd(&ones_cnt[fr], cnt);
                        }
                    }
                }
            }
        } else {
            for (int x = tid; x < F * nbytes; x += NT) {
                const int ff = x / nbytes;
                const int qb = x - ff * nbytes;
                const int64_t fr = fb * F + ff;
                if (fr < a.n_frames) {
                    uint32_t byte = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (qb * 8 + k < n) byte |= (uint32_t)(lds_f32(smem, (uint32_t)(ff * n + qb * 8 + k) * CB) > 0.0f) << k;
                    if (a.hard) a.hard[fr * nbytes + qb] = (uint8_t)byte;
                    if (byte) atomicAdd(&ones_cnt[ff], (int)__popc(byte));
                }
            }
        }
        if (a.soft) {
            const int64_t base = fb * F * (int64_t)n, total = a.n_frames * (int64_t)n;
            for (int cg = tid; cg < cols; cg += NT)
                if (base + cg < total) reinterpret_cast<float *>(a.soft)[base + cg] = lds_f32(smem, (uint32_t)cg * CB);
        }
        if (a.work && tid == 0) *claim_word = claimed;
        lds_barrier();
        if (row_valid && r == 0 && my_frame < a.n_frames) {
            if (a.iters) a.iters[my_frame] = my_iters;
            if (a.conv) a.conv[my_frame] = (uint8_t)my_conv;
            if (a.ones) a.ones[my_frame] = ones_cnt[f];
        }
        const int64_t after = a.work ? 2 * (int64_t)gridDim.x + __builtin_amdgcn_readfirstlane(*claim_word)
                                     : fb_next + gridDim.x;
        lds_barrier();
        fb = fb_next;
        fb_next = after;
    }
}
