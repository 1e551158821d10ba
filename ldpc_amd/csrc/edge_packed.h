// float32 min-sum for SHORT codes given as an edge list (802.11n n = 1944, toy codes ...): as many frames
// per workgroup as the LDS holds, and every thread working on the SAME check row (and the same columns)
// of RPT frames at once.  Included by ldpc_hip.hip.
//
// Why a second short-code kernel: with one frame per compute unit a pass over rows of 8 edges is a chain
// of LDS latencies and two barriers with nothing to overlap them (measured: 5,000 clocks per iteration
// for 7,000 edges, where the NearEarth kernel spends 7,400 on 32,700).  Here
//   * the frames of a workgroup come in groups of RPT (2 or 4) whose records are interleaved word by word:
//     a column's record is { posterior x RPT, message of its 1st check x RPT, 2nd check x RPT, ... }, so ONE
//     ds_read_b64 / ds_read_b128 fetches an edge's posterior (or message) for all frames of the group, with
//     one address register per edge for all of them;
//   * a record is as long as the column's degree asks (edge_slots.h reserves 4, 8 or 12 slots for every
//     column: 101 KB per frame at rate 1/2, 40 KB here), padded to an odd number of (RPT-word) units:
//     neighbouring lanes work on neighbouring columns of a circulant, whose records have one length,
//     and an even stride would put them on half or a quarter of the LDS banks (measured: 1.5 x);
//   * thread t owns check row t mod m of group t div m, and the columns t, t + NT, ... of the groups'
//     G * n columns: the RPT frames' reductions run side by side in its registers.
// Frames of a group that have stopped are carried along (their posteriors are kept by a select in the
// variable pass, their messages no longer matter).  Arithmetic and order are those of the other min-sum
// kernels (ldpc.py:150-163, 174-202, 295-324, 326-337): results are bit-identical.
#pragma once

template <int RPT> struct LdsVec;
template <> struct LdsVec<2> { typedef float __attribute__((ext_vector_type(2))) type; };
template <> struct LdsVec<4> { typedef float __attribute__((ext_vector_type(4))) type; };

template <int RPT>
__device__ __forceinline__ void lds_get_vec(uint32_t addr, float (&out)[RPT])
{
    typedef typename LdsVec<RPT>::type V;
    const V v = *reinterpret_cast<const __attribute__((address_space(3))) V *>((uintptr_t)addr);
#pragma unroll
    for (int j = 0; j < RPT; ++j) out[j] = v[j];
}

template <int RPT>
__device__ __forceinline__ void lds_put_vec(uint32_t addr, const float (&in)[RPT])
{
    typedef typename LdsVec<RPT>::type V;
    V v;
#pragma unroll
    for (int j = 0; j < RPT; ++j) v[j] = in[j];
    *reinterpret_cast<__attribute__((address_space(3))) V *>((uintptr_t)addr) = v;
}

struct PackedGeom {
    enum { YTOT = 8 };         // channel values a thread holds: (columns per thread) x RPT
    uint32_t GS;               // bytes per group of RPT frames: RPT * 4 * (rec_words + 2): the records + one dummy record
    uint32_t flag_off, cont_off, claim_off, ones_off, bytes;
    __host__ __device__ PackedGeom(int rec_words, int F, int RPT)
    {
        GS = 4u * (uint32_t)RPT * (uint32_t)(rec_words + 2);
        flag_off = GS * (uint32_t)(F / RPT);  // [2][F] "syndrome of frame non-zero" (16-byte aligned: GS is)
        cont_off = flag_off + 2u * F * 4;     // [2]    "some frame goes on"
        claim_off = cont_off + 8;             // [1]    block of frames this workgroup decodes after the next one
        ones_off = claim_off + 8;             // [F]
        // + 64 * RPT: the variable pass reads a column's slots four at a time and may run up to 11 units past the last record
        bytes = (ones_off + (uint32_t)F * 4 + 64u * RPT + 15) / 16 * 16;
    }
};

template <int DC, int RPT>
__global__ void __launch_bounds__(1024)
packed_minsum_kernel(const SlotTables tb, const DecodeArgs a)
{
    enum { YG = PackedGeom::YTOT / RPT, CH = 4, NCH = DC / CH, DVMAX = 12, UNIT = 4 * RPT };
    static_assert(DC % CH == 0 && DC <= 32 && (RPT == 2 || RPT == 4), "degree bounds");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = tb.m, n = tb.n, F = tb.F;
    const PackedGeom G(tb.rec_words, F, RPT);
    const uint32_t GS = G.GS;
    const int NT = blockDim.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int groups = F / RPT;
    const int gcols = groups * n;                              // columns of all groups
    const int nbytes = (n + 7) >> 3;
    int *flags = reinterpret_cast<int *>(smem + G.flag_off);
    int *cont = reinterpret_cast<int *>(smem + G.cont_off);
    int *ones_cnt = reinterpret_cast<int *>(smem + G.ones_off);
    const float INF = __uint_as_float(0x7f800000u);

    // ---- per-launch set-up: my row, the LDS addresses of its edges
    const bool row_valid = tid < groups * m;
    const int grp = row_valid ? tid / m : 0;
    const int r = row_valid ? tid - grp * m : 0;
    const int f0 = grp * RPT;                                  // my frames: f0 .. f0 + RPT - 1
    const int e0 = row_valid ? tb.row_ptr[r] : 0;
    const int deg = row_valid ? tb.row_ptr[r + 1] - e0 : 0;
    const uint32_t dummy = (uint32_t)UNIT * (uint32_t)tb.rec_words;   // inside a group: posterior +inf, one message
    uint32_t radr[DC], madr[DC];
#pragma unroll
    for (int e = 0; e < DC; ++e) {
        uint32_t pa = dummy, ma = dummy + UNIT;
        if (e < deg) {
            pa = (uint32_t)UNIT * (uint32_t)tb.col_rec[tb.col_idx[e0 + e]];
            ma = pa + (uint32_t)UNIT * (1u + (uint32_t)tb.edge_slot[e0 + e]);
        }
        radr[e] = opaque(pa + (uint32_t)grp * GS);
        madr[e] = opaque(ma + (uint32_t)grp * GS);
    }
    const unsigned long long spare_odd = __ballot(((DC - deg) & 1) != 0);
    // my columns: record address, group (one byte each), degree (four bits each; 15 = no column).
    // Groups and degrees are unpacked from laundered copies inside the loops: unpacked once, outside,
    // they would occupy registers for the whole launch.
    uint32_t crec[YG];
    uint32_t colg = 0, cdeg = 0;
#pragma unroll
    for (int k = 0; k < YG; ++k) {
        const int cg = k * NT + tid;
        crec[k] = 0;
        uint32_t dg = 15u;
        if (cg < gcols) {
            const int gg = cg / n, c = cg - gg * n;
            crec[k] = (uint32_t)gg * GS + (uint32_t)UNIT * (uint32_t)tb.col_rec[c];
            colg |= (uint32_t)gg << (8 * k);
            dg = (uint32_t)(tb.col_ptr[c + 1] - tb.col_ptr[c]);
        }
        cdeg |= dg << (4 * k);
    }
    auto col_degree = [](uint32_t cd, int k) { return (int)((cd >> (4 * k)) & 15u); };
    auto col_group = [](uint32_t cg, int k) { return (int)((cg >> (8 * k)) & 0xffu); };
    if (tid < groups) {
        float inf[RPT], zero[RPT];
#pragma unroll
        for (int j = 0; j < RPT; ++j) { inf[j] = INF; zero[j] = 0.0f; }
        lds_put_vec<RPT>((uint32_t)tid * GS + dummy, inf);
        lds_put_vec<RPT>((uint32_t)tid * GS + dummy + UNIT, zero);
    }

    const int64_t n_fblocks = (a.n_frames + F - 1) / F;
    // channel values of my columns in their RPT frames, prefetched one block of frames ahead (float32 inputs)
    float ypf[YG][RPT];
#pragma unroll
    for (int k = 0; k < YG; ++k)
#pragma unroll
        for (int j = 0; j < RPT; ++j) ypf[k][j] = 0.0f;
    auto fetch = [&](int64_t fb, float (&dst)[YG][RPT]) {
        const int t = (int)opaque((uint32_t)tid);             // addresses recomputed here, not kept across the decode
#pragma unroll
        for (int k = 0; k < YG; ++k) {
            const int cg = k * NT + t;
            if (cg < gcols) {
                const int gg = cg / n, c = cg - gg * n;
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    const int64_t fr = fb * F + gg * RPT + j;
                    if (fr < a.n_frames)
                        dst[k][j] = a.llr_f64 ? (float)reinterpret_cast<const double *>(a.llr)[fr * n + c]
                                              : reinterpret_cast<const float *>(a.llr)[fr * n + c];
                }
            }
        }
    };
    if ((int64_t)blockIdx.x < n_fblocks && !a.llr_f64) fetch(blockIdx.x, ypf);

    // Blocks of frames are claimed from a device counter when a.work is set, two blocks ahead
    // (as in the other short-code kernel).
    int *claim_word = reinterpret_cast<int *>(smem + G.claim_off);
    int64_t fb_next = (int64_t)blockIdx.x + gridDim.x;
    for (int64_t fb = blockIdx.x; fb < n_fblocks;) {
        if (a.llr_f64) fetch(fb, ypf);
        // bit j: my frame f0 + j exists and has not stopped
        uint32_t act = 0;
#pragma unroll
        for (int j = 0; j < RPT; ++j)
            if (row_valid && fb * F + f0 + j < a.n_frames) act |= 1u << j;
        // ---- load the block: posterior = channel value, messages = +0 (fresh decoder, ldpc.py:121)
        float ycur[YG][RPT];
        {
            const uint32_t cd0 = opaque(cdeg);
            float zero[RPT];
#pragma unroll
            for (int j = 0; j < RPT; ++j) zero[j] = 0.0f;
#pragma unroll
            for (int k = 0; k < YG; ++k) {
#pragma unroll
                for (int j = 0; j < RPT; ++j) ycur[k][j] = ypf[k][j] * a.llr_scale + 0.0f;      // "+ 0": -0.0 -> +0.0
                const int dg = col_degree(cd0, k);
                if (dg != 15) {
                    lds_put_vec<RPT>(crec[k], ycur[k]);
#pragma unroll
                    for (int sl = 0; sl < DVMAX; ++sl)
                        if (sl < dg) lds_put_vec<RPT>(crec[k] + (uint32_t)UNIT * (1u + sl), zero);
                }
            }
        }
        if (tid < 2 * F) flags[tid] = 0;
        if (tid < 2) cont[tid] = 0;
        if (tid < F) ones_cnt[tid] = 0;
        int my_iters[RPT];
        uint32_t conv = 0;                                         // bit j: frame f0 + j ended as a codeword
#pragma unroll
        for (int j = 0; j < RPT; ++j) my_iters[j] = 0;
        lds_barrier();
        int claimed = 0;
        if (a.work && tid == 0) claimed = atomicAdd(a.work, 1);
        if (fb_next < n_fblocks && !a.llr_f64) fetch(fb_next, ypf);

        for (int it = 0;; ++it) {
            const int par = it & 1;
            // ================= check pass (+ syndrome of the current posterior): my row in my RPT frames
            if (act != 0) {
                float min1[RPT], min2[RPT];
                uint32_t idx[RPT], R[RPT];
                unsigned long long synd[RPT];
#pragma unroll
                for (int j = 0; j < RPT; ++j) { min1[j] = INF; min2[j] = INF; idx[j] = 0; R[j] = 0; synd[j] = spare_odd; }
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    float p[CH][RPT], c[CH][RPT];
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        lds_get_vec<RPT>(radr[k * CH + u], p[u]);
                        lds_get_vec<RPT>(madr[k * CH + u], c[u]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < RPT; ++j) {
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int s = k * CH + u;
                            synd[j] ^= __ballot(p[u][j] > 0.0f);                        // slicer, ldpc.py:47
                            const float v = p[u][j] - c[u][j];                          // receive, ldpc.py:161
                            R[j] = __builtin_amdgcn_alignbit(R[j], __float_as_uint(v), 31);   // (R << 1) | sign(v), ldpc.py:139
                            const float av = fabsf(v);                                  // ldpc.py:178
                            const bool lt = av < min1[j];
                            idx[j] = lt ? (uint32_t)s : idx[j];
                            min2[j] = __builtin_amdgcn_fmed3f(min1[j], min2[j], av);
                            min1[j] = fminf(min1[j], av);
                        }
                        // the chunk is finished before the next one's loads are issued (both sides are volatile
                        // statements); otherwise all of a pass's loads are hoisted to its top
                        asm volatile("" : "+v"(min1[j]), "+v"(min2[j]), "+v"(idx[j]), "+v"(R[j]), "+s"(synd[j]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                uint32_t T[RPT], m1[RPT], m2[RPT];
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    uint32_t Rj = R[j];
                    if (DC < 32) Rj <<= (32 - DC);
                    uint32_t S = __popc(Rj) & 1u;                                       // spare edges are positive
                    if (a.sign_mode == LDPC_SIGN_CORRECTED) S ^= (uint32_t)deg & 1u;
                    T[j] = Rj ^ (S ? (0xffffffffu << (32 - DC)) : 0u);
                    float q1 = min1[j], q2 = min2[j];
                    if (a.ms_alpha != 1.0f || a.ms_beta != 0.0f) {
                        q1 = ms_correct(q1, a.ms_alpha, a.ms_beta);
                        q2 = ms_correct(q2, a.ms_alpha, a.ms_beta);
                    }
                    m1[j] = __float_as_uint(q1);
                    m2[j] = __float_as_uint(q2);
                    if (((act >> j) & 1u) && ((synd[j] >> lane) & 1ull)) {
                        flags[par * F + f0 + j] = 1;
                        cont[par] = 1;
                    }
                }
#pragma unroll
                for (int s = 0; s < DC; ++s) {
                    float out[RPT];
#pragma unroll
                    for (int j = 0; j < RPT; ++j) {
                        const uint32_t mag = idx[j] == (uint32_t)s ? m2[j] : m1[j];     // ldpc.py:190-191
                        out[j] = __uint_as_float(mag | ((T[j] << s) & 0x80000000u));
                    }
                    lds_put_vec<RPT>(madr[s], out);
                    if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            lds_barrier();
            // ================= stop test (ldpc.py:332-335): my frames, and whether any frame goes on
            const int any_bad = __builtin_amdgcn_readfirstlane(cont[par]);
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                if (((act >> j) & 1u) && flags[par * F + f0 + j] == 0) {
                    my_iters[j] = it;
                    conv |= 1u << j;
                    act &= ~(1u << j);
                }
            }
            if (any_bad == 0 || it >= a.max_iters) {
#pragma unroll
                for (int j = 0; j < RPT; ++j)
                    if ((act >> j) & 1u) my_iters[j] = it;
                break;
            }
            if (tid < F) flags[(par ^ 1) * F + tid] = 0;
            if (tid == 0) cont[par ^ 1] = 0;
            // ================= variable pass of the frames that go on (ldpc.py:295-311, 321-323)
            {
                const uint32_t cd = opaque(cdeg), cgp = opaque(colg);
#pragma unroll
                for (int k = 0; k < YG; ++k) {
                    const int dg = col_degree(cd, k);
                    if (dg != 15) {
                        int go[RPT];                               // the group's frames: syndrome non-zero, frame goes on
                        {
                            const uint32_t fa = G.flag_off + 4u * (uint32_t)(par * F + col_group(cgp, k) * RPT);
                            float fl[RPT];
                            lds_get_vec<RPT>(fa, fl);
#pragma unroll
                            for (int j = 0; j < RPT; ++j) go[j] = (int)__float_as_uint(fl[j]);
                        }
                        int any = 0;
#pragma unroll
                        for (int j = 0; j < RPT; ++j) any |= go[j];
                        if (any != 0) {
                            const uint32_t rec = crec[k];
                            // The column's messages in ascending check index.  Slots are read four at a time so
                            // that their latencies overlap; what is read beyond the degree (the next record)
                            // counts as +0, which changes no sum (x + 0 == x, and a sum of -0 terms meets a
                            // channel value that is never -0 in the end).
                            float acc[RPT], old[RPT];
#pragma unroll
                            for (int j = 0; j < RPT; ++j) acc[j] = 0.0f;
                            lds_get_vec<RPT>(rec, old);
#pragma unroll
                            for (int g4 = 0; g4 < DVMAX; g4 += 4) {
                                if (g4 < dg) {
                                    float v[4][RPT];
#pragma unroll
                                    for (int u = 0; u < 4; ++u) lds_get_vec<RPT>(rec + (uint32_t)UNIT * (uint32_t)(1 + g4 + u), v[u]);
#pragma unroll
                                    for (int u = 0; u < 4; ++u)
#pragma unroll
                                        for (int j = 0; j < RPT; ++j) acc[j] = acc[j] + (g4 + u < dg ? v[u][j] : 0.0f);
                                }
                            }
                            float nw[RPT];
#pragma unroll
                            for (int j = 0; j < RPT; ++j) nw[j] = go[j] != 0 ? acc[j] + ycur[k][j] : old[j];   // channel last, ldpc.py:322
                            lds_put_vec<RPT>(rec, nw);
                        }
                    }
                }
            }
            lds_barrier();
        }

        // ================= outputs
        if ((n & 7) == 0) {
            // A wave holds 64 consecutive columns (of one group, or the end of one and the start of the next):
            // the hard decisions of frame j are one ballot, and lane b < 8 stores byte b of it.  Columns and
            // frames are whole bytes long, so a byte never straddles two groups.
            const int w0 = (int)opaque((uint32_t)(tid & ~63));
#pragma unroll
            for (int k = 0; k < YG; ++k) {
                const int cg0 = k * NT + w0;
                if (cg0 < gcols) {                                  // uniform per wave
                    float pv[RPT];
                    lds_get_vec<RPT>(crec[k], pv);
                    const bool mine = cg0 + lane < gcols;
                    const int cb = cg0 + 8 * lane;                  // first column (of all groups) of my byte, lane < 8
                    const int gb = cb / n;
                    const int qb = (cb - gb * n) >> 3;
#pragma unroll
                    for (int j = 0; j < RPT; ++j) {
                        const unsigned long long bits = __ballot(mine && pv[j] > 0.0f);
                        const int64_t fr = fb * F + gb * RPT + j;
                        if (lane < 8 && cb < gcols && fr < a.n_frames) {
                            const uint32_t byte = (uint32_t)(bits >> (8 * lane)) & 0xffu;
                            if (a.hard) a.hard[fr * nbytes + qb] = (uint8_t)byte;
                            if (byte) atomicAdd(&ones_cnt[gb * RPT + j], (int)__popc(byte));
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
                    const uint32_t base = (uint32_t)(ff / RPT) * GS + 4u * (uint32_t)(ff % RPT);
                    uint32_t byte = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int c = qb * 8 + k;
                        if (c < n) byte |= (uint32_t)(lds_f32(smem, base + (uint32_t)UNIT * (uint32_t)tb.col_rec[c]) > 0.0f) << k;
                    }
                    if (a.hard) a.hard[fr * nbytes + qb] = (uint8_t)byte;
                    if (byte) atomicAdd(&ones_cnt[ff], (int)__popc(byte));
                }
            }
        }
        if (a.soft) {
            const int t = (int)opaque((uint32_t)tid);
#pragma unroll
            for (int k = 0; k < YG; ++k) {
                const int cg = k * NT + t;
                if (cg < gcols) {
                    const int gg = cg / n, c = cg - gg * n;
                    float pv[RPT];
                    lds_get_vec<RPT>(crec[k], pv);
#pragma unroll
                    for (int j = 0; j < RPT; ++j) {
                        const int64_t fr = fb * F + gg * RPT + j;
                        if (fr < a.n_frames) reinterpret_cast<float *>(a.soft)[fr * n + c] = pv[j];
                    }
                }
            }
        }
        if (a.work && tid == 0) *claim_word = claimed;
        lds_barrier();
        if (row_valid && r == 0) {
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int64_t fr = fb * F + f0 + j;
                if (fr < a.n_frames) {
                    if (a.iters) a.iters[fr] = my_iters[j];
                    if (a.conv) a.conv[fr] = (uint8_t)((conv >> j) & 1u);
                    if (a.ones) a.ones[fr] = ones_cnt[f0 + j];
                }
            }
        }
        const int64_t after = a.work ? 2 * (int64_t)gridDim.x + __builtin_amdgcn_readfirstlane(*claim_word)
                                     : fb_next + gridDim.x;
        lds_barrier();
        fb = fb_next;
        fb_next = after;
    }
}
