// Float32 min-sum on block-circulant codes with heavy circulants: the state of a gym episode, where
// every action leaves a circulant of up to 7 hot bits and the environment keeps them all
// (ldpc_env.py:293-317; hot-bit budget models.py:296, 368), so that after a few dozen actions a check
// row has 50 ... 112 edges.  Explicit messages no longer fit the LDS (4 bytes x 55,000 ... 114,000
// edges); a check row's outgoing messages are kept in the compressed form plain min-sum allows
// (ldpc.py:190-191: two magnitudes per row, one sign per edge).
//
// This header is the part the host and the device code share: limits, the LDS layout and the
// tables of a code.  The kernel is in qc_heavy.hip.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

enum {
    HQ_WMAX = 7,        // largest circulant weight (what a gym action installs at most: models.py:296, 368)
    HQ_CPT = 8,         // block-columns per thread group: ceil(nb / mb) <= HQ_CPT
    HQ_MAXMB = 8,       // block-rows
    HQ_NW = 4,          // sign words of a check row: up to 128 edges
    HQ_BLK = 16,        // words of one block in the column table
    HQ_CH = 8,          // edges per step of the check pass
    HQ_PAD = 4          // rows are padded to a multiple of this with dummy edges (the walk can end in mid-step)
};

// LDS layout of one frame.  Everything a lane reads at "its position + a shift" is stored TWICE, at
// index x and at x + Z, so that (i + s) and (c - s + Z) need no reduction modulo Z: one vector add
// makes the address.
//   posteriors   [nb][2Z] float
//   dummy        [Zp] float, +inf: what the padding edges of a row read (never a minimum, sign +)
//   row records  [nwu][mb][2Z] x 16 bytes { 2*min1, 2*min2, column-table word of the arg-min edge, sign word w }:
//                the messages a row sent last, one record per sign word so that a column fetches all it needs
//                about one edge with one 16-byte read (nwu <= 2), or
//   row records  [mb][2Z] x 16 bytes { 2*min1, 2*min2, column-table word of the arg-min edge, - } and sign words
//                [nwu][mb][2Z] x 4 (nwu 3 or 4: the replicated form would not fit)
//   flags        4 words
//   row table    [2][mb][tabw] words (HeavyTables::rowtab)
struct HeavyGeom {
    uint32_t Z, Z2, Zp, mb, nb, nwu;
    bool split;                 // sign words apart from the magnitudes
    uint32_t dummy_off, rec_off, rec_wstride, sw_off, sw_wstride, flag_off, rowtab_off, tabw, bytes;
    __host__ __device__ HeavyGeom(int Z_, int Zp_, int mb_, int nb_, int nwu_)
        : Z(Z_), Z2(2 * Z_), Zp(Zp_), mb(mb_), nb(nb_), nwu(nwu_)
    {
        split = nwu > 2;
        dummy_off = nb * Z2 * 4;
        rec_off = (dummy_off + Zp * 4 + 15) & ~15u;
        rec_wstride = mb * Z2 * 16;
        const uint32_t rec_bytes = (split ? 1 : nwu) * rec_wstride;
        sw_off = rec_off + rec_bytes;
        sw_wstride = mb * Z2 * 4;
        flag_off = sw_off + (split ? nwu * sw_wstride : 0);
        rowtab_off = flag_off + 16;                 // [mb][tabw] words: the check pass's edge table, as long as the
        tabw = 32 * (nwu <= 2 ? 2 : 4);             // kernel variant's unrolled walk (which reads ahead of the row's end)
        bytes = rowtab_off + 2 * mb * tabw * 4;     // ... followed by [mb][tabw] words: the edges' column-table words
    }
};

struct HeavyTables {
    int Z, Zp, mb, nb, n, nwu;
    int deg[HQ_MAXMB];          // edges of a row of block-row b
    int dpad[HQ_MAXMB];         // ... padded to a multiple of HQ_PAD
    int ncol[HQ_MAXMB];         // block-columns thread group g owns: j = g + q mb < nb, q < ncol[g]
    // [2][mb][HeavyGeom::tabw], copied into LDS at launch.  First half: byte address of the posterior that
    // lane 0 of block-row b reads along edge e (lane i adds 4 i); padding edges point to the dummy row.
    // Second half: the edge's word in the column table (below), which a row stores for its arg-min edge.
    const uint32_t *rowtab;
    // [mb][HQ_CPT * mb + 2] blocks of 16 words (split layout: 32), read with scalar loads (the next block's
    // while this one is summed); for thread group g its t-th block, t = q mb + b, is block (b, g + q mb).
    // Two words per shift s, shifts descending, k = 0 .. weight - 1:
    //   word 2k      byte address of the record of row (Z - s) of block-row b, in the copy that belongs to the
    //                edge's sign word (lane c adds 16 c)
    //   word 2k + 1  s << 18 | edge number in the row's list << 5 | (31 - edge number mod 32)
    //   word 14      weight of the block
    //   word 16 + k  split layout: byte address of that row's sign word (lane c adds 4 c)
    const uint32_t *coltab;
};

// qc_heavy.hip
const void *heavyqc_kernel(int nwu);
int heavyqc_launch(const void *fn, const HeavyTables &tb, const struct DecodeArgs &a, int grid, int block, size_t lds,
                   hipStream_t stream);
