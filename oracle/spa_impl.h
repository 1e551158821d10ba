/*
 * TEST INFRASTRUCTURE -- not product code.
 *
 * Type-generic CPU statement of the flooding sum-product decoder the HIP library
 * offers as algo 1.  PARITY UNPINNED: the reference has no sum-product rule at all
 * (its only check-node update is min-sum, ldpc.py:174-202), so no fixture of the
 * reference covers this file; it is the definition the GPU kernel is compared with.
 * Everything except the check-node magnitude rule is the reference's decoder:
 * schedule and stop test (ldpc.py:326-337), v2c = posterior - previous c2v
 * (ldpc.py:161), sign(0) = +1 (ldpc.py:139), ordered column sum with the channel
 * value added last (ldpc.py:298-303, 322), bit = x > 0 (ldpc.py:47).
 *
 * Check rule, in the log domain with phi(x) = -ln tanh(x/2) = log1p(2/expm1(x)):
 *     F_e  = phi(clamp(|v2c_e|)),           clamp(x) = min(max(x, 1e-12), 80)
 *     F1   = the largest F of the row (the first one on ties), rest = sum of the
 *            others accumulated in edge order (a displaced F1 is added when it is
 *            displaced)
 *     X_e  = rest                      if e holds F1
 *          = (rest - F_e) + F1         otherwise
 *     |c2v_e| = phi(clamp(X_e)),  sign as in the min-sum decoder.
 * Included twice by ldpc_oracle.c (REAL = double, float).
 */
static REAL FN(spa_clamp)(REAL x)
{
    return x < (REAL)1e-12 ? (REAL)1e-12 : (x > (REAL)80 ? (REAL)80 : x);
}

static void FN(spa_frame)(const oracle_graph *g, int sign_mode, int max_iters, REAL llr_scale,
                          const REAL *y, uint8_t *hard, int32_t *iters_out,
                          uint8_t *conv_out, REAL *soft_out,
                          REAL *c2v, REAL *llr, REAL *post)
{
    const int m = g->m, n = g->n;
    const int32_t *rp = g->row_ptr, *ci = g->col_idx;
    const int32_t *cp = g->col_ptr, *ce = g->col_edge;
    int it = 0;
    int ok;

    for (int c = 0; c < n; ++c) {
        llr[c] = y[c] * llr_scale + (REAL)0;             /* + 0: -0.0 -> +0.0 */
        post[c] = llr[c];
    }
    for (int e = 0; e < g->E; ++e) c2v[e] = (REAL)0;

    ok = FN(syndrome_is_zero)(g, post);
    while (!ok && it < max_iters) {
        ++it;
        for (int r = 0; r < m; ++r) {
            const int e0 = rp[r], e1 = rp[r + 1];
            REAL F1 = (REAL)-1, rest = (REAL)0;
            int arg = -1, neg = 0;
            for (int e = e0; e < e1; ++e) {
                const REAL v = post[ci[e]] - c2v[e];
                const int minus_v = v < (REAL)0;                 /* sign(0) = +1 */
                neg ^= minus_v;
                const REAL F = PHI(FN(spa_clamp)(FABS(v)));
                if (F > F1) {
                    if (arg >= 0) rest = rest + F1;
                    F1 = F;
                    arg = e;
                } else {
                    rest = rest + F;
                }
                c2v[e] = minus_v ? -F : F;                       /* F > 0 (the clamp): its sign carries the edge's, for any degree */
            }
            if (sign_mode == 1 && ((e1 - e0) & 1)) neg ^= 1;
            for (int e = e0; e < e1; ++e) {
                const REAL sf = c2v[e];
                const REAL x = (e == arg) ? rest : (rest - FABS(sf)) + F1;
                const REAL mag = PHI(FN(spa_clamp)(x));
                const int minus = (sf < (REAL)0) ^ neg;
                c2v[e] = minus ? -mag : mag;
            }
        }
        for (int c = 0; c < n; ++c) {
            REAL acc = (REAL)0;
            for (int k = cp[c]; k < cp[c + 1]; ++k) acc = acc + c2v[ce[k]];
            post[c] = acc + llr[c];
        }
        ok = FN(syndrome_is_zero)(g, post);
    }
    for (int c = 0; c < n; ++c) {
        if (hard) hard[c] = post[c] > (REAL)0;
        if (soft_out) soft_out[c] = post[c];
    }
    *iters_out = it;
    *conv_out = (uint8_t)ok;
}
