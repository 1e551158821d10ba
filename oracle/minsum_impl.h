/*
 * TEST INFRASTRUCTURE -- not product code.
 *
 * Type-generic body of the CPU restatement of the reference decoder; included
 * twice by ldpc_oracle.c with REAL = double (arithmetic of the reference's CPU
 * decoder, ldpc.py:29 LDPC_DECIMAL_DATA_TYPE = float64) and REAL = float
 * (arithmetic of the reference's numba.cuda decoder, ldpcCUDA.py:538-548, all
 * float32).  Compile with -ffp-contract=off: every rounding below is part of
 * the contract.
 */

/* One frame of flooding min-sum.  Restates
 *   ldpcDecoder.decoderMainLoop        ldpc.py:326-337
 *   ldpcDecoder.isCodeword             ldpc.py:249-268  (+ slicer ldpc.py:43-48)
 *   ldpcDecoder.decoderStep            ldpc.py:313-324
 *   variableNodesToCheckNodes/receive  ldpc.py:282-293, 150-163
 *   checkNode.setSign                  ldpc.py:135-141
 *   checkNode.calcOutgoingValues       ldpc.py:174-202
 *   checkNodesToVariableNodes/update   ldpc.py:295-311, 85-88
 * With keep_c2v == 0 every frame gets a FRESH decoder (check-node messages
 * zeroed, checkNode.__init__ ldpc.py:121).  With keep_c2v != 0 the messages in
 * c2v are kept from the previous frame, which is what happens when one
 * reference decoder object decodes several frames: decoderSet resets only the
 * variable nodes (ldpc.py:270-274; evaluateCode ldpc.py:356,377).
 *
 * c2v, v2c: scratch [E]; post: scratch [n].
 * sign_mode 0 = reference arithmetic (c2v sign = product of ALL signs times the
 * edge's own sign, ldpc.py:190-191); sign_mode 1 = additionally multiply by
 * (-1)^degree, which is the textbook rule for the reference's bit polarity
 * (0 -> -1, ldpc.py:62-66); the two coincide for even check degree.
 * alpha, beta: |c2v| = max(alpha*mag - beta, 0); (1, 0) is the reference's plain rule.
 */
static void FN(decode_frame)(const oracle_graph *g, int sign_mode, int keep_c2v, int max_iters,
                             REAL alpha, REAL beta,
                             const REAL *y, uint8_t *hard, int32_t *iters_out,
                             uint8_t *conv_out, REAL *soft_out,
                             REAL *c2v, REAL *v2c, REAL *post)
{
    const int m = g->m, n = g->n;
    const int32_t *rp = g->row_ptr, *ci = g->col_idx;
    const int32_t *cp = g->col_ptr, *ce = g->col_edge;
    int it = 0;
    int ok;

    for (int c = 0; c < n; ++c) post[c] = y[c];
    if (!keep_c2v)                                      /* a re-used decoder keeps them, ldpc.py:270-274 */
        for (int e = 0; e < g->E; ++e) c2v[e] = (REAL)0;

    ok = FN(syndrome_is_zero)(g, post);                 /* ldpc.py:327 */
    while (!ok && it < max_iters) {                      /* ldpc.py:332 */
        ++it;
        /* variable -> check: value minus what this check sent last time (ldpc.py:161) */
        for (int r = 0; r < m; ++r)
            for (int e = rp[r]; e < rp[r + 1]; ++e)
                v2c[e] = post[ci[e]] - c2v[e];
        /* check update (ldpc.py:174-191) */
        for (int r = 0; r < m; ++r) {
            const int e0 = rp[r], e1 = rp[r + 1];
            REAL min1 = (REAL)INFINITY, min2 = (REAL)INFINITY;
            int arg = -1, neg = 0;
            for (int e = e0; e < e1; ++e) {
                const REAL a = FABS(v2c[e]);
                if (v2c[e] < (REAL)0) neg ^= 1;          /* sign(0) = +1, ldpc.py:139 */
                if (a < min1) { min2 = min1; min1 = a; arg = e; }
                else if (a < min2) { min2 = a; }
            }
            REAL S = neg ? (REAL)-1 : (REAL)1;
            if (sign_mode == 1 && ((e1 - e0) & 1)) S = -S;
            for (int e = e0; e < e1; ++e) {
                const REAL sg = (v2c[e] < (REAL)0) ? (REAL)-1 : (REAL)1;
                REAL mag = (e == arg) ? min2 : min1;
                if (alpha != (REAL)1 || beta != (REAL)0) {         /* normalised / offset variant (not in the reference) */
                    mag = alpha * mag - beta;
                    if (!(mag > (REAL)0)) mag = (REAL)0;
                }
                c2v[e] = mag * S * sg;                    /* ldpc.py:190-191 */
            }
        }
        /* variable update: running sum from 0 over checks in ascending index
         * (ldpc.py:292, 298-303), channel value added last (ldpc.py:322) */
        for (int c = 0; c < n; ++c) {
            REAL acc = (REAL)0;
            for (int k = cp[c]; k < cp[c + 1]; ++k) acc = acc + c2v[ce[k]];
            post[c] = acc + y[c];
        }
        ok = FN(syndrome_is_zero)(g, post);              /* ldpc.py:335 */
    }
    for (int c = 0; c < n; ++c) {
        if (hard) hard[c] = post[c] > (REAL)0;           /* ldpc.py:46-47 */
        if (soft_out) soft_out[c] = post[c];
    }
    *iters_out = it;
    *conv_out = (uint8_t)ok;
}
