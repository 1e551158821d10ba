/*
 * TEST INFRASTRUCTURE -- not product code.
 *
 * CPU restatement ("oracle") of the reference's LDPC decode path, used only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker / timed CPU baseline.  The shipped library never links or calls it.
 *
 * Follows /root/reference/ldpc.py:43-66 (slicer, addAWGN, modulate),
 * :135-202 (check node), :249-337 (decoder loop) and nothing else.  Pinned by
 * tests/golden/ (outputs of the reference's own source executed in the build
 * container, see oracle/gen_golden.py).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct oracle_graph {
    int m, n, E;
    int32_t *row_ptr;   /* [m+1]                                              */
    int32_t *col_idx;   /* [E] columns of each row, ascending (ldpc.py:244)   */
    int32_t *col_ptr;   /* [n+1]                                              */
    int32_t *col_edge;  /* [E] edge ids of each column, ascending row
                           (= order in which checks update a variable,
                           ldpc.py:298-303)                                    */
} oracle_graph;

/* Graph from CSR (what ldpcDecoder.__init__ derives from dense H, ldpc.py:225-247). */
oracle_graph *oracle_graph_create(int m, int n, const int32_t *row_ptr, const int32_t *col_idx)
{
    oracle_graph *g = (oracle_graph *)calloc(1, sizeof(*g));
    if (!g) return NULL;
    g->m = m; g->n = n; g->E = row_ptr[m];
    g->row_ptr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1));
    g->col_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g->E > 0 ? g->E : 1));
    g->col_ptr = (int32_t *)calloc((size_t)(n + 1), sizeof(int32_t));
    g->col_edge = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g->E > 0 ? g->E : 1));
    memcpy(g->row_ptr, row_ptr, sizeof(int32_t) * (size_t)(m + 1));
    memcpy(g->col_idx, col_idx, sizeof(int32_t) * (size_t)g->E);
    for (int e = 0; e < g->E; ++e) g->col_ptr[col_idx[e] + 1]++;
    for (int c = 0; c < n; ++c) g->col_ptr[c + 1] += g->col_ptr[c];
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    memcpy(fill, g->col_ptr, sizeof(int32_t) * (size_t)n);
    for (int r = 0; r < m; ++r)                 /* rows ascending => per-column ascending row */
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e)
            g->col_edge[fill[col_idx[e]]++] = e;
    free(fill);
    return g;
}

void oracle_graph_destroy(oracle_graph *g)
{
    if (!g) return;
    free(g->row_ptr); free(g->col_idx); free(g->col_ptr); free(g->col_edge); free(g);
}

int oracle_graph_edges(const oracle_graph *g) { return g->E; }

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

/* isCodeword (ldpc.py:249-268): H . (x > 0) mod 2 == 0 */
#define SYNDROME_BODY                                                         \
    for (int r = 0; r < g->m; ++r) {                                          \
        int p = 0;                                                            \
        for (int e = g->row_ptr[r]; e < g->row_ptr[r + 1]; ++e)               \
            p ^= (x[g->col_idx[e]] > 0);                                      \
        if (p) return 0;                                                      \
    }                                                                         \
    return 1;

static int f64_syndrome_is_zero(const oracle_graph *g, const double *x) { SYNDROME_BODY }
static int f32_syndrome_is_zero(const oracle_graph *g, const float *x) { SYNDROME_BODY }

#define REAL double
#define FABS fabs
#define PHI(x) log1p(2.0 / expm1(x))
#define FN(name) CAT(f64_, name)
#include "minsum_impl.h"
#include "spa_impl.h"
#undef REAL
#undef FABS
#undef PHI
#undef FN

#define REAL float
#define FABS fabsf
/* Experiment switch (oracle_set_spa_phi_form): 1 = phi restated the way the float32 message-plane kernel
 * computes it (qc_planes.h planes_phi: three branches on exp2 / log2 / reciprocal with fused polynomial steps),
 * on the host's exp2f / log2f.  Tells whether the GPU's disagreement with this statement on frames that never
 * converge comes from the form of phi or from the last bits of the hardware's transcendental units. */
static int oracle_spa_phi_form = 0;
static float phi_device_form(float x)
{
    const float t = exp2f(x * -1.4426950408889634f);
    const float t2 = t * t;
    const float series = (t + t) * fmaf(t2, fmaf(t2, fmaf(t2, 0.14285715f, 0.2f), 0.33333334f), 1.0f);
    const float om_small = x * fmaf(-x, fmaf(-x, fmaf(-x, fmaf(-x, 0.008333334f, 0.041666668f), 0.16666667f), 0.5f), 1.0f);
    const float om = x < 0.0625f ? om_small : 1.0f - t;
    const float lg = log2f((1.0f + t) * (1.0f / om)) * 0.6931471805599453f;
    return x >= 2.0f ? series : lg;
}
#define PHI(x) (oracle_spa_phi_form ? phi_device_form(x) : log1pf(2.0f / expm1f(x)))
#define FN(name) CAT(f32_, name)
#include "minsum_impl.h"
#include "spa_impl.h"
#undef REAL
#undef FABS
#undef PHI
#undef FN

/* Magnitude correction of the min-sum drivers (1, 0 = the reference's plain rule); float on
 * purpose: the HIP library takes the two parameters as float. */
static float oracle_ms_alpha = 1.0f, oracle_ms_beta = 0.0f;
void oracle_set_minsum_scaling(float alpha, float beta) { oracle_ms_alpha = alpha; oracle_ms_beta = beta; }
void oracle_set_spa_phi_form(int form) { oracle_spa_phi_form = form; }

/* Batch drivers.  y: [frames][n]; hard: [frames][n] bytes 0/1 or NULL;
 * iters: [frames]; conv: [frames]; soft: [frames][n] or NULL.
 * state: NULL = fresh decoder per frame; else [E] check-to-variable messages of
 * a re-used decoder (in/out, row-major edge order), frames decoded in order on
 * one thread.  threads <= 0 -> 1 thread.  Returns 0, or -1 on allocation failure. */
#define BATCH_BODY(REAL_T, PFX)                                               \
    int fail = 0;                                                             \
    const float ms_alpha = oracle_ms_alpha, ms_beta = oracle_ms_beta;         \
    if (threads <= 0 || state) threads = 1;                                   \
    _Pragma("omp parallel num_threads(threads)")                              \
    {                                                                         \
        REAL_T *c2v = (REAL_T *)malloc(sizeof(REAL_T) * (size_t)(g->E + 1));  \
        REAL_T *v2c = (REAL_T *)malloc(sizeof(REAL_T) * (size_t)(g->E + 1));  \
        REAL_T *post = (REAL_T *)malloc(sizeof(REAL_T) * (size_t)(g->n + 1)); \
        if (!c2v || !v2c || !post) {                                          \
            _Pragma("omp atomic write")                                       \
            fail = 1;                                                         \
        } else {                                                              \
            _Pragma("omp for schedule(dynamic, 1)")                           \
            for (int64_t f = 0; f < n_frames; ++f)                            \
                PFX##decode_frame(g, sign_mode, state != NULL, max_iters,     \
                                  (REAL_T)ms_alpha, (REAL_T)ms_beta,          \
                                  y + (size_t)f * g->n,                       \
                                  hard ? hard + (size_t)f * g->n : NULL,      \
                                  iters + f, conv + f,                        \
                                  soft ? soft + (size_t)f * g->n : NULL,      \
                                  state ? state : c2v, v2c, post);            \
        }                                                                     \
        free(c2v); free(v2c); free(post);                                     \
    }                                                                         \
    return fail ? -1 : 0;

int oracle_minsum_f64(const oracle_graph *g, int sign_mode, int max_iters, int64_t n_frames,
                      const double *y, uint8_t *hard, int32_t *iters, uint8_t *conv,
                      double *soft, double *state, int threads)
{
    BATCH_BODY(double, f64_)
}

int oracle_minsum_f32(const oracle_graph *g, int sign_mode, int max_iters, int64_t n_frames,
                      const float *y, uint8_t *hard, int32_t *iters, uint8_t *conv,
                      float *soft, float *state, int threads)
{
    BATCH_BODY(float, f32_)
}

/* Sum-product batch drivers (always a fresh decoder per frame; `state` must be NULL). */
#define SPA_BATCH_BODY(REAL_T, PFX)                                           \
    int fail = 0;                                                             \
    if (state) return -2;                                                     \
    if (threads <= 0) threads = 1;                                            \
    _Pragma("omp parallel num_threads(threads)")                              \
    {                                                                         \
        REAL_T *c2v = (REAL_T *)malloc(sizeof(REAL_T) * (size_t)(g->E + 1));  \
        REAL_T *llr = (REAL_T *)malloc(sizeof(REAL_T) * (size_t)(g->n + 1));  \
        REAL_T *post = (REAL_T *)malloc(sizeof(REAL_T) * (size_t)(g->n + 1)); \
        if (!c2v || !llr || !post) {                                          \
            _Pragma("omp atomic write")                                       \
            fail = 1;                                                         \
        } else {                                                              \
            _Pragma("omp for schedule(dynamic, 1)")                           \
            for (int64_t f = 0; f < n_frames; ++f)                            \
                PFX##spa_frame(g, sign_mode, max_iters, (REAL_T)llr_scale,    \
                               y + (size_t)f * g->n,                          \
                               hard ? hard + (size_t)f * g->n : NULL,         \
                               iters + f, conv + f,                           \
                               soft ? soft + (size_t)f * g->n : NULL,         \
                               c2v, llr, post);                               \
        }                                                                     \
        free(c2v); free(llr); free(post);                                     \
    }                                                                         \
    return fail ? -1 : 0;

int oracle_spa_f64(const oracle_graph *g, int sign_mode, int max_iters, int64_t n_frames, double llr_scale,
                   const double *y, uint8_t *hard, int32_t *iters, uint8_t *conv,
                   double *soft, double *state, int threads)
{
    SPA_BATCH_BODY(double, f64_)
}

int oracle_spa_f32(const oracle_graph *g, int sign_mode, int max_iters, int64_t n_frames, double llr_scale,
                   const float *y, uint8_t *hard, int32_t *iters, uint8_t *conv,
                   float *soft, float *state, int threads)
{
    SPA_BATCH_BODY(float, f32_)
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
