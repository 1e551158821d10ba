"""TEST INFRASTRUCTURE -- not product code.

Python face of the CPU oracle: a ctypes binding of oracle/libldpc_oracle.so
(C restatement of /root/reference/ldpc.py:249-337) and a small pure-NumPy
restatement of the same contract used to cross-check the C code on tiny cases.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The shipped package (ldpc_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, 'libldpc_oracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('ldpc_oracle.c', 'minsum_impl.h', 'spa_impl.h', 'Makefile')]
    srcs = [s for s in srcs if os.path.exists(s)]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libldpc_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get('LDPC_ORACLE_SO')            # another build of the same sources (oracle/Makefile: asan)
        if not so:
            so = os.path.join(_HERE, 'libldpc_oracle.so')
            if not os.path.exists(so):
                build()
        L = ctypes.CDLL(so)
        L.oracle_graph_create.restype = ctypes.c_void_p
        L.oracle_graph_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_graph_destroy.argtypes = [ctypes.c_void_p]
        for name in ('oracle_minsum_f64', 'oracle_minsum_f32', 'oracle_spa_f64', 'oracle_spa_f32'):
            if hasattr(L, name):
                fn = getattr(L, name)
                fn.restype = ctypes.c_int
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_set_minsum_scaling.argtypes = [ctypes.c_float, ctypes.c_float]
        _LIB = L
    return _LIB


def dense_to_csr(H):
    """Row adjacency in ascending column order (ldpc.py:244)."""
    H = np.asarray(H)
    m, n = H.shape
    rows, cols = np.nonzero(H)          # row-major => ascending columns inside a row
    row_ptr = np.zeros(m + 1, np.int32)
    np.add.at(row_ptr, rows + 1, 1)
    row_ptr = np.cumsum(row_ptr, dtype=np.int64).astype(np.int32)
    return m, n, row_ptr, cols.astype(np.int32)


def make_llrs(seed, snr_db, n, n_frames):
    """Host-seeded channel outputs, float32, as the reference's GPU driver draws
    them (ldpcCUDA.py:509-517, 536, 581): one RandomState(seed) stream, frame
    after frame; all-zero codeword -> BPSK -1 (ldpc.py:62-66)."""
    prng = np.random.RandomState(seed)
    sigma = np.sqrt(0.5 / (10 ** (snr_db / 10)))
    out = np.empty((n_frames, n), dtype=np.float32)
    for k in range(n_frames):
        noise = np.float32(prng.normal(0, sigma, n))
        out[k] = np.float32(-1) + noise
    return out


class Oracle:
    """CPU decoder for a dense 0/1 parity-check matrix H."""

    def __init__(self, H):
        self.m, self.n, self.row_ptr, self.col_idx = dense_to_csr(H)
        self._g = lib().oracle_graph_create(self.m, self.n,
                                            self.row_ptr.ctypes.data, self.col_idx.ctypes.data)
        if not self._g:
            raise MemoryError('oracle_graph_create failed')

    def __del__(self):
        try:
            if self._g:
                lib().oracle_graph_destroy(self._g)
                self._g = None
        except Exception:
            pass

    def _run(self, fn_name, dtype, y, max_iters, sign_mode, want_soft, threads, extra=(), state=None):
        y = np.ascontiguousarray(np.atleast_2d(y), dtype=dtype)
        F, n = y.shape
        assert n == self.n
        hard = np.zeros((F, n), np.uint8)
        iters = np.zeros(F, np.int32)
        conv = np.zeros(F, np.uint8)
        soft = np.zeros((F, n), dtype) if want_soft else None
        fn = getattr(lib(), fn_name)
        rc = fn(ctypes.c_void_p(self._g), ctypes.c_int(sign_mode), ctypes.c_int(max_iters),
                ctypes.c_int64(F), *extra, ctypes.c_void_p(y.ctypes.data),
                ctypes.c_void_p(hard.ctypes.data), ctypes.c_void_p(iters.ctypes.data),
                ctypes.c_void_p(conv.ctypes.data),
                ctypes.c_void_p(soft.ctypes.data if want_soft else None),
                ctypes.c_void_p(state.ctypes.data if state is not None else None), ctypes.c_int(threads))
        if rc != 0:
            raise RuntimeError(f'{fn_name} failed: {rc}')
        return hard, iters, conv, soft

    def new_state(self, prec=64):
        """Check-node messages of a freshly built decoder (all zero), for minsum(state=...)."""
        return np.zeros(len(self.col_idx) + 1, np.float64 if prec == 64 else np.float32)

    def minsum(self, y, max_iters=50, prec=64, sign_mode=0, want_soft=True, threads=1, state=None,
               alpha=1.0, beta=0.0):
        """-> (hard uint8 [F,n], iters int32 [F], converged uint8 [F], soft [F,n] or None).
        state (from new_state, updated in place): decode the frames in order on ONE
        re-used decoder instead of a fresh decoder per frame."""
        dt = np.float64 if prec == 64 else np.float32
        if state is not None:
            assert state.dtype == dt and state.flags.c_contiguous
        lib().oracle_set_minsum_scaling(float(alpha), float(beta))     # alpha, beta: |c2v| = max(alpha*min - beta, 0)
        try:
            return self._run('oracle_minsum_f64' if prec == 64 else 'oracle_minsum_f32', dt, y, max_iters,
                             sign_mode, want_soft, threads, state=state)
        finally:
            lib().oracle_set_minsum_scaling(1.0, 0.0)

    def spa(self, y, llr_scale, max_iters=50, prec=64, sign_mode=0, want_soft=True, threads=1, phi_form=0):
        """Sum-product (tanh rule) on LLR = llr_scale * y.  No counterpart in the reference.
        phi_form=1 (float32 only, an experiment): phi in the three-branch form of the GPU kernel."""
        lib().oracle_set_spa_phi_form(int(phi_form))
        if prec == 64:
            return self._run('oracle_spa_f64', np.float64, y, max_iters, sign_mode, want_soft, threads,
                             extra=(ctypes.c_double(llr_scale),))
        return self._run('oracle_spa_f32', np.float32, y, max_iters, sign_mode, want_soft, threads,
                         extra=(ctypes.c_double(llr_scale),))


# --------------------------------------------------------------------------
# Pure-NumPy restatement (one frame), vectorised over edges.  Same contract:
# ldpc.py:161 (v2c), :135-141/:174-191 (check), :298-303/:322 (ordered sum,
# channel last), :47 (x > 0).  Used to cross-check the C oracle.
def minsum_numpy(H, y, max_iters=50, dtype=np.float64):
    H = np.asarray(H)
    m, n = H.shape
    er, ec = np.nonzero(H)                       # edges row-major
    E = len(er)
    y = np.asarray(y, dtype=dtype)
    row_start = np.searchsorted(er, np.arange(m))
    row_end = np.searchsorted(er, np.arange(m), side='right')
    # per column: edge ids in ascending row order, as a padded [n, dv] table
    order = np.lexsort((er, ec))
    col_sorted = ec[order]
    col_start = np.searchsorted(col_sorted, np.arange(n))
    col_cnt = np.bincount(ec, minlength=n)
    dv = int(col_cnt.max()) if E else 0

    def syndrome_zero(x):
        hardbits = (x > 0).astype(np.int64)
        s = np.zeros(m, np.int64)
        np.add.at(s, er, hardbits[ec])
        return not np.any(s % 2)

    post = y.copy()
    c2v = np.zeros(E, dtype)
    it = 0
    ok = syndrome_zero(post)
    while (not ok) and it < max_iters:
        it += 1
        v2c = post[ec] - c2v
        sgn = np.where(v2c < 0, dtype(-1), dtype(1))
        mag = np.abs(v2c)
        new = np.empty(E, dtype)
        for r in range(m):
            a, b = row_start[r], row_end[r]
            if b == a:
                continue
            mg = mag[a:b]
            S = np.prod(sgn[a:b])
            idx = np.argsort(mg, kind='stable')
            m0 = idx[0]
            min1 = mg[m0]
            min2 = mg[idx[1]] if b - a > 1 else dtype(np.inf)
            out = min1 * S * sgn[a:b]
            out[m0] = min2 * S * sgn[a + m0]
            new[a:b] = out
        c2v = new
        acc = np.zeros(n, dtype)
        for k in range(dv):                      # k-th check (ascending row) of every column
            has = col_cnt > k
            eid = order[col_start[has] + k]
            acc[has] = acc[has] + c2v[eid]
        post = acc + y
        ok = syndrome_zero(post)
    return (post > 0).astype(np.uint8), it, ok, post
