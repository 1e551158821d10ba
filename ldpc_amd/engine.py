"""Python face of libldpc_hip.so: one `Decoder` = one ldpc_handle (one device,
one stream, one code).  NumPy in / NumPy out for the host entry points; raw
device pointers (e.g. torch tensors' data_ptr()) for the device entry points.
"""
import ctypes

import numpy as np

from . import _lib, codes

ALGO_MINSUM = 0
ALGO_SUMPRODUCT = 1
SIGN_REFERENCE = 0
SIGN_CORRECTED = 1

COUNTER_FIELDS = ('frames', 'frame_errors', 'bit_errors', 'iterations', 'uncoded_errors', 'not_codeword')


class LdpcHipError(RuntimeError):
    pass


def library_path():
    return _lib.library_path()


def release_pool():
    """Free what closed decoders left for their successors (streams, table allocations, small buffers);
    returns the number of sets freed.  For a long-running process that is done decoding."""
    return int(_lib.load().ldpc_release_pool())


def device_count():
    return int(_lib.load().ldpc_device_count())


def _check(rc):
    if rc != 0:
        msg = _lib.load().ldpc_last_error().decode('utf-8', 'replace')
        raise LdpcHipError(f'libldpc_hip error {rc}: {msg}')


def _ptr(x):
    """Device pointer of a torch tensor / int address / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if hasattr(x, 'data_ptr'):
        return ctypes.c_void_p(x.data_ptr())
    raise TypeError(f'cannot take a device pointer from {type(x)!r}')


COMM_ID_BYTES = 128


def comm_unique_id():
    """bytes [128] that rank 0 hands to every rank before Comm(...) (ldpc_comm_unique_id)."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(_lib.load().ldpc_comm_unique_id(buf))
    return buf.raw


class Comm:
    """The C library's own RCCL communicator (one rank per GPU), for callers without torch.distributed:
    allreduce(counters) replaces a uint64 array by its sum over the ranks."""

    def __init__(self, device, rank, world, unique_id):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f'unique_id must be {COMM_ID_BYTES} bytes')
        self._L = _lib.load()
        self._c = ctypes.c_void_p()
        _check(self._L.ldpc_comm_create(int(device), int(rank), int(world), ctypes.c_char_p(bytes(unique_id)),
                                        ctypes.byref(self._c)))
        self.rank, self.world = int(rank), int(world)

    def allreduce(self, counters):
        c = np.ascontiguousarray(counters, dtype=np.uint64)
        _check(self._L.ldpc_allreduce_counters(self._c, c.ctypes.data, int(c.size)))
        return c

    def close(self):
        if getattr(self, '_c', None) is not None and self._c:
            self._L.ldpc_comm_destroy(self._c)
            self._c = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Decoder:
    """LDPC decoder / code evaluator on one HIP device.

    `code` is a codes.QCCode, or a dense 0/1 matrix H (block-circulant
    structure is detected; anything else goes through the CSR path).
    """

    def __init__(self, code, device=0, circulant_size=None):
        L = _lib.load()
        self._L = L
        self._h = ctypes.c_void_p()
        if not isinstance(code, codes.QCCode):
            H = np.asarray(code)
            if H.ndim != 2:
                raise ValueError('H must be a 2-D matrix')
            qc = codes.detect_qc(H, [circulant_size] if circulant_size else None)
            code = qc if qc is not None else H
        if isinstance(code, codes.QCCode):
            self.qc = code
            self.m, self.n = code.m, code.n
            ptr, sh = code.flat()
            rc = L.ldpc_create_qc(device, code.Z, code.mb, code.nb, ptr.ctypes.data,
                                  sh.ctypes.data if sh.size else None, ctypes.byref(self._h))
            if rc == -3:   # LDPC_E_UNSUPPORTED: outside the QC kernels' limits
                code = code.to_dense()
            else:
                _check(rc)
        if not isinstance(code, codes.QCCode):
            self.qc = None
            m, n, row_ptr, col_idx = codes.dense_to_csr(code)
            self.m, self.n = m, n
            _check(L.ldpc_create_csr(device, m, n, row_ptr.ctypes.data,
                                     col_idx.ctypes.data if col_idx.size else None,
                                     ctypes.byref(self._h)))
        self.device = device
        self.nbytes = (self.n + 7) // 8

    # -- lifetime
    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._L.ldpc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- stream control
    def set_stream(self, hip_stream):
        """hip_stream: integer hipStream_t, e.g. torch.cuda.current_stream().cuda_stream
        (0 = the legacy default stream); None = back to the handle's own stream."""
        if hip_stream is None:
            _check(self._L.ldpc_set_stream(self._h, None, 1))
        else:
            _check(self._L.ldpc_set_stream(self._h, ctypes.c_void_p(int(hip_stream)), 0))

    def set_minsum_scaling(self, alpha=1.0, beta=0.0):
        """Normalised / offset min-sum for later decodes: |message| = max(alpha*min - beta, 0).
        (1, 0) is the reference's plain min-sum."""
        _check(self._L.ldpc_set_minsum_scaling(self._h, float(alpha), float(beta)))

    def set_pass_counters(self, d_counts):
        """Device pointer to two uint64 (or None): the message-plane kernels add their numbers of full
        and of syndrome-only check passes to them (measurement aid)."""
        _check(self._L.ldpc_set_pass_counters(self._h, _ptr(d_counts)))

    def sync(self):
        _check(self._L.ldpc_sync(self._h))

    def launch_info(self, algo=ALGO_MINSUM, prec=32):
        g, b, l, f = (ctypes.c_int() for _ in range(4))
        _check(self._L.ldpc_get_launch_info(self._h, algo, prec, ctypes.byref(g), ctypes.byref(b),
                                            ctypes.byref(l), ctypes.byref(f)))
        return dict(grid=g.value, block=b.value, lds_bytes=l.value, frames_per_wg=f.value)

    # -- host entry point
    def reset_state(self):
        """Chained decoding starts from fresh check nodes again."""
        _check(self._L.ldpc_reset_state(self._h))

    def decode(self, llr, max_iters=50, prec=64, algo=ALGO_MINSUM, sign_mode=SIGN_REFERENCE,
               want_soft=False, want_hard=True, chained=False):
        """Decode host frames.  llr: [F, n] (or [n]) channel values; float64 input is
        passed through as float64, anything else as float32.

        chained=True decodes the frames one after another on the handle's stateful
        decoder (check-node messages carry over, as when one reference ldpcDecoder
        object is re-used); the default gives every frame a fresh decoder.

        Returns dict(hard=uint8 [F, n] 0/1 (or None), iters=int32 [F], converged=uint8 [F],
        ones=int32 [F], soft=float32/float64 [F, n] or None)."""
        y = np.atleast_2d(llr)
        y = np.ascontiguousarray(y, dtype=np.float64 if y.dtype == np.float64 else np.float32)
        llr_bits = 64 if y.dtype == np.float64 else 32
        F, n = y.shape
        if n != self.n:
            raise ValueError(f'frame length {n} != code length {self.n}')
        hard = np.zeros((F, self.nbytes), np.uint8) if want_hard else None
        iters = np.zeros(F, np.int32)
        conv = np.zeros(F, np.uint8)
        ones = np.zeros(F, np.int32)
        soft = np.zeros((F, n), np.float64 if prec == 64 else np.float32) if want_soft else None
        fn = self._L.ldpc_decode_chained if chained else self._L.ldpc_decode
        _check(fn(self._h, algo, prec, sign_mode, int(max_iters), F, y.ctypes.data, llr_bits,
                                   hard.ctypes.data if want_hard else None, iters.ctypes.data,
                                   conv.ctypes.data, ones.ctypes.data,
              soft.ctypes.data if want_soft else None))
        bits = None
        if want_hard:
            bits = np.unpackbits(hard, axis=1, bitorder='little')[:, :n]
        return dict(hard=bits, hard_packed=hard, iters=iters, converged=conv, ones=ones, soft=soft)

    # -- device entry points (pointers: torch CUDA tensors or integer addresses)
    def decode_device(self, d_llr, n_frames, max_iters=50, prec=32, algo=ALGO_MINSUM,
                      sign_mode=SIGN_REFERENCE, llr_scale=1.0, llr_bits=32, d_hard=None, d_iters=None,
                      d_conv=None, d_ones=None, d_soft=None, noalloc=False):
        """noalloc=True: the entry point that reaches no hipMalloc (its scratch is a static device
        array of the library; calls that overlap on one GPU race)."""
        fn = self._L.ldpc_decode_device_noalloc if noalloc else self._L.ldpc_decode_device
        _check(fn(self._h, algo, prec, sign_mode, int(max_iters), int(n_frames),
                                          _ptr(d_llr), int(llr_bits), float(llr_scale), _ptr(d_hard), _ptr(d_iters),
                                          _ptr(d_conv), _ptr(d_ones), _ptr(d_soft)))

    def channel_awgn_device(self, d_llr, n_frames, snr_db, seed, first_frame=0, d_sigma_hat=None,
                            d_uncoded=None):
        _check(self._L.ldpc_channel_awgn_device(self._h, float(snr_db), int(seed), int(first_frame),
                                                int(n_frames), _ptr(d_llr), _ptr(d_sigma_hat),
                                                _ptr(d_uncoded)))

    # -- Monte-Carlo driver
    def eval_awgn(self, snr_db, frames_per_snr, seed, first_frame=0, max_iters=50, prec=32,
                  algo=ALGO_MINSUM, sign_mode=SIGN_REFERENCE, per_frame=False):
        """-> (counters uint64 [n_snr, 6] (COUNTER_FIELDS), per_frame float32 [n_snr, F, 4] or None)"""
        snr = np.ascontiguousarray(np.atleast_1d(snr_db), dtype=np.float64)
        cnt = np.zeros((len(snr), 6), np.uint64)
        pf = np.zeros((len(snr), int(frames_per_snr), 4), np.float32) if per_frame else None
        _check(self._L.ldpc_eval_awgn(self._h, algo, prec, sign_mode, int(max_iters), len(snr),
                                      snr.ctypes.data, int(frames_per_snr), int(seed), int(first_frame),
                                      cnt.ctypes.data, pf.ctypes.data if per_frame else None))
        return cnt, pf
