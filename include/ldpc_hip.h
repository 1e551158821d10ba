/*
 * ldpc_hip.h -- C ABI of libldpc_hip.so: LDPC code evaluation on AMD MI355X (gfx950).
 *
 * This is the drop-in boundary for the one data-parallel hot path of
 * Omer-Sella/ldpc: channel -> iterative decode -> syndrome early stop ->
 * error counting.  Every entry point names the reference interface it replaces
 * (paths relative to the reference checkout).  Plain C, ctypes/cffi loadable,
 * no C++ or torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative LDPC_E_* code; the text
 *     of the last error of the calling thread is in ldpc_last_error();
 *   - the caller owns every buffer it passes; the library owns only what hangs
 *     off an ldpc_handle;
 *   - one handle = one device + one stream + one code; a handle must not be
 *     used from two threads at once, distinct handles may be;
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with LDPC_E_DEVICE.
 *
 * Decoder contract (identical to the reference's, bit for bit, see
 * ldpc.py:326-337): channel values y (positive = bit 1, ldpc.py:43-48,62-66);
 * if H*(y>0) == 0 return after 0 iterations; otherwise at most max_iters
 * flooding iterations of plain min-sum (no scaling/offset, ldpc.py:174-202),
 * variable update = sum of check messages in ascending check index with the
 * channel value added last (ldpc.py:298-303, 322), stop as soon as the
 * syndrome of the posterior hard decision is zero (ldpc.py:332-335).
 */
#ifndef LDPC_HIP_H
#define LDPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ldpc_handle ldpc_handle;

enum {
    LDPC_OK = 0,
    LDPC_E_ARG = -1,      /* bad argument (NULL, size, range)                    */
    LDPC_E_DEVICE = -2,   /* no HIP device / HIP runtime error                   */
    LDPC_E_UNSUPPORTED = -3, /* code or option outside what the kernels cover    */
    LDPC_E_NOMEM = -4
};

enum { LDPC_ALGO_MINSUM = 0, LDPC_ALGO_SUMPRODUCT = 1 };
enum { LDPC_SIGN_REFERENCE = 0, LDPC_SIGN_CORRECTED = 1 };

/* Text of the last error raised on the calling thread ("" if none). */
const char *ldpc_last_error(void);

/* Number of visible HIP devices (0 if none / runtime unusable).
 * Replaces numba.cuda.gpus / cuda.select_device, ldpcCUDA.py:42-45. */
int ldpc_device_count(void);

/*
 * Build a decoder for a quasi-cyclic parity-check matrix made of mb x nb
 * circulant blocks of size Z.  Block (b, j), 0 <= b < mb, 0 <= j < nb, is a sum
 * of cyclic shifts: row i of the block has ones at columns (i + s) mod Z for
 * every s in shifts[blk_ptr[b*nb+j] .. blk_ptr[b*nb+j+1]) (distinct, any order,
 * possibly none).  This is the matrix fileHandler.readMatrixFromFile(...,
 * isGenerator=False) builds from a hot-position file (fileHandler.py:137-142,
 * 161-181) and wifiMatrices.getWifiParityMatrix builds from its shift table
 * (wifiMatrices.py:12-34).
 * Replaces ldpcDecoder.__init__ (ldpc.py:225-247) and the dense device upload
 * of ldpcCUDA.py:538-564.  Limits: mb*roundup(Z,64) <= 1024, nb <= 32*mb
 * (LDPC_E_UNSUPPORTED beyond them: build the same matrix with ldpc_create_csr).
 * Any check degree: up to 128 edges per check (the state of a long gym episode,
 * ldpc_env.py:293-317) the block-circulant kernels apply; beyond that, or when a
 * frame's state does not fit the 160 KB of LDS in the arithmetic asked for, the code
 * is decoded from its edge list instead (slower, same results).
 */
int ldpc_create_qc(int device, int Z, int mb, int nb, const int32_t *blk_ptr,
                   const int32_t *shifts, ldpc_handle **out);

/*
 * Host helper: the shift lists (ldpc_create_qc's blk_ptr / shifts) of a dense row-major matrix whose
 * entries are elem_bytes = 1, 4 or 8 bytes wide (any non-zero pattern counts as 1; a float64 matrix of
 * 0.0 / 1.0 qualifies), if it is block-circulant with circulants of size Z.  The reference hands its
 * decoders the dense matrix (ldpc.py:225-247, ldpc_env.py:370).  Returns 0 and fills blk_ptr
 * [(m/Z)*(n/Z)+1] and shifts [<= shifts_cap] (ascending inside a block), 1 if the matrix is not
 * block-circulant with this Z, a negative code for bad arguments or too small a shifts array.
 * No device is touched.
 */
int ldpc_dense_to_qc(const void *H, int elem_bytes, int m, int n, int Z, int32_t *blk_ptr,
                     int32_t *shifts, int shifts_cap);

/*
 * Build a decoder for an arbitrary sparse H given as CSR (row_ptr[m+1],
 * col_idx ascending inside a row).  Replaces ldpcDecoder.__init__
 * (ldpc.py:225-247) for matrices that are not block-circulant.  Any check
 * degree (above 64 edges per check: one explicit message per edge in global
 * memory, the slowest path).
 */
int ldpc_create_csr(int device, int m, int n, const int32_t *row_ptr,
                    const int32_t *col_idx, ldpc_handle **out);

void ldpc_destroy(ldpc_handle *h);

/* A destroyed handle leaves its stream, its table allocation and its small buffers (at most 8 sets per device)
 * for the next handle on the same device: the gym builds a new decoder at every step (ldpc_env.py:313, 370).
 * A long-running host that is done decoding gives them back with this call; returns the number of sets freed.
 * No counterpart in the reference (numba keeps its device arrays until the process ends, ldpcCUDA.py:538-564). */
int ldpc_release_pool(void);

/* Code dimensions of a handle. */
int ldpc_get_dims(const ldpc_handle *h, int *m, int *n, int *edges);

/* Min-sum magnitude correction for every later min-sum decode on this handle:
 * |check-to-variable message| = max(alpha * min - beta, 0) (normalised and/or offset min-sum).
 * The default alpha = 1, beta = 0 is the reference's plain rule (ldpc.py:174-202) and leaves
 * every result bit-identical; the reference has no counterpart for other values (an optional
 * decoder variant).  Needs alpha > 0, beta >= 0. */
int ldpc_set_minsum_scaling(ldpc_handle *h, float alpha, float beta);

/* Measurement aid, no counterpart in the reference: d_counts points to two uint64 in device memory
 * (or NULL to switch off, the default).  Every later float32 decode of the message-plane kernels adds
 * to them the number of full check passes (d_counts[0]) and of syndrome-only passes (d_counts[1]) its
 * workgroups ran -- what the on-chip roofline of bench.py is computed from.  The caller zeroes them. */
int ldpc_set_pass_counters(ldpc_handle *h, uint64_t *d_counts);

/* Make the handle launch on an existing HIP stream: hip_stream is a hipStream_t
 * passed as void* (NULL is the legacy default stream, which is a valid choice);
 * use_own != 0 ignores hip_stream and goes back to the stream the handle
 * created for itself.  A change of stream waits for what the handle has queued
 * on the old one (its scratch buffers are shared between launches).  No counterpart in the reference (numba's default
 * stream, ldpcCUDA.py:594-662). */
int ldpc_set_stream(ldpc_handle *h, void *hip_stream, int use_own);

/* Block until everything queued on the handle's stream has finished. */
int ldpc_sync(ldpc_handle *h);

/*
 * Decode n_frames frames held in HOST memory.
 *   llr        [n_frames][n] channel values (row-major), float32 (llr_bits 32:
 *              what ldpcCUDA.addAWGN produces, ldpcCUDA.py:514) or float64
 *              (llr_bits 64: what ldpc.addAWGN produces, ldpc.py:57-59)
 *   hard_bits  [n_frames][ceil(n/8)] or NULL; bit c of a frame is
 *              (byte c>>3) >> (c&7) & 1  (numpy.packbits bitorder='little')
 *   iters      [n_frames] completed iterations (0 if y is already a codeword)
 *   converged  [n_frames] 1 = 'Codeword', 0 = 'Not a codeword'
 *   ones       [n_frames] or NULL: number of 1 bits in the hard decision
 *              (= bit errors against the all-zero codeword, ldpc.py:381)
 *   soft_out   [n_frames][n] float32 (prec 32) or float64 (prec 64), or NULL:
 *              the posterior vector decoderMainLoop returns (ldpc.py:337)
 * prec: 32 = float32 arithmetic (the reference's GPU path, ldpcCUDA.py:538-548),
 *       64 = float64 arithmetic (the reference's CPU path, ldpc.py:29).
 *       The rule has no normalisation: on a code full of 4-cycles (a circulant of weight 9, say)
 *       a frame that never converges grows its messages by up to (column degree - 1) per iteration
 *       and can leave float32 within 50 iterations; its soft values and hard decisions are then
 *       infinities / NaNs as in any float32 implementation of the rule (iteration count and
 *       status stay meaningful).  float64 has the range the reference relies on.
 * Synchronous.  Replaces ldpcDecoder.decoderMainLoop (ldpc.py:326-337) and the
 * per-frame body of ldpcCUDA.evaluateCodeCuda (ldpcCUDA.py:583-662).
 */
int ldpc_decode(ldpc_handle *h, int algo, int prec, int sign_mode, int max_iters,
                int64_t n_frames, const void *llr, int llr_bits, uint8_t *hard_bits,
                int32_t *iters, uint8_t *converged, int32_t *ones, void *soft_out);

/*
 * Stateful decoder: same arguments and outputs as ldpc_decode, but the frames
 * are decoded ONE AFTER ANOTHER on a decoder whose check-node messages are not
 * cleared between frames -- the first iteration of a frame subtracts the
 * messages the checks sent in the last completed iteration of the previous
 * frame.  This is what re-using one reference ldpcDecoder object does
 * (decoderSet resets only the variable nodes, ldpc.py:270-274; one decoder for
 * all frames in evaluateCode, ldpc.py:356,377, and for all SNR points in
 * evaluateCodeAtSingleTransmission, ldpc.py:397,424).  The state lives in the
 * handle (one per prec) and survives across calls; ldpc_reset_state returns
 * the handle to a freshly constructed decoder.  One workgroup, sequential:
 * meant for exact reproduction of the reference's CPU evaluation, not for
 * throughput.
 */
int ldpc_decode_chained(ldpc_handle *h, int algo, int prec, int sign_mode, int max_iters,
                        int64_t n_frames, const void *llr, int llr_bits, uint8_t *hard_bits,
                        int32_t *iters, uint8_t *converged, int32_t *ones, void *soft_out);

/* Forget the chained decoder's check-node messages (= a new ldpcDecoder(H), ldpc.py:114-133). */
int ldpc_reset_state(ldpc_handle *h);

/*
 * Same as ldpc_decode, with every buffer in DEVICE memory of the handle's device; queued on
 * the handle's stream, returns without waiting.  llr_scale multiplies the
 * channel values on load (1.0 for min-sum parity with the reference, whose
 * "LLR" is the raw channel value; 2/sigma^2 for sum-product).
 */
int ldpc_decode_device(ldpc_handle *h, int algo, int prec, int sign_mode, int max_iters,
                       int64_t n_frames, const void *d_llr, int llr_bits, float llr_scale,
                       uint8_t *d_hard_bits, int32_t *d_iters, uint8_t *d_converged,
                       int32_t *d_ones, void *d_soft_out);

/*
 * ldpc_decode_device for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arguments, outputs and results as ldpc_decode_device; the scratch memory
 * the kernels need -- the counter workgroups claim their frames from, and the per-workgroup
 * workspace of the edge-list kernels -- lives in a static device array of
 * LDPC_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends,
 * so the call reaches no hipMalloc / hipFree.  A code whose kernel needs a workspace gets as
 * many workgroups as fit the array (results do not depend on that; it only runs slower).
 * Fails with LDPC_E_UNSUPPORTED if not even one workspace fits, or if n_frames is beyond
 * 2^31 - 1 - 4 * (workgroups launched), or if the decode needs the handle's edge list and that
 * is not on the device yet (ldpc_create_qc defers it for codes of 2 x 16 circulants of size 511,
 * which need it only for float64 sum-product and the table-driven kernel: run such a decode once
 * through ldpc_decode_device first).  All calls on one GPU share the array: two of them
 * that overlap in time (different handles or streams on the same GPU) race; calls on one
 * stream, or separated by a synchronisation, are safe.
 */
#define LDPC_NOALLOC_SCRATCH_BYTES (4u << 20)
int ldpc_decode_device_noalloc(ldpc_handle *h, int algo, int prec, int sign_mode, int max_iters,
                               int64_t n_frames, const void *d_llr, int llr_bits, float llr_scale,
                               uint8_t *d_hard_bits, int32_t *d_iters, uint8_t *d_converged,
                               int32_t *d_ones, void *d_soft_out);

/*
 * Fill d_llr [n_frames][n] (device) with BPSK/AWGN channel outputs of the
 * all-zero codeword: y = -1 + sigma*N(0,1), sigma = sqrt(0.5 / 10^(snr_db/10))
 * (ldpc.py:51-66; no code-rate term).  Noise is counter based: sample (frame
 * first_frame+f, position c) depends only on (seed, first_frame+f, c), so any
 * sharding over devices sees the same frames.  Optional per-frame outputs:
 *   d_sigma_hat [n_frames] float: sqrt(sum(noise^2)/n)  (sigmaActual, ldpc.py:58)
 *   d_uncoded   [n_frames] int32: channel hard-decision errors (ldpcCUDA.py:586-587)
 * Queued on the handle's stream.  Replaces addAWGN/modulate/slicer on the
 * host (ldpcCUDA.py:497-517, 581-587).
 */
int ldpc_channel_awgn_device(ldpc_handle *h, double snr_db, uint64_t seed, uint64_t first_frame,
                             int64_t n_frames, float *d_llr, float *d_sigma_hat,
                             int32_t *d_uncoded);

/*
 * Monte-Carlo evaluation: for every SNR point generate frames_per_snr frames
 * on the device (ldpc_channel_awgn_device, frame ids first_frame ..
 * first_frame+frames_per_snr-1, stream id = index of the SNR point), decode
 * them and count.  Host outputs:
 *   counters  [n_snr][6] uint64: frames, frame errors (hard decision != 0),
 *             decoded bit errors, sum of iterations, uncoded bit errors,
 *             frames with status 'Not a codeword'
 *   per_frame [n_snr][frames_per_snr][4] float or NULL: sigma_hat, uncoded
 *             errors, decoded errors, iterations  (the columns of a
 *             berStatistics row, common.py:149-157)
 * A sweep of at most 8 points and 8192 frames in all runs as one launch, the
 * points' frames side by side (same noise and results as point after point).
 * Synchronous.  Replaces the SNR x transmissions loops of ldpc.evaluateCode
 * (ldpc.py:350-388) and ldpcCUDA.evaluateCodeCuda (ldpcCUDA.py:576-664).
 */
int ldpc_eval_awgn(ldpc_handle *h, int algo, int prec, int sign_mode, int max_iters,
                   int n_snr, const double *snr_db, int64_t frames_per_snr,
                   uint64_t seed, uint64_t first_frame, uint64_t *counters,
                   float *per_frame);

/*
 * The path's one exchange step for callers that have no collective of their own: the error
 * counters of the ranks' shares of a sweep, summed over RCCL (one process per GPU).  The
 * reference adds up its devices' berStatistics on the host after its process pool returns
 * (ldpcCUDA.py:897-912); a Python caller of this package uses torch.distributed instead.
 * RCCL is opened when the first of these calls is made (the copy a framework in the process
 * already loaded, else the system's); without it they return LDPC_E_UNSUPPORTED.
 *   ldpc_comm_unique_id   rank 0 fills id_out [LDPC_COMM_ID_BYTES] and hands the bytes to every
 *                         rank by the caller's own means (file, socket, launcher)
 *   ldpc_comm_create      collective over all `world` ranks: every rank calls it with the same id,
 *                         its own rank and the GPU it decodes on (one rank per GPU)
 *   ldpc_allreduce_counters  collective: counters [count] (host, e.g. ldpc_eval_awgn's
 *                         [n_snr][6]) are replaced by their sums over the ranks.  Synchronous.
 * A communicator is not thread-safe; distinct ones are.
 */
#define LDPC_COMM_ID_BYTES 128
typedef struct ldpc_comm ldpc_comm;
int ldpc_comm_unique_id(void *id_out);
int ldpc_comm_create(int device, int rank, int world, const void *id, ldpc_comm **out);
int ldpc_allreduce_counters(ldpc_comm *comm, uint64_t *counters, int count);
void ldpc_comm_destroy(ldpc_comm *comm);

/* Launch geometry the decoder uses for (algo, prec): workgroups, threads per
 * workgroup, dynamic LDS bytes, frames per workgroup.  For reports. */
int ldpc_get_launch_info(ldpc_handle *h, int algo, int prec, int *grid, int *block,
                         int *lds_bytes, int *frames_per_wg);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_H */
