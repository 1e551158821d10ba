"""Mirror of the Python entry points of the reference's ``ldpc.py`` on top of
the HIP engine.  Same names, argument order, return values and assertion
behaviour; the arithmetic of every decode is done on the GPU by libldpc_hip.so
in float64 (the reference's LDPC_DECIMAL_DATA_TYPE, ldpc.py:29), which makes
hard decisions, soft outputs and iteration counts bit-identical to the
reference's for a freshly constructed decoder.

The reference's decoder object is stateful -- decoderSet only resets variable
nodes (ldpc.py:270-274), so check-node messages of the previous frame leak into
the first iteration of the next one.  ldpcDecoder, evaluateCode (one decoder for
all frames, ldpc.py:356,377) and evaluateCodeAtSingleTransmission (one decoder
for all SNR points, ldpc.py:397,424) reproduce that through the library's
chained decode; pass freshDecoderPerFrame=True to evaluateCode for independent
frames decoded in parallel.

Deliberate difference: testCodeUsingMultiprocessing runs all transmissions of
an SNR point as one GPU batch (each with a fresh decoder, as each pool task of
the reference builds its own, ldpc.py:397) instead of a 2-process pool
(ldpc.py:454-458).
"""
import copy
import hashlib
import os
import threading
import time

import numpy as np

from . import codes, common
from .engine import Decoder

projectDir = os.environ.get('SWIFT')          # ldpc.py:11
LDPC_LOCAL_PRNG = np.random.RandomState(7134066)   # ldpc.py:24
LDPC_DATA_TYPE = np.int64
LDPC_INT_DATA_TYPE = np.int64
LDPC_DECIMAL_DATA_TYPE = np.float64
LDPC_SEED_DATA_TYPE = np.int64
LDPC_MAX_SEED = 2 ** 31 - 1
LDPC_HIP_DEVICE = int(os.environ.get('LDPC_HIP_DEVICE', '0'))


def slicer(vector, length):
    """Hard decision: 1 where vector > 0, else 0 (ldpc.py:43-48)."""
    out = np.ones(length, dtype=LDPC_INT_DATA_TYPE)
    out[np.where(np.asarray(vector) <= 0)] = 0
    return out


def addAWGN(vector, length, SNRdb, prng):
    """vector + N(0, sigma), sigma = sqrt(0.5 / 10^(SNRdb/10)); returns
    (noisy, sigma, sigmaActual) (ldpc.py:51-60)."""
    SNR = 10 ** (SNRdb / 10)
    sigma = np.sqrt(0.5 / SNR)
    noise = prng.normal(0, sigma, length)
    sigmaActual = np.sqrt((np.sum(noise ** 2)) / length)
    return vector + noise, sigma, sigmaActual


def modulate(vector, length):
    """BPSK: 0 -> -1, anything else -> +1 (ldpc.py:62-66)."""
    out = np.ones(length, dtype=LDPC_DECIMAL_DATA_TYPE)
    out[np.where(np.asarray(vector) == 0)] = -1
    return out


# ------------------------------------------------------------------ decoder cache
_DECODERS = {}              # key -> [engine, users]; insertion order = least recently used first
_DECODERS_MAX = 16
_DECODERS_LOCK = threading.Lock()


def _code_key(H):
    """(what to build the engine from, cache key).  A dense matrix is first read as block-circulant
    (codes.detect_qc verifies every entry): its shift lists are the key, a few hundred integers
    instead of a hash over the 8.4 M entries the gym hands over at every step (ldpc_env.py:313, 370).
    Only a matrix without that structure is hashed."""
    if isinstance(H, codes.QCCode):
        return H, H.key()
    Ha = np.asarray(H)
    qc = codes.detect_qc(Ha)
    if qc is not None:
        return qc, qc.key()
    bits = np.ascontiguousarray(Ha != 0)
    return Ha, ('dense', bits.shape, hashlib.sha1(np.packbits(bits).tobytes()).hexdigest())


def _evict_idle():
    """Close least recently used engines nobody holds while the cache is over its size (lock held)."""
    if len(_DECODERS) > _DECODERS_MAX:
        for k in [k for k, e in _DECODERS.items() if e[1] == 0][:len(_DECODERS) - _DECODERS_MAX]:
            _DECODERS.pop(k)[0].close()


class _Lease:
    """An engine of the cache, held for the duration of a `with` block.  An engine in use is never
    closed: eviction skips it (the cache may exceed its size for a while), and an engine that was
    dropped from the cache while in use is closed by its last user."""

    def __init__(self, H, device=None, physical_device=None):
        device = LDPC_HIP_DEVICE if device is None else device
        physical_device = device if physical_device is None else physical_device
        code, key = _code_key(H)
        key = (device,) + tuple(key)
        with _DECODERS_LOCK:
            entry = _DECODERS.pop(key, None)
            if entry is None:
                entry = [Decoder(code, device=physical_device), 0]
            _DECODERS[key] = entry                       # most recently used last
            entry[1] += 1
            _evict_idle()
        self._key, self._entry = key, entry
        self.engine = entry[0]

    def __enter__(self):
        return self.engine

    def __exit__(self, *exc):
        with _DECODERS_LOCK:
            self._entry[1] -= 1
            if self._entry[1] == 0 and _DECODERS.get(self._key) is not self._entry:
                self._entry[0].close()                   # dropped from the cache meanwhile: the last user closes it
            _evict_idle()
        return False


def _engine_lease(H, device=None, physical_device=None):
    """`with _engine_lease(H, d) as eng:` -- HIP decoder for H (dense array or codes.QCCode) from a
    cache keyed by content.  `device` is the caller's (logical) device number and part of the key:
    a handle serves one host thread at a time, so two logical devices never share one even when they
    map to the same physical device (`physical_device`, default = `device`)."""
    return _Lease(H, device, physical_device)


class ldpcDecoder:
    """Mirror of ldpc.ldpcDecoder (ldpc.py:223-337).

    Like the reference object it is STATEFUL: decoderSet resets only the variable
    nodes (ldpc.py:270-274), so the check nodes start a frame with the messages
    they sent in the last completed iteration of the previous frame decoded by
    this object.  Each ldpcDecoder therefore owns its own engine handle."""

    def __init__(self, H, device=None):
        self.parityMatrix = H
        if isinstance(H, codes.QCCode):
            m, n = H.m, H.n
        else:
            m, n = np.asarray(H).shape
        self.numberOfVariableNodes = n
        self.numberOfCheckNodes = m
        self.codewordLength = n
        self.softVector = np.zeros(n, dtype=LDPC_DECIMAL_DATA_TYPE)
        self._engine = Decoder(H, device=LDPC_HIP_DEVICE if device is None else device)

    def isCodeword(self, modulatedVector):
        """('Codeword' | 'Not a codeword', hard decision int64[n]) (ldpc.py:249-268).
        Evaluated by the decode kernel with an iteration budget of 0 (no state is touched)."""
        r = self._engine.decode(np.asarray(modulatedVector, dtype=LDPC_DECIMAL_DATA_TYPE), 0, prec=64)
        status = 'Codeword' if r['converged'][0] else 'Not a codeword'
        return status, r['hard'][0].astype(LDPC_INT_DATA_TYPE)

    def decoderMainLoop(self, fromChannel, maxNumberOfIterations):
        """-> (status, binaryVector int64[n], softVector float64[n], iterations) (ldpc.py:326-337)"""
        y = np.asarray(fromChannel, dtype=LDPC_DECIMAL_DATA_TYPE)
        r = self._engine.decode(y, maxNumberOfIterations, prec=64, want_soft=True, chained=True)
        it = int(r['iters'][0])
        status = 'Codeword' if r['converged'][0] else 'Not a codeword'
        # zero iterations: the reference returns a copy of its input (ldpc.py:328)
        soft = np.copy(y) if it == 0 else r['soft'][0]
        self.softVector = soft
        return status, r['hard'][0].astype(LDPC_INT_DATA_TYPE), soft, it

    def decodeBatch(self, fromChannel, maxNumberOfIterations, wantSoft=False, chained=True):
        """Many frames ([F, n]) in one call; not in the reference.  chained=True decodes
        them in order on this object's state (= calling decoderMainLoop F times);
        chained=False gives every frame a fresh decoder and runs them in parallel.
        -> engine result dict."""
        return self._engine.decode(np.asarray(fromChannel, dtype=LDPC_DECIMAL_DATA_TYPE),
                                   maxNumberOfIterations, prec=64, want_soft=wantSoft, chained=chained)


def testModulationAndSlicingRoundTrip():
    vectorLength = 100
    v = np.random.randint(0, 1, vectorLength, dtype=LDPC_DATA_TYPE)
    assert np.all(v == slicer(modulate(v, vectorLength), vectorLength))
    return 'OK'


def _status(flag):
    return 'Codeword' if flag else 'Not a codeword'


def evaluateCode(numberOfTransmissions, seed, SNRpoints, messageSize, codewordSize, numberOfIterations, H,
                 G='None', freshDecoderPerFrame=False):
    """SNR x transmissions sweep of the all-zero codeword (ldpc.py:350-388).

    As in the reference the noise comes from the module-global LDPC_LOCAL_PRNG
    (the seed argument only has to be positive, ldpc.py:353-355, 372), in
    float64, and berUncoded is reported as 0 (ldpc.py:374).  All frames are
    drawn first, in the reference's order, then decoded in that order on one
    stateful decoder (or independently, in parallel, with freshDecoderPerFrame)."""
    assert (seed > 0)
    assert hasattr(SNRpoints, "__len__")
    decoder = ldpcDecoder(H)
    berStats = common.berStatistics()
    codeword = np.zeros(codewordSize, dtype=LDPC_INT_DATA_TYPE)
    modulatedCodeword = modulate(codeword, codewordSize)
    frames, meta = [], []
    for SNR in SNRpoints:
        for _ in range(numberOfTransmissions):
            dirty, sigma, sigmaActual = addAWGN(modulatedCodeword, codewordSize, SNR, LDPC_LOCAL_PRNG)
            frames.append(dirty)
            meta.append((SNR, sigma, sigmaActual))
    if frames:
        start = time.time()
        r = decoder.decodeBatch(np.stack(frames), numberOfIterations, chained=not freshDecoderPerFrame)
        timeTotal = time.time() - start
        for k, (SNR, sigma, sigmaActual) in enumerate(meta):
            berDecoded = int(np.count_nonzero(r['hard'][k] != codeword))
            berStats.addEntry(SNR, sigma, sigmaActual, 0, berDecoded, int(r['iters'][k]),
                              numberOfIterations, _status(r['converged'][k]))
        print("Time it took the decoder:")
        print(timeTotal)
        print("And the throughput is:")
        print(len(frames) * codewordSize / max(timeTotal, 1e-12))
    return berStats


def _single_transmission_frames(localPrng, SNRpoints, messageSize, codewordSize, G):
    """Frames of evaluateCodeAtSingleTransmission (ldpc.py:402-422): the AWGN sample
    is drawn (it advances LDPC_LOCAL_PRNG and supplies sigmaActual) but the frame
    that is decoded is the clean codeword with its first symbol negated
    (ldpc.py:416-418)."""
    frames, meta = [], []
    for SNR in SNRpoints:
        if isinstance(G, str) and G == 'None':
            codeword = np.zeros(codewordSize, dtype=LDPC_INT_DATA_TYPE)
        else:
            message = localPrng.randint(0, 2, messageSize, dtype=LDPC_INT_DATA_TYPE)
            codeword = G.dot(message) % 2
        modulatedCodeword = modulate(codeword, codewordSize)
        _, sigma, sigmaActual = addAWGN(modulatedCodeword, codewordSize, SNR, LDPC_LOCAL_PRNG)
        dirty = copy.copy(modulatedCodeword)
        dirty[0] = dirty[0] * -1
        berUncoded = int(np.count_nonzero(slicer(dirty, codewordSize) != codeword))
        frames.append(dirty)
        meta.append((SNR, sigma, sigmaActual, berUncoded, codeword))
    return frames, meta


def evaluateCodeAtSingleTransmission(seed, SNRpoints, messageSize, codewordSize, numberOfIterations, H,
                                     G='None'):
    """One frame per SNR point (ldpc.py:391-430)."""
    assert (seed > 0)
    assert hasattr(SNRpoints, "__len__")
    localPrng = np.random.RandomState(seed)
    decoder = ldpcDecoder(H)
    berStats = common.berStatistics()
    frames, meta = _single_transmission_frames(localPrng, SNRpoints, messageSize, codewordSize, G)
    if frames:
        r = decoder.decodeBatch(np.stack(frames), numberOfIterations)
        for k, (SNR, sigma, sigmaActual, berUncoded, codeword) in enumerate(meta):
            berDecoded = int(np.count_nonzero(r['hard'][k] != codeword))
            berStats.addEntry(SNR, sigma, sigmaActual, berUncoded, berDecoded, int(r['iters'][k]),
                              numberOfIterations, _status(r['converged'][k]))
    return berStats


def constantFunction(const):
    def g(x):
        return const
    return g


def testCodeUsingMultiprocessing(seed, SNRpoints, messageSize, codewordSize, numberOfIterations,
                                 numberOfTransmissions, H, method=None, reference=None, G='None'):
    """numberOfTransmissions single-transmission evaluations per SNR point, merged
    with berStatistics.union (ldpc.py:438-476).  The reference maps them over a
    process pool; here they form one GPU batch per SNR point.  With `method`
    set the sweep stops at the first point whose BER at the lowest evaluated SNR
    exceeds `reference` (ldpc.py:465-475)."""
    bStats = common.berStatistics()
    seeds = LDPC_LOCAL_PRNG.randint(0, LDPC_MAX_SEED, numberOfTransmissions, dtype=LDPC_SEED_DATA_TYPE)
    decoder = ldpcDecoder(H)
    for s in SNRpoints:
        start = time.time()
        frames, meta = [], []
        for sd in seeds:
            assert (sd > 0)
            f, mt = _single_transmission_frames(np.random.RandomState(sd), [s], messageSize, codewordSize, G)
            frames += f
            meta += mt
        if frames:
            r = decoder.decodeBatch(np.stack(frames), numberOfIterations, chained=False)
            for k, (SNR, sigma, sigmaActual, berUncoded, codeword) in enumerate(meta):
                one = common.berStatistics()
                one.addEntry(SNR, sigma, sigmaActual, berUncoded,
                             int(np.count_nonzero(r['hard'][k] != codeword)), int(r['iters'][k]),
                             numberOfIterations, _status(r['converged'][k]))
                bStats = bStats.union(one)
        if method is None:
            end = time.time()
            print(" Time it took the decoder at snr " + str(s) + " is:")
            print(end - start)
            print("And the throughput is: ")
            print(numberOfTransmissions * codewordSize / max(end - start, 1e-12))
        else:
            snrAxis, averageSnrAxis, berData, averageNumberOfIterations = bStats.getStats(messageSize)
            print(berData)
            if berData[0] > reference:
                print("*** Evaluation failed at " + str(len(bStats.snrAxis)))
                break
    return bStats


def testNearEarth(numOfTransmissions=50):
    """ldpc.py:480-501: NearEarth, SNR 3.0..3.6 dB, 50 iterations, seed 460101."""
    print("*** in test near earth")
    nearEarthParity = codes.near_earth()
    roi = [3.0, 3.2, 3.4, 3.6]
    codewordSize = 8176
    messageSize = 7154
    numOfIterations = 50
    start = time.time()
    bStats = testCodeUsingMultiprocessing(460101, roi, messageSize, codewordSize, numOfIterations,
                                          numOfTransmissions, nearEarthParity)
    end = time.time()
    print('****Time it took for code evaluation == %d' % (end - start))
    print('****Throughput == ' + str((8176 * len(roi) * numOfTransmissions) / max(end - start, 1e-12))
          + 'bits per second.')
    return bStats


def testWifi(numOfTransmissions=50):
    """ldpc.py:504-525, the 802.11n n = 1944 rate 5/6 code over the same SNR points.  The reference marks
    it decommissioned: its evaluation call is commented out, so it stops at `return bStats` with a
    NameError; here the call its NearEarth twin makes (ldpc.py:495) is made for the WiFi matrix."""
    print("*** in test wifi")
    wifiParity = codes.wifi_1944_r56()
    roi = [3.0, 3.2, 3.4, 3.6]
    codewordSize = 1944
    messageSize = 1620
    numOfIterations = 50
    start = time.time()
    bStats = testCodeUsingMultiprocessing(460101, roi, messageSize, codewordSize, numOfIterations,
                                          numOfTransmissions, wifiParity)
    end = time.time()
    print('Time it took for code evaluation == %d' % (end - start))
    print('Throughput == ' + str((codewordSize * len(roi) * numOfTransmissions) / max(end - start, 1e-12))
          + 'bits per second.')
    return bStats


def main():
    print("*** In ldpc.py main function.")
    return testNearEarth()


if __name__ == '__main__':
    main()
