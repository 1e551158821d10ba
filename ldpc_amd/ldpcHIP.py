"""GPU-flavoured evaluation entry points: the names the gym environment and the
post-processing scripts of the reference import from ``ldpcCUDA``
(ldpc_env.py:370, postProcessing.py:47, test_ldpc.py:25-28), served by
libldpc_hip.so.  Drop-in use:   ``import ldpc_amd.ldpcHIP as ldpcCUDA``.

Arithmetic is float32 like the reference's numba.cuda kernels
(ldpcCUDA.py:538-548); pass prec=64 for the CPU decoder's float64 arithmetic.

Differences from ldpcCUDA.evaluateCodeCuda that are intended:
  * the stop test runs after every iteration (the reference reads its flag
    back only when iterator % 6 == 0 and its flag kernel covers half the rows,
    ldpcCUDA.py:465-482, 637-639), so iteration counts are the exact ones
    ldpc.py reports, not {0, 1, 7, 13, ...};
  * all devices are driven from this process with one host thread per device
    instead of one forked process per device (ldpcCUDA.py:901-906).
"""
import concurrent.futures
import time

import numpy as np

from . import codes, common
from .engine import device_count
from . import ldpc as _ldpc


def _noise_frames(localPrng, SNRpoints, numOfTransmissions, n):
    """Channel outputs exactly as the reference draws them: one float32 noise
    vector per frame from RandomState(seed), SNR point after SNR point
    (ldpcCUDA.py:509-517, 576-587)."""
    Y = np.empty((len(SNRpoints) * numOfTransmissions, n), dtype=np.float32)
    meta = []
    minus_one = np.float32(-1)
    k = 0
    for s in range(len(SNRpoints)):
        SNR = 10 ** (SNRpoints[s] / 10)
        sigma = np.sqrt(0.5 / SNR)
        for _ in range(numOfTransmissions):
            noise = np.float32(localPrng.normal(0, sigma, n))
            sigmaActual = np.sqrt((np.sum(noise ** 2)) / n)
            Y[k] = minus_one + noise
            meta.append((SNRpoints[s], sigma, sigmaActual, int(np.count_nonzero(Y[k] > 0))))
            k += 1
    return Y, meta


def evaluateCodeCuda(seed, SNRpoints, numberOfIterations, parityMatrix, numOfTransmissions, G='None',
                     cudaDeviceNumber=0, prec=32, deviceRng=False):
    """Evaluate a parity-check matrix on one device (ldpcCUDA.py:40-673).

    Returns common.berStatistics with one row per frame:
    (SNR, sigma, sigmaActual, berUncoded, berDecoded, iterations, numberOfIterations, 'test')
    -- the status column is the literal 'test' as in the reference (ldpcCUDA.py:664).

    deviceRng=False: frames are drawn on the host from RandomState(seed) exactly as
    the reference does.  deviceRng=True: frames are generated on the device from a
    counter-based generator keyed by `seed` (same statistics, different samples,
    no host work per frame)."""
    assert (seed > 0)
    assert hasattr(SNRpoints, "__len__")
    n_dev = max(device_count(), 1)
    with _ldpc._engine_lease(parityMatrix, cudaDeviceNumber, cudaDeviceNumber % n_dev) as eng:
        return _evaluate_on(eng, seed, SNRpoints, numberOfIterations, numOfTransmissions, prec, deviceRng)


def _evaluate_on(eng, seed, SNRpoints, numberOfIterations, numOfTransmissions, prec, deviceRng, budget=None):
    """budget: iterations the decoder may run when that is not the number reported in the rows
    (wifiCUDA.py:535, 579, 646)."""
    n = eng.n
    budget = numberOfIterations if budget is None else budget
    berStats = common.berStatistics()
    if deviceRng:
        cnt, pf = eng.eval_awgn(list(SNRpoints), numOfTransmissions, int(seed), max_iters=budget,
                                prec=prec, per_frame=True)
        frames = pf.tolist()                    # Python floats at once (indexing the array element by element is slower)
        for s in range(len(SNRpoints)):
            sigma = np.sqrt(0.5 / (10 ** (SNRpoints[s] / 10)))
            for sa, unc, dec, it in frames[s]:
                berStats.addEntry(SNRpoints[s], sigma, sa, int(unc), int(dec), int(it),
                                  numberOfIterations, 'test')
        return berStats
    localPrng = np.random.RandomState(seed)
    Y, meta = _noise_frames(localPrng, SNRpoints, numOfTransmissions, n)
    if len(meta):
        r = eng.decode(Y, budget, prec=prec, want_hard=False)
        for k, (snr, sigma, sigmaActual, berUncoded) in enumerate(meta):
            berStats.addEntry(snr, sigma, sigmaActual, berUncoded, int(r['ones'][k]), int(r['iters'][k]),
                              numberOfIterations, 'test')
    return berStats


def evaluateMatrixAndEpsilon(parityMatrix, epsilon, numberOfIterations=50, cudaDeviceNumber=0, prec=32):
    """One frame with the caller's noise vector (ldpcCUDA.py:677-828): the all-zero codeword,
    modulated to -1 (float32, ldpcCUDA.py:504-507), plus `epsilon`; returns
    (berUncoded, berDecoded, iterator, totalTime) -- channel hard-decision errors, errors left after
    decoding, iterations run, seconds spent decoding.  `iterator` is the exact count, as everywhere
    in this module (the reference looks at its stop flag every sixth iteration, ldpcCUDA.py:774-776)."""
    n_dev = max(device_count(), 1)
    with _ldpc._engine_lease(parityMatrix, cudaDeviceNumber, cudaDeviceNumber % n_dev) as eng:
        fromChannel = np.full(eng.n, -1, dtype=np.float32) + np.asarray(epsilon).reshape(eng.n)
        berUncoded = int(np.count_nonzero(fromChannel > 0))          # slicer, ldpcCUDA.py:497-501
        start = time.time()
        r = eng.decode(np.asarray(fromChannel, dtype=np.float64 if prec == 64 else np.float32).reshape(1, eng.n),
                       numberOfIterations, prec=prec, want_hard=False)
        totalTime = time.time() - start
    return berUncoded, int(r['ones'][0]), int(r['iters'][0]), totalTime


def evaluateCodeCudaWrapper(seeds, SNRpoints, numberOfIterations, parityMatrix, numOfTransmissions, G='None',
                            numberOfCudaDevices=4, prec=32, deviceRng=False):
    """numOfTransmissions // numberOfCudaDevices frames per device, device d seeded with
    seeds[d], rows concatenated with berStatistics.add (ldpcCUDA.py:891-932).  Logical
    devices beyond the visible ones wrap around (device d -> d mod visible)."""
    berStats = common.berStatistics()
    newNumOfTransmissions = numOfTransmissions // numberOfCudaDevices
    parityMatrix, _ = _ldpc._code_key(parityMatrix)       # the structure of a dense matrix is read once, not per device
    from .engine import device_count
    if numberOfCudaDevices <= 1 or device_count() <= 1:
        # one card serves every logical device in turn: starting a thread per call costs a fifth of a gym step
        for d in range(numberOfCudaDevices):
            berStats = berStats.add(evaluateCodeCuda(seeds[d], SNRpoints, numberOfIterations, parityMatrix,
                                                     newNumOfTransmissions, 'None', d, prec, deviceRng))
        return berStats
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, numberOfCudaDevices)) as pool:
        futs = [pool.submit(evaluateCodeCuda, seeds[d], SNRpoints, numberOfIterations, parityMatrix,
                            newNumOfTransmissions, 'None', d, prec, deviceRng)
                for d in range(numberOfCudaDevices)]
        for f in futs:
            berStats = berStats.add(f.result())
    return berStats


def testConcurrentFutures(numberOfCudaDevices=1, prec=32):
    """ldpcCUDA.testConcurrentFutures (ldpcCUDA.py:871-888): 15 NearEarth frames per SNR point on every
    device, device d seeded with the d-th draw of RandomState(7134066).randint(0, 2**31 - 1), rows
    concatenated in device order.  (As shipped the reference's function stops with a NameError: the
    generator and its bounds are locals of evaluateCodeCuda, ldpcCUDA.py:47-49; this is what it spells out.)"""
    nearEarthParity = codes.near_earth()
    numOfTransmissions = 15
    roi = [3.0, 3.2, 3.4, 3.6]
    numOfIterations = 50
    seeds = np.random.RandomState(7134066).randint(0, 2 ** 31 - 1, numberOfCudaDevices, dtype=np.int64)
    return evaluateCodeCudaWrapper(seeds, roi, numOfIterations, nearEarthParity,
                                   numOfTransmissions * numberOfCudaDevices, 'None', numberOfCudaDevices, prec=prec)


def testNearEarth(numOfTransmissions=60, graphics=False, prec=32):
    """ldpcCUDA.testNearEarth (ldpcCUDA.py:832-868): returns (berStatistics, status);
    status is 'OK' iff the BER at the two highest SNR points is 0."""
    status = 'Near earth problem'
    print("*** in test near earth")
    nearEarthParity = codes.near_earth()
    roi = [3.0, 3.2, 3.4, 3.6]
    codewordSize = 8176
    numOfIterations = 50
    start = time.time()
    bStats = evaluateCodeCuda(460101, roi, numOfIterations, nearEarthParity, numOfTransmissions, prec=prec)
    end = time.time()
    print('Time it took for code evaluation == %d' % (end - start))
    print('Throughput == ' + str((8176 * len(roi) * numOfTransmissions) / max(end - start, 1e-12))
          + 'bits per second.')
    a, b, c, d = bStats.getStats(codewordSize)
    scatterSnr, scatterBer, scatterItr, snrAxis, averageSnrAxis, berData, avgItr = bStats.getStatsV2()
    p1 = np.poly1d(np.polyfit(scatterSnr, scatterBer, 1))
    pTotalInteg = (np.poly1d([1]) - p1).integ()
    reward = pTotalInteg(roi[-1]) - pTotalInteg(roi[0])
    print(reward)
    print("berDecoded " + str(c))
    if (c[-1] == 0) and (c[-2] == 0):
        status = 'OK'
    return bStats, status
