"""The 802.11n flavour of the GPU evaluation module: the names the reference's ``wifiCUDA`` exposes
(wifiCUDA.py:496-683), served by the same engine as ``ldpcHIP``.  Drop-in use:
``import ldpc_amd.wifiHIP as wifiCUDA``.

The reference keeps a second copy of its kernels with the n = 1944 dimensions written in
(wifiCUDA.py:40-60); here the decoder takes its dimensions from the matrix, so the evaluation
function is the one of ``ldpcHIP`` (same host-seeded frames: RandomState(seed), one float32 noise
vector per frame, wifiCUDA.py:472-481, 503, 553).
"""
import time

from . import ldpcHIP, wifiMatrices
from . import ldpc as _ldpc
from .engine import device_count

MAX_NUMBER_OF_ITERATIONS = 40        # wifiCUDA.py:535


def evaluateCodeCuda(seed, SNRpoints, numberOfIterations, parityMatrix, numOfTransmissions, G='None',
                     cudaDeviceNumber=0, prec=32, deviceRng=False):
    """wifiCUDA.evaluateCodeCuda (wifiCUDA.py:496-657).  As there, the decoder runs at most 40 iterations
    whatever `numberOfIterations` says (wifiCUDA.py:535, 579); the argument is only written into the rows
    (wifiCUDA.py:646).  The device argument is this package's addition."""
    assert (seed > 0)
    assert hasattr(SNRpoints, "__len__")
    n_dev = max(device_count(), 1)
    with _ldpc._engine_lease(parityMatrix, cudaDeviceNumber, cudaDeviceNumber % n_dev) as eng:
        return ldpcHIP._evaluate_on(eng, seed, SNRpoints, numberOfIterations, numOfTransmissions, prec, deviceRng,
                                    budget=MAX_NUMBER_OF_ITERATIONS)


def testWifi(numOfTransmissions=50):
    """wifiCUDA.testWifi (wifiCUDA.py:660-682): the n = 1944 rate 5/6 code at 3.0 ... 3.6 dB, seed 460101;
    returns (berStatistics, status), status 'OK' iff the decoded BER at the two highest points is 0."""
    status = 'WIFI problem'
    print("*** in test near wifi")
    wifiParity = wifiMatrices.getWifiParityMatrix()
    roi = [3.0, 3.2, 3.4, 3.6]
    codewordSize = 1944
    numOfIterations = 50
    start = time.time()
    bStats = evaluateCodeCuda(460101, roi, numOfIterations, wifiParity, numOfTransmissions)
    end = time.time()
    print('Time it took for code evaluation == %d' % (end - start))
    print('Throughput == ' + str((codewordSize * len(roi) * numOfTransmissions) / max(end - start, 1e-12))
          + 'bits per second.')
    a, b, c, d = bStats.getStats(codewordSize)
    print("berDecoded " + str(c))
    if (c[2] == 0) and (c[3] == 0):
        status = 'OK'
    return bStats, status


def main():
    print("*** In wifiCUDA.py main function.")
    testWifi()
    return
