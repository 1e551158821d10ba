"""Result container of the code-evaluation path: a mirror of the reference's
``common.berStatistics`` (common.py:142-227) and of the line fit its callers
apply to the scatter data (``recursiveLinearFit``, common.py:293-303).

Row layout (common.py:155):
    [snr, snrDbActual, sigma, sigmaActual, berUncoded, berDecoded,
     numberOfDecoderIterations, maximalNumberOfIterations, wasDecoded]
where berUncoded / berDecoded are error COUNTS of one frame and
snrDbActual = 10*log10(1 / (2*sigmaActual^2)) (common.py:153-154).
"""
import numpy as np

DECIMAL = np.float32   # COMMON_DECIMAL_DATA_TYPE, common.py:13


class berStatistics:
    def __init__(self, codewordSize=8176):
        self.stats = []
        self.snrAxis = set()
        self.codewordSize = codewordSize
        self.evaluatedPoints = 0

    def addEntry(self, snr, sigma, sigmaActual, berUncoded, berDecoded, numberOfDecoderIterations,
                 maximalNumberOfIterations, wasDecoded):
        assert sigmaActual != 0
        snrDbActual = 10 * np.log10(1 / (2 * (sigmaActual ** 2)))
        self.stats.append([snr, snrDbActual, sigma, sigmaActual, berUncoded, berDecoded,
                           numberOfDecoderIterations, maximalNumberOfIterations, wasDecoded])
        self.snrAxis.add(snr)

    def getRawStats(self):
        return self.stats

    def _merged(self, rhs, rows):
        out = berStatistics()            # like the reference: codeword size is not carried over
        out.stats = rows
        out.snrAxis = self.snrAxis | rhs.snrAxis
        return out

    def union(self, rhs):
        """Rows of both, sorted (common.py:167-172)."""
        return self._merged(rhs, sorted(self.stats + rhs.stats))

    def add(self, rhs):
        """Rows of both, concatenated in order (common.py:174-180)."""
        return self._merged(rhs, self.stats + rhs.stats)

    def getStatsV2(self, codewordSize=None, fixAverageIterations=False):
        """-> scatterSNR, scatterBER, scatterITR, snrAxis, averageSnrAxis, berData,
        averageNumberOfIterations (all float32 except snrAxis), common.py:195-227.

        The reference never accumulates the per-point iteration sum (a statement
        without assignment, common.py:219), so its averageNumberOfIterations is
        all zeros; that is reproduced unless fixAverageIterations is set."""
        if codewordSize is None:
            codewordSize = self.codewordSize
        rows = self.stats
        snrAxis = np.array(sorted(self.snrAxis))
        p = len(snrAxis)
        # The scatter columns are the rows' columns rounded to float32.  The per-point sums are what the reference's
        # loop leaves in its float32 arrays (common.py:210-222): row after row, sum[at] = sum[at] + value, where
        # the actual SNR is a float64 scalar (added in float64, the sum rounded to float32 when it is stored) and
        # the error count an integer (added in float32).  Same arithmetic here on scalars, without a search of the
        # SNR axis per row (that search was most of a gym step's host time).
        scatterSNR = np.array([r[1] for r in rows], np.float64).astype(DECIMAL)
        scatterBER = (np.array([r[5] for r in rows], np.float64) / codewordSize).astype(DECIMAL)
        scatterITR = np.array([r[6] for r in rows], np.float64).astype(DECIMAL)
        where = {v: t for t, v in enumerate(snrAxis.tolist())}
        f32 = np.float32
        snr_acc = [0.0] * p                      # float32 values held as Python floats
        err_acc = [f32(0)] * p
        itr_acc = [f32(0)] * p
        cnt_acc = [f32(0)] * p
        one = f32(1)
        for row in rows:
            at = where.get(float(row[0]))
            if at is None:                       # (a point that is not on the axis counts nowhere, as in the reference)
                continue
            v = row[1]
            if isinstance(v, (np.floating, float)) and not isinstance(v, np.float32):
                snr_acc[at] = float(f32(snr_acc[at] + float(v)))
            else:
                snr_acc[at] = float(f32(snr_acc[at]) + f32(v))
            err_acc[at] = err_acc[at] + f32(row[5])
            if fixAverageIterations:
                itr_acc[at] = itr_acc[at] + f32(row[6])
            cnt_acc[at] = cnt_acc[at] + one
        snrSum = np.array(snr_acc, np.float64).astype(DECIMAL)
        errSum = np.array(err_acc, DECIMAL)
        itrSum = np.array(itr_acc, DECIMAL)
        count = np.array(cnt_acc, DECIMAL)
        with np.errstate(divide='ignore', invalid='ignore'):
            return (scatterSNR, scatterBER, scatterITR, snrAxis, snrSum / count,
                    errSum / (count * codewordSize), itrSum / count)

    def getStats(self, codewordSize=None):
        """Deprecated 4-tuple wrapper (common.py:162-165)."""
        _, _, _, snrAxis, averageSnrAxis, berData, avgItr = self.getStatsV2(codewordSize)
        return snrAxis, averageSnrAxis, berData, avgItr


def recursiveLinearFit(xData, yData, numberOfIterations=10, earlyStopping=False):
    """Repeated degree-1 fit that drops the points where the fitted line is not
    positive (common.py:293-303).  Returns (snr, ber, p, trendP, itr)."""
    ber = np.array(yData, copy=True)
    snr = np.array(xData, copy=True)
    itr = 0
    p = None
    trendP = None
    while itr < numberOfIterations:
        p = np.polyfit(snr, ber, 1)
        trendP = np.poly1d(p)
        keep = trendP(snr) > 0
        itr += 1
        if keep.all():
            # nothing dropped: every remaining round would fit the same points again and get the same
            # line (the gym calls this once per step; ten fits were most of a step's host time)
            itr = max(itr, numberOfIterations)
            break
        ber = ber[keep]
        snr = snr[keep]
    return snr, ber, p, trendP, itr


# ---------------------------------------------------------------------------
# The gym environment's observation format: rows 0 and 511 of the 1022 x 8176 parity matrix,
# each padded to 8192 bits (one zero after every 511) and packed to 1024 bytes.
_OBS_Z = 511
_OBS_BLOCKS = 16


def _observation_mask():
    mask = np.ones(8192, dtype=bool)
    mask[(np.arange(_OBS_BLOCKS) + 1) * (_OBS_Z + 1) - 1] = False       # common.py:352-354
    return mask


def observation_to_first_columns(compressedMatrix):
    """The two unpadded 8176-bit vectors of an observation (common.py:355-359)."""
    c = np.asarray(compressedMatrix, dtype=np.uint8)
    half = len(c) // 2
    mask = _observation_mask()
    return np.unpackbits(c[:half])[mask], np.unpackbits(c[half:])[mask]


def uncompress(compressedMatrix):
    """Observation -> dense int32 parity matrix, as the reference does it (common.py:349-365):
    every 511-bit segment becomes scipy.linalg.circulant(segment), i.e. the segment is the
    block's first COLUMN (M[i, j] = segment[(i - j) mod 511]), not its first row."""
    top = observation_to_first_columns(compressedMatrix)
    Z = _OBS_Z
    H = np.zeros((2 * Z, _OBS_BLOCKS * Z), dtype=np.int32)
    i = np.arange(Z)
    for b in range(2):
        for j in range(_OBS_BLOCKS):
            for loc in np.flatnonzero(top[b][j * Z:(j + 1) * Z]):
                H[b * Z + i, j * Z + (i - loc) % Z] = 1
    return H


def compress(H):
    """Dense parity matrix -> observation (LdpcEnv.compress, ldpc_env.py:379-393): rows 0 and 511."""
    H = np.asarray(H)
    mask = _observation_mask()
    out = []
    for r in (0, _OBS_Z):
        padded = np.zeros(8192, dtype=np.uint8)
        padded[mask] = H[r, :] != 0
        out.append(np.packbits(padded))
    return np.hstack(out)


# ---------------------------------------------------------------------------
# Monte-Carlo confidence intervals of an evaluation (not in the reference, which reports point
# estimates only, common.py:195-227).
def wilson_interval(errors, trials, z=1.96):
    """Wilson score interval (default 95 %) of a proportion errors / trials, as (low, high)."""
    if trials <= 0:
        return 0.0, 1.0
    p = errors / trials
    d = 1.0 + z * z / trials
    mid = (p + z * z / (2 * trials)) / d
    half = z * np.sqrt((p * (1 - p) + z * z / (4 * trials)) / trials) / d
    return max(mid - half, 0.0), min(mid + half, 1.0)


def ber_interval(bit_errors_per_frame, n, z=1.96):
    """(ber, low, high): bit error rate with a normal-approximation interval whose sampling unit is
    the FRAME -- bit errors come in bursts inside the frames that fail, so treating the n bits of a
    frame as independent trials would understate the spread by orders of magnitude."""
    e = np.asarray(bit_errors_per_frame, dtype=np.float64)
    if e.size == 0:
        return 0.0, 0.0, 1.0
    ber = e.mean() / n
    half = z * e.std(ddof=1) / np.sqrt(e.size) / n if e.size > 1 else 1.0
    return ber, max(ber - half, 0.0), ber + half
