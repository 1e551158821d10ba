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
        k = len(rows)
        scatterSNR = np.zeros(k, DECIMAL)
        scatterBER = np.zeros(k, DECIMAL)
        scatterITR = np.zeros(k, DECIMAL)
        snrAxis = np.array(sorted(self.snrAxis))
        p = len(snrAxis)
        snrSum = np.zeros(p, DECIMAL)
        errSum = np.zeros(p, DECIMAL)
        itrSum = np.zeros(p, DECIMAL)
        count = np.zeros(p, DECIMAL)
        for t, row in enumerate(rows):
            scatterSNR[t] = row[1]
            scatterBER[t] = row[5] / codewordSize
            scatterITR[t] = row[6]
            at = np.where(snrAxis == row[0])
            snrSum[at] = snrSum[at] + row[1]
            errSum[at] = errSum[at] + row[5]
            if fixAverageIterations:
                itrSum[at] = itrSum[at] + row[6]
            count[at] = count[at] + 1
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
        ber = ber[keep]
        snr = snr[trendP(snr) > 0]
        itr += 1
    return snr, ber, p, trendP, itr
