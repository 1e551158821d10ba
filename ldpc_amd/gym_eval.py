"""The code-evaluation step of the gym environment (LdpcEnv.evaluateCode / calcReward,
ldpc_env.py:319-371, and the part of LdpcEnv.step that calls them, ldpc_env.py:217-226)
on the HIP engine: a parity matrix that changes every step (one circulant replaced), a few
SNR points x a few frames, reward = area between the constant 1 and the line fitted through
the per-frame (SNR, BER) scatter.  A new matrix costs a table upload, never a recompilation.
"""
import numpy as np

from . import common
from . import ldpcHIP

LDPC_ENV_NUMBER_OF_ITERATIONS = 50          # ldpc_env.py:21
LDPC_ENV_NUMBER_OF_TRANSMISSIONS = 10       # ldpc_env.py:22
LDPC_ENV_MAX_SEED = 2 ** 31 - 1             # ldpc_env.py:27


class CodeEvaluator:
    """Holds what LdpcEnv holds for evaluation: SNR points, iteration and frame budgets, the
    generator the per-evaluation seeds are drawn from, and the number of devices."""

    def __init__(self, seed=7134066, SNRpoints=(3.0, 3.2, 3.4), numberOfIterations=LDPC_ENV_NUMBER_OF_ITERATIONS,
                 numberOfTransmissions=LDPC_ENV_NUMBER_OF_TRANSMISSIONS, numberOfDevices=1,
                 rewardForBadCandidate=0.0, prec=32, deviceRng=False):
        self.localPRNG = np.random.RandomState(seed)
        self.SNRpoints = np.array(SNRpoints, dtype=np.float32)
        self.numberOfIterations = int(numberOfIterations)
        self.numberOfTransmissions = int(numberOfTransmissions)
        self.cudaDevices = int(numberOfDevices)
        self.rewardForBadCandidate = rewardForBadCandidate
        self.prec = prec
        self.deviceRng = deviceRng
        self.pConst = np.poly1d([1])

    def evaluateCode(self, parityMatrix):
        """ldpc_env.py:347-371: per-device seeds from the environment's generator, then the
        multi-device driver.  (The extra single seed is drawn and unused, as in the reference.)"""
        seeds = self.localPRNG.randint(0, LDPC_ENV_MAX_SEED, self.cudaDevices, dtype=np.int32)
        self.localPRNG.randint(0, LDPC_ENV_MAX_SEED, 1, dtype=np.int32)
        seeds = np.maximum(seeds, 1)            # the driver asserts seed > 0 (ldpcCUDA.py:532)
        return ldpcHIP.evaluateCodeCudaWrapper(seeds, self.SNRpoints, self.numberOfIterations, parityMatrix,
                                               self.numberOfTransmissions, 'None', self.cudaDevices,
                                               prec=self.prec, deviceRng=self.deviceRng)

    def calcReward(self, berStats):
        """ldpc_env.py:319-344 on the scatter data of ldpc_env.py:218."""
        scatterSnr, scatterBer, _, _, _, berData, _ = berStats.getStatsV2()
        if len(berData) < 2:
            return self.rewardForBadCandidate
        _, _, p1, _, _ = common.recursiveLinearFit(scatterSnr, scatterBer)
        pTotalInteg = (self.pConst - p1).integ()
        return pTotalInteg(self.SNRpoints[-1]) - pTotalInteg(self.SNRpoints[0])

    def step(self, parityMatrix):
        """(reward, berStatistics) of one candidate matrix: dense array, QCCode, or the
        2048-byte observation of the environment."""
        from . import codes
        H = parityMatrix
        if isinstance(H, np.ndarray) and H.ndim == 1 and H.dtype == np.uint8 and H.size == 2048:
            H = codes.observation_to_qc(H)
        berStats = self.evaluateCode(H)
        return self.calcReward(berStats), berStats
