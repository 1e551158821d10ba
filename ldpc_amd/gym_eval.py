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


class LdpcEnv:
    """The environment object of the reference (LdpcEnv, ldpc_env.py:49-407) as far as it drives the
    evaluation path: the same constructor arguments, attributes, `reset`, `step(action)`,
    `replaceCirculant`, `evaluateCode`, `calcReward`, `compress` / `uncompress`, with the state kept as
    per-circulant shift lists instead of a dense 1022 x 8176 array (`state` builds the dense matrix on
    request).  A step therefore hands the engine 2 x 16 short lists, not 8.4 million matrix entries.
    It is a plain object: no gym base class, spaces or rendering (ldpc_env.py:100-107, 286-288).

    Reproduced as they are: every action is legal (the coordinate test of ldpc_env.py:306 cannot fail
    for 1 + 4 coordinate bits), `done` stays False (the accumulated decoding time is never updated,
    ldpc_env.py:260, 372-375), and in `replacementOnly` mode the source block is read at
    (xr, xr) and, through NumPy views, ends up in both (x, y) and (xr, yr) (ldpc_env.py:191-196)."""

    def __init__(self, replacementOnly=False, seed=7134066, numberOfCudaDevices=4, resetHammingDistance='MAXIMUM',
                 prec=32, deviceRng=False):
        from . import codes
        self.replacementOnly = replacementOnly
        self.messageSize = 7156                                        # sic, ldpc_env.py:84
        self.codewordSize = 8176
        self.SNRpoints = np.array([3.0, 3.2, 3.4], dtype=np.float32)
        self.BERpoints = np.ones(len(self.SNRpoints), dtype=np.float32)
        self.resetValue = codes.near_earth()
        self._code = self.resetValue
        self.circulantSize = 511
        self.xBits = 1
        self.yBits = 4
        self.actionBits = (self.xBits + self.yBits + self.circulantSize if not replacementOnly
                           else 2 * self.xBits + 2 * self.yBits)
        self.observationUint8 = 2048
        self.ldpcDecoderNumOfIterations = LDPC_ENV_NUMBER_OF_ITERATIONS
        self.ldpcDecoderNumOfTransmissions = LDPC_ENV_NUMBER_OF_TRANSMISSIONS
        self.xCoordinateBinaryToInt = np.flipud(2 ** np.arange(self.xBits))
        self.yCoordinateBinaryToInt = np.flipud(2 ** np.arange(self.yBits))
        self.seed = seed
        self.rewardForIllegalAction = -2.0
        self.rewardForBadCandidate = -2.0
        self.scatterSnr = self.SNRpoints
        self.scatterBer = self.BERpoints
        self.scatterItr = np.ones(len(self.SNRpoints), dtype=np.float32)
        self.accumulatedEvaluationTime = 0
        self.berStats = common.berStatistics()
        self.pConst = np.poly1d([1])
        self.localPRNG = np.random.RandomState(seed)
        self.cudaDevices = numberOfCudaDevices
        self.resetHammingDistance = resetHammingDistance
        self._evaluator = CodeEvaluator(SNRpoints=self.SNRpoints, numberOfIterations=self.ldpcDecoderNumOfIterations,
                                        numberOfTransmissions=self.ldpcDecoderNumOfTransmissions,
                                        numberOfDevices=numberOfCudaDevices,
                                        rewardForBadCandidate=self.rewardForBadCandidate, prec=prec, deviceRng=deviceRng)
        self._evaluator.localPRNG = self.localPRNG                      # one generator, as in the reference
        self.observed_state = self.compress()

    # ---- state
    @property
    def code(self):
        """The current parity-check matrix as shift lists (ldpc_amd.codes.QCCode)."""
        return self._code

    @property
    def state(self):
        """The dense matrix the reference keeps in `self.state` (built on request)."""
        return self._code.to_dense()

    def reset(self):
        """ldpc_env.py:267-284"""
        self._code = self.resetValue
        self.observed_state = self.compress()
        self.BERpoints = np.ones(len(self.SNRpoints), dtype=np.float32)
        self.accumulatedEvaluationTime = 0
        return self.observed_state

    def replaceCirculant(self, xCoordinate, yCoordinate, newCirculant):
        """ldpc_env.py:293-317.  newCirculant: a Z x Z circulant matrix (row i = row 0 rotated right by i,
        what `circulant(firstRow).T` is), or the hot positions of its first row."""
        if (xCoordinate > (2 ** self.xBits)) or (yCoordinate > (2 ** self.yBits)):
            return 'Illegal action'
        c = np.asarray(newCirculant)
        hot = np.flatnonzero(c[0]) if c.ndim == 2 else c.astype(np.int64)
        self._code = self._code.with_block(int(xCoordinate), int(yCoordinate), sorted(int(s) for s in hot))
        return 'OK'

    def step(self, action):
        """ldpc_env.py:169-264 -> (observation, reward, done, {})"""
        done = False
        action = np.asarray(action)
        assert (action.shape[0] == self.actionBits if self.replacementOnly
                else action.shape[0] == self.xBits + self.yBits + self.circulantSize)
        xCoordinate = int(self.xCoordinateBinaryToInt.dot(action[0:self.xBits]))
        yCoordinate = int(self.yCoordinateBinaryToInt.dot(action[self.xBits:self.xBits + self.yBits]))
        if self.replacementOnly:
            o = self.xBits + self.yBits
            xrCoordinate = int(self.xCoordinateBinaryToInt.dot(action[o:o + self.xBits]))
            yrCoordinate = int(self.yCoordinateBinaryToInt.dot(action[o + self.xBits:o + self.xBits + self.yBits]))
            source = self._code.blocks[xrCoordinate][xrCoordinate]       # (xr, xr), ldpc_env.py:192
            status = self.replaceCirculant(xCoordinate, yCoordinate, np.asarray(source, dtype=np.int64))
            if status == 'OK':
                # the reference passes a view of block (x, y), which by now holds the new content
                status = self.replaceCirculant(xrCoordinate, yrCoordinate,
                                               np.asarray(self._code.blocks[xCoordinate][yCoordinate], dtype=np.int64))
            else:
                status = 'FAILED'
        else:
            circulantFirstRow = action[self.xBits + self.yBits:]
            status = self.replaceCirculant(xCoordinate, yCoordinate, np.flatnonzero(circulantFirstRow))
        if status == 'OK':
            self.berStats = self.evaluateCode()
            (self.scatterSnr, self.scatterBer, self.scatterItr, snrAxis, averageSnrAxis, berData,
             averageNumberOfIterations) = self.berStats.getStatsV2()
            self.BERpoints = berData
            reward = self.calcReward()
        else:
            reward = self.rewardForIllegalAction
        self.observed_state = self.compress()
        return self.observed_state, reward, done, {}

    def evaluateCode(self):
        """ldpc_env.py:353-377"""
        return self._evaluator.evaluateCode(self._code)

    def calcReward(self, colour=None):
        """ldpc_env.py:319-345"""
        if len(self.BERpoints) < 2:
            return self.rewardForBadCandidate
        snr, ber, p1, trendP, itr = common.recursiveLinearFit(self.scatterSnr, self.scatterBer)
        pTotalInteg = (self.pConst - p1).integ()
        return pTotalInteg(self.SNRpoints[-1]) - pTotalInteg(self.SNRpoints[0])

    def compress(self):
        """ldpc_env.py:379-394: rows 0 and 511 of the matrix, a zero bit after every 511, packed to 2 x 1024 bytes.
        Row 0 of a block has its ones at the block's shifts."""
        rows = []
        for b in range(2):
            padded = np.zeros(8192, dtype=np.uint8)
            for j in range(16):
                sh = np.asarray(self._code.blocks[b][j], dtype=np.int64)
                padded[j * 512 + sh] = 1
            rows.append(np.packbits(padded))
        self.observed_state = np.hstack(rows)
        return self.observed_state

    def uncompress(self):
        """ldpc_env.py:396-401 -> (row 0, row 511) of the matrix, 8176 bits each"""
        half = len(self.observed_state) // 2
        mask = np.ones(8192, dtype=bool)
        mask[(np.arange(16) + 1) * 512 - 1] = False
        return np.unpackbits(self.observed_state[:half])[mask], np.unpackbits(self.observed_state[half:])[mask]

    def constantFunction(self, alpha):
        def g(x):
            return alpha
        return g


def testCompressionRoundrip():
    """ldpc_env.py:410-418: the observation unpacks to rows 0 and 511 of the state, 1000 random states.  The
    reference draws unstructured random matrices (only the two rows matter to it); here the state is block-circulant
    by construction, so every round installs random circulants."""
    testEnvironment = LdpcEnv(numberOfCudaDevices=1)
    prng = np.random.RandomState(1)
    for i in range(1000):
        firstRow, secondRow = testEnvironment.uncompress()
        dense0 = np.zeros(8176, dtype=np.uint8)
        dense1 = np.zeros(8176, dtype=np.uint8)
        for j in range(16):
            dense0[j * 511 + np.asarray(testEnvironment.code.blocks[0][j], dtype=np.int64)] = 1
            dense1[j * 511 + np.asarray(testEnvironment.code.blocks[1][j], dtype=np.int64)] = 1
        assert np.all(firstRow == dense0), "First row does not match"
        assert np.all(secondRow == dense1), "Second row does not match"
        for _ in range(3):
            testEnvironment.replaceCirculant(int(prng.randint(2)), int(prng.randint(16)),
                                             prng.choice(511, int(prng.randint(0, 9)), replace=False))
        testEnvironment.observed_state = testEnvironment.compress()
    return 'OK'
