"""Code-construction entry points of the reference's ``fileHandler`` that sit on
the decode path: dense parity-check matrices from hot-position files
(fileHandler.py:137-181), and systematic generator matrices from hexadecimal
first-row files (fileHandler.py:68-135, 151-160).
"""
import numpy as np

from . import codes

GENERAL_CODE_MATRIX_DATA_TYPE = np.int32


def hotLocationsToCirculant(locationList, circulantSize):
    """circulantSize x circulantSize matrix with ones at (i, (i + loc) mod size)
    for every loc in locationList (fileHandler.py:137-142)."""
    return codes.hot_locations_to_circulant(locationList, circulantSize, GENERAL_CODE_MATRIX_DATA_TYPE)


def readMatrixFromFile(fileName, dim0, dim1, circulantSize, isRow=True, isHex=True, isGenerator=True):
    """Dense dim0 x dim1 parity-check matrix from a text file holding, one line
    per circulant, the hot positions of its first row; lines run block-row after
    block-row (fileHandler.py:144-181 with isGenerator=False)."""
    if isGenerator:
        # [I | B], dim0 x (dim0 + 2*circulantSize) (fileHandler.py:151-160)
        return codes.SystematicQCEncoder(codes.parse_generator_file(fileName, dim0, circulantSize),
                                         circulantSize).to_dense(GENERAL_CODE_MATRIX_DATA_TYPE)
    return readCodeFromFile(fileName, dim0, dim1, circulantSize).to_dense(GENERAL_CODE_MATRIX_DATA_TYPE)


def readCodeFromFile(fileName, dim0, dim1, circulantSize):
    """Same file, returned as shift lists (codes.QCCode) -- the form the kernels use."""
    return codes.parse_hot_position_file(fileName, dim0, dim1, circulantSize)


def nearEarthParity():
    """The matrix readMatrixFromFile('codeMatrices/nearEarthParity.txt', 1022, 8176, 511,
    True, False, False) returns (ldpc.py:482), from the table built into this package."""
    return codes.near_earth().to_dense(GENERAL_CODE_MATRIX_DATA_TYPE)


def testFileHandler():
    H = nearEarthParity()
    assert H.shape == (1022, 8176) and int(H.sum()) == 32704
    return 'OK'


# ---- hexadecimal names and .mat code instances (fileHandler.py:36-135, 183-231)

_HEX_DIGITS = np.array(list('0123456789ABCDEF'))


def nibbleToHex(inputArray):
    """Hex digit of a 4-element array weighted 8, 4, 2, 1 (fileHandler.py:36-52).  Like the
    reference, values above 15 (elements that are not bits) come back as decimal text."""
    n = int(np.dot([8, 4, 2, 1], np.asarray(inputArray, dtype=np.int64)))
    return _HEX_DIGITS[n] if 10 <= n <= 15 else str(n)


def binaryArraytoHex(inputArray):
    """(array of digits, string) for an array whose length is a multiple of 4
    (fileHandler.py:54-65).  The array keeps one character per nibble, as numpy's ``str``
    dtype does in the reference."""
    a = np.asarray(inputArray, dtype=np.int64)
    assert a.size % 4 == 0
    vals = a.reshape(-1, 4) @ np.array([8, 4, 2, 1], dtype=np.int64)
    digits = [_HEX_DIGITS[v] if 10 <= v <= 15 else str(int(v)) for v in vals]
    return np.array([d[:1] for d in digits], dtype='<U1').reshape(-1), ''.join(digits)


def hexStringToBinaryArray(hexString):
    """Bits of the digits 0-9, A-F of hexString, four per digit, most significant first; every
    other character (lower-case digits and the newline included) adds nothing
    (fileHandler.py:68-123)."""
    vals = np.array([int(ch, 16) for ch in hexString if ch in '0123456789ABCDEF'], dtype=np.int64)
    bits = (vals[:, None] >> np.array([3, 2, 1, 0])) & 1
    return bits.reshape(-1).astype(GENERAL_CODE_MATRIX_DATA_TYPE)


def hexToCirculant(hexStr, circulantSize):
    """Circulant whose FIRST ROW is the bit string of hexStr without its leading pad bit
    (fileHandler.py:126-135; the branch for strings shorter than the circulant yields a
    matrix of zeros there and here)."""
    bits = hexStringToBinaryArray(hexStr)
    if bits.size < circulantSize:
        bits = np.zeros(circulantSize - bits.size, dtype=GENERAL_CODE_MATRIX_DATA_TYPE)
    else:
        bits = bits[1:]
    idx = (np.arange(bits.size)[None, :] - np.arange(bits.size)[:, None]) % bits.size
    return bits[idx]


def binaryMatrixToHexString(binaryMatrix, circulantSize):
    """Concatenated hex of the first row of every circulant, block-row after block-row; each row
    is preceded by ONE element holding the number 4 - circulantSize % 4 (fileHandler.py:183-200),
    so, as in the reference, only circulant sizes with size % 4 == 3 pass the length assertion."""
    binaryMatrix = np.asarray(binaryMatrix)
    m, n = binaryMatrix.shape
    assert m % circulantSize == 0
    assert n % circulantSize == 0
    lead = 4 - (circulantSize % 4)
    out = []
    for r in range(m // circulantSize):
        for k in range(n // circulantSize):
            row = np.concatenate(([lead], binaryMatrix[r * circulantSize, k * circulantSize:(k + 1) * circulantSize]))
            out.append(binaryArraytoHex(row)[1])
    return ''.join(out)


def codeInstanceName(parityMatrix, circulantSize):
    """'<Z>_<M>_<N>_<sha224 of the hex string>' (fileHandler.py:209-210)."""
    import hashlib
    m, n = np.asarray(parityMatrix).shape
    hexName = binaryMatrixToHexString(parityMatrix, circulantSize)
    return hexName, '%d_%d_%d_%s' % (circulantSize, m // circulantSize, n // circulantSize,
                                     hashlib.sha224(hexName.encode('utf-8')).hexdigest())


def saveCodeInstance(parityMatrix, circulantSize, codewordSize, evaluationData=None, path=None, evaluationTime=0,
                     numberOfNonZero=0, fileName=None):
    """MATLAB workspace with the matrix and, if given, its evaluation (fileHandler.py:203-231):
    keys parityMatrix, fileName, nonZero and snrData, berData, itrData, averageSnrAxis,
    averageNumberOfIterations, evaluationTime.  Without fileName the file is called
    '<Z>_<M>_<N>_<sha224>.mat'.  Returns the name the reference returns: the hex string, or
    fileName when one was given."""
    import scipy.io
    if fileName is None:
        fileName, hashed = codeInstanceName(parityMatrix, circulantSize)
        fileNameWithPath = path + hashed
    else:
        fileNameWithPath = path + fileName
    workspace = {'parityMatrix': parityMatrix, 'fileName': fileName}
    if evaluationData is not None:
        scatterSNR, scatterBER, scatterITR, _, averageSnrAxis, _, averageNumberOfIterations = evaluationData.getStatsV2()
        workspace.update(snrData=scatterSNR, berData=scatterBER, itrData=scatterITR, averageSnrAxis=averageSnrAxis,
                         averageNumberOfIterations=averageNumberOfIterations, evaluationTime=evaluationTime)
    workspace['nonZero'] = numberOfNonZero
    scipy.io.savemat(fileNameWithPath + '.mat', workspace)
    return fileName


def loadCodeInstance(fileNameWithPath, circulantSize=None):
    """Reads a workspace written by saveCodeInstance (here or by the reference): returns
    (codes.QCCode or dense matrix, dict of the remaining entries).  The matrix comes back as shift
    lists when circulantSize is given or can be read from a '<Z>_<M>_<N>_...' file name."""
    import os
    import scipy.io
    ws = scipy.io.loadmat(fileNameWithPath)
    H = np.asarray(ws.pop('parityMatrix'))
    if circulantSize is None:
        head = os.path.basename(fileNameWithPath).split('_')[0]
        circulantSize = int(head) if head.isdigit() else None
    code = codes.dense_to_qc(H, circulantSize) if circulantSize else H
    return code, {k: v for k, v in ws.items() if not k.startswith('__')}
