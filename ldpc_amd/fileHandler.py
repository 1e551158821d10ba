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
