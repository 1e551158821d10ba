"""802.11n parity-check matrix the reference evaluates (wifiMatrices.py:6-34):
n = 1944, Z = 81, rate 5/6 -- the only WiFi matrix it carries."""
import numpy as np

from . import codes

WIFI_LDPC_DATA_TYPE = np.int64


def getWifiParityMatrix(codewordSize=1944, circulantSize=81, rate=5 / 6):
    """Dense 324 x 1944 matrix: block (i, j) is the identity shifted so that row r
    has its one at column (r + shift) mod 81; empty where the table has no entry."""
    if codewordSize != 1944 or circulantSize != 81 or rate != 5 / 6:
        raise ValueError('only the (1944, Z=81, rate 5/6) matrix is defined, as in the reference')
    return codes.wifi_1944_r56().to_dense(WIFI_LDPC_DATA_TYPE)


def getWifiCode():
    return codes.wifi_1944_r56()
