"""ldpc_amd -- LDPC code evaluation on AMD MI355X (gfx950).

Host-side mirror of the Python entry points of Omer-Sella/ldpc's decode path
(``ldpc.py``, the GPU driver of ``ldpcCUDA.py``, ``common.berStatistics``,
``fileHandler.readMatrixFromFile``, ``wifiMatrices.getWifiParityMatrix``) on
top of ``libldpc_hip.so`` (hand-written HIP kernels behind a C ABI, see
include/ldpc_hip.h).  There is no CPU fallback: decoding without the built
library or without a HIP device raises.
"""
from . import codes, common  # noqa: F401
from .engine import (ALGO_MINSUM, ALGO_SUMPRODUCT, COUNTER_FIELDS, SIGN_CORRECTED, SIGN_REFERENCE,  # noqa: F401
                     Comm, Decoder, LdpcHipError, comm_unique_id, device_count, library_path, release_pool)

__all__ = ['codes', 'common', 'Comm', 'comm_unique_id', 'Decoder', 'LdpcHipError', 'device_count', 'library_path', 'release_pool', 'ALGO_MINSUM',
           'ALGO_SUMPRODUCT', 'SIGN_REFERENCE', 'SIGN_CORRECTED', 'COUNTER_FIELDS']
