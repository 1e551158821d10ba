"""Parity-check matrices of the codes the evaluation path is used with, and the
conversions between their three representations:

  dense H (what every reference entry point takes, int array m x n)
  QC shift lists (Z, mb, nb, per-block list of shifts) -- what the HIP kernels use
  CSR (row_ptr, col_idx) -- fallback for matrices that are not block-circulant

Circulant convention (fileHandler.py:137-142, wifiMatrices.py:21-25): a block
with hot position s has a one at (row i, column (i + s) mod Z).
"""
import numpy as np

# CCSDS 131.1-O-2 "Near-Earth" (8176,7156) code: hot positions of the first row
# of each 511x511 circulant, 2 block-rows x 16 block-columns.  Same numbers as
# the reference's data file codeMatrices/nearEarthParity.txt (line 16*b+j+1).
NEAR_EARTH_Z = 511
NEAR_EARTH_HOT = (
    ((0, 176), (12, 239), (0, 352), (24, 431), (0, 392), (151, 409), (0, 351), (9, 359),
     (0, 307), (53, 329), (0, 207), (18, 281), (0, 399), (202, 457), (0, 247), (36, 261)),
    ((99, 471), (130, 473), (198, 435), (260, 478), (215, 420), (282, 481), (48, 396), (193, 445),
     (273, 430), (302, 451), (96, 379), (191, 386), (244, 467), (364, 470), (51, 382), (192, 414)),
)

# IEEE 802.11n, n = 1944, Z = 81, rate 5/6 prototype matrix (-1 = empty block).
# The only WiFi matrix the reference carries (wifiMatrices.py:6-9).
WIFI_1944_Z = 81
WIFI_1944_R56 = (
    (13, 48, 80, 66, 4, 74, 7, 30, 76, 52, 37, 60, -1, 49, 73, 31, 74, 73, 23, -1, 1, 0, -1, -1),
    (69, 63, 74, 56, 64, 77, 57, 65, 6, 16, 51, -1, 64, -1, 68, 9, 48, 62, 54, 27, -1, 0, 0, -1),
    (51, 15, 0, 80, 24, 25, 42, 54, 44, 71, 71, 9, 67, 35, -1, 58, -1, 29, -1, 53, 0, -1, 0, 0),
    (16, 29, 36, 41, 44, 56, 59, 37, 50, 24, -1, 65, 4, 65, 52, -1, 4, -1, 73, 52, 1, -1, -1, 0),
)


# The other three n = 1944, Z = 81 prototype matrices of IEEE 802.11n-2009 (Annex R; the text's
# Table R.3): rates 1/2, 2/3 and 3/4.  NOT in the reference (it carries rate 5/6 only), so nothing
# of the reference pins them: they are restated from the standard, and tests/test_wifi_rates.py checks
# what can be checked without it (the standard's dual-diagonal parity structure, full rank, girth >= 6,
# decoding behaviour).
_ = -1
WIFI_1944_R12 = (
    (57, _, _, _, 50, _, 11, _, 50, _, 79, _, 1, 0, _, _, _, _, _, _, _, _, _, _),
    (3, _, 28, _, 0, _, _, _, 55, 7, _, _, _, 0, 0, _, _, _, _, _, _, _, _, _),
    (30, _, _, _, 24, 37, _, _, 56, 14, _, _, _, _, 0, 0, _, _, _, _, _, _, _, _),
    (62, 53, _, _, 53, _, _, 3, 35, _, _, _, _, _, _, 0, 0, _, _, _, _, _, _, _),
    (40, _, _, 20, 66, _, _, 22, 28, _, _, _, _, _, _, _, 0, 0, _, _, _, _, _, _),
    (0, _, _, _, 8, _, 42, _, 50, _, _, 8, _, _, _, _, _, 0, 0, _, _, _, _, _),
    (69, 79, 79, _, _, _, 56, _, 52, _, _, _, 0, _, _, _, _, _, 0, 0, _, _, _, _),
    (65, _, _, _, 38, 57, _, _, 72, _, 27, _, _, _, _, _, _, _, _, 0, 0, _, _, _),
    (64, _, _, _, 14, 52, _, _, 30, _, _, 32, _, _, _, _, _, _, _, _, 0, 0, _, _),
    (_, 45, _, 70, 0, _, _, _, 77, 9, _, _, _, _, _, _, _, _, _, _, _, 0, 0, _),
    (2, 56, _, 57, 35, _, _, _, _, _, 12, _, _, _, _, _, _, _, _, _, _, _, 0, 0),
    (24, _, 61, _, 60, _, _, 27, 51, _, _, 16, 1, _, _, _, _, _, _, _, _, _, _, 0),
)
WIFI_1944_R23 = (
    (61, 75, 4, 63, 56, _, _, _, _, _, _, 8, _, 2, 17, 25, 1, 0, _, _, _, _, _, _),
    (56, 74, 77, 20, _, _, _, 64, 24, 4, 67, _, 7, _, _, _, _, 0, 0, _, _, _, _, _),
    (28, 21, 68, 10, 7, 14, 65, _, _, _, 23, _, _, _, 75, _, _, _, 0, 0, _, _, _, _),
    (48, 38, 43, 78, 76, _, _, _, _, 5, 36, _, 15, 72, _, _, _, _, _, 0, 0, _, _, _),
    (40, 2, 53, 25, _, 52, 62, _, 20, _, _, 44, _, _, _, _, 0, _, _, _, 0, 0, _, _),
    (69, 23, 64, 10, 22, _, 21, _, _, _, _, _, 68, 23, 29, _, _, _, _, _, _, 0, 0, _),
    (12, 0, 68, 20, 55, 61, _, 40, _, _, _, 52, _, _, _, 44, _, _, _, _, _, _, 0, 0),
    (58, 8, 34, 64, 78, _, _, 11, 78, 24, _, _, _, _, _, 58, 1, _, _, _, _, _, _, 0),
)
WIFI_1944_R34 = (
    (48, 29, 28, 39, 9, 61, _, _, _, 63, 45, 80, _, _, _, 37, 32, 22, 1, 0, _, _, _, _),
    (4, 49, 42, 48, 11, 30, _, _, _, 49, 17, 41, 37, 15, _, 54, _, _, _, 0, 0, _, _, _),
    (35, 76, 78, 51, 37, 35, 21, _, 17, 64, _, _, _, 59, 7, _, _, 32, _, _, 0, 0, _, _),
    (9, 65, 44, 9, 54, 56, 73, 34, 42, _, _, _, 35, _, _, _, 46, 39, 0, _, _, 0, 0, _),
    (3, 62, 7, 80, 68, 26, _, 80, 55, _, 36, _, 26, _, 9, _, 72, _, _, _, _, _, 0, 0),
    (26, 75, 33, 21, 69, 59, 3, 38, _, _, _, 35, _, 62, 36, 26, _, _, 1, _, _, _, _, 0),
)
del _
WIFI_1944_TABLES = {'1/2': WIFI_1944_R12, '2/3': WIFI_1944_R23, '3/4': WIFI_1944_R34, '5/6': WIFI_1944_R56}


class QCCode:
    """Quasi-cyclic code as per-block shift lists.

    blocks[b][j] is a tuple of distinct shifts (possibly empty); row i of block
    (b, j) has ones at columns (i + s) mod Z.
    """

    def __init__(self, Z, blocks):
        self.Z = int(Z)
        self.blocks = tuple(tuple(tuple(int(s) for s in blk) for blk in row) for row in blocks)
        self.mb = len(self.blocks)
        self.nb = len(self.blocks[0])
        for row in self.blocks:
            if len(row) != self.nb:
                raise ValueError('ragged block table')
            for blk in row:
                if len(set(blk)) != len(blk) or any(s < 0 or s >= self.Z for s in blk):
                    raise ValueError('shifts of a block must be distinct and in [0, Z)')
        self.m = self.mb * self.Z
        self.n = self.nb * self.Z

    @property
    def nnz(self):
        return self.Z * sum(len(blk) for row in self.blocks for blk in row)

    def row_degrees(self):
        return [sum(len(blk) for blk in row) for row in self.blocks]

    def col_degrees(self):
        return [sum(len(self.blocks[b][j]) for b in range(self.mb)) for j in range(self.nb)]

    def flat(self):
        """(blk_ptr int32[mb*nb+1], shifts int32[...]) as ldpc_create_qc takes them."""
        ptr = [0]
        sh = []
        for row in self.blocks:
            for blk in row:
                sh.extend(blk)
                ptr.append(len(sh))
        return np.asarray(ptr, np.int32), np.asarray(sh, np.int32)

    def to_dense(self, dtype=np.int32):
        Z = self.Z
        H = np.zeros((self.m, self.n), dtype=dtype)
        i = np.arange(Z)
        for b, row in enumerate(self.blocks):
            for j, blk in enumerate(row):
                for s in blk:
                    H[b * Z + i, j * Z + (i + s) % Z] = 1
        return H

    def with_block(self, b, j, shifts):
        """Copy with block (b, j) replaced -- the gym's replaceCirculant (ldpc_env.py:293-317)."""
        blocks = [list(row) for row in self.blocks]
        blocks[b][j] = tuple(shifts)
        return QCCode(self.Z, blocks)

    def key(self):
        return ('qc', self.Z, self.blocks)


def near_earth():
    return QCCode(NEAR_EARTH_Z, NEAR_EARTH_HOT)


def near_earth_zeroed(b, j):
    """NearEarth with circulant (b, j) zeroed (testMatricesGeneratorScript.py:27-31)."""
    return near_earth().with_block(b, j, ())


def wifi_1944_r56():
    return QCCode(WIFI_1944_Z, [[() if s < 0 else (s,) for s in row] for row in WIFI_1944_R56])


def wifi_1944(rate):
    """802.11n n = 1944 code of rate '1/2', '2/3', '3/4' or '5/6' (only 5/6 is in the reference)."""
    return QCCode(WIFI_1944_Z, [[() if s < 0 else (s,) for s in row] for row in WIFI_1944_TABLES[rate]])


def wifi_1944_all():
    return {rate: wifi_1944(rate) for rate in ('1/2', '2/3', '3/4', '5/6')}


def _dense_to_qc_native(H, Z):
    """dense_to_qc through the library's host helper (several threads over the rows); NotImplemented when
    the array's layout is not one it reads."""
    if not (H.flags.c_contiguous and H.dtype.itemsize in (1, 4, 8) and H.dtype.kind in 'biuf'):
        return NotImplemented
    from . import _lib
    try:
        L = _lib.load()
    except ImportError:
        return NotImplemented
    m, n = H.shape
    mb, nb = m // Z, n // Z
    ptr = np.zeros(mb * nb + 1, np.int32)
    cap = 64 * mb * nb
    while True:
        sh = np.zeros(cap, np.int32)
        rc = L.ldpc_dense_to_qc(H.ctypes.data, H.dtype.itemsize, m, n, Z, ptr.ctypes.data, sh.ctypes.data, cap)
        if rc == -1 and cap < Z * mb * nb:           # more shifts than the array held: a dense matrix; once more with room for all
            cap = Z * mb * nb
            continue
        break
    if rc == 1:
        return None
    if rc != 0:
        return NotImplemented
    return QCCode(Z, [[tuple(int(x) for x in sh[ptr[b * nb + j]:ptr[b * nb + j + 1]]) for j in range(nb)] for b in range(mb)])


def dense_to_qc(H, Z):
    """QCCode of a dense H if every Z x Z block is a sum of cyclic shifts, else None."""
    H = np.asarray(H)
    m, n = H.shape
    if Z <= 0 or m % Z or n % Z:
        return None
    native = _dense_to_qc_native(H, Z)
    if native is not NotImplemented:
        return native
    mb, nb = m // Z, n // Z
    i = np.arange(Z)
    blocks = []
    for b in range(mb):
        row = []
        for j in range(nb):
            blk = H[b * Z:(b + 1) * Z, j * Z:(j + 1) * Z]
            hot = np.flatnonzero(blk[0])
            if np.count_nonzero(blk) != Z * len(hot):
                return None
            for s in hot:
                if not np.all(blk[i, (i + s) % Z] != 0):
                    return None
            row.append(tuple(int(s) for s in hot))
        blocks.append(row)
    return QCCode(Z, blocks)


def detect_qc(H, candidates=None):
    """Largest circulant size Z for which H is block-circulant and the kernels'
    limits hold (mb * roundup(Z, 64) <= 1024 threads), or None."""
    H = np.asarray(H)
    m, n = H.shape
    if candidates is None:
        g = int(np.gcd(m, n))
        candidates = [z for z in range(g, 7, -1) if g % z == 0]
    for Z in candidates:
        if m % Z or n % Z:
            continue
        if (m // Z) * ((Z + 63) // 64 * 64) > 1024 or -(-(n // Z) // (m // Z)) > 32:
            continue
        qc = dense_to_qc(H, Z)
        if qc is not None:
            return qc
    return None


def gym_episode_state(base, n_actions, seed):
    """The code after n_actions steps of the gym environment from `base`: every action replaces one
    circulant by one with 1 ... 7 hot bits and the state keeps them all (ldpc_env.py:293-317;
    hot-bit budget: models.py:296, 368).  Synthetic actions from RandomState(seed), for benchmarks."""
    prng = np.random.RandomState(seed)
    state = base
    for _ in range(int(n_actions)):
        w = int(prng.randint(1, 8))
        state = state.with_block(int(prng.randint(base.mb)), int(prng.randint(base.nb)),
                                 sorted(prng.choice(base.Z, w, replace=False).tolist()))
    return state


def dense_to_csr(H):
    """(m, n, row_ptr int32[m+1], col_idx int32[E]) with ascending columns per row,
    the adjacency ldpcDecoder.__init__ derives with np.where (ldpc.py:237-247)."""
    H = np.asarray(H)
    m, n = H.shape
    rows, cols = np.nonzero(H)
    row_ptr = np.zeros(m + 1, np.int64)
    np.add.at(row_ptr, rows + 1, 1)
    return m, n, np.cumsum(row_ptr).astype(np.int32), cols.astype(np.int32)


def hot_locations_to_circulant(locations, Z, dtype=np.int32):
    """Z x Z circulant with ones at (i, (i + loc) mod Z) (fileHandler.py:137-142)."""
    C = np.zeros((Z, Z), dtype=dtype)
    i = np.arange(Z)
    for s in locations:
        C[i, (i + int(s)) % Z] = 1
    return C


def parse_hot_position_file(path, dim0, dim1, Z):
    """QCCode from a text file with one circulant per line ("a, b, ..." = hot
    positions of its first row), block-row after block-row -- the layout
    fileHandler.readMatrixFromFile(..., isGenerator=False) reads
    (fileHandler.py:161-181; that function only handles two block-rows, this
    one handles dim0 // Z)."""
    with open(path) as fid:
        lines = [ln.strip() for ln in fid.readlines() if ln.strip()]
    mb, nb = dim0 // Z, dim1 // Z
    if len(lines) < mb * nb:
        raise ValueError(f'{path}: expected {mb * nb} lines, found {len(lines)}')
    blocks = [[tuple(sorted(set(int(t) for t in lines[b * nb + j].split(',') if t.strip() != '')))
               for j in range(nb)] for b in range(mb)]
    return QCCode(Z, blocks)


def observation_to_qc(compressedMatrix):
    """The gym observation (2048 bytes) as a QCCode, without building the dense matrix.
    common.uncompress (common.py:349-365) turns every 511-bit segment into a circulant whose
    first COLUMN is the segment, so a hot position `loc` of the segment is the shift
    (-loc) mod 511 of this module's convention (row i has its ones at (i + shift) mod Z)."""
    from .common import observation_to_first_columns
    top = observation_to_first_columns(compressedMatrix)
    Z = 511
    blocks = [[sorted(int((-loc) % Z) for loc in np.flatnonzero(top[b][j * Z:(j + 1) * Z]))
               for j in range(16)] for b in range(2)]
    return QCCode(Z, blocks)


# ---------------------------------------------------------------------------
# Systematic quasi-cyclic generators G = [I | B], B made of Z x Z circulants given by the
# first row of each (hexadecimal, most significant bit first; fileHandler.py:68-135, 144-160).
def parse_generator_file(path, dim0, Z):
    """First rows of the circulants of B: uint8 [dim0 // Z][2][Z].  Line 2i holds the left,
    line 2i+1 the right circulant of block-row i.  As in the reference (hexStringToBinaryArray and
    hexToCirculant, fileHandler.py:68-135): only the characters 0-9 and A-F count (anything else on
    the line is skipped), four bits each, most significant first, and the leading bit is dropped;
    what remains must be exactly Z bits -- a malformed file fails here instead of producing a
    generator of the wrong shape."""
    with open(path) as fid:
        lines = [ln.strip() for ln in fid.readlines()]
    rows = []
    for i in range(dim0 // Z):
        pair = []
        for ln in (lines[2 * i], lines[2 * i + 1]):
            digits = [int(ch, 16) for ch in ln if ch in '0123456789ABCDEF']
            bits = np.array([(d >> k) & 1 for d in digits for k in (3, 2, 1, 0)], dtype=np.uint8)
            if len(bits) != Z + 1:
                raise ValueError(f'generator line holds {len(bits)} bits, expected {Z + 1} (a pad bit and {Z} of the circulant)')
            pair.append(bits[1:].copy())
        rows.append(pair)
    return np.array(rows, dtype=np.uint8)


class SystematicQCEncoder:
    """codeword = [message | message . B mod 2] for G = [I | B] (readMatrixFromFile with
    isGenerator=True, fileHandler.py:151-160).  Row r = i*Z + t of B is the first row of
    block-row i's circulants rotated right by t."""

    def __init__(self, first_rows, Z):
        first_rows = np.asarray(first_rows, dtype=np.uint8)
        self.Z = int(Z)
        self.kb, self.pb = first_rows.shape[0], first_rows.shape[1]
        self.k = self.kb * self.Z
        self.n = self.k + self.pb * self.Z
        idx = (np.arange(Z)[None, :] - np.arange(Z)[:, None]) % Z          # [t, col] -> position in the first row
        B = np.zeros((self.k, self.pb * Z), dtype=np.uint8)
        for i in range(self.kb):
            for j in range(self.pb):
                B[i * Z:(i + 1) * Z, j * Z:(j + 1) * Z] = first_rows[i, j][idx]
        self.B = B

    def encode(self, messages):
        msg = np.atleast_2d(np.asarray(messages)).astype(np.uint8)
        if msg.shape[1] != self.k:
            raise ValueError(f'message length {msg.shape[1]} != {self.k}')
        parity = (msg.astype(np.int32) @ self.B.astype(np.int32)) & 1
        return np.concatenate([msg, parity.astype(np.uint8)], axis=1)

    def to_dense(self, dtype=np.int32):
        return np.hstack([np.eye(self.k, dtype=dtype), self.B.astype(dtype)])
