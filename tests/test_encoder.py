"""Generator side of the path (the G argument of the reference's evaluation functions,
ldpc.py:409-414): the Near-Earth generator file the reference ships
(codeMatrices/nearEarthGenerator.txt, kept as a data fixture) against this package's
parity-check table, and decoding of random codewords on the GPU."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, build_code

GEN = os.path.join(GOLDEN, 'nearEarthGenerator.txt')


@pytest.fixture(scope='module')
def encoder():
    from ldpc_amd import codes
    return codes.SystematicQCEncoder(codes.parse_generator_file(GEN, 7154, 511), 511)


def test_generator_rows_are_codewords(encoder):
    """H . G^T = 0: pins the hexadecimal format (bit order, dropped pad bit, rotation sense)
    against the parity-check matrix, which the golden decodes pin independently."""
    _, H = build_code('nearEarth')
    assert encoder.k == 7154 and encoder.n == 8176
    prng = np.random.RandomState(3)
    msgs = prng.randint(0, 2, (40, 7154))
    msgs[0] = 0
    msgs[1] = 0
    msgs[1, 5] = 1                      # a single row of G
    msgs[2] = 0
    msgs[2, 7153] = 1
    cw = encoder.encode(msgs)
    assert cw.shape == (40, 8176) and np.array_equal(cw[:, :7154], msgs)
    assert not ((H.astype(np.int64) @ cw.T.astype(np.int64)) & 1).any()
    assert cw[1].sum() > 1 and cw[0].sum() == 0


def test_read_matrix_from_file_generator_branch(encoder):
    from ldpc_amd import fileHandler
    G = fileHandler.readMatrixFromFile(GEN, 7154, 8176, 511, True, True, True)
    assert G.shape == (7154, 8176) and G.dtype == np.int32
    assert np.array_equal(G[:, :7154], np.eye(7154, dtype=np.int32))
    m = np.zeros(7154, np.int64)
    m[[3, 600, 7000]] = 1
    assert np.array_equal((m @ G) & 1, encoder.encode(m)[0])


@pytest.mark.gpu
def test_random_codewords_decode_like_the_all_zero_word(encoder):
    """The reference evaluates with the all-zero codeword only.  For a code with even check
    degrees the decoder is symmetric: sending codeword c over noise n gives the mirror image
    of sending 0 over noise n*(1 - 2c) -- same iteration count, decisions XOR c."""
    import ldpc_amd
    from ldpc_amd import codes
    from ldpc_amd import ldpc as L
    prng = np.random.RandomState(12)
    F, n = 24, 8176
    cw = encoder.encode(prng.randint(0, 2, (F, 7154)))
    sigma = np.sqrt(0.5 / 10 ** 0.34)
    noise = prng.normal(0, sigma, (F, n))
    x = 2.0 * cw - 1.0                                    # modulate, ldpc.py:62-66
    y = x + noise
    y0 = -1.0 - x * noise                                 # all-zero word, mirrored noise: y = -x * y0
    with ldpc_amd.Decoder(codes.near_earth()) as eng:
        for prec in (64, 32):
            a = eng.decode(y.astype(np.float32), 50, prec=prec)
            b = eng.decode(y0.astype(np.float32), 50, prec=prec)
            assert np.array_equal(a['iters'], b['iters']) and np.array_equal(a['converged'], b['converged'])
            assert np.array_equal(a['hard'] ^ cw, b['hard'])
            assert a['converged'].sum() >= F - 2 and np.array_equal(a['hard'][a['converged'] == 1], cw[a['converged'] == 1])
    # the reference's G branch (one frame per SNR point, first symbol flipped)
    G = encoder.to_dense().T                               # ldpc.py:412 computes G.dot(message)
    st = L.evaluateCodeAtSingleTransmission(77, [3.0, 3.4], 7154, 8176, 50, codes.near_earth().to_dense(), G)
    rows = st.getRawStats() if hasattr(st, 'getRawStats') else None
    assert rows is not None and len(rows) == 2
