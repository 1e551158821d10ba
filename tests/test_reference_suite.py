"""The reference's own test file (test_ldpc.py) with the module swap applied: its tests of the evaluation path, in
its words.  Its other tests (binarySpace, models, utilityFunctions) belong to the agent side, which is not rebuilt."""
import pytest


def test_roundtripCompression():
    """test_ldpc.py:20-23"""
    from ldpc_amd import gym_eval as ldpc_env
    status = ldpc_env.testCompressionRoundrip()
    assert status == 'OK'


@pytest.mark.gpu
def test_ldpcCUDAdecoder():
    """test_ldpc.py:25-28"""
    import ldpc_amd.ldpcHIP as ldpcCUDA
    _, status = ldpcCUDA.testNearEarth(graphics=False)
    assert status == 'OK'


def test_fileHandler_nearEarth():
    """test_ldpc.py:31-33"""
    import ldpc_amd.fileHandler as fileHandler
    assert fileHandler.testFileHandler() == 'OK'


def test_modulationAndSlicingRoundTrip():
    """test_ldpc.py:46-48"""
    import ldpc_amd.ldpc as ldpc
    assert ldpc.testModulationAndSlicingRoundTrip() == 'OK'


def test_gym_ldpc_environment_is_able_to_init():
    """test_ldpc.py:50-53 (the attribute it looks at, gpuDeviceNumber, is not one LdpcEnv has, ldpc_env.py:80-148;
    the device count and the observation are)."""
    from ldpc_amd.gym_eval import LdpcEnv
    ldpcEnvironment = LdpcEnv()
    assert ldpcEnvironment.cudaDevices == 4
    assert ldpcEnvironment.observed_state.shape == (2048,)


@pytest.mark.gpu
def test_wifiCUDAdecoder():
    """wifiCUDA.main (wifiCUDA.py:685-698) runs testWifi; same form as test_ldpcCUDAdecoder"""
    import ldpc_amd.wifiHIP as wifiCUDA
    bStats, status = wifiCUDA.testWifi(10)
    assert status in ('OK', 'WIFI problem') and len(bStats.getRawStats()) == 40
