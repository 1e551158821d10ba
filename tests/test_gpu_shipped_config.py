"""GPU suite (-m gpu): the configuration users get.  Every other GPU test runs with LDPC_HIP_TEST_HOOKS=1
(tests/conftest.py), under which the library re-selects its kernel on every call and reads the environment;
this one starts a fresh process WITHOUT it and checks the HIP path against the CPU oracle and the reference's
golden results there: cached kernel choice (every combination is called twice), edge tables built on first
need, no environment variable honoured.  ldpc.py:326-337 through the path a caller of the library takes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_shipped_configuration_against_oracle_and_golden():
    env = {k: v for k, v in os.environ.items() if not k.startswith('LDPC_HIP_')}
    p = subprocess.run([sys.executable, os.path.join(HERE, 'shipped_config_child.py')], env=env, capture_output=True,
                       text=True, timeout=800)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert lines, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    out = json.loads(lines[-1])
    assert not out['failed'], json.dumps(out['failed'], indent=1)
    assert p.returncode == 0 and out['checks'] >= 100, (p.returncode, out['checks'])
