"""Host code under AddressSanitizer and UndefinedBehaviorSanitizer (-m "not gpu"; the device code has no
sanitizer on this hardware): the CPU oracle and the library's host-only helper ldpc_dense_to_qc, built by
`make -C oracle asan` and `make -C ldpc_amd/csrc asan`, run in a child Python with the sanitizer run-times
preloaded.  The deterministic-output double run that goes with it is test_full_size_batch_properties."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    out = subprocess.run(['gcc', '-print-file-name=' + name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.skipif(shutil.which('gcc') is None or shutil.which('g++') is None, reason='needs gcc and g++')
def test_oracle_and_host_helper_under_asan_ubsan():
    asan, ubsan = _runtime('libasan.so'), _runtime('libubsan.so')
    if not asan or not ubsan:
        pytest.skip('sanitizer run-times not installed')
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'asan'], stdout=subprocess.DEVNULL)
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'ldpc_amd', 'csrc'), 'asan'], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan + ':' + ubsan,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1',          # (the interpreter itself leaks at exit)
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1',
               LDPC_ORACLE_SO=os.path.join(ROOT, 'oracle', '_asan', 'libldpc_oracle.so'),
               LDPC_HOST_SO=os.path.join(ROOT, 'ldpc_amd', 'csrc', 'build', 'asan', 'libldpc_host.so'),
               OMP_NUM_THREADS='4')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'asan_child.py')], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0 and 'SANITIZED OK' in p.stdout, (p.returncode, p.stdout[-1500:], p.stderr[-4000:])
    assert 'ERROR: AddressSanitizer' not in p.stderr and 'runtime error' not in p.stderr, p.stderr[-4000:]
