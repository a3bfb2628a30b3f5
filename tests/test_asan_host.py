"""Host-side AddressSanitizer pass over the C ABI (sanitizers run on the CPU build only - GPU ASan is not
available on the pool).  ``make asan-check`` builds libmorgana_hip with its HOST code instrumented and drives the entry points'
argument validation, workspace / split planning, error formatting and launch set-up from tests/asan/abi_host_check.c; the oracle's C
leg gets the same treatment from oracle/asan_check.c."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(directory, target):
    proc = subprocess.run(['make', '-C', os.path.join(REPO, directory), '-j4', target], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          universal_newlines=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-4000:]
    assert 'AddressSanitizer' not in proc.stdout and 'runtime error' not in proc.stdout, proc.stdout[-4000:]
    return proc.stdout


def test_c_abi_host_code_under_address_sanitizer():
    out = _make(os.path.join('morgana_amd', 'csrc'), 'asan-check')
    assert '0 unexpected results' in out


def test_oracle_c_leg_under_address_sanitizer():
    out = _make('oracle', 'asan')
    assert 'oracle_c asan check ok' in out
