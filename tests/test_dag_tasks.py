"""The task list of the single-launch factorisation (csrc/dag_tasks.hpp) is a static schedule whose correctness
condition is purely combinatorial: walking the list in order, with the leaf lane advancing as far as the already
listed tasks allow, every dependency of every task is satisfied (nothing points forwards: the ticket order then cannot
deadlock), every tile receives every panel exactly once and ends final.  tests/native/check_dag_tasks.cpp checks that
on the host for 7 sizes x tall rows x depth x window x tail threshold x fill; no GPU involved."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_task_list_dependencies_point_backwards(tmp_path):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    exe = str(tmp_path / 'check_dag_tasks')
    src = os.path.join(ROOT, 'tests', 'native', 'check_dag_tasks.cpp')
    subprocess.check_call([hipcc, '-O1', '-std=c++17', '-o', exe, src])
    out = subprocess.run([exe], capture_output=True, text=True)
    sys.stdout.write(out.stdout[-2000:])
    assert out.returncode == 0 and '0 bad configurations' in out.stdout


@pytest.mark.timeout(600)
def test_rbfmath_against_ieee_and_libm(tmp_path):
    """csrc/rbfmath.hpp (the arithmetic core of the covariance assembly) compiled for the HOST: its division and square
    root must equal the IEEE operations bit for bit and its exp must stay within 1 ulp of libm's on 2 x 10^7 samples
    (tests/native/check_rbfmath.cpp) -- the device runs the same text."""
    exe = str(tmp_path / 'check_rbfmath')
    src = os.path.join(ROOT, 'tests', 'native', 'check_rbfmath.cpp')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-Wno-unknown-pragmas', '-o', exe, src])
    out = subprocess.run([exe], capture_output=True, text=True)
    sys.stdout.write(out.stdout[-2000:])
    assert out.returncode == 0 and 'division mismatches 0, sqrt mismatches 0, exp worst 1 ulp' in out.stdout
