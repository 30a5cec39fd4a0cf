"""The C++ host layer (include/zkmi.hpp) replaying the reference's own unit tests for the path
(tests/cpp/test_reference_api.cpp) on the GPU.  The binary is built in-tree by
__graft_entry__.build() / `make -C tests/cpp`; here it is run once and its verdict checked."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cpp", "test_reference_api")


def test_reference_unit_tests_through_the_cpp_host_layer():
    assert os.path.exists(BIN), "tests/cpp/test_reference_api is not built (python -c 'import __graft_entry__ as g; g.build()')"
    p = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "0 failure(s)" in p.stdout
    for name in ("test_full_sumcheck_pass", "test_full_sumcheck_fail", "test_partial_sumcheck_pass", "test_gkr_sum_check",
                 "test_evaluate_4_variables", "test_operation_ws", "test_circuit", "test_transcript"):
        assert f"ok     {name}" in p.stdout
