"""The optimised CPU variant (oracle/fast.c: in-place fused fold, OpenMP) used for bench.py's second CPU-baseline
line must produce the faithful restatement's proof bit for bit, for any thread count."""
import numpy as np
import pytest

import oracle_lib as O

SEED = 0x5EED0000


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 12, 15])
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_fast_cpu_variant_matches_faithful_oracle(log_n, threads):
    table = O.gen_table(SEED + log_n, 0, 1 << log_n)
    claim, coeffs, ch = O.sumcheck_prove(table)
    fclaim, fcoeffs, fch = O.sumcheck_prove_fast(table, absorb=True, threads=threads)
    assert np.array_equal(fclaim, claim) and np.array_equal(fcoeffs, coeffs) and np.array_equal(fch, ch)


def test_fast_rejects_non_power_of_two():
    with pytest.raises(ValueError):
        O.sumcheck_prove_fast(np.zeros((3, 4), dtype=np.uint64))
