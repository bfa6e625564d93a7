"""GPU: a few seeds of the parity fuzzer (tests/fuzz_parity.py) inside the suite -- random genome, index shape, pass
knobs and read mix per seed, every record of the three layouts against the oracle.  The long campaigns are run by hand
(`python tests/fuzz_parity.py --seeds 0:700`; profiles/r3/fuzz_in_contract_700.txt)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [3, 36, 54, 57, 101, 202, 303, 404])
def test_fuzz_seed(oracle, seed):
    import sbwt_amd as sb
    import fuzz_parity
    sb.device_count()
    problems, desc = fuzz_parity.run_seed(seed, sb, oracle, n_reads=8000, verbose=False)
    assert not problems, desc + "\n" + "\n".join(problems)
