"""GPU: a few seeds of the parity fuzzer (tests/fuzz_parity.py) inside the suite -- random genome, index shape, pass
knobs and read mix per seed, every record of the three layouts against the oracle.  The long campaigns are run by hand
(`python tests/fuzz_parity.py --seeds 0:700`; profiles/r3/fuzz_in_contract_700.txt)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [3, 36, 54, 57, 101, 202, 303, 404, 100400, 100401, 100402, 100403,
                                  200000, 200001, 200002])  # (from 100000 on: reads up to 512 bases, -m -1, FASTQ -> SAM; from 200000 on: the device-pointer entry points too)
def test_fuzz_seed(oracle, seed):
    import sbwt_amd as sb
    import fuzz_parity
    sb.device_count()
    problems, desc = fuzz_parity.run_seed(seed, sb, oracle, n_reads=8000, verbose=False)
    assert not problems, desc + "\n" + "\n".join(problems)


@pytest.mark.parametrize("seed,lists_longer", [(34, True), (49, True), (110, True), (120, True), (37, False), (46, False),
                                               (78, False), (143, False), (100610, True), (100643, True), (200100, False), (200101, False), (300222, False)])
def test_fuzz_seed_beyond_the_sort_depth(oracle, seed, lists_longer):
    """Reads of L bases on an index whose suffix order stops before L + interval - 1 bases (upstream's builder stops at
    sort_len 256: reads of 194 bases and more).  Upstream's depth-first locate then reports a row every time its chain
    meets a sampled row -- more hits than the interval has rows (searcher.hpp:464 compares with ==, and the count can jump
    over it), and a one-row interval whose first position the filters drop goes on to a second one (searcher.hpp:440 breaks
    the loop only).  The library reports the same lists: the longer ones in slots behind all others (seeds 34, 49, 110,
    120: through k_multi_chain, k_multi_bfs and k_multi), the one-row reads through the depth-first locate (the others;
    each of these seeds failed before that).  Seed 300222: an index sampled every 128 rows, every result forced into wide[]
    and wide[] too small at first -- k_multi met records that were never written (zero rows) and read out of bounds."""
    import sbwt_amd as sb
    import fuzz_parity
    sb.device_count()
    problems, desc = fuzz_parity.run_seed(seed, sb, oracle, n_reads=12000, verbose=False, beyond=True)
    assert not problems, desc + "\n" + "\n".join(problems)
    if lists_longer:
        assert not desc.endswith(": 0"), desc


@pytest.mark.parametrize("seed", [0, 1, 5, 7, 11])
def test_fuzz_builder_seed(oracle, seed):
    """tests/fuzz_builder.py: a random FASTA through `sbwt build` on the GPU and through the oracle's builder -- the index
    files byte for byte (a genome with a duplication longer than sort_len: the mapping only), then FASTQ -> SAM on the
    GPU's files (400 seeds: profiles/r3/fuzz_builder_400.txt)"""
    import sbwt_amd as sb
    import fuzz_builder
    sb.device_count()
    problems, desc = fuzz_builder.run_seed(seed, sb, oracle, verbose=False)
    assert not problems, desc + "\n" + "\n".join(problems)
