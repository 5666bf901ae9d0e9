"""The native generator of the synthetic collections (include/debwt_synth.h) against its numpy definition
(debwt_amd/synth.py): same packed text, same separators, same census -- for every shape the bench and the tests use.
Host-only."""
import numpy as np
import pytest

from debwt_amd import api, synth
from debwt_amd import synth_native as SN

SHAPES = [
    (300_000, 4, 3, {}),                                        # pan-genome of chromosomes (the bench's shape)
    (250_000, 1, 1, {}),                                        # one record: no SNPs (synth.pan_genome(L, 1))
    (1_000_000, 1, 7, {}),                                      # synth.chromosomes
    (400_000, 3, 1, {}),                                        # synth.pan_genome(L, 3)
    (800_000, 2, 3, dict(lowcx_fraction=0.03, alu_copies=300)),  # distribution R
    (5000, 2, 2, {}), (1500, 3, 1, {}),                         # tiny: no repeat families below 2000 bases
    (400_000, 2, 2, dict(repeat_coverage=0.0, seed=synth.SEED_U)),   # distribution U
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"{s[0]}x{s[1]}x{s[2]}{'R' if s[3].get('alu_copies') else ''}")
def test_native_text_equals_numpy_definition(shape):
    gl, g, c, kw = shape
    s = SN.Synth(gl, g, c, threads=5, **kw)
    recs = s.records_numpy()
    words, n, sep = api.pack_records(recs)
    w2, census = s.words()
    assert n == s.n and len(recs) == s.nrec and np.array_equal(sep, s.sep())
    m = min(len(words), len(w2))
    assert np.array_equal(words[:m], w2[:m]) and not w2[m:].any() and not words[m:].any()
    want = np.zeros(4, dtype=np.int64)
    for r in recs:
        want += np.bincount(r, minlength=4)
    assert (want == census.astype(np.int64)).all()
    # any word range, any thread count
    s1 = SN.Synth(gl, g, c, threads=1, **kw)
    lo, hi = 3, min(1000, s.nwords)
    assert np.array_equal(s1.words(lo, hi)[0], w2[lo:hi])
    assert np.array_equal(s.codes(g - 1), np.concatenate(recs[(g - 1) * c:]))


def test_named_workloads_match_existing_generators():
    s = SN.Synth(2_000_000, 1, 4)
    assert all(np.array_equal(a, b) for a, b in zip(s.records_numpy(), synth.chromosomes(2_000_000, 4)))
    s = SN.Synth(500_000, 3, 1)
    assert all(np.array_equal(a, b) for a, b in zip(s.records_numpy(), synth.pan_genome(500_000, 3)))
    gl, g, c, _ = SN.WORKLOADS["pan10x3G"]
    assert gl * g == 30_000_000_000 and g * c == 240                # BASELINE configs[4]
    assert sum(synth.chromosome_lengths(gl, c)) == gl


def test_invalid_specs_are_rejected():
    with pytest.raises(RuntimeError):
        SN.Synth(100, 1, 4)                                          # records of <= 32 bases
