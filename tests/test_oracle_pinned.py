"""Pins the oracle against the reference's own stage functions run live (oracle/_ref/ref_driver,
compiled by oracle/Makefile from /root/reference/src).  Only runs where /root/reference exists
(the build container); on the GPU box the committed golden vectors carry the pin."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src"),
                                reason="reference sources only exist in the build container")


def _run_ref(driver, recs, k, threads=1):
    from debwt_amd import fasta
    d = tempfile.mkdtemp(prefix="pin_", dir="/tmp")
    try:
        fa, out = os.path.join(d, "in.fa"), os.path.join(d, "OUT")
        fasta.write_fasta(fa, recs)
        p = subprocess.run([driver, d, fa, out, str(k), str(threads)], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-800:]
        return (np.fromfile(out, dtype=np.uint64), np.fromfile(out + ".#", dtype=np.uint64),
                int(np.fromfile(out + ".$", dtype=np.uint64)[0]),
                np.fromfile(out + ".kmerInfo", dtype=np.uint64).reshape(-1, 2))
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("seed,k", [(1, 32), (2, 16), (3, 12), (4, 27)])
def test_oracle_equals_live_reference(oracle, seed, k):
    from debwt_amd import synth
    driver = oracle.build_ref()
    assert driver
    recs = synth.pan_genome(15000 + 1000 * seed, 2 + seed, seed=0xABC0 + seed)
    sym = oracle.sym_from_codes(recs)
    w, h, d, ki = _run_ref(driver, recs, k)
    ow, oh, od, _ = oracle.build_bwt(sym, k)
    assert np.array_equal(w, ow) and np.array_equal(h, oh) and d == od
    km, ct = oracle.kmer_count(sym, k)
    assert np.array_equal(ki[:, 0], km) and np.array_equal(ki[:, 1], ct)
