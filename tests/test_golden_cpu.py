"""CPU tests on the committed cfg1 fixtures (tests/golden/make_golden.py): the oracle reproduces its own golden TSVs,
and the reference-faithful Elias-Fano probe mode gives the same rows as plain-word probes."""
import os
import subprocess

from tests import util

G = os.path.join(util.ROOT, "tests", "golden")


def test_oracle_reproduces_golden_tsv(oracle_lib):
    idx = oracle_lib.Index.load(os.path.join(G, "cfg1.idx"))
    assert (idx.bins, idx.bin_size, idx.categories, idx.host_index) == (2, 11957, ["microbial", "host"], 1)
    fq = os.path.join(G, "cfg1_reads.fastq.gz")
    assert idx.dehost_files(fq) == open(os.path.join(G, "cfg1_expected.tsv")).read()
    assert idx.dehost_files(fq, run_extract=True, num_reads_to_fit=20) == open(os.path.join(G, "cfg1_expected_extract.tsv")).read()
    # -t 4 processes a chunk in parallel but adds reads in input order in this restatement: same rows
    assert idx.dehost_files(fq, threads=4) == open(os.path.join(G, "cfg1_expected.tsv")).read()
    idx.use_ef(True)
    assert idx.dehost_files(fq) == open(os.path.join(G, "cfg1_expected.tsv")).read()
    idx.free()


def test_oracle_cli_matches_library(oracle_lib):
    exe = os.path.join(util.ROOT, "oracle", "charon_oracle")
    env = dict(os.environ, CHARON_KDE_TABLES=os.path.join(util.ROOT, "charon_amd", "data", "default_kde.txt"))
    out = subprocess.run([exe, "dehost", "--db", os.path.join(G, "cfg1"), os.path.join(G, "cfg1_reads.fastq.gz")], env=env,
                         stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out == open(os.path.join(G, "cfg1_expected.tsv")).read()
