"""CPU tests of the `charon` front end's command line (no GPU is touched: every case ends in the argument parser).
Flags, integer limits and exit codes follow src/main.cpp:49-65 and src/dehost_main.cpp:208-312 of the reference."""
import os
import subprocess

import pytest

from tests import util

EXE = os.path.join(util.ROOT, "charon_amd", "bin", "charon")
G = os.path.join(util.ROOT, "tests", "golden")


def run(args):
    p = subprocess.run([EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build()


def test_version_and_help():
    rc, out, _ = run(["--version"])
    assert rc == 0 and "charon" in out
    rc, out, _ = run(["dehost", "--help"])
    assert rc == 0
    for flag in ("--db", "--extract", "--prefix", "--chunk_size", "--lo_hi_threshold", "--num_reads_to_fit", "--dist", "--min_length",
                 "--min_quality", "--min_compression", "--confidence", "--host_unique_prop_lo_threshold", "--min_proportion_diff",
                 "--min_probability_diff", "--log", "--threads"):
        assert flag in out
    assert run([])[0] != 0 and run(["frobnicate"])[0] != 0


def test_parse_errors_exit_non_zero():
    fq, db = os.path.join(G, "cfg1_reads.fastq.gz"), os.path.join(G, "cfg1.idx")
    for args in (["dehost", fq],                                   # --db is required
                 ["dehost", "--db", db],                           # <fastaq> is required
                 ["dehost", "--db", db, "/no/such/file.fq"],       # ExistingFile
                 ["dehost", "--db", "/no/such/index", fq],         # ExistingPath
                 ["dehost", "--db", db, "--chunk_size", "256", fq],      # uint8_t
                 ["dehost", "--db", db, "-t", "300", fq],
                 ["dehost", "--db", db, "--confidence", "-1", fq],
                 ["dehost", "--db", db, "--num_reads_to_fit", "65536", fq],  # uint16_t
                 ["dehost", "--db", db, "--min_quality", "abc", fq],
                 ["dehost", "--db", db, "--frobnicate", fq],
                 ["dehost", "--db", db, "-p", G, fq],              # NonexistentPath
                 ["dehost", "--db", db, fq, fq, fq]):              # at most two read files
        rc, out, err = run(args)
        assert rc != 0 and out == "" and err, args
