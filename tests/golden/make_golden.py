#!/usr/bin/env python3
"""Regenerates the committed cfg1 fixtures (BASELINE config 1 shape, reduced to 200 reads to stay small):

  cfg1_host.fasta        seeded 10 kb random "host" genome
  cfg1.idx               2-category index: my.fasta -> microbial (bin 0), cfg1_host.fasta -> host (bin 1); k=19, w=41
  cfg1_reads.fastq.gz    200 synthetic 1 kb reads (45/45/10 % microbial / host / random, 5 % substitutions, phred 40)
  cfg1_expected.tsv      `dehost` TSV of the CPU ORACLE on those inputs (chunk_size 100, -t 1)
  cfg1_expected_extract.tsv   same with --extract microbial --num_reads_to_fit 20 (training path)

The reference binary cannot be built or run in this environment (its dependencies are fetched at configure time), so the
expected TSVs are ORACLE output, i.e. a regression pin for the two implementations in this repository, not reference output.
"""
import gzip
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tests import util  # noqa: E402


def main():
    r = util.rng(42)
    host = util.random_seq(r, 10000)
    with open(os.path.join(HERE, "cfg1_host.fasta"), "w") as f:
        f.write(">host_genome seeded random 10 kb\n")
        for i in range(0, len(host), 70):
            f.write(host[i:i + 70].decode() + "\n")
    micro = []
    name = None
    for line in open(os.path.join(HERE, "my.fasta")):
        line = line.strip()
        if line.startswith(">"):
            micro.append("")
        elif line:
            micro[-1] += line
    idx = po.Index.from_fasta([(os.path.join(HERE, "my.fasta"), "microbial"), (os.path.join(HERE, "cfg1_host.fasta"), "host")],
                              ["microbial", "host"])
    idx.store(os.path.join(HERE, "cfg1.idx"))
    reads = util.sample_reads(r, [m.encode() for m in micro[1:5]] + [host] * 4, 200, 1000, sub_rate=0.05, random_fraction=0.1)
    reads[7] = reads[7][:500] + b"NNNNNNNN" + reads[7][508:]
    reads[11] = b"ACGT" * 40
    reads[12] = host[100:118]
    with gzip.open(os.path.join(HERE, "cfg1_reads.fastq.gz"), "wt", compresslevel=9) as f:
        for i, s in enumerate(reads):
            f.write("@r%d synthetic read %d\n%s\n+\n%s\n" % (i, i, s.decode(), "I" * len(s)))
    fq = os.path.join(HERE, "cfg1_reads.fastq.gz")
    open(os.path.join(HERE, "cfg1_expected.tsv"), "w").write(idx.dehost_files(fq))
    open(os.path.join(HERE, "cfg1_expected_extract.tsv"), "w").write(idx.dehost_files(fq, run_extract=True, num_reads_to_fit=20))
    print("bin_size", idx.bin_size, "rows", len(reads))


if __name__ == "__main__":
    main()
