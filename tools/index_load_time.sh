set -e
ROOT=$PWD; W=$(mktemp -d /tmp/chload.XXXX); trap 'rm -rf $W' EXIT; cd $W
python3 - <<'PY'
import numpy as np
rng=np.random.default_rng(1)
for i,n in enumerate([400_000_000, 100_000_000]):
    a=rng.integers(0,4,n,dtype=np.uint8); s=np.frombuffer(b"ACGT",np.uint8)[a]
    with open(f"g{i}.fa","wb") as f:
        for c in range(0,n,50_000_000):
            f.write(b">chr%d\n"%c); f.write(s[c:c+50_000_000].tobytes()); f.write(b"\n")
open("in.tsv","w").write("g0.fa\thost\ng1.fa\tmicrobial\n")
r=rng.integers(0,4,(2000,300),dtype=np.uint8)
with open("r.fq","w") as f:
    for i in range(2000):
        f.write("@r%d\n%s\n+\n%s\n"%(i,np.frombuffer(b"ACGT",np.uint8)[r[i]].tobytes().decode(),"I"*300))
PY
( time $ROOT/charon_amd/bin/charon index -t 16 in.tsv ) 2>&1 | tail -5
ls -l in.tsv.idx || ls -l
( time $ROOT/charon_amd/bin/charon dehost r.fq --db in.tsv.idx -t 4 --log dehost.log > out.tsv ) 2>&1 | tail -5; tail -20 dehost.log; wc -l out.tsv
