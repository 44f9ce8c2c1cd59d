import os, random, subprocess, gzip, sys
random.seed(int(sys.argv[1]) if len(sys.argv)>1 else 1)
EXE="/tmp/asan/charon"
os.makedirs("/tmp/asan/w",exist_ok=True)
def rec_fq(i):
    n=random.choice([0,1,2,19,40,41,100,1000,5000])
    s="".join(random.choice("ACGTNacgtnRYKM") for _ in range(n))
    return "@r%d desc\n%s\n+\n%s\n"%(i,s,"".join(chr(random.randint(33,73)) for _ in range(n)))
def rec_fa(i):
    n=random.choice([0,1,50,500,3000])
    s="".join(random.choice("ACGTN") for _ in range(n))
    w=random.choice([60,70,10**6])
    return ">r%d\n"%i+"\n".join(s[j:j+w] for j in range(0,max(n,1),w))+"\n"
bad=0
for it in range(400):
    fq=random.random()<0.6
    body="".join((rec_fq if fq else rec_fa)(i) for i in range(random.randint(0,30)))
    b=bytearray(body.encode())
    # mutations
    for _ in range(random.choice([0,0,1,2,5])):
        if not b: break
        op=random.random(); p=random.randrange(len(b))
        if op<0.3: del b[p:p+random.randint(1,50)]
        elif op<0.6: b[p]=random.choice(b"\n\r@>+ \t\x00\xffXZ-*")
        elif op<0.8: b[p:p]=bytes(random.choice(b"\n\r@>+ACGT") for _ in range(random.randint(1,20)))
        else: b=b[:p]
    if random.random()<0.2: b=b.replace(b"\n",b"\r\n")
    ext=(".fastq" if fq else ".fasta")
    gz=random.random()<0.3
    path="/tmp/asan/w/t%d%s%s"%(it%8,ext,".gz" if gz else "")
    data=bytes(b)
    if gz:
        data=gzip.compress(data)
        if random.random()<0.3 and len(data)>10: data=data[:random.randrange(1,len(data))]
    open(path,"wb").write(data)
    p=subprocess.run([EXE,"_records",path,str(random.choice([1,3,1000])),str(random.choice([64,1000,1<<20]))],stdout=subprocess.PIPE,stderr=subprocess.PIPE,env=dict(os.environ,ASAN_OPTIONS="detect_leaks=0"))
    e=p.stderr.decode(errors="replace")
    if "Sanitizer" in e or "runtime error" in e or p.returncode<0:
        bad+=1; print("ITER",it,path,p.returncode); print(e[:3000]); 
        os.rename(path,"/tmp/asan/w/crash%d_%s"%(it,os.path.basename(path)))
        if bad>3: break
print("done, bad =",bad)
