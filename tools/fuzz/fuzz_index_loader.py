import os, random, subprocess, sys, struct
random.seed(int(sys.argv[1]))
TAIL = len(sys.argv) > 2 and sys.argv[2] == "tail"   # mutate only the sd_vector's two select_support_mcl blocks (the last 632 bytes of the fixture)
EXE="/tmp/asan/charon"
src=open("/root/repo/tests/golden/cfg1.idx","rb").read()
open("/tmp/asan/w/r.fq","w").write("@a\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\n"+"I"*48+"\n")
bad=0; msgs={}
for it in range(300):
    b=bytearray(src)
    hdr=random.random()<0.7
    for _ in range(random.choice([1,1,2,4])):
        if len(b)<2: break
        p=random.randrange(min(len(b),400) if hdr else len(b))
        if TAIL: p=random.randrange(max(0,len(b)-632),len(b))
        op=random.random()
        if op<0.4: b[p]=random.randrange(256)
        elif op<0.6: b[p:p+8]=struct.pack("<Q",random.choice([0,1,2**63,2**64-1,2**40,2**58,random.getrandbits(64)]))
        elif op<0.8: b=b[:p]
        else: del b[p:p+random.randint(1,16)]
    open("/tmp/asan/w/m.idx","wb").write(bytes(b))
    p=subprocess.run([EXE,"dehost","--db","/tmp/asan/w/m.idx","/tmp/asan/w/r.fq","--log","/tmp/asan/w/log"],stdout=subprocess.PIPE,stderr=subprocess.PIPE,env=dict(os.environ,ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=0:max_allocation_size_mb=8192"),timeout=120)
    e=p.stderr.decode(errors="replace")
    key=e.strip().split("\n")[-1][:90] if e.strip() else "rc=%d"%p.returncode
    msgs[key]=msgs.get(key,0)+1
    if "Sanitizer" in e or "runtime error" in e or p.returncode<0:
        bad+=1; print("ITER",it,p.returncode); print(e[:2500]); os.rename("/tmp/asan/w/m.idx","/tmp/asan/w/crash_%d.idx"%it)
        if bad>3: break
for k,v in sorted(msgs.items(),key=lambda x:-x[1])[:25]: print(v,k)
print("bad =",bad)
