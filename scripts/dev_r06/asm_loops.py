#!/usr/bin/env python3
"""Dev tool (round 6): the loops of one kernel in a device listing (make -C csrc asm K=11), by back edge, with what
scripts/kernel_resources.py cannot say -- WHERE the waits sit: per loop its instruction count, buffer loads, scratch accesses and
the s_waitcnt vmcnt(...) in program order.  A rolling prefetch reads vmcnt(10), vmcnt(11), vmcnt(11) ...; vmcnt(10) ... vmcnt(0)
inside the first 33 instructions of a loop is a prefetch drained at the latch (profiles/r06/NOTES.md).
usage: asm_loops.py [mangled-kernel-name-prefix] [listing]   (default: k_zbwd<11, 2, FROM_T, !HAS_GD, !NT, WALK> in tu_taps_11.gfx950.s)"""
import re,sys
KN=sys.argv[1] if len(sys.argv)>1 else '_Z6k_zbwdILi11ELi2ELb1ELb0ELb0ELb1EE'
txt=open(sys.argv[2] if len(sys.argv)>2 else 'differentiable-point-clouds_amd/csrc/tu_taps_11.gfx950.s').read()
m=re.search(r"^(%s\w+):.*?\n(.*?)\.end_amdhsa_kernel" % KN, txt, re.S|re.M)
body=m.group(2).split('\n')
labels={}
for i,l in enumerate(body):
    mm=re.match(r"^(\.LBB\d+_\d+):",l)
    if mm: labels[mm.group(1)]=i
loops=[]
for i,l in enumerate(body):
    mm=re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)",l)
    if mm:
        t=mm.group(1) or mm.group(2)
        if t in labels and labels[t]<i: loops.append((labels[t],i,t))
for a,b,t in sorted(loops):
    n=sum(1 for l in body[a:b] if l.startswith('\t') and not l.strip().startswith(('.',';')))
    loads=sum(1 for l in body[a:b] if 'buffer_load' in l)
    if n<400 or loads==0: continue
    sc=[(j,body[j].strip().split(';')[0][:60]) for j in range(a,b) if 'scratch_' in body[j]]
    vm=[body[j].strip().replace('s_waitcnt ','') for j in range(a,b) if 's_waitcnt vmcnt' in body[j]]
    print("loop %s %d-%d instrs %d loads %d scratch %d waits %s" % (t,a,b,n,loads,len(sc),vm[:16]))
    for x in sc[:24]: print("      ",x)
