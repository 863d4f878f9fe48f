#!/usr/bin/env python3
"""LDS layout of k_describe's transposed intermediate (orbx_describe.hip: TCS, c_reach.gbase, c_reach.vlane), chosen by a bank model.

Model: 32 banks x 4 bytes; a b32 / b16 access is served 32 lanes at a time, a b128 access 8 lanes at a time; a group costs as many
cycles as its busiest bank has distinct dwords.  (It tracks the measured SQ_LDS_IDX_ACTIVE of the kernel within ~10 %.)  Free
parameters: the column stride T (dwords), a few dwords of gap in front of each group of four columns (the whole block must stay
inside seven workgroups per CU), and which lane of the vertical pass takes which (column pair, first row) run.  The script searches
gaps at random and lane orders by pair swaps and prints the tables that are pasted into orbx_describe.hip; the kernel's
static_asserts check that the pasted runs cover exactly the disc.  python tools/desc_lds_layout.py"""
import numpy as np, math, random, itertools, sys
def hh():
    out=[]
    for c in range(40):
        ax=abs(c-18) if c<37 else 99
        best=0
        for d in range(19):
            tx=2*ax-1 if ax>0 else 0; ty=2*d-1 if d>0 else 0
            if tx*tx+ty*ty<=4*340: best=d
        out.append(best if c<37 else 0)
    return out
H=hh()
items=[]
for r in range(43):
    gs=[c>>2 for c in range(37) if abs(r-21)<=H[c]+3]
    for g in range(min(gs),max(gs)+1): items.append((r,g))
nitems=len(items)
while len(items)<384: items.append(items[nitems-1])
vl=[]
for cp in range(19):
    h=max(H[2*cp],H[2*cp+1]); top=(18-h)&~1; bot=18+h
    r0=top
    while r0<=bot:
        rs=38-12 if r0+12>38 else r0
        vl.append((cp,rs)); r0+=12
nv=len(vl)
while len(vl)<64: vl.append(vl[nv-1])
def cyc(dw, G=32):
    # dw: list of dword indices (or None) for 64 lanes; groups of G lanes
    tot=0
    for g0 in range(0,64,G):
        banks={}
        for d in dw[g0:g0+G]:
            if d is None: continue
            banks.setdefault(d%32,set()).add(d)
        tot+=max([len(v) for v in banks.values()]+[1])
    return tot
def cost(T,Gb,items,vl,parts=False):
    # Gb[g]: dword base of group g; column c at Gb[c>>2] + (c&3)*T
    hw=0
    for it in range(6):
        chunk=items[it*64:(it+1)*64]
        for jj in range(4):
            hw+=cyc([Gb[cg]+jj*T+(r>>1) for r,cg in chunk])
    vr=0
    for k in range(9):
        for col in (0,1):
            vr+=cyc([Gb[(2*cp+col)>>2]+((2*cp+col)&3)*T+(r0>>1)+k for cp,r0 in vl])
    hr=0
    for it in range(6):
        chunk=items[it*64:(it+1)*64]
        # b128: 8 lanes per pass, 4 dwords each
        t=0
        for g0 in range(0,64,8):
            banks={}
            for r,cg in chunk[g0:g0+8]:
                for d in range(r*12+cg, r*12+cg+4): banks.setdefault(d%32,set()).add(d)
            t+=max(len(v) for v in banks.values())
        hr+=t
    vw=0
    for i in range(12):
        vw+=cyc([((r0+i)*40+2*cp)//4 for cp,r0 in vl])
    if parts: return hw,vr,hr,vw
    return hw+vr+hr+vw
T=23
Gb=[g*4*T for g in range(10)]
print("current",cost(T,Gb,items,vl,True))
# search gaps
best=None
random.seed(1)
for T in (23,22):
    budget=(5851-2080)//4-40*T
    for trial in range(3000):
        gaps=[0]*10
        rem=budget
        for g in range(1,10):
            x=random.randint(0,min(rem,7)); gaps[g]=x; rem-=x
        Gb=[]; acc=0
        for g in range(10):
            acc+=gaps[g]; Gb.append(g*4*T+acc)
        c=cost(T,Gb,items,vl)
        if best is None or c<best[0]: best=(c,T,list(Gb),list(gaps)); 
print("best gaps",best, cost(best[1],best[2],items,vl,True))
# now optimise V lane order and H item order by random swaps (hill climbing) for the best layout
T,Gb=best[1],best[2]
def vcost(vl): 
    vr=0
    for k in range(9):
        for col in (0,1):
            vr+=cyc([Gb[(2*cp+col)>>2]+((2*cp+col)&3)*T+(r0>>1)+k for cp,r0 in vl])
    vw=0
    for i in range(12):
        vw+=cyc([((r0+i)*40+2*cp)//4 for cp,r0 in vl])
    return vr+vw
cur=list(vl); cc=vcost(cur)
for step in range(20000):
    i,j=random.randrange(64),random.randrange(64)
    if (i<32)==(j<32): continue
    cur[i],cur[j]=cur[j],cur[i]
    c=vcost(cur)
    if c<=cc: cc=c
    else: cur[i],cur[j]=cur[j],cur[i]
print("V after swaps",cc)
def hcost(items):
    hw=0;hr=0
    for it in range(6):
        chunk=items[it*64:(it+1)*64]
        for jj in range(4):
            hw+=cyc([Gb[cg]+jj*T+(r>>1) for r,cg in chunk])
        for g0 in range(0,64,8):
            banks={}
            for r,cg in chunk[g0:g0+8]:
                for d in range(r*12+cg, r*12+cg+4): banks.setdefault(d%32,set()).add(d)
            hr+=max(len(v) for v in banks.values())
    return hw+hr
curh=list(items); ch=hcost(curh)
for step in range(60000):
    i,j=random.randrange(384),random.randrange(384)
    curh[i],curh[j]=curh[j],curh[i]
    c=hcost(curh)
    if c<=ch: ch=c
    else: curh[i],curh[j]=curh[j],curh[i]
print("H after swaps",ch)
print("T =", T)
print("gbase (dwords) =", Gb)
print("vlane (cp, r0) =", ", ".join("{%d, %d}" % (a, b) for a, b in cur))
