"""Model of the Line tile kernel's LDS bank-pair loads when a work item's segments are SORTED before they walk (round 4).

The kernel is bound by the LDS atomic pipe: one ds_add_u64 per visited cell, 64 segments of a wave on 64 unrelated cells,
cost ~ 2 cycles x (lanes on the fullest of the 32 bank pairs) (tools/ubench_lds_patterns.hip).  Idea tested here: give the
window a row pitch that is a multiple of 32 cells, so that the bank pair of a cell is x mod 32; x-major segments walking the
same way then keep their bank offsets for the whole walk, and a wave whose 64 segments were dealt two per bank pair would stay
conflict-free.  Segments are classed (x-major by sign of the x step; y-major by slope bucket and sign), sorted by the bank pair
of their first cell and dealt round-robin to the waves of their class.

Result (4096 segments of half length 16 in a 128 x 96 window): mean load of the fullest bank pair
    arrival order                 5.0   (pitch 108 or 128)
    sorted, x-major classes       2.9-3.0   (Poisson-uneven banks: some waves still get 3-4 lanes on one pair)
    sorted, y-major classes       4.4-5.0   (the bank changes only at minor steps, at a rate that is the segment's slope:
                                             the dealt order is gone after a few steps; more slope buckets = classes
                                             too small to fill waves)
    all                           3.5-3.85
i.e. 25-30 % fewer LDS cycles for an in-kernel sort of every 4096 records (three barriers per round in a kernel that has
none in its walk) and a 128-cell pitch -- not built.  A second, transposed window for the y-major half does not fit the LDS.
Run: python tools/line_bank_model2.py"""
import numpy as np
rng=np.random.default_rng(3)
P=128
def seg(hl=16.0):
    th=rng.uniform(0,np.pi); fx=rng.uniform(18,110); fy=rng.uniform(18,78)
    px,py=np.float32(hl)*np.float32(np.cos(th)),np.float32(hl)*np.float32(np.sin(th))
    x0,y0=int(np.round(fx-px)),int(np.round(fy-py)); x1,y1=int(np.round(fx+px)),int(np.round(fy+py))
    dx,dy=abs(x1-x0),abs(y1-y0); sx=1 if x0<x1 else -1; sy=1 if y0<y1 else -1
    err=dx-dy; cx,cy=x0,y0; out=[]
    while True:
        out.append((cx,cy))
        if cx==x1 and cy==y1: break
        e2=2*err
        if e2>-dy: err-=dy; cx+=sx
        if e2<dx: err+=dx; cy+=sy
    return out,(dx,dy,sx,sy)
N=4096
S=[seg() for _ in range(N)]
def wave_cost(idx, pitch):
    # sum over steps of max bank-pair load among active lanes
    L=max(len(S[i][0]) for i in idx); tot=0
    for j in range(L):
        b=[(S[i][0][j][0]+pitch*S[i][0][j][1])%32 for i in idx if j<len(S[i][0])]
        tot+=np.bincount(b,minlength=32).max()
    return tot,L
def total(groups,pitch):
    c=0;steps=0
    for g in groups:
        t,L=wave_cost(g,pitch); c+=t; steps+=L
    return c,steps
# baseline: arrival order, pitch 108
base=[list(range(i,i+64)) for i in range(0,N,64)]
for pitch in (108,128):
    c,s=total(base,pitch); print("random order pitch",pitch,"avg maxload %.2f"%(c/s),"wave-steps",s)
# scheme: class = xmajor? (sx) : (4 + slope bucket*2 + (sx*sy>0)) ; sort by (class, bank of first cell), deal within class
def scheme(nslope, pitch=128):
    keys=[]
    for i,(cells,(dx,dy,sx,sy)) in enumerate(S):
        b0=(cells[0][0]+pitch*cells[0][1])%32
        if dx>=dy: cls=(0 if sx>0 else 1)
        else:
            sl=min(int(nslope*dx/max(dy,1)),nslope-1)
            # minor step direction in x matters: bank += sx at minor steps
            cls=2+sl*2+(0 if sx>0 else 1)
        keys.append((cls,b0,i))
    keys.sort()
    groups=[]
    from itertools import groupby
    for cls,it in groupby(keys,key=lambda k:k[0]):
        mem=[k[2] for k in it]; nw=(len(mem)+63)//64
        for k in range(nw): groups.append(mem[k::nw])
    return groups
for ns in (1,2,4,8):
    g=scheme(ns); c,s=total(g,128)
    # split x-major vs y-major stats
    print("scheme slope buckets",ns,"avg maxload %.2f"%(c/s),"wave-steps",s,"waves",len(g))
print("--- per class (8 slope buckets)")
keys={}
for i,(cells,(dx,dy,sx,sy)) in enumerate(S):
    b0=(cells[0][0]+128*cells[0][1])%32
    if dx>=dy: cls=(0 if sx>0 else 1)
    else: cls=2+min(int(4*dx/max(dy,1)),3)*2+(0 if sx>0 else 1)
    keys.setdefault(cls,[]).append((b0,i))
for cls in sorted(keys):
    mem=[i for b,i in sorted(keys[cls])]; nw=(len(mem)+63)//64
    g=[mem[k::nw] for k in range(nw)]
    c,s=total(g,128); print(cls,len(mem),"avg maxload %.2f"%(c/s))
    if cls==0:
        t,L=wave_cost(g[0],128)
        for j in (0,1,5,10,20,30):
            b=[(S[i][0][j][0]+128*S[i][0][j][1])%32 for i in g[0] if j<len(S[i][0])]
            print(" step",j,"lanes",len(b),"max",np.bincount(b,minlength=32).max())
