import numpy as np
rng=np.random.default_rng(1)
def seg_cells(hl=16.0):
    th=rng.uniform(0,np.pi); fx,fy=rng.uniform(40,60,2)
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
    return out
segs=[seg_cells() for _ in range(4000)]
def cost(L, nb=32):
    tot=0; n=0
    for i in range(0,len(segs)-1,2):
        A,B=segs[i],segs[i+1]
        cells=A[:32]+B[:32]
        banks=[(x+L*y)%nb for x,y in cells]
        tot+=np.bincount(banks,minlength=nb).max(); n+=1
    return tot/n
# random lane=segment mapping at a given step: 64 cells from 64 different segments
def cost_random(nb=32):
    tot=0;n=0
    for t in range(300):
        idx=rng.integers(0,len(segs),64); j=rng.integers(0,24)
        banks=[(segs[i][j][0]+rng.integers(0,1000)+ (segs[i][j][1]+rng.integers(0,1000)))%nb for i in idx]
        tot+=np.bincount(banks,minlength=nb).max(); n+=1
    return tot/n
print("random f64 max load", cost_random(32), " u32:", cost_random(64))
for L in [1,3,5,7,9,11,13,15,17,19,21,23,25,27,29,31]:
    print(L, "f64 maxload %.2f"%cost(L,32), " u32(L+32k) %.2f / %.2f"%(cost(L,64),cost(L+32,64)))
