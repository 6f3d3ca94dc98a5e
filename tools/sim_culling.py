"""Developer model of the culled 1-NN search (CPU only: numpy + scipy): which boxes can a source point not rule out?

Real synthetic scans, exact nearest-neighbour distances from scipy's kd-tree, the kernel's own structure (sorted
points cut into 128-point chunks and 16-point sub-blocks with bounding boxes, waves of 128 consecutive sorted sources,
a chunk processed when some source's box distance is within its bound, a (source, sub-block) item evaluated likewise)
-- and, per ORDERING of the points, the counts the kernel's cost follows: candidate / processed chunks per wave,
listed sources, evaluated items.  With the Hilbert order the model reproduces what the kernel's counters measured in
round 2 (7 processed chunks per wave, ~50 listed sources per chunk, ~3 sub-blocks per source).  It is how round 3
found that the target's ORDER, not the kernel, was the thing to change: finer curve cells,
2-D curves, anisotropic cells, regrouped sources and a curve / kd hybrid do nothing; kd order of the target cuts the
items by 27-46 % on same-place, 4 m-apart and different-world pairs.

    python tools/sim_culling.py                      (about two minutes)
    python tools/sim_culling.py --real SCAN.bin      only the curve-order / kd-order comparison, on a real KITTI scan
                                                     (x y z i float32): its even points as the target, its odd points
                                                     moved by a small pose + 2 cm noise as the source.  On the
                                                     reference's sample scan (s2s_libtorch/000000.bin, 124 668 points;
                                                     build container only, not committed): processed chunks per wave
                                                     9.5 -> 5.8, evaluated items 523 -> 296.
"""
import os, sys, time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gloc3d_amd import synth
from scipy.spatial import cKDTree

def hilbert_keys(ix, bits):
    # Skilling transpose, 3-D, ix: [n,3] uint64 in [0, 2^bits)
    X=[ix[:,0].astype(np.uint64).copy(), ix[:,1].astype(np.uint64).copy(), ix[:,2].astype(np.uint64).copy()]
    M=np.uint64(1)<<np.uint64(bits-1)
    Q=M
    while Q>1:
        P=Q-np.uint64(1)
        for i in range(3):
            m=(X[i]&Q)!=0
            X[0]=np.where(m, X[0]^P, X[0])
            t=(X[0]^X[i])&P
            t=np.where(m, np.uint64(0), t)
            X[0]^=t; X[i]^=t
        Q>>=np.uint64(1)
    X[1]^=X[0]; X[2]^=X[1]
    t=np.zeros_like(X[0]); Q=M
    while Q>1:
        t=np.where((X[2]&Q)!=0, t^(Q-np.uint64(1)), t); Q>>=np.uint64(1)
    for i in range(3): X[i]^=t
    key=np.zeros_like(X[0])
    for b in range(bits):
        for i in range(3):
            key|=((X[i]>>np.uint64(b))&np.uint64(1))<<np.uint64(3*b+(2-i))
    return key

def sort_scan(P, cell, bits):
    mn=P.min(0)
    ix=np.clip(np.floor((P-mn)/cell),0,(1<<bits)-1).astype(np.uint64)
    k=hilbert_keys(ix,bits)
    o=np.argsort(k,kind='stable')
    return P[o]

def boxes(P, m):
    n=len(P); nb=(n+m-1)//m
    pad=np.concatenate([P, np.repeat(P[-1:], nb*m-n,0)])
    pad=pad.reshape(nb,m,3)
    return pad.min(1), pad.max(1)

def lb2(p, lo, hi):
    e=np.maximum(np.maximum(lo[None]-p[:,None], p[:,None]-hi[None]),0)
    return (e*e).sum(-1)

w=synth.make_world(1001)
A=synth.lidar_scan(w,None,seed=1001)[:,:3].astype(np.float64)
T=synth.se3(5.0,(0.5,-0.3,0.1))
B=synth.lidar_scan(w,T,seed=1002)[:,:3].astype(np.float64)
Bw=B@T[:3,:3].T+T[:3,3]
# a slightly wrong pose as in mid-ICP: 3 cm / 0.1 deg off
E=synth.se3(0.1,(0.03,-0.02,0.01)); Bm=Bw@E[:3,:3].T+E[:3,3]
tree=cKDTree(A)
def _kd_order_simple(P):
    n=len(P); out=np.empty(n,np.int64)
    sys.setrecursionlimit(10000)
    def rec(ids, lo):
        m=len(ids)
        if m<=16:
            out[lo:lo+m]=ids; return
        pts=P[ids]
        ax=np.argmax(pts.max(0)-pts.min(0))
        unit=128 if m>128 else 16
        half=((m//2+unit-1)//unit)*unit
        if half>=m: half=m-unit if m>unit else m//2
        part=np.argpartition(pts[:,ax],half-1 if half>0 else 0)
        rec(ids[part[:half]],lo); rec(ids[part[half:]],lo+half)
    rec(np.arange(n),0)
    return out

if len(sys.argv) > 2 and sys.argv[1] == "--real":
    R=np.fromfile(sys.argv[2],np.float32).reshape(-1,4)[:,:3].astype(np.float64)
    rng=np.random.default_rng(0)
    Tr=synth.se3(1.0,(0.25,-0.15,0.02))
    tgt=R[0::2]; srcp=R[1::2]@Tr[:3,:3].T+Tr[:3,3]+rng.normal(0,0.02,(len(R[1::2]),3))
    tr=cKDTree(tgt)
    Sh=sort_scan(srcp,0.25,10); dd,_=tr.query(Sh)
    for tag,Tg in (("curve order",sort_scan(tgt,0.25,10)),("kd order",tgt[_kd_order_simple(tgt)])):
        slo,shi=boxes(Tg,16); clo,chi=boxes(Tg,128); cand=[];proc=[];lst=[];items=[]
        for wv in range(0,len(Sh)//128,4):
            S=Sh[wv*128:(wv+1)*128]; bd=(dd[wv*128:(wv+1)*128]+0.03)**2
            wl,wh=S.min(0),S.max(0)
            e=np.maximum(np.maximum(clo-wh, wl-chi),0); lbw=(e*e).sum(-1)
            c=np.nonzero(lbw<=bd.max())[0]
            L=lb2(S,clo[c],chi[c])<=bd[:,None]
            cand.append(len(c)); proc.append(L.any(0).sum()); lst.append(L.sum()); items.append((lb2(S,slo,shi)<=bd[:,None]).sum())
        print(f"{sys.argv[2]} ({len(R)} points), target in {tag}: candidate chunks {np.mean(cand):.1f} processed {np.mean(proc):.1f} listings {np.mean(lst):.0f} items {np.mean(items):.0f} per wave")
    sys.exit(0)

for cell,bits in ((0.25,10),(0.0625,12),(0.015,14),(0.004,16)):
    As=sort_scan(A,cell,bits); Bs=sort_scan(Bm,cell,bits)
    slo,shi=boxes(As,16); clo,chi=boxes(As,128)
    d,_=tree.query(Bs)       # exact NN distance (ideal bound)
    # warm bound: distance to the NN of the previous pass (pose off by E): approx d_prev = NN dist under true pose, new dist to same target
    dprev,jprev=tree.query(sort_scan(Bw,cell,bits)) if False else (None,None)
    sel=np.arange(0,len(Bs),16)
    for name,slack in (("ideal",0.0),("+3cm",0.03)):
        bound=(d[sel]+slack)**2
        ns=(lb2(Bs[sel],slo,shi)<=bound[:,None]).sum(1)
        nc=(lb2(Bs[sel],clo,chi)<=bound[:,None]).sum(1)
        print(f"cell {cell} bits {bits} {name}: sub-blocks/source {ns.mean():.2f}  chunks/source {nc.mean():.2f}")
    # wave level: groups of 128 sources; wave box + max bound -> candidate chunks, processed chunks (any source passes)
    nw=len(Bs)//128
    cand=[];proc=[];listed=[]
    for wv in range(0,nw,8):
        S=Bs[wv*128:(wv+1)*128]; bd=(d[wv*128:(wv+1)*128]+0.03)**2
        wl,wh=S.min(0),S.max(0)
        e=np.maximum(np.maximum(clo-wh, wl-chi),0); lbw=(e*e).sum(-1)
        c=np.nonzero(lbw<=bd.max())[0]
        L=lb2(S,clo[c],chi[c])<=bd[:,None]
        cand.append(len(c)); proc.append(L.any(0).sum()); listed.append(L.sum()/max(L.any(0).sum(),1))
    print(f"   per wave: candidate chunks {np.mean(cand):.1f} processed {np.mean(proc):.1f} listed/processed chunk {np.mean(listed):.1f}  sub-block diag mean {np.linalg.norm(shi-slo,axis=1).mean():.3f} chunk diag {np.linalg.norm(chi-clo,axis=1).mean():.3f}")

print("---- target-aligned grouping (cell 0.25, bits 10) ----")
cell,bits=0.25,10
As=sort_scan(A,cell,bits); clo,chi=boxes(As,128); slo,shi=boxes(As,16)
treeS=cKDTree(As)
# previous pass: pose Bw (true), this pass: Bm (moved by E): prev NN index known
dp,jp=treeS.query(Bw)
d,_=treeS.query(Bm)
bound=(np.linalg.norm(Bm-As[jp],axis=1))**2     # distance to the previous NN under the new pose
grp=jp//128
order=np.argsort(grp,kind='stable')
cand=[];proc=[];lst=[];sizes=[];items=[]
ug=np.unique(grp)
for g in ug[::8]:
    idx=np.nonzero(grp==g)[0]
    S=Bm[idx]; bd=bound[idx]
    wl,wh=S.min(0),S.max(0)
    e=np.maximum(np.maximum(clo-wh, wl-chi),0); lbw=(e*e).sum(-1)
    c=np.nonzero(lbw<=bd.max())[0]
    L=lb2(S,clo[c],chi[c])<=bd[:,None]
    cand.append(len(c)); proc.append(L.any(0).sum()); lst.append(L.sum()); sizes.append(len(idx))
print(f"groups {len(ug)} mean size {np.mean(sizes):.1f}; per group: candidate chunks {np.mean(cand):.1f} processed {np.mean(proc):.1f} listings {np.mean(lst):.1f} ({np.mean(lst)/np.mean(sizes):.2f}/source)")
# baseline with the same bound definition: source-order waves
Bs_order=np.argsort(hilbert_keys(np.clip(np.floor((Bm-Bm.min(0))/cell),0,1023).astype(np.uint64),10),kind='stable')
cand=[];proc=[];lst=[]
for wv in range(0,len(Bm)//128,8):
    idx=Bs_order[wv*128:(wv+1)*128]
    S=Bm[idx]; bd=bound[idx]
    wl,wh=S.min(0),S.max(0)
    e=np.maximum(np.maximum(clo-wh, wl-chi),0); lbw=(e*e).sum(-1)
    c=np.nonzero(lbw<=bd.max())[0]
    L=lb2(S,clo[c],chi[c])<=bd[:,None]
    cand.append(len(c)); proc.append(L.any(0).sum()); lst.append(L.sum())
print(f"source-order waves: candidate chunks {np.mean(cand):.1f} processed {np.mean(proc):.1f} listings {np.mean(lst):.1f} ({np.mean(lst)/128:.2f}/source)")

print("---- kd-order (median splits on the widest axis, aligned to 128 / 16) ----")
def kd_order(P):
    n=len(P); idx=np.arange(n)
    out=np.empty(n,np.int64)
    stack=[(0,n,idx)]
    res=[]
    def rec(ids, lo):
        m=len(ids)
        if m<=16:
            out[lo:lo+m]=ids; return
        pts=P[ids]
        ax=np.argmax(pts.max(0)-pts.min(0))
        unit=128 if m>128 else 16
        half=((m//2+unit-1)//unit)*unit
        if half>=m: half=m-unit if m>unit else m//2
        part=np.argpartition(pts[:,ax],half-1 if half>0 else 0)
        l=ids[part[:half]]; r=ids[part[half:]]
        rec(l,lo); rec(r,lo+half)
    import sys; sys.setrecursionlimit(10000)
    rec(idx,0)
    return out
t0=time.time(); o=kd_order(A); print("kd build (numpy)",time.time()-t0)
Ak=A[o]
slo,shi=boxes(Ak,16); clo,chi=boxes(Ak,128)
tree=cKDTree(A)
ob=kd_order(Bm); Bk=Bm[ob]
d,_=tree.query(Bk)
sel=np.arange(0,len(Bk),16)
for name,slack in (("ideal",0.0),("+3cm",0.03)):
    bound=(d[sel]+slack)**2
    ns=(lb2(Bk[sel],slo,shi)<=bound[:,None]).sum(1)
    nc=(lb2(Bk[sel],clo,chi)<=bound[:,None]).sum(1)
    print(f"kd {name}: sub-blocks/source {ns.mean():.2f}  chunks/source {nc.mean():.2f}")
nw=len(Bk)//128
cand=[];proc=[];listed=[];items=[]
for wv in range(0,nw,8):
    S=Bk[wv*128:(wv+1)*128]; bd=(d[wv*128:(wv+1)*128]+0.03)**2
    wl,wh=S.min(0),S.max(0)
    e=np.maximum(np.maximum(clo-wh, wl-chi),0); lbw=(e*e).sum(-1)
    c=np.nonzero(lbw<=bd.max())[0]
    L=lb2(S,clo[c],chi[c])<=bd[:,None]
    cand.append(len(c)); proc.append(L.any(0).sum()); listed.append(L.sum()/max(L.any(0).sum(),1))
    it=(lb2(S,slo,shi)<=bd[:,None]).sum()
    items.append(it)
print(f"   per wave: candidate chunks {np.mean(cand):.1f} processed {np.mean(proc):.1f} listed/processed chunk {np.mean(listed):.1f} items/wave {np.mean(items):.0f}  sub-block diag mean {np.linalg.norm(shi-slo,axis=1).mean():.3f} chunk diag {np.linalg.norm(chi-clo,axis=1).mean():.3f}")
# same items/wave metric for the hilbert order
As=sort_scan(A,0.25,10); Bs=sort_scan(Bm,0.25,10); slo,shi=boxes(As,16); d2,_=tree.query(Bs)
items=[]
for wv in range(0,nw,8):
    S=Bs[wv*128:(wv+1)*128]; bd=(d2[wv*128:(wv+1)*128]+0.03)**2
    items.append((lb2(S,slo,shi)<=bd[:,None]).sum())
print(f"hilbert items/wave {np.mean(items):.0f}")

print("---- hybrid: Hilbert runs of R points, kd inside ----")
def hybrid_order(P, R):
    Ps=sort_scan(P,0.25,10)
    out=np.empty_like(Ps)
    for a in range(0,len(Ps),R):
        seg=Ps[a:a+R]
        out[a:a+len(seg)]=seg[kd_order(seg)]
    return out
def wave_stats(Bsrc, Atgt, tag):
    slo,shi=boxes(Atgt,16); clo,chi=boxes(Atgt,128)
    d,_=tree.query(Bsrc)
    nw=len(Bsrc)//128
    cand=[];proc=[];listed=[];items=[]
    for wv in range(0,nw,8):
        S=Bsrc[wv*128:(wv+1)*128]; bd=(d[wv*128:(wv+1)*128]+0.03)**2
        wl,wh=S.min(0),S.max(0)
        e=np.maximum(np.maximum(clo-wh, wl-chi),0); lbw=(e*e).sum(-1)
        c=np.nonzero(lbw<=bd.max())[0]
        L=lb2(S,clo[c],chi[c])<=bd[:,None]
        cand.append(len(c)); proc.append(L.any(0).sum()); listed.append(L.sum())
        items.append((lb2(S,slo,shi)<=bd[:,None]).sum())
    print(f"{tag}: candidate {np.mean(cand):.1f} processed {np.mean(proc):.1f} listings/wave {np.mean(listed):.0f} items/wave {np.mean(items):.0f}")
Ah=sort_scan(A,0.25,10); Bh=sort_scan(Bm,0.25,10)
wave_stats(Bh,Ah,"hilbert/hilbert")
wave_stats(Bh,Ak,"src hilbert / tgt kd")
wave_stats(Bk,Ak,"kd/kd")
for R in (8192,4096,2048,1024,512):
    wave_stats(hybrid_order(Bm,R),hybrid_order(A,R),f"hybrid R={R} both")
wave_stats(Bh,hybrid_order(A,8192),"src hilbert / tgt hybrid 8192")

print("---- anisotropic keys: z cells coarser ----")
def hilbert2d_keys(ix, iy, bits):
    # standard xy2d
    n=np.uint64(1)<<np.uint64(bits)
    x=ix.astype(np.uint64).copy(); y=iy.astype(np.uint64).copy()
    d=np.zeros_like(x)
    s=n>>np.uint64(1)
    while s>0:
        rx=((x&s)>0).astype(np.uint64); ry=((y&s)>0).astype(np.uint64)
        d+=s*s*((np.uint64(3)*rx)^ry)
        # rotate
        m=(ry==0)
        fl=m&(rx==1)
        x=np.where(fl,(s-np.uint64(1))-(x&(s-np.uint64(1)))|(x&~(s-np.uint64(1)))*0 + 0*x, x) if False else x
        xs=x&(s-np.uint64(1)); ys=y&(s-np.uint64(1))
        xs2=np.where(fl,(s-np.uint64(1))-xs,xs); ys2=np.where(fl,(s-np.uint64(1))-ys,ys)
        xs3=np.where(m,ys2,xs2); ys3=np.where(m,xs2,ys2)
        x=xs3; y=ys3
        s>>=np.uint64(1)
    return d
def sort2d(P, cell, bits, zcell=None, zbits=0):
    mn=P.min(0)
    ix=np.clip(np.floor((P[:,0]-mn[0])/cell),0,(1<<bits)-1).astype(np.uint64)
    iy=np.clip(np.floor((P[:,1]-mn[1])/cell),0,(1<<bits)-1).astype(np.uint64)
    k=hilbert2d_keys(ix,iy,bits)
    if zcell:
        iz=np.clip(np.floor((P[:,2]-mn[2])/zcell),0,(1<<zbits)-1).astype(np.uint64)
        k=(k<<np.uint64(zbits))|iz
    return P[np.argsort(k,kind='stable')]
for cell in (0.25,0.5,1.0):
    wave_stats(sort2d(Bm,cell,10),sort2d(A,cell,10),f"2-D hilbert xy cell {cell}")
wave_stats(sort2d(Bm,0.25,10,1.0,4),sort2d(A,0.25,10,1.0,4),"2-D hilbert 0.25 + z minor 1.0")
def sort3d_aniso(P, cell, zcell):
    mn=P.min(0)
    ix=np.clip(np.floor((P-mn)/np.array([cell,cell,zcell])),0,1023).astype(np.uint64)
    return P[np.argsort(hilbert_keys(ix,10),kind='stable')]
for zc in (1.0,4.0):
    wave_stats(sort3d_aniso(Bm,0.25,zc),sort3d_aniso(A,0.25,zc),f"3-D hilbert xy 0.25, z cell {zc}")

print("---- other pairs: hilbert vs kd target order (sources hilbert) ----")
def pair_stats(Asrc_world, Atgt, tag):
    global tree
    tree=cKDTree(Atgt)
    Bh=sort_scan(Asrc_world,0.25,10)
    wave_stats(Bh, sort_scan(Atgt,0.25,10), tag+" tgt hilbert")
    wave_stats(Bh, Atgt[kd_order(Atgt)], tag+" tgt kd")
w2=synth.make_world(2002)
N=synth.lidar_scan(w2,synth.se3(7.0,(1.5,-0.7,0)),seed=5001)[:,:3].astype(np.float64)
pair_stats(Bm, N, "negative (other world)")
T4=synth.se3(2.0,(4.0,0.8,0.0))
C=synth.lidar_scan(w,T4,seed=1003)[:,:3].astype(np.float64)
Cw=C@T4[:3,:3].T+T4[:3,3]
pair_stats(Cw@E[:3,:3].T+E[:3,3], A, "positive 4 m apart")

print("---- kd targets: source waves in hilbert order vs grouped by the previous NN's chunk ----")
tree=cKDTree(A)
Ak=A[kd_order(A)]; clo,chi=boxes(Ak,128); slo,shi=boxes(Ak,16)
treeK=cKDTree(Ak)
dp,jp=treeK.query(Bw)                      # previous pass (true pose): NN position in kd order
bound=(np.linalg.norm(Bm-Ak[jp],axis=1))**2  # this pass's initial bound: distance to the previous NN
def grp_stats(groups, tag):
    cand=[];proc=[];lst=[];items=[];sz=[]
    for idx in groups:
        S=Bm[idx]; bd=bound[idx]
        wl,wh=S.min(0),S.max(0)
        e=np.maximum(np.maximum(clo-wh, wl-chi),0); lbw=(e*e).sum(-1)
        c=np.nonzero(lbw<=bd.max())[0]
        L=lb2(S,clo[c],chi[c])<=bd[:,None]
        cand.append(len(c)); proc.append(L.any(0).sum()); lst.append(L.sum()); sz.append(len(idx))
        items.append((lb2(S,slo,shi)<=bd[:,None]).sum())
    n=np.sum(sz)
    print(f"{tag}: groups {len(groups)} (of mean {np.mean(sz):.0f}); per 128 sources: candidate {np.sum(cand)/n*128:.1f} processed {np.sum(proc)/n*128:.1f} listings {np.sum(lst)/n*128:.0f} items {np.sum(items)/n*128:.0f}")
oh=np.argsort(hilbert_keys(np.clip(np.floor((Bm-Bm.min(0))/0.25),0,1023).astype(np.uint64),10),kind='stable')
grp_stats([oh[w*128:(w+1)*128] for w in range(0,len(Bm)//128,8)],"hilbert-order waves")
g=jp//128
og=np.argsort(g,kind='stable'); gs=g[og]
starts=np.nonzero(np.r_[1,np.diff(gs)])[0]; ends=np.r_[starts[1:],len(gs)]
grp_stats([og[a:b] for a,b in list(zip(starts,ends))[::8]],"grouped by prev-NN chunk")
ok=kd_order(Bm)
grp_stats([ok[w*128:(w+1)*128] for w in range(0,len(Bm)//128,8)],"kd-order waves")
