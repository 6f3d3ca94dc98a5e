"""dev (CPU only): what share of the moved source points could skip the culled search after looking at ONE sub-block -- the kd leaf
of their previous neighbour -- because their search ball lies inside that leaf's REGION (the cells of a kd order tile space)?
A bench-like scan pair at three pose errors, bounds of 1 / 1.5 / 2 x the true distance: 39 % / 25 % / 18 % -- not enough to pay for
evaluating that sub-block for every source outside the work queue (LAB_NOTES Part 0)."""
import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
from gloc3d_amd import synth
from scipy.spatial import cKDTree
w = synth.make_world(1001)
A = synth.lidar_scan(w, None, seed=1001)[:, :3].astype(np.float64)            # target
Tgt = synth.se3(1.5, (0.3, -0.2, 0.02))
B = synth.lidar_scan(w, Tgt, seed=1002)[:, :3].astype(np.float64)              # source, sensor frame
# exact relative pose: points of B in A's frame
Bw = B @ Tgt[:3,:3].T + Tgt[:3,3]
SB=16
def kd_order(P):
    n=len(P); idx=np.arange(n)
    # pad to SB*2^L
    L=0
    while SB*(1<<L) < n: L+=1
    regions={}
    out=[]
    def rec(ids, lo, hi, size):
        if size<=SB or len(ids)<=SB:
            out.append((ids, lo.copy(), hi.copy())); return
        pts=P[ids]
        ext=pts.max(0)-pts.min(0); a=int(np.argmax(ext))
        order=ids[np.argsort(pts[:,a],kind='stable')]
        half=size//2
        left=order[:half]; right=order[half:]
        if len(right)==0:
            out.append((ids, lo.copy(), hi.copy())); return
        m=0.5*(P[left][:,a].max()+P[right][:,a].min())
        hl=hi.copy(); hl[a]=m; lr=lo.copy(); lr[a]=m
        rec(left, lo, hl, half); rec(right, lr, hi, half)
    rec(idx, np.full(3,-np.inf), np.full(3,np.inf), SB*(1<<L))
    return out
t=time.time(); leaves=kd_order(A); print('leaves',len(leaves), time.time()-t)
leaf_of=np.empty(len(A),int); lo=np.empty((len(leaves),3)); hi=np.empty((len(leaves),3)); tlo=np.empty((len(leaves),3)); thi=np.empty((len(leaves),3))
for i,(ids,l,h) in enumerate(leaves):
    leaf_of[ids]=i; lo[i]=l; hi[i]=h; tlo[i]=A[ids].min(0); thi[i]=A[ids].max(0)
tree=cKDTree(A)
for name,err in (('converged (exact pose)',None),('2 cm / 0.05 deg off',synth.se3(0.05,(0.02,0.01,0.0))),('10 cm / 0.3 deg off',synth.se3(0.3,(0.08,-0.06,0.01)))):
    P = Bw if err is None else Bw @ err[:3,:3].T + err[:3,3]
    d,j=tree.query(P)
    # bound: the previous pass's neighbour under the new pose ~ here the true NN (optimistic) and 1.5x (pessimistic)
    for f in (1.0,1.5,2.0):
        r=(d*f)[:,None]
        lf=leaf_of[j]
        inside=((P-r>=lo[lf])&(P+r<=hi[lf])).all(1)
        # per wave of 128 consecutive (Hilbert-ish: use sorting by coarse morton) -- approximate with sorted by leaf of NN
        print(f'{name}: bound x{f}: sources whose ball lies inside the neighbour leaf region: {inside.mean():.3f}; median d {np.median(d):.3f} m; leaf region size median {np.median(np.minimum(hi-lo,50)[np.isfinite(hi-lo).all(1)],0)}')
