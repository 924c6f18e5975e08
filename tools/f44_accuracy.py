"""fp32 forward error of Winograd F(m x m, 3x3) against the fp64 direct convolution on a 48 -> 48 layer (glorot filter, unit-normal
input): transforms, products and accumulation rounded to fp32 stage by stage as the kernels do.  python tools/f44_accuracy.py"""
import numpy as np
from fractions import Fraction as Fr
def toom(points, m, r):
    """Winograd F(m,r) matrices (AT, G, BT) from interpolation points (last = infinity), via the Toom-Cook construction in exact arithmetic."""
    n = m + r - 1
    pts = points[:n-1]
    # Vandermonde-based: AT (m x n), G (n x r), BT (n x n)
    AT = [[Fr(p)**i for p in pts] + [Fr(1) if i == m-1 else Fr(0)] for i in range(m)]
    G = []
    for k,p in enumerate(pts):
        den = Fr(1)
        for j,q in enumerate(pts):
            if j != k: den *= (Fr(p)-Fr(q))
        G.append([Fr(p)**i/den for i in range(r)])
    G.append([Fr(0)]*(r-1)+[Fr(1)])
    # BT: rows = coefficients of prod_{j!=k}(x - p_j) ; last row = prod_j (x-p_j)
    def polymul(a,b):
        c=[Fr(0)]*(len(a)+len(b)-1)
        for i,x in enumerate(a):
            for j,y in enumerate(b): c[i+j]+=x*y
        return c
    BT=[]
    for k in range(n-1):
        poly=[Fr(1)]
        for j,q in enumerate(pts):
            if j!=k: poly=polymul(poly,[-Fr(q),Fr(1)])
        BT.append(poly+[Fr(0)]*(n-len(poly)))
    poly=[Fr(1)]
    for q in pts: poly=polymul(poly,[-Fr(q),Fr(1)])
    BT.append(poly)
    f=lambda M: np.array([[float(x) for x in row] for row in M])
    return f(AT), f(G), f(BT)
def check(AT,G,BT,m,r):
    rng=np.random.default_rng(0)
    d=rng.standard_normal(m+r-1); g=rng.standard_normal(r)
    y=AT@((G@g)*(BT@d))
    ref=np.array([sum(d[i+k]*g[k] for k in range(r)) for i in range(m)])
    return np.abs(y-ref).max()
def conv_wino(x,w,AT,G,BT,m,dt):
    # x: [H+2, W+2, C] padded, w: [3,3,C,K]; returns [H,W,K]; everything rounded to dt after each stage as the kernel would (transforms in dt, products accumulated in dt via sequential K? use float32 matmul)
    H=x.shape[0]-2; W=x.shape[1]-2; C=x.shape[2]; K=w.shape[3]; n=m+2
    AT=AT.astype(dt);G=G.astype(dt);BT=BT.astype(dt)
    U=np.einsum('ai,ijck,bj->abck',G,w.astype(dt),G).astype(dt)           # n n C K
    out=np.zeros((H,W,K),dt)
    for ty in range(0,H,m):
        for tx in range(0,W,m):
            d=x[ty:ty+n,tx:tx+n].astype(dt)
            t=np.einsum('ai,ijc->ajc',BT,d).astype(dt)
            V=np.einsum('ajc,bj->abc',t,BT).astype(dt)
            M=np.einsum('abc,abck->abk',V,U).astype(dt)
            t2=np.einsum('ia,abk->ibk',AT,M).astype(dt)
            out[ty:ty+m,tx:tx+m]=np.einsum('ibk,jb->ijk',t2,AT).astype(dt)
    return out
def direct(x,w,dt):
    H=x.shape[0]-2; W=x.shape[1]-2
    out=np.zeros((H,W,w.shape[3]),dt)
    for ky in range(3):
        for kx in range(3):
            out+=np.einsum('hwc,ck->hwk',x[ky:ky+H,kx:kx+W].astype(dt),w[ky,kx].astype(dt)).astype(dt)
    return out
rng=np.random.default_rng(1)
C=K=48;H=W=24
x=np.zeros((H+2,W+2,C)); x[1:-1,1:-1]=rng.standard_normal((H,W,C))
lim=np.sqrt(6/(9*C+9*K)); w=rng.uniform(-lim,lim,(3,3,C,K))
ref=direct(x,w,np.float64)
sc=np.abs(ref).max()
print('direct fp32 err', np.abs(direct(x,w,np.float32)-ref).max()/sc)
for name,pts,m in [('F(2x2) pts 0,1,-1',[0,1,-1],2),('F(4x4) pts 0,1,-1,2,-2',[0,1,-1,2,-2],4),('F(4x4) pts 0,1,-1,1/2,-1/2',[0,1,-1,Fr(1,2),Fr(-1,2)],4),
                   ('F(4x4) pts 0,1,-1,1/2,-2',[0,1,-1,Fr(1,2),-2],4),('F(4x4) pts 0,1,-1,2,-1/2',[0,1,-1,2,Fr(-1,2)],4), ('F(3x3) pts 0,1,-1,2',[0,1,-1,2],3),('F(3x3) pts 0,1,-1,1/2',[0,1,-1,Fr(1,2)],3)]:
    AT,G,BT=toom(pts,m,3)
    e64=check(AT,G,BT,m,3)
    y=conv_wino(x,w,AT,G,BT,m,np.float32)
    err=np.abs(y-ref)
    print(f'{name:34s} exact-check {e64:.1e}  fp32 max err/max|y| {err.max()/sc:.2e}  rms {np.sqrt((err**2).mean())/sc:.2e}')
