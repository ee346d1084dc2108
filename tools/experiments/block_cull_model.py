#!/usr/bin/env python3
"""Round 6: a numpy model of the block-level band cull's bound (csrc/gs_clip.hip bandBoundRejects), used BEFORE the kernels were written to
see which term of the bound is loose on the bench scene, and to predict what the block test, smaller blocks, size classes in the storage
order and a per-splat wave test would keep ("Block-level band cull").  CPU only (the Morton order is host code).
Findings on 1 M splats at 1920x1080, centre band (29, 34) of the cost-balanced 8-way cut: blocks of 256 Morton neighbours span 5.5 tile
rows and are judged by their largest splat (median block bound sqrt(lam) 0.24 against a median splat's 0.048: +-6.5 rows) -> 67 % of the
blocks kept; blocks of 64: 53 %; size classes (storage order by floor(log2 smax), then Morton): 55 % at 256; the per-splat bound: 25 % of
the splats are candidates, 44 % of the waves of 64 Morton neighbours hold one, 53 % of the blocks of 256.  usage: block_cull_model.py [N]"""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudagaussianrenderer_amd as gs
from cudagaussianrenderer_amd import scenes
import tempfile
N=int(sys.argv[1]) if len(sys.argv)>1 else 200000
W,H=1920,1080
with tempfile.TemporaryDirectory() as tmp:
    path=os.path.join(tmp,'s.ply'); scenes.write_synthetic_ply(path,N,sh_degree=0,seed=1234)
    ply=gs.parse_ply(path)
pos=ply['position'][:,:3].astype(np.float64); sr=ply['scaleAndRotation']
import ctypes as C
order=np.zeros(N,np.int32)
gs.lib().gs_morton_order(np.ascontiguousarray(ply['position']).ctypes.data_as(C.POINTER(C.c_float)),N,order.ctypes.data_as(C.POINTER(C.c_int32)))
pos=pos[order]; s=sr[order,:3].astype(np.float64)
smax=s.max(1)
nb=(N+255)//256
pad=nb*256-N
def blk(a,f,fill):
    a2=np.concatenate([a,np.full((pad,)+a.shape[1:],fill)]) if pad else a
    return f(a2.reshape(nb,256,*a.shape[1:]),axis=1)
cmin=blk(pos,np.min,1e30); cmax=blk(pos,np.max,-1e30); lam=blk(smax**2,np.max,0)
print('blocks',nb,'median block diag',np.median(np.linalg.norm(cmax-cmin,axis=1)),'mean',np.mean(np.linalg.norm(cmax-cmin,axis=1)), 'p90',np.percentile(np.linalg.norm(cmax-cmin,axis=1),90))
print('median sqrt(lam)',np.median(np.sqrt(lam)),'median smax',np.median(smax),'p99 smax',np.percentile(smax,99))
cams=[gs.camera_from_pose(p,q,W,H) for p,q in scenes.orbit_camera_poses(ply['boundsMin'],ply['boundsMax'],8)]
ty=(H+15)//16
def test(cam,rb,re,mode):
    v=np.array(cam.view[:]).reshape(4,4).T  # v[row][col]
    cotX,cotY=cam.fovCotangent[0],cam.fovCotangent[1]; ds,db=cam.depthScaleBias[0],cam.depthScaleBias[1]
    lo=np.zeros((nb,3)); hi=np.zeros((nb,3))
    for r in range(3):
        p0=v[r,:3][None,:]*cmin; p1=v[r,:3][None,:]*cmax
        lo[:,r]=v[r,3]+np.minimum(p0,p1).sum(1); hi[:,r]=v[r,3]+np.maximum(p0,p1).sum(1)
    zA=(-1-db)/ds; zB=(1-db)/ds
    zLo=np.maximum(lo[:,2],min(zA,zB)); zHi=np.minimum(hi[:,2],max(zA,zB))
    rej=zLo>zHi
    dn=-zHi; df=-zLo
    cyMax=np.minimum(cotY*hi[:,1]/np.where(hi[:,1]>=0,dn,df),1.001); cyMin=np.maximum(cotY*lo[:,1]/np.where(lo[:,1]<=0,dn,df),-1.001)
    cxMax=cotX*hi[:,0]/np.where(hi[:,0]>=0,dn,df); cxMin=cotX*lo[:,0]/np.where(lo[:,0]<=0,dn,df)
    rej|=(cyMin>1.001)|(cyMax<-1.001)|(cxMin>1.001)|(cxMax<-1.001)
    bump=(1/np.pi)*(2/W)*(2/H)
    if mode==0: jac=(cotX**2+cotY**2+2.02)/dn**2; half=3*np.sqrt(lam*jac+2*bump)
    elif mode==1:
        cy2=np.maximum(cyMin**2,cyMax**2); cx2=np.minimum(np.maximum(cxMin**2,cxMax**2),1.002)
        jac=(cotX**2+cx2+cotY**2+cy2)/dn**2; half=3*np.sqrt(lam*jac+2*bump)
    tps=H/16
    top=(cyMax+half+1)*0.5*tps; bot=(cyMin-half+1)*0.5*tps
    rej|=(top<rb)|(bot>re)
    return ~rej, half*0.5*tps, (cyMax-cyMin)*0.5*tps
bands=[(0, 17), (17, 24), (24, 29), (29, 34), (34, 38), (38, 43), (43, 50), (50, 68)]
for mode in (0,1):
  for b in bands[3:4]+bands[:1]:
    ks=[];hs=[];sp=[]
    for cam in cams:
        k,h,spn=test(cam,b[0],b[1],mode); ks.append(k.mean()); hs.append(np.median(h[k])); sp.append(np.median(spn[k]))
    print('mode',mode,'band',b,'kept frac',np.mean(ks),'median half rows',np.mean(hs),'median centre span rows',np.mean(sp), 'ideal', (b[1]-b[0])/ty)
# exact per-splat: fraction of splats whose true box reaches band (approx: centre row +- 3*smax*cotY/d*sqrt2)

print('---- size classes x block size (band (29,34) and (0,17)), kept fraction of SPLATS')
pos0=ply['position'][:,:3].astype(np.float64); s0=sr[:,:3].astype(np.float64); smax0=s0.max(1)
# morton rank of every splat
rank=np.empty(N,np.int64); rank[order]=np.arange(N)
def run(classes, B):
    global nb,pad,cmin,cmax,lam
    if classes:
        cls=np.floor(np.log2(smax0)).astype(np.int64)
        key=cls*(1<<40)+rank
    else: key=rank
    o=np.argsort(key,kind='stable')
    p=pos0[o]; sm=smax0[o]
    nb=(N+B-1)//B; pad=nb*B-N
    def blk(a,f,fill):
        a2=np.concatenate([a,np.full((pad,)+a.shape[1:],fill)]) if pad else a
        return f(a2.reshape(nb,B,*a.shape[1:]),axis=1)
    cmin=blk(p,np.min,1e30); cmax=blk(p,np.max,-1e30); lam=blk(sm**2,np.max,0)
    out=[]
    for b in (bands[3],bands[0]):
        ks=[]
        for cam in cams:
            k,h,spn=test(cam,b[0],b[1],1); ks.append(k.mean())
        out.append(round(float(np.mean(ks)),3))
    return out
for classes in (0,1):
    for B in (256,128,64):
        print('classes',classes,'block',B,run(classes,B))

print('---- per-splat bound, wave-64 any() in Morton order: fraction of waves that project (within kept 256-blocks)')
p=pos0[order]; sm=smax0[order]
def persplat(cam,rb,re):
    global nb,pad,cmin,cmax,lam
    nb=N; cmin=p; cmax=p; lam=sm**2
    k,h,spn=test(cam,rb,re,1)
    return k
for b in (bands[3],bands[0]):
    cand=[];wv=[];bl=[]
    for cam in cams:
        k=persplat(cam,b[0],b[1])
        cand.append(k.mean())
        kk=np.concatenate([k,np.zeros((-N)%256,bool)])
        wv.append(kk.reshape(-1,64).any(1).mean()); bl.append(kk.reshape(-1,256).any(1).mean())
    print('band',b,'candidate splats',round(np.mean(cand),3),'waves with a candidate',round(np.mean(wv),3),'256-blocks with a candidate',round(np.mean(bl),3))
