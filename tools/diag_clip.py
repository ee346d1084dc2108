import sys; sys.path.insert(0,'.')
import numpy as np
import cudagaussianrenderer_amd as gs
from oracle import gsref
from tests.gpu_helpers import StagedFrame
for (w,h) in [(1024,1024),(1920,1080)]:
    pos, sr, col = gs.generate_random_gaussians(200000, 1234, 0.01, 0.5)
    cam = gs.camera_from_bounds([-4]*3,[4]*3,w,h)
    f = StagedFrame(pos, sr, np.zeros_like(pos), cam, w, h); f.begin()
    xy,z,ell,con = f.run_clip()
    rxy,rz,rell,rcon = gsref.evaluate_splat_clip_data(pos, sr, cam, w, h)
    vis, rvis = xy[:,0]!=-128, rxy[:,0]!=-128
    print('vis mismatch', (vis!=rvis).sum(), 'visible', vis.sum())
    b = vis&rvis
    print('xy max abs', np.abs(xy[b]-rxy[b]).max(), 'z', np.abs(z[b]-rz[b]).max())
    cn = np.abs(rcon[b][:,:3]).max(1)
    rel = np.abs(con[b][:,:3]-rcon[b][:,:3]).max(1)/cn
    cond = rcon[b][:,0]*rcon[b][:,2]/(rcon[b][:,0]*rcon[b][:,2]-rcon[b][:,1]**2)
    print('conic rel err: max', rel.max(), 'p99.9', np.quantile(rel,0.999), 'cond max', cond.max(), 'max rel/cond', (rel/cond).max())
    er = np.abs(ell[b][:,2:]-rell[b][:,2:])/np.maximum(np.abs(rell[b][:,2:]),1e-9)
    print('extent rel err max', er.max(0))
    cross = np.abs(ell[b][:,0]*rell[b][:,1]-ell[b][:,1]*rell[b][:,0])
    an = rell[b][:,2]/np.maximum(rell[b][:,3],1e-20)
    print('cross max', cross.max(), 'cross*(an-1) max', (cross*(an-1)).max(), 'nan ell', np.isnan(ell).sum(), np.isnan(rell).sum())
