"""What useful-lane fraction a compositing list granularity can reach: the product reach mask (csrc/reach_mask.h, host build) against an fp64 brute force
of the alpha test on a cfg3-like scene scaled to 480x272 (CPU only; termination not modelled).  python tools/exp_mask_ceiling.py"""
import sys, os, ctypes as C, subprocess, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gs_sdf_amd.synth as synth
from oracle import oracle
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out='/tmp/libreach_mask_host.so'
subprocess.check_call(["g++","-O2","-std=c++17","-shared","-fPIC","-ffp-contract=off","-I",ROOT+"/gs-sdf_amd/csrc",ROOT+"/tests/cpp/reach_mask_host.cpp","-o",out])
lib=C.CDLL(out); lib.reach_masks4x4.restype=None
W,H=480,272; N=int(1_000_000*(W*H)/(1920*1080))
sc=synth.make_scene(N,W,H,0,seed=0)
n=lambda t:t.detach().numpy()
means,quats=n(sc["means"]),n(sc["quats"]); scales=np.exp(n(sc["log_scales"])); opac_all=1/(1+np.exp(-n(sc["logit_opacities"])))
vm=n(synth.make_views(1)); K=n(sc["K"])
pr=oracle.projection_2dgs_fwd(means,quats,scales,vm,K,W,H)
tpg,ids,flat,offs=oracle.tile_encode(W,H,16,pr["means2d"],pr["radii"],pr["depths"],pr["camera_ids"],1)
tw=(W+15)//16; th=(H+15)//16
offs=np.asarray(offs).reshape(-1); T=tw*th
tile_of=np.repeat(np.arange(T), np.diff(np.append(offs, len(flat))))
idx=np.asarray(flat)
P=len(idx); print('pairs',P,'M',len(pr["gaussian_ids"]))
rt=pr["ray_transforms"].astype(np.float32); m2d=pr["means2d"].astype(np.float32); opac=opac_all[pr["gaussian_ids"]].astype(np.float32)
txy=np.stack([(tile_of%tw)*16,(tile_of//tw)*16],1).astype(np.float32)
masks=np.zeros(P,np.uint16)
a=[np.ascontiguousarray(v) for v in (rt[idx].reshape(P,9), m2d[idx], opac[idx], txy)]
lib.reach_masks4x4(C.c_int64(P),*(v.ctypes.data_as(C.c_void_p) for v in a),masks.ctypes.data_as(C.c_void_p))
px=np.arange(16)[None,:]+0.5
tot_set=tot_kept_blocks=tot_keep=0; tot_88=0; tot_22=0
row_len=np.zeros((T,16),np.int64); row_len_exact=np.zeros((T,16),np.int64); quad_len=np.zeros((T,64),np.int64)
yy,xx=np.meshgrid(np.arange(16),np.arange(16),indexing='ij')
bit=4*(2*(yy>>3)+(xx>>3))+2*((yy>>2)&1)+((xx>>2)&1)
for s0 in range(0,P,20000):
    sl=slice(s0,min(P,s0+20000)); ii=idx[sl]; t=txy[sl]
    X=(t[:,0:1].astype(np.float64)+px)[:,None,:].repeat(16,1); Y=(t[:,1:2].astype(np.float64)+px)[:,:,None].repeat(16,2)
    Mm=rt[ii].astype(np.float64)
    hu=X[...,None]*Mm[:,None,None,2,:]-Mm[:,None,None,0,:]; hv=Y[...,None]*Mm[:,None,None,2,:]-Mm[:,None,None,1,:]
    z=np.cross(hu,hv)
    with np.errstate(divide='ignore',invalid='ignore'):
        s=z[...,:2]/z[...,2:3]; g3=(s**2).sum(-1)
    g2=2.0*((X-m2d[ii,0:1,None])**2+(Y-m2d[ii,1:2,None])**2)
    sig=0.5*np.where(np.isfinite(g3),np.minimum(g3,g2),g2)
    alpha=np.minimum(0.999,opac[ii].astype(np.float64)[:,None,None]*np.exp(-sig))
    keep=(z[...,2]!=0)&(alpha>=1/255)&(X<W)&(Y<H)
    mk=masks[sl]
    setb=((mk[:,None].astype(np.int64)>>np.arange(16)[None])&1).astype(bool)
    kb=np.zeros((len(ii),16),bool)
    for b in range(16): kb[:,b]=(keep&(bit[None]==b)).any(axis=(1,2))
    assert not (kb&~setb).any()
    tot_set+=setb.sum(); tot_kept_blocks+=kb.sum(); tot_keep+=keep.sum()
    # 2x2 blocks ceiling
    k22=keep.reshape(len(ii),8,2,8,2).any(axis=(2,4)); tot_22+=k22.sum()
    tl=tile_of[sl]
    np.add.at(row_len,(tl[:,None].repeat(16,1),np.arange(16)[None].repeat(len(ii),0)),setb.astype(np.int64))
    np.add.at(row_len_exact,(tl[:,None].repeat(16,1),np.arange(16)[None].repeat(len(ii),0)),kb.astype(np.int64))
    # quad q of a wave: wave w = 8x8 quadrant (2 (y>>3) + (x>>3)), 16 quads of 2x2 pixels inside it
    qy,qx=np.meshgrid(np.arange(8),np.arange(8),indexing='ij')
    qid=(16*(2*(qy>>2)+(qx>>2))+4*(qy&3)+(qx&3)).reshape(-1)
    np.add.at(quad_len,(tl[:,None].repeat(64,1),qid[None].repeat(len(ii),0)),k22.reshape(len(ii),64).astype(np.int64))
print('kept pixels',tot_keep)
print('mask blocks (4x4)',tot_set,'-> useful lanes',tot_keep/(16*tot_set))
print('exact 4x4 blocks',tot_kept_blocks,'-> useful lanes ceiling',tot_keep/(16*tot_kept_blocks))
print('exact 2x2 blocks',tot_22,'-> useful lanes ceiling',tot_keep/(4*tot_22))

# wave iterations of the compositing loop (every wave runs max over its lists' lengths; batches of 256 staged splats ignored)
rows_now=row_len.reshape(T,4,4).max(axis=2).sum(); rows_exact=row_len_exact.reshape(T,4,4).max(axis=2).sum(); quads=quad_len.reshape(T,4,16).max(axis=2).sum()
print('wave iterations: row lists, current mask',rows_now,'| row lists, exact 4x4 mask',rows_exact,'| quad (2x2) lists, exact mask',quads)
print('lane-iterations (64 x wave iterations) per kept pixel:',64*rows_now/tot_keep,64*rows_exact/tot_keep,64*quads/tot_keep)
