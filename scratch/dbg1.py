import sys, numpy as np, torch
sys.path.insert(0,'.')
from oracle import vnet_oracle as O
from vnet_tensorflow_amd import ops
dev=torch.device('cuda',0)
def g(a): return torch.as_tensor(np.ascontiguousarray(a),dtype=torch.float32).to(dev)
def report(tag, got, ref):
    got=got.detach().cpu().numpy().astype(np.float64); err=np.abs(got-ref)
    print(tag,'max',err.max(),'ref max',np.abs(ref).max())
    # per-channel, per-z, per-y, per-x error
    print('  per-channel max err', np.round(err.reshape(-1,err.shape[-1]).max(0),3)[:40])
    for ax,name in ((1,'z'),(2,'y'),(3,'x')):
        other=tuple(i for i in range(5) if i!=ax)
        print('  per-%s max err'%name, np.round(err.max(axis=other),3))
rng=np.random.default_rng(0)
# case 1: Cin=3
x=rng.standard_normal((1,6,6,18,3)); w=rng.standard_normal((5,5,5,3,16))*0.1; b=rng.standard_normal(16)
y=ops.conv(g(x),g(w),g(b),5,1); report('conv5 cin3', y, O.conv_nd_fwd(x,w,1)+b)
# case 2: down 64->128 at 8^3
x=rng.standard_normal((1,8,8,8,64)); w=rng.standard_normal((2,2,2,64,128))*0.1; b=rng.standard_normal(128)
y=ops.conv(g(x),g(w),g(b),2,2); report('down 64->128', y, O.conv_nd_fwd(x,w,2)+b)
# case 3: up 32->16
x=rng.standard_normal((1,4,8,16,32)); w=rng.standard_normal((2,2,2,16,32))*0.2; b=rng.standard_normal(16)
X,W,Bv=O.Var(x),O.Var(w),O.Var(b); yr=O.deconvolution(X,W,Bv,(8,16,32),2)
y=ops.conv_transpose2(g(x),g(w),g(b),(8,16,32)); report('up 32->16', y, yr.v)
