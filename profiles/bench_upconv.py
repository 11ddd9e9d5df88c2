import sys, torch
sys.path.insert(0,'.')
from vnet_tensorflow_amd import ops
dev=torch.device('cuda',0)
for (P,ci) in ((8,256),(16,128),(32,64),(64,32)):
    x=torch.randn(1,P,P,P,ci,device=dev); w=torch.nn.Parameter(torch.randn(2,2,2,ci//2,ci,device=dev)*0.05); b=torch.zeros(ci//2,device=dev)
    with torch.no_grad():
        for _ in range(5): y=ops.conv_transpose2(x,w,b,(2*P,2*P,2*P))
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(50): y=ops.conv_transpose2(x,w,b,(2*P,2*P,2*P))
        e1.record(); torch.cuda.synchronize()
    print("up %d^3 %d->%d: %.1f us"%(P,ci,ci//2,e0.elapsed_time(e1)/50*1e3))
