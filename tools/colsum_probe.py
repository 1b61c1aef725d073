"""Column sums (bias gradients) of tall matrices: error against fp64 and time per call.  Run on the GPU box."""
import torch, sys
sys.path.insert(0,'.')
from tf2_gnn_amd import ops
dev=torch.device('cuda',0)
def t(fn, iters=10):
    for _ in range(2): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/iters*1000
for M,N in [(1151896,384),(30000,320),(30000,1280),(5001,7)]:
    x=torch.randn((M,N),device=dev)
    out=ops.colsum(x)
    ref=x.double().sum(0)
    print(M,N, float((out.double()-ref).abs().max()/ref.abs().max()), round(t(lambda: ops.colsum(x)),1),'us')
