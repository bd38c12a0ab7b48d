import torch, time, sys
sys.path.insert(0, '/root/repo')
from clip_dplm_amd import ops
dev='cuda'
B,L,d,V=512,256,480,33
ids=torch.randint(0,V,(B,L),device=dev)
dx=torch.randn(B*L,d,device=dev)
tab=torch.zeros(V,d,device=dev)
def run(): ops.embed_bwd(ids, dx, tab)
run(); torch.cuda.synchronize()
s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
print('embed_bwd us', s.elapsed_time(e)*100)
ref=torch.zeros(V,d,device=dev); ref.index_add_(0, ids.view(-1), dx)
tab.zero_(); run()
t1=tab.clone(); tab.zero_(); run(); print('bitwise reproducible', torch.equal(t1, tab))
print('maxerr', (tab-ref).abs().max().item(), ref.abs().max().item())
