import sys, torch
sys.path.insert(0, '/root/repo')
import clip_dplm_amd as K
torch.manual_seed(0)
dev = torch.device('cuda:0')
m = K.ProteinRNACLIP(esm="esm2_t12_35M_UR50D").to(dev).train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
opt = K.FusedAdamW(m, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)
g = torch.Generator().manual_seed(1)
ids = torch.randint(4, 24, (512, 256), generator=g).to(dev)
rna = torch.randn(512, 256, 768, generator=g).to(dev)
ls = []
for i in range(60):
    opt.zero_grad(); l = m.loss(rna, ids); l.backward(); gn = opt.step()
    if i % 6 == 0 or i == 59:
        ls.append(round(l.item(), 4)); print(i, l.item(), float(gn.sqrt()), flush=True)
assert all(torch.isfinite(torch.tensor(ls)))
assert ls[-1] < ls[0] - 0.5, ls
print('OK loss decreased', ls[0], '->', ls[-1])
