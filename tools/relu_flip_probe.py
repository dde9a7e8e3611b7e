import sys, random, numpy as np, torch, os
sys.path.insert(0,'/root/repo')
from omr_a2s_multimodal_transformer_amd import synthetic as syn
from oracle import ref_cpu as R
import torch.nn.functional as F
V=50; w2i,i2w=syn.make_vocab(V)
x,xl,y_in,y_out=syn.synthetic_unimodal_batch(2,64,160,12,V,w2i["<sos>"],w2i["<eos>"],seed=9)
allow={1,2}
saved={}
orig=R.conv_block
def conv_block(sd,p,x,stride,drop=None,dp=0.5):
    if 'conv_blocks.3.' not in p: return orig(sd,p,x,stride,drop,dp)
    pos=random.randint(1,3)
    tag=str(x.dtype)
    def keep(name,t):
        t.retain_grad(); saved[(tag,name)]=t; return t
    x=keep('in',x)
    a=keep('c1',F.conv2d(x,sd[p+"conv1.weight"],sd[p+"conv1.bias"],padding=1))
    x=keep('r1',F.relu(a))
    if pos==1: x=R.mix_dropout(drop,x,dp)
    a=keep('c2',F.conv2d(x,sd[p+"conv2.weight"],sd[p+"conv2.bias"],padding=1))
    x=keep('r2',F.relu(a))
    if pos==2: x=R.mix_dropout(drop,x,dp)
    x=keep('in_out',R.instance_norm(x))
    a=keep('c3',F.conv2d(x,sd[p+"conv3.weight"],sd[p+"conv3.bias"],padding=1,stride=stride))
    x=keep('r3',F.relu(a))
    if pos==3: x=R.mix_dropout(drop,x,dp)
    return x
R.conv_block=conv_block
def run(dt):
    sd={k:v.to(dt).requires_grad_(True) for k,v in syn.seeded_state_dict(syn.transformer_shapes(V),41).items()}
    def fn(site,kind,p,shape,channel):
        if site not in allow: return torch.ones((),dtype=dt)
        return syn.seeded_dropout_mask(5,site,p,shape,channel).to(dt)
    random.seed(3)
    torch.set_default_dtype(dt)
    lo=R.transformer_forward(sd,x.to(dt),xl,y_in,R.OracleCfg(),64,160,drop=R.DropPlan(fn))
    R.ce_loss(lo,y_out).backward()
    torch.set_default_dtype(torch.float32)
run(torch.float32); run(torch.float64)
for name in ['r3','c3','in_out','r2','c2','r1','c1','in']:
    a=saved[('torch.float32',name)]; b=saved[('torch.float64',name)]
    fe=((a.double()-b).norm()/b.norm()).item()
    ge=((a.grad.double()-b.grad).norm()/b.grad.norm()).item()
    print(f"{name:7s} fwd err {fe:.2e} grad err {ge:.2e}")
r2=saved[('torch.float64','r2')]
v=r2.var(dim=(2,3),unbiased=False); mu=r2.mean(dim=(2,3))
print("min var", v.min().item(), "planes with var<1e-6:", (v<1e-6).sum().item(), "of", v.numel())
print("planes all-zero:", (r2.abs().amax(dim=(2,3))==0).sum().item())
# per-plane gradient error at r2
a=saved[('torch.float32','r2')].grad.double(); b=saved[('torch.float64','r2')].grad
pe=((a-b).flatten(2).norm(dim=2)/ (b.flatten(2).norm(dim=2)+1e-30))
idx=pe.flatten().topk(5)
print("worst planes", idx.values, idx.indices, "their var", v.flatten()[idx.indices], "mean", mu.flatten()[idx.indices])
print("their grad norm share", (b.flatten(2).norm(dim=2).flatten()[idx.indices]/b.norm()))
