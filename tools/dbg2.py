import sys, os
sys.path.insert(0, 'dl-unet_amd'); sys.path.insert(0, '.')
import torch, numpy as np
import network
from oracle import prng, oracle_c
S = int(sys.argv[1]) if len(sys.argv)>1 else 188
B = int(sys.argv[2]) if len(sys.argv)>2 else 2
params = prng.make_params(0)
net = network.Unet(); net.load_state_dict({k: torch.from_numpy(v) for k,v in params.items()}); net=net.to('cuda:0')
x = prng.make_input(1,B,S); dl = prng.make_cotangent(2,(B,2,S-184,S-184))
y = net(torch.from_numpy(x).cuda()); y.backward(torch.from_numpy(dl).cuda())
p64 = {k:v.astype(np.float64) for k,v in params.items()}
rl, rg = oracle_c.unet_fwd_bwd(p64, x.astype(np.float64), dlogits=dl.astype(np.float64))
l32, g32 = oracle_c.unet_fwd_bwd(params, x, dlogits=dl)
def ne(a,b): return np.abs(np.asarray(a,np.float64)-b).max()/max(np.abs(b).max(),1e-300)
print("logits: hip %.3g  c_f32 %.3g" % (ne(y.detach().cpu().numpy(), rl), ne(l32, rl)))
for k,p in net.named_parameters():
    print("%-18s hip %.3g   c_f32 %.3g   hip-vs-c_f32 %.3g" % (k, ne(p.grad.cpu().numpy(), rg[k]), ne(g32[k], rg[k]), ne(p.grad.cpu().numpy(), g32[k].astype(np.float64))))
