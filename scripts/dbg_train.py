import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests'))
import numpy as np, torch
import golden_cases as gc
from amyloid_yolo_paper_amd import cfg_gen, parse_config, synth
from amyloid_yolo_paper_amd.models import Darknet
name, C_, S, B, seed = gc.TRAIN_CASES[0]
z = np.load(f'tests/golden/{name}.npz')
cfg = cfg_gen.write_cfg(C_, '/tmp/cfgd'); defs = parse_config.parse_model_config(cfg)
params = synth.synth_params(defs, seed=7)
m = Darknet(cfg, precision='fp32')
sd = m.state_dict()
for i, p in params.items():
    for k, nme in (("weight", f"conv_{i}.weight"), ("bias", f"conv_{i}.bias"), ("gamma", f"batch_norm_{i}.weight"), ("beta", f"batch_norm_{i}.bias"), ("mean", f"batch_norm_{i}.running_mean"), ("var", f"batch_norm_{i}.running_var")):
        if k in p: sd[f"module_list.{i}.{nme}"].copy_(torch.from_numpy(p[k]))
m = m.to('cuda').train()
tg = torch.from_numpy(gc.train_targets(B, C_, S, seed)); x = torch.from_numpy(gc.model_inputs(S, B, 10))
loss, out = m(x, tg); loss.backward()
print('loss', loss.item(), float(z['loss']))
for li in (105, 104, 93, 81, 80, 73, 42, 2, 1, 0):
    conv = m.module_list[li][0]
    g = conv.weight.grad.cpu().numpy(); ref = z[f'gw{li}']
    if ref.shape != g.shape: g = g.reshape(-1)[:: max(1, g.size // 65536)]
    print(li, 'w relerr', np.abs(g-ref).max()/np.abs(ref).max(), 'scale', np.abs(ref).max())
    if conv.bias is not None:
        print('   b relerr', np.abs(conv.bias.grad.cpu().numpy()-z[f'gb{li}']).max()/np.abs(z[f'gb{li}']).max())
    else:
        bn = m.module_list[li][1]
        print('   gamma relerr', np.abs(bn.weight.grad.cpu().numpy()-z[f'ggamma{li}']).max()/np.abs(z[f'ggamma{li}']).max(), 'beta', np.abs(bn.bias.grad.cpu().numpy()-z[f'gbeta{li}']).max()/np.abs(z[f'gbeta{li}']).max())

# ---- compare dhead and a few activation gradients against the CPU oracle's autograd
from oracle.darknet_oracle import OracleDarknet
from amyloid_yolo_paper_amd.train_engine import train_forward
o = OracleDarknet(cfg); o.set_params(params); o.require_grad()
xo = x.clone()
# hook: keep grads of layer outputs
import torch.nn.functional as F
loss_o, _ = o.forward(xo, tg, train_bn=True, collect=True)
acts = {li: o.layer_outputs[li] for li in range(len(o.layer_outputs)) if o.defs[li]['type'] == 'convolutional'}
for t in acts.values(): t.retain_grad()
loss_o.backward()
m2 = Darknet(cfg, precision='fp32'); m2.load_state_dict(m.state_dict()); 
sd2 = m2.state_dict()
for i, p in params.items():
    for k, nme in (("weight", f"conv_{i}.weight"), ("bias", f"conv_{i}.bias"), ("gamma", f"batch_norm_{i}.weight"), ("beta", f"batch_norm_{i}.bias"), ("mean", f"batch_norm_{i}.running_mean"), ("var", f"batch_norm_{i}.running_var")):
        if k in p: sd2[f"module_list.{i}.{nme}"].copy_(torch.from_numpy(p[k]))
m2 = m2.to('cuda').train()
lossd, outd, stt = train_forward(m2, x, tg)
for yl, hl in ((82, 81), (94, 93), (106, 105)):
    d = stt.dhead[yl].cpu().numpy(); r = acts[hl].grad.numpy()
    print('dhead', hl, 'relerr', np.abs(d - r).max() / np.abs(r).max(), 'max', np.abs(r).max())
    bad = np.abs(d - r) > 1e-4 * np.abs(r).max()
    if bad.any():
        idx = np.argwhere(bad)
        print('  nbad', len(idx), 'channels', sorted(set(idx[:, 1].tolist()))[:20], 'first', idx[:4].tolist())
        i0 = tuple(idx[0]); print('  got', d[i0], 'ref', r[i0])

from amyloid_yolo_paper_amd.train_engine import train_backward
m2._keep_dval = {}
grads = train_backward(m2, stt)
for li in sorted(m2._keep_dval, reverse=True)[:40]:
    d = m2._keep_dval[li].cpu().numpy(); r = acts[li].grad.numpy()
    print('dY', li, 'relerr', np.abs(d - r).max() / np.abs(r).max(), 'max', np.abs(r).max(), 'mean abs', np.abs(r).mean())
rec = stt.conv[80]
bn = m2.module_list[80][1]
print('bn80 mean/invstd', rec['mean'][:4].cpu().numpy(), rec['invstd'][:4].cpu().numpy())
zz = rec['z'].cpu().numpy(); print('z80 ch0 values', zz[:, 0].reshape(-1)[:8], 'std', zz[:,0].std())
gb = dict(zip([id(p) for p in m2.parameters()], grads))
print('dbeta80 mine', gb[id(bn.bias)][:6].cpu().numpy(), 'ref', z['gbeta80'][:6])
print('dgamma80 mine', gb[id(bn.weight)][:6].cpu().numpy(), 'ref', z['ggamma80'][:6])
for li in (101, 96, 91):
    d = m2._keep_dval[li].cpu().numpy(); r = acts[li].grad.numpy()
    err = np.abs(d - r); bad = err > 1e-3 * np.abs(r).max()
    idx = np.argwhere(bad)
    print('layer', li, 'shape', d.shape, 'nbad', len(idx), 'of', d.size, 'batches', np.bincount(idx[:,0]), 'ys', np.bincount(idx[:,2]), 'xs', np.bincount(idx[:,3]), 'nch', len(set(idx[:,1].tolist())))
    i0 = tuple(idx[0]); print('  eg', i0, d[i0], r[i0])
# check leaky zero / BN of layer 102: compare z,y with oracle
yo = o.layer_outputs[102].detach().numpy(); ym = stt.conv[102]['y'].cpu().numpy()
print('y102 relerr', np.abs(yo-ym).max(), 'num exact zeros', (ym==0).sum(), 'num |y|<1e-6', (np.abs(ym)<1e-6).sum())
nflip = 0
for li in sorted(stt.conv):
    e = m2._graph[li]
    if not e['bn']: continue
    yo = o.layer_outputs[li].detach().numpy(); ym = stt.conv[li]['y'].cpu().numpy()
    flip = (yo > 0) != (ym > 0)
    if flip.any():
        idx = np.argwhere(flip)
        nflip += len(idx)
        print('layer', li, 'sign flips', len(idx), 'values mine/ref', [(float(ym[tuple(i)]), float(yo[tuple(i)])) for i in idx[:3]], 'at', idx[:3].tolist())
print('total flips', nflip)
