import sys, torch
sys.path.insert(0, 'synthesis-in-style_amd')
import networks.trans_u_net.vit_seg_modeling_resnet_skip as R
dev = torch.device('cuda:0')
torch.manual_seed(3)
net = R.ResNetV2((1, 2, 2), 1).to(dev)
x = torch.randn(2, 3, 128, 128, device=dev)
names = ['feat', 's0', 's1', 's2'] + [n for n, _ in net.named_parameters()]
def run():
    net.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        feat, skips = net(x)
    loss = feat.float().square().mean() + sum(s.float().mean() for s in skips)
    loss.backward()
    return [feat] + list(skips) + [p.grad.clone() for p in net.parameters()]
R._WS_BANK = False
a = run(); a2 = run()
R._WS_BANK = True
b = run(); b2 = run()
for n, u, v, u2, v2 in zip(names, a, b, a2, b2):
    if not torch.equal(u, v) or not torch.equal(u, u2) or not torch.equal(v, v2):
        print(n, tuple(u.shape), 'bank-vs-layer', (u.float() - v.float()).abs().max().item(), 'layer rerun', (u.float()-u2.float()).abs().max().item(), 'bank rerun', (v.float()-v2.float()).abs().max().item(), 'scale', u.float().abs().max().item())
print('done')
