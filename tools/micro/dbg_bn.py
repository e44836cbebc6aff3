import os, sys, torch
sys.path.insert(0, 'synthesis-in-style_amd')
import sis_hip
dev = torch.device('cuda:0')
b, c, h, w, relu, use_res = 16, 64, 32, 32, True, True
gen = torch.Generator().manual_seed(b * 5 + c + h)
x = (torch.randn(b, c, h, w, generator=gen) * 2 + 0.5).to(dev)
res = torch.randn(b, c, h, w, generator=gen).to(dev)
dy = torch.randn(b, c, h, w, generator=gen).to(dev)
gamma, beta = (1 + 0.1 * torch.randn(c, generator=gen)).to(dev), (0.1 * torch.randn(c, generator=gen)).to(dev)
def run(single):
    os.environ["SIS_BN_SINGLE_PASS"] = "1" if single else "0"
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    if single:
        y, mean, invstd, mask = sis_hip.bn_fused_fwd(x, res, gamma, beta, rm, rv, 1e-5, 3e-4, relu, want_mask=relu)
    else:
        mean, invstd = sis_hip.bn_stats(x, rm, rv, 1e-5, 3e-4)
        y, mask = sis_hip.bn_act_fwd(x, res, mean, invstd, gamma, beta, relu, want_mask=relu)
    grads = sis_hip.bn_act_bwd(dy, None, x, mean, invstd, gamma, relu, use_res, mask=mask)
    return dict(y=y, mean=mean, invstd=invstd, rm=rm, rv=rv, mask=mask, dx=grads[0], dres=grads[1], dgamma=grads[2], dbeta=grads[3])
a, t = run(True), run(False)
for k in a:
    if not torch.equal(a[k], t[k]):
        d = (a[k].double() - t[k].double()).abs()
        print(k, 'max diff', d.max().item(), 'n diff', int((d > 0).sum()), 'scale', t[k].double().abs().max().item())
print('done')
