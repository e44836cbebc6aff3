import sys, torch
import torch.nn.functional as F
sys.path.insert(0, 'synthesis-in-style_amd')
import sis_hip
dev = torch.device('cuda:0')
x = torch.randn(2, 3, 300, 280, generator=torch.Generator().manual_seed(300)).to(dev)
size = (600, 560)
y = sis_hip.upsample_bilinear(x, *size)
r32 = F.interpolate(x.cpu(), size=size, mode='bilinear', align_corners=True).to(dev)
g32 = F.interpolate(x, size=size, mode='bilinear', align_corners=True)
r64 = F.interpolate(x.double(), size=size, mode='bilinear', align_corners=True)
for name, r in (('cpu32', r32), ('gpu32', g32), ('f64', r64)):
    d = (y.double() - r.double()).abs()
    i = d.argmax().item()
    idx = []
    for sz in reversed(y.shape):
        idx.append(i % sz); i //= sz
    print(name, d.max().item(), 'at', idx[::-1])
print('cpu32 vs f64', (r32.double() - r64).abs().max().item(), 'gpu32 vs f64', (g32.double() - r64).abs().max().item())
