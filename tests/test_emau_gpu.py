"""GPU parity of EMANet's EM attention unit on its own kernels (csrc/emau.hip) against the reference's composition
(networks/ema_net/network.py:229-247: bmm -> softmax -> column normalisation -> bmm -> l2norm, three rounds, then mu z^T and
ReLU) evaluated in float64 on the same inputs.

Tolerance: every product is an exact-fp32 MFMA chain, the exponentials are ``expf``; what differs from a float64 evaluation
is fp32 rounding of 512- / 1024-term sums, amplified by three rounds of a contraction-free fixed-point iteration: bases to
1e-5 of their (unit) norm per column, reconstruction to 2e-5 of max|ref| (measured values are written to gpurun_out/)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(x, mu0, stages):
    b, c = x.shape[0], x.shape[1]
    x = x.double().view(b, c, -1)
    mu = mu0.double().repeat(b, 1, 1)
    x_t = x.permute(0, 2, 1)
    for _ in range(stages):
        z = torch.softmax(torch.bmm(x_t, mu), dim=2)
        z_ = z / (1e-6 + z.sum(dim=1, keepdim=True))
        mu = torch.bmm(x, z_)
        mu = mu / (1e-6 + mu.norm(dim=1, keepdim=True))
    return torch.relu(mu.matmul(z.permute(0, 2, 1))), mu


@pytest.mark.parametrize("b,c,h,w,stages", [(16, 512, 32, 32, 3), (2, 512, 32, 32, 3), (3, 128, 16, 8, 1), (1, 64, 16, 16, 2),
                                            (5, 256, 8, 48, 4)])
def test_emau_kernels_vs_float64_composition(device, b, c, h, w, stages):
    import sis_hip
    gen = torch.Generator().manual_seed(b * 1000 + c + h)
    # conv1's output in the network: O(1) activations with structure, so that the softmax is neither flat nor one-hot
    x = torch.randn(b, c, h, w, generator=gen) * 0.5 + torch.randn(b, c, 1, 1, generator=gen) * 0.3
    mu0 = torch.randn(1, c, 64, generator=gen)
    mu0 = mu0 / (1e-6 + mu0.norm(dim=1, keepdim=True))
    assert sis_hip.emau_supported(x.to(device), mu0.to(device))
    y, mu = sis_hip.emau_forward(x.to(device), mu0.to(device), stages)
    y_ref, mu_ref = _reference(x, mu0, stages)
    assert tuple(y.shape) == (b, c, h, w) and tuple(mu.shape) == (b, c, 64)
    mu_err = ((mu.cpu().double() - mu_ref).norm(dim=1).max()).item()          # columns have unit norm
    y_err = (y.cpu().double().view(b, c, -1) - y_ref).abs().max().item() / y_ref.abs().max().item()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"emau_parity_{b}_{c}_{h}x{w}_{stages}.json"), "w") as f:
        json.dump({"mu_column_l2": mu_err, "recon_max_over_max": y_err}, f)
    assert mu_err < 1e-5, mu_err
    assert y_err < 2e-5, y_err
    again = sis_hip.emau_forward(x.to(device), mu0.to(device), stages)
    assert torch.equal(again[0], y) and torch.equal(again[1], mu), "no atomics: bitwise repeatable"


def test_emau_module_uses_the_kernels_and_matches_the_library_path(device, monkeypatch):
    """EMAU.forward on the kernels vs the same module with SIS_HIP_EMAU off (torch.bmm / softmax): outputs, bases, and the
    gradient reaching conv2 / the block input agree; conv1 receives no gradient on either path (reference :229-240)."""
    import sis_hip
    import networks.ema_net.network as N
    torch.manual_seed(3)
    unit = N.EMAU(512, 64, 3).to(device).train()
    x = torch.randn(4, 512, 32, 32, device=device)

    def run(flag):
        monkeypatch.setattr(N, "_HIP_EMAU", flag)
        unit.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        sis_hip.library_calls(reset=True)
        y, mu = unit(xi)
        calls = sis_hip.library_calls(reset=True)
        y.square().mean().backward()
        return y.detach(), mu.detach(), xi.grad, unit.conv2[0].weight.grad.clone(), unit.conv1.weight.grad, calls

    y1, mu1, gx1, gw1, gc1, calls1 = run(True)
    y0, mu0, gx0, gw0, gc0, calls0 = run(False)
    assert gc1 is None and gc0 is None
    assert calls1["fallback"] == {} and "ema_net.EMAU (bmm / softmax rounds)" in calls0["fallback"]
    assert (mu1 - mu0).norm(dim=1).max().item() < 2e-5
    assert (y1 - y0).abs().max().item() <= 2e-4 * y0.abs().max().item()
    assert ((gw1 - gw0).norm() / gw0.norm()).item() < 1e-3 and ((gx1 - gx0).norm() / gx0.norm()).item() < 1e-3
