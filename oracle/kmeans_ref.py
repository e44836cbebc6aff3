"""ORACLE (test infrastructure, not product): nearest-centre assignment exactly as the reference computes it,
/root/reference/stylegan_code_finder/segmentation/gan_local_edit/factor_catalog.py:47-62 (``pairwise_distance``:
``((A - B) ** 2).sum(-1)`` then ``argmin``) after ``ptutils.partial_flat`` (:23-26), plus the third-party
``make_image`` conversion (clamp to [-1,1], add 1, div 2, mul 255, truncating uint8 cast, NHWC).

Pins: ``tests/golden/kmeans_reference.npz`` holds label maps computed by the reference's own ``FactorCatalog.predict``
(module loaded by file path in the build container, tests/golden/make_golden_kmeans.py); ``predict`` below must
reproduce them bit for bit (tests/test_oracle_cpu.py).  ``make_image`` is third-party (``pytorch_training``, not in the
reference tree): parity unpinned for its rounding, restated from its public behaviour.

``predict_ordered`` states the ASSOCIATION of the fp32 adds inside torch's ``.sum(dim=-1)`` explicitly (ATen
SumKernel.cpp ``vectorized_inner_sum`` / ``row_sum`` / ``multi_row_sum``: per vector lane four interleaved running
sums with a cascade every 16 steps, the four merged in order, then the lanes added in order after the scalar tail), so that a device kernel can reproduce near-tie decisions exactly instead of
"whenever the margin is large enough"; it is pinned against ``predict`` (torch's own sum, 8-float vectors under
both the AVX2 and the AVX-512 dispatch) in tests/test_oracle_cpu.py."""
import torch


def predict(X, centres):
    b, c, h, w = X.shape
    flat = X.permute(0, 2, 3, 1).contiguous().view(-1, c)
    d = ((flat.unsqueeze(1) - centres.unsqueeze(0)) ** 2.0).sum(dim=-1)
    return torch.argmin(d, dim=1).reshape(b, h, w), d.reshape(b, h, w, -1)


def host_vector_lanes():
    """Floats per vector of torch's CPU float sum kernel: 8 under the AVX2 and (measured in the build container,
    ATEN_CPU_CAPABILITY default and =avx2) the AVX-512 dispatch alike."""
    return 8


def _multi_row_sum(rows):
    """ATen SumKernel.cpp ``multi_row_sum``: rows [n, ...] -> running sum over n with a cascade every 16 rows."""
    n = rows.shape[0]
    level_power = max(4, (max(n, 1) - 1).bit_length() // 4)
    step, mask0 = 1 << level_power, (1 << level_power) - 1
    acc = [torch.zeros_like(rows[0]) for _ in range(4)] if n else [torch.zeros(rows.shape[1:], dtype=rows.dtype)] * 4
    i = 0
    while i + step <= n:
        for _ in range(step):
            acc[0] = acc[0] + rows[i]
            i += 1
        for j in range(1, 4):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = torch.zeros_like(acc[j])
            if i & (mask0 << (j * level_power)):
                break
    while i < n:
        acc[0] = acc[0] + rows[i]
        i += 1
    for j in range(1, 4):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def _row_sum(elems):
    """ATen ``row_sum``: elems [size, ...] summed over ``size`` as four interleaved partial sums (ILP factor 4)."""
    size = elems.shape[0]
    size_ilp = size // 4
    partial = _multi_row_sum(elems[:size_ilp * 4].reshape(size_ilp, 4, *elems.shape[1:]))  # [4, ...]
    partial = [partial[k] for k in range(4)]
    for i in range(size_ilp * 4, size):
        partial[0] = partial[0] + elems[i]
    for k in range(1, 4):
        partial[0] = partial[0] + partial[k]
    return partial[0]


def ordered_sum(terms, lanes=8):
    """Sum over the last axis with torch's CPU association for an inner (contiguous) reduction of ``size`` floats:
    ``vectorized_inner_sum`` when size >= lanes, ``scalar_inner_sum`` below."""
    size = terms.shape[-1]
    t = terms.movedim(-1, 0)  # [size, ...]
    if size < lanes:
        return _row_sum(t)
    nvec = size // lanes
    vec_acc = _row_sum(t[:nvec * lanes].reshape(nvec, lanes, *t.shape[1:]))  # [lanes, ...]
    total = torch.zeros(t.shape[1:], dtype=terms.dtype)
    for c in range(nvec * lanes, size):  # scalar tail first
        total = total + t[c]
    for lane in range(lanes):
        total = total + vec_acc[lane]
    return total


def predict_ordered(X, centres, lanes=8):
    """``predict`` with the summation order written out (see the module docstring); fp32 in, fp32 arithmetic."""
    b, c, h, w = X.shape
    flat = X.permute(0, 2, 3, 1).contiguous().view(-1, c)
    diff = flat.unsqueeze(1) - centres.unsqueeze(0)
    d = ordered_sum(diff * diff, lanes)
    return torch.argmin(d, dim=1).reshape(b, h, w), d.reshape(b, h, w, -1)


def make_image(t):
    return t.detach().clamp(min=-1, max=1).add(1).div(2).mul(255).type(torch.uint8).permute(0, 2, 3, 1).contiguous()
