"""GPU parity, part 6 ("next" rows): nearest-centre label maps and the uint8 image conversion.

Label maps are integer outputs: they must be bit-exact wherever the best and second-best squared distances of the
fp64 oracle differ by more than fp32 rounding of the sums (the kernel accumulates channels in order, torch's
reduction is pairwise -- both are fp32 sums of the same terms)."""
import numpy as np
import pytest
import torch

from oracle import kmeans_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,c,h,w,k", [(2, 128, 64, 64, 20), (1, 512, 16, 16, 7), (3, 32, 5, 7, 40), (2, 64, 32, 32, 16),
                                       (1, 16, 8, 8, 1)])
def test_kmeans_assign_matches_reference_rule(device, b, c, h, w, k):
    from segmentation.gan_local_edit.factor_catalog import FactorCatalog
    gen = torch.Generator().manual_seed(c + k)
    x = torch.randn(b, c, h, w, generator=gen)
    centres = torch.randn(k, c, generator=gen)
    ref32, _ = kmeans_ref.predict(x, centres)
    ref64, d64 = kmeans_ref.predict(x.double(), centres.double())
    cat = FactorCatalog(k, cluster_centers=centres)
    got = cat.predict(x.to(device)).cpu()
    assert got.dtype == torch.int64 and tuple(got.shape) == (b, h, w)
    if k > 1:
        top2 = d64.topk(2, dim=-1, largest=False).values
        decided = (top2[..., 1] - top2[..., 0]) > 1e-4 * top2[..., 0]
    else:
        decided = torch.ones(b, h, w, dtype=torch.bool)
    assert torch.equal(got[decided], ref64[decided])
    assert decided.float().mean() > 0.99
    assert (got == ref32).float().mean() > 0.999
    flat = x.permute(0, 2, 3, 1).reshape(-1, c)
    assert torch.equal(cat.pairwise_distance(flat.to(device)).cpu()[decided.reshape(-1)], ref64.reshape(-1)[decided.reshape(-1)])


def test_kmeans_ties_go_to_lowest_index(device):
    import sis_hip
    x = torch.zeros(1, 4, 2, 2, device=device)
    centres = torch.tensor([[1., 0, 0, 0], [0, 1., 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]], device=device)
    assert torch.equal(sis_hip.kmeans_assign(x, centres).cpu(), torch.full((1, 2, 2), 2))


def test_kmeans_on_generator_activations_256(device):
    """Layer keys "12"/"13" of the shipped dataset config are [B,128,256,256]."""
    import sis_hip
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(2, 128, 256, 256, generator=gen)
    centres = torch.randn(20, 128, generator=gen)
    got = sis_hip.kmeans_assign(x.to(device), centres.to(device)).cpu()
    ref, d = kmeans_ref.predict(x[:1, :, :64], centres)
    assert (got[:1, :64] == ref).float().mean() > 0.999


def test_make_image_u8(device):
    import sis_hip
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(3, 3, 32, 40, generator=gen) * 0.8
    x[0, 0, 0, 0], x[0, 1, 0, 0], x[0, 2, 0, 0] = -1.0, 1.0, 0.0
    got = sis_hip.make_image_u8(x.to(device)).cpu()
    ref = kmeans_ref.make_image(x)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (3, 32, 40, 3)
    assert (got.int() - ref.int()).abs().max().item() <= 1  # (x+1)/2*255 may round across an integer boundary
    assert (got == ref).float().mean() > 0.999
    assert got[0, 0, 0].tolist() == [0, 255, 127]


def test_create_dataset_loop_shards_and_writes(device, tmp_path):
    """Hot loop of create_dataset_for_segmentation.py on a small generator: two 'ranks' cover disjoint id ranges,
    files land in the reference's directory layout, PNG = [image | label] side by side."""
    import argparse
    import numpy as np
    from PIL import Image
    import create_dataset_for_segmentation as cds
    centres = tmp_path / "c.npy"
    np.save(centres, np.random.RandomState(0).standard_normal((5, 512)).astype(np.float32))
    cfg = {"image_size": 32, "latent_size": 512, "n_mlp": 2, "seed": 3, "catalogs": {"7": str(centres)}, "label_layer": 7}
    args = argparse.Namespace(checkpoint=None, config=None, num_images=7, save_to=str(tmp_path / "out"), batch_size=3,
                              truncate=True)
    torch.manual_seed(0)
    d0, r0 = cds.build_dataset(args, cfg, rank=0, world_size=2)
    d1, r1 = cds.build_dataset(args, cfg, rank=1, world_size=2)
    assert (r0, r1) == ((0, 4), (4, 7)) and d0 == 4 and d1 == 3
    files = sorted((tmp_path / "out").rglob("*.png"))
    assert [f.name for f in files] == [f"{i:04d}.png" for i in range(7)]
    assert files[0].parent.name == "0" and files[0].parent.parent.name == "0"
    im = np.asarray(Image.open(files[5]))
    assert im.shape == (32, 64, 3) and im.dtype == np.uint8
    assert set(np.unique(im[:, 32:])) <= {0, 63, 127, 191, 255}


def test_label_and_encode_side_stream_is_bit_identical(device, monkeypatch):
    """utils.dataset_creation.label_and_encode: label maps and uint8 pixels issued on the side stream while the next
    batch is generated equal the same-stream results bit for bit (index / byte work), batch after batch."""
    from networks.stylegan2.model import Generator
    from segmentation.gan_local_edit.factor_catalog import FactorCatalog
    from utils.dataset_creation import label_and_encode
    torch.manual_seed(3)
    g = Generator(64, 64, 2, channel_multiplier=1).to(device).eval()
    rng = np.random.RandomState(5)
    catalogs = {k: FactorCatalog(cluster_centers=rng.randn(7, c).astype(np.float32)) for k, c in ((4, 512), (5, 512), (9, 256))}
    zs = [torch.randn(6, 64, device=device) for _ in range(4)]
    noise = g.make_noise()

    def run():
        jobs = []
        with torch.no_grad():
            for z in zs:  # forward of batch i+1 is issued while the label pass of batch i may still be running
                image, acts = g([z], noise=noise, return_intermediate_activations=True)
                jobs.append(label_and_encode(image, acts, catalogs))
                del image, acts
        out = []
        for pixels, labels, ready in jobs:
            if ready is not None:
                ready.synchronize()
            out.append((pixels.cpu(), {k: v.cpu() for k, v in labels.items()}))
        return out

    monkeypatch.setenv("SIS_LABEL_STREAM", "0")
    want = run()
    monkeypatch.setenv("SIS_LABEL_STREAM", "1")
    got = run()
    for (p0, l0), (p1, l1) in zip(want, got):
        assert p1.dtype == torch.uint8 and torch.equal(p0, p1)
        for k in l0:
            assert torch.equal(l0[k], l1[k])
