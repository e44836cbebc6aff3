"""GPU parity, part 6 ("next" rows): nearest-centre label maps and the uint8 image conversion.

Both are integer / byte outputs: bit-exact against the fp32 oracle.  The kernel reproduces the association of the
fp32 adds of torch's CPU ``.sum(dim=-1)`` (csrc/dataset_ops.hip; oracle/kmeans_ref.py::predict_ordered, pinned against
torch's own sum and against label maps of the reference's own FactorCatalog in tests/test_oracle_cpu.py), so near-ties
break exactly as in the reference."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import kmeans_ref

pytestmark = pytest.mark.gpu


def _kmeans_cases():
    spec = importlib.util.spec_from_file_location("make_golden_kmeans", os.path.join(os.path.dirname(__file__), "golden",
                                                                                     "make_golden_kmeans.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("b,c,h,w,k", [(2, 128, 64, 64, 20), (1, 512, 16, 16, 7), (3, 32, 5, 7, 40), (2, 64, 32, 32, 16),
                                       (1, 16, 8, 8, 1), (1, 20, 3, 3, 5), (1, 1056, 6, 6, 9), (2, 40, 9, 8, 64),
                                       (1, 256, 16, 16, 24), (1, 384, 8, 8, 12), (1, 128, 7, 7, 24),
                                       (2, 1, 9, 9, 5), (1, 3, 16, 16, 24), (1, 4, 8, 8, 6), (2, 5, 7, 9, 16), (1, 7, 12, 12, 32)])  # (multiples of 128: the prefetching / packed path; odd HW: its one-pixel form)
def test_kmeans_assign_matches_reference_rule(device, b, c, h, w, k):
    from segmentation.gan_local_edit.factor_catalog import FactorCatalog
    gen = torch.Generator().manual_seed(c + k)
    x = torch.randn(b, c, h, w, generator=gen)
    centres = torch.randn(k, c, generator=gen)
    centres[1::2] = centres[0:2 * (k // 2):2] * (1 + 1e-7 * torch.randn(k // 2, c, generator=gen))  # near-duplicate pairs
    ref32, _ = kmeans_ref.predict(x, centres)
    ordered, _ = kmeans_ref.predict_ordered(x, centres)
    assert torch.equal(ref32, ordered)  # the oracle's two statements agree on this host too
    cat = FactorCatalog(k, cluster_centers=centres)
    got = cat.predict(x.to(device)).cpu()
    assert got.dtype == torch.int64 and tuple(got.shape) == (b, h, w)
    assert torch.equal(got, ref32)
    flat = x.permute(0, 2, 3, 1).reshape(-1, c)
    assert torch.equal(cat.pairwise_distance(flat.to(device)).cpu(), ref32.reshape(-1))


def test_kmeans_assign_matches_reference_fixture(device, golden_dir):
    """Label maps computed by the reference's own FactorCatalog.predict (tests/golden/make_golden_kmeans.py)."""
    import sis_hip
    mk = _kmeans_cases()
    g = np.load(os.path.join(golden_dir, "kmeans_reference.npz"))
    for i in range(len(mk.CASES)):
        x, centres = mk.case_inputs(i)
        got = sis_hip.kmeans_assign(x.to(device), centres.to(device)).cpu()
        assert torch.equal(got, torch.from_numpy(g[f"labels{i}"].astype(np.int64))), f"case {i} {mk.CASES[i]}"


@pytest.mark.parametrize("mfma", ["1", "0"])
@pytest.mark.parametrize("b,c,h,w,k", [(4, 512, 64, 64, 24), (2, 128, 256, 256, 24), (3, 48, 30, 34, 32), (2, 16, 64, 64, 8), (2, 64, 16, 24, 32),
                                       (1, 512, 32, 32, 1), (2, 32, 16, 16, 3)])
def test_kmeans_fast_pass_plus_exact_refinement(device, b, c, h, w, k, mfma, monkeypatch):
    """(``mfma``: the first pass on the matrix cores -- csrc/dataset_ops.hip kmeans_mfma_kernel, HW % 128 == 0 and C % 32 == 0 -- or
    the VALU one.)  The two-pass form (fast distances decide the pixels whose argmin cannot depend on the order of the adds, the exact-order
    kernel revisits the rest) returns the label map of the exact-order kernel alone, bit for bit: unit-variance activations at
    the generator's key shapes (few open pixels), plus pixels planted exactly between two centres and on top of a duplicated
    centre (open by construction), plus a NaN pixel."""
    import sis_hip
    gen = torch.Generator().manual_seed(c + k + h)
    x = torch.randn(b, c, h, w, generator=gen)
    centres = torch.randn(k, c, generator=gen)
    if k >= 4:
        centres[3] = centres[1]                                   # exact duplicate: ties go to the lower index
        x[0, :, 0, 0] = centres[1]
        x[0, :, 0, 1] = (centres[0] + centres[2]) / 2             # equidistant up to rounding
        x[-1, :, h - 1, w - 1] = centres[k - 1] * (1 + 1e-7)
        x[0, 0, 1, 1] = float("nan")
    xd, cd = x.to(device), centres.to(device)
    monkeypatch.setenv("SIS_KMEANS_MFMA", mfma)
    monkeypatch.setenv("SIS_KMEANS_FAST", "0")
    exact = sis_hip.kmeans_assign(xd, cd)
    monkeypatch.setenv("SIS_KMEANS_FAST", "1")
    two_pass = sis_hip.kmeans_assign(xd, cd)
    first_pass = sis_hip.lib().sis_last_kernel().decode()
    takes_mfma = mfma == "1" and (h * w) % 128 == 0 and c % 64 == 0
    assert first_pass == ("kmeans_mfma_kernel" if takes_mfma else "kmeans_fast_kernel"), first_pass
    assert torch.equal(two_pass, exact)
    assert int(two_pass.min()) >= 0 and int(two_pass.max()) < k
    if k >= 4:
        assert int(exact[0, 0, 0]) == 1


def test_kmeans_ties_go_to_lowest_index(device):
    import sis_hip
    x = torch.zeros(1, 8, 2, 2, device=device)
    centres = torch.zeros(4, 8, device=device)
    centres[0, 0] = centres[1, 1] = 1.0  # centres 2 and 3 tie at distance 0
    assert torch.equal(sis_hip.kmeans_assign(x, centres).cpu(), torch.full((1, 2, 2), 2))


def test_kmeans_on_generator_activations_256(device):
    """Layer keys "12"/"13" of the shipped dataset config are [B,128,256,256]."""
    import sis_hip
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(2, 128, 256, 256, generator=gen)
    centres = torch.randn(20, 128, generator=gen)
    got = sis_hip.kmeans_assign(x.to(device), centres.to(device)).cpu()
    ref, d = kmeans_ref.predict(x[:1, :, :64], centres)
    assert torch.equal(got[:1, :64], ref)


def test_make_image_u8(device):
    import sis_hip
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(3, 3, 32, 40, generator=gen) * 0.8
    x[0, 0, 0, 0], x[0, 1, 0, 0], x[0, 2, 0, 0] = -1.0, 1.0, 0.0
    got = sis_hip.make_image_u8(x.to(device)).cpu()
    ref = kmeans_ref.make_image(x)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (3, 32, 40, 3)
    assert torch.equal(got, ref)  # byte work: every byte, add / div / mul as separate fp32 operations like the oracle
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
    from networks import get_stylegan2_generator
    torch.manual_seed(0)
    g = get_stylegan2_generator(32, 512, n_mlp=2)
    with torch.no_grad():
        for name, p in g.named_parameters():
            if name.endswith("noise.weight"):
                p.normal_(0.0, 0.3)  # the per-batch device noise must matter for the world-size check below
    ckpt = tmp_path / "g.pt"
    torch.save({"g_ema": g.state_dict()}, ckpt)
    args = argparse.Namespace(checkpoint=str(ckpt), config=None, num_images=7, save_to=str(tmp_path / "out"), batch_size=3,
                              truncate=True)
    torch.manual_seed(0)  # mean_latent(4096) draws from the device RNG before the loop seeds it
    d0, r0 = cds.build_dataset(args, cfg, rank=0, world_size=2)
    torch.manual_seed(0)
    d1, r1 = cds.build_dataset(args, cfg, rank=1, world_size=2)
    assert (r0, r1) == ((0, 4), (4, 7)) and d0 == 4 and d1 == 3
    # the image an id maps to does not depend on the world size: a single-rank run writes the same files
    single = argparse.Namespace(**{**vars(args), "save_to": str(tmp_path / "single")})
    torch.manual_seed(0)
    assert cds.build_dataset(single, cfg, rank=0, world_size=1) == (7, (0, 7))
    for f in sorted((tmp_path / "out").rglob("*.png")):
        twin = tmp_path / "single" / f.relative_to(tmp_path / "out")
        assert np.array_equal(np.asarray(Image.open(f)), np.asarray(Image.open(twin))), f.name
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


def test_tensor_hand_off_equals_the_png_round_trip(device, tmp_path):
    """the batch ``SynthesisSegmentationLoader`` hands to the trainer equals, bit for bit, what the
    reference's PNG pipeline yields for the same sample: [image | label colours] PNG written as
    create_dataset_for_segmentation.py:84-99 does, read back with PIL, split, ToTensor + Normalize(0.5, 0.5), colours ->
    class ids, nearest resize (data/segmentation_dataset.py:44-63, restated here with numpy / PIL)."""
    from PIL import Image
    from data.device_dataset import SynthesisSegmentationLoader
    from networks.stylegan2.model import Generator
    from segmentation.gan_local_edit.factor_catalog import FactorCatalog
    torch.manual_seed(4)
    g = Generator(64, 64, 2, channel_multiplier=1).to(device).eval()
    layer = 7  # [B, C, 32, 32]: the label map is resized to the image's 64 x 64
    with torch.no_grad():
        _, acts = g([torch.randn(1, 64, device=device)], return_intermediate_activations=True)
    rng = np.random.RandomState(9)
    catalogs = {layer: FactorCatalog(cluster_centers=rng.randn(6, acts[layer].shape[1]).astype(np.float32))}
    class_of_cluster = torch.tensor([0, 1, 2, 1, 0, 2])
    colours = np.array([[0, 0, 0], [255, 0, 0], [0, 0, 255]], dtype=np.uint8)  # class -> colour of the PNG's right half
    loader = SynthesisSegmentationLoader(g, catalogs, layer, batch_size=3, class_of_cluster=class_of_cluster, image_size=64,
                                         seed=11, num_batches=1)
    torch.manual_seed(21)  # make_noise() draws from the device RNG
    batch = next(iter(loader))
    assert batch["images"].dtype == torch.float32 and tuple(batch["images"].shape) == (3, 3, 64, 64)
    assert batch["segmented"].dtype == torch.int64 and tuple(batch["segmented"].shape) == (3, 1, 64, 64)
    assert batch["images"].is_cuda and batch["images"].abs().max() <= 1
    # the PNG path on the same sample
    import sis_hip
    torch.manual_seed(21)
    z = torch.randn(3, 64, generator=torch.Generator().manual_seed(11)).to(device)
    with torch.no_grad():
        image, acts = g([z], noise=g.make_noise(), return_intermediate_activations=True)
    rgb = sis_hip.make_image_u8(image).cpu().numpy()
    classes = class_of_cluster[catalogs[layer].predict(acts[layer]).cpu()].numpy()
    for i in range(3):
        lab = colours[np.repeat(np.repeat(classes[i], 2, 0), 2, 1)]  # label image rendered at the image's resolution
        Image.fromarray(np.concatenate([rgb[i], lab], axis=1)).save(tmp_path / f"{i}.png")
        png = Image.open(tmp_path / f"{i}.png")
        left = np.asarray(png.crop((0, 0, png.width // 2, png.height)))
        right = np.asarray(png.crop((png.width // 2, 0, png.width, png.height)))
        want_img = (torch.from_numpy(left.copy()).permute(2, 0, 1).float().div(255) - 0.5) / 0.5  # ToTensor + Normalize
        want_cls = torch.from_numpy((right[..., None, :] == colours[None, None]).all(-1).argmax(-1))
        assert torch.equal(batch["images"][i].cpu(), want_img)
        assert torch.equal(batch["segmented"][i, 0].cpu(), want_cls)


def test_loader_leaves_grad_mode_on_and_feeds_an_updater(device):
    """The loader's generator body synthesises under ``no_grad`` but must hand the batch over OUTSIDE it: the updater keeps the
    iterator suspended at its ``yield`` across forward / backward (updater/segmentation_updater.py:47-73 in the reference
    holds its iterators the same way), and a generator suspended inside ``no_grad`` leaves grad mode off in the consumer."""
    import yaml
    from data.device_dataset import SynthesisSegmentationLoader
    from networks.stylegan2.model import Generator
    from segmentation.gan_local_edit.factor_catalog import FactorCatalog
    from training_builder.ema_net_train_builder import EMANetTrainBuilder
    torch.manual_seed(4)
    g = Generator(64, 64, 2, channel_multiplier=1).to(device).eval()
    layer = 7
    with torch.no_grad():
        _, acts = g([torch.randn(1, 64, device=device)], return_intermediate_activations=True)
    rng = np.random.RandomState(9)
    catalogs = {layer: FactorCatalog(cluster_centers=rng.randn(6, acts[layer].shape[1]).astype(np.float32))}
    loader = SynthesisSegmentationLoader(g, catalogs, layer, batch_size=2, class_of_cluster=torch.tensor([0, 1, 2, 1, 0, 2]),
                                         image_size=64, seed=11)
    it = iter(loader)
    next(it)
    assert torch.is_grad_enabled(), "the suspended loader left grad mode off"
    next(it)
    assert torch.is_grad_enabled()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "synthesis-in-style_amd", "configs", "segmenter", "ema_net_resnet50_256.yaml")))
    cfg.update(fine_tune=None, batch_size=2, image_size=64)
    torch.manual_seed(0)
    builder = EMANetTrainBuilder(cfg, loader, None, rank=0, world_size=1)
    updater = builder.get_updater()
    before = [p.detach().clone() for p in builder.get_network().parameters()][:4]
    for _ in range(4):  # two eager iterations, the graph capture, one replay
        updater.update()
    torch.cuda.synchronize()
    after = list(builder.get_network().parameters())[:4]
    assert all(torch.isfinite(a).all() for a in after)
    assert any(not torch.equal(a, b) for a, b in zip(after, before)), "the step fed by the loader did not train"


def test_make_noise_on_the_device_is_one_launch_of_consecutive_maps(device, monkeypatch):
    """``Generator.make_noise()`` on a HIP device draws all maps with ONE generator launch and returns them as consecutive
    slices of that buffer (13 launches on the critical path of the dataset loop otherwise; SIS_NOISE_ONE_LAUNCH=0: one tensor
    per map as in the reference, networks/stylegan2/model.py:443-452).  Same shapes, dense, 64-byte aligned, unit variance; the
    generator takes either form and the same maps give the same image."""
    import networks.stylegan2.model as M
    g = M.Generator(64, 64, 2, channel_multiplier=1).to(device).eval()
    want = [(1, 1, 4, 4)] + [(1, 1, 2 ** i, 2 ** i) for i in range(3, 7) for _ in range(2)]
    torch.manual_seed(5)
    one = g.make_noise()
    assert [tuple(n.shape) for n in one] == want
    assert all(n.is_contiguous() and n.data_ptr() % 64 == 0 for n in one)
    base = one[0].untyped_storage().data_ptr()
    assert all(n.untyped_storage().data_ptr() == base for n in one), "one buffer behind all maps"
    ends = [n.data_ptr() + n.numel() * 4 for n in one]
    assert all(a == b.data_ptr() for a, b in zip(ends[:-1], one[1:])), "consecutive slices"
    flat = torch.cat([n.flatten() for n in one])
    assert abs(float(flat.mean())) < 0.05 and abs(float(flat.std()) - 1) < 0.05
    monkeypatch.setattr(M, "_NOISE_ONE_LAUNCH", False)
    many = g.make_noise()
    assert [tuple(n.shape) for n in many] == want and len({n.untyped_storage().data_ptr() for n in many}) == len(many)
    z = torch.randn(2, 64, device=device)
    with torch.no_grad():
        img_views, _ = g([z], noise=one)
        img_copies, _ = g([z], noise=[n.clone() for n in one])
    assert torch.equal(img_views, img_copies)
