"""GPU: input staging -- DeviceLoader over the reference's batch structure, bf16 / uint8 transport."""
import pytest
import torch

from conftest import assert_close, rel_err
from oracle import healnet_cpu as O

KW = dict(n_modalities=2, channel_dims=[40, 3], num_spatial_axes=[1, 2], out_dims=4, depth=1, l_c=8, l_d=16, x_heads=2, l_heads=2,
          cross_dim_head=8, latent_dim_head=8)


def _oracle(model, kw, feats):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        return O.fusion_forward(sd, O.FusionConfig(**kw), [f.float().cpu() for f in feats])

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


def _batches(n, b, gen):
    out = []
    for _ in range(n):
        feats = [torch.rand(b, 1, 40, generator=gen), torch.rand(b, 9, 11, 3, generator=gen)]
        out.append((feats, torch.randint(0, 2, (b,), generator=gen), torch.rand(b, generator=gen), torch.randint(0, 4, (b,), generator=gen)))
    return out


def test_device_loader_yields_the_reference_batch_structure_in_order(hn):
    gen = torch.Generator().manual_seed(0)
    batches = _batches(5, 3, gen)
    model = hn.HealNet(n_modalities=2, channel_dims=[40, 3], num_spatial_axes=[1, 2], out_dims=4, depth=1, l_c=8, l_d=16,
                       x_heads=2, l_heads=2, cross_dim_head=8, latent_dim_head=8).eval().to(DEV)
    got = []
    with torch.no_grad():
        for (features, censorship, event_time, y_disc) in hn.etl.DeviceLoader(batches, DEV, depth=2):
            assert all(f.is_cuda for f in features) and censorship.is_cuda and y_disc.is_cuda and event_time.is_cuda
            assert isinstance(features, list) and len(features) == 2
            got.append(model(features))
        assert len(got) == 5
        for y, (features, c, t, yd) in zip(got, batches):
            assert torch.equal(y, model([f.to(DEV) for f in features]))       # same data, same order, no torn copies
            assert_close(y.cpu(), _oracle(model, KW, features), rel=2e-4, what="staged batch vs oracle")


def test_bf16_transport_equals_host_side_rounding(hn):
    gen = torch.Generator().manual_seed(1)
    batches = _batches(3, 2, gen)
    model = hn.HealNet(n_modalities=2, channel_dims=[40, 3], num_spatial_axes=[1, 2], out_dims=4, depth=1, l_c=8, l_d=16,
                       x_heads=2, l_heads=2, cross_dim_head=8, latent_dim_head=8).eval().to(DEV)
    with torch.no_grad():
        for (features, c, t, yd), (f0, _, _, _) in zip(hn.etl.DeviceLoader(batches, DEV, transport="bf16"), batches):
            assert all(f.dtype == torch.bfloat16 for f in features)
            assert c.dtype == torch.int64 and t.dtype == torch.float32       # labels / times untouched
            want = model([f.to(torch.bfloat16).float().to(DEV) for f in f0])
            assert torch.equal(model(features), want)
            # ... and the oracle on the same bf16-rounded values (the reference would see exactly these after a host-side cast)
            assert_close(want.cpu(), _oracle(model, KW, [f.to(torch.bfloat16) for f in f0]), rel=2e-4, what="bf16 transport vs oracle")


def test_uint8_image_transport_is_totensor_exact(hn):
    gen = torch.Generator().manual_seed(2)
    model = hn.HealNet(n_modalities=2, channel_dims=[40, 3], num_spatial_axes=[1, 2], out_dims=4, depth=2, l_c=8, l_d=16,
                       x_heads=2, l_heads=2, cross_dim_head=8, latent_dim_head=8).eval().to(DEV)
    tab = torch.rand(4, 1, 40, generator=gen).to(DEV)
    img8 = torch.randint(0, 256, (4, 17, 13, 3), generator=gen, dtype=torch.uint8)
    img8[0, 0, 0] = torch.tensor([0, 255, 128], dtype=torch.uint8)
    with torch.no_grad():
        want = model([tab, img8.float().div(255).to(DEV)])      # what ToTensor hands the reference
        assert torch.equal(model([tab, img8.to(DEV)]), want)
        kw2 = dict(KW, depth=2)
        assert_close(want.cpu(), _oracle(model, kw2, [tab, img8.float().div(255)]), rel=2e-4, what="uint8 transport vs oracle")
        low = hn.HealNet(n_modalities=2, channel_dims=[40, 3], num_spatial_axes=[1, 2], out_dims=4, depth=2, l_c=8, l_d=16,
                         x_heads=2, l_heads=2, cross_dim_head=8, latent_dim_head=8, core_precision="bf16").eval().to(DEV)
        low.load_state_dict(model.state_dict())
        assert rel_err(low([tab, img8.to(DEV)]), want) <= 2e-2


def test_bag_padding_mask(hn):
    bag = torch.rand(2, 50, 6, device=DEV)
    bag[0, 30:] = 0
    bag[1, 45:] = 0
    m = hn.etl.bag_padding_mask(bag)
    assert m.shape == (2, 50) and int(m[0].sum()) == 30 and int(m[1].sum()) == 45
