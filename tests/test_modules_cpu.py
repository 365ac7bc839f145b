"""CPU: host-side logic of the drop-in modules -- parameter names, creation (RNG) order, growth bookkeeping, aliasing,
zero_grad semantics -- against the reference-captured golden key lists and hashes.  No kernels run here."""
import pytest
import torch

from golden_util import PROGAN_CASES, load, sha


@pytest.mark.parametrize("case", PROGAN_CASES)
def test_state_dict_keys_shapes_and_same_seed_init_equal_reference(case):
    from musicgan_amd.networks import Discriminator, Generator
    g = load(f"progan_{case}.npz")
    torch.manual_seed(int(g["seed"]))
    gen = Generator(int(g["rand_channels"]), end_layer=int(g["g_end_layer"]))
    disc = Discriminator(start_layer=int(g["d_start_layer"]))
    for _ in range(int(g["n_grow"])):
        assert gen.next_layer() and disc.next_layer()
    ws = float(g["wscale"])
    if ws != 1.0:
        with torch.no_grad():
            for net in (gen, disc):
                for k, p in net.named_parameters():
                    if k.endswith("weight"):
                        p.mul_(ws)
    gsd, dsd = gen.state_dict(), disc.state_dict()
    assert list(gsd.keys()) == list(g["g_keys"]) and list(dsd.keys()) == list(g["d_keys"])
    assert [str(tuple(v.shape)) for v in gsd.values()] == list(g["g_shapes"])
    assert [str(tuple(v.shape)) for v in dsd.values()] == list(g["d_shapes"])
    assert [sha(v) for v in gsd.values()] == list(g["g_sha"])
    assert [sha(v) for v in dsd.values()] == list(g["d_sha"])
    assert gen.curr_layer == int(g["g_curr_layer"]) and disc.curr_layer == int(g["d_curr_layer"])


def test_growth_bookkeeping_and_aliasing():
    from musicgan_amd.networks import Discriminator, Generator
    g = load("progan_shapes.npz")
    torch.manual_seed(5)
    gen, disc = Generator(8), Discriminator(7)
    assert gen.down_sample == 7 and gen.curr_layer == 0 and disc.curr_layer == 7
    n_g0 = len(list(gen.parameters()))
    for i in range(10):
        assert [int(gen.growing), int(disc.growing)] == list(g["growing"][i])
        old_head = list(gen.end_block_params())
        old_stem = list(disc.start_block_parameters())
        grew_g, grew_d = gen.next_layer(), disc.next_layer()
        assert grew_g == grew_d == (i < 7)
        if grew_g:
            sd = gen.state_dict(keep_vars=True)
            assert sd["_Generator__last_end_block.0.0.weight"] is old_head[0]  # same Parameter object, re-used
            dsd = disc.state_dict(keep_vars=True)
            assert dsd["_Discriminator__last_start_block.1.0.weight"] is old_stem[0]
            assert all(p is not q for p in gen.end_block_params() for q in old_head)
    assert gen.curr_layer == 7 and disc.curr_layer == 0
    assert list(gen.state_dict().keys()) == list(g["g_keys_final"])
    assert list(disc.state_dict().keys()) == list(g["d_keys_final"])
    # parameters(): all 8 blocks + head + aliased previous head (deduplicated by nn.Module)
    assert len(list(gen.parameters())) == n_g0 + 2


def test_constructor_asserts_like_reference():
    from musicgan_amd.networks import Discriminator, Generator
    with pytest.raises(AssertionError):
        Generator(8, end_layer=8)
    with pytest.raises(AssertionError):
        Discriminator(start_layer=10)


def test_zero_grad_sets_none_and_state_dict_roundtrip():
    from musicgan_amd.networks import Generator
    torch.manual_seed(1)
    a = Generator(8, end_layer=3)
    for p in a.parameters():
        p.grad = torch.zeros_like(p)
    a.zero_grad()
    assert all(p.grad is None for p in a.parameters())
    torch.manual_seed(2)
    b = Generator(8, end_layer=3)
    b.load_state_dict(a.state_dict())
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


def test_losses_match_oracle():
    from musicgan_amd import networks
    from oracle import progan as O
    yr, yf = torch.randn(7, 1), torch.randn(7, 1)
    assert torch.equal(networks.wasserstein_discriminator_loss(yr, yf), O.w_disc_loss(yr, yf))
    assert torch.equal(networks.wasserstein_generator_loss(yf), O.w_gen_loss(yf))
    p = torch.rand(5, 1) * 0.8 + 0.1
    assert torch.allclose(networks.generator_loss(p), -torch.log2(p).mean())
    assert torch.allclose(networks.discriminator_loss(p, p), -(torch.log2(p) + torch.log2(1 - p)).mean())
