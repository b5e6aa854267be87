"""GPU: the small-batch latency path (Next 4): ``HealNet.capture`` -- one inference forward as a HIP graph owned by
the model (static workspace / statistics / input buffers), replayed with one host call.  README.md:96-110 of the reference
calls the model at b = 1; BASELINE configs[0] is b = 4.

  * replay == eager forward bit for bit, == the CPU oracle within the fp32 tolerance, on new input VALUES of the captured shapes
    (3-modality README shape at reduced volume, cfg1 at its full image size, b = 1 and 4);
  * attention weights exported after a replay belong to the replayed inputs;
  * shape / missing-modality changes are refused loudly; masks and `return_embeddings` are part of the capture;
  * the cached inference descriptor (ops.Spec.model_cached) follows re-homed parameters.
"""
import pytest
import torch

from conftest import assert_close
from oracle import healnet_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


def _oracle(model, kw, ins, **extra):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        return O.fusion_forward(sd, O.FusionConfig(**kw), [None if t is None else t.clone() for t in ins], **extra)


@pytest.mark.parametrize("b", [1, 4])
@pytest.mark.parametrize("case", ["cfg1", "readme3"])
def test_graph_replay_equals_eager_and_oracle(hn, case, b):
    if case == "cfg1":
        kw = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4)
        shapes = [(1, 2000), (224, 224, 3)]
    else:
        kw = dict(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4)
        shapes = [(1, 2000), (64, 48, 3), (3, 40, 36, 3)]
    torch.manual_seed(21)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(22)
    first = [torch.rand(b, *s, generator=gen) for s in shapes]
    graph = model.capture([t.to(DEV) for t in first])
    for trial in range(3):                                   # new VALUES of the captured shapes
        ins = [torch.rand(b, *s, generator=gen) for s in shapes]
        with torch.no_grad():
            got = graph([t.to(DEV) for t in ins]).clone()
            eager = model([t.to(DEV) for t in ins])
        assert torch.equal(got, eager), f"trial {trial}: graph replay differs from the eager forward"
        if case == "readme3" or trial == 0:
            assert_close(got.cpu(), _oracle(model, kw, ins), rel=1e-3, floor=0.0, abs_floor=1e-5, what=f"{case} b={b} graph replay vs oracle")
    # the statistics / trace buffers are static: attention weights after a replay belong to the replayed inputs
    ins = [torch.rand(b, *s, generator=gen) for s in shapes]
    with torch.no_grad():
        graph([t.to(DEV) for t in ins])
        w_graph = [None if w is None else w.clone() for w in model.get_attention_weights()]
        model([t.to(DEV) for t in ins])
        w_eager = model.get_attention_weights()
    assert len(w_graph) == len(w_eager)
    for a, e in zip(w_graph, w_eager):
        assert (a is None) == (e is None)
        if a is not None:
            assert torch.equal(a, e)
    # in-place weight updates are seen by the next replay (parameters are read in place)
    with torch.no_grad():
        model.to_logits[2].bias.add_(0.5)
        got = graph([t.to(DEV) for t in ins]).clone()
        assert torch.equal(got, model([t.to(DEV) for t in ins]))


def test_graph_refuses_other_shapes_and_keeps_mask_and_embeddings(hn):
    kw = dict(n_modalities=2, channel_dims=[30, 3], num_spatial_axes=[1, 2], out_dims=3, depth=2, l_c=16, l_d=32, x_heads=2, l_heads=2,
              cross_dim_head=16, latent_dim_head=16)
    torch.manual_seed(31)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(32)
    seq = torch.rand(2, 40, 30, generator=gen)
    img = torch.rand(2, 8, 5, 3, generator=gen)
    mask = torch.rand(2, 40, generator=gen) > 0.3
    mask[:, 0] = True
    g = model.capture([seq.to(DEV), None], mask=mask.to(DEV), return_embeddings=True)
    seq2 = torch.rand(2, 40, 30, generator=gen)
    mask2 = torch.rand(2, 40, generator=gen) > 0.5
    mask2[:, 1] = True
    got = g([seq2.to(DEV), None], mask=mask2.to(DEV)).clone()
    want = _oracle(model, kw, [seq2, None], mask=mask2, return_embeddings=True)
    assert got.shape == (2, 16, 32)
    assert_close(got.cpu(), want, rel=1e-3, floor=0.0, abs_floor=1e-5, what="masked graph replay, embeddings")
    with pytest.raises(ValueError):
        g([seq2.to(DEV)[:1], None])
    with pytest.raises(ValueError):
        g([seq2.to(DEV), img.to(DEV)])
    model.train()
    model2 = hn.HealNet(**kw, attn_dropout=0.1).train().to(DEV)
    with pytest.raises(RuntimeError):
        model2.capture([seq.to(DEV), None])


def test_cached_descriptor_follows_rehomed_parameters(hn):
    kw = dict(n_modalities=2, channel_dims=[30, 3], num_spatial_axes=[1, 2], out_dims=3, depth=2, l_c=16, l_d=32, x_heads=2, l_heads=2,
              cross_dim_head=16, latent_dim_head=16)
    torch.manual_seed(41)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(42)
    ins = [torch.rand(3, 1, 30, generator=gen), torch.rand(3, 8, 6, 3, generator=gen)]
    with torch.no_grad():
        a = model([t.to(DEV) for t in ins]).clone()
        a2 = model([t.to(DEV) for t in ins])
        assert torch.equal(a, a2)
        # re-home every parameter (new storage, new values): the memoised descriptor must not be reused
        for p in model.parameters():
            p.data = (p.data * 1.5).clone()
        b_ = model([t.to(DEV) for t in ins])
    assert_close(b_.cpu(), _oracle(model, kw, ins), rel=1e-3, floor=0.0, abs_floor=1e-5, what="after re-homing the parameters")
    assert not torch.equal(a, b_)
    flat = hn.train.flatten_parameters(model)                # re-homes everything once more
    with torch.no_grad():
        c = model([t.to(DEV) for t in ins])
    assert torch.equal(c, b_)
    assert flat.numel > 0


# ------------------------------------------------------------------------------------------------
# healnet_amd.train.GraphedStep: zero_grad + tape forward + loss + fused backward as ONE graph replay (round 4)
# ------------------------------------------------------------------------------------------------
def _bag_model(hn, dropout):
    kw = dict(n_modalities=2, channel_dims=[40, 96], num_spatial_axes=[1, 1], out_dims=4, depth=2)
    if dropout:      # one of the reference's tuned shapes in small: odd latent width, one narrow cross head, both dropouts on (staged route)
        kw.update(l_c=17, l_d=62, x_heads=1, cross_dim_head=27, l_heads=8, latent_dim_head=16, self_per_cross_attn=0,
                  attn_dropout=0.3, ff_dropout=0.2)
    torch.manual_seed(31)
    model = hn.HealNet(**kw).train().to(DEV)
    flat = hn.train.flatten_parameters(model)
    return model, flat


def _batch(gen, b=4):
    return ([torch.rand(b, 1, 40, generator=gen).to(DEV), torch.rand(b, 300, 96, generator=gen).to(DEV)],
            (torch.randint(0, 4, (b,), generator=gen).to(DEV), torch.randint(0, 2, (b,), generator=gen).to(DEV)))


def _loss(hn):
    return lambda logits, y, c: hn.train.surv_nll_loss(logits, y, c).loss


def test_graphed_step_equals_eager_bit_for_bit(hn):
    model, flat = _bag_model(hn, dropout=False)
    gen = torch.Generator().manual_seed(32)
    ins, la = _batch(gen)
    step = hn.train.GraphedStep(model, _loss(hn), ins, la)
    for trial in range(3):
        ins, la = _batch(gen)                                 # new VALUES of the captured shapes
        loss_g, out_g = step(ins, la)
        loss_g, out_g, grads_g = loss_g.clone(), out_g.clone(), flat.grads.clone()
        flat.zero_grad()
        out_e = model(list(ins))
        loss_e = _loss(hn)(out_e, *la)
        loss_e.backward()
        assert torch.equal(out_g, out_e.detach()) and torch.equal(loss_g, loss_e.detach()), trial
        assert torch.equal(grads_g, flat.grads), f"trial {trial}: replayed gradients differ from the eager step"
        with torch.no_grad():                                 # parameters are read in place: an update is seen by the next replay
            flat.params.mul_(1.0 + 1e-3 * (trial + 1))
    # a new input signature (another bag length) is captured on first sight and replayed from then on (round 5; it used to raise)
    assert step.captures == 1
    short = [ins[0], ins[1][:, :100].contiguous()]
    for trial in range(2):
        loss_g, out_g = step(short, la)
        loss_g, grads_g = loss_g.clone(), flat.grads.clone()
        flat.zero_grad()
        loss_e = _loss(hn)(model(list(short)), *la)
        loss_e.backward()
        assert torch.equal(loss_g, loss_e.detach()) and torch.equal(grads_g, flat.grads)
    assert step.captures == 2
    step(ins, la)
    assert step.captures == 2                                 # the first signature's graph is still there
    with pytest.raises(ValueError, match="flatten_parameters"):
        hn.train.GraphedStep(hn.HealNet(n_modalities=1, channel_dims=[8], num_spatial_axes=[1], out_dims=2, depth=1).train().to(DEV),
                             _loss(hn), [torch.rand(2, 1, 8, device=DEV)], ())


def test_graphed_step_draws_fresh_dropout_masks(hn):
    """A captured launch bakes the Philox offset into its arguments; the device word (hn_rng.offset_dev) the graph increments makes
    every replay draw new masks, and a replay is reproduced bit for bit by an eager step run at the same counter."""
    model, flat = _bag_model(hn, dropout=True)
    assert model.runs_staged()
    gen = torch.Generator().manual_seed(33)
    ins, la = _batch(gen)
    step = hn.train.GraphedStep(model, _loss(hn), ins, la)
    baked = model._rng_offset                                 # the offset the capture baked into the graph
    losses, grads = [], []
    for _ in range(4):
        loss, _ = step(ins, la)
        losses.append(float(loss))
        grads.append(flat.grads.clone())
    assert len(set(losses)) == 4, f"replays repeated a mask: {losses}"
    assert not torch.equal(grads[0], grads[1])
    # same inputs, same counter -> same masks: eager with (offset = baked, word = w) must equal the replay that ran at word w
    w = int(step.word.item())
    loss_g, _ = step(ins, la)                                 # runs at word w + 1
    loss_g, grads_g = float(loss_g), flat.grads.clone()
    model._rng_offset = baked - 1                             # the eager forward advances it to `baked`
    flat.zero_grad()
    loss_e = _loss(hn)(model(list(ins)), *la)                 # word is w + 1 now
    loss_e.backward()
    assert int(step.word.item()) == w + 1
    assert float(loss_e) == loss_g and torch.equal(flat.grads, grads_g)
    step.close()
    assert "_hn_rng_word" not in model.__dict__


def test_graphed_step_serves_the_reference_training_loop(hn):
    """The loop body of healnet/main.py:425-467 on GraphedStep with no example batch: an "epoch" whose last batch is short and a
    loader that alternates two bag lengths -- every (shape) signature is captured once (after its eager warm-up) and replayed; the
    parameters after the epochs equal those of the same loop run eagerly (same optimizer, same data), bit for bit."""
    def run(graphed):
        model, flat = _bag_model(hn, dropout=False)
        opt = hn.train.FusedL1Adam(flat, lr=1e-3, l1=1e-5)
        step = hn.train.GraphedStep(model, _loss(hn), warmup=1) if graphed else None
        gen = torch.Generator().manual_seed(77)
        losses = []
        for epoch in range(2):
            for it, (b, n) in enumerate([(4, 300), (4, 200), (4, 300), (4, 200), (3, 300)]):      # the last batch of the epoch is short
                ins = [torch.rand(b, 1, 40, generator=gen).to(DEV), torch.rand(b, n, 96, generator=gen).to(DEV)]
                la = (torch.randint(0, 4, (b,), generator=gen).to(DEV), torch.randint(0, 2, (b,), generator=gen).to(DEV))
                if graphed:
                    loss, logits = step(ins, la)
                else:
                    opt.zero_grad()
                    logits = model(list(ins))
                    loss = _loss(hn)(logits, *la)
                    loss.backward()
                opt.step()
                losses.append(float(loss))
        if graphed:
            assert step.captures == 3, step.captures
            step.close()
        return losses, flat.params.clone()

    losses_e, params_e = run(False)
    losses_g, params_g = run(True)
    assert losses_g == losses_e
    assert torch.equal(params_g, params_e)


def test_graphed_steps_share_the_models_dropout_word(hn):
    """Two GraphedStep objects on one model: the dropout word belongs to the model and is reference-counted -- closing
    one leaves the other's replays drawing fresh masks; the last close detaches it."""
    model, flat = _bag_model(hn, dropout=True)
    gen = torch.Generator().manual_seed(35)
    ins, la = _batch(gen)
    a = hn.train.GraphedStep(model, _loss(hn), ins, la)
    b = hn.train.GraphedStep(model, _loss(hn), ins, la)
    assert a.word is b.word
    a.close()
    assert "_hn_rng_word" in model.__dict__
    losses = [float(b(ins, la)[0]) for _ in range(3)]
    assert len(set(losses)) == 3, losses
    b.close()
    assert "_hn_rng_word" not in model.__dict__
    with pytest.raises(RuntimeError, match="after close"):
        b(ins, la)


def test_scratch_of_a_capture_does_not_outlive_its_graph(hn):
    """Round 5 regression: the per-stream workspace cache used to keep a buffer that had been allocated DURING a capture, i.e. from
    that graph's private memory pool.  After the graph was destroyed the next capture on the same stream still found the buffer, and a
    replay of it faulted ("Memory access fault by GPU node").  Sequence: an inference graph with a large workspace is captured,
    replayed and destroyed; a GraphedStep then captures two signatures and replays them alternately; results equal the eager step."""
    import gc
    kw = dict(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4)
    torch.manual_seed(51)
    big = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(52)
    ins3 = [torch.rand(4, *s, generator=gen).to(DEV) for s in [(1, 2000), (64, 48, 3), (3, 40, 36, 3)]]
    g = big.capture(ins3)
    with torch.no_grad():
        assert torch.equal(g(ins3).clone(), big(ins3))
    del g, big
    gc.collect()
    torch.cuda.synchronize()
    model, flat = _bag_model(hn, dropout=False)
    ins, la = _batch(gen)
    short = [ins[0], ins[1][:, :100].contiguous()]
    step = hn.train.GraphedStep(model, _loss(hn), ins, la)
    for trial in range(3):
        for batch in (ins, short, ins):
            loss_g, _ = step(batch, la)
            loss_g, grads_g = loss_g.clone(), flat.grads.clone()
            flat.zero_grad()
            loss_e = _loss(hn)(model(list(batch)), *la)
            loss_e.backward()
            torch.cuda.synchronize()
            assert torch.equal(loss_g, loss_e.detach()) and torch.equal(grads_g, flat.grads), trial
    assert step.captures == 2
    step.close()
