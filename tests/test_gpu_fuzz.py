"""GPU: randomised whole-model parity (tools/fuzz_forward.py): random modality mixes, head / latent sizes, batch sizes, key
masks, missing modalities, weight tying -- inference forward, tape-recording forward and backward against the CPU oracle.
Every fast path has entry conditions on these shapes (context layout, GEMM routes, split geometry, one-token look-ahead);
the BASELINE configs exercise only a few of them.  (Found in round 1: the packed training layout for D = 16 / 17 on a
32-column row.)

The counts here are the in-suite sample (the GPU suite has a wall-clock budget, and every case costs an oracle
forward on the host); the wide sweeps are the same tools run by hand with larger --n (profiles/*fuzz.log)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _fuzz():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_forward.py")
    spec = importlib.util.spec_from_file_location("fuzz_forward", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [0, 1])
def test_random_configurations_forward_and_backward(seed):
    assert _fuzz().main(["--n", "20", "--seed", str(seed), "--backward"]) == 0


def test_random_configurations_forward_only_wider_sweep():
    assert _fuzz().main(["--n", "40", "--seed", "7"]) == 0


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_random_configurations_reduced_precision_cores(precision):
    """Inference forward with the bf16 / bf16x3 attention core (tolerances 2e-2 / 1e-3); the taping forward stays fp32."""
    assert _fuzz().main(["--n", "25", "--seed", "11", "--core-precision", precision]) == 0


def test_random_medium_size_configurations():
    """Larger shapes, so that the size-gated routes run: 128x128 GEMM of the patch-bag projection, split-K latent GEMMs, Q + K/V in
    one launch, one-token look-ahead, many-split merges, row-chunked GEMV above 32 samples."""
    assert _fuzz().main(["--n", "10", "--seed", "0", "--scale", "medium"]) == 0
    assert _fuzz().main(["--n", "4", "--seed", "4", "--scale", "medium", "--backward"]) == 0


def test_random_configurations_on_the_latent_chain():
    """l_d = 128 models whose latent side runs on the fused chain kernel (all head / projection kinds, masks, missing
    modalities, the verbose quirk, embeddings), plus the attention export through the chain's trace slots."""
    assert _fuzz().main(["--n", "30", "--seed", "21", "--scale", "chain"]) == 0
    assert _fuzz().main(["--n", "15", "--seed", "22", "--scale", "chain", "--attn"]) == 0
    assert _fuzz().main(["--n", "6", "--seed", "23", "--scale", "chain", "--backward"]) == 0


def test_random_configurations_under_dropout():
    """Training mode with random attention / feed-forward dropout: the oracle replays the masks the build exports
    (hn_dropout_mask).  (Found in round 1: dropout on a modality with D == 16 / 32 exactly had no binding to run on.)"""
    assert _fuzz().main(["--n", "20", "--seed", "41", "--backward", "--dropout"]) == 0


def test_random_configurations_attention_export():
    """get_attention_weights() / get_attention_importance() after the inference forward (statistics + chained trace slots)
    and after the taping forward (views of the tape), against the oracle's probabilities."""
    assert _fuzz().main(["--n", "25", "--seed", "61", "--attn"]) == 0


def test_random_op_level_blocks():
    """tools/fuzz_ops.py: PreNorm(Attention) (cross / self, masks, odd head and context widths, attn_weights on demand) and
    PreNorm(FeedForward) through the granular C-ABI entry points."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_ops.py")
    spec = importlib.util.spec_from_file_location("fuzz_ops", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(["--n", "100", "--seed", "0"]) == 0


def test_random_modality_mixes_on_the_layer_chains(monkeypatch):
    """tools/fuzz_layer_chain.py: default latent widths, random mixes / order / presence of one-token, image and patch-bag modalities,
    batch sizes on both sides of the layer chains' size gate -- the host's all-or-nothing route plan (api_fusion.hip) and the
    segment programs it builds (lchain.hip) against the oracle.  (The wide sweep: 100 cases in profiles/r06_fuzz_layer_chain.log.)"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_layer_chain.py")
    spec = importlib.util.spec_from_file_location("fuzz_layer_chain", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(["--n", "14", "--seed", "5"]) == 0
