"""Per-engine path options (include/pmgt_ops.h) and the engine-state regressions of round 2's review:
two engines with different options in one process, the separate backward call at the headline hidden size (the
LayerNorm-from-output decision must be re-derived, not remembered), a partial-sum arena that flushes between the two
regions of one producer, the output ring under changing pair counts, graph capture with outputs of its own."""
import numpy as np
import pytest
import torch

from tests import golden_util as gu
from tests.test_engine_gpu import dev_batch, inject_for, make_engine

pytestmark = pytest.mark.gpu


def test_two_engines_with_different_options_keep_their_own_paths():
    case = gu.model_case("m3")          # d = 256, H = 8, S = 32: every fused kernel applies
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    a = make_engine(case, dtype="bf16")
    b = make_engine(case, dtype="bf16")
    for k in ("no_fused_qkvc_attention", "no_fused_attention_bwd", "store_ln_input"):
        b.set_option(k, 1)
        assert b.get_option(k) and not a.get_option(k)
    with pytest.raises(RuntimeError, match="unknown option"):
        a.set_option("no_such_option", 1)
    with pytest.raises(KeyError):
        a.get_option("no_such_option")
    prof = {}
    for rnd in range(2):                # interleaved: neither engine's choice leaks into the other's next step
        for name, eng in (("a", a), ("b", b)):
            eng.rng_state[1] = 0
            eng.profile_begin()
            out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
            prof[name] = eng.profile_end()
            assert np.isfinite(out["loss"].item())
    assert "fwd.qkvc_attention" in prof["a"] and "bwd.attention_wgrad" in prof["a"]
    assert "fwd.qkvc_attention" not in prof["b"] and "bwd.attention_wgrad" not in prof["b"] and "bwd.wgrad_qkvc" in prof["b"]
    cos = torch.nn.functional.cosine_similarity(a.grads, b.grads, dim=0).item()
    assert cos > 0.999, cos
    b.set_option("no_fused_qkvc_attention", 0)
    b.profile_begin()
    b.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
    assert "fwd.qkvc_attention" in b.profile_end()


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_separate_backward_call_at_hidden_256_rederives_the_layernorm_carrier(dropout):
    """pmgt_encode_train + pmgt_encode_backward (PMGT_NCF, PMGTModel autograd) in bf16 at d = 256: the forward does not store the
    pre-LayerNorm sums, and the backward -- a separate call that carves its own buffers -- must know that (round 2 read the
    never-written buffers there).  Checked against the same engine with stored inputs and against the fp32 engine."""
    case = gu.model_case("m3")
    tgt = case["batch"][1]              # the pair sequences: 10 x B sequences of 32
    ids, mask = tgt["node_ids"].cuda(), tgt["attention_mask"].cuda()
    w = torch.from_numpy(np.random.RandomState(5).standard_normal(tuple(ids.shape) + (256,)).astype(np.float32)).cuda()
    grads = {}
    for key, dtype, stored in (("y", "bf16", 0), ("x", "bf16", 1), ("f32", "fp32", 0)):
        eng = make_engine(case, dtype=dtype, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
        eng.set_option("store_ln_input", stored)
        g = torch.Generator().manual_seed(3)
        for l in range(case["cfg"]["num_hidden_layers"]):
            for ln in ("attention.output.LayerNorm", "output.LayerNorm"):       # gamma away from 1, beta away from 0
                eng.view(f"bert.encoder.layer.{l}.{ln}.weight").copy_((0.7 + 0.6 * torch.rand(256, generator=g)).cuda())
                eng.view(f"bert.encoder.layer.{l}.{ln}.bias").copy_((0.2 * torch.randn(256, generator=g)).cuda())
        # the workspace of the call comes from torch's caching allocator: hand it a block full of NaN bit patterns, so that a
        # backward pass that reads a buffer the forward never wrote shows up as NaN gradients, not as plausible noise
        nbytes = int(eng.lib.pmgt_workspace_bytes(eng.h, ids.shape[0], ids.shape[1], 1, 1))
        poison = torch.full((nbytes // 4 + 1,), float("nan"), device="cuda")[: nbytes // 4].view(torch.uint8)[:nbytes]
        torch.cuda.synchronize()
        del poison
        last, state = eng.encode_train(ids=ids, attention_mask=mask, training=dropout > 0)
        eng.grads.fill_(float("nan"))
        eng.encode_backward(state, w.to(last.dtype))
        torch.cuda.synchronize()
        grads[key] = eng.grads.clone()
        n_bert = eng.entry("nfr_loss.projections.0.weight")["offset"]
        assert torch.isfinite(grads[key][:n_bert]).all(), key
        grads[key] = grads[key][:n_bert]
    cos_xy = torch.nn.functional.cosine_similarity(grads["y"], grads["x"], dim=0).item()
    assert cos_xy > 0.999, cos_xy
    if dropout == 0.0:
        for key in ("y", "x"):
            cos = torch.nn.functional.cosine_similarity(grads[key], grads["f32"], dim=0).item()
            assert cos > 0.995, (key, cos)


def test_partial_sum_arena_that_holds_one_producer_gives_the_same_gradients():
    """With the arena sized for ONE producer every take flushes what is queued; a flush between the slab and the bias slab of one
    weight-gradient launch would rewind the arena under a region that is handed out but not queued yet."""
    case = gu.model_case("m3")
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    res = []
    for small in (0, 1):
        eng = make_engine(case, dtype="bf16", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        eng.set_option("small_arena", small)
        eng.profile_begin()
        out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
        prof = eng.profile_end()
        res.append((out["loss"].item(), eng.grads.clone(), prof["bwd.slab_reduce"][0]))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])          # the reductions sum in a fixed order whatever launch carries them
    assert res[1][2] > res[0][2]                      # ... and the small arena really did flush more often


def test_output_ring_follows_the_pair_count_and_graph_capture_owns_its_outputs():
    from pmgt_amd.trainer import Trainer
    case = gu.model_case("m3")
    tgt, pair, num_pairs, labels = dev_batch(case["batch"])
    eng = make_engine(case, dtype="bf16")
    B = tgt["node_ids"].shape[0]
    full = eng.pretrain_step((tgt, pair, num_pairs, labels), training=False, want_hidden=False)
    ref_logits = full["logits"].clone()
    ptrs = set()
    for drop in range(0, 6):            # a different pair count per step, as live eval / training batches have
        np_ = num_pairs.clone()
        np_[-1] -= min(drop, int(np_[-1]) - 1)
        P = int(np_.sum())
        sub = ({k: v for k, v in tgt.items()}, {k: v[:P] for k, v in pair.items()}, np_, labels[:P])
        out = eng.pretrain_step(sub, training=False, want_hidden=False)
        assert out["logits"].shape[0] == P
        # (bf16: a different token count selects other GEMM tiles for the same rows -- rounding noise of a few 1e-3 on logits of O(0.1))
        torch.testing.assert_close(out["logits"][: P - 1], ref_logits[: P - 1], rtol=1e-2, atol=4e-3)
        ptrs.add(out["logits"].data_ptr())
    assert len(ptrs) <= eng.OUTPUT_RING            # one ring serves every pair count: no allocation per new P
    assert len(eng._out_rings) == 1
    for b in range(1, eng.OUTPUT_RINGS_MAX + 3):   # many shapes: least-recently-used rings go, never all at once
        eng._outputs(b, 10 * b, 32, False)
        assert 1 <= len(eng._out_rings) <= eng.OUTPUT_RINGS_MAX
    # a captured step writes outputs of its own: eager steps of the same shape do not clobber them
    tr = Trainer(eng, lr=1e-3, max_grad_norm=5.0)
    batch = (tgt, pair, num_pairs, labels)
    replay = tr.capture_step(batch, warmup=1)
    l1 = replay().clone()
    torch.cuda.synchronize()
    held = replay.outputs["logits"]
    snap = held.clone()
    for _ in range(eng.OUTPUT_RING + 1):
        tr.train_step(batch)
    torch.cuda.synchronize()
    assert torch.equal(held, snap)                 # untouched by the eager steps
    assert np.isfinite(l1.item()) and replay.outputs["loss"].data_ptr() == replay().data_ptr()


def test_layernorm_carrier_guard_switches_to_stored_inputs():
    case = gu.model_case("m3")
    eng = make_engine(case, dtype="bf16")
    assert not eng.get_option("store_ln_input") and eng.check_layernorm_carrier() < 1.0
    bad = {k: v.clone() for k, v in case["params"].items()}
    bad["bert.encoder.layer.1.output.LayerNorm.weight"][7] = 1e-3
    bad["bert.encoder.layer.1.output.LayerNorm.bias"][7] = 0.5
    with pytest.warns(UserWarning, match="store_ln_input"):
        eng.load_params(bad)
    assert eng.get_option("store_ln_input")
    out = eng.pretrain_step(dev_batch(case["batch"]), training=True, backward=True, want_hidden=False)
    assert np.isfinite(out["loss"].item()) and torch.isfinite(eng.grads).all()


def test_module_load_state_dict_runs_the_layernorm_carrier_guard():
    """Round-3 advisor finding: PMGT.load_state_dict writes a checkpoint through the Parameter views and never went through
    Engine.load_params, so a checkpoint with |beta / gamma| >> 1 (or gamma == 0) trained on x^ recovered from the bf16 output."""
    from tests.test_surface_gpu import build
    case = gu.model_case("m3")
    model = build(case, dtype="bf16")
    assert not model.engine.get_option("store_ln_input")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd["bert.encoder.layer.0.attention.output.LayerNorm.weight"][3] = 0.0       # a dead channel: x^ is not recoverable from y at all
    with pytest.warns(UserWarning, match="store_ln_input"):
        model.load_state_dict(sd)
    assert model.engine.get_option("store_ln_input")
    # the encoder module on its own engine takes the same route
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.modeling_pmgt import PMGTModel
    enc = PMGTModel(PMGTConfig(**case["cfg"]), dtype="bf16")
    esd = {k: v.clone() for k, v in enc.state_dict().items()}
    esd["encoder.layer.1.output.LayerNorm.weight"][5] = 1e-3
    esd["encoder.layer.1.output.LayerNorm.bias"][5] = 1.0
    with pytest.warns(UserWarning, match="store_ln_input"):
        enc.load_state_dict(esd)
    assert enc.engine.get_option("store_ln_input")


def test_backward_refuses_options_changed_since_the_forward():
    """Round-3 advisor finding: the backward re-derives which buffers the forward filled from the CURRENT options; a flip between
    pmgt_encode_train and pmgt_encode_backward read never-written buffers.  Now the forward's bits are remembered per workspace."""
    case = gu.model_case("m3")
    eng = make_engine(case, dtype="bf16")
    tgt = case["batch"][1]
    ids, mask = tgt["node_ids"].cuda(), tgt["attention_mask"].cuda()
    last, state = eng.encode_train(ids=ids, attention_mask=mask, training=True)
    eng.set_option("store_ln_input", 1)
    with pytest.raises(RuntimeError, match="options changed since the forward"):
        eng.encode_backward(state, torch.ones_like(last))
    eng.set_option("store_ln_input", 0)                  # back to what the forward saw: accepted
    eng.encode_backward(state, torch.ones_like(last))
    eng.set_option("eager_reduce", 1)                    # scheduling-only bits do not count
    last, state = eng.encode_train(ids=ids, attention_mask=mask, training=True)
    eng.set_option("eager_reduce", 0)
    eng.encode_backward(state, torch.ones_like(last))
    torch.cuda.synchronize()
    assert torch.isfinite(eng.grads).all()


def test_options_are_frozen_while_a_captured_step_lives():
    import gc
    from pmgt_amd.trainer import Trainer
    case = gu.model_case("m3")
    eng = make_engine(case, dtype="bf16")
    tr = Trainer(eng, lr=1e-3, max_grad_norm=5.0)
    replay = tr.capture_step(dev_batch(case["batch"]), warmup=1)
    replay()
    eng.set_option("store_ln_input", 0)                  # no change: fine
    with pytest.raises(RuntimeError, match="captured step"):
        eng.set_option("store_ln_input", 1)
    torch.cuda.synchronize()
    del replay
    gc.collect()
    eng.set_option("store_ln_input", 1)                  # the graph is gone


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_attention_backward_pairs_one_interval_apart_is_bit_identical_to_lockstep(dropout):
    """The fused attention backward runs the two (sequence, head) pairs of a step one barrier interval apart (pair 1 in phase k - 1 while
    pair 0 is in phase k); `lockstep_attention_bwd` keeps both in the same phase.  Same arithmetic per element, different schedule:
    every gradient must agree BIT FOR BIT, with an odd number of sequences (a last step with one live pair) and with dropout."""
    case = gu.model_case("m3")          # d = 256, H = 8, S = 32
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    outs = []
    for lock in (0, 1):
        eng = make_engine(case, dtype="bf16", hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
        eng.set_option("lockstep_attention_bwd", lock)
        eng.profile_begin()
        out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
        assert "bwd.attention_wgrad" in eng.profile_end()
        torch.cuda.synchronize()
        outs.append((out["loss"].item(), eng.grads.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.isfinite(outs[0][1]).all() and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("S,Tq,dropout", [(31, 7, 0.0), (32, 5, 0.1), (9, 13, 0.1)])
def test_embedding_sum_is_recomputed_in_the_backward_at_ragged_shapes(S, Tq, dropout):
    """Table mode at hidden size 256 in bf16: the token phase of the embedding (embed_tok8_fwd/bwd_kernel: two positions x four sequences per
    wave) does not store its pre-LayerNorm sum, the backward recomputes it from the mixed row + position + role rows
    (pmgt/pmgt/modeling_pmgt.py:199-208).  An odd S leaves the last pair's second half-wave without a position, a sequence count that
    is not a multiple of four leaves row slots empty: both must neither write nor count.  Checked against the per-token path
    (no_segment_sum: generic kernels, stored sum) and against the fp32 engine."""
    from oracle import pmgt_oracle as po
    n = 40
    cfg = po.default_cfg(hidden_size=256, num_attention_heads=8, num_hidden_layers=1, intermediate_size=256, hidden_dropout_prob=dropout,
                         attention_probs_dropout_prob=0.0)
    case = dict(cfg=cfg, params=po.synth_params(cfg, 7), tables=po.synth_tables(n, cfg["feat_hidden_sizes"], 8), n_nodes=n)
    g = torch.Generator().manual_seed(100 * S + Tq)
    ids = torch.randint(2, n + 2, (Tq, S), generator=g)
    mask = torch.ones(Tq, S)
    ids[1, S - 3:] = 0
    mask[1, S - 3:] = 0
    w = torch.randn(Tq, S, 256, generator=g)
    res = {}
    for key, dtype, off in (("tok8", "bf16", 0), ("per_token", "bf16", 1), ("f32", "fp32", 0)):
        eng = make_engine(case, dtype=dtype)
        eng.set_option("no_segment_sum", off)
        eng.view("bert.embeddings.LayerNorm.weight").copy_((0.7 + 0.6 * torch.rand(256, generator=torch.Generator().manual_seed(3))).cuda())
        eng.view("bert.embeddings.LayerNorm.bias").copy_((0.2 * torch.randn(256, generator=torch.Generator().manual_seed(4))).cuda())
        nbytes = int(eng.lib.pmgt_workspace_bytes(eng.h, Tq, S, 1, 1))
        poison = torch.full((nbytes // 4 + 1,), float("nan"), device="cuda")      # (a read of a never-written buffer shows up as NaN)
        torch.cuda.synchronize()
        del poison
        last, state = eng.encode_train(ids=ids.cuda(), attention_mask=mask.cuda(), training=dropout > 0)
        eng.grads.fill_(float("nan"))
        eng.encode_backward(state, w.cuda().to(last.dtype))
        torch.cuda.synchronize()
        n_bert = eng.entry("nfr_loss.projections.0.weight")["offset"]
        assert torch.isfinite(eng.grads[:n_bert]).all() and torch.isfinite(last.float()).all(), key
        res[key] = (last.float().clone(), eng.grads[:n_bert].clone(), eng)
    # same dropout masks (same seed, step, site, row, column) in both bf16 paths: they differ by summation order only
    assert float((res["tok8"][0] - res["per_token"][0]).abs().max()) < 8e-2
    cos = torch.nn.functional.cosine_similarity(res["tok8"][1], res["per_token"][1], dim=0).item()
    assert cos > 0.999, cos
    eng = res["tok8"][2]
    for k in ("bert.embeddings.LayerNorm.weight", "bert.embeddings.LayerNorm.bias", "bert.embeddings.position_embeddings.weight",
              "bert.embeddings.role_embeddings.weight", "bert.embeddings.feat_linear.0.weight"):
        e = eng.entry(k)
        a = res["tok8"][1][e["offset"]: e["offset"] + e["numel"]]
        b = res["per_token"][1][e["offset"]: e["offset"] + e["numel"]]
        rel = ((a - b).norm() / b.norm()).item()
        assert rel < 2e-2, (k, rel)
        if k.startswith("bert.embeddings.position"):       # rows of positions >= S get no gradient at all
            assert float(a.view(*e["shape"])[S:].abs().max()) == 0.0
        if dropout == 0.0:
            c = res["f32"][1][e["offset"]: e["offset"] + e["numel"]]
            assert ((a - c).norm() / c.norm()).item() < 3e-2, k
    if dropout == 0.0:
        assert float((res["tok8"][0] - res["f32"][0]).abs().max()) < 8e-2


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_side_stream_weight_gradients_are_bit_identical_to_in_line_ones(dtype):
    """Option side_stream_wgrad: the dense weight-gradient GEMMs of a layer run on the engine's side stream next to the data-gradient chain
    (fork behind their operands' producers, joined before a launch overwrites an operand and before anything sums their slabs).  Same
    kernels, same slabs, same reduction order: gradients and the updated parameters must be EQUAL to the in-line schedule's, step after
    step (a missing wait shows up as a difference or a NaN here).  Opt-in: it measured slower (profiles/r04/NOTES.md section 9)."""
    case = gu.model_case("m3")
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    res = {}
    for key, on in (("side", 1), ("inline", 0)):
        eng = make_engine(case, dtype=dtype, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        eng.set_option("side_stream_wgrad", on)
        gs = []
        for step in range(3):
            eng.grads.fill_(float("nan"))
            out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
            gs.append(eng.grads.clone())
            eng.optimizer_step(lr=1e-3, weight_decay=1e-2, max_grad_norm=5.0)
        torch.cuda.synchronize()
        res[key] = (gs, eng.params.clone(), out["loss"].item())
    for a, b in zip(res["side"][0], res["inline"][0]):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)
    assert torch.equal(res["side"][1], res["inline"][1]) and res["side"][2] == res["inline"][2]


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_shortcut_layer_attention_backward_skips_gradient_free_query_tiles_exactly(dropout):
    """Last layer of the training fast path: target and pair sequences are read at row 0 only, so rows 16 .. 31 of their d ctx are zero and the
    fused attention backward lets the waves of that query tile skip scores / softmax / softmax backward (they write zero dS / P images, zero
    dQ rows and a zero dC half: what the full computation produces, as exact zeros).  Must be bit-identical to running them
    (option no_cls_only_attention_bwd), gradients and loss, with and without dropout."""
    case = gu.model_case("m3")
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    res = []
    for off in (0, 1):
        eng = make_engine(case, dtype="bf16", hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
        eng.set_option("no_cls_only_attention_bwd", off)
        eng.grads.fill_(float("nan"))
        eng.profile_begin()
        out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
        prof = eng.profile_end()
        assert prof["bwd.attention_wgrad"][0] == case["cfg"]["num_hidden_layers"]       # the fused kernel ran on every layer, the shortcut one included
        res.append((out["loss"].item(), eng.grads.clone()))
    assert torch.isfinite(res[0][1]).all()
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])
