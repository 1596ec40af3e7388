"""The reference's Python surface on top of the HIP engine: PMGTConfig / PMGTModel / PMGT /
PMGTForPreTrainingOutput / get_optimizer + DenseSparseAdamW / trainer step, written the way the
reference's own callers use them (pmgt/pmgt/trainer.py:118-160, pmgt/base_trainer.py:35-68), checked
against the golden vectors."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import pmgt_oracle as po
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def build(case, dtype="fp32"):
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.models import PMGT
    cfg = PMGTConfig(**case["cfg"])
    model = PMGT(node_size=case["n_nodes"], config=cfg, feat_init_emb=[t.numpy() for t in case["tables"]], dtype=dtype)
    sd = model.state_dict()
    for k, v in case["params"].items():
        sd[k] = v
    model.load_state_dict(sd)
    return model


def test_state_dict_keys_are_the_reference_keys():
    case = gu.model_case("m1")
    model = build(case)
    want = {n for n, _ in po.param_shapes(case["cfg"])} | {"feat_embeddings.0.weight", "feat_embeddings.1.weight",
                                                          "bert.embeddings.position_ids", "bert.embeddings.role_ids"}
    assert set(model.state_dict().keys()) == want
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    assert set(trainable) == {n for n, _ in po.param_shapes(case["cfg"])}
    assert not model.feat_embeddings._modules["0"].weight.requires_grad
    # parameters are views of ONE flat buffer
    lo = model.engine.params.data_ptr()
    assert all(lo <= p.data_ptr() < lo + 4 * model.engine.n_params for n, p in model.named_parameters() if p.requires_grad)


@pytest.mark.parametrize("name", ["f3", "f1", "f4"])
def test_other_modality_counts_through_the_reference_surface(name):
    """len(feat_hidden_sizes) != 2 (modeling_pmgt.py:163-173,195-201,549-569; models.py:38-54): reference key names, eval loss /
    logits / inference CLS rows and the training loss against the fixtures generated from the reference's PMGT."""
    case = gu.model_case(name)
    gold = case["gold"]
    nf = len(case["cfg"]["feat_hidden_sizes"])
    model = build(case)
    keys = set(model.state_dict().keys())
    assert {f"feat_embeddings.{i}.weight" for i in range(nf)} <= keys and f"feat_embeddings.{nf}.weight" not in keys
    assert {f"bert.embeddings.feat_linear.{i}.weight" for i in range(nf)} <= keys
    assert {f"nfr_loss.projections.{i}.bias" for i in range(nf)} <= keys
    assert tuple(model.state_dict()["bert.embeddings.attention.1.weight"].shape) == (nf, nf * case["cfg"]["hidden_size"])
    model.eval()
    batch = case["batch"]
    out = model(*batch)
    np.testing.assert_allclose(out[0].item(), gold["eval_loss"], rtol=1e-4)
    np.testing.assert_allclose(out[1].cpu().numpy(), gold["eval_logits"], rtol=1e-4, atol=1e-5)
    inf = model(batch[0])
    np.testing.assert_allclose(inf[0][:, 0].detach().cpu().numpy(), gold["inf_cls"], rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError, match="feature tensors"):       # one tensor per modality, like the reference's zip (:195-198)
        feats = [f.cuda() for f in po.gather_feats(batch[0]["node_ids"], case["tables"])]
        model.bert(*feats, feats[0])
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.engine import Engine
    with pytest.raises(ValueError, match="modalities"):
        Engine(PMGTConfig(**dict(case["cfg"], feat_hidden_sizes=[8, 8, 8, 8, 8])), dtype="fp32")


def test_forward_outputs_index_like_the_reference():
    case = gu.model_case("m1")
    gold = case["gold"]
    model = build(case).eval()
    batch = case["batch"]
    out = model(*batch)                                         # eval: loss, logits, hidden
    np.testing.assert_allclose(out[0].item(), gold["eval_loss"], rtol=1e-4)
    np.testing.assert_allclose(out[1].cpu().numpy(), gold["eval_logits"], rtol=1e-4, atol=1e-5)
    assert out.loss is out["loss"] and out.pooler_output is None and out.hidden_states is None
    tup = model(*batch, return_dict=False)
    assert tup[3] is None and torch.equal(tup[2], out.last_hidden_state)
    inf = model(batch[0])                                       # inference: loss None -> [0] is last_hidden_state
    assert inf.loss is None
    assert inf[0].requires_grad                                 # like the reference: autograd reaches the encoder weights
    np.testing.assert_allclose(inf[0][:, 0].detach().cpu().numpy(), gold["inf_cls"], rtol=1e-4, atol=1e-4)
    full = model(*batch, output_attentions=True, output_hidden_states=True)
    assert len(full.hidden_states) == 3 and len(full.attentions) == 2
    np.testing.assert_allclose(full.attentions[1].cpu().numpy(), gold["eval_attn_1"], rtol=1e-4, atol=1e-6)
    with pytest.raises(AssertionError, match="labels must be passed"):
        model(batch[0], batch[1])
    # PMGTModel.forward(*input_feat_embeds): the materialised-input entry of the reference
    feats = po.gather_feats(batch[0]["node_ids"], case["tables"])
    enc = model.bert(*feats, attention_mask=batch[0]["attention_mask"])
    np.testing.assert_allclose(enc[0].detach().cpu().numpy(), gold["eval_last_hidden"], rtol=1e-4, atol=1e-4)
    assert model.bert(*feats, return_dict=False)[1] is None


def test_autograd_step_with_reference_optimizer_surface():
    """loss = net(*batch)[0]; loss.backward(); clip; optimizer.step()  — the reference's training step."""
    from pmgt_amd.optimizers import get_optimizer
    case = gu.model_case("m1")
    gold = case["gold"]
    model = build(case).train()
    args = types.SimpleNamespace(model=model, decay=1e-2, lr=1e-3, optim="adamw", gradient_max_norm=5.0)
    opt = get_optimizer(args)
    assert len(opt.param_groups) == 2 and opt.param_groups[1]["weight_decay"] == 0.0
    ids = case["batch"][0]["node_ids"]
    for s in range(len(gold["opt_losses"])):
        masked, m2, tidx = gu.nfr_inject(gold, ids, case["n_nodes"], "opt_", f"_{s}")
        full = torch.full_like(ids, -1)
        full[:, 1:][m2] = tidx
        opt.zero_grad()
        loss = model(*case["batch"], nfr_inject=(masked.cuda(), full.cuda()))[0]
        loss.backward()
        assert model.bert.encoder.layer._modules["0"].attention.self.query.weight.grad is not None
        opt.step()
        np.testing.assert_allclose(loss.item(), gold["opt_losses"][s], rtol=2e-4)
    for k, v in model.named_parameters():
        if v.requires_grad:
            gu.check_stored(gold, "final/" + k, v.detach().cpu().numpy(), 2e-3, 2e-5)


def test_gradient_accumulation_scales_like_autograd():
    case = gu.model_case("m4")
    model = build(case).train()
    ids = case["batch"][0]["node_ids"]
    masked, m2, tidx = gu.nfr_inject(case["gold"], ids, case["n_nodes"])
    full = torch.full_like(ids, -1)
    full[:, 1:][m2] = tidx
    inj = (masked.cuda(), full.cuda())
    (model(*case["batch"], nfr_inject=inj)[0] / 2).backward()
    (model(*case["batch"], nfr_inject=inj)[0] / 2).backward()
    w = model.bert.embeddings.feat_linear._modules["0"].weight
    gu.check_stored(case["gold"], "grad/bert.embeddings.feat_linear.0.weight", w.grad.cpu().numpy(), 2e-3, 1e-6)


def test_trainer_eval_export_and_live_pipeline():
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.datasets import MCNSampler
    from pmgt_amd.engine import Engine
    from pmgt_amd.graph import synthetic_graph
    from pmgt_amd.models import reference_init, synthetic_features
    from pmgt_amd.trainer import Trainer, evaluate, export_embeddings
    n, S = 500, 16
    graph = synthetic_graph(n, 4000, seed=1)
    eng = Engine(PMGTConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64), dtype="bf16")
    reference_init(eng, 0)
    eng.set_tables(*synthetic_features(n, seed=1))
    smp = MCNSampler(graph, S - 1)
    tr = Trainer(eng, lr=1e-3, max_grad_norm=5.0)
    live = tr.run_live(smp, np.arange(2, n + 2), batch_size=32, steps=6, threads=4)
    assert live["nodes_per_s"] > 0 and np.isfinite(tr.last_loss.item())
    ev = evaluate(eng, smp, np.arange(2, 130), batch_size=64, threads=4)
    assert 0.0 <= ev["val/auc"] <= 1.0 and np.isfinite(ev["loss/val"])
    emb = export_embeddings(eng, smp, n, batch_size=256, threads=4)
    assert emb.shape == (n, 64) and emb.dtype == np.float32 and np.isfinite(emb).all()


def _live_world(seed=0):
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.datasets import MCNSampler
    from pmgt_amd.engine import Engine
    from pmgt_amd.graph import synthetic_graph
    from pmgt_amd.models import reference_init, synthetic_features
    n, S = 500, 16
    graph = synthetic_graph(n, 4000, seed=1)
    eng = Engine(PMGTConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64), dtype="bf16", seed=seed)
    reference_init(eng, 0)
    eng.set_tables(*synthetic_features(n, seed=1))
    return eng, MCNSampler(graph, S - 1), np.arange(2, n + 2)


def test_live_graph_replay_on_a_fresh_trainer_equals_eager_steps():
    """Round-4 advisor finding (high): run_live(graphs=True) on a trainer that had never stepped captured the lazy zero-fill of the Adam
    moments into slot 0's graph, so every replay of that slot reset both moments.  The moments now exist before any capture; N replayed
    steps on a fresh trainer leave the parameters N eager steps leave (same batches: the threaded sampler draws per-target counter
    streams; same dropout: device-side counters), and the moments keep growing."""
    from pmgt_amd.trainer import Trainer
    steps, res = 8, {}
    for graphs in (False, True):
        eng, smp, ids = _live_world()
        assert eng.exp_avg is None
        tr = Trainer(eng, lr=1e-3, max_grad_norm=5.0)
        tr.run_live(smp, ids, batch_size=32, steps=steps, threads=3, depth=3, graphs=graphs)
        torch.cuda.synchronize()
        assert int(eng.opt_step.item()) == steps
        res[graphs] = (eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone())
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
    assert float(res[True][2].abs().sum()) > 0


def test_live_graph_cache_follows_hyperparameters_and_keeps_its_workspace():
    """Round-4 advisor findings (medium): captured steps freeze lr / weight decay / clip as kernel arguments and address the workspace by
    raw pointer.  A changed learning rate must re-capture (lr = 0 leaves the parameters untouched); a later, larger eager call makes the
    engine allocate a new workspace while the replays keep the old one alive; drop_captured_steps() frees the option lock."""
    from pmgt_amd.trainer import Trainer
    eng, smp, ids = _live_world()
    tr = Trainer(eng, lr=1e-3, max_grad_norm=5.0)
    tr.run_live(smp, ids, batch_size=32, steps=4, threads=3, depth=2, graphs=True)
    assert len(tr._live_replays) == 2
    old_ws = eng._ws
    assert all(r.keep[0] is old_ws for r in tr._live_replays.values())
    tr.train_step(_eager_batch(smp, ids[:96]))                 # three times the rows: a larger workspace
    assert eng._ws is not old_ws and all(r.keep[0] is old_ws for r in tr._live_replays.values())
    tr.run_live(smp, ids, batch_size=32, steps=4, threads=3, depth=2, graphs=True)                      # replays over the old workspace: still valid
    torch.cuda.synchronize()
    assert torch.isfinite(eng.params).all()
    before = eng.params.clone()
    tr.lr = 0.0
    tr.weight_decay = 0.0
    tr.run_live(smp, ids, batch_size=32, steps=3, threads=3, depth=2, graphs=True)
    torch.cuda.synchronize()
    assert torch.equal(before, eng.params)                     # the lr = 1e-3 graphs were not replayed
    with pytest.raises(RuntimeError, match="captured step"):
        eng.set_option("store_ln_input", 1)
    tr.drop_captured_steps()
    import gc
    gc.collect()
    eng.set_option("store_ln_input", 1)


def test_replays_interleaved_with_a_larger_eager_step_equal_the_all_eager_run():
    """Round-5 advisor finding: a captured step keeps replaying on the workspace it was captured over, while a later, larger eager call
    makes the engine carve a NEW workspace -- capture_step's public API allows both to alternate.  What a step needs from earlier steps
    (parameters, moments, the RNG / AdamW counters) lives in the engine, and each forward rebuilds its weight mirrors in its own workspace
    from the parameters: replay, replay, larger eager step, replay leaves exactly the parameters and moments the same four steps leave
    when all of them run eagerly."""
    from pmgt_amd.trainer import Trainer
    res = {}
    for graphs in (False, True):
        eng, smp, ids = _live_world()
        tr = Trainer(eng, lr=1e-3, max_grad_norm=5.0)
        small = [_eager_batch(smp, ids[32 * i: 32 * (i + 1)]) for i in range(3)]
        big = _eager_batch(smp, ids[100:196])
        static = tuple({k: v.clone() for k, v in x.items()} if isinstance(x, dict) else x.clone() for x in small[0])

        def feed(b):
            for dst, src in zip(static, b):
                if isinstance(dst, dict):
                    for k in dst:
                        dst[k].copy_(src[k])
                else:
                    dst.copy_(src)
        if graphs:
            replay = tr.capture_step(static, warmup=2)        # two eager steps on batch 0, then the capture (not executed)
            ws0 = eng._ws
            step_small = lambda b: (feed(b), replay())
        else:
            tr.train_step(small[0])
            tr.train_step(small[0])
            step_small = lambda b: tr.train_step(b)
        step_small(small[0])
        step_small(small[1])
        tr.train_step(big)                                    # three times the rows: the engine re-allocates its workspace
        if graphs:
            assert eng._ws is not ws0 and replay.keep[0] is ws0
        step_small(small[2])
        tr.train_step(big)
        torch.cuda.synchronize()
        assert int(eng.opt_step.item()) == 7
        res[graphs] = (eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone())
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)


def test_eager_steps_after_run_live_graphs_survive_a_layernorm_carrier_flip():
    """Round-5 advisor finding: the periodic carrier guard of an EAGER optimizer step raised whenever captured steps were alive -- a user
    mixing run_live(graphs=True) with later eager train_step calls met a RuntimeError at the flip.  The trainer now drops the captured steps
    it owns, switches, and run_live re-captures on its next call; a replay handle the CALLER holds still gets the engine's error."""
    from pmgt_amd.trainer import Trainer
    eng, smp, ids = _live_world()
    tr = Trainer(eng, lr=1e-4, max_grad_norm=5.0, check_carrier_every=2)
    tr.run_live(smp, ids, batch_size=32, steps=4, threads=3, depth=2, graphs=True)
    assert len(tr._live_replays) == 2 and not eng.get_option("store_ln_input")
    eng.view("bert.encoder.layer.0.output.LayerNorm.bias").fill_(20.0)        # |beta / gamma| = 20 > 8, behind the engine's back
    b = _eager_batch(smp, ids[:32])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr.train_step(b)
        tr.train_step(b)                                      # optimizer step 6: the guard runs, drops the trainer's replays, switches
    assert eng.get_option("store_ln_input") and len(tr._live_replays) == 0
    tr.run_live(smp, ids, batch_size=32, steps=3, threads=3, depth=2, graphs=True)      # re-captured under the new option
    torch.cuda.synchronize()
    assert len(tr._live_replays) == 2 and torch.isfinite(eng.params).all()
    # a handle the caller holds cannot be dropped by the trainer: the error names the remedy
    eng2, smp2, ids2 = _live_world()
    tr2 = Trainer(eng2, lr=1e-4, max_grad_norm=5.0, check_carrier_every=1)
    b2 = _eager_batch(smp2, ids2[:32])
    replay = tr2.capture_step(b2, warmup=1)
    eng2.view("bert.encoder.layer.0.output.LayerNorm.bias").fill_(20.0)
    with pytest.raises(RuntimeError, match="captured steps of this engine are alive"):
        tr2.train_step(b2)
    del replay


def _eager_batch(smp, targets):
    from pmgt_amd.datasets import MODE_TRAIN
    tgt, pair, num_pairs, labels = smp.batch(np.asarray(targets), MODE_TRAIN, threads=2, base_seed=3, counter=0)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}
    return cu(tgt), cu(pair), num_pairs.cuda(), labels.cuda()


def test_live_graph_replays_run_the_layernorm_carrier_guard():
    """Round-4 advisor finding (medium): replays never ran the guard that switches the backward to stored LayerNorm inputs when
    |beta / gamma| drifts past the bound.  run_live now looks every `check_carrier_every` replays, drops the captured steps, switches and
    re-captures."""
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.datasets import MCNSampler
    from pmgt_amd.engine import Engine
    from pmgt_amd.graph import synthetic_graph
    from pmgt_amd.models import reference_init, synthetic_features
    from pmgt_amd.trainer import Trainer
    n, S = 300, 32
    graph = synthetic_graph(n, 3000, seed=2)
    eng = Engine(PMGTConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=8, intermediate_size=256), dtype="bf16")
    reference_init(eng, 0)
    eng.set_tables(*synthetic_features(n, seed=2))
    smp = MCNSampler(graph, S - 1)
    tr = Trainer(eng, lr=1e-4, max_grad_norm=5.0, check_carrier_every=2)
    ids = np.arange(2, n + 2)
    tr.run_live(smp, ids, batch_size=16, steps=4, threads=3, depth=2, graphs=True)
    assert not eng.get_option("store_ln_input") and len(tr._live_replays) == 2
    eng.view("bert.encoder.layer.0.output.LayerNorm.bias").fill_(20.0)        # |beta / gamma| = 20 > 8 (set behind the engine's back)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr.run_live(smp, ids, batch_size=16, steps=6, threads=3, depth=2, graphs=True)
    torch.cuda.synchronize()
    assert eng.get_option("store_ln_input") and len(tr._live_replays) == 2
    assert torch.isfinite(eng.params).all() and np.isfinite(tr.last_loss.item())


@pytest.mark.parametrize("name", list(gu.NCF_CASES))
def test_ncf_second_caller_on_hip_encoder(name):
    """PMGT_NCF with the item tower on the HIP engine (pmgt_encode_train / pmgt_encode_backward through autograd):
    logits, loss, encoder gradients and head gradients equal the reference's (tests/golden/ncf_*.npz)."""
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.pmgt_ncf import PMGT_NCF
    c = gu.ncf_case(name)
    gold = c["gold"]
    model = PMGT_NCF(user_num=c["users"], item_num=c["n_nodes"], factor_num=c["factor"], num_layers=c["num_layers"],
                     model=c["model"], config=PMGTConfig(**c["cfg"]), dtype="fp32")
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in list(c["params"].items()) + list(c["head"].items()):
            sd[k].copy_(v)
    model.set_features([t.numpy() for t in c["tables"]])
    model.train()
    logits = model(c["user"], c["item"])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, c["labels"].cuda())
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), gold["logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(loss.item(), gold["loss"], rtol=1e-4)
    bad = []
    for k, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, k                   # frozen feature tables
            continue
        g = p.grad.cpu().numpy().astype(np.float64)
        scale = float(np.sqrt((g ** 2).mean()))
        try:
            gu.check_stored(gold, "grad/" + k, g, 2e-3, 2e-3 * scale + 1e-8)
        except AssertionError as e:
            bad.append((k, str(e).splitlines()[3:6]))
    assert not bad, bad
    # eval-mode call without autograd goes through the plain inference entry and gives the same logits (dropout is 0)
    model.eval()
    with torch.no_grad():
        np.testing.assert_allclose(model(c["user"], c["item"]).cpu().numpy(), gold["logits"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["m2", "f3", "f1", "f4"])      # two modalities as the trainer builds; three, one and four
def test_encoder_backward_on_materialised_features_matches_oracle(name):
    """PMGTModel.forward(*input_feat_embeds) with autograd (the reference's own call form, pmgt_ncf/models.py:83-89)
    against torch autograd through the CPU oracle: arbitrary upstream gradient on the whole last_hidden_state."""
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.modeling_pmgt import PMGTModel
    case = gu.model_case(name)
    cfg = case["cfg"]
    model = PMGTModel(PMGTConfig(**cfg), dtype="fp32")
    with torch.no_grad():
        sd = model.state_dict()
        for k, v in case["params"].items():
            if k.startswith("bert."):
                sd[k[5:]].copy_(v)
    tgt = case["batch"][1]
    feats = po.gather_feats(tgt["node_ids"], case["tables"])
    w = torch.from_numpy(np.random.RandomState(3).standard_normal(tuple(tgt["node_ids"].shape) + (cfg["hidden_size"],)).astype(np.float32))
    out = model(*[f.cuda() for f in feats], attention_mask=tgt["attention_mask"].cuda())[0]
    (out * w.cuda()).sum().backward()
    p = {k: v.clone().requires_grad_(True) for k, v in case["params"].items() if k.startswith("bert.")}
    ref = po.encoder_fwd(p, cfg, feats, tgt["attention_mask"])[0]
    (ref * w).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    grms = max(float(v.grad.double().pow(2).mean().sqrt()) for v in p.values())
    for k, v in model.named_parameters():
        a, b = v.grad.cpu().numpy(), p["bert." + k].grad.numpy()
        scale = float(np.sqrt((b.astype(np.float64) ** 2).mean()))
        # key.bias gradients are exactly 0 in theory (softmax shift invariance): absolute floor from the global scale
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-3 * scale + 1e-5 * grms, err_msg=k)


def test_lightning_style_checkpoint_roundtrip(tmp_path):
    """Checkpoints carry the reference's key names under PL's `net.` prefix (pmgt/base_trainer.py:99-110): save from
    one model, load into a fresh one (different init) and get identical eval outputs; a bare state_dict loads too."""
    from pmgt_amd import io as pio
    case = gu.model_case("m1")
    a = build(case).eval()
    path = str(tmp_path / "last.ckpt")
    pio.save_checkpoint(a, path, epoch=3)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    keys = set(ck["state_dict"])
    assert ck["epoch"] == 3 and all(k.startswith("net.") for k in keys)
    assert {"net." + n for n, _ in po.param_shapes(case["cfg"])} <= keys
    assert {"net.feat_embeddings.0.weight", "net.feat_embeddings.1.weight", "net.bert.embeddings.position_ids",
            "net.bert.embeddings.role_ids"} <= keys
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.models import PMGT
    b = PMGT(node_size=case["n_nodes"], config=PMGTConfig(**case["cfg"]), dtype="fp32", seed=123).eval()
    pio.load_checkpoint(b, path)
    with torch.no_grad():
        oa, ob = a(*case["batch"]), b(*case["batch"])
    assert torch.equal(oa.prediction_logits, ob.prediction_logits) and oa.loss.item() == ob.loss.item()
    c = PMGT(node_size=case["n_nodes"], config=PMGTConfig(**case["cfg"]), dtype="fp32", seed=7).eval()
    pio.load_checkpoint(c, a.state_dict())
    with torch.no_grad():
        assert torch.equal(c(*case["batch"]).prediction_logits, oa.prediction_logits)


@pytest.mark.gpu
def test_graft_entry_build_then_smoke_in_one_process():
    """build() loads the HIP library before smoke() touches torch.cuda: the library must bind PyTorch-ROCm's HIP runtime,
    not the system one (a process that mapped the system runtime first could not initialise the device)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke ok" in r.stdout
