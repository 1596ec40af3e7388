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
