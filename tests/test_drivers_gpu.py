"""The validation and export drivers against the oracle on the SAME sampled contexts (fixed seed): the reference's
`_validation_and_test_step` / `_valid_and_test_epoch_end` (pmgt/pmgt/trainer.py:162-195: eval-mode forward, 1 positive + 1
negative per target, sigmoid(logits) vs labels -> loss/val, val/auc) and `inference` (pmgt/pmgt/trainer.py:153-154,259-275;
pmgt/base_trainer.py:400-407: CLS hidden state of every node in id order -> [N, d] fp32)."""
import numpy as np
import pytest
import torch

from oracle import pmgt_oracle as po

pytestmark = pytest.mark.gpu

N, S = 300, 16
CFG = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, hidden_dropout_prob=0.1,
           attention_probs_dropout_prob=0.1, beta=0.5)         # dropout on in the config: eval / inference must ignore it


@pytest.fixture(scope="module")
def world():
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.datasets import MCNSampler
    from pmgt_amd.engine import Engine
    from pmgt_amd.graph import synthetic_graph
    graph = synthetic_graph(N, 1500, seed=8)
    ocfg = po.default_cfg(**CFG)
    params = po.synth_params(ocfg, 12)
    tables = po.synth_tables(N, ocfg["feat_hidden_sizes"], 13)
    eng = Engine(PMGTConfig(**CFG), dtype="fp32", seed=0)
    eng.load_params(params)
    eng.set_tables(*[t.numpy() for t in tables])
    return dict(eng=eng, smp=MCNSampler(graph, S - 1), ocfg=ocfg, params=params, tables=tables)


def test_evaluate_matches_oracle_on_the_same_contexts(world):
    from pmgt_amd.datasets import MODE_EVAL
    from pmgt_amd.trainer import evaluate, roc_auc_score
    from pmgt_amd.datasets import train_valid_split
    _, valid = train_valid_split(N, 0.3, seed=4)                 # the reference's split (pmgt/pmgt/trainer.py:45-52)
    bs, seed, threads = 32, 17, 3
    got = evaluate(world["eng"], world["smp"], valid, batch_size=bs, threads=threads, seed=seed)
    preds, labs, loss_sum = [], [], 0.0
    for lo in range(0, len(valid), bs):                          # ragged last batch included
        tg = valid[lo: lo + bs]
        batch = world["smp"].batch(tg, MODE_EVAL, threads=threads, base_seed=seed, counter=lo)
        assert torch.equal(batch[2], torch.full((len(tg),), 2))  # 1 positive + 1 negative per target (datasets.py:125-145)
        ref = po.pretrain_forward(world["params"], world["ocfg"], world["tables"], batch, training=False)
        preds.append(torch.sigmoid(ref["logits"]).numpy())
        labs.append(batch[3].numpy())
        loss_sum += ref["loss"].item() * len(tg)
    preds, labs = np.concatenate(preds), np.concatenate(labs)
    np.testing.assert_allclose(got["loss/val"], loss_sum / len(valid), rtol=1e-4)
    assert abs(got["val/auc"] - roc_auc_score(labs, preds)) < 1e-6
    from sklearn.metrics import roc_auc_score as sk
    assert abs(got["val/auc"] - sk(labs, preds)) < 1e-6
    # same seed -> same contexts -> same numbers; another seed re-samples the contexts (Q10)
    again = evaluate(world["eng"], world["smp"], valid, batch_size=bs, threads=threads, seed=seed)
    assert again == got
    assert evaluate(world["eng"], world["smp"], valid, batch_size=bs, threads=threads, seed=seed + 1) != got


def test_export_embeddings_matches_oracle_on_the_same_contexts(world):
    from pmgt_amd.datasets import MODE_INFERENCE
    from pmgt_amd.trainer import export_embeddings
    bs, seed, threads = 128, 23, 3
    got = export_embeddings(world["eng"], world["smp"], N, batch_size=bs, threads=threads, seed=seed)
    assert got.shape == (N, CFG["hidden_size"]) and got.dtype == np.float32
    want = np.empty_like(got)
    ids = np.arange(2, N + 2)                                    # node-id order (pmgt/pmgt/trainer.py:259-266)
    for lo in range(0, N, bs):
        tg = ids[lo: lo + bs]
        tgt = world["smp"].batch(tg, MODE_INFERENCE, threads=threads, base_seed=seed, counter=lo)
        assert np.array_equal(tgt["node_ids"][:, 0].numpy(), tg)
        h, _, _ = po.encoder_fwd(world["params"], world["ocfg"], po.gather_feats(tgt["node_ids"], world["tables"]), tgt["attention_mask"])
        want[lo: lo + len(tg)] = h[:, 0].detach().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
