"""End-to-end parity of the HIP engine (through the C ABI) against the golden vectors produced by the
reference (tests/golden/*.npz) and against the CPU oracle on the same seeded inputs.

Tolerances: fp32 parity mode — losses rtol 1e-4 (the bar BASELINE.json's north_star states), logits
and hidden states 1e-4, gradients rtol 2e-3 / atol 2e-5 of the tensor scale; bf16 perf mode — loss
rtol 2e-2 and gradient cosine similarity > 0.99 (the reference has no bf16 path to compare with)."""
import numpy as np
import pytest
import torch

from oracle import pmgt_oracle as po
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def make_engine(case, dtype="fp32", **cfg_over):
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.engine import Engine
    cfg = dict(case["cfg"])
    cfg.update(cfg_over)
    eng = Engine(PMGTConfig(**cfg), dtype=dtype, seed=0)
    eng.load_params(case["params"])
    eng.set_tables(*[t.numpy() for t in case["tables"]])
    return eng


def dev_batch(batch):
    tgt, pair, num_pairs, labels = batch
    cu = lambda d: {k: v.cuda() for k, v in d.items()}
    return cu(tgt), cu(pair), num_pairs.cuda(), labels.cuda()


def inject_for(case, prefix="nfr_", suffix=""):
    ids = case["batch"][0]["node_ids"]
    masked, m2, tidx = gu.nfr_inject(case["gold"], ids, case["n_nodes"], prefix, suffix)
    full = torch.full_like(ids, -1)
    full[:, 1:][m2] = tidx
    return (masked.cuda(), full.cuda()), (masked, m2, tidx)


@pytest.mark.parametrize("name", list(gu.MODEL_CASES))
def test_eval_forward_matches_reference(name):
    case = gu.model_case(name)
    gold = case["gold"]
    eng = make_engine(case)
    out = eng.pretrain_step(dev_batch(case["batch"]), training=False)
    np.testing.assert_allclose(out["loss"].item(), gold["eval_loss"], rtol=1e-4)
    np.testing.assert_allclose(out["logits"].cpu().numpy(), gold["eval_logits"], rtol=1e-4, atol=1e-5)
    gu.check_stored(gold, "eval_last_hidden", out["last_hidden_state"].float().cpu().numpy(), 1e-4, 1e-4)
    tgt = case["batch"][0]
    last, hs, pr = eng.encode(ids=tgt["node_ids"].cuda(), attention_mask=tgt["attention_mask"].cuda(),
                              output_hidden_states=True, output_attentions=True)
    for i in range(case["cfg"]["num_hidden_layers"] + 1):
        gu.check_stored(gold, f"eval_hidden_{i}", hs[i].float().cpu().numpy(), 1e-4, 1e-4)
    for i in range(case["cfg"]["num_hidden_layers"]):
        gu.check_stored(gold, f"eval_attn_{i}", pr[i].cpu().numpy(), 1e-4, 1e-6)
    np.testing.assert_allclose(last[:, 0].float().cpu().numpy(), gold["inf_cls"], rtol=1e-4, atol=1e-4)
    # materialised-feature entry (PMGTModel.forward(*input_feat_embeds)) gives the same result
    feats = po.gather_feats(tgt["node_ids"], case["tables"])
    last2, _, _ = eng.encode(feats=[f.cuda() for f in feats], attention_mask=tgt["attention_mask"].cuda())
    np.testing.assert_allclose(last2.float().cpu().numpy(), last.float().cpu().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", list(gu.MODEL_CASES))
def test_train_loss_and_grads_match_reference(name):
    case = gu.model_case(name)
    gold = case["gold"]
    eng = make_engine(case)
    inj, _ = inject_for(case)
    out = eng.pretrain_step(dev_batch(case["batch"]), training=True, backward=True, nfr_inject=inj)
    np.testing.assert_allclose(out["loss"].item(), gold["train_loss"], rtol=1e-4)
    np.testing.assert_allclose(out["logits"].cpu().numpy(), gold["train_logits"], rtol=1e-4, atol=1e-5)
    bad = []
    for k, g in eng.named_views(grad=True).items():
        a = g.cpu().numpy().astype(np.float64)
        scale = float(gold["grad/" + k + "@norm"]) / np.sqrt(a.size) if ("grad/" + k) not in gold.files else \
            float(np.sqrt((gold["grad/" + k].astype(np.float64) ** 2).mean()))
        try:
            gu.check_stored(gold, "grad/" + k, a, 2e-3, 2e-3 * scale + 1e-9)
        except AssertionError as e:
            bad.append((k, str(e).splitlines()[3:6]))
    assert not bad, bad
    # accumulate flag: a second identical call doubles the gradients
    g1 = eng.grads.clone()
    eng.rng_state[1] = 0
    eng.pretrain_step(dev_batch(case["batch"]), training=True, backward=True, accumulate=True, nfr_inject=inj)
    torch.testing.assert_close(eng.grads, 2 * g1, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("name", ["m1_beta1", "m1_beta0", "e_script"])
def test_dead_attention_branch_is_skipped_and_its_gradients_are_exact_zeros(name):
    """beta in {0, 1} (the author's script trains with --beta 1.0, scripts/run_pmgt.sh:24): one softmax branch is multiplied by exactly 0
    (pmgt/pmgt/modeling_pmgt.py:519-521).  The default path skips it (test_train_loss_and_grads_match_reference above checks that path against
    the reference's loss, logits and gradients); here it is compared with the general arithmetic (`no_beta_skip`): same loss and live
    gradients up to fp32 round-off, and the dead projections' gradients EXACTLY zero in both -- what autograd reports."""
    case = gu.model_case(name)
    inj, _ = inject_for(case)
    res = []
    for off in (0, 1):
        eng = make_engine(case)
        eng.set_option("no_beta_skip", off)
        out = eng.pretrain_step(dev_batch(case["batch"]), training=True, backward=True, nfr_inject=inj)
        torch.cuda.synchronize()
        res.append((out["loss"].item(), eng.grads.clone(), eng))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-6)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-4, atol=1e-8)
    dead = ("query", "key") if case["cfg"]["beta"] == 1.0 else ("ctx_attention",)
    for _, _, eng in res:
        for l in range(case["cfg"]["num_hidden_layers"]):
            for nm in dead:
                for wb in ("weight", "bias"):
                    assert float(eng.view(f"bert.encoder.layer.{l}.attention.self.{nm}.{wb}", grad=True).abs().max()) == 0.0, (l, nm, wb)
            live = eng.view(f"bert.encoder.layer.{l}.attention.self.value.weight", grad=True)
            assert float(live.abs().max()) > 0.0


@pytest.mark.parametrize("name", ["m1", "m1_pad", "m4"])
def test_clip_adamw_curve_matches_reference(name):
    case = gu.model_case(name)
    gold = case["gold"]
    eng = make_engine(case)
    batch = dev_batch(case["batch"])
    for s in range(len(gold["opt_losses"])):
        inj, _ = inject_for(case, "opt_", f"_{s}")
        out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj)
        if np.isnan(gold["opt_losses"][s]):          # reference quirk Q2: empty mask -> NaN loss
            assert torch.isnan(out["loss"]).item() and out["nfr_count"].item() == 0
            return
        eng.optimizer_step(lr=1e-3, weight_decay=1e-2, max_grad_norm=5.0)
        np.testing.assert_allclose(out["loss"].item(), gold["opt_losses"][s], rtol=2e-4)
        np.testing.assert_allclose(eng.grad_norm().item(), gold["opt_gradnorms"][s], rtol=1e-3)
    for k, v in eng.named_views().items():
        gu.check_stored(gold, "final/" + k, v.cpu().numpy(), 2e-3, 2e-5)


@pytest.mark.parametrize("name", ["m1", "m3"])
def test_bf16_mode_tracks_fp32(name):
    case = gu.model_case(name)
    gold = case["gold"]
    eng = make_engine(case, dtype="bf16")
    inj, _ = inject_for(case)
    out = eng.pretrain_step(dev_batch(case["batch"]), training=True, backward=True, nfr_inject=inj)
    np.testing.assert_allclose(out["loss"].item(), gold["train_loss"], rtol=2e-2)
    ref = make_engine(case, dtype="fp32")
    ref.pretrain_step(dev_batch(case["batch"]), training=True, backward=True, nfr_inject=inj)
    cos = torch.nn.functional.cosine_similarity(eng.grads, ref.grads, dim=0).item()
    assert cos > 0.99, cos


def test_device_nfr_masking_statistics_and_dropout_training():
    """Device-generated NFR masks (no injection) hit the reference ratios, never touch the target
    position or padding, and a dropout-on training step stays finite and decreases the loss."""
    case = gu.model_case("m3")
    eng = make_engine(case, dtype="fp32", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    batch = dev_batch(case["batch"])
    counts = []
    for _ in range(20):
        out = eng.pretrain_step(batch, training=True, backward=False)
        counts.append(out["nfr_count"].item())
        assert np.isfinite(out["loss"].item())
    valid = int((case["batch"][0]["node_ids"][:, 1:] != 0).sum())
    assert abs(np.mean(counts) / valid - 0.16) < 0.06
    losses = []
    for _ in range(12):
        out = eng.pretrain_step(batch, training=True, backward=True)
        eng.optimizer_step(lr=1e-3, weight_decay=1e-2, max_grad_norm=5.0)
        losses.append(out["loss"].item())
    assert np.all(np.isfinite(losses)) and np.mean(losses[-3:]) < np.mean(losses[:3])


@pytest.mark.parametrize("name,dtype", [("m1", "fp32"), ("m1_pad", "fp32"), ("m3", "fp32"), ("m4", "fp32"), ("m3", "bf16")])
def test_last_layer_shortcut_matches_full_path(name, dtype):
    """Training fast path (no last_hidden_state requested): the last layer's attn-out/FFN blocks run only on the
    rows the loss reads.  Losses and every gradient must equal the full path (and hence the reference)."""
    case = gu.model_case(name)
    gold = case["gold"]
    inj, _ = inject_for(case)
    batch = dev_batch(case["batch"])
    res = {}
    for label, want_hidden in (("full", True), ("fast", False)):
        eng = make_engine(case, dtype=dtype)
        out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=want_hidden)
        res[label] = (out["loss"].item(), out["logits"].cpu(), eng.grads.clone())
    tol = 1e-5 if dtype == "fp32" else 2e-2
    np.testing.assert_allclose(res["fast"][0], res["full"][0], rtol=tol)
    if dtype == "fp32":
        np.testing.assert_allclose(res["fast"][0], gold["train_loss"], rtol=1e-4)
    torch.testing.assert_close(res["fast"][1], res["full"][1], rtol=tol, atol=tol)
    g0, g1 = res["full"][2], res["fast"][2]
    rel = ((g0 - g1).norm() / g0.norm()).item()
    assert rel < (1e-5 if dtype == "fp32" else 3e-2), rel


def test_side_stream_reductions_give_identical_gradients():
    """Partial-sum reductions on the engine's second stream (opt-in; double-buffered slabs, fork/join by events) must
    be a pure scheduling change: losses and every gradient bit-identical to the single-stream run, repeated to catch
    races."""
    case = gu.model_case("m3")
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    ref = make_engine(case, dtype="bf16", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    assert not ref.get_option("side_stream_reduce")          # off by default
    out0 = ref.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
    torch.cuda.synchronize()
    eng = make_engine(case, dtype="bf16", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    eng.set_overlap(True)
    for _ in range(8):
        eng.rng_state[1] = 0
        out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
        torch.cuda.synchronize()
        assert out["loss"].item() == out0["loss"].item()
        assert torch.equal(eng.grads, ref.grads)


@pytest.mark.parametrize("name", ["m3"])
def test_fused_qkvc_attention_in_the_engine_matches_unfused(name):
    """bf16 engine at the headline shape (d=256, H=8, S=32): the fused projection+attention forward is on by
    default; switching it off must give the same loss and gradients up to bf16 round-off of the context."""
    case = gu.model_case(name)
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    res = {}
    for off in (0, 1):
        eng = make_engine(case, dtype="bf16", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        eng.set_option("no_fused_qkvc_attention", off)
        eng.profile_begin()
        out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
        prof = eng.profile_end()
        res[off] = (out["loss"].item(), eng.grads.clone(), prof)
    assert "fwd.qkvc_attention" in res[0][2] and "fwd.qkvc_attention" not in res[1][2]
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-3)
    rel = ((res[0][1] - res[1][1]).norm() / res[1][1].norm()).item()
    assert rel < 1e-2, rel


def test_phase_records_are_the_launch_order_of_the_step():
    """pmgt_profile_sequence / pmgt_profile_records (round 6): the phases in LAUNCH order -- what tools/make_traffic.py walks next to a counter
    run's dispatch order, and what bench.py takes a phase's median launch from.  The sequence is the forward's phases, then the losses, then
    the backward's in reverse layer order; records carry one non-negative time per entry; the totals of profile_end() are their sums; a second
    identical step records the identical sequence."""
    case = gu.model_case("m3")
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    eng = make_engine(case, dtype="bf16", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    seqs = []
    for _ in range(2):
        eng.profile_begin()
        eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
        seq = eng.profile_sequence()
        recs = eng.profile_records()
        prof = eng.profile_end()
        assert [n for n, _ in recs] == seq and all(t >= 0 for _, t in recs)
        for name, (cnt, ms) in prof.items():
            mine = [t for n, t in recs if n == name]
            assert len(mine) == cnt and abs(sum(mine) - ms) <= 1e-3 * max(ms, 1e-3), name
        seqs.append(seq)
    assert seqs[0] == seqs[1]
    seq = seqs[0]
    assert seq[0] == "mirror"
    fwd = [i for i, n in enumerate(seq) if n.startswith("fwd.")]
    loss = [i for i, n in enumerate(seq) if n.startswith("loss.")]
    bwd = [i for i, n in enumerate(seq) if n.startswith("bwd.")]
    assert max(fwd) < min(bwd) and min(loss) > min(fwd) and max(loss) < max(bwd)
    L = case["cfg"]["num_hidden_layers"]
    assert seq.count("fwd.qkvc_attention") == L and seq.count("bwd.attention_wgrad") + seq.count("bwd.attention") == L


@pytest.mark.parametrize("name,dtype", [("m1", "fp32"), ("m3", "fp32"), ("m3", "bf16")])
def test_table_projection_mode_matches_per_token_projection(name, dtype):
    """Small graphs: the engine projects the whole feature table once and gathers projected rows by node id
    (default whenever 2 (N+2) <= tokens).  The per-token path (large graphs) must give the same loss/gradients."""
    case = gu.model_case(name)
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    res = {}
    for off in (0, 1):
        eng = make_engine(case, dtype=dtype)
        eng.set_option("no_table_projection", off)
        eng.set_option("no_segment_sum", 1)           # isolate the forward: same per-token weight-gradient GEMM in both runs
        out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj)
        res[off] = (out["loss"].item(), out["last_hidden_state"].float().clone(), eng.grads.clone())
    if dtype == "fp32":
        np.testing.assert_allclose(res[0][0], case["gold"]["train_loss"], rtol=1e-4)
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])       # identical forward values
    rel = ((res[0][2] - res[1][2]).norm() / res[1][2].norm()).item()
    assert rel < 1e-6, rel


@pytest.mark.parametrize("name,dtype", [("m1", "fp32"), ("m1_pad", "fp32"), ("m3", "fp32"), ("m3", "bf16")])
def test_segment_sum_wgrad_matches_per_token_wgrad(name, dtype):
    """Table mode backward: the feature-projection weight gradient goes through a stable sort by node id + ordered
    segment sums + a GEMM over N+2 rows.  Must equal the per-token GEMM (and the reference), run to run identical."""
    case = gu.model_case(name)
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    res = {}
    for off in (0, 1, 0):
        eng = make_engine(case, dtype=dtype)
        eng.set_option("no_segment_sum", off)
        eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj)
        res.setdefault(off, []).append(eng.grads.clone())
    assert torch.equal(res[0][0], res[0][1])                      # deterministic
    for k in ("bert.embeddings.feat_linear.0.weight", "bert.embeddings.feat_linear.1.weight", "bert.embeddings.feat_linear.0.bias"):
        e = eng.entry(k)
        a = res[0][0][e["offset"]: e["offset"] + e["numel"]]
        b = res[1][0][e["offset"]: e["offset"] + e["numel"]]
        rel = ((a - b).norm() / b.norm()).item()
        # bf16: the two paths differ by one rounding of the mixed feature in the forward, so every gradient moves by
        # bf16 noise; against the fp32 engine both are equally close (1.2 % on these tensors, a one-off comparison script)
        assert rel < (2e-6 if dtype == "fp32" else 2e-2), (k, rel)
        if dtype == "fp32":
            gu.check_stored(case["gold"], "grad/" + k, a.view(*e["shape"]).cpu().numpy(), 2e-3, 2e-3 * float(b.abs().mean()) + 1e-9)
    # the modality-attention parameters are differentiated per node instead of per token in this mode: same sums,
    # different order
    for k in ("bert.embeddings.attention.1.weight", "bert.embeddings.attention.1.bias"):
        e = eng.entry(k)
        a = res[0][0][e["offset"]: e["offset"] + e["numel"]]
        b = res[1][0][e["offset"]: e["offset"] + e["numel"]]
        rel = ((a - b).norm() / b.norm()).item()
        assert rel < (1e-5 if dtype == "fp32" else 2e-2), (k, rel)
    other = torch.ones_like(res[0][0], dtype=torch.bool)
    for k in ("bert.embeddings.feat_linear.0.weight", "bert.embeddings.feat_linear.1.weight", "bert.embeddings.feat_linear.0.bias",
              "bert.embeddings.feat_linear.1.bias", "bert.embeddings.attention.1.weight", "bert.embeddings.attention.1.bias"):
        e = eng.entry(k)
        other[e["offset"]: e["offset"] + e["numel"]] = False
    # everything downstream sees the mixed feature through one extra rounding (bf16) / a different FMA contraction (fp32)
    rel = ((res[0][0][other] - res[1][0][other]).norm() / res[1][0][other].norm()).item()
    assert rel < (1e-5 if dtype == "fp32" else 2e-2), rel


def test_whole_step_replays_as_one_hip_graph():
    """The training step (NFR masks on device, forward, losses, backward, clip + AdamW) captured once into a hipGraph:
    replays must walk the same parameter trajectory as eager steps (device-side step counters keep advancing)."""
    from pmgt_amd.trainer import Trainer
    case = gu.model_case("m3")
    batch = dev_batch(case["batch"])

    def fresh():
        eng = make_engine(case, dtype="bf16", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        return eng, Trainer(eng, lr=1e-3, weight_decay=1e-2, max_grad_norm=5.0)

    eng_a, tr_a = fresh()
    la = [tr_a.train_step(batch).item() for _ in range(2 + 1 + 4)]          # warm-up + captured step + 4 replays
    torch.cuda.synchronize()
    eng_b, tr_b = fresh()
    replay = tr_b.capture_step(batch, warmup=2)                             # 2 eager steps, then the capture (records, does not run)
    lb = [replay().item() for _ in range(5)]
    torch.cuda.synchronize()
    # capture itself does not execute: eager trajectory steps 3..7 == replays 1..5
    np.testing.assert_allclose(lb, la[2:], rtol=0, atol=0)
    assert torch.equal(eng_a.params, eng_b.params)
    assert np.all(np.diff(lb) != 0)                                         # not one frozen step replayed


@pytest.mark.parametrize("name,dtype,bound", [("curve_c", "fp32", 1e-4), ("curve_c", "bf16", 1e-3),
                                              ("curve_c2", "fp32", 1e-4), ("curve_c2", "bf16", 1e-3)])
def test_loss_curve_matches_reference(name, dtype, bound):
    """north_star: "pre-training loss curve matching the reference within 1e-3".  G9 = optimizer steps of the reference (its
    sampler, model, clip, DenseSparseAdamW) on fresh batches: curve_c = 30 steps of a toy configuration, curve_c2 = 20 steps
    at the BENCHMARK shapes (7 252-node graph, L4 H8 d256 S32, B = 32, lr 1e-4: every fused kernel, table mode and the
    last-layer shortcut are on the path).  Here: C++ sampler (regenerates the same batches from the seed) -> HIP engine
    through the C ABI.  Per step: fp32 within 1e-4 relative (and gradient norms within 5e-3), bf16 within 1e-3."""
    c = gu.curve_case(name)
    gold = c["gold"]
    eng = make_engine(dict(cfg=c["cfg"], params=c["params"], tables=c["tables"]), dtype=dtype)
    losses, norms = [], []
    for step, batch in gu.curve_batches(c):
        ids = batch[0]["node_ids"]
        if step == 0:
            assert np.array_equal(ids.numpy(), gold["tgt_ids_0"])
        masked, m2, tidx = po.nfr_masking(ids, c["n_nodes"], torch.from_numpy(gold[f"r1_{step}"]),
                                          torch.from_numpy(gold[f"repl_{step}"]), torch.from_numpy(gold[f"r2_{step}"]))
        full = torch.full_like(ids, -1)
        full[:, 1:][m2] = tidx
        out = eng.pretrain_step(dev_batch(batch), training=True, backward=True, nfr_inject=(masked.cuda(), full.cuda()),
                                want_hidden=False)
        eng.optimizer_step(lr=c["lr"], weight_decay=1e-2, max_grad_norm=5.0)
        losses.append(out["loss"].item())
        norms.append(eng.grad_norm().item())
    rel = np.abs(np.array(losses) / gold["losses"] - 1.0)
    print(f"loss-curve max relative deviation ({name}, {dtype}): {rel.max():.2e}")
    assert rel.max() < bound, (rel.max(), losses, gold["losses"])
    if dtype == "fp32":
        np.testing.assert_allclose(norms, gold["gradnorms"], rtol=5e-3)


def test_head_major_qkvc_layout_is_transparent():
    """Training at the headline shape stores Q|K|V|C (and its gradient) head-major between the fused forward and the
    attention backward; the data-gradient GEMM reads a column-permuted W^T copy and the weight-gradient GEMM writes
    its rows back in q | k | v | c order.  Switching the layout off changes nothing but summation order."""
    case = gu.model_case("m3")
    batch = dev_batch(case["batch"])
    inj, _ = inject_for(case)
    res = {}
    for off in (0, 1):
        eng = make_engine(case, dtype="bf16", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        eng.set_option("no_head_major", off)
        out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=False)
        res[off] = (out["loss"].item(), eng.grads.clone())
    assert res[0][0] == res[1][0]                      # the forward is bit-identical
    # the data-gradient GEMM sums its K = 4d products in permuted order: bf16 round-off of dX, then propagated
    rel = ((res[0][1] - res[1][1]).norm() / res[1][1].norm()).item()
    assert rel < 5e-3, rel
    eng = make_engine(case, dtype="bf16")
    for k in ("bert.encoder.layer.3.attention.self.query.weight", "bert.encoder.layer.3.attention.self.ctx_attention.bias",
              "bert.encoder.layer.0.attention.self.value.weight"):
        e = eng.entry(k)
        a, b = res[0][1][e["offset"]: e["offset"] + e["numel"]], res[1][1][e["offset"]: e["offset"] + e["numel"]]
        assert ((a - b).norm() / b.norm()).item() < 1e-2, k      # rows land at the right q | k | v | c positions


# ---- C4 / C5 shapes (BASELINE.json configs 4 and 5: d = 512, H = 8 -> head size 64, S = 64) against the oracle ----------------
def _c4_case(B=4, n=3000, L=2, seed=11):
    """A graph large enough for the per-token (gather-fused) feature projection: 2 (N + 2) > 12 B S tokens."""
    from pmgt_amd.datasets import MODE_TRAIN, MCNSampler
    from pmgt_amd.graph import synthetic_graph
    S = 64
    cfg = po.default_cfg(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, hidden_size=512, num_attention_heads=8,
                         num_hidden_layers=L, intermediate_size=512, beta=0.5)
    graph = synthetic_graph(n, 6 * n, seed=seed)
    smp = MCNSampler(graph, max_ctx_neigh=S - 1)
    tgt, pair, num_pairs, labels = smp.batch(np.arange(2, 2 + B), MODE_TRAIN, threads=2, base_seed=seed, counter=0)
    batch = (tgt, pair, num_pairs, labels)
    g = torch.Generator().manual_seed(seed)
    ids = tgt["node_ids"]
    r1, r2 = torch.rand(B, S - 1, generator=g), torch.rand(B, S - 1, generator=g)
    repl = torch.randint(2, n + 2, (B * (S - 1),), generator=g)
    masked, m2, tidx = po.nfr_masking(ids, n, r1, repl, r2)
    full = torch.full_like(ids, -1)
    full[:, 1:][m2] = tidx
    return dict(cfg=cfg, params=po.synth_params(cfg, 5), tables=po.synth_tables(n, cfg["feat_hidden_sizes"], 6), batch=batch,
                n_nodes=n, inj=(masked.cuda(), full.cuda()), inj_cpu=(masked, m2, tidx))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp8"])
def test_c4_shapes_match_the_oracle(dtype):
    """d = 512 / head size 64 / S = 64 take other kernels than the headline shape: cooperative attention, K = 512 tile GEMMs,
    gather-fused feature projection and its gathered weight gradient (token mode), unfused fp8 Q|K|V|C projection."""
    case = _c4_case()
    eng = make_engine(case, dtype=dtype)
    tables, cfg = case["tables"], dict(case["cfg"])
    if dtype == "fp8":
        tables = [po.fake_quant_tensor(t)[0] for t in tables]
        cfg["fp8"] = True
    p = {k: v.clone().requires_grad_(True) for k, v in case["params"].items()}
    ref = po.pretrain_forward(p, cfg, tables, case["batch"], training=True, nfr_inject=case["inj_cpu"])
    ref["loss"].backward()
    out = eng.pretrain_step(dev_batch(case["batch"]), training=True, backward=True, nfr_inject=case["inj"])
    tol_loss = 1e-4 if dtype == "fp32" else 2e-2
    np.testing.assert_allclose(out["loss"].item(), ref["loss"].item(), rtol=tol_loss)
    np.testing.assert_allclose(out["gsr"].item(), ref["gsr"].item(), rtol=tol_loss)
    np.testing.assert_allclose(out["nfr"].item(), ref["nfr"].item(), rtol=tol_loss)
    np.testing.assert_allclose(out["logits"].cpu().numpy(), ref["logits"].detach().numpy(), rtol=0, atol=1e-4 if dtype == "fp32" else 3e-2)
    worst = (1.0, None)
    gscale = max(float(p[e["name"]].grad.abs().max()) for e in eng.entries)
    for e in eng.entries:
        a, b = eng.view(e["name"], grad=True).reshape(-1).cpu().double(), p[e["name"]].grad.reshape(-1).double()
        if float(b.abs().max()) < 1e-6 * gscale:      # key.bias: the softmax over keys is invariant to it, its gradient is round-off
            assert float(a.abs().max()) < 1e-4 * gscale, e["name"]
            continue
        if dtype == "fp32":
            scale = float(b.abs().max()) + 1e-12
            assert float((a - b).abs().max()) < 2e-3 * scale + 1e-9, e["name"]
        c = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
        worst = min(worst, (c, e["name"]))
    assert worst[0] > (0.9999 if dtype == "fp32" else 0.97), worst
    flat_a = torch.cat([eng.view(e["name"], grad=True).reshape(-1) for e in eng.entries]).cpu()
    flat_b = torch.cat([p[e["name"]].grad.reshape(-1) for e in eng.entries])
    assert torch.nn.functional.cosine_similarity(flat_a, flat_b, dim=0).item() > (0.99999 if dtype == "fp32" else 0.99)


# ---- sequence lengths 65 .. 100: the reference allows S <= max_position_embeddings = 100 (configuration_pmgt.py:23,
#      modeling_pmgt.py:189-193); the MFMA kernels stop at 64, above that the one-row-per-lane kernel runs 128-lane groups ----------
def _long_case(S, cfgkw, B=3, n=500, seed=41):
    from pmgt_amd.datasets import MODE_TRAIN, MCNSampler
    from pmgt_amd.graph import synthetic_graph
    cfg = po.default_cfg(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, beta=0.3, **cfgkw)
    graph = synthetic_graph(n, 20 * n, seed=seed)              # dense enough for > 64 distinct context neighbours ...
    smp = MCNSampler(graph, max_ctx_neigh=S - 1, hop_sampling_sizes=(32, 8, 4))
    batch = smp.batch(np.arange(2, 2 + B), MODE_TRAIN, threads=2, base_seed=seed, counter=0)
    batch[0]["attention_mask"][0, S - 7:] = 0                  # ... and a few padded keys on top (they must be masked out)
    batch[0]["node_ids"][0, S - 7:] = 0
    ids = batch[0]["node_ids"]
    g = torch.Generator().manual_seed(seed)
    r1, r2 = torch.rand(B, S - 1, generator=g), torch.rand(B, S - 1, generator=g)
    repl = torch.randint(2, n + 2, (B * (S - 1),), generator=g)
    masked, m2, tidx = po.nfr_masking(ids, n, r1, repl, r2)
    full = torch.full_like(ids, -1)
    full[:, 1:][m2] = tidx
    return dict(cfg=cfg, params=po.synth_params(cfg, seed), tables=po.synth_tables(n, cfg["feat_hidden_sizes"], seed + 1), batch=batch,
                n_nodes=n, inj=(masked.cuda(), full.cuda()), inj_cpu=(masked, m2, tidx))


@pytest.mark.parametrize("S,dtype,cfgkw", [
    (100, "fp32", dict(hidden_size=64, num_attention_heads=2, num_hidden_layers=2, intermediate_size=64)),      # head size 32
    (80, "fp32", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64)),       # head size 16
    (100, "bf16", dict(hidden_size=256, num_attention_heads=8, num_hidden_layers=2, intermediate_size=256)),    # headline model, S = 100
    (65, "bf16", dict(hidden_size=128, num_attention_heads=2, num_hidden_layers=1, intermediate_size=128)),     # head size 64, first length past the MFMA path
])
def test_sequence_lengths_up_to_max_position_embeddings(S, dtype, cfgkw):
    case = _long_case(S, cfgkw)
    assert int(case["batch"][1]["attention_mask"].sum(1).max()) == S          # full-length contexts are really in the batch
    eng = make_engine(case, dtype=dtype)
    p = {k: v.clone().requires_grad_(True) for k, v in case["params"].items()}
    ref = po.pretrain_forward(p, case["cfg"], case["tables"], case["batch"], training=True, nfr_inject=case["inj_cpu"])
    ref["loss"].backward()
    out = eng.pretrain_step(dev_batch(case["batch"]), training=True, backward=True, nfr_inject=case["inj"])
    tol = 1e-4 if dtype == "fp32" else 2e-2
    np.testing.assert_allclose(out["loss"].item(), ref["loss"].item(), rtol=tol)
    np.testing.assert_allclose(out["logits"].cpu().numpy(), ref["logits"].detach().numpy(), rtol=0, atol=1e-4 if dtype == "fp32" else 3e-2)
    np.testing.assert_allclose(out["last_hidden_state"].float().cpu().numpy(), ref["last_hidden_state"].detach().numpy(), rtol=0,
                               atol=2e-4 if dtype == "fp32" else 8e-2)
    flat_a = torch.cat([eng.view(e["name"], grad=True).reshape(-1) for e in eng.entries]).cpu().double()
    flat_b = torch.cat([p[e["name"]].grad.reshape(-1) for e in eng.entries]).double()
    if dtype == "fp32":
        assert float((flat_a - flat_b).abs().max()) < 2e-3 * float(flat_b.abs().max())
    assert torch.nn.functional.cosine_similarity(flat_a, flat_b, dim=0).item() > (0.99999 if dtype == "fp32" else 0.99)
    # inference entry with attention probabilities at S = 100 (rows sum to 1, padded keys get 0)
    tgt = case["batch"][0]
    last, hs, pr = eng.encode(ids=tgt["node_ids"].cuda(), attention_mask=tgt["attention_mask"].cuda(), output_attentions=True)
    assert pr.shape[-2:] == (S, S)
    np.testing.assert_allclose(pr.sum(-1).cpu().numpy(), 1.0, atol=1e-3 if dtype == "fp32" else 2e-2)
    assert float(pr[:, 0, :, :, S - 7:].abs().max()) == 0.0


def test_sequence_length_past_max_position_embeddings_is_refused():
    case = _long_case(16, dict(hidden_size=64, num_attention_heads=2, num_hidden_layers=1, intermediate_size=64))
    eng = make_engine(case, dtype="fp32")
    ids = torch.randint(2, 100, (2, 101)).cuda()
    with pytest.raises(Exception, match="max_position_embeddings"):
        eng.encode(ids=ids)


@pytest.mark.parametrize("mode", ["table", "token"])
@pytest.mark.parametrize("hidden,heads,feats", [(512, 8, [256, 128, 64]), (256, 8, [64, 32, 128, 96]), (512, 8, [96])])
def test_modality_counts_at_wider_hidden_sizes(hidden, heads, feats, mode):
    """Three, four and one modalities at hidden sizes 256 / 512 (two 256-column passes per row in the embedding kernels), per-node
    (table mode) and per-token projections, against autograd through the oracle on the batch and NFR draws of fixture m1."""
    base = gu.model_case("m1")
    cfg = po.default_cfg(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, hidden_size=hidden, num_attention_heads=heads,
                         num_hidden_layers=1, intermediate_size=hidden, beta=0.5, feat_hidden_sizes=feats)
    case = dict(base, cfg=cfg, params=po.synth_params(cfg, 41), tables=po.synth_tables(base["n_nodes"], feats, 42))
    inj, inj_cpu = inject_for(case)
    p = {k: v.clone().requires_grad_(True) for k, v in case["params"].items()}
    ref = po.pretrain_forward(p, cfg, case["tables"], case["batch"], training=True, nfr_inject=inj_cpu)
    ref["loss"].backward()
    grms = max(float(v.grad.double().pow(2).mean().sqrt()) for v in p.values())
    for dtype in ("fp32", "bf16"):
        eng = make_engine(case, dtype=dtype)
        if mode == "token":
            eng.set_option("no_table_projection", True)
        out = eng.pretrain_step(dev_batch(case["batch"]), training=True, backward=True, nfr_inject=inj)
        np.testing.assert_allclose(out["loss"].item(), ref["loss"].item(), rtol=1e-4 if dtype == "fp32" else 2e-2)
        np.testing.assert_allclose(out["nfr"].item(), ref["nfr"].item(), rtol=1e-4 if dtype == "fp32" else 2e-2)
        for k, g in eng.named_views(grad=True).items():
            a, b = g.float().cpu(), p[k].grad
            if dtype == "fp32":
                scale = float(b.double().pow(2).mean().sqrt())
                # (key.bias gradients are exactly 0 in theory -- softmax shift invariance: absolute floor from the global scale)
                np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-3, atol=2e-3 * scale + 1e-5 * grms, err_msg=k)
            elif k.startswith("bert.embeddings.") and (k.endswith("weight") or "feat_linear" in k):
                if float(b.abs().max()) == 0.0:          # one modality: the softmax over a single score has no gradient
                    assert float(a.abs().max()) <= 1e-6 * grms, k
                    continue
                cos = torch.nn.functional.cosine_similarity(a.reshape(-1), b.reshape(-1), dim=0).item()
                assert cos > 0.99, (k, cos)
