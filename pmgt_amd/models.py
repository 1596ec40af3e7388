"""Initialisation and workload helpers for the PMGT engine.

`reference_init` reproduces the reference's initial parameter DISTRIBUTIONS:
everything under PMGTModel gets PMGTPretrainedModel._init_weights (pmgt/pmgt/modeling_pmgt.py:44-58:
Linear/Embedding weights N(0, initializer_range), biases 0, LayerNorm (1, 0)); PMGT itself never calls
it, so the NFR projections keep torch's nn.Linear default (kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in))
for weight and bias) — fixture tests/golden/init_stats.npz pins these statistics.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def init_value(name: str, shape, config, generator: torch.Generator) -> torch.Tensor:
    """The reference's initial distribution of ONE parameter (pmgt/pmgt/modeling_pmgt.py:44-58 under `bert.`;
    nn.Linear's default under `nfr_loss.`, pmgt/pmgt/models.py:31-54) -- pure torch, no device."""
    if name.startswith("nfr_loss."):
        bound = 1.0 / math.sqrt(config.hidden_size)            # kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)), bias likewise
        return (torch.rand(shape, generator=generator) * 2 - 1) * bound
    if name.endswith("LayerNorm.weight"):
        return torch.ones(shape)
    if name.endswith(".bias"):
        return torch.zeros(shape)
    return torch.randn(shape, generator=generator) * config.initializer_range


def reference_init(engine, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    for e in engine.entries:
        engine.view(e["name"]).copy_(init_value(e["name"], e["shape"], engine.config, g))


def synthetic_features(n_nodes: int, feat_sizes=(1536, 768), seed: int = 0):
    """fp32 [N+2, F] ~ N(0,1) with rows 0 (<pad>) and 1 (<mask>) zero (notebooks/PMGT.ipynb cell 30)."""
    rs = np.random.RandomState(seed)
    out = []
    for f in feat_sizes:
        a = rs.standard_normal((n_nodes + 2, f)).astype(np.float32)
        a[:2] = 0
        out.append(a)
    return out


def train_flops_per_node(d, I, L, S, Fv=1536, Ft=768, pairs=10):
    """Algorithmic FLOPs of one pre-training step per target node:
    forward per token 2*F*d + 8d + L(10 d^2 + 4 d I + 6 S d); training = 3x encoder terms + 2x the
    feature projection (frozen tables: no dgrad) + the NFR head (3 * M * 2 d F, M ~ 0.16 (S-1))."""
    F = Fv + Ft
    seqs = pairs + 2
    tok = seqs * S
    enc = L * (10 * d * d + 4 * d * I + 6 * S * d) + 8 * d
    proj = 2 * F * d
    nfr = 3 * 0.16 * (S - 1) * 2 * d * F
    return tok * (3 * enc + 2 * proj) + nfr


def encoder_flops_per_node(d, I, L, S, pairs=10):
    """The encoder's share of train_flops_per_node (SURVEY.md section 8d "encoder 4.45" at C2): 3 x the forward flops of the L layers and
    the embedding mix over the 12 S tokens of a target -- no feature projection, no NFR head.  The denominator of north_star's
    ">= 30 % bf16 MFMA utilisation on the PMGT encoder"."""
    return (pairs + 2) * S * 3 * (L * (10 * d * d + 4 * d * I + 6 * S * d) + 8 * d)


def executed_flops_per_node(d, I, L, S, n_nodes, batch, Fv=1536, Ft=768, pairs=10, shortcut=True, dead_dot_branch=False):
    """FLOPs the engine actually EXECUTES per target node for the same step (same conventions as train_flops_per_node), with its
    two structural savings taken out: table mode (2 (N + 2) <= tokens: the feature projection and its weight gradient run over
    N + 2 table rows per step instead of one row per token) and the last-layer shortcut (the last layer's attention-output and
    FFN blocks, forward and backward, run on the B target-CLS + pairs * B pair-CLS + masked rows only).  `dead_dot_branch`: beta == 1
    with the dead dot-product branch skipped (V | C projected only, one softmax branch)."""
    F = Fv + Ft
    seqs = pairs + 2
    tok = seqs * S
    masked = 0.16 * (S - 1)
    per_tok_attn = 8 * d * d + 6 * S * d                     # Q|K|V|C projection + the two score / context products
    if dead_dot_branch:
        per_tok_attn = 4 * d * d + 4 * S * d                 # V | C projection; C C^T scores + P V (the dot-product scores are never formed)
    per_tok_dense = 2 * d * d + 4 * d * I                    # attention output + FFN
    full_layers = L - 1 if shortcut else L
    enc = tok * (full_layers * (per_tok_attn + per_tok_dense) + 8 * d)
    if shortcut:
        enc += tok * per_tok_attn + (1 + pairs + masked) * per_tok_dense
    table_mode = 2 * (n_nodes + 2) <= batch * tok
    proj_rows = (n_nodes + 2) / batch if table_mode else tok
    nfr = 3 * masked * 2 * d * F
    return 3 * enc + 2 * proj_rows * 2 * F * d + nfr


# ================================================================================================
# PMGT: the reference's pre-training module surface (pmgt/pmgt/models.py:22-176)
# ================================================================================================
import torch.nn as nn  # noqa: E402

from .configuration_pmgt import PMGTConfig  # noqa: E402
from .engine import Engine  # noqa: E402
from .modeling_pmgt import PMGTForPreTrainingOutput, PMGTModel, PMGTPretrainedModel, _attach_params  # noqa: E402


class _PretrainLoss(torch.autograd.Function):
    """Connects the engine's fused forward+backward to autograd: the step already left d loss / d params
    in a flat scratch buffer; backward() hands the per-parameter views (scaled by the incoming gradient)
    to autograd so `.grad` accumulation, `loss / accum` scaling and optimizers behave as usual."""

    @staticmethod
    def forward(ctx, loss, scratch, names, engine, *params):
        ctx.scratch, ctx.names, ctx.engine = scratch, names, engine
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        ctx.scratch.mul_(grad_out)
        eng = ctx.engine
        views = []
        for n in ctx.names:
            e = eng.entry(n)
            views.append(ctx.scratch[e["offset"]: e["offset"] + e["numel"]].view(*e["shape"]).clone())
        return (None, None, None, None, *views)


class PMGT(PMGTPretrainedModel):
    """Same constructor / forward as the reference's `PMGT`.  One call runs all B(1 + pairs + 1) sequences
    through the HIP engine as a single batched encoder pass; in training mode the same call also runs
    the backward (device work is fused; autograd only distributes the resulting gradients)."""

    def __init__(self, node_size: int, random_node_ratio: float = 0.2 * 0.1, mask_node_ratio: float = 0.2 * 0.8,
                 config: PMGTConfig = None, feat_init_emb=None, dtype: str = "bf16", device: str = "cuda:0", seed: int = 0):
        super().__init__()
        config = config if config is not None else PMGTConfig()
        self.node_size = node_size
        self.random_node_ratio = random_node_ratio
        self.mask_node_ratio = mask_node_ratio
        self.config = config
        self.engine = Engine(config, dtype=dtype, device=device, seed=seed)
        self.bert = PMGTModel(config, engine=self.engine)
        _attach_params(self, self.engine, "nfr_loss.")          # creates self.projections.{0,1}.{weight,bias}
        self._reroot("projections", "nfr_loss")                 # -> nfr_loss.projections.* like the reference
        self.feat_embeddings = nn.Module()
        for i, f in enumerate(config.feat_hidden_sizes):
            holder = nn.Module()
            holder.register_parameter("weight", nn.Parameter(torch.zeros(node_size + 2, f), requires_grad=False))
            self.feat_embeddings.add_module(str(i), holder)
        self.bert._init_weights()                               # PMGTModel.__init__ -> init_weights() (modeling_pmgt.py:74)
        self._init_nfr_default()
        if feat_init_emb is not None:
            assert len(feat_init_emb) == len(config.feat_hidden_sizes)
            self.set_features(feat_init_emb)

    def _reroot(self, child: str, under: str):
        sub = self._modules.pop(child)
        holder = nn.Module()
        holder.add_module(child, sub)
        self.add_module(under, holder)

    def _init_nfr_default(self):
        """PMGT never runs _init_weights on itself: the NFR projections keep nn.Linear's default init
        (pmgt/pmgt/models.py:31-54)."""
        with torch.no_grad():
            for n, p in self.nfr_loss.named_parameters():
                bound = 1.0 / math.sqrt(self.config.hidden_size)
                p.uniform_(-bound, bound)

    def set_features(self, feat_init_emb):
        with torch.no_grad():
            for i, w in enumerate(feat_init_emb):
                self.feat_embeddings._modules[str(i)].weight.copy_(torch.as_tensor(w))
        self.engine.set_tables(*[emb.weight for emb in self.feat_embeddings.children()])

    def load_state_dict(self, state_dict, strict: bool = True):
        out = super().load_state_dict(state_dict, strict=strict)      # writes the engine's parameters through the Parameter views
        self.engine.set_tables(*[emb.weight for emb in self.feat_embeddings.children()])
        self.engine.check_layernorm_carrier()
        return out

    def forward(self, target_node_inputs, pair_node_inputs=None, num_pairs=None, labels=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, nfr_inject=None):
        if pair_node_inputs is not None:
            assert labels is not None, "labels must be passed, when set pair_node_inputs"
            assert num_pairs is not None, "num_pairs must be passed, when set pair_node_inputs"
        cfg = self.config
        output_attentions = output_attentions if output_attentions is not None else cfg.output_attentions
        output_hidden_states = output_hidden_states if output_hidden_states is not None else cfg.output_hidden_states
        return_dict = return_dict if return_dict is not None else cfg.use_return_dict
        eng = self.engine
        dev = eng.device
        to = lambda d: {k: v.to(dev) for k, v in d.items()}
        tgt = to(target_node_inputs)
        hidden = attn = None
        if output_attentions or output_hidden_states or pair_node_inputs is None:
            enc = self.bert.encode_ids(tgt["node_ids"], tgt["attention_mask"], output_attentions, output_hidden_states)
            hidden, attn = enc.hidden_states, enc.attentions
            if pair_node_inputs is None:        # inference: loss is None, so out[0] is last_hidden_state
                if not return_dict:
                    return (None, None) + enc.to_tuple()
                return PMGTForPreTrainingOutput(last_hidden_state=enc.last_hidden_state, hidden_states=hidden, attentions=attn)
        batch = (tgt, to(pair_node_inputs), num_pairs.to(dev), labels.to(dev))
        want_grad = self.training and torch.is_grad_enabled()
        scratch = torch.empty_like(eng.params) if want_grad else None
        out = eng.pretrain_step(batch, training=self.training, backward=want_grad, nfr_inject=nfr_inject,
                                random_node_ratio=self.random_node_ratio, mask_node_ratio=self.mask_node_ratio,
                                grad_buffer=scratch)
        # the engine hands out its outputs from a small ring of persistent buffers: this module surface returns copies, so a
        # caller may keep the results of many steps (e.g. a validation epoch's list of outputs) like the reference's
        loss, logits = out["loss"].clone(), out["logits"].clone()
        if want_grad:
            names = [n for n, p in self.named_parameters() if p.requires_grad]
            params = [p for _, p in self.named_parameters() if p.requires_grad]
            loss = _PretrainLoss.apply(loss, scratch, [self._engine_name(n) for n in names], eng, *params)
        last = out["last_hidden_state"].to(torch.float32, copy=True)      # a copy in fp32 mode too (.float() would alias the ring buffer)
        if not return_dict:
            return (loss, logits, last, None) + tuple(v for v in (hidden, attn) if v is not None)
        return PMGTForPreTrainingOutput(loss=loss, prediction_logits=logits, last_hidden_state=last,
                                        pooler_output=None, hidden_states=hidden, attentions=attn)

    @staticmethod
    def _engine_name(module_name: str) -> str:
        return module_name            # module tree == engine entry names (bert.*, nfr_loss.*)
