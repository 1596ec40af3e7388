"""`PMGTModel` and the output containers with the reference's Python surface
(pmgt/pmgt/modeling_pmgt.py:65-152,572-579), backed by the HIP engine instead of torch.nn ops.

Parameters keep the reference's state_dict names (`embeddings.feat_linear.0.weight`,
`encoder.layer.3.attention.self.ctx_attention.bias`, ...): every nn.Parameter is a VIEW into the
engine's flat fp32 buffer, so checkpoints load/save with the reference keys while the kernels see one
contiguous buffer.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .configuration_pmgt import PMGTConfig
from .engine import Engine


class ModelOutput(OrderedDict):
    """Minimal restatement of transformers' ModelOutput: attribute + key access, and integer indexing /
    `to_tuple()` over the fields that are not None (so `out[0]` is `loss` when there is one and
    `last_hidden_state` otherwise — the behaviour pmgt/pmgt/trainer.py:154,157,166-167 relies on)."""

    _fields: Tuple[str, ...] = ()

    def __init__(self, **kwargs):
        super().__init__()
        for f in self._fields:
            v = kwargs.get(f)
            object.__setattr__(self, f, v)
            if v is not None:
                super().__setitem__(f, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return super().__getitem__(k)
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


class BaseModelOutputWithPooling(ModelOutput):
    _fields = ("last_hidden_state", "pooler_output", "hidden_states", "attentions")


class PMGTForPreTrainingOutput(ModelOutput):
    """pmgt/pmgt/modeling_pmgt.py:572-579."""
    _fields = ("loss", "prediction_logits", "last_hidden_state", "pooler_output", "hidden_states", "attentions")


def _attach_params(root: nn.Module, engine: Engine, prefix: str):
    """Create the nested module tree + Parameter views for every engine entry under `prefix`."""
    for e in engine.entries:
        name = e["name"]
        if not name.startswith(prefix):
            continue
        parts = name[len(prefix):].split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        param = nn.Parameter(engine.view(name), requires_grad=True)
        mod.register_parameter(parts[-1], param)


class PMGTPretrainedModel(nn.Module):
    """Init rule of the reference (pmgt/pmgt/modeling_pmgt.py:44-58): Linear/Embedding weights
    N(0, initializer_range), biases 0, LayerNorm (1, 0)."""
    config_class = PMGTConfig
    base_model_prefix = "pmgt"

    def _init_weights(self):
        std = self.config.initializer_range
        with torch.no_grad():
            for n, p in self.named_parameters():
                if n.endswith("LayerNorm.weight"):
                    p.fill_(1.0)
                elif n.endswith(".bias"):
                    p.zero_()
                else:
                    p.normal_(mean=0.0, std=std)


class _EncoderFn(torch.autograd.Function):
    """Autograd bridge for callers that put their own head on the encoder output (PMGT_NCF,
    pmgt/pmgt_ncf/models.py:86-89): forward keeps the activations on the device, backward turns
    d loss / d last_hidden_state into the `bert.*` parameter gradients with the HIP backward kernels."""

    @staticmethod
    def forward(ctx, engine, ids, feats, mask, training, names, *params):
        last, state = engine.encode_train(ids=ids, feats=feats, attention_mask=mask, training=training)
        ctx.engine, ctx.state, ctx.names = engine, state, names
        return last.float()

    @staticmethod
    def backward(ctx, d_last):
        eng = ctx.engine
        scratch = torch.empty_like(eng.params)
        eng.encode_backward(ctx.state, d_last, grad_buffer=scratch)
        ctx.state = None
        views = []
        for n in ctx.names:
            e = eng.entry(n)
            views.append(scratch[e["offset"]: e["offset"] + e["numel"]].view(*e["shape"]).clone())
        return (None, None, None, None, None, None, *views)


class PMGTModel(PMGTPretrainedModel):
    """Encoder with the reference call signature.  `forward(*input_feat_embeds, attention_mask=...)` takes
    already-gathered features [T, S, F_m] (the compatible, materialised-input entry); the fused
    gather + projection path is entered through `PMGT.forward` / `encode_ids`, which see node ids."""

    def __init__(self, config: PMGTConfig, add_pooling_layer: bool = False, engine: Optional[Engine] = None,
                 dtype: str = "bf16", device: str = "cuda:0"):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError(
                f"The hidden size ({config.hidden_size}) is not a multiple of the number of attention "
                f"heads ({config.num_attention_heads})")
        if add_pooling_layer:
            raise NotImplementedError("the reference never enables the pooler on this path (modeling_pmgt.py:66,72)")
        self.config = config
        self.engine = engine if engine is not None else Engine(config, dtype=dtype, device=device)
        _attach_params(self, self.engine, "bert.")      # engine names carry PMGT's `bert.` prefix
        emb = self._modules["embeddings"]
        emb.register_buffer("position_ids", torch.arange(config.max_position_embeddings).unsqueeze(0))
        emb.register_buffer("role_ids", torch.LongTensor([0] + [1] * (config.max_position_embeddings - 1)).unsqueeze(0))
        self.pooler = None
        if engine is None:
            self._init_weights()

    def load_state_dict(self, state_dict, strict: bool = True):
        out = super().load_state_dict(state_dict, strict=strict)      # writes the engine's parameters through the Parameter views
        self.engine.check_layernorm_carrier()
        return out

    def forward(self, *input_feat_embeds, attention_mask=None, head_mask=None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        first = input_feat_embeds[0]
        assert all(first.size()[:-1] == f.size()[:-1] for f in input_feat_embeds[1:]), \
            "All features are same dim except last one"
        assert head_mask is None, "head masks are not supported by the HIP path (the reference passes None)"
        cfg = self.config
        output_attentions = output_attentions if output_attentions is not None else cfg.output_attentions
        output_hidden_states = output_hidden_states if output_hidden_states is not None else cfg.output_hidden_states
        return_dict = return_dict if return_dict is not None else cfg.use_return_dict
        if self._wants_grad() and not (output_attentions or output_hidden_states):
            last = self._with_grad(None, list(input_feat_embeds), attention_mask)
            return self._wrap(last, None, None, return_dict)
        last, hs, pr = self.engine.encode(feats=list(input_feat_embeds), attention_mask=attention_mask,
                                          output_hidden_states=output_hidden_states, output_attentions=output_attentions)
        return self._wrap(last, hs, pr, return_dict)

    def encode_ids(self, node_ids, attention_mask=None, output_attentions=False, output_hidden_states=False,
                   return_dict=True):
        """Same as forward() but on node ids: the feature gather is fused into the projection GEMM."""
        if self._wants_grad() and not (output_attentions or output_hidden_states):
            last = self._with_grad(node_ids, None, attention_mask)
            return self._wrap(last, None, None, return_dict)
        last, hs, pr = self.engine.encode(ids=node_ids, attention_mask=attention_mask,
                                          output_hidden_states=output_hidden_states, output_attentions=output_attentions)
        return self._wrap(last, hs, pr, return_dict)

    def _wants_grad(self) -> bool:
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _with_grad(self, ids, feats, mask):
        named = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
        names = ["bert." + n for n, _ in named]
        return _EncoderFn.apply(self.engine, ids, feats, mask, self.training, names, *[p for _, p in named])

    @staticmethod
    def _wrap(last, hs, pr, return_dict):
        last = last.float()
        hidden = tuple(h.float() for h in hs) if hs is not None else None
        attn = tuple(p for p in pr) if pr is not None else None
        if not return_dict:      # (sequence_output, pooled_output) + encoder_outputs[1:]  (modeling_pmgt.py:144-145)
            return (last, None) + tuple(v for v in (hidden, attn) if v is not None)
        return BaseModelOutputWithPooling(last_hidden_state=last, pooler_output=None, hidden_states=hidden, attentions=attn)
