"""PMGT_NCF — the reference's downstream fine-tuning head (pmgt/pmgt_ncf/models.py:15-121), kept as the second
caller of the encoder boundary: the item tower is `PMGTModel` on the HIP engine (node-id entry, feature gather fused,
encoder gradients flow through `pmgt_encode_train` / `pmgt_encode_backward`, feature tables frozen); the NCF
part (user embeddings, MLP stack, optional GMF branch, predict layer) is a handful of tiny dense ops outside the
hot path and stays in torch."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from .configuration_pmgt import PMGTConfig
from .modeling_pmgt import PMGTModel, PMGTPretrainedModel


class MLPLayer(nn.Module):
    """Linear -> Dropout -> ReLU (pmgt/pmgt_ncf/models.py:108-121)."""

    def __init__(self, input_hidden_size: int, output_hidden_size: int, dropout: float = 0.0):
        super().__init__()
        self.linear = nn.Linear(input_hidden_size, output_hidden_size)
        self.dropout = nn.Dropout(dropout)
        self.act = nn.ReLU()

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return self.act(self.dropout(self.linear(hidden_states)))


class PMGT_NCF(PMGTPretrainedModel):
    def __init__(self, user_num: int, item_num: int, factor_num: int = 32, num_layers: int = 3, emb_dropout: float = 0.0,
                 dropout: float = 0.0, model: str = "MLP", config: PMGTConfig = None, dtype: str = "bf16",
                 device: str = "cuda:0"):
        super().__init__()
        assert model in ["MLP", "NeuMF-end"]
        config = config if config is not None else PMGTConfig()
        assert config.hidden_size == factor_num * 2 ** (num_layers - 1), \
            "item embedding (hidden_size) must match the user embedding width (pmgt/pmgt_ncf/models.py:49-63)"
        self.config = config
        self.factor_num, self.num_layers, self.model = factor_num, num_layers, model
        self.bert = PMGTModel(config, dtype=dtype, device=device)
        self.engine = self.bert.engine
        self.feat_embeddings = nn.Module()          # frozen, idx 0 <pad>, idx 1 <mask> (:36-47)
        for i, f in enumerate(config.feat_hidden_sizes):
            holder = nn.Module()
            holder.register_parameter("weight", nn.Parameter(torch.zeros(item_num + 2, f), requires_grad=False))
            self.feat_embeddings.add_module(str(i), holder)
        self.mlp_user_embeddings = nn.Embedding(user_num, factor_num * (2 ** (num_layers - 1)))
        self.emb_dropout = nn.Dropout(emb_dropout)
        self.mlp_layers = nn.Sequential(*[MLPLayer(self._get_input_size(i), self._get_input_size(i) // 2, dropout=dropout)
                                          for i in range(num_layers)])
        if model == "NeuMF-end":
            self.gmf_user_embeddings = nn.Embedding(user_num, factor_num)
            self.gmf_item_embeddings = nn.Embedding(item_num, factor_num)
            self.predict_layer = nn.Linear(factor_num * 2, 1)
        else:
            self.register_parameter("gmf_user_embeddings", None)
            self.register_parameter("gmf_item_embeddings", None)
            self.predict_layer = nn.Linear(factor_num, 1)
        dev = self.engine.device
        for m in (self.mlp_user_embeddings, self.mlp_layers, self.predict_layer):
            m.to(dev)
        if model == "NeuMF-end":
            self.gmf_user_embeddings.to(dev)
            self.gmf_item_embeddings.to(dev)

    def _get_input_size(self, i: int) -> int:
        return self.factor_num * 2 ** (self.num_layers - i)

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

    def forward(self, user: torch.LongTensor, item: Dict[str, torch.Tensor]) -> torch.Tensor:
        dev = self.engine.device
        user = user.to(dev)
        node_ids = item["node_ids"].to(dev)
        mlp_user_embeds = self.mlp_user_embeddings(user)
        item_embeds = self.bert.encode_ids(node_ids, attention_mask=item["attention_mask"].to(dev))[0][:, 0]
        interaction = self.emb_dropout(torch.cat([mlp_user_embeds, item_embeds], dim=-1))
        mlp_outputs = self.mlp_layers(interaction)
        if self.model == "NeuMF-end":
            gmf_outputs = self.gmf_user_embeddings(user) * self.gmf_item_embeddings(node_ids[:, 0] - 2)
            outputs = torch.cat([self.emb_dropout(gmf_outputs), mlp_outputs], dim=-1)
        else:
            outputs = mlp_outputs
        return self.predict_layer(outputs).view(-1)
