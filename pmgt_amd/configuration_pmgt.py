"""PMGTConfig — same fields, defaults and keyword surface as the reference's
`pmgt/pmgt/configuration_pmgt.py:9-41`, as a plain attribute bag (no `transformers` dependency:
the installed transformers 5.x no longer matches the 4.11.2 API the reference was written for)."""
import copy
from typing import Any, Dict


class PMGTConfig:
    model_type = "pmgt"

    def __init__(
        self,
        hidden_size=128,
        feat_hidden_sizes=(1536, 768),
        num_hidden_layers=5,
        num_attention_heads=1,
        intermediate_size=128,
        hidden_act="gelu",
        hidden_dropout_prob=0.1,
        attention_probs_dropout_prob=0.1,
        max_position_embeddings=100,
        initializer_range=0.02,
        layer_norm_eps=1e-12,
        beta=0.5,  # diversity promoting attention weight
        **kwargs,
    ):
        # PretrainedConfig kwargs the hot path reads (transformers 4.11.2 defaults)
        self.output_attentions = kwargs.pop("output_attentions", False)
        self.output_hidden_states = kwargs.pop("output_hidden_states", False)
        self.return_dict = kwargs.pop("return_dict", True)
        self.chunk_size_feed_forward = kwargs.pop("chunk_size_feed_forward", 0)
        self.position_embedding_type = kwargs.pop("position_embedding_type", "absolute")
        for k, v in kwargs.items():
            setattr(self, k, v)
        self.hidden_size = hidden_size
        self.feat_hidden_sizes = list(feat_hidden_sizes)
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.hidden_act = hidden_act
        self.intermediate_size = intermediate_size
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.initializer_range = initializer_range
        self.layer_norm_eps = layer_norm_eps
        self.beta = beta
        if hidden_act != "gelu":
            raise ValueError("the HIP path implements hidden_act='gelu' (exact erf form), the reference default")
        if self.position_embedding_type != "absolute":
            raise ValueError("only position_embedding_type='absolute' is implemented (the reference default)")

    @property
    def use_return_dict(self) -> bool:
        return self.return_dict

    def to_dict(self) -> Dict[str, Any]:
        d = copy.deepcopy(self.__dict__)
        d["model_type"] = self.model_type
        return d

    def __repr__(self):
        return f"PMGTConfig {self.to_dict()}"
