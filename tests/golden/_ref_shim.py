"""Compatibility shim that lets the read-only reference (transformers==4.11.2 era) import
against the transformers 5.x installed in the development container.

Tool code for `make_golden.py` only: it runs in the dev container where /root/reference exists,
never on the GPU box, and is never imported by the product (`pmgt_amd`), the tests or the bench.
It restates the two 4.11.2 mixin helpers the reference relies on
(`get_extended_attention_mask` -> (1-m)*-10000, `get_head_mask(None)` -> [None]*L) and re-exports
helpers that moved between transformers releases.
"""
import sys

REFERENCE_ROOT = "/root/reference"


def install():
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer

    def _removed(*a, **k):  # head pruning is off the hot path
        raise NotImplementedError

    mu.find_pruneable_heads_and_indices = _removed
    _orig = mu.PreTrainedModel.init_weights

    def _init_weights(self):
        return self.post_init() if not hasattr(self, "all_tied_weights_keys") else _orig(self)

    mu.PreTrainedModel.init_weights = _init_weights

    def _ext_mask(self, attention_mask, input_shape, device=None):
        return (1.0 - attention_mask[:, None, None, :].to(self.dtype)) * -10000.0

    def _head_mask(self, head_mask, num_hidden_layers, is_attention_chunked=False):
        assert head_mask is None
        return [None] * num_hidden_layers

    mu.PreTrainedModel.get_extended_attention_mask = _ext_mask
    mu.PreTrainedModel.get_head_mask = _head_mask
