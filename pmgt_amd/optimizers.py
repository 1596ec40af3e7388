"""`DenseSparseAdamW` with the reference's constructor (pmgt/optimizers.py:127-170) and
`get_optimizer` (pmgt/base_trainer.py:35-68), executed as ONE fused clip + AdamW launch over the
engine's flat buffers (dense branch, pmgt/optimizers.py:256-270; PMGT has no sparse gradients)."""
from __future__ import annotations

import torch
from torch.optim import Optimizer


class DenseSparseAdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        if not 0.0 <= weight_decay:
            raise ValueError("Invalid weight_decay value: {}".format(weight_decay))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm     # PL's gradient_clip_val (pmgt/base_trainer.py:314), fused into the step
        self.engine = None
        self._decay_synced = False

    def bind(self, engine):
        """Attach the engine whose flat buffers hold these parameters."""
        self.engine = engine
        return self

    def _sync(self):
        eng = self.engine
        lo, hi = eng.params.data_ptr(), eng.params.data_ptr() + eng.params.numel() * 4
        mask = torch.zeros(eng.n_params, dtype=torch.uint8)
        lrs, wds = set(), set()
        for g in self.param_groups:
            lrs.add(g["lr"])
            if g["weight_decay"] > 0:
                wds.add(g["weight_decay"])
            for p in g["params"]:
                assert lo <= p.data_ptr() < hi, "parameter does not live in the bound engine's flat buffer"
                off = (p.data_ptr() - lo) // 4
                if g["weight_decay"] > 0:
                    mask[off: off + p.numel()] = 1
        assert len(lrs) == 1 and len(wds) <= 1, "the fused step supports one lr and one non-zero weight decay"
        eng.decay_mask.copy_(mask.to(eng.device))
        self._decay_synced = True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        assert self.engine is not None, "call .bind(engine) (get_optimizer does it)"
        if not self._decay_synced:
            self._sync()
        eng = self.engine
        # gather .grad of the parameter views into the flat gradient buffer when autograd produced them
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    off = (p.data_ptr() - eng.params.data_ptr()) // 4
                    dst = eng.grads[off: off + p.numel()].view_as(p)
                    if p.grad.data_ptr() != dst.data_ptr():
                        dst.copy_(p.grad)
        g0 = self.param_groups[0]
        wd = max(g["weight_decay"] for g in self.param_groups)
        eng.optimizer_step(lr=g0["lr"], weight_decay=wd, betas=g0["betas"], eps=g0["eps"], max_grad_norm=self.max_grad_norm)
        return loss


def get_optimizer(args) -> Optimizer:
    """pmgt/base_trainer.py:35-68: two groups, no weight decay on names containing 'bias' / 'LayerNorm.weight'.
    `args` needs .model (a pmgt_amd PMGT), .decay, .lr, .optim and optionally .gradient_max_norm."""
    model = args.model
    no_decay = ["bias", "LayerNorm.weight"]
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    param_groups = [
        {"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": args.decay, "lr": args.lr},
        {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0, "lr": args.lr},
    ]
    if args.optim == "adamw":
        return DenseSparseAdamW(param_groups, max_grad_norm=getattr(args, "gradient_max_norm", None)).bind(model.engine)
    if args.optim == "sgd":
        return torch.optim.SGD(param_groups)
    raise ValueError(f"Optimizer {args.optim} is not supported")
