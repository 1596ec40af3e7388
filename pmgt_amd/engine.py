"""Device-side state of one PMGT replica and thin wrappers over the C ABI (include/pmgt_capi.h, include/pmgt_ops.h).

PyTorch is used here only as plumbing: device memory (torch tensors own every buffer handed to the
library), the current HIP stream, and torch.distributed for the data-parallel gradient all-reduce.
All math runs in libpmgt_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    """Flat fp32 parameters/gradients, frozen feature tables, RNG state and scratch for one GPU."""

    def __init__(self, config, dtype: str = "bf16", device: str = "cuda:0", seed: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("pmgt_amd.Engine needs a HIP device; there is no CPU fallback")
        self.lib = _lib.hip()
        self.config = config
        self.dtype_name = dtype
        # "fp8": the bf16 engine with e4m3 feature tables and fp8-MFMA feature / Q|K|V|C projections (include/pmgt_capi.h)
        self.dtype_code = {"fp32": _lib.DTYPE_F32, "bf16": _lib.DTYPE_BF16, "fp8": _lib.DTYPE_FP8}[dtype]
        self.torch_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp8": torch.bfloat16}[dtype]
        self.tables: List[torch.Tensor] = []          # frozen feature tables, one per modality (set_tables)
        self.table_scale: tuple = ()                  # fp8 mode: value = e4m3 byte * table_scale[m]
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        feats = [int(f) for f in config.feat_hidden_sizes]
        if not 1 <= len(feats) <= _lib.MAX_FEATS:
            raise ValueError(f"feat_hidden_sizes={feats}: the HIP path takes 1 .. {_lib.MAX_FEATS} modalities "
                             "(the reference's trainer builds two: visual, textual; pmgt/pmgt/trainer.py:114-125)")
        self.n_feats = len(feats)
        self.cfg_c = _lib.PMGTConfigC(
            config.hidden_size, config.num_hidden_layers, config.num_attention_heads, config.intermediate_size,
            len(feats), (C.c_int * _lib.MAX_FEATS)(*feats), config.max_position_embeddings, config.layer_norm_eps, config.beta,
            config.hidden_dropout_prob, config.attention_probs_dropout_prob, self.dtype_code)
        self.h = self.lib.pmgt_engine_create(C.byref(self.cfg_c))
        if not self.h:
            raise ValueError(self.lib.pmgt_last_error().decode())
        self.n_params = int(self.lib.pmgt_param_count(self.h))
        self.entries: List[dict] = []
        name = C.create_string_buffer(256)
        off, numel = C.c_int64(), C.c_int64()
        rows, cols, decay = C.c_int(), C.c_int(), C.c_int()
        for i in range(self.lib.pmgt_param_num_entries(self.h)):
            _lib.check(self.lib.pmgt_param_entry(self.h, i, name, 256, C.byref(off), C.byref(numel), C.byref(rows),
                                                 C.byref(cols), C.byref(decay)))
            shape = (rows.value, cols.value) if cols.value > 0 else (rows.value,)
            self.entries.append(dict(name=name.value.decode(), offset=off.value, numel=numel.value, shape=shape,
                                     decay=bool(decay.value)))
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        dm = torch.zeros(self.n_params, dtype=torch.uint8)
        for e in self.entries:
            if e["decay"]:
                dm[e["offset"]: e["offset"] + e["numel"]] = 1
        self.decay_mask = dm.to(self.device)
        self.rng_state = torch.tensor([seed, 0], dtype=torch.int64, device=self.device)
        self.n_nodes = 0
        self._ws: Optional[torch.Tensor] = None
        # optimizer state (created lazily)
        self.exp_avg = self.exp_avg_sq = None
        self.opt_step = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.opt_scalars = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.opt_scratch = torch.zeros(1024, dtype=torch.float32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.pmgt_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------
    def entry(self, name: str) -> dict:
        if not hasattr(self, "_by_name"):
            self._by_name = {e["name"]: e for e in self.entries}
        return self._by_name[name]

    def view(self, name: str, grad: bool = False) -> torch.Tensor:
        e = self.entry(name)
        buf = self.grads if grad else self.params
        return buf[e["offset"]: e["offset"] + e["numel"]].view(*e["shape"])

    def named_views(self, grad: bool = False) -> Dict[str, torch.Tensor]:
        return {e["name"]: self.view(e["name"], grad) for e in self.entries}

    def load_params(self, params: Dict[str, torch.Tensor]):
        for e in self.entries:
            self.view(e["name"]).copy_(params[e["name"]].to(torch.float32))
        self.check_layernorm_carrier()

    # The default bf16 path at hidden size 256 keeps no copy of a LayerNorm's input: the backward takes the normalised row from the
    # LayerNorm OUTPUT, x^ = (y - beta) / gamma.  y is bf16, so x^ comes back with an absolute error of 2^-9 (|x^| + |beta / gamma|):
    # harmless for the reference's initialisation (gamma = 1, beta = 0) and anything near it, not for a checkpoint whose LayerNorm has
    # |beta| >> |gamma| in some channel.  Loading parameters checks that ratio and falls back to stored inputs ("store_ln_input").
    LN_CARRIER_MAX_RATIO = 8.0
    LN_CARRIER_MIN_GAMMA = 1e-6

    def layernorm_carrier_ratio(self) -> float:
        """max over the encoder's LayerNorm channels of |beta| / |gamma| (inf with a (nearly) zero gamma)."""
        worst = 0.0
        for e in self.entries:
            if e["name"].endswith("LayerNorm.weight") and "encoder.layer" in e["name"]:
                g = self.view(e["name"]).abs()
                b = self.view(e["name"][:-len("weight")] + "bias").abs()
                if bool((g < self.LN_CARRIER_MIN_GAMMA).any()):
                    return float("inf")
                worst = max(worst, float((b / g).max()))
        return worst

    def carrier_needs_stored_inputs(self) -> bool:
        """True when the parameters no longer allow x^ from the LayerNorm output and the engine still runs that form."""
        return not self.get_option("store_ln_input") and self.layernorm_carrier_ratio() > self.LN_CARRIER_MAX_RATIO

    def check_layernorm_carrier(self) -> float:
        """max over the encoder's LayerNorm channels of |beta| / |gamma|; above LN_CARRIER_MAX_RATIO -- or with a channel whose
        gamma is (nearly) zero, where x^ cannot be recovered from the output at all -- the engine switches to stored LayerNorm
        inputs.  Called wherever parameters arrive in bulk (load_params, the modules' load_state_dict), by the Trainer every
        `check_carrier_every` optimizer steps and before a step is captured into a graph."""
        worst = self.layernorm_carrier_ratio()
        if worst > self.LN_CARRIER_MAX_RATIO and not self.get_option("store_ln_input"):
            if getattr(self, "_live_graphs", 0) > 0:
                raise RuntimeError(f"pmgt_amd: LayerNorm |beta / gamma| reaches {worst:.1f}, so the backward must switch to stored "
                                   "LayerNorm inputs, but captured steps of this engine are alive and keep the old kernels: call "
                                   "Trainer.drop_captured_steps() (run_live does it by itself) and capture again")
            import warnings
            warnings.warn(f"pmgt_amd: LayerNorm |beta / gamma| reaches {worst:.1f}: x^ from the LayerNorm output would lose "
                          "precision, storing LayerNorm inputs instead (option store_ln_input)")
            self.set_option("store_ln_input", 1)
        return worst

    # ---- path options (include/pmgt_ops.h): state of THIS engine, not of the process ----------------
    def set_option(self, key: str, value) -> None:
        if getattr(self, "_live_graphs", 0) > 0 and bool(value) != self.get_option(key):
            # a captured step keeps the kernels it was captured with, while the workspace would be re-carved for the new option
            raise RuntimeError(f"pmgt_amd: option {key!r} cannot change while a captured step of this engine is alive "
                               "(drop the replay handle first)")
        _lib.check(self.lib.pmgt_engine_set_option(self.h, key.encode(), 1 if value else 0))
        self.__dict__.pop("_ws_bytes", None)          # workspace carving depends on some options

    def get_option(self, key: str) -> bool:
        v = int(self.lib.pmgt_engine_get_option(self.h, key.encode()))
        if v < 0:
            raise KeyError(self.lib.pmgt_last_error().decode())
        return bool(v)

    def set_tables(self, *tables):
        """Frozen feature tables [N+2, F_m], one per modality in the order of `feat_hidden_sizes` (visual, textual for the
        reference's trainer; pmgt/pmgt/models.py:40-54), cast once to the engine dtype."""
        if len(tables) == 1 and isinstance(tables[0], (list, tuple)):
            tables = tuple(tables[0])
        if len(tables) != self.n_feats:
            raise ValueError(f"set_tables: {len(tables)} tables for {self.n_feats} modalities (models.py:49-50)")
        tabs, scales = [], []
        for a, F in zip(tables, self.config.feat_hidden_sizes):
            if tuple(a.shape)[1:] != (F,) or (tabs and a.shape[0] != tabs[0].shape[0]):
                raise ValueError(f"set_tables: table of shape {tuple(a.shape)}, expected [n_nodes + 2, {F}]")
            t = torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a)
            if self.dtype_name == "fp8":
                # one scale per table: value = e4m3 byte * (max|table| / 448), quantised once by the library's kernel
                t = t.to(self.device, dtype=torch.float32).contiguous()
                amax = float(t.abs().max())
                scale = np.float32(amax) / np.float32(448.0) if amax > 0 else np.float32(1.0)
                inv = np.float32(448.0) / np.float32(amax) if amax > 0 else np.float32(1.0)
                q = torch.empty(t.shape, dtype=torch.uint8, device=self.device)
                _lib.check(self.lib.pmgt_quantize_e4m3(_ptr(t), _ptr(q), t.numel(), float(inv), _stream()))
                tabs.append(q)
                scales.append(float(scale))
            else:
                tabs.append(t.to(self.device, dtype=self.torch_dtype).contiguous())
                scales.append(0.0)
        self.tables = tabs
        self.table_scale = tuple(scales)
        self.n_nodes = int(tabs[0].shape[0] - 2)

    def dequantized_tables(self):
        """fp8 mode: the feature values the kernels see (fp32 [N+2, F_m]) -- what a checker must use as the tables."""
        assert self.dtype_name == "fp8"
        out = []
        for q, sc in zip(self.tables, self.table_scale):
            o = torch.empty(q.shape, dtype=torch.float32, device=self.device)
            _lib.check(self.lib.pmgt_dequantize_e4m3(_ptr(q), _ptr(o), q.numel(), sc, _stream()))
            out.append(o)
        return out

    def cast(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 device tensor -> engine dtype through the library's own cast kernel."""
        x = x.to(self.device, torch.float32).contiguous()
        out = torch.empty(x.shape, dtype=self.torch_dtype, device=self.device)
        _lib.check(self.lib.pmgt_cast_from_f32(self.dtype_code, _ptr(x), _ptr(out), x.numel(), _stream()))
        return out

    def _tensors(self, grad_buffer: Optional[torch.Tensor] = None):
        g = self.grads if grad_buffer is None else grad_buffer
        ptrs = [t.data_ptr() for t in self.tables] + [None] * (_lib.MAX_FEATS - len(self.tables))
        scales = list(self.table_scale) + [0.0] * (_lib.MAX_FEATS - len(self.table_scale))
        return _lib.TensorsC(self.params.data_ptr(), g.data_ptr(), (C.c_void_p * _lib.MAX_FEATS)(*ptrs), self.n_nodes,
                             self.rng_state.data_ptr(), (C.c_float * _lib.MAX_FEATS)(*scales))

    def _feat_args(self, feats):
        """[n_seq, S, F_m] tensors, one per modality -> (device copies kept alive by the caller, host pointer array)."""
        if len(feats) != self.n_feats:
            raise ValueError(f"{len(feats)} feature tensors for {self.n_feats} modalities (modeling_pmgt.py:195-201)")
        dev = [f.to(self.device, self.torch_dtype).contiguous() for f in feats]
        return dev, (C.c_void_p * self.n_feats)(*[f.data_ptr() for f in dev])

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    def _workspace_bytes(self, n_seq: int, S: int, B: int, training: bool) -> int:
        key = (n_seq, S, B, bool(training))
        cache = self.__dict__.setdefault("_ws_bytes", {})
        if key not in cache:
            cache[key] = int(self.lib.pmgt_workspace_bytes(self.h, n_seq, S, B, 1 if training else 0))
        return cache[key]

    OUTPUT_RING = 4
    OUTPUT_RINGS_MAX = 8

    def _outputs(self, B: int, P: int, S: int, want_hidden: bool, private: bool = False):
        """Output tensors of pretrain_step from a ring of OUTPUT_RING persistent sets (no allocator call and no memset launch
        on the step's critical path): what a call returns stays valid until OUTPUT_RING - 1 further calls of the same kind
        have been made -- clone a result that must live longer.  A ring is keyed by (B, S, want_hidden) and sized for the
        LARGEST pair count seen so far (the pair count changes with almost every live batch: logits are a [:P] view of a
        capacity-sized buffer, grown geometrically); rings are evicted least-recently-used, never all at once.
        private = True hands out a set of its own that no later call reuses (hipGraph capture: replays keep writing it)."""
        d = self.config.hidden_size

        def make(cap):
            return (torch.empty(3, dtype=torch.float32, device=self.device),
                    torch.empty(cap, dtype=torch.float32, device=self.device),
                    torch.empty(B, S, d, dtype=self.torch_dtype, device=self.device) if want_hidden else None,
                    torch.zeros(1, dtype=torch.int32, device=self.device))
        if private:
            loss, logits, hidden, count = make(P)
            return loss, logits, hidden, count
        key = (B, S, bool(want_hidden))
        rings = self.__dict__.setdefault("_out_rings", {})
        ring = rings.pop(key, None)
        if ring is None or ring["cap"] < P:
            cap = max(P, 0 if ring is None else int(ring["cap"] * 1.25) + 1, B * 10)
            ring = dict(next=0, cap=cap, sets=[make(cap) for _ in range(self.OUTPUT_RING)])
        rings[key] = ring                     # dicts keep insertion order: re-inserting marks the ring most recently used
        while len(rings) > self.OUTPUT_RINGS_MAX:
            rings.pop(next(iter(rings)))
        loss, logits, hidden, count = ring["sets"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % self.OUTPUT_RING
        return loss, logits[:P], hidden, count

    # ---- PMGT.forward (+ backward) ---------------------------------------------------------------
    def pretrain_step(self, batch, training: bool, backward: bool = False, accumulate: bool = False,
                      nfr_inject=None, random_node_ratio: float = 0.02, mask_node_ratio: float = 0.16,
                      want_hidden: bool = True, grad_buffer: Optional[torch.Tensor] = None, private_outputs: bool = False):
        """batch = (target_dict, pair_dict, num_pairs, labels) of device tensors (pmgt_collate_fn layout).
        nfr_inject = (masked_ids [B,S] int64, nfr_targets [B,S] int64 with -1 = not masked)."""
        tgt, pair, num_pairs, labels = batch
        ids = tgt["node_ids"].contiguous()
        B, S = ids.shape
        P = int(pair["node_ids"].shape[0])
        n_seq = B + P + (B if training else 0)
        ws = self._workspace(self._workspace_bytes(n_seq, S, B, training))
        loss, logits, hidden, count = self._outputs(B, P, S, want_hidden, private_outputs)
        if not training:
            count.zero_()                # only the training path writes the number of masked rows
        keep = [ids, tgt["attention_mask"].contiguous(), pair["node_ids"].contiguous(),
                pair["attention_mask"].contiguous(), num_pairs.contiguous(), labels.to(torch.float32).contiguous()]
        bc = _lib.BatchC(B, P, S, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(),
                         keep[4].data_ptr(), keep[5].data_ptr(), 0, 0, random_node_ratio, mask_node_ratio)
        if nfr_inject is not None:
            mi, mt = nfr_inject[0].contiguous(), nfr_inject[1].contiguous()
            keep += [mi, mt]
            bc.nfr_masked_ids, bc.nfr_targets = mi.data_ptr(), mt.data_ptr()
        oc = _lib.OutputsC(loss.data_ptr(), logits.data_ptr(), 0 if hidden is None else hidden.data_ptr(), count.data_ptr())
        flags = (_lib.FLAG_TRAINING if training else 0) | (_lib.FLAG_BACKWARD if backward else 0) | \
                (_lib.FLAG_ACCUMULATE if accumulate else 0)
        tc = self._tensors(grad_buffer)
        _lib.check(self.lib.pmgt_pretrain_step(self.h, C.byref(tc), C.byref(bc), C.byref(oc), _ptr(ws), ws.numel(),
                                               flags, _stream()))
        self._raise_hook_error()
        return dict(loss=loss[0], gsr=loss[1], nfr=loss[2], losses=loss, logits=logits, last_hidden_state=hidden,
                    nfr_count=count)

    # ---- PMGTModel.forward ---------------------------------------------------------------------------
    def encode(self, ids: Optional[torch.Tensor] = None, feats=None, attention_mask: Optional[torch.Tensor] = None,
               output_hidden_states: bool = False, output_attentions: bool = False):
        cfg = self.config
        if ids is not None:
            n_seq, S = ids.shape
        else:
            n_seq, S = feats[0].shape[:2]
        d, L, H = cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads
        ws = self._workspace(self.lib.pmgt_workspace_bytes(self.h, n_seq, S, 1, 0))
        last = torch.empty(n_seq, S, d, dtype=self.torch_dtype, device=self.device)
        hs = torch.empty(L + 1, n_seq, S, d, dtype=self.torch_dtype, device=self.device) if output_hidden_states else None
        pr = torch.empty(L, n_seq, H, S, S, dtype=torch.float32, device=self.device) if output_attentions else None
        m = None if attention_mask is None else attention_mask.to(self.device, torch.float32).contiguous()
        tc = self._tensors()
        if ids is not None:
            ids = ids.to(self.device).contiguous()
            _lib.check(self.lib.pmgt_encode_ids(self.h, C.byref(tc), _ptr(ids), _ptr(m), n_seq, S, _ptr(last), _ptr(hs),
                                                _ptr(pr), _ptr(ws), ws.numel(), _stream()))
        else:
            dev, fp = self._feat_args(feats)
            _lib.check(self.lib.pmgt_encode_feats(self.h, C.byref(tc), fp, _ptr(m), n_seq, S, _ptr(last),
                                                  _ptr(hs), _ptr(pr), _ptr(ws), ws.numel(), _stream()))
        return last, hs, pr

    # ---- PMGTModel.forward with activations kept for a backward driven by the caller's head ------------------
    def encode_train(self, ids: Optional[torch.Tensor] = None, feats=None, attention_mask: Optional[torch.Tensor] = None,
                     training: bool = True):
        """Returns (last_hidden_state [n_seq, S, d] in the engine dtype, state) — pass `state` to encode_backward."""
        if ids is not None:
            n_seq, S = ids.shape
            ids = ids.to(self.device).contiguous()
            dev, fp = None, None
        else:
            n_seq, S = feats[0].shape[:2]
            dev, fp = self._feat_args(feats)
        m = torch.ones(n_seq, S, dtype=torch.float32, device=self.device) if attention_mask is None else \
            attention_mask.to(self.device, torch.float32).contiguous()
        nbytes = int(self.lib.pmgt_workspace_bytes(self.h, n_seq, S, 1, 1))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)      # owned by this call's state, not the shared cache
        last = torch.empty(n_seq, S, self.config.hidden_size, dtype=self.torch_dtype, device=self.device)
        tc = self._tensors()
        flags = _lib.FLAG_TRAINING if training else 0
        _lib.check(self.lib.pmgt_encode_train(self.h, C.byref(tc), _ptr(ids), fp, _ptr(m), n_seq, S, _ptr(last),
                                              _ptr(ws), nbytes, flags, _stream()))
        return last, dict(ws=ws, feats=dev, feat_ptrs=fp, n_seq=n_seq, S=S, flags=flags)

    def encode_backward(self, state: dict, d_last: torch.Tensor, accumulate: bool = False,
                        grad_buffer: Optional[torch.Tensor] = None):
        """d loss / d last_hidden_state -> gradients of the `bert.*` entries in self.grads (or grad_buffer)."""
        d_last = d_last.to(self.device, self.torch_dtype).contiguous()
        assert tuple(d_last.shape) == (state["n_seq"], state["S"], self.config.hidden_size)
        tc = self._tensors(grad_buffer)
        flags = state["flags"] | (_lib.FLAG_ACCUMULATE if accumulate else 0)
        ws = state["ws"]
        _lib.check(self.lib.pmgt_encode_backward(self.h, C.byref(tc), state["feat_ptrs"], _ptr(d_last),
                                                 state["n_seq"], state["S"], _ptr(ws), ws.numel(), flags, _stream()))
        self._raise_hook_error()

    # ---- clip + AdamW ----------------------------------------------------------------------------------
    def ensure_optimizer_state(self):
        """Adam moments, allocated and zero-filled OUTSIDE any stream capture: a zero-fill recorded into a captured step would
        reset both moments on every replay (the device-side step counter keeps counting, so nothing would look wrong)."""
        if self.exp_avg is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("pmgt_amd: the optimizer state must exist before a step is captured "
                                   "(call Engine.ensure_optimizer_state() first)")
            self.exp_avg = torch.zeros_like(self.params)
            self.exp_avg_sq = torch.zeros_like(self.params)

    def optimizer_step(self, lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=None):
        self.ensure_optimizer_state()
        ac = _lib.AdamC(self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.decay_mask.data_ptr(), lr, weight_decay,
                        betas[0], betas[1], eps, float(max_grad_norm) if max_grad_norm else 0.0,
                        self.opt_step.data_ptr(), self.opt_scalars.data_ptr(), self.opt_scratch.data_ptr())
        tc = self._tensors()
        _lib.check(self.lib.pmgt_optimizer_step(self.h, C.byref(tc), C.byref(ac), _stream()))

    def set_grad_ready_hook(self, fn=None):
        """fn(offset, numel) is called on the launching thread, in stream order, as soon as grads[offset: offset + numel]
        is final during a backward pass (buckets: NFR head, layers L-1 .. 0, embeddings); None removes the hook.
        An exception raised by `fn` is re-raised by the pretrain_step / encode_backward call that triggered it."""
        self._hook_error = None
        if fn is None:
            self._grad_cb = _lib.GRAD_READY_FN(0)
        else:
            def _cb(_user, off, numel):
                try:
                    fn(int(off), int(numel))
                except BaseException as exc:         # never unwind through the C frames
                    self._hook_error = exc
            self._grad_cb = _lib.GRAD_READY_FN(_cb)            # keep the thunk alive as long as the engine holds its address
        self.lib.pmgt_engine_set_grad_ready_callback(self.h, self._grad_cb, None)

    def _raise_hook_error(self):
        err, self._hook_error = getattr(self, "_hook_error", None), None
        if err is not None:
            raise err

    def set_overlap(self, on: bool):
        """Partial-sum reductions of the backward pass on the engine's side stream, or (default) on the caller's stream."""
        self.set_option("side_stream_reduce", on)

    # ---- phase timers -------------------------------------------------------------------------------------
    def profile_begin(self):
        _lib.check(self.lib.pmgt_profile_begin(self.h))

    def profile_sequence(self):
        """Phase names recorded since profile_begin(), in launch order (no wait; call before profile_end())."""
        buf = C.create_string_buffer(1 << 18)
        _lib.check(self.lib.pmgt_profile_sequence(self.h, buf, len(buf)))
        return buf.value.decode().split()

    def profile_records(self):
        """[(phase, ms)] of every launch group recorded since profile_begin(), in launch order (waits for the events; call before profile_end())."""
        buf = C.create_string_buffer(1 << 20)
        _lib.check(self.lib.pmgt_profile_records(self.h, buf, len(buf)))
        return [(ln.split()[0], float(ln.split()[1])) for ln in buf.value.decode().splitlines()]

    def profile_end(self) -> Dict[str, tuple]:
        """{phase: (launch groups, total ms)} measured with HIP events on the launch stream."""
        buf = C.create_string_buffer(1 << 16)
        _lib.check(self.lib.pmgt_profile_end(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out

    def grad_norm(self) -> torch.Tensor:
        """Pre-clip global gradient norm of the last optimizer_step (device scalar)."""
        return self.opt_scalars[3]
