"""Trainer step of the PMGT pre-training hot path (reference: `PMGTTrainerModel.training_step`,
pmgt/pmgt/trainer.py:156-160, driven by PL's loop with `gradient_clip_val`, `DenseSparseAdamW` from
`get_optimizer`, and DDP when gpus > 1 — pmgt/base_trainer.py:35-68,309-322).

Data parallelism: one process per GPU, every rank holds a full replica (graph on the
host, feature tables + weights on the device); the ONLY exchange is one all-reduce(AVG) of the flat
fp32 gradient buffer per optimizer step over RCCL/xGMI (12 MB at L4/d256: far below one link's
bandwidth-delay product, so a single un-bucketed collective is the right shape); the clip uses the
post-reduce global norm, identical on all ranks.  Each rank takes indices rank::world of one seeded
permutation (DistributedSampler semantics).
"""
from __future__ import annotations

import queue
import threading
import time
from typing import Optional

import numpy as np
import torch

from .datasets import MODE_EVAL, MODE_INFERENCE, MODE_TRAIN
from .parallel import BucketedAllReduce, allreduce_mean_, broadcast_, gather_predictions, world


class Trainer:
    def __init__(self, engine, lr: float = 1e-3, weight_decay: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: Optional[float] = None, world_size: int = 1, accumulate_grad_batches: int = 1,
                 random_node_ratio: float = 0.02, mask_node_ratio: float = 0.16, overlap_allreduce: bool = True,
                 buckets: str = "two", check_carrier_every: int = 200, force_exchange: bool = False):
        self.engine = engine
        # force_exchange: run the data-parallel exchange (bucketed all-reduce from the engine's callback, wait in front of the optimizer)
        # even with ONE rank -- the N > 1 code path unchanged on a single-rank process group, so that RCCL executes it on a one-GPU box
        self.force_exchange = bool(force_exchange)
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.max_grad_norm = max_grad_norm
        self.world_size = world_size
        self.accum = max(1, accumulate_grad_batches)
        self.random_node_ratio, self.mask_node_ratio = random_node_ratio, mask_node_ratio
        self.last_loss = None
        self._micro = 0
        # every N optimizer steps re-check that the LayerNorm parameters still allow x^ = (y - beta) / gamma from the bf16 output
        # (Engine.check_layernorm_carrier: one small device -> host read per LayerNorm; 0 = only when parameters are loaded or a
        # step is captured)
        self.check_carrier_every = int(check_carrier_every)
        self._opt_steps = 0
        # world_size > 1: per-bucket all-reduce started from the engine's gradient-ready hook while the backward pass of
        # the earlier layers is still running (overlap_allreduce=False: ONE blocking all-reduce after the backward pass)
        self._exchange = None
        # buckets: "layer" = one collective per engine bucket (NFR head, every layer, embeddings); "two" = NFR head + encoder
        # layers as one collective (issued when layer 0's gradients are final), embeddings as the second; "one" = the whole buffer
        # after the backward pass (engine option one_bucket).  Default "two" = bench.py's default (one policy for the library and the measurement);
        # PROVISIONAL: chosen on one-rank RCCL runs (profiles/r05/rccl_single_rank_exchange.txt), where transport is free -- no multi-GPU A/B exists
        if buckets not in ("layer", "two", "one"):
            raise ValueError(f"buckets={buckets!r}: expected 'layer', 'two' or 'one'")
        self.buckets = buckets
        if (world_size > 1 or self.force_exchange) and overlap_allreduce:
            bounds = ()
            if buckets == "two" and engine.config.num_hidden_layers > 0:
                bounds = (engine.entry("bert.encoder.layer.0.attention.self.query.weight")["offset"],)
            engine.set_option("one_bucket", buckets == "one")
            self._exchange = BucketedAllReduce(engine.grads, boundaries=bounds)
            engine.set_grad_ready_hook(self._exchange.bucket_ready)

    def broadcast_parameters(self, src: int = 0):
        """DDP constructor semantics: every replica starts from rank `src`'s parameters."""
        if self.world_size > 1 or self.force_exchange:
            broadcast_(self.engine.params, src=src, force=self.force_exchange)

    def training_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        """loss = net(*batch)[0] with gradients left in engine.grads (pmgt/pmgt/trainer.py:156-160).  The returned device
        scalar lives in the engine's output ring (valid for the next Engine.OUTPUT_RING - 1 steps; clone it to keep it)."""
        if self._exchange is not None:      # gradients are exchanged once per optimizer step: on the last micro-batch
            self._exchange.enabled = (self.world_size > 1 or self.force_exchange) and self._micro == self.accum - 1
        out = self.engine.pretrain_step(batch, training=True, backward=True, accumulate=self._micro > 0,
                                        random_node_ratio=self.random_node_ratio, mask_node_ratio=self.mask_node_ratio,
                                        want_hidden=False, private_outputs=getattr(self, "_capturing", False))
        self.last_loss = out["loss"]
        self.last_outputs = out
        return out["loss"]

    def optimizer_step(self):
        eng = self.engine
        if self.world_size > 1 or self.force_exchange:
            done = self._exchange.wait() if self._exchange is not None else 0
            if done == 0:
                allreduce_mean_(eng.grads, force=self.force_exchange)
            elif done != eng.n_params:
                raise RuntimeError(f"gradient exchange covered {done} of {eng.n_params} elements")
        if self.accum > 1:
            eng.grads.div_(self.accum)
        eng.optimizer_step(lr=self.lr, weight_decay=self.weight_decay, betas=self.betas, eps=self.eps,
                           max_grad_norm=self.max_grad_norm)
        self._opt_steps += 1
        if self.check_carrier_every and self._opt_steps % self.check_carrier_every == 0 and not getattr(self, "_capturing", False):
            self._check_carrier()

    def _check_carrier(self):
        """The periodic LayerNorm-carrier guard of an EAGER step.  When the guard has to switch the engine to stored LayerNorm inputs while
        captured steps are alive, the ones this trainer owns (run_live(graphs=True)) are dropped first -- run_live re-captures them on its next
        call, exactly as its own periodic check does -- so a user who mixes run_live(graphs=True) with later eager train_step calls does not
        meet an error at the flip.  Replay handles the CALLER holds (capture_step) cannot be dropped from here: that case keeps the engine's
        error, which names the remedy."""
        eng = self.engine
        flips = eng.layernorm_carrier_ratio() > eng.LN_CARRIER_MAX_RATIO and not eng.get_option("store_ln_input")
        if flips and getattr(eng, "_live_graphs", 0) > 0 and self.__dict__.get("_live_replays"):
            import gc
            self.drop_captured_steps()
            gc.collect()
        eng.check_layernorm_carrier()

    def train_step(self, batch) -> torch.Tensor:
        """One micro-batch; steps the optimizer every `accumulate_grad_batches` calls."""
        loss = self.training_step(batch)
        self._micro += 1
        if self._micro == self.accum:
            self.optimizer_step()
            self._micro = 0
        return loss

    # ---- the whole step as ONE hipGraph ---------------------------------------------------------------------------
    def capture_step(self, batch, warmup: int = 2, capture_error_mode: str = "global"):
        """Captures train_step(batch) (mask -> forward -> losses -> backward -> clip + AdamW) into a hipGraph and
        returns `replay()`: the library never syncs or allocates and keeps every data-dependent count (masked rows,
        dropout step, AdamW step) on the device, so the captured launches stay valid step after step.  New batches
        are fed by copying into the tensors of `batch` (static input buffers), as with any captured graph.
        `warmup` eager steps run first (one-time kernel attribute calls are not capturable).  Single-GPU step only:
        the gradient all-reduce is not captured."""
        assert self.world_size == 1 and self.accum == 1 and not self.force_exchange, "capture covers the single-GPU, non-accumulating step"
        dev = self.engine.device
        # replays never run the Python-side guard: decide "x^ from the LayerNorm output or from stored inputs" once, on the
        # parameters as they are now, before the kernels are frozen into the graph
        self.engine.check_layernorm_carrier()
        # Adam moments exist before the capture: their zero-fill must not become a node of the graph (it would reset them on every replay)
        self.engine.ensure_optimizer_state()
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            for _ in range(warmup):
                self.train_step(batch)
            st.synchronize()
            graph = torch.cuda.CUDAGraph()
            self._capturing = True        # the graph gets an output set of its own: replays keep writing it, eager steps never do
            try:
                with torch.cuda.graph(graph, stream=st, capture_error_mode=capture_error_mode):
                    loss = self.train_step(batch)
                    outputs = self.last_outputs
            finally:
                self._capturing = False
        torch.cuda.current_stream(dev).wait_stream(st)

        def replay():
            graph.replay()
            self.last_loss = loss
            return loss
        import weakref
        eng = self.engine
        eng._live_graphs = getattr(eng, "_live_graphs", 0) + 1      # Engine.set_option refuses changes while a captured step lives
        weakref.finalize(graph, lambda: setattr(eng, "_live_graphs", eng._live_graphs - 1))
        replay.graph = graph
        replay.outputs = outputs          # loss / logits / nfr_count the replays write (kept alive with the graph)
        # everything the captured launches address by raw pointer lives as long as the replay handle: the engine's workspace (a later,
        # larger call makes the engine allocate a NEW one and drop its reference to this one), the static input tensors, the moments
        replay.keep = (eng._ws, batch, eng.exp_avg, eng.exp_avg_sq, eng.params, eng.grads)
        return replay

    def _hyper_key(self):
        """What a captured step froze into kernel arguments: a replay is only valid for the same values."""
        return (float(self.lr), float(self.weight_decay), tuple(float(b) for b in self.betas), float(self.eps),
                None if self.max_grad_norm is None else float(self.max_grad_norm), float(self.random_node_ratio), float(self.mask_node_ratio))

    def drop_captured_steps(self):
        """Forgets every step run_live(graphs=True) captured (call after changing lr / weight decay / clip / ratios / engine options by
        hand; run_live itself re-captures when the hyper-parameters it was captured with no longer match).  Waits for the GPU first: a
        graph must not be destroyed while a replay of it is still executing."""
        reps = self.__dict__.get("_live_replays")
        if reps:
            torch.cuda.synchronize(self.engine.device)
            reps.clear()

    # ---- live input pipeline: threaded C++ MCNSampling -> pinned buffers -> side-stream H2D ------------
    def run_live(self, sampler, node_ids: np.ndarray, batch_size: int, steps: int, threads: int = 8, depth: int = 3,
                 stall_timeout_s: float = 120.0, graphs: bool = False):
        """Training steps fed by the live host pipeline (the reference: a DataLoader over PMGTDataset, pmgt/pmgt/trainer.py:84-105):
        ONE producer thread runs the threaded C++ sampler into a pinned host slot and issues the slot's async H2D copies on a
        side stream into that slot's PRE-ALLOCATED device buffers (no allocator call, no record_stream on the step's path); the
        launch thread orders each step behind its copies with one event and hands the slot back with a completion event.
        graphs=True: the step is captured once per slot over that slot's device buffers (train-mode batches have a fixed shape: every
        target brings max_total_samples pairs) and replayed -- ONE launch per step, so a launch thread that loses its CPU for a
        millisecond in the middle of a step's ~70 launches (a shared host) no longer shows up as GPU idle time inside the step."""
        eng = self.engine
        dev = eng.device
        copy_stream = torch.cuda.Stream(device=dev)
        # the slots (pinned host + device buffers) live as long as the trainer: a second pass over the same shapes re-uses them -- and, with
        # graphs=True, the steps captured over them
        skey = (int(sampler.S), int(sampler.max_pairs(MODE_TRAIN)), batch_size, depth)      # shapes, not id(sampler): an id can be re-used
        cache = self.__dict__.setdefault("_live_slots", {})
        if skey not in cache:
            sl = [sampler.alloc(batch_size, MODE_TRAIN, pinned=True) for _ in range(depth)]
            cache[skey] = (sl, [{k: torch.empty_like(v, device=dev) for k, v in s_.items()} for s_ in sl])
        slots, dslots = cache[skey]
        n = len(node_ids)
        t_sample, t_wait, t_copy = [0.0], [0.0], [0.0]

        def produce(step, slot, done):
            ts = time.perf_counter()
            if done is not None:
                done.synchronize()     # the slot (pinned + device buffers) may be refilled once the step that read it is done
            lo = (step * batch_size) % max(n - batch_size, 1)
            tg = np.resize(node_ids[lo:], batch_size)
            t1 = time.perf_counter()
            tgt, pair, num_pairs, labels = sampler.batch(tg, MODE_TRAIN, out=slots[slot], threads=threads,
                                                        base_seed=7, counter=step * batch_size)
            t2 = time.perf_counter()
            P = int(pair["node_ids"].shape[0])
            d = dslots[slot]
            with torch.cuda.stream(copy_stream):
                for k, cnt in (("tgt_ids", batch_size), ("tgt_mask", batch_size), ("pair_ids", P), ("pair_mask", P),
                               ("num_pairs", batch_size), ("labels", P)):
                    d[k][:cnt].copy_(slots[slot][k][:cnt], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            b = ({"node_ids": d["tgt_ids"][:batch_size], "attention_mask": d["tgt_mask"][:batch_size]},
                 {"node_ids": d["pair_ids"][:P], "attention_mask": d["pair_mask"][:P]}, d["num_pairs"][:batch_size], d["labels"][:P])
            t3 = time.perf_counter()
            t_wait[0] += t1 - ts
            t_sample[0] += t2 - t1
            t_copy[0] += t3 - t2
            return b, ev

        pipe = ProducerPipeline(produce, steps, depth, stall_timeout_s=stall_timeout_s)
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        t0 = None
        pipe.start()
        # GPU-side view of the pipeline: events around every step on the launch stream; the gap between one step's end and the
        # next step's start is time the GPU had nothing of this stream to run (input not there yet, or the launch thread late)
        ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        t_launch = 0.0
        replays = self.__dict__.setdefault("_live_replays", {})       # (slot buffers, shape, hyper-parameters) -> captured step, kept across calls
        hyper = self._hyper_key()
        if graphs and any(k[-1] != hyper for k in replays):
            self.drop_captured_steps()       # lr / weight decay / clip / ratios changed since the capture: those are frozen kernel arguments
        checked_at = -1
        try:
            for i, (slot, (b, ev)) in enumerate(pipe):
                if t0 is None:
                    t0 = time.perf_counter()       # sustained rate: the clock starts when the first batch is there (the fill is reported)
                tl = time.perf_counter()
                torch.cuda.current_stream().wait_event(ev)
                ev_a[i].record()
                key = (b[0]["node_ids"].data_ptr(), tuple(b[0]["node_ids"].shape), tuple(b[1]["node_ids"].shape), hyper) if graphs else None
                if graphs and self.check_carrier_every and self._opt_steps % self.check_carrier_every == 0 and self._opt_steps != checked_at:
                    # replays never run optimizer_step's Python-side guard: look at the LayerNorm parameters here (one small read per
                    # LayerNorm every N steps); when they no longer allow x^ from the LayerNorm output, the captured steps are dropped,
                    # the engine switches to stored inputs and the slots are captured again below
                    checked_at = self._opt_steps
                    if eng.carrier_needs_stored_inputs():
                        self.drop_captured_steps()
                        eng.check_layernorm_carrier()
                if graphs and key in replays:
                    self.last_loss = replays[key]()
                    self._opt_steps += 1
                elif graphs and self.world_size == 1 and self.accum == 1 and not self.force_exchange:
                    # first batch of this slot: record the step (nothing executes during capture), then replay it like every later one.
                    # thread_local: the producer thread keeps issuing its own copies / event waits while this thread captures
                    if replays and eng.carrier_needs_stored_inputs():
                        self.drop_captured_steps()       # (capture_step switches the option; it must not find live graphs then)
                    replays[key] = self.capture_step(b, warmup=0, capture_error_mode="thread_local")      # (counts one optimizer step: the recording)
                    self.last_loss = replays[key]()
                else:
                    self.train_step(b)
                ev_b[i].record()
                pipe.release(slot, ev_b[i])   # the launch thread does not wait for the GPU: the producer does, before it refills
                t_launch += time.perf_counter() - tl
        finally:
            torch.cuda.synchronize()
            pipe.close()
        el = time.perf_counter() - t0
        idle = sum(ev_b[i - 1].elapsed_time(ev_a[i]) for i in range(1, steps))
        busy = sum(ev_a[i].elapsed_time(ev_b[i]) for i in range(steps))
        return {"nodes_per_s": round(steps * batch_size / el, 1), "ms_per_step": round(el / steps * 1e3, 3),
                "pipeline_fill_ms": round((t0 - t_start) * 1e3, 3),
                "sampler_threads": threads, "steps": steps, "pipeline_depth": depth, "graph_replay": bool(graphs),
                "gpu_step_ms": round(busy / steps, 3),
                "gpu_idle_ms_per_step": round(idle / max(steps - 1, 1), 3),
                "launch_thread_busy_ms_per_step": round(t_launch / steps * 1e3, 3),
                "launch_thread_waiting_for_input_ms_per_step": round(pipe.starved_s / steps * 1e3, 3),
                "producer_ms_per_batch": {"sampling": round(t_sample[0] / steps * 1e3, 3), "h2d_issue": round(t_copy[0] / steps * 1e3, 3),
                                          "waiting_for_a_free_slot": round(t_wait[0] / steps * 1e3, 3)}}


class PipelineError(RuntimeError):
    """The producer thread of a ProducerPipeline died or stalled; the original exception (if any) is the __cause__."""


class ProducerPipeline:
    """`depth` reusable slots filled by ONE producer thread and drained in order by the calling thread (the host side of
    `Trainer.run_live`: sampler -> pinned slot -> async copy).  `produce(step, slot, token)` runs on the producer thread;
    `token` is whatever the consumer passed to `release(slot, token)` when it handed the slot back (None the first time).
    A producer that raises (sampler ValueError for an isolated / out-of-range node, a failed pin or copy) or stops
    delivering for `stall_timeout_s` does not leave the consumer blocked: iteration raises PipelineError instead."""

    def __init__(self, produce, steps: int, depth: int, stall_timeout_s: float = 120.0, poll_s: float = 0.2):
        self.produce, self.steps, self.depth = produce, steps, depth
        self.stall_timeout_s, self.poll_s = stall_timeout_s, poll_s
        self.free_q: "queue.Queue" = queue.Queue()
        self.ready_q: "queue.Queue" = queue.Queue()
        for i in range(depth):
            self.free_q.put((i, None))
        self.starved_s = 0.0
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        try:
            for step in range(self.steps):
                while True:                      # a consumer that stopped early must not leave this thread blocked
                    if self._stop.is_set():
                        return
                    try:
                        slot, token = self.free_q.get(timeout=self.poll_s)
                        break
                    except queue.Empty:
                        continue
                self.ready_q.put(("item", slot, self.produce(step, slot, token)))
        except BaseException as exc:             # delivered to the consumer, which re-raises
            self.ready_q.put(("error", None, exc))

    def start(self):
        self._th.start()

    def release(self, slot: int, token=None):
        self.free_q.put((slot, token))

    def close(self):
        self._stop.set()
        if self._th.is_alive():
            self._th.join(timeout=5.0)

    def __iter__(self):
        for _ in range(self.steps):
            t0 = time.perf_counter()
            while True:
                try:
                    kind, slot, payload = self.ready_q.get(timeout=self.poll_s)
                    break
                except queue.Empty:
                    waited = time.perf_counter() - t0
                    if not self._th.is_alive() and self.ready_q.empty():
                        raise PipelineError("input pipeline: the producer thread exited without delivering a batch")
                    if waited > self.stall_timeout_s:
                        raise PipelineError(f"input pipeline: no batch for {waited:.0f} s (producer stalled)")
            self.starved_s += time.perf_counter() - t0
            if kind == "error":
                raise PipelineError(f"input pipeline: producer failed: {payload!r}") from payload
            yield slot, payload


def roc_auc_score(labels: np.ndarray, scores: np.ndarray) -> float:
    """sklearn.metrics.roc_auc_score for binary labels (what `_valid_and_test_epoch_end` logs as val/auc,
    pmgt/pmgt/trainer.py:182-195): Mann-Whitney U with midranks for ties."""
    labels = np.asarray(labels).astype(bool)
    scores = np.asarray(scores, dtype=np.float64)
    n_pos, n_neg = int(labels.sum()), int((~labels).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(scores, kind="mergesort")
    s = scores[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty_like(ranks)
    r[order] = ranks
    return float((r[labels].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


@torch.no_grad()
def evaluate(engine, sampler, node_ids: np.ndarray, batch_size: int = 256, threads: int = 8, seed: int = 0,
             distributed: bool = False):
    """Validation pass (pmgt/pmgt/trainer.py:162-195): eval-mode forward with 1 positive + 1 negative
    per target, sigmoid(logits) vs labels -> {'loss/val', 'val/auc'}.  `loss/val` is the mean of the per-batch losses
    (what `self.log("loss/val", ...)` aggregates over an epoch, weighted by batch size).  distributed=True under an
    initialised process group: rank r evaluates node_ids[r::W] and the predictions of all ranks are gathered, so every
    rank reports the same AUC over the whole validation set (the reference's AUC is per rank: no sync_dist); every node
    draws from the stream of its GLOBAL index, so the result equals the single-process evaluation of the same list."""
    node_ids = np.asarray(node_ids)
    rank, ws = world() if distributed else (0, 1)
    mine = node_ids[rank::ws]
    preds, labs = [np.empty(0, np.float32)], [np.empty(0, np.float32)]
    loss_sum = 0.0
    for lo in range(0, len(mine), batch_size):
        tg = mine[lo: lo + batch_size]
        tgt, pair, num_pairs, labels = sampler.batch(tg, MODE_EVAL, threads=threads, base_seed=seed, counter=rank + ws * lo,
                                                      counter_stride=ws)      # item j of this rank = item rank + ws * j of the list
        cu = lambda d: {k: v.to(engine.device) for k, v in d.items()}
        out = engine.pretrain_step((cu(tgt), cu(pair), num_pairs.to(engine.device), labels.to(engine.device)),
                                   training=False, want_hidden=False)
        preds.append(torch.sigmoid(out["logits"]).cpu().numpy())
        labs.append(labels.numpy())
        loss_sum += out["loss"].item() * len(tg)
    preds, labs = np.concatenate(preds), np.concatenate(labs)
    n_total = len(mine)
    if ws > 1:
        import torch.distributed as dist
        preds, labs = gather_predictions(preds, labs)
        parts = [None] * ws
        dist.all_gather_object(parts, (loss_sum, n_total))
        loss_sum, n_total = sum(p[0] for p in parts), sum(p[1] for p in parts)
    return {"loss/val": float(loss_sum / max(n_total, 1)), "val/auc": roc_auc_score(labs, preds)}


@torch.no_grad()
def export_embeddings(engine, sampler, n_nodes: int, batch_size: int = 1024, threads: int = 8, seed: int = 0) -> np.ndarray:
    """Inference / export (pmgt/pmgt/trainer.py:153-154,259-275; pmgt/base_trainer.py:400-407): CLS hidden
    state of every node in id order as fp32 [N, d] (contexts are still randomly sampled, as in the reference)."""
    out = np.empty((n_nodes, engine.config.hidden_size), dtype=np.float32)
    ids = np.arange(2, n_nodes + 2)
    for lo in range(0, n_nodes, batch_size):
        tg = ids[lo: lo + batch_size]
        tgt = sampler.batch(tg, MODE_INFERENCE, threads=threads, base_seed=seed, counter=lo)
        last, _, _ = engine.encode(ids=tgt["node_ids"].to(engine.device), attention_mask=tgt["attention_mask"].to(engine.device))
        out[lo: lo + len(tg)] = last[:, 0].float().cpu().numpy()
    return out
