"""Host pipeline of Trainer.run_live (producer thread -> slots -> launch thread): a producer that dies or stalls must
surface as an exception on the consumer, never as a hang (round-1 advisor finding)."""
import threading
import time

import numpy as np
import pytest

from pmgt_amd.trainer import PipelineError, ProducerPipeline


def test_items_arrive_in_order_and_slots_are_recycled_with_their_tokens():
    seen_tokens = []

    def produce(step, slot, token):
        seen_tokens.append(token)
        return step * 10

    pipe = ProducerPipeline(produce, steps=7, depth=3)
    pipe.start()
    got = []
    for slot, item in pipe:
        got.append((slot, item))
        pipe.release(slot, ("done", item))
    pipe.close()
    assert [i for _, i in got] == [0, 10, 20, 30, 40, 50, 60]
    assert [s for s, _ in got] == [0, 1, 2, 0, 1, 2, 0]
    assert seen_tokens[:3] == [None, None, None] and seen_tokens[3:] == [("done", 0), ("done", 10), ("done", 20), ("done", 30)]


def test_sampler_error_reaches_the_consumer():
    """The C++ sampler refuses an isolated / out-of-range node with ValueError (rc -2 / -3): the launch thread re-raises."""
    from pmgt_amd.datasets import MODE_TRAIN, MCNSampler
    from pmgt_amd.graph import synthetic_graph
    smp = MCNSampler(synthetic_graph(50, 200, seed=1), 7)

    def produce(step, slot, token):
        tg = np.array([2, 3, 4, 5]) if step < 2 else np.array([2, 3, 10 ** 6, 5])     # step 2: node id outside the graph
        return smp.batch(tg, MODE_TRAIN, threads=2, base_seed=1, counter=4 * step)

    pipe = ProducerPipeline(produce, steps=5, depth=2)
    pipe.start()
    n = 0
    t0 = time.time()
    with pytest.raises(PipelineError) as ei:
        for slot, item in pipe:
            n += 1
            pipe.release(slot)
    pipe.close()
    assert n == 2 and isinstance(ei.value.__cause__, ValueError)
    assert time.time() - t0 < 10.0


def test_stalled_producer_times_out_and_early_consumer_exit_frees_the_thread():
    gate = threading.Event()

    def produce(step, slot, token):
        if step == 1:
            gate.wait(5.0)
        return step

    pipe = ProducerPipeline(produce, steps=3, depth=2, stall_timeout_s=0.5, poll_s=0.05)
    pipe.start()
    with pytest.raises(PipelineError, match="stalled"):
        for slot, item in pipe:
            pipe.release(slot)
    gate.set()
    pipe.close()
    assert not pipe._th.is_alive()
    # a consumer that stops early (exception in its own step) does not leave the producer blocked on the free queue
    pipe = ProducerPipeline(lambda step, slot, token: step, steps=100, depth=2, poll_s=0.05)
    pipe.start()
    it = iter(pipe)
    next(it)
    pipe.close()
    assert not pipe._th.is_alive()
