"""CPU swap of KV blocks on the GPU engine — SURVEY §8 f4 (second half); block_manager.rs:870-1010, scheduler.rs:303-338,826-955,
cache::swap_blocks (runner.rs:1641-1645).  A preempted sequence's blocks are copied to pinned host memory, its GPU blocks
are reused by the other sequence, and after the swap-in (into different blocks) it must continue exactly as if it had never
left: greedy tokens equal those of an engine with a cache large enough never to preempt."""
import numpy as np
import pytest

from oracle import model as om
from tests.test_gpu_engine import small_cfg
from vllm_rs_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _run(eng, prompts, max_tokens):
    rids = [eng.add_request(p, max_tokens=max_tokens, ignore_eos=True) for p in prompts]
    n = 0
    while eng.has_unfinished():
        eng.step()
        n += 1
        assert n < 5000
    return [np.asarray(eng.output(r)).tolist() for r in rids]


@pytest.mark.parametrize("fp8", [False, True])
def test_swapped_sequence_continues_bit_identically(fp8):
    cfg = small_cfg()
    w = om.make_random_checkpoint(cfg, 3)
    r = np.random.default_rng(5)
    prompts = [r.integers(5, 500, size=100).astype(np.uint32) for _ in range(2)]
    kw = dict(max_num_seqs=4, max_model_len=512, use_graph=False, fp8_kvcache=fp8)
    big = Engine(cfg, num_gpu_blocks=64, **kw).load_weights(w)
    want = _run(big, prompts, 200)
    assert big.swap_stats()[2] == 0
    big.close()
    # min_tokens_left_for_swap (scheduler.rs:50, 1000 in the reference) keeps the sequence out until the other one is done:
    # with 0 it would come straight back, find no free slot and be dropped by the no-progress rule (engine.rs:1103-1120)
    small = Engine(cfg, num_gpu_blocks=9, cpu_mem_fold=1.0, swap_cooling_ms=-1, min_tokens_left_for_swap=300, **kw).load_weights(w)
    got = _run(small, prompts, 200)
    cpu, free_cpu, out_blocks, in_blocks = small.swap_stats()
    small.close()
    assert out_blocks >= 3 and in_blocks == out_blocks and free_cpu == cpu, (cpu, free_cpu, out_blocks, in_blocks)
    assert [len(g) for g in got] == [200, 200]
    assert got == want
