"""Bit-reproducibility of the tensor-parallel forward under oversubscription: Llama-3-70B widths over eight runner processes that
share ONE GPU (the setting of tests/test_gpu_tp.py::test_tp8_llama3_70b_widths_one_gpu), the same prefill and the same decode
step run REPS times each; every rank's logits are compared bit for bit with the first run and with rank 0.
    python tools/tp8_repro.py [reps]"""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import model as om
from tests.test_gpu_engine import small_cfg, simple_tables, prefill_inputs
from vllm_rs_amd.runner import TPEngine
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60


def main():
    cfg = small_cfg(hidden_size=8192, intermediate_size=28672, num_layers=1, num_heads=64, num_kv_heads=8, head_dim=128, vocab_size=1024, rope_theta=500000.0, quant_method="gptq")
    world = 8
    w = om.make_random_checkpoint(cfg, 21)
    r = np.random.default_rng(21)
    prompts = [r.integers(1, 1023, size=n).tolist() for n in (19, 6)]
    bt = simple_tables([len(p) + 4 for p in prompts])
    ids, pos, slots, ctx, cu = prefill_inputs(prompts, bt)
    dids = np.array([5, 9], np.uint32)
    dpos = np.array([19, 6], np.int64)
    dslots = np.array([int(bt[b, dpos[b] // 64]) * 64 + dpos[b] % 64 for b in range(2)], np.int64)
    dctx = (dpos + 1).astype(np.uint32)
    bad = 0
    with TPEngine(cfg, world, devices=[0] * world, transport="ipc", tensors=w, num_gpu_blocks=16, max_num_seqs=8, max_model_len=512, use_graph=False, timeout=900) as tp:
        for name, args in (("prefill 19+6", (ids, pos, slots, bt, ctx, cu)), ("decode 2", (dids, dpos, dslots, bt, dctx))):
            first = None
            for rep in range(REPS):
                g = tp.forward_raw(*args)
                if first is None:
                    first = g[0].copy()
                for k in range(world):
                    if not np.array_equal(g[k].view(np.uint32), first.view(np.uint32)):
                        bad += 1
                        print(f"{name} rep {rep} rank {k}: differs from the first run, max |d| {float(np.abs(g[k] - first).max()):.4f}; from rank 0 of this run {float(np.abs(g[k] - g[0]).max()):.4f}", flush=True)
            print(f"{name}: {REPS} runs x {world} ranks compared", flush=True)
    print("non-reproducible (run, rank) pairs:", bad)


if __name__ == "__main__":
    main()
