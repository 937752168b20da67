"""debug aid: Llama-3-70B widths sharded over eight runner processes on one GPU: rank-vs-rank and rank-vs-oracle logit differences"""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import model as om, oracle as orc
from tests.test_gpu_engine import small_cfg, simple_tables, prefill_inputs
from vllm_rs_amd.runner import TPEngine
H = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
I = int(sys.argv[2]) if len(sys.argv) > 2 else 28672


def main():
    cfg = small_cfg(hidden_size=H, intermediate_size=I, num_layers=1, num_heads=64, num_kv_heads=8, head_dim=128, vocab_size=1024, rope_theta=500000.0, quant_method="gptq")
    world = 8
    w = om.make_random_checkpoint(cfg, 21)
    o8 = om.OracleModel(cfg, w, num_blocks=16, tp_world=world)
    r = np.random.default_rng(21)
    prompts = [r.integers(1, 1023, size=n).tolist() for n in (19, 6)]
    bt = simple_tables([len(p) + 4 for p in prompts])
    ids, pos, slots, ctx, cu = prefill_inputs(prompts, bt)
    ref = o8.forward(ids, pos, slots, bt, ctx, cu)
    with TPEngine(cfg, world, devices=[0] * world, transport="ipc", tensors=w, num_gpu_blocks=16, max_num_seqs=8, max_model_len=512, use_graph=False, timeout=900) as tp:
        for rep in range(2):
            g = tp.forward_raw(ids, pos, slots, bt, ctx, cu)
            print("rep", rep, "max |rank r - rank 0|:", [float(np.abs(g[k] - g[0]).max()) for k in range(world)])
            print("   max |rank r - oracle|:", [round(float(np.abs(g[k] - ref).max()), 4) for k in range(world)], "scale", float(np.sqrt((ref ** 2).mean())))


if __name__ == "__main__":
    main()
