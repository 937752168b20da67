"""decode over an FP8 KV cache at long contexts (the bench's long_context_decode_fp8_kv leg, alone)"""
import sys
sys.path.insert(0, ".")
import bench
from vllm_rs_amd import _lib, engine as E
L = _lib.load()
cfg = dict(E.LLAMA3_8B)
e8 = E.Engine(cfg, max_num_seqs=32, max_model_len=8192, num_gpu_blocks=8192, use_graph=True, seed=1234, fp8_kvcache=True, cpu_mem_fold=0.0).init_synthetic()
for bs, ctx in ((1, 8000), (32, 1024), (32, 4096)):
    dtl, _, _ = bench.run_decode(e8, bench.make_prompts(bs, ctx, cfg["vocab_size"], seed=77 + ctx), 4, 16, lambda: L.vra_device_sync())
    print(f"fp8 kv bs={bs} ctx={ctx}: {dtl * 1e3 / 16:.3f} ms/step")
e8.close()
