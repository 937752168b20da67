"""one prefill-heavy run for profiling: two T-token prompts (argv[1], default 4096) through the Llama-3-8B-shape engine (eager launches)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from vllm_rs_amd import engine as E

MODEL = {"llama3-8b": E.LLAMA3_8B, "qwen2-7b-awq": E.QWEN2_7B}[os.environ.get("VRA_PREFILL_MODEL", "llama3-8b")]  # (profiling aid)
eng = E.Engine(dict(MODEL), max_num_seqs=8, max_model_len=8192, num_gpu_blocks=512, use_graph=False).init_synthetic()
r = np.random.default_rng(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for _ in range(2):
    rid = eng.add_request(r.integers(1000, 100000, size=T).astype(np.uint32), max_tokens=2, ignore_eos=True)
    while eng.has_unfinished():
        eng.step()
    t = eng.times(rid)
    print("ttft ms", t["first_token_ms"] - t["created_ms"])
eng.close()
