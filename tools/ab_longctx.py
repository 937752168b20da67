"""A/B of environment variants on ONE box for the long-context decode legs of bench.py (bs 32 at ctx 1024 / 4096, bs 1 at ctx 8000):
    python tools/ab_longctx.py rounds "VAR=1,VAR2=x" "VAR=2" ...     ('-' = no variables)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import bench
from vllm_rs_amd import _lib, engine as E
L = _lib.load()
cfg = dict(E.LLAMA3_8B)
out = {}
for fp8 in (0, 1):
    eng = E.Engine(cfg, max_num_seqs=32, max_model_len=8192, num_gpu_blocks=2304, use_graph=True, seed=1234, cpu_mem_fold=0.0, fp8_kvcache=bool(fp8)).init_synthetic()
    for bs, ctx, steps in ((32, 1024, 16), (32, 4096, 8), (1, 8000, 16)):
        dt, _, _ = bench.run_decode(eng, bench.make_prompts(bs, ctx, cfg["vocab_size"], seed=77 + ctx), 4, steps, L.vra_device_sync)
        out["%%sbs%%d_ctx%%d_ms" %% ("fp8_" if fp8 else "", bs, ctx)] = round(dt * 1e3 / steps, 4)
    eng.close()
print(json.dumps(out))
'''
rounds = int(sys.argv[1])
for r in range(rounds):
    for v in sys.argv[2:]:
        env = dict(os.environ)
        for kv in filter(None, (v if v != "-" else "").split(",")):
            k, _, val = kv.partition("=")
            env[k] = val
        p = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(f"round {r} {v:28s} {line[-1] if line else 'FAILED: ' + p.stderr[-400:]}", flush=True)
