"""A/B of library builds on ONE box: for every library given (VRA_LIB of a child process each time, alternating `rounds` times) the
Llama-3-8B int4 decode step at bs 1 (and the batches listed) plus the four dequant-GEMV launches of a layer in isolation.
    python tools/ab_libs.py rounds libA.so libB.so ...      (paths relative to vllm_rs_amd/; 'default' = libvllm_rs_amd.so)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time
sys.path.insert(0, %r)
import bench
from vllm_rs_amd import _lib, engine as E
L = _lib.load()
cfg = dict(E.LLAMA3_8B)
eng = E.Engine(cfg, max_num_seqs=32, max_model_len=8192, num_gpu_blocks=2048, use_graph=True, seed=1234, cpu_mem_fold=0.0).init_synthetic()
out = {}
for bs in %r:
    dt, ms, _ = bench.run_decode(eng, bench.make_prompts(bs, 128, cfg["vocab_size"]), 16, 256 if bs == 1 else 64, L.vra_device_sync)
    out["bs%%d_ms_per_step" %% bs] = round(dt * 1e3 / (256 if bs == 1 else 64), 4)
for m in (1,):
    out["family_us_m%%d" %% m] = [round(eng.bench_gemm(w, m, 320) * 1e3, 2) for w in range(4)]
for ctx in (1024, 8000):
    lc, _, _ = bench.run_decode(eng, bench.make_prompts(1, ctx, cfg["vocab_size"], seed=77 + ctx), 4, 16, L.vra_device_sync)
    out["bs1_ctx%%d_ms" %% ctx] = round(lc * 1e3 / 16, 4)
lc, _, _ = bench.run_decode(eng, bench.make_prompts(32, 128, cfg["vocab_size"], seed=43), 8, 64, L.vra_device_sync)
out["bs32_ms"] = round(lc * 1e3 / 64, 4)
print(json.dumps(out))
'''


def main():
    rounds = int(sys.argv[1])
    libs = sys.argv[2:]
    batches = [1, 2, 4]
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ)
            name, _, evars = lib.partition("@")  # "lib.so@VAR=1,VAR2=x": environment of that child
            for kv in filter(None, evars.split(",")):
                k, _, v = kv.partition("=")
                env[k] = v
            if name != "default":
                env["VRA_LIB"] = os.path.join(ROOT, "vllm_rs_amd", name)
            p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, batches)], env=env, capture_output=True, text=True, cwd=ROOT)
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:]
            print(f"round {r} {lib:28s} {line}", flush=True)


if __name__ == "__main__":
    main()
