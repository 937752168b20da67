"""Llama-3-70B widths (one layer) sharded over eight runner processes that share ONE GPU — the oversubscribed stand-in of
tests/test_gpu_tp.py::test_tp8_llama3_70b_widths_one_gpu — run REPS prefill and REPS decode forwards in one session, every
forward compared with the TP ORACLE (not only with the first run), and on a deviation traced to the first rank and stage of layer 0
that leaves the per-stage oracle (tests/tp_stages.py).  A forward that fails loudly (bounded wait expired) is reported with the
waiter's view (slice, peer, epoch expected, flag read) and the session goes on with fresh runner processes.
    python tools/tp8_stress.py [reps] [sessions] [wall-clock budget s] [hardware queues per runner, cycled over the sessions: e.g. 1,4]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from oracle import model as om
from tests.test_gpu_engine import small_cfg, simple_tables, prefill_inputs, check_logits
from tests.tp_stages import first_deviation, oracle_stages
from vllm_rs_amd.runner import TPEngine
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 25
SESSIONS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
BUDGET = float(sys.argv[3]) if len(sys.argv) > 3 else 420.0
HWQ = (sys.argv[4] if len(sys.argv) > 4 else "1").split(",")
BF16 = 0


def main():
    t_start = time.time()
    cfg = small_cfg(hidden_size=8192, intermediate_size=28672, num_layers=1, num_heads=64, num_kv_heads=8, head_dim=128, vocab_size=1024, rope_theta=500000.0, quant_method="gptq")
    world = 8
    w = om.make_random_checkpoint(cfg, 21)
    r = np.random.default_rng(21)
    prompts = [r.integers(1, 1023, size=n).tolist() for n in (19, 6)]
    bt = simple_tables([len(p) + 4 for p in prompts])
    pre = prefill_inputs(prompts, bt)
    dids = np.array([5, 9], np.uint32)
    dpos = np.array([19, 6], np.int64)
    dslots = np.array([int(bt[b, dpos[b] // 64]) * 64 + dpos[b] % 64 for b in range(2)], np.int64)
    dctx = (dpos + 1).astype(np.uint32)
    o = om.OracleModel(cfg, w, num_blocks=16, tp_world=world)
    ost = om.OracleModel(cfg, w, num_blocks=16, tp_world=world)
    cases = [("prefill 19+6", (pre[0], pre[1], pre[2], bt, pre[3], pre[4])), ("decode 2", (dids, dpos, dslots, bt, dctx, None))]
    refs = [o.forward(*a) for _, a in cases]
    stages = [oracle_stages(ost, *a) for _, a in cases]
    print(f"# oracle ready after {time.time() - t_start:.0f} s", flush=True)
    stats = dict(forwards=0, deviating=0, loud=0)
    for sess in range(SESSIONS):
        if time.time() - t_start > BUDGET:
            break
        t0 = time.time()
        import os
        os.environ["VRA_TP_SHARED_HW_QUEUES"] = HWQ[sess % len(HWQ)]
        print(f"# session {sess}: GPU_MAX_HW_QUEUES={HWQ[sess % len(HWQ)]} in every runner", flush=True)
        try:
            with TPEngine(cfg, world, devices=[0] * world, transport="ipc", tensors=w, num_gpu_blocks=16, max_num_seqs=8, max_model_len=512, use_graph=False, timeout=300, snapshots=True) as tp:
                print(f"# session {sess}: 8 runners up after {time.time() - t0:.0f} s", flush=True)
                for (name, args), ref, st in zip(cases, refs, stages):
                    t1 = time.time()
                    for rep in range(REPS):
                        if time.time() - t_start > BUDGET:
                            break
                        g = tp.forward_raw(*[a for a in args if a is not None])
                        stats["forwards"] += 1
                        same = all(np.array_equal(g[k].view(np.uint32), g[0].view(np.uint32)) for k in range(world))
                        try:
                            import contextlib, io
                            with contextlib.redirect_stdout(io.StringIO()):
                                check_logits(g[0], ref, f"{name} rep {rep}", BF16)
                            ok = True
                        except AssertionError as e:
                            ok = False
                            stats["deviating"] += 1
                            print(f"session {sess} {name} rep {rep}: {e}; ranks identical: {same}", flush=True)
                            print(first_deviation(tp.snapshots(), st, BF16) or "   (every stage of layer 0 agrees with the oracle: the deviation is behind the layer)", flush=True)
                        if ok and not same:
                            print(f"session {sess} {name} rep {rep}: ranks disagree", flush=True)
                    print(f"# session {sess} {name}: {REPS} forwards in {time.time() - t1:.1f} s", flush=True)
        except (RuntimeError, TimeoutError) as e:
            stats["loud"] += 1
            stats.setdefault("loud_sessions", []).append((sess, HWQ[sess % len(HWQ)]))
            print(f"session {sess}: LOUD failure after {time.time() - t0:.0f} s: {str(e)[-2500:]}", flush=True)
    print("tp8_stress:", stats, f"({time.time() - t_start:.0f} s)")


if __name__ == "__main__":
    main()
