"""Kernel X (dequant pass + 256-row dense GEMM, csrc/gemm_dense.cuh) against kernel D, at the Llama-3-8B prefill shapes:
    python tools/gemm_dense_microbench.py [--check] [--tile 0|128|256 (| split-K slices << 16)] [M ...]
--check: sampled outputs against a float64 product of the dequantised weights (transpose-detecting random data), before timing."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from vllm_rs_amd import ops

L = ops.lib()
args = sys.argv[1:]
check = "--check" in args
tile = 0
if "--tile" in args:
    tile = int(args[args.index("--tile") + 1])
    del args[args.index("--tile"):args.index("--tile") + 2]
Ms = [int(v) for v in args if not v.startswith("--")] or [2048, 4096]
SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 14336), "down": (14336, 4096)}
e0, e1 = L.vra_event_create(), L.vra_event_create()
L.vra_debug_set_dense_prefill_min_rows(0)  # "D" below means kernel D: keep vra_wna16_gemm off the dense path


def bf16_to_f32(a):
    return (a.astype(np.uint32) << 16).view(np.float32)


def timed(fn, reps):
    for _ in range(2):
        fn()
    L.vra_device_sync()
    L.vra_event_record(e0, 0)
    for _ in range(reps):
        fn()
    L.vra_event_record(e1, 0)
    return L.vra_event_elapsed_ms(e0, e1) / reps


for M in Ms:
    tot_d = tot_x = tot_q = 0.0
    for which, (K, N) in SHAPES.items():
        dual = which == "gate_up"
        nt = 2 if dual else 1
        ws = [ops.DevBuf(K * N // 2) for _ in range(nt)]
        sc = [ops.DevBuf(K // 128 * N * 2) for _ in range(nt)]
        for i, w in enumerate(ws):
            L.vra_fill_hash_u32(w.ptr, K * N // 8, 11 + i, 0)
        for i, s in enumerate(sc):
            L.vra_fill_uniform(s.ptr, K // 128 * N, 21 + i, 0.002, 0.02, 0, 0)
        x = ops.DevBuf(M * K * 2)
        L.vra_fill_normal(x.ptr, M * K, 3, 0.0, 1.0, 0, 0)
        out_d, out_x = ops.DevBuf(M * N * 2), ops.DevBuf(M * N * 2)
        wd = ops.DevBuf(K * N * 2 * nt)
        nv = N * nt

        def run_d():  # (called with the dense path switched off: vra_wna16_gemm would take it from 768 rows)
            if dual:
                L.vra_wna16_gate_up_silu(x.ptr, ws[0].ptr, sc[0].ptr, None, ws[1].ptr, sc[1].ptr, None, out_d.ptr, M, K, N, 128, 0, 0, 0, 0)
            else:
                L.vra_wna16_gemm(x.ptr, ws[0].ptr, sc[0].ptr, None, None, None, out_d.ptr, M, K, N, 128, 0, 0, 0, 0)

        def run_q():
            for i in range(nt):
                L.vra_wna16_dequant_frag(ws[i].ptr, sc[i].ptr, None, wd.ptr, K, N, 128, 0, 0, 0, i if dual else 0, nt, 0)

        def run_x():
            L.vra_dense_frag_gemm(x.ptr, wd.ptr, None, None, out_x.ptr, M, K, nv, int(dual), 0, tile, 0)

        run_q()
        run_x()
        ops.check_error()
        if check:
            run_d()
            od = bf16_to_f32(out_d.numpy(np.uint16, (M, N)))
            ox = bf16_to_f32(out_x.numpy(np.uint16, (M, N)))
            xs = bf16_to_f32(x.numpy(np.uint16, (M, K))).astype(np.float64)
            rng = np.random.default_rng(5)
            rows = np.unique(np.concatenate([[0, 1, 15, 16, 127, 128, 255, 256, M - 1], rng.integers(0, M, 24)]))
            rows = rows[rows < M]
            cols = np.unique(np.concatenate([[0, 1, 15, 16, 63, 64, 255, 256, N - 1], rng.integers(0, N, 40)]))
            dense = []
            for i in range(nt):
                t = ops.DevBuf(K * N * 2)
                L.vra_wna16_dequant(ws[i].ptr, sc[i].ptr, None, t.ptr, K, N, 128, 0, 0, 0, 0)
                dense.append(bf16_to_f32(t.numpy(np.uint16, (K, N)))[:, cols].astype(np.float64))
                del t
            ref = xs[rows] @ dense[0]
            if dual:
                g = bf16_to_f32((ref.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16))  # (truncation is enough for a sanity check)
                u = xs[rows] @ dense[1]
                ref = (g / (1.0 + np.exp(-g.astype(np.float64)))) * u
            got = ox[np.ix_(rows, cols)].astype(np.float64)
            scale = np.abs(ref).max()
            err = np.abs(got - ref).max() / scale
            errd = np.abs(od[np.ix_(rows, cols)] - ref).max() / scale
            full = np.abs(ox.astype(np.float64) - od).max() / np.abs(od).max()
            print(f"  check {which}: |X - f64| / scale {err:.2e}   |D - f64| / scale {errd:.2e}   max |X - D| / max|D| over all outputs {full:.2e}", flush=True)
        reps = 6
        td, tq, tx = timed(run_d, reps), timed(run_q, reps), timed(run_x, reps)
        ops.check_error()
        flops = 2.0 * M * K * N * nt
        print(f"M={M:5d} {which:8s} D {td * 1e3:8.1f} us {flops / td / 1e9:7.1f} TF | X {tx * 1e3:8.1f} us {flops / tx / 1e9:7.1f} TF + dequant {tq * 1e3:6.1f} us"
              f" = {flops / (tx + tq) / 1e9:7.1f} TF", flush=True)
        tot_d += td
        tot_x += tx
        tot_q += tq
        del ws, sc, x, out_d, out_x, wd
    print(f"M={M:5d} layer: kernel D {tot_d * 1e3:.0f} us; kernel X {tot_x * 1e3:.0f} + dequant {tot_q * 1e3:.0f} = {(tot_x + tot_q) * 1e3:.0f} us", flush=True)
