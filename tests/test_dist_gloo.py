"""world_size-2 `gloo` tests on CPU of the N>1 paths (no GPU):

* bench.py's multi-rank contract: independent replicas per rank (weak scaling, no data-path collective), barrier, the
  job's time = MAX over ranks, value = units of all ranks / that time
* the tensor-parallel sharding rules of SURVEY.md §8(e) (wna16.rs:35-40,127-152; distributed.rs:325-396,438-455):
  column-parallel q/k/v/gate/up shard N, row-parallel o/down shard K in units of the quantisation group, ONE
  all-reduce(sum) in the storage dtype after each row-parallel GEMM — restated with the oracle on two ranks and
  compared with the unsharded oracle result.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BF16 = 0


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("OMP_NUM_THREADS", "2")


def _bench_worker(rank, world, port, q):
    _init(rank, world, port)
    import bench
    d = bench.NodeRendezvous(rank, world)   # key = MASTER_PORT, as under torch.distributed.run
    d.barrier()
    local_s = 0.5 + 0.25 * rank            # rank 1 is the slow one
    t = bench.max_over_ranks(d, local_s)
    batch, steps = 4, 10
    value = batch * steps * world / t      # whole-job aggregate, as bench.py's `value`
    d.barrier()
    if rank == 0:
        q.put((t, value))
    d.close()


def test_bench_aggregation_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, 29611, q)) for r in range(2)]
    [p.start() for p in procs]
    t, value = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert t == pytest.approx(0.75) and value == pytest.approx(4 * 10 * 2 / 0.75)


def test_single_rank_needs_no_process_group():
    import bench
    assert bench.max_over_ranks(None, 1.25) == 1.25 and bench.NodeRendezvous(0, 1).max(2.5) == 2.5


def test_bench_gpus_n_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts two ranks of itself (RANK / LOCAL_RANK / WORLD_SIZE + a private
    rendezvous key); here, without GPUs, each rank stops at the device check — after the spawn, with its rank in the message"""
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "VRA_BENCH_RDZV"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert "no HIP device for rank 0" in p.stderr and "no HIP device for rank 1" in p.stderr, p.stderr[-2000:]
    # a launcher that started another number of ranks than --gpus asks for is an error, not a silently different line
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-extras"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert p.returncode != 0 and "--gpus 4 but the launcher started 2" in p.stderr


def test_bench_gpus_8_spawns_eight_ranks_that_rendezvous():
    """the 8-GPU run must work first time.  `python bench.py --gpus 8` starts eight ranks of itself; they meet on the
    node-local rendezvous socket BEFORE anything touches a device (rank 0 says so), then — here, without GPUs — every rank stops at
    the device check with its own rank in the message"""
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "VRA_BENCH_RDZV"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode != 0
    assert "rendezvous of 8 ranks complete" in p.stderr, p.stderr[-2000:]
    for r in range(8):
        assert f"no HIP device for rank {r}" in p.stderr, p.stderr[-2000:]


def test_bench_aggregation_eight_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 8, 29613, q)) for r in range(8)]
    [p.start() for p in procs]
    t, value = q.get(timeout=300)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert t == pytest.approx(0.5 + 0.25 * 7) and value == pytest.approx(4 * 10 * 8 / (0.5 + 0.25 * 7))


def test_bench_tp8_fails_loudly_without_gpus_and_counts_a_rank_s_bytes():
    """`bench.py --tp 8 --model llama3-70b` has no CPU fallback; the per-rank roofline bytes of its JSON line follow SURVEY §8(d)"""
    import subprocess
    import bench
    from vllm_rs_amd import engine as E
    b = bench.tp_rank_algorithmic_bytes(dict(E.LLAMA3_70B), 8, 1)
    layer_w = (8192 * 1024 + 2 * 8192 * 128 + 1024 * 8192 + 3 * 8192 * 3584) // 2  # packed int4 bytes of one rank's layer
    assert 80 * layer_w < b - 128256 * 8192 * 2 < 80 * layer_w * 1.05  # + scales / zeros / activations: a few percent
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tp", "8", "--model", "llama3-70b", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "needs a GPU" in (p.stderr + p.stdout), (p.stderr + p.stdout)[-1500:]


def _shard_cols(q, rank, world):
    """column parallel: packed tensors are stored [in, out] so the OUTPUT dim is dim 1 (wna16.rs:35-40);
    qzeros in units of 8 columns."""
    N = q["scales"].shape[1]
    n0, n1 = rank * N // world, (rank + 1) * N // world
    return dict(idx=q["idx"][:, n0:n1], scales=q["scales"][:, n0:n1], zeros=None if q["zeros"] is None else q["zeros"][:, n0:n1]), (n0, n1)


def _shard_rows(q, rank, world, g):
    """row parallel: shard K (dim 0 of qweight in units of 8 rows, scales/qzeros rows in units of the group)."""
    K = q["idx"].shape[0]
    k0, k1 = rank * K // world, (rank + 1) * K // world
    assert (K // world) % g == 0, "K/world must be a multiple of the group size (SURVEY §8e)"
    return dict(idx=q["idx"][k0:k1], scales=q["scales"][k0 // g:k1 // g], zeros=None if q["zeros"] is None else q["zeros"][k0 // g:k1 // g]), (k0, k1)


def _tp_worker(rank, world, port, q):
    _init(rank, world, port)
    from oracle import oracle as orc
    dist.init_process_group("gloo")
    r = np.random.default_rng(5)            # same stream on every rank: identical full tensors
    H, I, M, g = 256, 512, 3, 128
    mk = lambda K, N: dict(idx=r.integers(0, 16, size=(K, N), dtype=np.uint8), zeros=None,
                           scales=orc.to_bf16((0.002 + 0.01 * r.random((K // g, N))).astype(np.float32)))
    gate, up, down = mk(H, I), mk(H, I), mk(I, H)
    x = orc.to_bf16(r.standard_normal((M, H)).astype(np.float32))
    res = orc.to_bf16(r.standard_normal((M, H)).astype(np.float32))
    # ---- unsharded reference (mlp.rs:451-469)
    act = orc.silu_mul(orc.wna16_gemm(x, gate["idx"], None, gate["scales"], g, BF16), orc.wna16_gemm(x, up["idx"], None, up["scales"], g, BF16), BF16)
    full = orc.add(orc.wna16_gemm(act, down["idx"], None, down["scales"], g, BF16), res, BF16)
    # ---- this rank's shard
    gs, (n0, n1) = _shard_cols(gate, rank, world)
    us, _ = _shard_cols(up, rank, world)
    ds, (k0, k1) = _shard_rows(down, rank, world, g)
    assert (n0, n1) == (k0, k1)             # the column shard of gate/up feeds the row shard of down directly
    a = orc.silu_mul(orc.wna16_gemm(x, gs["idx"], None, gs["scales"], g, BF16), orc.wna16_gemm(x, us["idx"], None, us["scales"], g, BF16), BF16)
    assert (a == act[:, n0:n1]).all()        # column-parallel outputs are exactly the slices of the full result
    part = orc.wna16_gemm(a, ds["idx"], None, ds["scales"], g, BF16)         # partial sum, rounded to bf16 per rank
    t = torch.from_numpy(orc.from_bf16(part).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)                                  # AllReduce (distributed.rs:438-455)
    out = orc.add(orc.to_bf16(t.numpy()), res, BF16)                          # residual after the reduction
    d = np.abs(orc.from_bf16(out) - orc.from_bf16(full))
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(orc.from_bf16(full)), 1e-30))) - 7)
    if rank == 0:
        # bf16 partial sums: the order of roundings differs from the single-GPU result by at most ~2 ulp
        q.put((float((d / np.maximum(ulp, 2.0 ** -9)).max()), float((d == 0).mean())))
    dist.barrier()
    dist.destroy_process_group()


def test_tp2_sharding_rules_with_gloo_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, 29612, q)) for r in range(2)]
    [p.start() for p in procs]
    worst, exact = q.get(timeout=180)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert worst <= 2.0 and exact > 0.5, (worst, exact)
