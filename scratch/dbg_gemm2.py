import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as orc
from tests.util import *
from vllm_rs_amd import ops
M, K, N = int(sys.argv[1]), 512, 256
r = rng(1)
q = make_quant(r, K, N, 128, BF16, False)
tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape)
def run(xf, tag):
    x = orc.to_bf16(xf.astype(np.float32))
    out = ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), None, M, K, N, 128)
    got = orc.from_bf16(out.numpy(np.uint16, (M, N)))
    ref = orc.from_bf16(orc.wna16_gemm(x, q["idx"], None, q["scales"], 128, BF16))
    bad = np.abs(got - ref) > 0.02 * np.abs(ref).max()
    print(tag, "bad:", int(bad.sum()), "per block:", bad.reshape(M, N // 16, 16).sum((0, 2)))
    return got, ref, bad
run(np.ones((M, K)), "ones")
run(np.tile((np.arange(M)[:, None] + 1) * 0.125, (1, K)), "row-const")
run(np.tile(((np.arange(K) % 8) + 1)[None, :] * 0.25, (M, 1)), "k-mod8")
run(np.tile(((np.arange(K) // 8 % 4) + 1)[None, :] * 0.25, (M, 1)), "oct-mod4")
run(np.tile(((np.arange(K) // 32) + 1)[None, :] * 0.25, (M, 1)), "k32 steps")
got, ref, bad = run(r.standard_normal((M, K)), "random")
print(np.argwhere(bad)[:40].T)
got2, _, bad2 = run(r.standard_normal((M, K)), "random-again")
