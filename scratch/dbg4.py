import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as orc
from tests.util import *
from vllm_rs_amd import ops
M, K, N = int(sys.argv[1]), 512, 256
r = rng(1)
q = make_quant(r, K, N, 128, BF16, False)
tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape)
x = orc.to_bf16(np.ones((M, K), np.float32))
ref = orc.from_bf16(orc.wna16_gemm(x, q["idx"], None, q["scales"], 128, BF16))
for it in range(4):
    out = ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), None, M, K, N, 128)
    got = orc.from_bf16(out.numpy(np.uint16, (M, N)))
    bad = np.abs(got - ref) > 0.02 * np.abs(ref).max()
    blocks = np.nonzero(bad.reshape(M, N // 16, 16).sum((0, 2)))[0]
    print("iter", it, "bad blocks", blocks)
    for b in blocks[:2]:
        sub = bad[:, b * 16:(b + 1) * 16]
        print("block", b)
        for m in range(M):
            print("".join("X" if v else "." for v in sub[m]), " got", got[m, b*16:(b*16+4)], "ref", ref[m, b*16:(b*16+4)])
