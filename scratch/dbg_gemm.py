import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as orc
from tests.util import *
from vllm_rs_amd import ops
M, K, N = int(sys.argv[1]), 512, 256
r = rng(1)
q = make_quant(r, K, N, 128, BF16, False)
x = rand_dt(r, (M, K), BF16)
tiled = ops.marlin_weight_repack(ops.dev(q["qweight"]), q["qweight"].shape)
out = ops.wna16_gemm(ops.dev(x), tiled, ops.dev(q["scales"]), None, M, K, N, 128)
got = orc.from_bf16(out.numpy(np.uint16, (M, N)))
ref = orc.from_bf16(orc.wna16_gemm(x, q["idx"], None, q["scales"], 128, BF16))
bad = np.abs(got - ref) > 0.05
print("bad per row:", bad.sum(1))
print("bad per 16-col block:", bad.reshape(M, N // 16, 16).sum((0, 2)))
# does got row m equal ref row m' for some other m'?
for m in range(min(M, 34)):
    d = np.abs(ref - got[m][None, :]).max(1)
    print(m, int(d.argmin()), float(d.min()))
