"""CPU tests of the HF checkpoint reader (vllm_rs_amd/checkpoint.py): safetensors parsing incl. BF16 and
sharded indexes, config.json mapping and the rejections the reference makes, f16 -> model-dtype conversion of scales."""
import json
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from vllm_rs_amd import checkpoint as ck  # noqa: E402

CODE = {"bf16": "BF16", "f16": "F16", "f32": "F32", "i32": "I32"}


def write_safetensors(path, tensors):
    """minimal writer (format: u64 header length, JSON header, raw little-endian data); tensors: name -> (array, kind)"""
    header, blobs, off = {}, [], 0
    for name, (a, kind) in tensors.items():
        b = np.ascontiguousarray(a).tobytes()
        header[name] = {"dtype": CODE[kind], "shape": list(a.shape), "data_offsets": [off, off + len(b)]}
        blobs.append(b)
        off += len(b)
    h = json.dumps(header).encode()
    h += b" " * ((8 - len(h) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(h)) + h + b"".join(blobs))


def hf_config(**kw):
    cfg = dict(architectures=["LlamaForCausalLM"], hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
               num_key_value_heads=1, vocab_size=128, max_position_embeddings=128, rms_norm_eps=1e-5, rope_theta=10000.0,
               torch_dtype="bfloat16", tie_word_embeddings=False)
    cfg.update(kw)
    return cfg


def test_read_safetensors_round_trip(tmp_path):
    r = np.random.default_rng(0)
    t = {"a.weight": (r.integers(0, 65536, size=(4, 6)).astype(np.uint16), "bf16"),
         "b.qweight": (r.integers(-2 ** 31, 2 ** 31 - 1, size=(3, 5)).astype(np.int32), "i32"),
         "c.scales": (r.standard_normal((2, 8)).astype(np.float16).view(np.uint16), "f16"),
         "d": (r.standard_normal(7).astype(np.float32), "f32")}
    p = tmp_path / "m.safetensors"
    write_safetensors(p, t)
    got = ck.read_safetensors(str(p))
    assert set(got) == set(t)
    for k, (a, kind) in t.items():
        assert got[k][1] == kind and got[k][0].shape == a.shape and np.array_equal(got[k][0], a)
    # agrees with the safetensors package where numpy has the dtype
    from safetensors.numpy import load_file
    ref = load_file(str(p)) if False else None  # (bf16 cannot be represented in numpy; the package is exercised below)
    q = tmp_path / "n.safetensors"
    from safetensors.numpy import save_file
    save_file({"x": t["b.qweight"][0], "y": t["d"][0]}, str(q))
    g2 = ck.read_safetensors(str(q))
    assert np.array_equal(g2["x"][0], t["b.qweight"][0]) and np.array_equal(g2["y"][0], t["d"][0])


def test_sharded_index(tmp_path):
    a = (np.arange(12, dtype=np.uint16).reshape(3, 4), "bf16")
    b = (np.arange(6, dtype=np.int32).reshape(2, 3), "i32")
    write_safetensors(tmp_path / "model-00001-of-00002.safetensors", {"model.norm.weight": a})
    write_safetensors(tmp_path / "model-00002-of-00002.safetensors", {"lm_head.qweight": b})
    json.dump({"weight_map": {"model.norm.weight": "model-00001-of-00002.safetensors", "lm_head.qweight": "model-00002-of-00002.safetensors"}},
              open(tmp_path / "model.safetensors.index.json", "w"))
    got = {n: (x, k) for n, x, k in ck.iter_tensors(str(tmp_path))}
    assert np.array_equal(got["model.norm.weight"][0], a[0]) and np.array_equal(got["lm_head.qweight"][0], b[0])


def test_parse_config_and_rejections():
    c = ck.parse_config(hf_config(quantization_config=dict(quant_method="gptq", bits=4, group_size=128, desc_act=False, sym=True)))
    assert c["arch"] == "llama" and c["quant_method"] == "gptq" and c["group_size"] == 128 and c["head_dim"] == 64 and c["dtype"] == ck.BF16
    q = ck.parse_config(hf_config(architectures=["Qwen2ForCausalLM"], torch_dtype="float16", quantization_config=dict(quant_method="awq", bits=4, group_size=64, zero_point=True)))
    assert q["arch"] == "qwen2" and q["attention_bias"] and q["quant_method"] == "awq" and q["dtype"] == ck.F16 and q["group_size"] == 64
    l3 = ck.parse_config(hf_config(rope_theta=500000.0, rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)))
    assert l3["rope_theta"] == 500000.0 and l3["rope_scaling"]["rope_type"] == "llama3" and l3["quant_method"] is None
    with pytest.raises(ValueError, match="desc_act"):
        ck.parse_config(hf_config(quantization_config=dict(quant_method="gptq", bits=4, desc_act=True)))
    with pytest.raises(ValueError, match="4-bit"):
        ck.parse_config(hf_config(quantization_config=dict(quant_method="gptq", bits=8)))
    with pytest.raises(ValueError, match="architecture"):
        ck.parse_config(hf_config(architectures=["MixtralForCausalLM"]))
    # Mistral-7B-v0.1: the window goes into the forward since round 6 (vra_model_config.sliding_window; llama.rs:46,284)
    mi = ck.parse_config(hf_config(architectures=["MistralForCausalLM"], sliding_window=4096, max_position_embeddings=32768))
    assert mi["arch"] == "llama" and mi["sliding_window"] == 4096
    assert "sliding_window" not in ck.parse_config(hf_config(architectures=["MistralForCausalLM"], sliding_window=None))
    q2 = ck.parse_config(hf_config(architectures=["Qwen2ForCausalLM"], sliding_window=131072, use_sliding_window=False))
    assert q2["arch"] == "qwen2" and "sliding_window" not in q2
    assert ck.parse_config(hf_config(architectures=["Qwen3ForCausalLM"], head_dim=128))["arch"] == "qwen3"
    from vllm_rs_amd.engine import model_config
    mc = model_config(l3)
    assert mc.rope_scaling_type == 2 and mc.rope_factor == 8.0 and mc.num_kv_heads == 1


def test_float_conversion_to_model_dtype():
    r = np.random.default_rng(1)
    f16 = (r.standard_normal(4096) * 0.01).astype(np.float16)
    # f16 scales of a bf16 model: cast through f32 with round-to-nearest-even (wna16.rs:97-109), same rule as the oracle
    got = ck.to_engine_tensor("x.scales", f16.view(np.uint16), "f16", ck.BF16)
    assert np.array_equal(got, orc.to_bf16(f16.astype(np.float32)))
    assert np.array_equal(ck.to_engine_tensor("x.scales", f16.view(np.uint16), "f16", ck.F16), f16.view(np.uint16))
    bf = orc.to_bf16(r.standard_normal(64).astype(np.float32))
    assert np.array_equal(ck.to_engine_tensor("w", bf, "bf16", ck.F16), orc.to_f16(orc.from_bf16(bf)))
    assert np.array_equal(ck.f32_to_bf16_bits(np.array([np.nan, 1.00390625, 1.01171875], np.float32))[1:], orc.to_bf16(np.array([1.00390625, 1.01171875], np.float32)))
    qw = r.integers(-2 ** 31, 2 ** 31 - 1, size=(4, 4)).astype(np.int32)
    assert ck.to_engine_tensor("x.qweight", qw, "i32", ck.BF16).dtype == np.uint32


def test_load_pretrained_from_golden_fixture(tmp_path):
    """the HF golden fixture rewritten as a real checkpoint directory: names, dtypes and shapes survive; g_idx is checked and dropped"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "hf_llama_tiny.npz"))
    meta = json.loads(bytes(z["cfg_json"]).decode())
    tensors = {k[2:]: (z[k], "bf16") for k in z.files if k.startswith("w:")}
    tensors["model.layers.0.self_attn.rotary_emb.inv_freq"] = (np.zeros(32, np.float32), "f32")
    write_safetensors(tmp_path / "model.safetensors", tensors)
    json.dump(hf_config(hidden_size=meta["hidden_size"], intermediate_size=meta["intermediate_size"], num_hidden_layers=meta["num_layers"],
                        num_attention_heads=meta["num_heads"], num_key_value_heads=meta["num_kv_heads"], vocab_size=meta["vocab_size"],
                        max_position_embeddings=meta["max_position_embeddings"], rope_theta=meta["rope_theta"]), open(tmp_path / "config.json", "w"))
    cfg, gen = ck.load_pretrained(str(tmp_path))
    got = dict(gen)
    assert cfg["hidden_size"] == meta["hidden_size"] and cfg["quant_method"] is None
    assert "model.layers.0.self_attn.rotary_emb.inv_freq" not in got
    assert set(got) == {k[2:] for k in z.files if k.startswith("w:")}
    for k, a in got.items():
        assert np.array_equal(a, z["w:" + k])
