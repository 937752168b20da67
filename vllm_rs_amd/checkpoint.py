"""HuggingFace checkpoint ingestion for the native runtime: `config.json` → model config,
`*.safetensors` (single file or sharded `model.safetensors.index.json`, src/utils/mod.rs:90-111) → tensors under their
HF names, handed to `vra_engine_load_tensor` (which does the TP slicing and the int4 repack on the device).

Mirrors what the reference reads: `Config` (src/utils/config.rs:218-255), `QuantConfig` (:735-757; only 4-bit,
`desc_act = false` is accepted — src/utils/mod.rs:1316-1318), and the tensors of `WNA16::new`
(src/models/layers/wna16.rs:56-152: qweight / qzeros / scales / g_idx / bias; scales and bias are stored f16 and
cast to the model dtype, :97-109).  Pure host code: numpy + a 30-line safetensors reader (mmap, zero copy); no torch.
"""
import json
import mmap
import os
import struct

import numpy as np

BF16, F16, F32 = 0, 1, 2
_ST_DTYPES = {"BF16": (np.uint16, "bf16"), "F16": (np.uint16, "f16"), "F32": (np.float32, "f32"), "I32": (np.int32, "i32"),
              "U32": (np.uint32, "u32"), "I64": (np.int64, "i64"), "U8": (np.uint8, "u8"), "I8": (np.int8, "i8"),
              "U16": (np.uint16, "u16"), "I16": (np.int16, "i16"), "BOOL": (np.uint8, "bool"), "F64": (np.float64, "f64")}
ARCHS = {"LlamaForCausalLM": "llama", "MistralForCausalLM": "llama", "Qwen2ForCausalLM": "qwen2", "Qwen3ForCausalLM": "qwen3"}


def f32_to_bf16_bits(x):
    """round-to-nearest-even, NaN quieted — same rule as the kernels (common.cuh BF16::from_f32)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return np.where(nan, ((u >> 16) | 0x40).astype(np.uint16), r)


def read_safetensors(path):
    """-> dict name -> (numpy array view into an mmap, kind) with kind in {'bf16','f16','f32','i32',...}; 16-bit floats
    come back as uint16 bit patterns."""
    f = open(path, "rb")
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    (hlen,) = struct.unpack("<Q", mm[:8])
    header = json.loads(mm[8:8 + hlen].decode("utf-8"))
    out = {}
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        if meta["dtype"] not in _ST_DTYPES:
            raise ValueError(f"{path}: tensor {name} has unsupported dtype {meta['dtype']}")
        npdt, kind = _ST_DTYPES[meta["dtype"]]
        b0, b1 = meta["data_offsets"]
        a = np.frombuffer(mm, dtype=npdt, count=(b1 - b0) // np.dtype(npdt).itemsize, offset=8 + hlen + b0)
        out[name] = (a.reshape(meta["shape"]), kind)
    return out


def parse_config(cfg_json, dtype=None):
    """HF config.json dict -> the dict `vllm_rs_amd.engine.model_config` consumes; raises on what the reference
    rejects (src/utils/mod.rs:1303-1330)."""
    archs = cfg_json.get("architectures") or ["LlamaForCausalLM"]
    if archs[0] not in ARCHS:
        raise ValueError(f"architecture {archs[0]} is not on this path (supported: {sorted(ARCHS)})")
    td = dtype or cfg_json.get("torch_dtype", "bfloat16")
    dt = {"bfloat16": BF16, "bf16": BF16, "float16": F16, "half": F16, "f16": F16}.get(td if isinstance(td, str) else "", None)
    if dt is None:
        dt = td if td in (BF16, F16) else BF16
    heads = cfg_json["num_attention_heads"]
    out = dict(arch=ARCHS[archs[0]], hidden_size=cfg_json["hidden_size"], intermediate_size=cfg_json["intermediate_size"],
               num_layers=cfg_json["num_hidden_layers"], num_heads=heads, num_kv_heads=cfg_json.get("num_key_value_heads", heads),
               head_dim=cfg_json.get("head_dim") or cfg_json["hidden_size"] // heads, vocab_size=cfg_json["vocab_size"],
               max_position_embeddings=cfg_json.get("max_position_embeddings", 4096), rms_norm_eps=cfg_json.get("rms_norm_eps", 1e-5),
               rope_theta=float(cfg_json.get("rope_theta", (cfg_json.get("rope_parameters") or {}).get("rope_theta", 10000.0))),
               rope_scaling=cfg_json.get("rope_scaling"), tie_word_embeddings=bool(cfg_json.get("tie_word_embeddings", False)),
               attention_bias=bool(cfg_json.get("attention_bias", ARCHS[archs[0]] == "qwen2")), dtype=dt, quant_method=None)
    # Sliding-window attention (Mistral-7B-v0.1: 4096; llama.rs:46,284 passes config.sliding_window into attention and mask): wired into
    # the engine's forward since round 6 (vra_model_config.sliding_window).  A window that covers every position is full causal attention.
    sw = cfg_json.get("sliding_window")
    if sw and cfg_json.get("use_sliding_window", True) and int(sw) < int(out["max_position_embeddings"]):
        out["sliding_window"] = int(sw)
    q = cfg_json.get("quantization_config")
    if q:
        method = q.get("quant_method", "").lower()
        if method not in ("gptq", "awq"):
            raise ValueError(f"quant_method {method!r} is not on this path (gptq, awq)")
        if q.get("bits", 4) != 4:
            raise ValueError("only 4-bit GPTQ/AWQ checkpoints are supported (wna16.rs:154-160)")
        if q.get("desc_act", False):
            raise ValueError("desc_act=true checkpoints are rejected, as in the reference (src/utils/mod.rs:1316-1318)")
        if method == "gptq" and q.get("sym", True) is False:
            raise ValueError("asymmetric GPTQ goes through gemm_half_q_half_alt in the reference (f16 only); not wired into the engine")
        out.update(quant_method=method, group_size=q.get("group_size", 128))
    return out


def iter_tensors(model_dir):
    """yields (hf_name, array, kind) over every tensor of the checkpoint (sharded or not)."""
    idx = os.path.join(model_dir, "model.safetensors.index.json")
    if os.path.exists(idx):
        files = sorted(set(json.load(open(idx))["weight_map"].values()))
    else:
        files = sorted(f for f in os.listdir(model_dir) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no .safetensors files in {model_dir}")
    for fn in files:
        for name, (a, kind) in read_safetensors(os.path.join(model_dir, fn)).items():
            yield name, a, kind


def to_engine_tensor(name, a, kind, model_dtype):
    """checkpoint tensor -> what vra_engine_load_tensor expects: packed int tensors as they are, every float tensor
    as 16-bit patterns of the MODEL dtype (scales/bias are f16 on disk even for bf16 models: wna16.rs:97-109)."""
    if kind in ("i32", "u32"):
        return np.ascontiguousarray(a).view(np.uint32)
    want = "bf16" if model_dtype == BF16 else "f16"
    if kind == want:
        return np.ascontiguousarray(a)
    if kind == "f16":
        f = a.view(np.float16).astype(np.float32)
    elif kind == "bf16":
        f = (a.astype(np.uint32) << 16).view(np.float32)
    elif kind == "f32":
        f = np.asarray(a, np.float32)
    else:
        raise ValueError(f"{name}: cannot convert {kind} to the model dtype")
    return f32_to_bf16_bits(f) if model_dtype == BF16 else f.astype(np.float16).view(np.uint16)


def load_pretrained(model_dir, dtype=None):
    """-> (cfg dict, generator of (name, engine-ready array)); rotary inv_freq buffers and unused tensors are skipped
    by the engine itself (unknown names are ignored there only if they end in `rotary_emb.inv_freq`)."""
    cfg = parse_config(json.load(open(os.path.join(model_dir, "config.json"))), dtype)

    def gen():
        for name, a, kind in iter_tensors(model_dir):
            if name.endswith("rotary_emb.inv_freq"):
                continue
            if name.endswith(".g_idx"):
                g = cfg.get("group_size", 128)
                K = a.shape[0]
                gg = g if g and g > 0 else K
                if not np.array_equal(np.asarray(a, np.int64), np.arange(K) // gg):
                    raise ValueError(f"{name}: non-trivial g_idx (act-order) is not supported (SURVEY Appendix A7)")
                continue
            yield name, to_engine_tensor(name, a, kind, cfg["dtype"])
    return cfg, gen()
