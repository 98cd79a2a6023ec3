"""Host-side logic that needs no GPU: weight name maps, config struct layout, sharding."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from gitcap.config import CGitCapConfig, git_base, git_large, git_tiny
from gitcap import weights as W
from gitcap.dist import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_configs_validate():
    for c in (git_base(6), git_base(0), git_large(6), git_tiny(2)):
        c.validate()
    assert git_base().tokens_per_frame == 197 and git_large().tokens_per_frame == 257
    assert git_base().image_tokens(6) == 1182 and git_large().image_tokens(6) == 1542     # model.py:90


def test_synthetic_weights_are_deterministic_and_complete():
    cfg = git_tiny(2)
    a, b = W.synthetic_weights(cfg, 0), W.synthetic_weights(cfg, 0)
    assert list(a) == list(W.canonical_shapes(cfg))
    for k in a:
        assert a[k].dtype == np.float32 and np.array_equal(a[k], b[k])
    assert not np.array_equal(a["head.w"], W.synthetic_weights(cfg, 1)["head.w"])
    W.check_shapes(cfg, a)


def test_hf_name_map_roundtrip():
    cfg = git_tiny(2)
    w = W.synthetic_weights(cfg, 3)
    back = W.from_hf_state_dict(cfg, W.to_hf_state_dict(cfg, w))
    for k in w:
        assert np.array_equal(w[k], back[k]), k


def test_ms_checkpoint_layout_import():
    """Build a dict with the key layout of the MS GenerativeImage2Text checkpoint the reference
    loads (model.py:736-738: image_encoder.transformer.resblocks.N.*, textual.transformer.encoder.layer.N.*,
    img_temperal_embedding.N) and check it lands on the canonical tensors."""
    cfg = git_tiny(2)
    w = W.synthetic_weights(cfg, 4)
    Dv, D, p = cfg.enc_width, cfg.dec_width, cfg.patch_size
    sd = {"image_encoder.conv1.weight": w["enc.patch_w"].reshape(Dv, 3, p, p),
          "image_encoder.class_embedding": w["enc.cls"], "image_encoder.positional_embedding": w["enc.pos"],
          "image_encoder.ln_pre.weight": w["enc.ln_pre.w"], "image_encoder.ln_pre.bias": w["enc.ln_pre.b"],
          "image_encoder.ln_post.weight": w["enc.ln_post.w"], "image_encoder.ln_post.bias": w["enc.ln_post.b"]}
    for i in range(cfg.enc_layers):
        s, q = f"image_encoder.transformer.resblocks.{i}.", f"enc.L{i}."
        sd.update({s + "ln_1.weight": w[q + "ln1.w"], s + "ln_1.bias": w[q + "ln1.b"],
                   s + "attn.in_proj_weight": w[q + "qkv.w"], s + "attn.in_proj_bias": w[q + "qkv.b"],
                   s + "attn.out_proj.weight": w[q + "proj.w"], s + "attn.out_proj.bias": w[q + "proj.b"],
                   s + "ln_2.weight": w[q + "ln2.w"], s + "ln_2.bias": w[q + "ln2.b"],
                   s + "mlp.c_fc.weight": w[q + "fc1.w"], s + "mlp.c_fc.bias": w[q + "fc1.b"],
                   s + "mlp.c_proj.weight": w[q + "fc2.w"], s + "mlp.c_proj.bias": w[q + "fc2.b"]})
    for f in range(cfg.num_frames):
        sd[f"img_temperal_embedding.{f}"] = w["temporal"][f].reshape(1, 1, Dv)
    t = "textual."
    sd.update({t + "visual_projection.0.weight": w["vproj.w"], t + "visual_projection.0.bias": w["vproj.b"],
               t + "visual_projection.1.weight": w["vproj.ln.w"], t + "visual_projection.1.bias": w["vproj.ln.b"],
               t + "embedding.words.weight": w["txt.word"], t + "embedding.positions.weight": w["txt.pos"],
               t + "embedding.layer_norm.weight": w["txt.ln.w"], t + "embedding.layer_norm.bias": w["txt.ln.b"],
               t + "output.weight": w["head.w"], t + "output.bias": w["head.b"]})
    for i in range(cfg.dec_layers):
        s, q = t + f"transformer.encoder.layer.{i}.", f"dec.L{i}."
        for j, n in enumerate(("query", "key", "value")):
            sd[s + f"attention.self.{n}.weight"] = w[q + "qkv.w"][j * D:(j + 1) * D]
            sd[s + f"attention.self.{n}.bias"] = w[q + "qkv.b"][j * D:(j + 1) * D]
        sd.update({s + "attention.output.dense.weight": w[q + "ao.w"], s + "attention.output.dense.bias": w[q + "ao.b"],
                   s + "attention.output.LayerNorm.weight": w[q + "ln1.w"], s + "attention.output.LayerNorm.bias": w[q + "ln1.b"],
                   s + "intermediate.dense.weight": w[q + "fc1.w"], s + "intermediate.dense.bias": w[q + "fc1.b"],
                   s + "output.dense.weight": w[q + "fc2.w"], s + "output.dense.bias": w[q + "fc2.b"],
                   s + "output.LayerNorm.weight": w[q + "ln2.w"], s + "output.LayerNorm.bias": w[q + "ln2.b"]})
    back = W.from_ms_state_dict(cfg, sd)
    for k in w:
        assert np.array_equal(w[k], back[k]), k


def test_hf_import_accepts_legacy_temporal_key_and_refuses_a_missing_one():
    """transformers 4.x (and the video checkpoints it wrote) spell the list `img_temperal_embedding`; 5.x
    `img_temporal_embedding`.  Both must import; a frame with neither is an error, not a row of zeros."""
    cfg = git_tiny(2)
    w = W.synthetic_weights(cfg, 5)
    sd = W.to_hf_state_dict(cfg, w)
    legacy = {k.replace("img_temporal_embedding", "img_temperal_embedding"): v for k, v in sd.items()}
    assert any("img_temperal_embedding" in k for k in legacy) and not any("img_temporal_embedding" in k for k in legacy)
    back = W.from_hf_state_dict(cfg, legacy)
    assert np.array_equal(back["temporal"], w["temporal"]) and np.abs(w["temporal"]).max() > 0
    broken = {k: v for k, v in sd.items() if not k.endswith("img_temporal_embedding.1")}
    with pytest.raises(KeyError):
        W.from_hf_state_dict(cfg, broken)


def test_check_shapes_rejects_wrong_and_missing():
    cfg = git_tiny(2)
    w = W.synthetic_weights(cfg, 0)
    bad = dict(w)
    bad["head.w"] = bad["head.w"][:-1]
    with pytest.raises(ValueError):
        W.check_shapes(cfg, bad)
    del bad["head.w"]
    with pytest.raises(KeyError):
        W.check_shapes(cfg, bad)


def test_ctypes_struct_matches_c_header(tmp_path):
    """Compile a probe against include/gitcap.h with gcc and compare sizeof/offsetof."""
    src = tmp_path / "probe.c"
    fields = [n for n, _ in CGitCapConfig._fields_]
    body = "".join(f'printf("{n} %zu\\n", offsetof(gitcap_config, {n}));' for n in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gitcap.h"\n'
                   'int main(){printf("size %zu\\n", sizeof(gitcap_config));' + body + "return 0;}")
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    assert int(out["size"]) == ctypes.sizeof(CGitCapConfig)
    for n in fields:
        assert int(out[n]) == getattr(CGitCapConfig, n).offset, n


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 16, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(128, 3, 8) == (48, 64)            # configs[3]: 16 clips per GPU
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def test_fp8_weight_quantisation_is_exact_in_bf16_and_close():
    import torch
    cfg = git_tiny(2)
    w = W.synthetic_weights(cfg, 0)
    q = W.quantize_weights_fp8(w)
    for k in w:
        if not W.is_gemm_weight(k):
            assert q[k] is w[k]
            continue
        t = torch.from_numpy(q[k])
        assert torch.equal(t.bfloat16().float(), t), k                  # representable in bf16 exactly
        rel = (t - torch.from_numpy(w[k])).abs() / torch.from_numpy(w[k]).abs().amax(dim=1, keepdim=True)
        assert float(rel.max()) < 2 ** -4, k                            # e4m3: 3 mantissa bits
        # idempotent: quantising again changes nothing
    q2 = W.quantize_weights_fp8(q)
    assert all(np.array_equal(q[k], q2[k]) for k in q)


def test_host_copy_is_a_memcpy_for_every_size():
    """gitcap_host_copy (the staging copy of host-fed submissions; no device work): plain memcpy semantics at every size -- below the
    4 MiB per-thread piece (one thread), across piece boundaries and with a ragged tail -- and argument checks."""
    import ctypes
    import torch
    from gitcap import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    for n in (0, 1, 4095, 4096, (4 << 20) - 1, (4 << 20) + 1, 9 * (1 << 20) + 123, 40 * (1 << 20) + 7):
        src = torch.randint(0, 256, (max(n, 1) + 64,), dtype=torch.uint8, generator=g)
        dst = torch.zeros_like(src)
        assert lib.gitcap_host_copy(ctypes.c_void_p(dst.data_ptr() + 32), ctypes.c_void_p(src.data_ptr() + 32), n) == 0
        assert torch.equal(dst[32:32 + n], src[32:32 + n]), n
        assert int(dst[:32].sum()) == 0 and int(dst[32 + n:].sum()) == 0, n       # nothing outside [32, 32 + n)
    assert lib.gitcap_host_copy(None, None, 0) == 0
    assert lib.gitcap_host_copy(None, ctypes.c_void_p(src.data_ptr()), 8) == -1
    assert lib.gitcap_host_copy(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), -1) == -1
