"""Stage-by-stage comparison of the device with the bf16-emulating oracle (oracle/git_oracle.py), two ways:

  * SHARED INPUTS: stage s of the oracle is fed the DEVICE's own input of stage s (the ViT's residual stream per block through
    gitcap_dbg_enc_tap, the decoder's per-layer hidden states through gitcap_hidden_states_*), so whatever the two computations
    of ONE stage disagree by is what is measured -- conditioning cannot amplify anything across a single stage;
  * END TO END: stage s of the oracle's own chain against the device's output of stage s (everything upstream included).

Used by tests/test_stress_layers_gpu.py (fixed tolerances on the shared-input numbers) and tools/stress_divergence.py (the table
profiles/r06_stress_divergence.md).  Reference path being localised: src/models/model.py:378 (image encoder), :412-418 (decoder).

Errors are stated in bf16 ulps OF THE ROW MAXIMUM: ulp(row) = 2^(floor(log2 max|ref row|) - 7), the spacing of bf16 numbers at the
row's largest magnitude -- the natural unit for a pipeline whose every GEMM operand is rounded to bf16 (a residual stream carrying
an outlier channel of 60 has ulp 0.25 whatever the other 767 channels hold)."""
import ctypes

import torch


def ulps_of_rowmax(dev: torch.Tensor, ref: torch.Tensor):
    """-> (max error in ulps of the row maximum, rms error in the same unit, max |row| seen, max abs error)."""
    d = (dev.double() - ref.double()).reshape(-1, ref.shape[-1])
    r = ref.double().reshape(-1, ref.shape[-1])
    rowmax = r.abs().amax(dim=1).clamp_min(2.0 ** -126)
    ulp = torch.exp2(torch.floor(torch.log2(rowmax)) - 7.0)
    e = d.abs() / ulp[:, None]
    return float(e.max()), float(e.pow(2).mean().sqrt()), float(rowmax.max()), float(d.abs().max())


def device_stages(m, fr: torch.Tensor, ids: torch.Tensor):
    """One synchronous pass of the device with every tap open -> dict of CPU tensors:
    enc [L][nf][N][Dv] (residual stream entering each block), visual [B][F*N][Dv], hidden [B][Ld+1][S_img+T][D], logits [B][T][V]."""
    cfg = m.cfg
    B, F = fr.shape[:2]
    N, Dv, L = cfg.tokens_per_frame, cfg.enc_width, cfg.enc_layers
    rows = B * F * N
    tap = torch.full((L, rows, Dv), float("nan"), dtype=torch.float32, device=m._dev)
    m._call("gitcap_dbg_enc_tap", ctypes.c_void_p(tap.data_ptr()))
    try:
        logits, vis, hid = m.forward_output_logits(fr, ids, output_hidden_states=True)
        torch.cuda.synchronize()
    finally:
        m._call("gitcap_dbg_enc_tap", None)
    return {"enc": tap.view(L, B * F, N, Dv).cpu(), "visual": torch.cat(vis, 0).cpu(),
            "hidden": torch.stack([h.cpu() for h in hid], 0), "logits": torch.cat(logits, 0).cpu()}


def stage_table(cfg, dev: dict, orc, fr: torch.Tensor, ids: torch.Tensor, img_only_decoder: bool = False):
    """Rows (stage, shared-input stats, end-to-end stats).  `orc`: a GitOracle with the device's rounding points.
    img_only_decoder: compare the decoder's image rows only (compute modes that treat image and text rows differently)."""
    B, F = fr.shape[:2]
    L, Ld = cfg.enc_layers, cfg.dec_layers
    S_img = F * cfg.tokens_per_frame
    out = []
    with torch.no_grad():
        # ---- the oracle's own chain (end to end) ----
        taps = []
        v_o = orc.encode_frames(fr, taps=taps)
        mem_o = orc.project(v_o)
        _, hid_o = orc.decoder_full(mem_o, ids, return_hidden=True)
        logits_o = orc._lin(hid_o[-1][:, S_img:], "head")
        enc_d, hid_d = dev["enc"], dev["hidden"]
        sel = (lambda t: t[:, :S_img]) if img_only_decoder else (lambda t: t)

        def add(stage, shared, e2e_dev, e2e_ref):
            out.append({"stage": stage, "shared": ulps_of_rowmax(*shared) if shared is not None else None,
                        "e2e": ulps_of_rowmax(e2e_dev, e2e_ref)})
        add("patch embed + ln_pre", (enc_d[0], orc.embed_frames(fr)), enc_d[0], taps[0])
        for i in range(L - 1):
            add(f"enc block {i}", (enc_d[i + 1], orc.enc_block(i, enc_d[i])), enc_d[i + 1], taps[i + 1])
        add(f"enc block {L - 1} + ln_post", (dev["visual"], orc.enc_post(orc.enc_block(L - 1, enc_d[L - 1]), B, F)), dev["visual"], v_o)
        add("visual projection", (hid_d[:, 0, :S_img], orc.project(dev["visual"])), hid_d[:, 0, :S_img], mem_o)
        if not img_only_decoder:
            add("text embedding", (hid_d[:, 0, S_img:], orc.embed_text(ids)), hid_d[:, 0, S_img:], hid_o[0][:, S_img:])
        for l in range(Ld):
            x_in = hid_d[:, l]
            ref = orc.dec_layer_img(l, x_in[:, :S_img]) if img_only_decoder else orc.dec_layer_full(l, x_in, S_img)
            add(f"dec layer {l}", (sel(hid_d[:, l + 1]), ref), sel(hid_d[:, l + 1]), sel(hid_o[l + 1]))
        if not img_only_decoder:
            add("vocabulary head", (dev["logits"], orc._lin(hid_d[:, Ld, S_img:], "head")), dev["logits"], logits_o)
    return out


def format_table(title: str, rows: list) -> str:
    lines = [f"### {title}", "",
             "| stage | shared inputs: max (ulp of row max) | rms (ulp) | max abs | end to end: max (ulp) | rms (ulp) | max abs | largest row max |",
             "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        s, e = r["shared"], r["e2e"]
        lines.append(f"| {r['stage']} | {s[0]:.3f} | {s[1]:.4f} | {s[3]:.4g} | {e[0]:.3f} | {e[1]:.4f} | {e[3]:.4g} | {e[2]:.1f} |")
    return "\n".join(lines) + "\n"
