"""Compile-time lint of the product kernels (no GPU: hipcc cross-compiles gfx950 here).  The round-4 gains came from places where the
hardware was waiting for the compiler -- `vmcnt(0)` where a counted wait was meant, a store drain in front of a load, spills.  These
checks keep the repaired shapes from coming back unnoticed (tools/isa_waits.py lists a kernel's memory operations and vmcnt waits in
program order from the compiler's assembly; profiles/r04_*_waits.txt, r04_ffn_txt_counted_staging_ab.txt)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# The expectations below (wait sequences, scratch bytes) describe the output of ONE compiler: they were taken with the hipcc of
# ROCm 7.2 (HIP version 7.2.x).  On another major.minor the checks still RUN, as expected failures (xfail, not strict): a different
# schedule is then a reason to re-read the listings and re-take the expectations, not a defect of the source -- but the guard does
# not go silent: the module prints the new wait sequences of every kernel it lints (pytest -rx shows them with the reason), and a
# check that still passes is reported as XPASS.  Without hipcc there is nothing to run: skipped.  The compile flags are read from
# csrc/Makefile (tools/isa_waits.py: product_flags), so what is linted is what is shipped.
EXPECTED_HIPCC = "7.2"
LINTED = (("ffn_txt.hip", "ffn_txt_kernelILi24ELb0"), ("ffn_txt.hip", "ffn_txt_kernelILi24ELb1"), ("gemm256.hip", "gemm256_kernelILi6E"),
          ("gemm_f8.hip", "gemm256f8_kernelILi6E"), ("skinny.hip", "skinny_head_kernelILi24ELb0"), ("skinny.hip", "skinny_rows3_kernelILi24ELi0"))


def _hipcc_state():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        return "missing", "needs /opt/rocm/bin/hipcc"
    from isa_waits import hipcc_version
    v = hipcc_version()
    if v == EXPECTED_HIPCC:
        return "ok", ""
    return "other", f"ISA expectations were taken with hipcc {EXPECTED_HIPCC}, this is {v}: re-take them from the sequences printed below (tools/isa_waits.py)"


_state, _why = _hipcc_state()
if _state == "missing":
    pytestmark = pytest.mark.skip(reason=_why)
elif _state == "other":
    pytestmark = pytest.mark.xfail(reason=_why, strict=False)

# kernels that are allowed a few dwords of scratch (the residual + LayerNorm epilogue of the 256 x 256 tile at 256 VGPRs), in bytes
# (ILi6ELb1 / ILi7ELb1: the opt-in fp8-compute instantiations that also write and count the e4m3 copy of the LayerNorm output)
SCRATCH_ALLOWED = {"gemm256_kernelILi6ELb1E": 80, "gemm256_kernelILi7ELb1E": 32,
                   "gemm256f8_kernelILi6E": 40, "gemm256f8_kernelILi7E": 32}
FILES = ["attention.hip", "ffn_txt.hip", "gemm.hip", "gemm256.hip", "gemm_f8.hip", "preproc.hip", "rowops.hip",
         "skinny.hip", "student.hip", "txtblock.hip"]


@pytest.fixture(scope="module")
def listings():
    from isa_waits import kernel_listings
    out = {}
    for f in FILES:
        res, _ = kernel_listings(f, "_Z")
        assert res, f
        out[f] = res
    if _state == "other":           # another compiler: show what it emits for the linted kernels instead of going silent
        print("\n" + _why)
        for f, pat in LINTED:
            for name, vg, scratch, toks in out[f]:
                if pat in name:
                    print(f"{f}: {name}: {vg} VGPRs, {scratch} B scratch\n    " + " ".join(toks))
    return out


def test_no_product_kernel_spills(listings):
    seen = 0
    for f, res in listings.items():
        for name, vg, scratch, toks in res:
            seen += 1
            allowed = max([v for k, v in SCRATCH_ALLOWED.items() if k in name] or [0])
            assert scratch is not None and scratch <= allowed, f"{f}: {name} uses {scratch} bytes of scratch (allowed {allowed})"
            assert "xs" not in toks and "xl" not in toks or allowed, f"{f}: {name} has scratch traffic"
    assert seen >= 120


def _one(listings, f, pat):
    hits = [r for r in listings[f] if pat in r[0]]
    assert len(hits) == 1, (f, pat, [h[0] for h in hits])
    return hits[0][3]


def test_text_ffn_waits_are_counted(listings):
    """ffn_txt (GIT-base width, bf16 and e4m3 weights): no LDS-DMA in the kernel (with one the compiler stops counting), and the next
    m-tile's rows are copied to LDS with the previous tile's 12 slab stores still in flight.  Stated as a property, not as the
    compiler's exact sequence: behind every run of a tile's 12 slab stores that is followed by more work, the first wait leaves at
    least those 12 stores outstanding (vmcnt(n), n >= 12) -- never a drain (vmcnt(0))."""
    for pat in ("ffn_txt_kernelILi24ELb0", "ffn_txt_kernelILi24ELb1"):
        toks = _one(listings, "ffn_txt.hip", pat)
        assert "D" not in toks, pat
        runs = 0
        i = 0
        while i < len(toks):
            if toks[i] == "S":
                j = i
                while j < len(toks) and toks[j] == "S":
                    j += 1
                nxt = next((t for t in toks[j:] if t.startswith("W") or t == "END"), "END")
                if j - i >= 12 and nxt != "END" and any(t in ("L", "l", "S") for t in toks[j:]):      # a tile's stores, more tiles follow
                    runs += 1
                    assert nxt.startswith("W") and int(nxt[1:]) >= 12, (pat, " ".join(toks[i:j + 8]))
                i = j
            else:
                i += 1
        assert runs >= 2, (pat, " ".join(toks))                  # the pair-form loop has at least two such places


def test_gemm_ln_epilogue_keeps_its_spills_out_of_the_row_loops(listings):
    """The residual + LayerNorm epilogue of the 256 x 256 tile (pre-LN form, the one at the 256-VGPR limit): its few dwords of
    scratch must not sit in phase 1 (bias + residual add, the fp32 x stores, the segment statistics) -- a scratch reload there is
    followed by vmcnt(0) in front of every row's store (round 5: +15 % per launch when a kernarg-layout change moved them there).
    Property: no scratch traffic in the two row loops between the end of the K loop and the statistics publish (the first buffer
    store), and the x stores of both half-blocks are waited for with counted waits (16 stores, first wait vmcnt(15))."""
    for f, pat in (("gemm256.hip", "gemm256_kernelILi6E"), ("gemm_f8.hip", "gemm256f8_kernelILi6E")):
        pres = [r for r in listings[f] if pat in r[0]]
        assert 1 <= len(pres) <= 2, [r[0] for r in pres]              # the pre-LN instantiation(s) of the tile kernel (gemm256: both K-loop forms)
        for name, _, _, toks in pres:
            bs = toks.index("BS")
            bars = [i for i, t in enumerate(toks[:bs]) if t == "|"]       # ... K loop | phase 1 | BS (the statistics publish)
            phase1 = toks[bars[-2]:bs]
            loops = phase1[phase1.index("W15"):]                          # from the first counted wait on: the two row loops
            assert "xs" not in loops and "xl" not in loops, (name, " ".join(phase1))
            s = " ".join(phase1)
            assert s.count("W15 S W14 S W13 S") == 2, (name, s)


def test_vocabulary_head_requests_rows_ahead_of_weights(listings):
    """skinny_head: the activation rows are requested first and waited for alone (counted), the weights stay in flight across the
    first barrier; nothing in front of that barrier waits for everything."""
    toks = _one(listings, "skinny.hip", "skinny_head_kernelILi24ELb0")
    first_bar = toks.index("|")
    head = toks[:first_bar]
    assert head.count("L") >= 30 and "W0" not in head, head
    assert any(t.startswith("W") and int(t[1:]) >= 24 for t in head), head


def test_row_prologue_three_waves_is_one_round_trip(listings):
    """skinny_rows3: the row's two slab groups per wave (2 x 8 slabs x 3 vectors) + bias / residual / gamma / beta are all requested
    before the single wait of the row loop."""
    toks = _one(listings, "skinny.hip", "skinny_rows3_kernelILi24ELi0")
    i = toks.index("[")
    j = toks.index("W0", i)
    assert toks[i + 1:j].count("L") >= 48 + 12, toks[i:j + 1]
