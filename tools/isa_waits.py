#!/usr/bin/env python3
"""Memory operations and vmcnt waits of a kernel, in program order, from the compiler's assembly (no GPU needed).

    python tools/isa_waits.py gemm256.hip 'gemm256_kernelILi6E'        # GEMM + residual + LayerNorm epilogue
    python tools/isa_waits.py txtblock.hip 'txt_block_kernelILi24ELb0ELi16ELb1'

Prints one token per instruction: L = global load (x4), l = narrower global load, D = global_load_lds (LDS-DMA), S / s / d =
global store x4 / x2 / x1, BL / BS = buffer load / store, Wn = s_waitcnt vmcnt(n), | = s_barrier, xs / xl = scratch (spill)
store / load, [ ] = a loop.  What to look for (profiles/r04_text_attention_phase_stamps.txt, r04_gemm_ln_epilogue_waits.txt):
"L W0 S L W0 S" (every load waits for the previous store: on gfx9 vmcnt counts stores), W0 right behind a counted wait written
in the source (LDS-DMA + plain loads in one kernel: the compiler stops counting), xs / xl inside a loop.
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "real-time-video-captioning_amd", "csrc")


def product_flags():
    """The compile flags of the product build, read from csrc/Makefile (CXXFLAGS with $(ARCH) / $(EXTRA) resolved to their defaults),
    so that what is linted is what is shipped."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    var = lambda n: re.search(r"^%s[ \t]*\??=[ \t]*(.*)$" % n, mk, re.M).group(1).strip()
    flags = var("CXXFLAGS").replace("$(ARCH)", var("ARCH")).replace("$(EXTRA)", var("EXTRA"))
    return [f for f in flags.split() if f not in ("-fPIC", "-Wall", "-Wno-unused-function")]


def hipcc_version():
    """'major.minor' of the HIP compiler (the ISA expectations of tests/test_isa_lint.py were taken on one specific compiler)."""
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
    m = re.search(r"HIP version:\s*(\d+)\.(\d+)", out)
    return "%s.%s" % (m.group(1), m.group(2)) if m else "unknown"


def kernel_listings(src, pat, extra_flags=()):
    """[(mangled name, NumVgprs, ScratchSize in bytes, tokens)] of every kernel of `src` whose mangled name contains `pat`."""
    src = src if os.path.exists(src) else os.path.join(CSRC, src)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc"] + product_flags() + list(extra_flags) + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-S",
                                                                             "--cuda-device-only", "-o", out, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    names = [l.split(":")[0] for l in text if re.match(r"^_Z\w+:", l)]
    res = []
    for name in [n for n in names if pat in n]:
        i = text.index(next(l for l in text if l.startswith(name + ":")))
        toks, vg, sc = [], None, None
        for l in text[i + 1:]:
            t = l.strip()
            if t.startswith("; NumVgprs:"):
                vg = t.split(":")[1].strip()
            if t.startswith("; ScratchSize:"):
                sc = t.split(":")[1].strip()
                break
            if "Loop Header" in t:
                toks.append("[")
            op = t.split()[0] if t and not t.startswith(";") and not t.startswith(".") else ""
            if op.startswith("global_load_lds"):
                toks.append("D")
            elif op == "global_load_dwordx4":
                toks.append("L")
            elif op.startswith("global_load"):
                toks.append("l")
            elif op == "global_store_dwordx4":
                toks.append("S")
            elif op == "global_store_dwordx2":
                toks.append("s")
            elif op.startswith("global_store"):
                toks.append("d")
            elif op.startswith("buffer_load"):
                toks.append("BL")
            elif op.startswith("buffer_store"):
                toks.append("BS")
            elif op.startswith("scratch_store"):
                toks.append("xs")
            elif op.startswith("scratch_load"):
                toks.append("xl")
            elif op == "s_barrier":
                toks.append("|")
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    toks.append("W" + m.group(1))
            elif op == "s_endpgm":
                toks.append("END")
        res.append((name, vg, int(sc) if sc is not None else None, toks))
    return res, names


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    res, names = kernel_listings(sys.argv[1], sys.argv[2])
    if not res:
        sys.exit("no kernel matches %r; kernels: %s" % (sys.argv[2], ", ".join(names)))
    for name, vg, sc, toks in res:
        print("%s\n  VGPRs %s, scratch %s bytes\n  %s\n" % (name, vg, sc, " ".join(toks)))


if __name__ == "__main__":
    main()
