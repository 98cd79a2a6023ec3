#!/bin/bash
# Runs on the GPU box (through gpurun): SQ counters of attn_full_kernel at S = 197 (ViT frames) and S = 1182 (decoder image
# prefix) -- is the kernel VALU-issue bound, as its cycle arithmetic says (docs/LAB_NOTEBOOK.md par. 6b)?  Separate --pmc passes
# (8 SQ slots each), kernel trace only.  Summarise with tools/attn_pmc_table.py.
set -e
out=gpurun_out/prof_attn_${1:-x}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L > $out/counters_available.txt 2>&1 || true
P="python3 tools/attn_bench.py"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $out/pmc_a --output-format csv -- $P > $out/pmc_a.out 2> $out/pmc_a.err || echo "pass a failed"
echo "pmc a done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $out/pmc_b --output-format csv -- $P > $out/pmc_b.out 2> $out/pmc_b.err || echo "pass b failed"
echo "pmc b done"
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $out/pmc_c --output-format csv -- $P > $out/pmc_c.out 2> $out/pmc_c.err || echo "pass c failed"
echo "pmc c done"
find $out -name "*.csv" -size +20M -delete
du -sh $out
