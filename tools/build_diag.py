# Builds tools/libgitcap_diag.so: a copy of csrc/ with s_memtime / s_memrealtime / HW_ID stamps in gemm256
# (read by tools/gemm_timeline.py through GEMM_DBG_PTR) plus the measured-and-rejected tile kernels of
# tools/experiments/ (gemm256p.hip: persistent tiles, gemm2b.hip: two workgroups per CU; docs/LAB_NOTEBOOK.md "What did not
# work"), reachable through gitcap_dbg_gemm(tile = 259 | 260 | 258).  The product library has neither.
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix='gitcap_diag_')
os.makedirs(os.path.join(tmp, 'pkg', 'gitcap')); os.makedirs(os.path.join(tmp, 'include'))
src = os.path.join(ROOT, 'real-time-video-captioning_amd', 'csrc')
dst = os.path.join(tmp, 'pkg', 'csrc')
shutil.copytree(src, dst, ignore=shutil.ignore_patterns('build'))
shutil.copy(os.path.join(ROOT, 'include', 'gitcap.h'), os.path.join(tmp, 'include'))
for f in ('gemm256p.hip', 'gemm2b.hip', 'gemm2w.hip'):
    shutil.copy(os.path.join(ROOT, 'tools', 'experiments', f), dst)
open(os.path.join(dst, 'kernels.h'), 'a').write(
    '\nhipError_t launch_gemm256p(const GemmArgs& a, int epi, hipStream_t s);\nbool gemm2b_ok(const GemmArgs& a);\n'
    'hipError_t launch_gemm2b(const GemmArgs& a, int epi, hipStream_t s);\nbool gemm2w_ok(const GemmArgs& a);\nhipError_t launch_gemm2w(const GemmArgs& a, int epi, hipStream_t s);\n')
def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert s.count(old) == 1, (path, old)
        s = s.replace(old, new)
    open(path, 'w').write(s)
patch(os.path.join(dst, 'gemm256.hip'), [
    ("    const int nt = a.K >> 6;\n", "    const int nt = a.K >> 6;\n    unsigned long long T0 = __builtin_amdgcn_s_memtime();\n    unsigned long long R0 = __builtin_amdgcn_s_memrealtime();\n"),
    ("    BARRIER();\n    if (grp == 1) BARRIER();", "    BARRIER();\n    unsigned long long T1 = __builtin_amdgcn_s_memtime();\n    if (grp == 1) BARRIER();"),
    ("    if (grp == 0) BARRIER();                       // matches group 1's extra leading barrier\n",
     "    if (grp == 0) BARRIER();                       // matches group 1's extra leading barrier\n    unsigned long long T2 = __builtin_amdgcn_s_memtime();\n"),
    ("        gemm_epilogue_wave<EPI>(a, acc, smem + wid * EPI_REGION, m0 + wm * 64, n0 + wn * 128, lane);\n",
     "        gemm_epilogue_wave<EPI>(a, acc, smem + wid * EPI_REGION, m0 + wm * 64, n0 + wn * 128, lane);\n"
     "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    unsigned long long T3 = __builtin_amdgcn_s_memtime();\n"
     "    if (EPI != EPI_PATCH_F32 && a.pos && lane == 0) {\n"
     "        unsigned long long* d = (unsigned long long*)a.pos + ((size_t)blockIdx.x * 8 + wid) * 8;\n"
     "        d[0] = T0; d[1] = T1; d[2] = T2; d[3] = T3;\n"
     "        d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID\n"
     "        d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID\n"
     "        d[6] = R0; d[7] = __builtin_amdgcn_s_memrealtime();\n    }\n"),
])
patch(os.path.join(dst, 'gitcap.hip'), [
    ("    if (tile != 64 && tile != 128 && tile != 256 && tile != 224 && tile != 257) return GITCAP_ERR_ARG;\n    hipError_t e = GITCAP_DBG_GEMM_DISPATCH(tile, a, epi, (hipStream_t)stream);",
     "    if (getenv(\"GEMM_DBG_PTR\")) a.pos = (const float*)strtoull(getenv(\"GEMM_DBG_PTR\"), nullptr, 0);\n    hipError_t e = GITCAP_DBG_GEMM_DISPATCH(tile, a, epi, (hipStream_t)stream);"),
    ("#ifndef GITCAP_DBG_GEMM_DISPATCH\n", "#define GITCAP_DBG_GEMM_DISPATCH(tile, a, epi, s) ((tile) == 258 ? launch_gemm2w(a, epi, s) : (tile) == 260 ? launch_gemm2b(a, epi, s) : (tile) == 259 ? launch_gemm256p(a, epi, s) : (tile) == 256 ? launch_gemm256(a, epi, s) : (tile) == 64 ? launch_gemm64(a, epi, s) : launch_gemm(a, epi, s))\n#ifndef GITCAP_DBG_GEMM_DISPATCH\n"),
])
srcs = 'SRCS=' + ' '.join(sorted(f for f in os.listdir(dst) if f.endswith('.hip')))
subprocess.check_call(['make', '-C', dst, '-j8', srcs])
shutil.copy(os.path.join(tmp, 'pkg', 'gitcap', 'libgitcap.so'), os.path.join(ROOT, 'tools', 'libgitcap_diag.so'))
shutil.rmtree(tmp)
print('wrote tools/libgitcap_diag.so')
