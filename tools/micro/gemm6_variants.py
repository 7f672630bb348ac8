"""Where do the cycles of gemm_bf16x6's main loop go?  Runs ON THE GPU BOX (scratch copy of the repository): patches the
kernel source textually into diagnostic variants (wrong results, valid timing), rebuilds librsaf.so for each and times two
shapes.  The product source in the repository is not touched (the box's copy is thrown away)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "robust_speech_analysis_framework_amd", "csrc", "gemm_bf16x6.hip")
orig = open(SRC).read()


def nodma(s):
    return s.replace("const bool refill = kt + 2 < nk;", "const bool refill = false;")


def nolds(s):
    # fragments from registers: every LDS fragment read becomes a copy of a loop-invariant register value
    s = s.replace("    int st = 0;                                              // stage of k-tile kt",
                  "    bf16x8 frag0 = *reinterpret_cast<const bf16x8*>(&smem6[lane * 8]);\n    int st = 0;")
    return re.sub(r"\*reinterpret_cast<const bf16x8\*>\(&img\[[^;]*\]\);", "frag0;", s)


def nobar(s):
    a = s.index("    for (int kt = 0; kt < nk; ++kt) {")
    b = s.index("    if (!grpB) __builtin_amdgcn_s_barrier();")
    return s[:a] + s[a:b].replace("__builtin_amdgcn_s_barrier();", "") + s[b:]


def nowait(s):
    return s.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n                __builtin_amdgcn_s_barrier();',
                     '__builtin_amdgcn_s_barrier();')


def samek(s):          # every k-tile fetches k-tile 0 again: same lines, L2 / TCP hits only
    return s.replace("const int rkt = kt + 2;", "const int rkt = 0;")


def halfdma(s):        # only the first DMA instruction triple of the wave (half the pieces)
    return s.replace("if (refill && live[i]) {", "if (refill && live[i] && i == 0) {")


def nostagger(s):
    return s.replace("if (grpB) __builtin_amdgcn_s_barrier();", "").replace("if (!grpB) __builtin_amdgcn_s_barrier();", "")


def bunched(s):        # all DMA instructions of the wave behind the first MFMA group of the second half
    return s.replace("if (d * HGROUPS / NDMA == grp_) {", "if (grp_ == 0) {")


def firsthalf(s):      # DMA between the MFMA groups of the FIRST half (needs the stage to be free: timing only)
    return s.replace("if (mt >= HM) {                              // DMA", "if (mt < HM) {  // DMA").replace(
        "const int grp_ = (mt - HM) * TN + nt;", "const int grp_ = mt * TN + nt;")


def blocked(s):        # k16-panel operand layout [K/16][rows][16]: a DMA piece = 1 KiB contiguous (timing only, M, N % 256 == 0)
    s = s.replace("voff[i] = 2u * ((unsigned)rr * (unsigned)p.lda + 8u * dch);", "voff[i] = 1024u * jj + 16u * lane;")
    s = s.replace("voff[i] = 2u * ((unsigned)rr * (unsigned)p.ldb + 8u * dch);", "voff[i] = 1024u * jj + 16u * lane;")
    s = s.replace("sbase[i] = p.A + z * p.sA + (int64_t)m0 * p.lda;", "sbase[i] = p.A + (int64_t)m0 * 16;")
    s = s.replace("sbase[i] = p.B + (int64_t)n0 * p.ldb;", "sbase[i] = p.B + (int64_t)n0 * 16;")
    s = s.replace("(int64_t)rkt * BK)", "(int64_t)rkt * BK * (isA_ ? p.M : p.N))")
    s = s.replace("(int64_t)(KT) * BK)", "(int64_t)(KT) * BK * (isA_ ? p.M : p.N))")
    return s


def regstage(s):       # timing only: k-tiles through registers (global_load_dwordx4, ds_write_b128 one k-tile later) instead of LDS-DMA
    s = s.replace("    int st = 0;                                              // stage of k-tile kt",
                  "    uint4 stg[NDMA];\n#pragma unroll\n    for (int d = 0; d < NDMA; ++d) stg[d] = make_uint4(0, 0, 0, 0);\n    int st = 0;")
    old = """                                __builtin_amdgcn_global_load_lds(G6_ADDR(sbase[i] + (pl * pstride[i] + (int64_t)rkt * kstride[i]), voff[i]),
                                    (lds_ptr6)(smem6 + rst * STAGE + ldsoff[i] + pl * (isA_ ? A_PLANE : B_PLANE)), 16, 0, 0);"""
    new = """                                *reinterpret_cast<uint4*>(smem6 + rst * STAGE + ldsoff[i] + pl * (isA_ ? A_PLANE : B_PLANE) + lane * 8) = stg[d];
                                stg[d] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sbase[i] + (pl * pstride[i] + (int64_t)rkt * kstride[i])) + voff[i]);"""
    assert old in s
    return s.replace(old, new)


VARIANTS = {"regstage": regstage,"blocked": blocked,"nostagger": nostagger, "bunched": bunched, "firsthalf": firsthalf,"nowait": nowait, "samek": samek, "halfdma": halfdma,"base": lambda s: s, "nodma": nodma, "nolds": nolds, "nobar": nobar, "nodma_nolds": lambda s: nolds(nodma(s)),
            "nodma_nolds_nobar": lambda s: nobar(nolds(nodma(s)))}
which = sys.argv[1:] or list(VARIANTS)
for name in which:
    src = VARIANTS[name](orig)
    assert name == "base" or src != orig, name
    open(SRC, "w").write(src)
    subprocess.run([sys.executable, "-c", "from robust_speech_analysis_framework_amd import build; build.build_library(verbose=False)"],
                   check=True, cwd=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm6_bench.py")], capture_output=True, text=True, cwd=ROOT)
    for line in r.stdout.splitlines():
        if line.startswith(("qkv", "ffn2", "out-proj")):
            m = re.search(r"bf16x6\s+([\d.]+) ms\s+([\d.]+) TF-eq", line)
            print(f"{name:20s} {line.split()[0]:10s} {m.group(1)} ms {m.group(2)} TF-eq", flush=True)
    if r.returncode != 0:
        print(r.stderr[-2000:])
open(SRC, "w").write(orig)
