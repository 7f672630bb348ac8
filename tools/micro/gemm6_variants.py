"""Where do the cycles of gemm_bf16x6's main loop go?  (Variants of the first round-2 structure - DMA between the MFMA groups of the second
half: bunched, next to the fragment reads, through registers, k16-panel addresses - are recorded in DESIGN.md 4.1; this file patches the
current structure.)  Runs ON THE GPU BOX (scratch copy of the repository): patches the
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
    return s.replace("const bool refill = rkt < nk && (grpB || kt >= 1);", "const bool refill = false;")


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
    s = s.replace('if (!grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // A: its share', '// A: its share')
    return s.replace("if (grpB) {\n                    if (refill)", "if (false) {\n                    if (refill)")


def samek(s):          # every k-tile fetches k-tile 0 again: same lines, L2 / TCP hits only
    return s.replace("const int rkt = kt + 1 + grpB;", "const int rkt = 0;")


def halfdma(s):        # only the first DMA instruction triple of the wave (half the pieces)
    return s.replace("for (int d = 0; d < NDMA; ++d) {\n                const int i = d / 3, pl = d % 3;", "for (int d = 0; d < NDMA / 2; ++d) {\n                const int i = d / 3, pl = d % 3;")


def nostagger(s):
    return s.replace("if (grpB) __builtin_amdgcn_s_barrier();", "").replace("if (!grpB) __builtin_amdgcn_s_barrier();", "")


def head_split(s):     # three DMA instructions behind the B-fragment reads, three behind the first A-fragment reads
    a = s.index("        if (refill) {\n#pragma unroll\n            for (int d = 0; d < NDMA; ++d) {")
    b = s.index("        bf16x8 afp[3];")
    blk = s[a:b]
    first = blk.replace("for (int d = 0; d < NDMA; ++d) {", "for (int d = 0; d < NDMA / 2; ++d) {")
    second = blk.replace("for (int d = 0; d < NDMA; ++d) {", "for (int d = NDMA / 2; d < NDMA; ++d) {")
    s = s[:a] + first + s[b:]
    # second block behind the MFMAs of m-tile 0
    marker = "            if (mt == HM - 1) {\n                // middle barrier"
    i = s.index(marker)
    return s[:i] + "            if (mt == 0) {\n" + second.replace("        if (refill) {", "            if (refill) {") + "            }\n" + s[i:]


def head_after_mt0(s):  # all six DMA instructions behind the MFMAs of m-tile 0
    a = s.index("        if (refill) {\n#pragma unroll\n            for (int d = 0; d < NDMA; ++d) {")
    b = s.index("        bf16x8 afp[3];")
    blk = s[a:b]
    s = s[:a] + s[b:]
    marker = "            if (mt == HM - 1) {\n                // middle barrier"
    i = s.index(marker)
    return s[:i] + "            if (mt == 0) {\n" + blk + "            }\n" + s[i:]


def head_interleaved(s):   # one DMA instruction behind every B-fragment read
    a = s.index("        if (refill) {\n#pragma unroll\n            for (int d = 0; d < NDMA; ++d) {")
    b = s.index("        bf16x8 afp[3];")
    s = s[:a] + s[b:]
    old = "            for (int pl = 0; pl < 3; ++pl) bf[nt][pl] = *reinterpret_cast<const bf16x8*>(&img[3 * A_PLANE + pl * B_PLANE + off]);"
    new = """            for (int pl = 0; pl < 3; ++pl) {
                bf[nt][pl] = *reinterpret_cast<const bf16x8*>(&img[3 * A_PLANE + pl * B_PLANE + off]);
                const int d = nt * 3 + pl;
                if (refill && d < NDMA) {
                    const int i = d / 3, pl2 = d % 3;
                    const bool isA_ = (wave + NW * i) < CFG::A_INSTR;
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_global_load_lds(G6_ADDR(sbase[i] + (pl2 * pstride[i] + (int64_t)rkt * kstride[i]), voff[i]),
                        (lds_ptr6)(smem6 + rst * STAGE + ldsoff[i] + pl2 * (isA_ ? A_PLANE : B_PLANE)), 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }"""
    assert old in s
    return s.replace(old, new)


VARIANTS = {"head_interleaved": head_interleaved, "base": lambda s: s, "nodma": nodma, "nolds": nolds, "nobar": nobar, "nodma_nolds": lambda s: nolds(nodma(s)),
            "nodma_nolds_nobar": lambda s: nobar(nolds(nodma(s))), "halfdma": halfdma, "samek": samek, "nowait": nowait,
            "nostagger": nostagger, "head_split": head_split, "head_after_mt0": head_after_mt0}
which = sys.argv[1:] or list(VARIANTS)
BUILD = [sys.executable, "-c", "from robust_speech_analysis_framework_amd import build; build.build_library(verbose=False)"]


def restore():
    """Whatever happened (failed build, interrupt): the product source and the built library are the real kernel again."""
    open(SRC, "w").write(orig)
    subprocess.run(BUILD, check=False, cwd=ROOT)


import atexit
atexit.register(restore)
for name in which:
    src = VARIANTS[name](orig)
    assert name == "base" or src != orig, name
    open(SRC, "w").write(src)
    subprocess.run(BUILD, check=True, cwd=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm6_bench.py")], capture_output=True, text=True, cwd=ROOT)
    for line in r.stdout.splitlines():
        if line.startswith(("qkv", "ffn2", "out-proj")):
            m = re.search(r"bf16x6 row-major\s+([\d.]+) ms\s+([\d.]+) TF-eq", line)
            u = re.search(r"as used \([^)]*\)\s+([\d.]+) ms\s+([\d.]+) TF-eq", line)
            print(f"{name:20s} {line.split()[0]:10s} row-major {m.group(1)} ms {m.group(2)} TF-eq | as used {u.group(1)} ms {u.group(2)} TF-eq", flush=True)
    if r.returncode != 0:
        print(r.stderr[-2000:])
