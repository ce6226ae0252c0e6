"""Static VALU instruction mix of a kernel from the compiler's ISA listing, priced with the issue rates measured on this chip
(profiles/r01_valu_rate.txt, tools/microbench/valu_rate.hip: cycles per wave64 instruction per SIMD at 4-8 waves per SIMD).

    python tools/isa_valu_mix.py <kernel name substring> [listing.s]  ->  JSON on stdout

Without a listing the extractor kernels are compiled with --save-temps into a temporary directory.  The mix is STATIC (every
instruction of the listing counts once, whatever its loop depth): it says which share of the kernel's vector instructions
issue in 2 cycles instead of 4, which is what bench.py's `roofline_valu` needs to price SQ_INSTS_VALU (a dynamic count).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# 2-cycle class of profiles/r01_valu_rate.txt: add / sub / logic / mov / right shifts / bitop3 / unpacked 16-bit / f32 add mul fma
TWO = re.compile(r"^v_(add|sub|subrev)_(u32|i32|co_u32|u16|i16|f32|nc_u32)|^v_(and|or|xor|not)_b32|^v_mov_b32|^v_lshrrev_b32|^v_ashrrev_i32|"
                 r"^v_bitop3_b32|^v_(max|min)_(u16|i16)|^v_(mul|fma|mac|fmac)_f32|^v_addc_co_u32|^v_subb_co_u32|^v_accvgpr")
SKIP = re.compile(r"^v_(readlane|readfirstlane|writelane|nop)")   # SALU-like / not counted as VALU work


def listing(kernel):
    d = tempfile.mkdtemp(prefix="isa_")
    src = os.path.join(ROOT, "active-orb-slam2_amd", "csrc", "extractor_kernels.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-munsafe-fp-atomics", "--save-temps=obj", "-c", src, "-o", os.path.join(d, "ek.o")],
                          cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
    return [os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s")][0]


def mix(kernel, path):
    lines = open(path).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(kernel), ln))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    counts = {}
    for ln in lines[start:end]:
        t = ln.strip().split()
        if not t or not t[0].startswith("v_") or SKIP.match(t[0]):
            continue
        op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
        counts[op] = counts.get(op, 0) + 1
    two = sum(c for op, c in counts.items() if TWO.match(op))
    tot = sum(counts.values())
    return dict(kernel=kernel, valu_instructions_static=tot, two_cycle=two, four_cycle=tot - two, two_cycle_share=two / tot,
                mean_cycles_per_instruction=(2.0 * two + 4.0 * (tot - two)) / tot,
                rates="profiles/r01_valu_rate.txt: 2 cycles per wave64 instruction for add / sub / and / or / xor / mov / lshrrev / "
                      "bitop3 / unpacked 16-bit max min add / f32 add mul fma, 4 for everything else the kernel uses (packed 16-bit, "
                      "compares, min / max 32-bit, mad / mul 24-bit, perm / alignbyte, lshl, bfe, mbcnt, cndmask)",
                top=sorted(counts.items(), key=lambda kv: -kv[1])[:24])


if __name__ == "__main__":
    k = sys.argv[1]
    print(json.dumps(mix(k, sys.argv[2] if len(sys.argv) > 2 else listing(k)), indent=1))
