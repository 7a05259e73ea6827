#!/usr/bin/env python3
"""isa_mix.py <out.json> -- static instruction mix of the hot kernels' LOOP BODIES (gfx950 assembly of the tree's .hip files).

For a kernel that is bound by VALU issue, time >= sum over its wave-level VALU instructions of the issue cycles of each form.  The
SQ counters give the NUMBER of VALU instructions a kernel executed (SQ_INSTS_VALU) but not their forms; the forms are read off the
assembly: every instruction inside a loop (a label .. a backward branch to it) of the kernel is classed by the issue rates measured
with profiles/tools/ubench/valu_rate.hip on this GPU (profiles/r02_c2/README.md):
    2 cycles per wave64 instruction: plain 32-bit VOP1 / VOP2 forms (v_xor/and/or/add_u32/sub/not/mov/lshrrev/lshlrev/bcnt/bfrev/min/max_u32, v_cndmask with vcc, v_cmp e32)
    4 cycles: three-operand VOP3 forms (v_bfi, v_and_or, v_or3, v_alignbit, v_lshl_or, v_lshl_add, v_add3, v_xad, v_bfe, v_med3, v_perm, v_mad/mul_*),
              carry forms (v_add_co / v_addc_co / v_sub_co / v_subb_co), 64-bit forms (v_lshlrev_b64, v_lshrrev_b64, v_lshl_add_u64, v_*_f64 ...),
              e64 encodings with SGPR masks (v_cndmask_b32_e64, v_cmp_*_e64), every DPP form, v_readlane / v_readfirstlane / v_writelane
bench.py multiplies the counter's instruction count with the mix's average cycles per instruction:  alu_ms = N_valu x cpi / (1024 SIMDs x clock).
Static mix of the loops, not a dynamic trace: a loop that runs more often than another one weighs the same here."""
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "lordfast_amd", "csrc")
KERNELS = ["lf_seed_search_kernel", "lf_edlib_rsweep_kernel", "lf_edlib_tb_kernel", "lf_render_kernel", "lf_vote_cell_kernel", "lf_seed_locate_kernel",
           "lf_chain_n2_kernel", "lf_chain_n2_big_kernel", "lf_ksw_mw_kernel", "lf_ksw_r4_kernel", "lf_hirsch_level_kernel", "lf_hband_level_kernel", "lf_hband_group_kernel", "lf_sam_write_kernel"]
FOUR = re.compile(r"^v_(bfi|and_or|or3|alignbit|alignbyte|lshl_or|lshl_add|add3|xad|bfe|med3|perm|mad|mul_|fma|add_co|addc_co|sub_co|subb_co|subrev_co|subbrev_co|"
                  r"lshlrev_b64|lshrrev_b64|ashrrev_i64|lshl_add_u64|readlane|readfirstlane|writelane|mbcnt|cvt_|rcp|div_|sad|min3|max3|add_lshl|xor3|cndmask_b32_e64|"
                  r"cmp_\w+_e64|cmpx_\w+_e64|[a-z_0-9]+_f64|[a-z_0-9]+_dpp|pk_)")


def tree_hash():
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(CSRC, "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def asm_of(src):
    out = "/tmp/isa_mix_" + os.path.basename(src) + ".s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-Wno-unused-result", "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read()


def kernels_in(asm):
    """-> {mangled name: [instruction lines]} for every kernel entry (.amdhsa_kernel names) of the file"""
    names = set(re.findall(r"\.amdhsa_kernel\s+(\S+)", asm))
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(\S+):\s*(;.*)?$", line)
        if m and m.group(1) in names:
            cur = m.group(1); out[cur] = []; continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None; continue
        out[cur].append(line)
    return out


def loop_mix(lines):
    labels, ins = {}, []
    for l in lines:
        t = l.strip()
        m = re.match(r"^(\.LBB\S+):", t)
        if m:
            labels[m.group(1)] = len(ins); continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        ins.append(t.split(";")[0].strip())
    in_loop = [False] * len(ins)
    for i, t in enumerate(ins):
        m = re.match(r"^s_c?branch\S*\s+(\.LBB\S+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            for k in range(labels[m.group(1)], i + 1):
                in_loop[k] = True
    sel = [t for t, f in zip(ins, in_loop) if f] or ins
    n2 = n4 = salu = vmem = lds = 0
    for t in sel:
        op = t.split()[0]
        if op.startswith("v_"):
            dpp = " row_" in t or " wave_" in t or "quad_perm" in t or "row_bcast" in t
            if dpp or FOUR.match(op):
                n4 += 1
            else:
                n2 += 1
        elif op.startswith("s_"):
            salu += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            vmem += 1
        elif op.startswith("ds_"):
            lds += 1
    nv = n2 + n4
    return dict(valu_in_loops=nv, two_cycle=n2, four_cycle=n4, cycles_per_valu=(2 * n2 + 4 * n4) / nv if nv else None, salu_in_loops=salu, vmem_in_loops=vmem,
                lds_in_loops=lds, all_instructions=len(ins), instructions_in_loops=len(sel))


def main():
    res = {"_meta": {"source_tree": tree_hash(), "method": __doc__.split("\n\n")[1][:400]}}
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        text = open(src).read()
        if not any(k in text for k in KERNELS):
            continue
        try:
            ks = kernels_in(asm_of(src))
        except subprocess.CalledProcessError as e:
            print("failed:", src, e.stderr[-500:].decode(), file=sys.stderr); continue
        for name, lines in ks.items():
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            short = re.sub(r"^void ", "", dem).split("(")[0]
            if any(short.startswith(k) for k in KERNELS):
                res[short] = loop_mix(lines)
    json.dump(res, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        if k != "_meta":
            print(f"{k:60s} loops: {v['valu_in_loops']:5d} VALU ({v['two_cycle']} x 2 + {v['four_cycle']} x 4 cycles) -> {v['cycles_per_valu']:.2f} cycles per VALU instruction; SALU {v['salu_in_loops']}, VMEM {v['vmem_in_loops']}, LDS {v['lds_in_loops']}")


if __name__ == "__main__":
    main()
