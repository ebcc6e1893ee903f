"""Static VALU mix of a kernel's ISA by gfx950 issue class (development tool).

Classes as measured by tools/ubench/pairbench (cycles per wave-instruction per SIMD at 8 waves / SIMD):
  fast-int   v_xor/and/or/add_u32/sub_u32/lshrrev_b32/mov/bitop3/cndmask        2.3 - 2.9
  fast-fp    v_fma/fmac/mul/add/sub_f32                                          2.3 - 2.5 (overlap with slow ops)
  slow       everything else (alignbit, lshl, cvt, cmp, min/max, 3-operand int,
             64-bit int, f64, packed)                                            4.1 - 4.9
  trans      v_rcp/v_sqrt/...                                                    8
usage: python tools/isa_mix.py file.s [first_line last_line]"""
import collections
import re
import sys

FAST_INT = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_mov_b32",
            "v_bitop3_b32", "v_cndmask_b32", "v_not_b32"}
FAST_FP = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64", "v_sqrt_f64"}
COST = {"fast-int": 2.7, "fast-fp": 2.45, "slow": 4.3, "trans": 8.2}


def classify(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in FAST_INT:
        return "fast-int"
    if base in FAST_FP:
        return "fast-fp"
    if base in TRANS:
        return "trans"
    return "slow"


def kernel_lines(asm_path, mangled):
    """The lines of one kernel of a hipcc -S listing (label `mangled:` ... s_endpgm)."""
    out, inside = [], False
    for line in open(asm_path):
        if line.startswith(mangled + ":"):
            inside = True
        if inside:
            out.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                break
    return out


def mix(lines):
    """{class: count} of the VALU instructions in `lines`."""
    counts = collections.Counter()
    for line in lines:
        m = re.match(r"\s+(v_[a-z0-9_]+)\b", line)
        if m:
            counts[classify(m.group(1))] += 1
    return dict(counts)


def main():
    lines = open(sys.argv[1]).read().splitlines()
    if len(sys.argv) > 3:
        lines = lines[int(sys.argv[2]) - 1:int(sys.argv[3])]
    counts = collections.Counter()
    ops = collections.Counter()
    other = collections.Counter()
    for line in lines:
        m = re.match(r"\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+)\b", line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            c = classify(op)
            counts[c] += 1
            ops[(c, re.sub(r"_(e32|e64)$", "", op))] += 1
        else:
            other[op.split("_")[0]] += 1
    total = sum(counts.values())
    cycles = sum(COST[c] * n for c, n in counts.items())
    print(f"VALU {total}: " + ", ".join(f"{c} {n} ({100 * n / total:.0f} %)" for c, n in counts.most_common()))
    print(f"mix-weighted cost {cycles / total:.2f} cycles / instruction if nothing overlapped; "
          f"slow+trans share of those cycles {100 * (COST['slow'] * counts['slow'] + COST['trans'] * counts['trans']) / cycles:.0f} %")
    print("other:", dict(other))
    for c in ("slow", "fast-int", "fast-fp", "trans"):
        print(f"  {c}: " + ", ".join(f"{op[2:]} {n}" for (cc, op), n in ops.most_common() if cc == c))


if __name__ == "__main__":
    main()
