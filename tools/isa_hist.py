#!/usr/bin/env python3
"""Instruction histogram of one kernel in a gfx950 .s file, weighted with the measured issue costs
(profiles/r01_isa_rates.txt).  usage: isa_hist.py file.s kernel_substring"""
import collections, re, sys
COST = collections.defaultdict(lambda: 4.5, {"v_add_u32": 2.7, "v_sub_u32": 2.7, "v_mov_b32": 2.5, "v_and_b32": 2.7,
        "v_or_b32": 2.7, "v_xor_b32": 2.7, "v_lshrrev_b32": 2.7, "v_lshlrev_b32": 2.7, "v_mad_u64_u32": 4.9,
        "s_nop": 1.0, "s_waitcnt": 0.0, "s_mov_b32": 0.0})
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l) and pat in l)
hist = collections.Counter()
for l in src[start + 1:]:
    if "s_endpgm" in l:
        break
    m = re.match(r"\s+([vsd][a-z0-9_]+)", l)
    if m:
        op = re.sub(r"_e(32|64)$", "", m.group(1))
        hist[op] += 1
tot = sum(hist.values()); cyc = sum(COST[o] * c for o, c in hist.items() if o.startswith("v_") or o.startswith("ds_"))
print(f"{pat}: {tot} instructions, ~{cyc:.0f} weighted VALU/LDS issue cycles per wave")
for o, c in hist.most_common(28):
    print(f"  {c:6d}  {o}")
