#!/usr/bin/env python3
"""Counts the vector-memory instructions of every kernel instance in a HIP source file (hipcc --cuda-device-only -S): before an
ablation's time is quoted, its instance must issue the loads of the full kernel (minus what the ablation is meant to remove).
    python3 tools/count_loads.py tools/ubench_mfma_bfly.hip -DBFLY_ABLATE
prints, per kernel: global_load_dwordx4 / other global + buffer loads / stores / v_mfma / ds_read_b128."""
import collections, re, subprocess, sys, tempfile, os

src, flags = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out] + flags,
                          stderr=subprocess.DEVNULL)
    kernel, counts = None, collections.OrderedDict()
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
            counts[kernel] = collections.Counter()
            continue
        if kernel is None:
            continue
        t = line.split()
        if not t:
            continue
        op = t[0]
        if op == "global_load_dwordx4" or op == "buffer_load_dwordx4":
            counts[kernel]["load_x4"] += 1
        elif op.startswith(("global_load", "buffer_load")):
            counts[kernel]["load_other"] += 1
        elif op.startswith(("global_store", "buffer_store")):
            counts[kernel]["store"] += 1
        elif op.startswith("v_mfma"):
            counts[kernel]["mfma"] += 1
        elif op == "ds_read_b128":
            counts[kernel]["ds_read_b128"] += 1
        elif op.startswith("scratch_"):
            counts[kernel]["scratch"] += 1
demangle = subprocess.run(["c++filt"] + list(counts), capture_output=True, text=True).stdout.split("\n")
for name, (k, c) in zip(demangle, counts.items()):
    if not any(c.values()):
        continue
    short = re.sub(r"\(hbmpc::mf::MfmaRowsArgs\)|void |hbmpc::mf::", "", name)
    print(f"{short:64s} load_x4 {c['load_x4']:3d}  other loads {c['load_other']:3d}  stores {c['store']:3d}  mfma {c['mfma']:4d}  "
          f"ds_read_b128 {c['ds_read_b128']:4d}  scratch {c['scratch']:3d}")
