#!/usr/bin/env python3
"""Register / spill / scratch figures of every kernel of the shipped sources, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage with the flags of csrc/Makefile).  Writes a ';'-separated table that
bench.py reads for roofline.resources:

    python tools/kernel_resources.py profiles/r04_kernel_resources.csv
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "monocular-visual-odometry_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Rpass-analysis=kernel-resource-usage -x hip -c".split()
FIELDS = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("TotalSGPRs", "sgpr"), ("VGPRs Spill", "vgpr_spill"), ("SGPRs Spill", "sgpr_spill"),
          ("ScratchSize [bytes/lane]", "scratch_bytes_per_lane"), ("Occupancy [waves/SIMD]", "occupancy"),
          ("LDS Size [bytes/block]", "static_lds_bytes")]


def short(name):
    """k_ba_lm<false, 32, 1>(BaBatch) -> k_ba_lm<false,32,1>"""
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace(", ", ",")


def main():
    out = sys.argv[1]
    rows = []
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + [src, "-o", "/dev/null"], cwd=CSRC,
                           capture_output=True, text=True)
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
            if m:
                dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                cur = dict(kernel=short(dem), file=src)
                rows.append(cur)
                continue
            for label, key in FIELDS:
                m = re.search(r"remark:\s+" + re.escape(label) + r": (\d+)", line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
    with open(out, "w") as o:
        o.write("# hipcc %s <file> (ROCm 7.2, flags of csrc/Makefile); instrumented instantiations (PROF = true) omitted\n" % " ".join(FLAGS))
        o.write(";".join(["kernel", "file"] + [k for _, k in FIELDS]) + "\n")
        for r in rows:
            if "<true" in r["kernel"]:
                continue
            o.write(";".join([r["kernel"], r["file"]] + [str(r.get(k, "")) for _, k in FIELDS]) + "\n")
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    main()
