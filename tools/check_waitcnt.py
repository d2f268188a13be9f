#!/usr/bin/env python3
"""Static check of the first-phase copies of the scan kernels: compiles one instantiation TU to gfx950 assembly and lists, for
every kernel, the s_waitcnt vmcnt(k) values that follow the tile's unconditional non-temporal load group, copy by copy
(a copy = a basic block that starts a fresh 3, 2, ... countdown).  A copy whose FIRST wait is vmcnt(0) or vmcnt(1) computes
nothing before all four loads of the tile have landed - measured at 6.3 instead of 7.4 TB/s - and the compiler produces
such copies now and then depending on unrelated code around them.
    python tools/check_waitcnt.py [scan_inst_u4_nt1.hip] [-DNAME ...]      exit status 1 if any copy is bad"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sliceslice-rs_amd", "csrc")


def main():
    args = sys.argv[1:]
    defs = [a for a in args if a.startswith("-D")]
    files = [a for a in args if not a.startswith("-D")] or ["scan_inst_u4_nt1.hip"]
    bad = 0
    for f in files:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                            os.path.join(CSRC, f), "-I" + os.path.join(ROOT, "include"), "-o", out] + defs,
                           check=True, stderr=subprocess.DEVNULL)
            lines = open(out).read().split("\n")
        kernels = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN2ss\w+:", l)]
        kernels.append((len(lines), None))
        for (a, name), (b, _) in zip(kernels, kernels[1:]):
            body = lines[a:b]
            first = next((i for i, l in enumerate(body) if "global_load_dwordx4" in l and " nt" in l and re.search(r"v\d+, s\[", l)), None)
            if first is None:
                continue
            # walk forward until the next load group; record the first vmcnt wait of every basic block that has one
            copies, cur = [], None
            for l in body[first + 4:]:
                if "global_load_dwordx4" in l and " nt" in l:
                    break
                if re.match(r"^\.LBB", l):
                    cur = None
                m = re.search(r"s_waitcnt vmcnt\((\d+)\)", l)
                if m:
                    k = int(m.group(1))
                    if k > 4:                      # the second phase (needle staging, compares): not a first-phase copy any more
                        break
                    if cur is None:
                        cur = [k]
                        copies.append(cur)
                    else:
                        cur.append(k)
            nbad = sum(1 for c in copies if c[0] <= 1)
            demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"{f}: {demangled[:70]:70s} copies {copies}  {'BAD ' + str(nbad) if nbad else 'ok'}")
            bad += nbad
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
