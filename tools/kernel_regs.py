#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of the kernels of a library (from the metadata notes of its gfx950 code objects):

    python3 tools/kernel_regs.py LIB.so [REGEX]
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_vgpr_hazard as H  # noqa: E402


def main():
    path = sys.argv[1]
    rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    for triple, blob in H.code_objects(path):
        if "gfx950" not in triple:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(blob)
        try:
            notes = H.run(H.LLVM + "/llvm-readelf", "--notes", f.name)
        finally:
            os.unlink(f.name)
        # one list item per kernel; it starts with "  - .agpr_count:" and every key of the kernel is indented by four
        cur = None
        for line in notes.splitlines():
            if re.match(r"  - \.\w+:", line):
                if cur and "vgpr_count" in cur:
                    show(cur, rx)
                cur = {}
                line = "    " + line[4:]
            m = re.match(r"    \.(\w+):\s*(.*)$", line)
            if m and cur is not None:
                cur[m.group(1)] = m.group(2).strip()
        if cur and "vgpr_count" in cur:
            show(cur, rx)


def demangle(name):
    """Enough of the Itanium scheme for this library's kernels: namespace, name and integer template arguments."""
    m = re.match(r"_ZN4cvec(\d+)", name)
    if not m:
        m2 = re.match(r"_Z(\d+)", name)
        if not m2:
            return name
        n = int(m2.group(1))
        return name[len(m2.group(0)):len(m2.group(0)) + n]
    n = int(m.group(1))
    at = len(m.group(0))
    base = name[at:at + n]
    rest = name[at + n:]
    args = []
    if rest.startswith("I"):
        for a in re.finditer(r"L[ijb](n?\d+)E", rest.split("EE", 1)[0] + "E"):
            args.append(a.group(1).replace("n", "-"))
    return base + ("<" + ", ".join(args) + ">" if args else "")


def show(k, rx):
    name = k.get("name", "?")
    if rx and not rx.search(name):
        return
    dem = demangle(name)
    print("%-60s vgpr %3s agpr %3s sgpr %3s spill v%s s%s scratch %s lds %s" % (
        dem[:60], k.get("vgpr_count"), k.get("agpr_count", "0"), k.get("sgpr_count"), k.get("vgpr_spill_count", "0"),
        k.get("sgpr_spill_count", "0"), k.get("private_segment_fixed_size", "0"), k.get("group_segment_fixed_size", "0")))


if __name__ == "__main__":
    main()
