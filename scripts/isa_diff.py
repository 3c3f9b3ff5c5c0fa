#!/usr/bin/env python3
"""Per-function diff of two `hipcc -S --cuda-device-only` outputs of csrc/engine.hip: which device functions changed?

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o new.s c-blosc_amd/csrc/engine.hip
    python scripts/isa_diff.py base.s new.s

Used before a GPU call to show that a change meant for one path (a new typesize, a new codec option) left the instruction
streams of the measured kernels untouched.  Labels and debug directives are normalised away."""
import re
import sys


def functions(path):
    out, name, body = {}, None, []
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+|k_\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            name, body = m.group(1), []
            out[name] = body
            continue
        if name is None:
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end"):
            name = None
            continue
        if not s or s.startswith((";", ".loc", ".file", ".cfi", ".p2align")):
            continue
        s = re.sub(r"\s*;.*$", "", s)                       # trailing comments carry basic-block numbers
        s = re.sub(r"\.L(BB|tmp|func_begin|func_end)\d+(_\d+)?", ".L", s)
        body.append(s)
    return out


def main():
    a, b = functions(sys.argv[1]), functions(sys.argv[2])
    same = [n for n in a if n in b and a[n] == b[n]]
    changed = [n for n in a if n in b and a[n] != b[n]]
    print(f"{len(same)} functions identical, {len(changed)} changed, {len(set(b) - set(a))} new, {len(set(a) - set(b))} gone")
    for tag, names in (("changed", changed), ("new", sorted(set(b) - set(a))), ("gone", sorted(set(a) - set(b)))):
        for n in names:
            la, lb = len(a.get(n, [])), len(b.get(n, []))
            print(f"  {tag:8s} {n[:110]}  ({la} -> {lb} lines)")


if __name__ == "__main__":
    main()
