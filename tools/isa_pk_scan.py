#!/usr/bin/env python3
"""Static scan of gfx950 ISA (hipcc -S --cuda-device-only) for the instruction pattern behind the packed-FP32 observation of
DESIGN.md section 11.3: a packed FP32 VALU op that selects register halves (op_sel / op_sel_hi) issued as the FIRST vector
instruction behind an EXEC write (s_and_saveexec / s_or_b64 exec / ... , or a block entry reached by such a branch).
usage: isa_pk_scan.py file.s [...]   -> per kernel: sites, with the distance (instructions) to the EXEC write / label"""
import re
import sys

EXECW = re.compile(r"^\s*(s_and_saveexec_b64|s_or_saveexec_b64|s_andn2_saveexec_b64|s_(or|and|andn2|xor|mov)_b64\s+exec)")
PK = re.compile(r"^\s*v_pk_(add|mul|fma)_f32\b.*op_sel")
LABEL = re.compile(r"^\.LBB\d+_\d+:")
FUNC = re.compile(r"^(_Z\w+):")
SKIP = re.compile(r"^\s*(;|\.|$)|^\s*;;#")


def scan(path, window=1):
    func, sites = None, []
    hist = []       # last real instructions / labels
    for ln in open(path):
        m = FUNC.match(ln)
        if m:
            func = m.group(1)
            hist = []
            continue
        if LABEL.match(ln):
            hist.append(("label", ln.strip()))
            continue
        if SKIP.match(ln):
            continue
        ins = ln.strip()
        if PK.match(ln):
            back = hist[-window:]
            why = [k for k, _ in back if k in ("label", "execw")]
            if why:
                sites.append((func, ins, [t for _, t in back]))
        hist.append(("execw" if EXECW.match(ln) else "ins", ins))
    return sites


for p in sys.argv[1:]:
    s = scan(p)
    per = {}
    for f, ins, back in s:
        per.setdefault(f, []).append((ins, back))
    print(f"{p}: {len(s)} packed-FP32 op_sel instructions directly behind an EXEC write or a block entry, in {len(per)} kernels")
    for f, v in sorted(per.items(), key=lambda kv: -len(kv[1]))[:8]:
        print(f"   {len(v):4d}  {f[:90]}")
        print(f"         e.g. {v[0][1][-1]}  ->  {v[0][0]}")
