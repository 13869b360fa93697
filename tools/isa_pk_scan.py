#!/usr/bin/env python3
"""Static check of the SHIPPED gfx950 code for packed-FP32 VALU instructions.

On MI355X a `v_pk_add_f32 ... op_sel:[0,1]` (the LOW result takes its operand from the HIGH register of the pair) delivered
the low result with that operand read as 0 in lanes 48..63, intermittently, when a second wave on the SIMD was in an MFMA loop
(instrumented kernel, tools/exp/pkf32_check.py).  Every kernel file but conv3x3_h16.hip is therefore built without
packed-FP32 ops (csrc/Makefile: -target-feature -packed-fp32-ops); the 16 x 16 halo conv keeps them (30 % of its time) and
its code holds only the op_sel_hi broadcast forms.  This script proves both facts on the binary -- the rule of
tests/test_host.py::test_shipped_code_has_no_low_lane_op_sel_packed_fp32:

    isa_pk_scan.py <libmadm_hip.so | file.s ...>      exit status 1 when a packed-FP32 instruction feeds a LOW result from
                                                      a HIGH register, or sits outside conv3x3_h16_kernel
    isa_pk_scan.py --strict ...                       exit status 1 on ANY packed-FP32 instruction (the no-packed build)

For .so / .o inputs the gfx950 code objects are cut out of the clang offload bundle and disassembled with llvm-objdump."""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
PK = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")
LOWSEL = re.compile(r"\bop_sel:\[[01,]*1[01,]*\]")      # a LOW result fed from a HIGH register


def code_objects(path):
    data = open(path, "rb").read()
    magic, pos, out = b"__CLANG_OFFLOAD_BUNDLE__", 0, []
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return out
        nb = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(nb):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode(errors="replace")
            off += tl
            if "gfx950" in triple and sz:
                out.append(data[i + o:i + o + sz])
        pos = i + 24


def disassemble(blob):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(blob)
        f.flush()
        return subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout


def scan_text(text):
    """-> (packed-FP32 instructions, of them with a low-lane op_sel, {kernel: count})"""
    func, total, low, per = "?", 0, 0, {}
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]* ?<([^>]+)>:$", ln.strip()) or re.match(r"^(_Z\w+):", ln)
        if m:
            func = m.group(1)
            continue
        if PK.search(ln):
            total += 1
            per[func] = per.get(func, 0) + 1
            if LOWSEL.search(ln):
                low += 1
    return total, low, per


def scan(path):
    if path.endswith(".s"):
        return scan_text(open(path).read())
    total, low, per = 0, 0, {}
    for blob in code_objects(path):
        t, l_, p = scan_text(disassemble(blob))
        total += t
        low += l_
        for k, v in p.items():
            per[k] = per.get(k, 0) + v
    return total, low, per


ALLOWED = "conv3x3_h16_kernelIDF16"     # the 16-bit instantiations of the one kernel that keeps packed FP32 (op_sel_hi forms only)

if __name__ == "__main__":
    bad = 0
    args = [a for a in sys.argv[1:] if a != "--strict"]
    strict = len(args) != len(sys.argv) - 1
    for p in args:
        total, low, per = scan(p)
        outside = {k: v for k, v in per.items() if ALLOWED not in k}
        print(f"{p}: {total} packed-FP32 VALU instructions ({low} of them feed a LOW result from a HIGH register) in {len(per)} "
              f"kernels, {sum(outside.values())} outside {ALLOWED}")
        for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:6]:
            print(f"   {v:6d}  {k[:100]}")
        bad += total if strict else low + sum(outside.values())
    sys.exit(1 if bad else 0)
