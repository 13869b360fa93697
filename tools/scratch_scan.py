#!/usr/bin/env python3
"""Kernels of the built library that use SCRATCH (private segment: register spills / stack), from the code objects' metadata
notes.  usage: python tools/scratch_scan.py [libmadm_hip.so]    (exit code 1 if any kernel uses scratch)

Why it matters here: a spilling kernel's reloads are VMEM traffic inside hand-scheduled loops, and the only two unexplained
numerical events of rounds 4-5 involved kernels with scratch (the rejected weights-direct 16 x 16 conv -- not bit-stable under the
multi-stream pipeline -- and the f32 instantiation of the 16 x 16 conv, whose generic epilogue spilled 400 .. 750 VGPRs: the f32
eval forward that once mismatched on a fresh box ran on it).  tests/test_host.py keeps the library scratch-free."""
import importlib.util
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernels(path):
    spec = importlib.util.spec_from_file_location("isa_pk_scan", os.path.join(HERE, "isa_pk_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = {}
    for i, blob in enumerate(mod.code_objects(path)):
        fn = f"/tmp/_scratch_scan_{os.getpid()}_{i}.o"
        with open(fn, "wb") as f:
            f.write(blob)
        try:
            out = subprocess.run([READELF, "--notes", fn], capture_output=True, text=True).stdout
        finally:
            os.unlink(fn)
        cur = {}
        for ln in out.splitlines():
            ln = ln.strip().lstrip("- ").strip()
            if ln.startswith(".name:"):
                cur["name"] = ln.split(":", 1)[1].strip()
            for k in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count"):
                if ln.startswith("." + k + ":"):
                    cur[k] = int(ln.split(":")[1])
            if ln.startswith(".wavefront_size:") and "name" in cur:
                rows[cur["name"]] = dict(cur)
                cur = {}
    return rows


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "..", "madm_amd", "libmadm_hip.so")
    rows = kernels(path)
    bad = {k: v for k, v in rows.items() if v.get("private_segment_fixed_size", 0) > 0 or v.get("vgpr_spill_count", 0) > 0}
    print(f"{path}: {len(rows)} kernels, {len(bad)} with scratch")
    for k, v in sorted(bad.items(), key=lambda kv: -kv[1].get("private_segment_fixed_size", 0)):
        print(f"   {v.get('private_segment_fixed_size', 0):5d} B/lane, {v.get('vgpr_spill_count', 0):4d} spilled VGPRs, {v.get('vgpr_count')} VGPRs  {k[:120]}")
    sys.exit(1 if bad else 0)
