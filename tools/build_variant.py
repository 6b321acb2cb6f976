#!/usr/bin/env python3
"""Measurement builds of libadayolo.so / libadaisp.so beside the in-tree ones: build/variants/<name>/lib*.so (they travel to the
GPU box with the snapshot; build/ is git-ignored). usage: build_variant.py <name> <yolo|isp> [extra hipcc flags ...]
Tools pick one up through ADAYOLO_LIB / ADAISP_LIB (adaptiveisp_amd/yolo/_lib.py, adaptiveisp_amd/_lib.py)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adaptiveisp_amd import build as B  # noqa: E402


def main():
    name, which, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    lib = {"yolo": "libadayolo.so", "isp": "libadaisp.so"}[which]
    spec = B.LIBS[lib]
    out = os.path.join(ROOT, "build", "variants", name)
    os.makedirs(out, exist_ok=True)
    hipcc = B._hipcc()
    flags = spec["flags"] + extra

    def one(s):
        obj = os.path.join(out, s.rsplit(".", 1)[0] + ".o")
        src = os.path.join(B.CSRC, s)
        stamp = obj + ".flags"
        line = " ".join(flags)
        deps = [src] + [os.path.join(B.CSRC, h) for h in spec["headers"]]
        if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == line and not B._stale(obj, deps):
            return obj
        B._compile(hipcc, src, obj, flags)
        open(stamp, "w").write(line)
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(one, spec["sources"]))
    target = os.path.join(out, lib)
    r = subprocess.run([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", target, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr)
    print(target)


if __name__ == "__main__":
    main()
