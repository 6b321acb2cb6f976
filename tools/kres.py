#!/usr/bin/env python3
"""Per-kernel registers / scratch / LDS of one .hip file: kres.py file.hip [filter] [-- hipcc flags]"""
import re, subprocess, sys
args = sys.argv[1:]
flags = args[args.index("--") + 1:] if "--" in args else []
args = args[:args.index("--")] if "--" in args else args
src, filt = args[0], (args[1] if len(args) > 1 else "")
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", "-", src, *flags],
                     capture_output=True, text=True)
if asm.returncode:
    sys.exit(asm.stderr[-3000:])
for blk in asm.stdout.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if filt in name:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("adaisp::(anonymous namespace)::", "").replace("void ", "")
        print(f"{dem:48s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
