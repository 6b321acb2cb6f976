"""Build the HIP shared libraries for gfx950 in-tree (csrc/*.so travel to the GPU box with the repo
snapshot; they are git-ignored). `python -m adaptiveisp_amd.build` or __graft_entry__.build()."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
ARCH = os.environ.get("ADAISP_ARCH", "gfx950")

# -ffp-contract=off: the ISP filters reproduce the reference's mul/add rounding sequence; fused
# multiply-adds are written explicitly (fmaf) where they are wanted.
ISP_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]

LIBS = {
    "libadaisp.so": dict(
        sources=["isp_pointwise.hip", "isp_conv.hip", "isp_nlm.hip", "isp_pool.hip", "isp_demosaic.hip", "isp_backward.hip",
                 "isp_policy.hip", "isp_trunk_train.hip", "isp_rl_train.hip", "isp_heads_train.hip", "isp_api.hip"],
        headers=["isp_internal.h", "../../include/adaisp.h"],
        flags=ISP_FLAGS),
    "libadayolo.so": dict(
        sources=["yolo_conv_dma.hip", "yolo_conv_dma2.hip", "yolo_conv_small.hip", "yolo_conv_pp.hip", "yolo_conv_chain.hip", "yolo_bneck.hip", "yolo_bneck_ws.hip", "yolo_conv_pp128.hip", "yolo_conv_k1.hip", "yolo_conv_pq.hip", "yolo_conv_ws.hip", "yolo_misc.hip", "yolo_stem_down.hip", "yolo_nms.hip", "yolo_train.hip", "yolo_loss.hip", "yolo_api.hip"],
        # (yolo_conv_chain.hip compiles the tile bodies of yolo_conv_pp.hip / yolo_conv_pp128.hip into its own translation unit:
        # a change to either rebuilds it)
        headers=["yolo_internal.h", "yolo_chain.h", "yolo_conv_pp.hip", "yolo_conv_pp128.hip", "../../include/adayolo.h"],
        flags=["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("ADAYOLO_EXTRA_FLAGS", "").split()),
}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP libraries cannot be built (and there is no fallback path)")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(hipcc, src, obj, flags):
    cmd = [hipcc, f"--offload-arch={ARCH}", *flags, "-c", src, "-o", obj]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force=False, verbose=True):
    hipcc = _hipcc()
    built = []
    for lib, spec in LIBS.items():
        hdrs = [os.path.join(CSRC, h) for h in spec["headers"]]
        # objects built with other flags (a measurement build: ADAYOLO_EXTRA_FLAGS=-DADAYOLO_MEASURE) are stale whatever their
        # time stamps say: the flag line of the last build is kept beside them
        flagfile = os.path.join(CSRC, "." + lib + ".flags")
        flagline = " ".join([ARCH, *spec["flags"]])
        if not os.path.exists(flagfile) or open(flagfile).read() != flagline:
            force_lib = True
        else:
            force_lib = force
        jobs = []
        for s in spec["sources"]:
            src, obj = os.path.join(CSRC, s), os.path.join(CSRC, s.rsplit(".", 1)[0] + ".o")
            if force_lib or _stale(obj, [src, *hdrs]):
                jobs.append((src, obj))
        with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
            list(ex.map(lambda j: _compile(hipcc, j[0], j[1], spec["flags"]), jobs))
        objs = [os.path.join(CSRC, s.rsplit(".", 1)[0] + ".o") for s in spec["sources"]]
        target = os.path.join(CSRC, lib)
        if force_lib or jobs or _stale(target, objs):
            r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", target, *objs],
                               cwd=CSRC, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed for {lib}:\n{r.stdout}\n{r.stderr}")
        with open(flagfile, "w") as f:
            f.write(flagline)
        built.append(target)
        if verbose:
            print(f"[adaptiveisp_amd.build] {target} ({len(jobs)} object(s) rebuilt)")
    return built


if __name__ == "__main__":
    build(force="--force" in sys.argv)
