#!/usr/bin/env bash
# Runs ON THE GPU BOX: kernel trace of the RL training loop as it ships (python -m adaptiveisp_amd.train, graph mode by default) and a
# TIMELINE of one steady-state iteration: busy / idle time, the largest idle gaps with the kernels around them, and per 100 us
# bucket which kernel families ran. Usage: gpurun -- 'bash tools/train_timeline.sh [extra env assignments for the run]'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R${PYTHONPATH:+:$PYTHONPATH}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/train_timeline" -o tl -- python3 -m adaptiveisp_amd.train --iters 40 --warmup 10 > "$OUT/train_timeline.log" 2>&1
tail -1 "$OUT/train_timeline.log" | cut -c1-200
python3 - "$(find $OUT/train_timeline -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def fam(n):
    n = n.replace("(anonymous namespace)::", "")
    for pat, f in (("k_td_", "td"), ("k_detloss", "detloss"), ("k_adam|k_gradsq|k_gradnorm", "adam"), ("k_tconv|k_tbn|k_twgrad|k_tdgrad|k_t[a-z]+_", "trunk"),
                   ("k_policy_tail|k_critic_planes|k_image_stats", "rl-small"), ("k_silu|k_zero_insert|k_upsample|k_image_grad|k_stem", "det-elem"),
                   ("k_conv|conv_tile|k_bneck|k_pq|k_pp|splitk", "det-conv"), ("k_nlm|k_pointwise|k_conv_rows|k_pool64|isp|k_param|k_sharpen|k_usm", "isp"),
                   ("Cijk|rocblas|gemm|Gemm", "gemm"), ("at::native|elementwise|reduce_kernel|vectorized|index|cat|CatArray", "aten")):
        if re.search(pat, n):
            return f
    return "other"
anch = [i for i, r in enumerate(rows) if "k_td_fwd" in r["Kernel_Name"]]
a0, a1 = anch[-6], anch[-5]
seg = rows[a0:a1]
t0 = int(seg[0]["Start_Timestamp"])
span = int(rows[a1]["Start_Timestamp"]) - t0
ivs = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"].split("(")[0][-50:], fam(r["Kernel_Name"])) for r in seg)
busy, last_end, gaps, prev = 0, 0, [], "(iteration anchor)"
for s, e, n, f in ivs:
    if s > last_end:
        gaps.append((s - last_end, last_end, prev, n))
    if e > last_end:
        busy += e - max(s, last_end)
        last_end, prev = e, n
print(f"one iteration (k_td_fwd to k_td_fwd): span {span/1e6:.3f} ms, some kernel running {busy/1e6:.3f} ms, nothing running {(span-busy)/1e6:.3f} ms, {len(seg)} kernels")
print("largest gaps with nothing running:")
for g, at, p, n in sorted(gaps, reverse=True)[:12]:
    print(f"   {g/1e3:7.1f} us at +{at/1e6:.3f} ms   after {p}   before {n}")
g, at, _, _ = max(gaps)
print(f"around the largest gap (+{at/1e6:.3f} ms): start offset us, duration us, queue, kernel")
for r in seg:
    s0, e0 = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if at - 60000 <= s0 <= at + g + 120000:
        print(f"   {s0/1e3:9.1f} {(e0-s0)/1e3:7.1f}  q{r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'].split('(')[0][-70:]}")
import glob, os
mc = glob.glob(os.path.join(os.path.dirname(sys.argv[1]), "*memory_copy_trace.csv"))
if mc:
    cp = list(csv.DictReader(open(mc[0])))
    print("memory copies of the iteration (start offset us, duration us, direction, bytes):")
    for r in cp:
        s0, e0 = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        if 0 <= s0 <= span:
            print(f"   {s0/1e3:9.1f} {(e0-s0)/1e3:7.1f}  {r.get('Direction', r.get('Kind', '?'))}  {r.get('Size', r.get('Bytes', '?'))}")
print("per 200 us: kernel-time by family (us; > 200 = overlap of streams)")
nb = span // 200000 + 1
buckets = [collections.Counter() for _ in range(nb)]
for s, e, n, f in ivs:
    b = s // 200000
    while s < e and b < nb:
        hi = min(e, (b + 1) * 200000)
        buckets[b][f] += hi - s
        s, b = hi, b + 1
for b, c in enumerate(buckets):
    print(f"   +{b*0.2:4.1f} ms  " + "  ".join(f"{k} {v/1e3:.0f}" for k, v in c.most_common()))
if os.environ.get("TIMELINE_ALL"):
    with open(os.environ["TIMELINE_ALL"], "w") as f:
        for r in seg:
            s0, e0 = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            f.write(f"{s0/1e3:9.1f} {(e0-s0)/1e3:7.1f} q{r.get('Queue_Id', '?')} {fam(r['Kernel_Name']):9s} {r['Kernel_Name'][:150]}\n")
fams = collections.Counter()
for s, e, n, f in ivs:
    fams[f] += e - s
print("kernel time by family (ms): " + "  ".join(f"{k} {v/1e6:.3f}" for k, v in fams.most_common()))
oth = collections.Counter()
for s, e, n, f in ivs:
    if f in ("other", "aten"):
        oth[n] += e - s
print("largest 'aten' / 'other' names: " + "; ".join(f"{k} {v/1e3:.0f}us" for k, v in oth.most_common(10)))
PY
