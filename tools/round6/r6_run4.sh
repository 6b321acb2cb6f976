set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_agent.py -x -q -m gpu -k "pool or fused or episode or golden" > $O/isp_tests.log 2>&1; echo "tests rc=$?" >> $O/isp_tests.log
timeout 600 python tools/isp_step_ab.py noshared=build/variants/poolnoshared/libadaisp.so --ops=0,2,5,7 > $O/isp_pool_shared_ab.txt 2>&1
ADAYOLO_LIB=build/variants/measure/libadayolo.so timeout 300 python tools/bneck_ws_stamps.py > $O/bneck_ws_stamps.txt 2>&1
timeout 600 python tools/eval_graph_prof.py 40 > $O/eval_graph_prof.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eval_stats -o eval -- python3 $R/tools/eval_graph_prof.py 40 > $O/eval_rocprof.log 2>&1
cp $O/eval_stats/eval_kernel_stats.csv $O/eval_config3_kernel_stats.csv
export ADAYOLO_LIB=$R/build/variants/measure/libadayolo.so
for V in 50 56; do
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/ldspmc -o v$V -- python3 $R/tools/conv_one.py 92 160 128 256 3 1 $V 6 > $O/ldspmc_$V.log 2>&1
done
unset ADAYOLO_LIB
cd $R
python3 - $O <<'PY'
import csv, glob, os, sys, collections
o = sys.argv[1]
for p in sorted(glob.glob(os.path.join(o, "ldspmc", "*_counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if "k_conv_pp" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        print(os.path.basename(p), k, m, "conflict frac", m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1)))
PY
tail -3 $O/isp_tests.log; cat $O/isp_pool_shared_ab.txt | tail -30; cat $O/bneck_ws_stamps.txt; cat $O/eval_graph_prof.txt
