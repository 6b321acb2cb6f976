cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh round6final > gpurun_out/round6final_refresh.log 2>&1
tail -3 gpurun_out/round6final_refresh.log
python - <<'P'
import json
d=json.loads(open('gpurun_out/round6final_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('detector'))
P
rm -rf gpurun_out/round6final_stats gpurun_out/round6final_pmc
