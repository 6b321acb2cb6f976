cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6z
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6z/smoke.txt 2>&1; echo "smoke rc=$?"
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/r6z/all_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r6z/all_gpu_tests.log
cp gpurun_out/parity_margins.txt gpurun_out/r6z/parity_margins_run4.txt
timeout 900 python bench.py > gpurun_out/r6z/bench.json 2> gpurun_out/r6z/bench.err
echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r6z/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'] if 'kernel' in d['roofline'] else '', d['roofline']['frac'], d['roofline'].get('detector'))
def find(o,k):
    if isinstance(o,dict):
        for kk,v in o.items():
            if kk==k: return v
            r=find(v,k)
            if r is not None: return r
print(json.dumps(find(d,'train_iteration')))
P
