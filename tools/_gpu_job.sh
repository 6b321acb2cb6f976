python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r2_t3.log
tail -3 gpurun_out/r2_t3.log
for ng in 1 2; do for rs in 15 30 60 120; do
  echo "== NG=$ng RS=$rs"
  ADAISP_CONV_NG=$ng ADAISP_CONV_RS=$rs python tools/kernel_times.py --shape 8,720,1280 --ops E,Shr,USM 2>&1 | grep -E "^(Shr|USM|E ) " | sed 's/^/C2 /'
  ADAISP_CONV_NG=$ng ADAISP_CONV_RS=$rs python tools/kernel_times.py --shape 4,2160,3840 --iters 10 --ops E,Shr,USM 2>&1 | grep -E "^(Shr|USM|E ) " | sed 's/^/4K /'
done; done > gpurun_out/r2_stencil_sweep.txt 2>&1
cat gpurun_out/r2_stencil_sweep.txt
