cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm" > $O/pytest_ops.log 2>&1; echo "ops rc=$?" > $O/rc.txt
python -m pytest tests/test_gpu_extractor.py tests/test_gpu_golden.py -m gpu -q -x > $O/pytest_ext.log 2>&1; echo "ext rc=$?" >> $O/rc.txt
T=$PWD/odise_amd/lib/libodise_hip_tools.so
for i in 1 2; do
  ODISE_HIP_LIB=$T python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inclusive > $O/bench_fold$i.json 2> $O/bench_fold$i.err; python -c "import json; d=json.load(open('$O/bench_fold$i.json')); print('fold$i', d['ms_per_step'], d['value'])" >> $O/rc.txt
  ODISE_CLIP_LN_KERNELS=1 ODISE_HIP_LIB=$T python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inclusive > $O/bench_ln$i.json 2> $O/bench_ln$i.err; python -c "import json; d=json.load(open('$O/bench_ln$i.json')); print('lnkern$i', d['ms_per_step'], d['value'])" >> $O/rc.txt
done
grep -E "passed|failed|Error|error" $O/pytest_ops.log $O/pytest_ext.log | tail -12; cat $O/rc.txt
