cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_fullsize_1280.py tests/test_gpu_golden.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -s > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r3b/rc.txt
ODISE_HIP_LIB=$PWD/odise_amd/lib/libodise_hip_tools.so timeout 600 python tools/epi16_ab.py > gpurun_out/r3b/epi16_ab.log 2>&1; echo "epi16 rc=$?" >> gpurun_out/r3b/rc.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inclusive > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err; echo "bench rc=$?" >> gpurun_out/r3b/rc.txt
tail -5 gpurun_out/r3b/pytest.log; cat gpurun_out/r3b/rc.txt; tail -3 gpurun_out/r3b/epi16_ab.log; cat gpurun_out/r3b/bench.json
