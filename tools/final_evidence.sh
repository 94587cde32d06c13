#!/bin/bash
# Round evidence from the tree's built libraries, one GPU box, one pass (run through gpurun): GPU test suite, smoke, the bench lines of the
# configurations, the same-box line of the PREVIOUS round's tree when a copy sits in _old_tree/ (git-ignored), stage timelines, kernel trace by
# shape + lane timeline, GEMM efficiency by shape, the round's micro-benchmarks, HBM counters of the dominant convolution and of the fused
# MSDeformAttn gather.  Output under gpurun_out/final/ (copied into profiles/ as rNN_*).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
T=$PWD/odise_amd/lib/libodise_hip_tools.so
if [ "$1" != "notests" ]; then
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
rocm-smi --showclocks --showpower --showtemp --showperflevel > $O/smi_before.txt 2>&1
python bench.py > $O/bench_full_b4_1024.json 2> $O/bench_full.err; echo "bench rc=$?" >> $O/rc.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inclusive > $O/bench_full_b4_1024_steps20.json 2> $O/bench_full20.err
if [ -d _old_tree ]; then (cd _old_tree && python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inclusive > ../$O/bench_full_b4_1024_steps20_previous_round_tree.json 2> ../$O/bench_prev.err); fi
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inclusive > $O/bench_full_b4_1024_steps20_again.json 2> $O/bench_full20b.err
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-inclusive --pipeline > $O/bench_full_b4_1024_encoder_prefetch.json 2> $O/bench_pipeline.err
python bench.py --stage unet --images 1 > $O/bench_unet_b1.json 2> $O/bench_unet_b1.err
python bench.py --stage unet --images 4 --no-cpu-baseline > $O/bench_unet_b4.json 2> $O/bench_unet_b4.err
python bench.py --stage unet --images 16 --no-cpu-baseline > $O/bench_unet_b16.json 2> $O/bench_unet_b16.err
python bench.py --vocab ade150 --images 8 --no-cpu-baseline --no-inclusive > $O/bench_ade150_b8_1024.json 2> $O/bench_ade150.err
python bench.py --vocab ade847 --size 1280 --images 2 --semantic-only --no-cpu-baseline --no-inclusive > $O/bench_ade847_b2_1280_semantic.json 2> $O/bench_ade847.err
for R in 1 2 3 4 5 6 7; do python bench.py --picture-rank $R --steps 2 --warmup 1 --no-cpu-baseline --no-inclusive 2>&1 >/dev/null | grep "segments per image"; done > $O/bench_rank_rehearsal_segments.txt 2>&1
echo "bench lines done" >> $O/rc.txt
python tools/stage_timeline.py --lanes 2 > $O/stage_timeline_2lanes.txt 2>&1
python tools/stage_timeline.py --lanes 1 > $O/stage_timeline_1lane.txt 2>&1
python tools/stage_timeline.py --lanes 2 --maskclip-passes 2 > $O/stage_timeline_2lanes_maskclip_one_pass.txt 2>&1
python tools/maskclip_bench.py > $O/maskclip_passes.txt 2>&1
python tools/conv_in_bench.py 2>&1 | head -1 > $O/conv_in_kernel.txt
python tools/msda_bench.py 20 4 > $O/msda_variants.txt 2>&1
python tools/clip_gemm_bench.py 5 > $O/clip_gemm_epilogues.txt 2>&1
python tools/attn_unet_bench.py 20 > $O/attention_pipelined.txt 2>&1
python tools/post_bench.py 2>&1 | grep -v "amdgpu.ids" > $O/post_bench.txt
echo "timelines + micro-benchmarks done" >> $O/rc.txt
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inclusive > $O/bench_full_b4_1024_traced.json 2> $O/prof.err
python tools/db_by_shape.py $O/prof/bench_results.db marker 60 > $O/bench_full_by_shape.txt 2>&1
python tools/lane_timeline.py $O/prof/bench_results.db > $O/lane_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_csv -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inclusive > /dev/null 2> $O/prof_csv.err
find $O/prof_csv -name "*kernel_stats.csv" -exec cp {} $O/bench_full_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_unet -o unet -- python bench.py --stage unet --images 16 --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $O/prof_unet.err
find $O/prof_unet -name "*kernel_stats.csv" -exec cp {} $O/unet_b16_kernel_stats.csv \;
ODISE_HIP_LIB=$T ODISE_GEMM_FLAGS=32 rocprofv3 --kernel-trace --output-format csv -d $O/eff -o t -- python tools/gemm_eff.py run 2> $O/eff_gemm.log > /dev/null
python tools/gemm_eff.py join $O/eff/t_kernel_trace.csv $O/eff_gemm.log > $O/gemm_efficiency_by_shape.txt 2>&1
echo "traces done" >> $O/rc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/conv_$c -o c -- python tools/one_conv.py -1 5 16 128 512 512 > $O/conv_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/msda_$c -o c -- python tools/msda_bench.py 3 4 > $O/msda_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/conv_mfma -o c -- python tools/one_conv.py -1 5 16 128 512 512 > $O/conv_mfma.log 2>&1
for d in conv_FETCH_SIZE conv_WRITE_SIZE conv_mfma; do echo "== $d"; python tools/pmc_avg.py $O/$d/c_counter_collection.csv conv3_halo; done > $O/conv_pmc.txt 2>&1
for d in msda_FETCH_SIZE msda_WRITE_SIZE; do for k in msda_fused_kernelILi3ELi4ELi8 msda_fused_kernelILi3ELi4ELi4 msda_forward_kernel msda_prepare_kernel; do echo "== $d $k"; python tools/pmc_avg.py $O/$d/c_counter_collection.csv $k; done; done > $O/msda_pmc.txt 2>&1
python tools/conv_traffic_json.py $O/conv_pmc.txt $O/dominant_conv_traffic.json 16 > /dev/null 2> $O/conv_traffic.err
echo "pmc done" >> $O/rc.txt
# keep the merge small: the raw traces stay on the box
rm -rf $O/eff $O/prof $O/prof_csv $O/prof_unet $O/conv_FETCH_SIZE $O/conv_WRITE_SIZE $O/conv_mfma $O/msda_FETCH_SIZE $O/msda_WRITE_SIZE
cat $O/rc.txt; tail -3 $O/pytest_gpu.log; for f in $O/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['ms_per_step'],2), round(d['value'],2), d['unit'], d.get('roofline',{}).get('frac'))"; done
head -3 $O/lane_timeline.txt; cat $O/conv_pmc.txt; cat $O/msda_pmc.txt; cat $O/bench_rank_rehearsal_segments.txt
