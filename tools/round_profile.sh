#!/bin/bash
# usage (on the GPU box): tools/round_profile.sh TAG -- the bench lines, the rocprofv3 kernel stats of the same command and
# the two PMC passes a round's profiles/ entries are made from.  Outputs under gpurun_out/TAG/.
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
export TMPDIR=/tmp
# the PMC passes first (their own rocprofv3 runs: counters are never combined with traces), so that the bench lines below read this build's
# roofline.traffic from profiles/pmc_traffic.json
export PMC_B=64 PMC_DTYPE=fp16x3
bash tools/pmc_pass.sh ${tag}_x3 FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" 2>&1 | tail -3
cd $R
python tools/pmc_traffic.py gpurun_out/${tag}_x3 $tag 64 > gpurun_out/${tag}_x3/traffic.md 2>&1
python tools/pmc_mfma.py gpurun_out/${tag}_x3 ${tag}_fp16x3 > /dev/null 2>&1
cp profiles/pmc_traffic.json profiles/${tag}_pmc_hbm_traffic.md profiles/${tag}_fp16x3_pmc_mfma_util.* gpurun_out/${tag}_x3/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err      # the driver's command
timeout 900 python tools/noisy_pipeline.py 2048 $O/noisy_pipeline_2048.json > /dev/null 2>&1
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_c3_bf16.json 2> $O/bench_c3_bf16.err
python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --workload c4 --line-workload random --no-cpu-baseline --no-parity > $O/bench_c4_random_lines.json 2> $O/bench_c4_random_lines.err
python bench.py --dtype fp8 --no-cpu-baseline > $O/bench_c3_fp8.json 2> $O/bench_c3_fp8.err
python bench.py --dtype fp8 --size 1080p --batch 64 --no-cpu-baseline > $O/bench_c5_fp8.json 2> $O/bench_c5_fp8.err
python bench.py --dtype bf16 --size 1080p --batch 64 --no-cpu-baseline > $O/bench_c5_bf16.json 2> $O/bench_c5_bf16.err
python bench.py --size 1080p --batch 64 --no-cpu-baseline > $O/bench_c5_fp16x3.json 2> $O/bench_c5_fp16x3.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_c3_traced.json 2> $O/bench_c3_traced.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*.csv" -size +2M -delete
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof16 -o trace -- python $R/bench.py --dtype bf16 --no-cpu-baseline --no-parity > $O/bench_c3_bf16_traced.json 2> $O/bench_c3_bf16_traced.err
find $O/prof16 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bf16.csv \;
find $O/prof16 -name "*.csv" -size +2M -delete
# the reference-precision engine, profiled the same way (the fp32 object of the main line comes from the same step)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof32 -o trace -- python $R/bench.py --dtype fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-parity > $O/bench_c3_fp32_traced.json 2> $O/bench_c3_fp32_traced.err
find $O/prof32 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_fp32.csv \;
find $O/prof32 -name "*.csv" -size +2M -delete
cd $R
# (PMC passes: tools/pmc_pass.sh in a call of their own BEFORE this script -- `python tools/pmc_traffic.py` turns them into
#  profiles/pmc_traffic.json, which the bench lines above read for roofline.traffic)
python tools/parity_large.py 4096 0.35 > $O/parity_large_4096.log 2>&1
python tools/latency.py > $O/latency.log 2>&1
SNCAL_BBX_TRACE=$O/bbx.bin python tools/dev/bbx_trace_run.py > /dev/null 2>&1; python tools/bbx_trace.py $O/bbx.bin > $O/bbx_trace.txt 2>&1; rm -f $O/bbx.bin
head -c 1500 $O/bench_c3.json; echo; head -c 600 $O/bench_c3_bf16.json; echo; head -c 600 $O/bench_c4.json; echo; head -c 600 $O/bench_c3_fp8.json; echo; head -c 600 $O/bench_c5_fp8.json; echo; head -c 600 $O/bench_c5_bf16.json; echo
cat $O/noisy_pipeline_2048.json; echo; head -8 $O/kernel_stats.csv; tail -12 $O/parity_large_4096.log; cat $O/bbx_trace.txt
