# everything profiles/r0N_* of the default bench is made of, in one call on one box (see profiles/README.md).  usage: tools/round_end_profiles.sh [r06]
# Order: PMC traffic + kernel traces first (their summary goes into profiles/ so that the bench lines taken right after carry
# roofline.traffic and roofline.rocprof_kernel_trace from the same box, same sources), then the bench lines, then the other workloads.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
bash $R/tools/pmc_bench_traffic.sh $O/pmc > $O/pmc.log 2>&1
rm -rf $O/pmc/FETCH_SIZE $O/pmc/WRITE_SIZE
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_regime -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-roofline --no-cpu-baseline --no-selfcheck > $O/prof_regime.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_regime/*.db | head -1) --csv $O/kernel_stats.csv --top 45 > $O/kernel_stats.txt; rm -rf $O/prof_regime
THEIA_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_serial -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-roofline --no-cpu-baseline --no-selfcheck > $O/prof_serial.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_serial/*.db | head -1) --csv $O/kernel_stats_serial.csv --top 45 > $O/kernel_stats_serial.txt; rm -rf $O/prof_serial
python $R/tools/add_trace_to_traffic.py $O/pmc/traffic.json $O/kernel_stats.csv $O/kernel_stats_serial.csv
cp $O/pmc/traffic.json $R/profiles/${TAG}_bench_pmc_traffic.json; cp $O/pmc/summary.txt $R/profiles/${TAG}_bench_pmc_traffic.txt
cp $O/pmc/traffic.json $O/${TAG}_bench_pmc_traffic.json; cp $O/pmc/summary.txt $O/${TAG}_bench_pmc_traffic.txt
python $R/tools/hbm_fractions.py $O/pmc/traffic.json $O/kernel_stats_serial.csv > $O/hbm_fractions.txt 2>&1
cd $R; THEIA_BENCH_GEMM_TABLE=1 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20steps.json 2> /dev/null; cat $O/bench_20steps.json | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --teacher-dtype fp32 > $O/bench_20steps_fp32_targets.json 2> /dev/null; cat $O/bench_20steps_fp32_targets.json | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph > $O/bench_20steps_graph.json 2> /dev/null
python bench.py --mode forward_feature > $O/forward_feature_b4096.json 2> /dev/null
python bench.py --backbone facebook/deit-small-patch16-224 --teachers cdiv --batch 16 --steps 50 --warmup 5 --no-cpu-baseline > $O/small_cdiv_b16_eager.json 2> /dev/null
python bench.py --backbone facebook/deit-small-patch16-224 --teachers cdiv --batch 16 --steps 50 --warmup 5 --no-cpu-baseline --graph > $O/small_cdiv_b16_graph.json 2> /dev/null
python bench.py --backbone facebook/deit-tiny-patch16-224 --teachers cdiv --batch 256 --steps 30 --warmup 5 --no-cpu-baseline --graph > $O/tiny_cdiv_b256_graph.json 2> /dev/null
# the other workloads: serial kernel traces, PMC traffic (into profiles/ first), then their self-checked lines
cd /tmp
for cfg in "tiny_cdiv_b256:--backbone facebook/deit-tiny-patch16-224 --teachers cdiv --batch 256" "small_5t_b256:--backbone facebook/deit-small-patch16-224 --batch 256" "small_5t_b256_fp8:--backbone facebook/deit-small-patch16-224 --batch 256 --precision fp8"; do n=${cfg%%:*}; a=${cfg#*:}
  THEIA_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o bench -- python $R/bench.py $a --steps 8 --warmup 2 --no-roofline --no-cpu-baseline --no-selfcheck > $O/prof_$n.log 2>&1
  python $R/tools/rocpd_stats.py $(ls $O/prof_$n/*.db | head -1) --csv $O/kernel_stats_serial_$n.csv --top 45 > $O/kernel_stats_serial_$n.txt; rm -rf $O/prof_$n
  PMC_BENCH_ARGS="$a" bash $R/tools/pmc_bench_traffic.sh $O/pmc_$n > $O/pmc_$n.log 2>&1
  rm -rf $O/pmc_$n/FETCH_SIZE $O/pmc_$n/WRITE_SIZE
  cp $O/pmc_$n/traffic.json $R/profiles/${TAG}_bench_pmc_traffic_$n.json; cp $O/pmc_$n/traffic.json $O/${TAG}_bench_pmc_traffic_$n.json
  python $R/tools/hbm_fractions.py $O/pmc_$n/traffic.json $O/kernel_stats_serial_$n.csv > $O/hbm_fractions_$n.txt 2>&1
done
cd $R
python bench.py --backbone facebook/deit-tiny-patch16-224 --teachers cdiv --batch 256 --steps 30 --warmup 5 --no-cpu-baseline > $O/tiny_cdiv_b256.json 2> $O/tiny.err
python bench.py --backbone facebook/deit-small-patch16-224 --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/small_5t_b256_bf16.json 2> $O/small.err
# BASELINE configs[3]: fp8 against bf16, arms alternating (64 steps: two scale-refresh steps inside the timed region)
for r in 1 2 3; do
  python bench.py --backbone facebook/deit-small-patch16-224 --batch 256 --steps 64 --warmup 5 --no-cpu-baseline --precision fp8 > $O/small_5t_b256_fp8_$r.json 2> $O/small_fp8.err
  python bench.py --backbone facebook/deit-small-patch16-224 --batch 256 --steps 64 --warmup 5 --no-cpu-baseline > $O/small_5t_b256_bf16_$r.json 2> /dev/null
done
cp $O/small_5t_b256_fp8_3.json $O/small_5t_b256_fp8.json
python bench.py --precision fp8 --steps 64 --warmup 5 --no-cpu-baseline > $O/base_5t_b128_fp8.json 2> /dev/null
python bench.py --steps 64 --warmup 5 --no-cpu-baseline > $O/base_5t_b128_bf16_64steps.json 2> /dev/null
python -c "
import json
for f in ['small_5t_b256_fp8_%d' % r for r in (1, 2, 3)] + ['small_5t_b256_bf16_%d' % r for r in (1, 2, 3)] + ['base_5t_b128_fp8', 'base_5t_b128_bf16_64steps']:
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('traffic'))
"
python -c "
import json
for f in ('tiny_cdiv_b256','small_5t_b256_bf16'):
    d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['bound'], r['achieved'], r['frac'], 'traffic', r.get('traffic'), r.get('traffic_source'), 'alg', r.get('traffic_algorithmic'))
"
timeout 300 python tools/determinism_check.py > $O/determinism.txt 2>&1; tail -n 1 $O/determinism.txt
python -m pytest tests -m gpu -q -s 2>&1 | grep -E "^\.*\[|passed|failed|Error|error|FAILED" > $O/pytest.txt; tail -n 5 $O/pytest.txt
