cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05cfg; mkdir -p $O
PMC_BENCH_ARGS="--backbone facebook/deit-tiny-patch16-224 --teachers cdiv --batch 256" bash tools/pmc_bench_traffic.sh $O/pmc_tiny > $O/pmc_tiny.log 2>&1
cp $O/pmc_tiny/traffic.json $R/profiles/r05_bench_pmc_traffic_tiny_cdiv_b256.json; cp $O/pmc_tiny/traffic.json $O/r05_bench_pmc_traffic_tiny_cdiv_b256.json
PMC_BENCH_ARGS="--backbone facebook/deit-small-patch16-224 --batch 256" bash tools/pmc_bench_traffic.sh $O/pmc_small > $O/pmc_small.log 2>&1
cp $O/pmc_small/traffic.json $R/profiles/r05_bench_pmc_traffic_small_5t_b256.json; cp $O/pmc_small/traffic.json $O/r05_bench_pmc_traffic_small_5t_b256.json
rm -rf $O/pmc_tiny/FETCH_SIZE $O/pmc_tiny/WRITE_SIZE $O/pmc_small/FETCH_SIZE $O/pmc_small/WRITE_SIZE
cd $R
python bench.py --backbone facebook/deit-tiny-patch16-224 --teachers cdiv --batch 256 --steps 30 --warmup 5 --no-cpu-baseline > $O/tiny_cdiv_b256.json 2> $O/tiny.err
python bench.py --backbone facebook/deit-small-patch16-224 --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/small_5t_b256_bf16.json 2> $O/small.err
python -c "
import json
for f in ('tiny_cdiv_b256','small_5t_b256_bf16'):
    d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, d['value'], d['ms_per_step'], 'traffic', r.get('traffic'), r.get('traffic_source'), 'alg', r.get('traffic_algorithmic'))
"
