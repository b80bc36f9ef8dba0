# everything profiles/r0N_* of the default bench is made of, in one call on one box (see profiles/README.md)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/f2; mkdir -p $O
bash $R/tools/pmc_bench_traffic.sh $O/pmc > $O/pmc.log 2>&1
cp $O/pmc/traffic.json $R/profiles/r03_bench_pmc_traffic.json; cp $O/pmc/summary.txt $R/profiles/r03_bench_pmc_traffic.txt
rm -rf $O/pmc/FETCH_SIZE $O/pmc/WRITE_SIZE
cd $R; THEIA_BENCH_GEMM_TABLE=1 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_20steps.json 2> /dev/null; cat $O/bench_20steps.json | cut -c1-200
python bench.py --backbone facebook/deit-tiny-patch16-224 --teachers cdiv --batch 256 --steps 20 --warmup 5 --no-roofline > $O/tiny_cdiv.json 2> $O/tiny.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_regime -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-roofline --no-cpu-baseline --no-selfcheck > $O/prof_regime.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_regime/*.db | head -1) --csv $O/kernel_stats.csv --top 45 > $O/kernel_stats.txt; rm -rf $O/prof_regime
THEIA_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_serial -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-roofline --no-cpu-baseline --no-selfcheck > $O/prof_serial.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_serial/*.db | head -1) --csv $O/kernel_stats_serial.csv --top 45 > $O/kernel_stats_serial.txt; rm -rf $O/prof_serial
cd $R; timeout 300 python tools/determinism_check.py > $O/determinism.txt 2>&1; tail -n 1 $O/determinism.txt
python -m pytest tests -m gpu -q -s 2>&1 | grep -E "^\.*\[|passed|failed|Error|error|FAILED" > $O/pytest.txt; tail -n 5 $O/pytest.txt
