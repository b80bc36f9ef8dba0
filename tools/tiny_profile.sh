cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tinyprof; mkdir -p $O
a="--backbone facebook/deit-tiny-patch16-224 --teachers cdiv --batch 256"
THEIA_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py $a --steps 8 --warmup 2 --no-roofline --no-cpu-baseline --no-selfcheck > $O/prof.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof/*.db | head -1) --csv $O/kernel_stats_serial.csv --top 60 > $O/kernel_stats_serial.txt; rm -rf $O/prof
cd $R; python tools/host_profile.py > $O/hostprof.txt 2>&1
