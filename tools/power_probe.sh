cd $GRAFT_REPO_ROOT; O=gpurun_out/power; mkdir -p $O
rocm-smi --showpower --showmaxpower --showclocks --showperflevel --showpowerprofile > $O/idle.txt 2>&1
( for i in $(seq 1 40); do echo "--- t=$i"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk"; sleep 0.5; done ) > $O/during_bench.txt 2>&1 &
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-selfcheck --no-roofline > $O/bench.json 2> $O/bench.err
wait
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'])"
grep -E "Power|sclk" $O/idle.txt | head -12
echo ==== during
grep -E "Power|sclk" $O/during_bench.txt | sed -n 30,60p
