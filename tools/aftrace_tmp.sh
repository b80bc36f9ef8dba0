cd $GRAFT_REPO_ROOT
cp theia_amd/lib/libtheia_hip.so /tmp/keep.so
python -m pytest tests/test_ops_gpu.py -q -k attention 2>&1 | tail -2
for r in 1 2; do for l in prev ds2; do cp build/alt/libtheia_$l.so theia_amd/lib/libtheia_hip.so; echo "== $l"; python tools/attn_bench.py 2>&1 | grep -i "bwd" | head -4; python tools/attn_bench.py --b 256 --h 3 2>&1 | grep -i "bwd" | head -2; done; done
cp build/alt/libtheia_aftrace.so theia_amd/lib/libtheia_hip.so
python tools/experiments/attn_bwd_phase_trace.py
cp /tmp/keep.so theia_amd/lib/libtheia_hip.so
