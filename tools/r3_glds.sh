timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_amortized.py -x -q -k "dense or logreg or exact_data or vae_golden or exact" 2>&1 | tail -5
for g in 1 0; do
for w in cfg4 cfg5; do BSVI_XGEMM_GLDS=$g python3 bench.py --workload $w --steps 50 --warmup 5 --other-configs off --traffic off --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('glds=$g $w', round(d['ms_per_step']*1e3,1), 'us', round(d['roofline']['achieved'],1), d['all_finite'])"; done; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 30 --warmup 3 --other-configs off --traffic off --no-cpu-baseline --spinup-ms 0 > /dev/null 2>&1
f=$(find /tmp/prof_cfg4 -name "*kernel_stats.csv" | head -1); head -3 $f | cut -c1-120
