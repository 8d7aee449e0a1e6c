BSVI_XGEMM_TALL=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_amortized.py -x -q -k "dense or logreg or exact_data or vae_golden or exact" 2>&1 | tail -3
for t in 1 0; do
for w in cfg4 cfg5; do BSVI_XGEMM_TALL=$t python3 bench.py --workload $w --steps 50 --warmup 5 --other-configs off --traffic off --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tall=$t $w', round(d['ms_per_step']*1e3,1), 'us', round(d['roofline']['achieved'],1), d['all_finite'])"; done; done
