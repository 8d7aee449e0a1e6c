mkdir -p gpurun_out/r3
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "dense or logreg" 2>&1 | tail -15
python3 bench.py --workload cfg4 --steps 50 --warmup 5 --other-configs off --traffic off --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cfg4 exact', round(d['ms_per_step']*1e3,1), 'us', d['roofline']['achieved'], d['all_finite'], d['final_loss'])"
BSVI_DENSE_XGEMM=0 python3 bench.py --workload cfg4 --steps 50 --warmup 5 --other-configs off --traffic off --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cfg4 f32  ', round(d['ms_per_step']*1e3,1), 'us', d['roofline']['achieved'], d['all_finite'], d['final_loss'])"
