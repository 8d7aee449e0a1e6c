#!/bin/bash
# iteration time of the in-kernel loop (cfg 1 program) against the number of waves of the workgroup
for n in 64 128 256 300 384 448 512; do
  python bench.py --no-cpu-baseline --samples $n 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['number_samples_per_gpu'], 'samples', d['config']['grid']['n_threads'], 'threads', round(d['device_ms_per_step']*1e3,3), 'us')"
done
