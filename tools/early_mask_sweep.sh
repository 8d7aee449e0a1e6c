for m in ${MASKS:-3 2 1 0 19 15 18 6}; do
  echo -n "mask $m: "
  BSVI_SPEC_DEFINES="#define SPEC_EARLY_MASK ${m}u" python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1e3,3))"
done
