mkdir -p gpurun_out/r5/ab2
for v in "a:" "b:--traffic off --other-configs off --no-cpu-baseline" "c:--traffic off" "d:--other-configs off" "e:--traffic off --other-configs off --no-cpu-baseline --spinup-ms 100"; do
  tag=${v%%:*}; flags=${v#*:}
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $flags > gpurun_out/r5/ab2/$tag.json 2> gpurun_out/r5/ab2/$tag.err
done
timeout 300 python tools/r5/one_shot_probe.py > gpurun_out/r5/ab2/probe.txt 2>&1
