mkdir -p gpurun_out/r3
{
python3 tools/gemm_shapes_probe.py default
BSVI_GEMM_HALF_BELOW=0 python3 tools/gemm_shapes_probe.py tile128
} > gpurun_out/r3/gemm_shapes_probe.txt 2>&1
cat gpurun_out/r3/gemm_shapes_probe.txt
