DYNMM_BENCH_SHAPES=gpurun_out/shapes_auto.txt python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
DYNMM_PRECISION=fp32 DYNMM_BENCH_SHAPES=gpurun_out/shapes_fp32.txt python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for f in auto fp32; do echo == $f; grep -E "igemm_dgrad" gpurun_out/shapes_$f.txt | head -8 | cut -c1-110; done
