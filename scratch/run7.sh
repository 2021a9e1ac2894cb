mkdir -p gpurun_out
for c in 0 128; do
  export DYNMM_IGEMM_TPIX_C64=$c; [ "$c" = "0" ] && unset DYNMM_IGEMM_TPIX_C64
  DYNMM_BENCH_SHAPES=gpurun_out/shapes7_c$c.txt python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench7_c$c.log 2>&1
  echo "C64=$c: $(tail -1 gpurun_out/bench7_c$c.log | cut -c60-170)"
  grep -E "igemm_(fwd|dgrad)" gpurun_out/shapes7_c$c.txt | grep -E "\(32, (64|128), (120|60), (160|80), (64|128), 3, [13], 1, 1\)" | cut -c1-120
done
