mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -m gpu -q --timeout=900 -k "conv" > gpurun_out/t6.log 2>&1
grep -E "^E   |FAILED|passed|failed" gpurun_out/t6.log | cut -c1-300 | head -10
for cfg in "0 0" "64 0" "64 128"; do
  set -- $cfg
  export DYNMM_IGEMM_TPIX=$1 DYNMM_IGEMM_TPIX_C64=$2
  [ "$1" = "0" ] && unset DYNMM_IGEMM_TPIX
  [ "$2" = "0" ] && unset DYNMM_IGEMM_TPIX_C64
  DYNMM_BENCH_SHAPES=gpurun_out/shapes6_t$1_c$2.txt python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench6_t$1_c$2.log 2>&1
  echo "TPIX=$1 C64=$2: $(tail -1 gpurun_out/bench6_t$1_c$2.log | cut -c60-170)"
  grep -E "igemm_(fwd|dgrad)" gpurun_out/shapes6_t$1_c$2.txt | grep -E "\(32, (64|128|256|512), (120|60|30|15), (160|80|40|20), (64|128|256|512), 3, 1, 1, 1\)" | cut -c1-120
done
