mkdir -p gpurun_out
for cfg in "0 0" "64 0" "128 0" "0 32" "64 32" "128 32"; do
  set -- $cfg
  export DYNMM_IGEMM_TPIX=$1 DYNMM_IGEMM_BK=$2
  [ "$1" = "0" ] && unset DYNMM_IGEMM_TPIX
  [ "$2" = "0" ] && unset DYNMM_IGEMM_BK
  DYNMM_BENCH_SHAPES=gpurun_out/shapes_t$1_b$2.txt python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_t$1_b$2.log 2>&1
  echo "TPIX=$1 BK=$2: $(tail -1 gpurun_out/bench_t$1_b$2.log | cut -c60-140)"
  grep -E "igemm_(fwd|dgrad)" gpurun_out/shapes_t$1_b$2.txt | grep -E "\(32, (64|128|256|512), (120|60|30|15), (160|80|40|20), (64|128|256|512), 3, 1, 1, 1\)" | cut -c1-120
done
