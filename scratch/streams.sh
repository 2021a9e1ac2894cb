for cfg in "0 0 " "1 0 " "1 1 " "0 1 " "1 1 --no-graph" "1 0 --no-graph" "0 1 --no-graph" "0 0 --no-graph"; do
  set -- $cfg
  echo "dual=$1 async_wgrad=$2 $3: $(DYNMM_DUAL_STREAM=$1 DYNMM_ASYNC_WGRAD=$2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing $3 2>&1 | tail -1 | cut -c60-170)"
done
