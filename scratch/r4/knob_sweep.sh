# scheduling knobs after the Winograd kernels (same box, one process each)
B="python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 20 --warmup 5"
run() { echo -n "$1: "; env $1 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run A=0
run DYNMM_WGRAD_STREAMS=1
run DYNMM_WGRAD_STREAMS=3
run DYNMM_WGRAD_V6_BLOCKS=256
run DYNMM_WGRAD_V6_BLOCKS=1024
run DYNMM_WGRAD_GROUP=2
run DYNMM_WGRAD_GROUP=6
run DYNMM_WINO_TILE=1
run DYNMM_WGRAD_WINO=0
run A=1
