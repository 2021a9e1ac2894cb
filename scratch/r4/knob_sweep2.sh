# joint sweep: weight-gradient streams x workgroups per v6 launch (how much of each CU the weight gradients may hold while the
# dependent chain's BatchNorm / input-gradient kernels run beside them)
B="python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 20 --warmup 5"
run() { echo -n "$*: "; env "$@" $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run A=0
run DYNMM_WGRAD_STREAMS=1 DYNMM_WGRAD_V6_BLOCKS=256
run DYNMM_WGRAD_STREAMS=1 DYNMM_WGRAD_V6_BLOCKS=384
run DYNMM_WGRAD_STREAMS=1 DYNMM_WGRAD_V6_BLOCKS=512
run DYNMM_WGRAD_STREAMS=2 DYNMM_WGRAD_V6_BLOCKS=128
run DYNMM_WGRAD_STREAMS=2 DYNMM_WGRAD_V6_BLOCKS=192
run DYNMM_WGRAD_STREAMS=3 DYNMM_WGRAD_V6_BLOCKS=128
run A=1
