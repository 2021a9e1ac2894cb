# step-level sweep of the scheduling knobs after the three-tap weight-gradient kernel (one box, --no-extra)
run() { printf "%-44s" "$*"; env "$@" timeout 250 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run A=0
run DYNMM_WGRAD_STREAMS=1
run DYNMM_WGRAD_STREAMS=3
run DYNMM_WGRAD_GROUP=3
run DYNMM_WGRAD_GROUP=6
run DYNMM_WGRAD_GROUP_AGE=3
run DYNMM_WGRAD_GROUP_AGE=12
run DYNMM_WGRAD_V6_BLOCKS=384
run DYNMM_WGRAD_V6_BLOCKS=1024
run DYNMM_WGRAD_V6_OCC=3
run A=1
