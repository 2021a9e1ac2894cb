# usage: sh scratch/r5/ab_cfg.sh VAR a b reps [bench args]: alternate two values of one environment switch on one box
V=$1; A=$2; Bv=$3; R=$4; shift 4
B="python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 20 --warmup 5 $@"
for i in $(seq $R); do for v in $A $Bv; do echo -n "$V=$v: "; env $V=$v $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done; done
