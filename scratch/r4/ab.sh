# usage: sh scratch/r4/ab.sh VAR a b [reps]: alternate two values of one environment switch on one box
B="python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 30 --warmup 5"
R=${4:-3}
for i in $(seq $R); do for v in $2 $3; do echo -n "$1=$v: "; env $1=$v $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done; done
