# alternate two builds of the library on one box: sh scratch/r4/ab_lib.sh <other.so> [reps]
B="python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 30 --warmup 5"
cp dynmm_amd/libdynmm_hip.so /tmp/new.so
R=${2:-3}
for i in $(seq $R); do
  cp /tmp/new.so dynmm_amd/libdynmm_hip.so; echo -n "new: "; $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  cp $1 dynmm_amd/libdynmm_hip.so; echo -n "base: "; $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
cp /tmp/new.so dynmm_amd/libdynmm_hip.so
