mkdir -p gpurun_out
for b in all4 uniform all0; do
  for c in "" "--no-compact"; do
    python bench.py --mode fwd --batch 16 --steps 10 --warmup 3 --branches $b $c --no-cpu-baseline --no-kernel-timing > gpurun_out/fwd_$b$c.log 2>&1
    echo "$b $c: $(tail -1 gpurun_out/fwd_$b$c.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["launch"])')"
  done
done
